// filtered_lrelu_mfma.hip -- fused bias -> up-FIR -> gain -> leaky ReLU -> clamp -> down-FIR for float16 /
// bfloat16 tensors on gfx950, with the four separable FIR stages run on the matrix cores.
//
// Semantics: exactly those of filtered_lrelu.hip (reference torch_utils/ops/filtered_lrelu.cu:139-1099,
// filtered_lrelu.cpp:16-210), including the 2-bit sign / clamp mask (write and read with offsets).
//
// Why MFMA: at 16-bit I/O the op moves ~5 bytes per output pixel but needs >= 72 FMAs per output pixel
// (12 + 24 + 24 + 12 for an up-2 / down-2 layer with 12-tap filters); the fp32 vector pipe caps that at about a
// third of the HBM roofline. A zero-insertion FIR is a banded (block-Toeplitz) matrix; as dense 32x32x16 f16
// MFMAs the band wastes ~7x the multiplies but the matrix pipe is 16x faster than the vector pipe.
//
// One workgroup = 4 waves = one 128 x 128 tile of the UP-SAMPLED plane (u = columns, v = rows). With
// X the input tile, A_y / A_x the zero-insertion up-sampling matrices and D_x / D_y the decimating ones:
//
//   stage A  T'[ic][v] = sum_ir X[ir][ic] * A_y[v][ir]     X^T is the A operand (LDS transpose read), A_y^T constant
//   stage B  U^T[u][v] = sum_ic A_x[u][ic] * T'[ic][v]     A_x constant, T' = stage A's accumulators (registers)
//   act      Z^T = clamp(lrelu(U^T)), sign / clamp mask    4 consecutive u per lane = one mask byte
//   stage C  W[ox][v]  = sum_u  D_x[ox][u] * Z^T[u][v]     D_x constant, Z^T from registers
//            W -> LDS as [v][ox]                            (the only intermediate that leaves the registers)
//   stage D  Y[oy][ox] = sum_v  D_y[oy][v] * W[v][ox]      W via LDS transpose read; lanes = ox: row-contiguous stores
//
// Every stage is a LEFT multiplication of the running matrix, whose 32x32 MFMA result layout (lane = column,
// registers = rows) is already the B-operand layout of the next MFMA up to a fixed permutation of k inside each
// 16-row chunk -- the constant operand is stored with the same permutation, so stages A -> B -> C chain through
// registers. Wave w owns rows v in [32w, 32w + 32) of the up-sampled tile through stages A - C; stage D splits the
// output blocks over the waves. Two barriers per tile. The constant operands ("fragments": 32 x 16 slices of
// the banded matrices, one per distinct band offset) are built once per workgroup in LDS; workgroups are
// persistent and walk over tiles.
//
// Arithmetic: f16 operands (bfloat16 tensors are converted: their 8-bit mantissa is exact in f16), f32
// accumulation; T', Z and W are rounded to f16 between stages (one rounding each, like the reference's
// non-fused path which rounds after every op); filter taps are rounded to f16.
// Algorithmic HBM bytes: (N_in + N_out) * 2 + mask bytes; see DESIGN.md.

#include "lvg_common.h"
#include "filtered_lrelu_args.h"

#ifdef LVG_MARKERS     // analysis builds only (tools/isa_count.py --regions): region labels in the listing
#define LVG_MARK(name) asm volatile("; LVGMARK " name)
#else
#define LVG_MARK(name)
#endif
// Analysis builds only (-DLVG_TIMING; tools/flrelu_check prints the table): shader-clock cycles each wave spends per region of the
// tile loop, read with s_memtime at the region boundaries and left in g_flreluTiming[(workgroup * 4 + wave) * 16 + region].
#ifdef LVG_TIMING
__device__ uint32_t g_flreluTiming[4096 * 16];
#define LVG_TICK(idx) { const uint32_t now_ = (uint32_t)__builtin_readcyclecounter(); tAcc[idx] += now_ - tLast; tLast = now_; }
extern "C" int lvg_flrelu_timing_read(uint32_t* host, int count)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_flreluTiming), (size_t)count * 4, 0, hipMemcpyDeviceToHost);
}
#else
#define LVG_TICK(idx)
#endif

// Tuning switches (A/B measured with tools/flrelu_check):
#ifndef LVG_ABL
#define LVG_ABL 0            // ablation builds only (timing experiments; results are WRONG): 1 no prefetch, 2 no y stores, 4 no activation math, 8 no stage D, 16 all y stores to plane 0 tile 0, 32 all x loads from plane 0
#endif
#ifndef LVG_MFMA_PIPELINE
#define LVG_MFMA_PIPELINE 1      // issue stage B of block b + 1 before the activation of block b
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;   // 4 waves
constexpr int kUpT = 128;       // edge of the up-sampled tile (4 blocks of 32)

constexpr int mdiv_up(int a, int b) { return (a + b - 1) / b; }

template <int UP, int DOWN, int FU, int FD, int TW, int TH, int MODE>
struct MG
{
    static constexpr int KU     = FU / UP;                                  // taps per output of an up stage
    static constexpr int IN_N   = (UP - 1 + kUpT - 1) / UP + KU;            // input rows / columns a tile touches
    static constexpr int IN_BLK = mdiv_up(IN_N, 32);                        // 32-blocks of input columns (stage A's M)
    static constexpr int IN_CH  = ((((kUpT - 1 + UP - 1) / UP) + KU - 1) >> 4) + 1;   // 16-chunks of input rows / columns
    static constexpr int X_ROWS = IN_CH * 16;
    static constexpr int SX     = 80;                                       // X row stride (halves). Narrower than the 96 columns stage A's transpose reads span for
                                                                            // UP = 2: columns 80..95 of a row alias the next row's first 16 (finite data); they only
                                                                            // feed rows 80..95 of T', which nothing reads.
    static constexpr int OBX    = mdiv_up(TW, 32);
    static constexpr int OBY    = mdiv_up(TH, 32);
    static constexpr int SW     = (OBX == 2) ? 60 : OBX * 32 + 4;           // W row stride (halves): 8-byte column writes of 16 rows hit 32 distinct banks; with two
                                                                            // 32-column blocks only 60 columns are kept (TW <= 56; the last 4 would be discarded anyway)
    static constexpr int NUC    = (UP == 2) ? 2 : (UP == 4 ? 3 : 1);        // distinct band offsets of an up stage
    static constexpr int NDC    = ((31 * DOWN + FD - 1 + 3) >> 4) + 1;      // ... of a down stage (+3: READ-mode column shift)
    static constexpr int IMG_FY = 0;                                        // fragment images: A_y natural k order (stage A)
    static constexpr int IMG_FX = NUC;                                      //                  A_x permuted k order, scaled (stage B)
    static constexpr int IMG_DX = 2 * NUC;                                  //                  D_x permuted (stage C)
    // D_y (stage D, natural k order) equals D_x except for the READ-mode column shift: without it stage D reads the D_x
    // images with two 8-byte reads per lane (lds_frag_nat_from_perm) and no separate images exist.
    static constexpr bool SHARE_D = MODE != LVG_SIGNS_READ;
    static constexpr int IMG_DY = SHARE_D ? IMG_DX : 2 * NUC + NDC;         //                  D_y natural (stage D)
    static constexpr int NIMG   = 2 * NUC + (SHARE_D ? 1 : 2) * NDC;
    static constexpr bool HAS_M = MODE != LVG_SIGNS_NONE;                   // mask tile in LDS (assembling the dwords in registers measured slower)
    // Register budget: 4 waves per SIMD (128 VGPRs) where the kernel fits without spilling (measured with
    // -Rpass-analysis=kernel-resource-usage), else 3 (168 VGPRs).
    static constexpr int WAVES  = (MODE == LVG_SIGNS_NONE || UP == 4 || (MODE == LVG_SIGNS_READ && DOWN == 4)) ? 4 : 3;
    static constexpr int TAPS   = (FU + FD + 3) / 4 * 4;
    // LDS map (bytes)
    static constexpr int OFF_TAPS = 0;
    static constexpr int OFF_LUT  = TAPS * 4;                               // 16 dwords (READ mode)
    static constexpr int OFF_BIAS = OFF_LUT + 64;                           // bias bits of the first 64 planes of this workgroup's range
    static constexpr int OFF_TMAX = OFF_BIAS + 256;                         // 2 dwords: max |x + bias| (f16 bits) of the tile in XL / of the tile being written
    static constexpr int OFF_TAB  = OFF_TMAX + 16;
    static constexpr int OFF_X    = OFF_TAB + NIMG * 1024;
    static constexpr int OFF_W    = OFF_X + X_ROWS * SX * 2 + 64;           // (+64: the aliased reads of the last row stay inside the allocation)
    static constexpr int OFF_M    = OFF_W + kUpT * SW * 2 + 16;
    static constexpr int SM       = 36;                                       // mask tile row stride (bytes): 9 dwords, lanes = rows hit distinct banks
    static constexpr int LDS_BYTES = OFF_M + (HAS_M ? kUpT * SM : 0);
    static_assert(FU % UP == 0 && FD % DOWN == 0, "filter sizes must be multiples of the rates");
    static_assert(IN_N % 2 == 0 && IN_N <= SX && IN_CH * 16 <= SX, "input tile geometry");
    static_assert((TW * DOWN) % 4 == 0 && (TW * DOWN) % UP == 0 && (TH * DOWN) % UP == 0, "tile origin must keep the mask byte and the up-sampling phase fixed");
    static_assert((TW - 1) * DOWN + FD - 1 + 3 < kUpT && (TH - 1) * DOWN + FD - 1 < kUpT, "tile does not fit the 128 x 128 up-sampled block");
    static_assert(OBX * OBY <= 4, "stage D: one output block per wave");
    static_assert(LDS_BYTES <= 64 * 1024, "LDS budget");
    static constexpr int CU_WGS = (160 * 1024 / LDS_BYTES) < WAVES ? (160 * 1024 / LDS_BYTES) : WAVES;   // resident workgroups per CU
    static_assert(OFF_TAB % 16 == 0 && OFF_X % 16 == 0 && OFF_W % 16 == 0 && OFF_M % 16 == 0, "alignment");
};

// Offset (in input samples, relative to the first input sample of a 32-output block) of up-stage class `cls`.
template <int UP> __host__ __device__ constexpr int up_class_offset(int cls) { return UP == 2 ? 16 * cls : (UP == 4 ? 8 * cls - 8 : 0); }

// 16-chunks of the input that output block b of an up stage needs: first chunk, count, class of the first, class step.
template <int UP> struct UpChunks
{
    __host__ __device__ static constexpr int first(int b)  { return UP == 2 ? b : (UP == 4 ? ((b & 1) ? (b - 1) / 2 : b / 2) : 2 * b); }
    __host__ __device__ static constexpr int count(int b)  { return UP == 2 ? 2 : (UP == 4 ? ((b & 1) ? 2 : 1) : 2); }
    __host__ __device__ static constexpr int cls0(int b)   { return UP == 2 ? 0 : (UP == 4 ? ((b & 1) ? 0 : 1) : 0); }
    __host__ __device__ static constexpr int step()        { return UP == 4 ? 2 : 1; }
};

__device__ __forceinline__ half8 lds_frag(const _Float16* tab, int img, int lane)
{
    return *reinterpret_cast<const half8*>(tab + img * 512 + lane * 8);
}

// The natural-order fragment (k = 8 g + j) out of a PERMUTED image: its first four k live in the image of lane (row, g' = 0),
// halves 4 g .. 4 g + 3, the last four in lane (row, g' = 1) at the same halves.
__device__ __forceinline__ half8 lds_frag_nat_from_perm(const _Float16* tab, int img, int lane)
{
    const int row = lane & 31, g = lane >> 5;
    const half4 lo = *reinterpret_cast<const half4*>(tab + img * 512 + row * 8 + 4 * g);
    const half4 hi = *reinterpret_cast<const half4*>(tab + img * 512 + (32 + row) * 8 + 4 * g);
    half8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &hi, 8);
    return r;
}

// MFMA operand (lane: index = lane & 31 along the COLUMNS of a row-major LDS matrix, k = 8 * (lane >> 5) + j along
// its ROWS) through the gfx950 transpose read: 16 lanes fetch a 4 (rows) x 16 (columns) block, lane s supplying
// the address of row (s >> 2), columns 4 * (s & 3) .. + 3, and lane l receiving column l of the 4 rows
// (measured: tools/probe_mfma_layout.hip).
__device__ __forceinline__ half8 lds_tr_operand(const _Float16* base, int stride, int row0, int col0, int lane)
{
    const int g = lane >> 5, hgrp = (lane >> 4) & 1, s = lane & 15;
    const _Float16* p = base + (row0 + 8 * g + (s >> 2)) * stride + col0 + 16 * hgrp + 4 * (s & 3);
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * stride));
    half8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &hi, 8);
    return r;
}

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// Rows 16 * h .. 16 * h + 15 of a 32x32 result as the B operand of the next MFMA (k order: see the header).
__device__ __forceinline__ half8 pack_chunk(const f32x16& c, int h)
{
    half8 r;
    #pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (_Float16)c[8 * h + j];
    return r;
}

__device__ __forceinline__ uint32_t h2_bits(half2v v) { uint32_t u; __builtin_memcpy(&u, &v, 4); return u; }
__device__ __forceinline__ half2v bits_h2(uint32_t u) { half2v v; __builtin_memcpy(&v, &u, 4); return v; }
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// One 16-bit element at a wave-uniform address through the scalar cache (the aligned dword that holds it).
__device__ __forceinline__ uint32_t scalar_load_u16(const uint16_t* ptr)
{
    const uint64_t a = (uint64_t)(uintptr_t)ptr;
    const uint64_t a4 = a & ~(uint64_t)3;
    uint32_t wd;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wd) : "s"(a4) : "memory");
    return (a & 2) ? (wd >> 16) : (wd & 0xffffu);
}

// Two stored elements (element 0 in the low half of the dword) + bias -> f16 pair. bfloat16 values beyond the
// f16 range saturate instead of turning into inf (inf * a zero tap of the banded matrix would be NaN).
template <class T> __device__ __forceinline__ half2v pair_plus_bias(uint32_t raw, half2v bias2, float bias);
template <> __device__ __forceinline__ half2v pair_plus_bias<f16_t>(uint32_t raw, half2v bias2, float) { return bits_h2(raw) + bias2; }
template <> __device__ __forceinline__ half2v pair_plus_bias<bf16_t>(uint32_t raw, half2v, float bias)
{
    const float a = __uint_as_float(raw << 16) + bias, b = __uint_as_float(raw & 0xffff0000u) + bias;
    half2v r;
    r[0] = (_Float16)__builtin_fminf(__builtin_fmaxf(a, -65504.0f), 65504.0f);
    r[1] = (_Float16)__builtin_fminf(__builtin_fmaxf(b, -65504.0f), 65504.0f);
    return r;
}

typedef short short2v __attribute__((ext_vector_type(2)));

// Activation of one 32 x 32 block of U^T held as an MFMA result (register r = pixel u = (r & 3) + 8 (r >> 2) + 4 g of
// this lane's row v), in packed f16: leaky ReLU, clamp, and the 2-bit mask codes (1 = negative, 2 = clamped).
// Registers 4q .. 4q + 3 are the four pixels of one mask byte (mbytes[2 q]). Result: the block as 8 packed dwords =
// the two B-operand chunks of the next MFMA. READ mode multiplies by (1, slope, 0) looked up from the stored codes.
// CLAMP = false is used for tiles whose input is provably too small to reach the clamp (see `xLimitBits` in the kernel):
// the clamp and its flag arithmetic are then no-ops and are skipped.
// The file is compiled with -fno-honor-nans (no canonicalisation ops around min / max): a NaN pre-activation
// comes out as -clamp instead of NaN.
template <int MODE, bool SLOPEMAX, bool CLAMP>
__device__ __forceinline__ void act_block(const f32x16& accU, uint32_t (&zp)[8], uint8_t* mbytes, const uint32_t* lut,
                                          half2v slope2, half2v clampP, half2v clampN, uint32_t clampBits)
{
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        half2v P[2];
        #pragma unroll
        for (int h = 0; h < 2; h++) { P[h][0] = (_Float16)accU[4 * q + 2 * h]; P[h][1] = (_Float16)accU[4 * q + 2 * h + 1]; }
        if (MODE == LVG_SIGNS_READ)
        {
            const uint32_t bits = mbytes[2 * q];
            #pragma unroll
            for (int h = 0; h < 2; h++) zp[2 * q + h] = h2_bits(P[h] * bits_h2(lut[(bits >> (4 * h)) & 15u]));
        }
        else
        {
            half2v L[2];
            #pragma unroll
            for (int h = 0; h < 2; h++)
            {
                const half2v ls = P[h] * slope2;
                if (SLOPEMAX) L[h] = __builtin_elementwise_max(P[h], ls);    // 0 <= slope <= 1
                else
                {
                    short2v pi, li;
                    __builtin_memcpy(&pi, &P[h], 4); __builtin_memcpy(&li, &ls, 4);
                    const short2v m = pi >> 15;                              // all ones where the half is negative (sign bit: -0.0 counts)
                    const short2v r = (li & m) | (pi & ~m);
                    __builtin_memcpy(&L[h], &r, 4);
                }
            }
            if (MODE == LVG_SIGNS_WRITE)
            {
                // bytes 1 and 3 of each pair carry the sign bits; "clamped" = sign bit of (clamp - |L|) as 16-bit integers
                const uint32_t S = __builtin_amdgcn_perm(h2_bits(P[1]), h2_bits(P[0]), 0x07050301u);
                uint32_t x = (S >> 7) & 0x01010101u;
                if (CLAMP)
                {
                    uint32_t Tb[2];
                    #pragma unroll
                    for (int h = 0; h < 2; h++)
                    {
                        const uint32_t a = h2_bits(L[h]) & 0x7fff7fffu;
                        short2v cv, av; __builtin_memcpy(&cv, &clampBits, 4); __builtin_memcpy(&av, &a, 4);
                        const short2v d = cv - av;
                        __builtin_memcpy(&Tb[h], &d, 4);
                    }
                    const uint32_t C = __builtin_amdgcn_perm(Tb[1], Tb[0], 0x07050301u);
                    x = (((S & ~C) >> 7) & 0x01010101u) | ((C >> 6) & 0x02020202u);   // code 2 replaces the sign bit
                }
                uint32_t y = x | (x >> 6);
                y = y | (y >> 12);
                mbytes[2 * q] = (uint8_t)y;
            }
            #pragma unroll
            for (int h = 0; h < 2; h++)
                zp[2 * q + h] = CLAMP ? h2_bits(__builtin_elementwise_min(__builtin_elementwise_max(L[h], clampN), clampP)) : h2_bits(L[h]);
        }
    }
}

struct TileCoord { int tileX, tileY, ch, nb, plane; };   // plane = nb * channels + ch

template <class T, int UP, int DOWN, int FU, int FD, int TW, int TH, int MODE, bool FASTLOAD>
__global__ __launch_bounds__(kThreads, (MG<UP, DOWN, FU, FD, TW, TH, MODE>::WAVES)) void filtered_lrelu_mfma_kernel(FlreluArgs p, int totalTiles)
{
    typedef MG<UP, DOWN, FU, FD, TW, TH, MODE> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float*     taps = reinterpret_cast<float*>(smem + G::OFF_TAPS);        // [0, FU): up taps, [FU, FU + FD): down taps (flipped)
    uint32_t*  lut  = reinterpret_cast<uint32_t*>(smem + G::OFF_LUT);      // READ mode: mask nibble -> pair of gradient factors
    uint32_t*  biasL = reinterpret_cast<uint32_t*>(smem + G::OFF_BIAS);    // storage bits of b[channel] for the planes this workgroup visits
    uint32_t*  tmax = reinterpret_cast<uint32_t*>(smem + G::OFF_TMAX);     // see xLimitBits
    _Float16*  tab  = reinterpret_cast<_Float16*>(smem + G::OFF_TAB);      // fragment images, 512 halves each, lane-major
    _Float16*  XL   = reinterpret_cast<_Float16*>(smem + G::OFF_X);        // input tile + bias [X_ROWS][SX]
    _Float16*  WL   = reinterpret_cast<_Float16*>(smem + G::OFF_W);        // W [128 v][SW]
    uint8_t*   ML   = smem + G::OFF_M;                                     // mask tile [128 v][SM bytes, 32 used] (not with MODE NONE)
    const int tid = threadIdx.x, lane = tid & 63, w = sgpr(tid >> 6);
    const int n = lane & 31, g = lane >> 5;

    // ---- this workgroup's contiguous range of tiles; tile -> (tileX, tileY, channel, sample) ----------------
    const int tileBeg = (int)((int64_t)totalTiles * blockIdx.x / gridDim.x);
    const int tileEnd = (int)((int64_t)totalTiles * (blockIdx.x + 1) / gridDim.x);
    TileCoord cur;
    {
        int bid = tileBeg;
        cur.tileX = bid % p.tilesX; bid /= p.tilesX;
        cur.tileY = bid % p.tilesY; bid /= p.tilesY;
        cur.plane = bid;
        cur.ch = bid % p.c; cur.nb = bid / p.c;
        cur.tileX = sgpr(cur.tileX); cur.tileY = sgpr(cur.tileY); cur.ch = sgpr(cur.ch); cur.nb = sgpr(cur.nb); cur.plane = sgpr(cur.plane);
    }
    const int planeBeg = cur.plane;
    if (tid < 64) biasL[tid] = ((const uint16_t*)p.b)[(planeBeg + tid) % p.c];
    if (tid < 2) tmax[tid] = 0u;

    // ---- once per workgroup: taps, fragment images, zero the padding of the input tile -------------------
    if (tid < FU)
    {
        float v = 0.0f;
        if (tid < p.fuN) v = p.fu ? p.fu[p.flip ? tid : p.fuN - 1 - tid] : 1.0f;
        taps[tid] = v;
    }
    else if (tid < FU + FD)
    {
        const int t = tid - FU;
        float v = 0.0f;
        if (t < p.fdN) v = p.fd ? p.fd[p.flip ? t : p.fdN - 1 - t] : 1.0f;
        taps[FU + t] = v;
    }
    for (int i = tid; i < (G::X_ROWS * G::SX + 32) / 2; i += kThreads) reinterpret_cast<uint32_t*>(XL)[i] = 0u;
    if (MODE == LVG_SIGNS_READ && tid < 16)
    {
        // mask codes of two neighbouring pixels -> (factor of pixel 0, factor of pixel 1): 0 -> 1, 1 -> slope, 2 / 3 -> 0
        half2v f;
        const int c0 = tid & 3, c1 = tid >> 2;
        f[0] = (_Float16)(c0 == 0 ? 1.0f : (c0 == 1 ? p.slope : 0.0f));
        f[1] = (_Float16)(c1 == 0 ? 1.0f : (c1 == 1 ? p.slope : 0.0f));
        lut[tid] = h2_bits(f);
    }
    __syncthreads();

    // Launch-constant geometry: the column shift that aligns the tile with the mask bytes in READ mode and the
    // zero-insertion phases (tile origins are multiples of 4 and of UP in the up-sampled plane).
    const int rOff = (MODE == LVG_SIGNS_READ) ? (p.sOfsX & 3) : 0;
    const int phX = ((UP - 1 - p.px0 - rOff) % UP + UP) % UP;
    const int phY = ((UP - 1 - p.py0) % UP + UP) % UP;
    {
        const float scale = (float)(UP * UP) * p.gain;
        for (int e = tid; e < G::NIMG * 512; e += kThreads)
        {
            const int img = e >> 9, idx = e & 511, L = idx >> 3, j = idx & 7, row = L & 31, gg = L >> 5;
            const bool perm = img >= G::IMG_FX && img < G::IMG_DX + G::NDC;
            const int k = perm ? ((j & 3) + 8 * (j >> 2) + 4 * gg) : (8 * gg + j);
            float v = 0.0f;
            if (img < G::IMG_DX)
            {
                // up stage: output row (local u' or v'), input sample kk relative to the block's first input sample
                const bool isX = img >= G::IMG_FX;
                const int ph = isX ? phX : phY;
                const int kk = up_class_offset<UP>(isX ? img - G::IMG_FX : img) + k;
                const int m = row + ph, i0 = m / UP, t = kk - i0;
                if (t >= 0 && t < G::KU) v = taps[(UP - 1 - m % UP) + t * UP] * (isX ? scale : 1.0f);
            }
            else
            {
                const bool isX = img < G::IMG_DX + G::NDC;
                const int cls = isX ? img - G::IMG_DX : img - (G::IMG_DX + G::NDC);
                const int t = 16 * cls + k - (isX ? rOff : 0) - row * DOWN;
                if (t >= 0 && t < FD) v = taps[FU + t];
            }
            tab[e] = (_Float16)v;
        }
    }
    // (barrier 1 of the first tile publishes the table)

    // ---- activation constants (packed f16) -----------------------------------------------------------------
    const _Float16 slope_h = (_Float16)p.slope;
    const half2v slope2 = {slope_h, slope_h};
    const bool slopeMax = p.slope <= 1.0f;                                  // lrelu(x) = max(x, slope * x) (slope >= 0 is asserted by the caller)
    const _Float16 clamp_h = (_Float16)(p.clamp < 65504.0f ? p.clamp : 65504.0f);    // no clamp = the largest finite f16
    const half2v clampP = {clamp_h, clamp_h}, clampN = {-clamp_h, -clamp_h};
    const uint32_t clampBits = h2_bits(clampP);
    // No pre-activation of a tile can exceed  scale * l1(up taps per phase)^2 * max |x + bias|  in magnitude (and leaky
    // ReLU with slope <= 1 only shrinks it), so tiles whose input maximum stays below clamp / that factor (5 % margin
    // for the f16 roundings) skip the clamp and the "clamped" flag arithmetic: xLimitBits = that threshold as f16 bits
    // (positive f16 numbers order like their bit patterns; inf / NaN inputs compare above every finite threshold).
    uint32_t xLimitBits = 0;
    {
        float l1 = 0.0f;
        for (int ph = 0; ph < UP; ph++)
        {
            float a = 0.0f;
            for (int k = ph; k < FU; k += UP) a += fabsf(taps[k]);
            l1 = fmaxf(l1, a);
        }
        const float lim = p.clamp / ((float)(UP * UP) * p.gain * l1 * l1 * 1.05f + 1e-30f);
        const _Float16 lh = (_Float16)fminf(lim, 60000.0f);
        uint16_t lb; __builtin_memcpy(&lb, &lh, 2);
        xLimitBits = ((float)lh <= lim && lb > 0) ? lb : (lb > 0 ? lb - 1u : 0u);      // round down
        if (!(p.slope <= 1.0f)) xLimitBits = 0;
    }

    // All per-lane address arithmetic is done ONCE here as 32-bit byte offsets from a per-tile scalar base pointer
    // (the launcher checks that a plane spans < 2^31 bytes), so a load / store in the tile loop is
    // "scalar base + lane offset" with no 64-bit vector math.
    // Input loader: thread = (row r0 of RPP, column pair qp); pass i handles row r0 + RPP * i.
    constexpr int PAIRS = G::IN_N / 2, RPP = kThreads / PAIRS, NPASS = mdiv_up(G::IN_N, RPP);
    const int ld_r0 = tid / PAIRS, ld_qp = tid - ld_r0 * PAIRS;
    const bool ld_active = tid < RPP * PAIRS;
    const int ld_lds0 = ld_r0 * G::SX + 2 * ld_qp;                          // halves
    const uint32_t ld_off0 = (uint32_t)(ld_r0 * (int)p.xs[2] + 2 * ld_qp * (int)p.xs[3]) * 2u;   // bytes from the tile's first input pixel
    const uint32_t ld_pass = (uint32_t)(RPP * (int)p.xs[2]) * 2u;
    const uint32_t ld_x1 = (uint32_t)((int)p.xs[3]) * 2u;
    uint32_t raw[NPASS];                                                    // prefetched pairs of the NEXT tile (storage bits)
    uint32_t mraw[5];                                                       // READ mode: prefetched (aligned) mask dwords of the next tile
    int mshiftN = 0, mvalidN = 0;                                           // ... their byte shift (uniform) and this thread's count of valid bytes
    uint32_t mokN = 0;                                                      // ... bit j: dword j lies inside the mask plane
    float biasN = 0.0f;
    // Stage D: this wave's output block and this lane's column.
    const int dBy = w / G::OBX, dBx = w - dBy * G::OBX;
    const uint32_t st_off0 = (uint32_t)(n * (int)p.ys[2] + (32 * dBx + 4 * g) * (int)p.ys[3]) * 2u;   // row n of the block, first of this lane's columns
    const uint32_t st_x1 = (uint32_t)((int)p.ys[3]) * 2u;
    // four consecutive outputs go out as one 8-byte store when they are contiguous and dword aligned
    const bool fastStore = p.ys[3] == 1 && (p.ys[2] & 1) == 0 && (p.ys[1] & 1) == 0 && (p.ys[0] & 1) == 0 && (((uintptr_t)p.y) & 3u) == 0;

    // Loads are UNCONDITIONAL from offsets clamped into the plane (no execution-mask juggling, no per-pass scalar
    // control flow); which of them are real pixels is decided when the tile is written to LDS: rows by one unsigned
    // compare per pass, columns by poisoning the row counter of lanes whose column pair lies outside the image.
    uint32_t ldRowN = 0;                                                    // next tile: this lane's first row minus the first valid row (poisoned: never valid)
    uint32_t ldColN = 0;                                                    // next tile, slow path: bit 0 / 1 = column ix / ix + 1 inside the image
    int ldSpanN = 0;                                                        // next tile: number of valid rows
    uint32_t negbN = 0;                                                     // next tile: (-bias, -bias) in storage bits
    auto issue_loads = [&](const TileCoord& tc)
    {
        const int uStart = tc.tileX * (TW * DOWN) - rOff, upY0 = tc.tileY * (TH * DOWN);
        const int inX0 = lvg_floor_div(uStart + UP - 1 - p.px0, UP);
        const int inY0 = lvg_floor_div(upY0 + UP - 1 - p.py0, UP);
        // "scalar plane base + 32-bit lane offset" addressing: the tile term keeps the sum inside this block, so the
        // compiler cannot hoist ten 64-bit lane addresses out of the tile loop (it did: +20 VGPRs and spills)
        const char* xpl = (const char*)((const T*)p.x + ((LVG_ABL & 32) ? 0 : ((int64_t)tc.nb * p.xs[0] + (int64_t)tc.ch * p.xs[1])));
        const int bi = tc.plane - planeBeg;
        // (planes beyond the 64 of the LDS table: a SCALAR load. A vector load here would put a vector-memory wait between the
        // prefetch loads below and their use a tile later -- the counter is in-order -- and stall every tile for a full memory latency)
        const uint32_t bb = bi < 64 ? biasL[bi] : scalar_load_u16((const uint16_t*)p.b + tc.ch);
        { T bt; const uint16_t b16 = (uint16_t)bb; __builtin_memcpy(&bt, &b16, 2); biasN = (float)to_acc(bt); }
        negbN = (bb ^ 0x8000u) * 0x10001u;                                    // (-bias, -bias): + bias = 0 outside the image
        const int rLo = max(0, -inY0);
        ldSpanN = max(0, min(G::IN_N, p.xh - inY0) - rLo);
        const int ix = inX0 + 2 * ld_qp;
        const bool c0 = (uint32_t)ix < (uint32_t)p.xw, c1 = (uint32_t)(ix + 1) < (uint32_t)p.xw;
        ldColN = (c0 ? 1u : 0u) | (c1 ? 2u : 0u);
        ldRowN = (FASTLOAD ? c0 : (c0 || c1)) ? (uint32_t)(ld_r0 - rLo) : 0x40000000u;
        // last byte offset a load may start at (the wrap of a negative offset is far above it)
        const uint32_t lastEl = (uint32_t)((p.xh - 1) * (int)p.xs[2] + (p.xw - 1) * (int)p.xs[3]) * 2u;
        uint32_t off = (uint32_t)(inY0 * (int)p.xs[2] + inX0 * (int)p.xs[3]) * 2u + ld_off0;
        #pragma unroll
        for (int i = 0; i < NPASS; i++)
        {
            if (FASTLOAD) raw[i] = *reinterpret_cast<const uint32_t*>(xpl + min(off, lastEl - 2u));   // pairs are dword aligned, never straddle the edge
            else
            {
                const uint32_t lo = *reinterpret_cast<const uint16_t*>(xpl + min(off, lastEl));
                const uint32_t hi = *reinterpret_cast<const uint16_t*>(xpl + min(off + ld_x1, lastEl));
                raw[i] = lo | (hi << 16);
            }
            off += ld_pass;
        }
        if (MODE == LVG_SIGNS_READ)
        {
            // 16 bytes of one mask row per thread, fetched as the 5 aligned dwords that cover them (rows of the mask
            // plane are dword aligned, the tile's first byte is not); write_tile() shifts them into place.
            const int row = tid >> 1, half = tid & 1;
            const int signByte0 = (uStart + p.sOfsX) >> 2;
            const int sy = upY0 + p.sOfsY + row;
            const bool rowOk = (uint32_t)sy < (uint32_t)p.sH;
            const uint8_t* spl = p.s + (int64_t)tc.plane * ((int64_t)p.sH * p.sWBytes);
            const int b0 = signByte0 + 16 * half, a0 = b0 & ~3;
            const uint32_t rowOff = (uint32_t)(sy * p.sWBytes);
            mshiftN = signByte0 & 3;
            mvalidN = p.swLimit - b0;                                        // bytes of this thread's 16 that carry pixels (may be <= 0 or >= 16)
            // unconditional loads like the pixels': dwords outside the plane are fetched from the plane's first dword and zeroed
            // when the tile is written (no execution-mask branches, no vector-memory waits here)
            mokN = 0;
            #pragma unroll
            for (int j = 0; j < 5; j++)
            {
                const int bx = a0 + 4 * j;
                const bool ok = rowOk && bx >= 0 && bx + 4 <= p.sWBytes;
                mokN |= ok ? (1u << j) : 0u;
                mraw[j] = *reinterpret_cast<const uint32_t*>(spl + (ok ? rowOff + (uint32_t)bx : 0u));
            }
        }
    };
    int slotW = 0;                                                          // tmax slot of the tile being written (alternates)
    auto write_tile = [&]()
    {
        const _Float16 bh = (_Float16)biasN;
        const half2v bias2 = {bh, bh};
        uint32_t mx2 = 0;
        if (ld_active)
        {
            #pragma unroll
            for (int i = 0; i < NPASS; i++)
            {
                const bool rowOk = (ldRowN + (uint32_t)(RPP * i)) < (uint32_t)ldSpanN;
                uint32_t v = rowOk ? raw[i] : negbN;
                if (!FASTLOAD)                                                // one of the two columns may lie outside the image
                    v = (v & ((ldColN & 1u) ? 0xffffu : 0u)) | (v & ((ldColN & 2u) ? 0xffff0000u : 0u)) |
                        (negbN & (((ldColN & 1u) ? 0u : 0xffffu) | ((ldColN & 2u) ? 0u : 0xffff0000u)));
                const half2v hv = pair_plus_bias<T>(v, bias2, biasN);
                *reinterpret_cast<half2v*>(XL + ld_lds0 + RPP * i * G::SX) = hv;
                if (MODE != LVG_SIGNS_READ)
                {
                    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
                    const uint32_t ab = h2_bits(hv) & 0x7fff7fffu;
                    ushort2v am, cm; __builtin_memcpy(&am, &ab, 4); __builtin_memcpy(&cm, &mx2, 4);
                    cm = __builtin_elementwise_max(cm, am);
                    __builtin_memcpy(&mx2, &cm, 4);
                }
            }
        }
        if (MODE != LVG_SIGNS_READ)
        {
            // wave maximum (DPP butterfly inside rows of 16, then row broadcasts), one LDS atomic per wave
            uint32_t m = max(mx2 & 0xffffu, mx2 >> 16);
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, false));    // row_half_mirror
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, false));    // row_mirror
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x142, 0xa, 0xf, false));    // row_bcast15 -> rows 1, 3
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x143, 0xc, 0xf, false));    // row_bcast31 -> rows 2, 3
            if (lane == 63) atomicMax(&tmax[slotW], m);
        }
        if (MODE == LVG_SIGNS_READ)
        {
            uint32_t* m = reinterpret_cast<uint32_t*>(ML + (tid >> 1) * G::SM + 16 * (tid & 1));
            #pragma unroll
            for (int d = 0; d < 4; d++)
            {
                const uint32_t lo = (mokN >> d) & 1u ? mraw[d] : 0u, hi = (mokN >> (d + 1)) & 1u ? mraw[d + 1] : 0u;
                uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)mshiftN);
                const int nv = mvalidN - 4 * d;                              // bytes at and beyond swLimit carry no pixels
                if (nv < 4) v = nv <= 0 ? 0u : (v & ((1u << (8 * nv)) - 1u));
                m[d] = v;
            }
        }
    };
    // One 32 x 32 block of U^T = A_x * T' (stage B) for column block b of the up-sampled tile.
    auto stage_b = [&](int b, const half8 (&tpk)[G::IN_BLK][2]) -> f32x16
    {
        f32x16 acc;
        #pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        #pragma unroll
        for (int t = 0; t < 2; t++)
        {
            if (t < UpChunks<UP>::count(b))
            {
                const int c = UpChunks<UP>::first(b) + t;
                const half8 fx = lds_frag(tab, G::IMG_FX + UpChunks<UP>::cls0(b) + t * UpChunks<UP>::step(), lane);
                acc = mfma(fx, tpk[c >> 1][c & 1], acc);
            }
        }
        return acc;
    };

    // Stage D of one tile: Y^T[ox][oy] = W^T * D_y^T from WL (all waves' rows), one 32 x 32 output block per wave.
    // Lanes = 32 output rows, registers 4q .. 4q + 3 = four consecutive ox -> 8-byte stores (16-byte stores after a half-wave
    // exchange, and 16 rows x 64 contiguous bytes per instruction through an LDS transpose, both measured SLOWER:
    // profiles/r03_flrelu_store_ab.log -- the stores are not issue-bound). `store` = false computes without writing (first trip
    // of the pipelined loop: WL holds nothing yet).
    auto stage_d = [&](const TileCoord& tc, bool store)
    {
        if (w < G::OBX * G::OBY && !(LVG_ABL & 8))
        {
            const int outX0 = tc.tileX * TW, outY0 = tc.tileY * TH;
            f32x16 accY;
            #pragma unroll
            for (int r = 0; r < 16; r++) accY[r] = 0.0f;
            #pragma unroll
            for (int cls = 0; cls < G::NDC; cls++)
            {
                const int c = 2 * dBy * DOWN + cls;
                if (c < 8)
                {
                    const half8 fd = G::SHARE_D ? lds_frag_nat_from_perm(tab, G::IMG_DX + cls, lane) : lds_frag(tab, G::IMG_DY + cls, lane);
                    const half8 wt = lds_tr_operand(WL, G::SW, 16 * c, 32 * dBx, lane);
                    accY = mfma(wt, fd, accY);
                }
            }
            const int rowsHere = min(TH, p.yh - outY0) - 32 * dBy;           // rows of this wave's block that exist (uniform)
            const int colRoom = min(TW, p.yw - outX0) - 32 * dBx - 4 * g;    // columns from this lane's first one that exist
            if (store && n < rowsHere && !(LVG_ABL & 2))
            {
                char* ypl = (char*)((T*)p.y + ((LVG_ABL & 16) ? 0 : ((int64_t)tc.nb * p.ys[0] + (int64_t)tc.ch * p.ys[1])));
                const uint32_t yoff = (uint32_t)((((LVG_ABL & 16) ? 0 : outY0) + 32 * dBy) * (int)p.ys[2] + ((LVG_ABL & 16) ? 0 : outX0) * (int)p.ys[3]) * 2u + st_off0;
                #pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    if (fastStore && 8 * q + 4 <= colRoom)
                    {
                        T t4[4];
                        #pragma unroll
                        for (int e = 0; e < 4; e++) t4[e] = from_acc<T>(accY[4 * q + e]);
                        uint2 v; __builtin_memcpy(&v, t4, 8);
                        *reinterpret_cast<uint2*>(ypl + (yoff + 16u * q)) = v;
                    }
                    else
                    {
                        #pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (8 * q + e < colRoom) *reinterpret_cast<T*>(ypl + (yoff + (uint32_t)(8 * q + e) * st_x1)) = from_acc<T>(accY[4 * q + e]);
                    }
                }
            }
        }
    };

    if (tileBeg < tileEnd) { issue_loads(cur); write_tile(); }
    TileCoord prv = cur;

    LVG_MARK("loop");
#ifdef LVG_TIMING
    uint32_t tAcc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t tLast = (uint32_t)__builtin_readcyclecounter();
#endif
    for (int tile = tileBeg; tile < tileEnd; tile++)
    {
        LVG_TICK(0);
        const int tileX = cur.tileX, tileY = cur.tileY;
        const int outX0 = tileX * TW, outY0 = tileY * TH;

        LVG_MARK("barrier1");
        __syncthreads();                                                    // barrier X: XL (ML, table) of this tile and WL of the previous tile visible
        LVG_TICK(1);

        // tmax[slotR] = max |x + bias| of the tile now in XL (written before barrier X); the other slot is cleared here for
        // the tile written after barrier Y
        const int slotR = (tile - tileBeg) & 1;
        const bool noClamp = MODE != LVG_SIGNS_READ && tmax[slotR] < xLimitBits;
        if (tid == 0) tmax[slotR ^ 1] = 0u;

        // ---- prefetch the next tile's input (and mask) into registers; it lands while this tile computes ----
        LVG_MARK("prefetch");
        TileCoord nxt = cur;
        if (tile + 1 < tileEnd)
        {
            if (++nxt.tileX == p.tilesX) { nxt.tileX = 0; if (++nxt.tileY == p.tilesY) { nxt.tileY = 0; ++nxt.plane; if (++nxt.ch == p.c) { nxt.ch = 0; ++nxt.nb; } } }
            if (!(LVG_ABL & 1)) issue_loads(nxt);
        }

        LVG_TICK(2);
        // ---- stage A: T'[ic][v] for this wave's 32 rows v ------------------------------------------------
        LVG_MARK("stageA");
        half8 tpk[G::IN_BLK][2];
        {
            f32x16 accA[G::IN_BLK];
            #pragma unroll
            for (int m = 0; m < G::IN_BLK; m++)
                #pragma unroll
                for (int r = 0; r < 16; r++) accA[m][r] = 0.0f;
            const int c0 = UpChunks<UP>::first(w), cnt = UpChunks<UP>::count(w), cls0 = UpChunks<UP>::cls0(w);
            #pragma unroll
            for (int t = 0; t < 2; t++)
            {
                if (t < cnt)
                {
                    const half8 fy = lds_frag(tab, G::IMG_FY + cls0 + t * UpChunks<UP>::step(), lane);
                    #pragma unroll
                    for (int m = 0; m < G::IN_BLK; m++)
                    {
                        const half8 xt = lds_tr_operand(XL, G::SX, 16 * (c0 + t), 32 * m, lane);
                        accA[m] = mfma(xt, fy, accA[m]);
                    }
                }
            }
            #pragma unroll
            for (int m = 0; m < G::IN_BLK; m++) { tpk[m][0] = pack_chunk(accA[m], 0); tpk[m][1] = pack_chunk(accA[m], 1); }
        }

        // ---- stage D of the PREVIOUS tile: independent of everything above, so its LDS-read -> 5-MFMA chain -> store
        //      latency overlaps with stage A / B of this tile instead of sitting alone between two barriers ----------
        LVG_TICK(3);
        LVG_MARK("stageD");
        stage_d(prv, tile > tileBeg);
        LVG_TICK(4);

        // ---- stages B, activation, C over the four 32-column blocks of u, software-pipelined: the MFMAs of block
        //      b + 1 are issued before the (vector-pipe) activation of block b, stage C of block b after it --------
        LVG_MARK("stageBC");
        f32x16 accW[G::OBX];
        #pragma unroll
        for (int bo = 0; bo < G::OBX; bo++)
            #pragma unroll
            for (int r = 0; r < 16; r++) accW[bo][r] = 0.0f;
        uint8_t* mrow = ML + (32 * w + n) * G::SM + g;
        f32x16 accU = stage_b(0, tpk);
        #pragma unroll
        for (int b = 0; b < 4; b++)
        {
            f32x16 accUn;
            if (LVG_MFMA_PIPELINE && b < 3) accUn = stage_b(b + 1, tpk);
            // Activation in packed f16 (act_block): registers 4q .. 4q + 3 are the four pixels of mask byte 8 b + 2 q + g.
            uint32_t zp[8];
            if (LVG_ABL & 4) { for (int i = 0; i < 8; i++) { half2v t; t[0] = (_Float16)accU[2 * i]; t[1] = (_Float16)accU[2 * i + 1]; zp[i] = h2_bits(t); } }
            else if (MODE == LVG_SIGNS_READ) act_block<MODE, true, false>(accU, zp, mrow + 8 * b, lut, slope2, clampP, clampN, clampBits);
            else if (slopeMax)
            {
                if (noClamp) act_block<MODE, true, false>(accU, zp, mrow + 8 * b, lut, slope2, clampP, clampN, clampBits);
                else         act_block<MODE, true, true>(accU, zp, mrow + 8 * b, lut, slope2, clampP, clampN, clampBits);
            }
            else act_block<MODE, false, true>(accU, zp, mrow + 8 * b, lut, slope2, clampP, clampN, clampBits);
            #pragma unroll
            for (int h = 0; h < 2; h++)
            {
                half8 z;
                __builtin_memcpy(&z, &zp[4 * h], 16);
                const int c = 2 * b + h;
                #pragma unroll
                for (int bo = 0; bo < G::OBX; bo++)
                {
                    const int cls = c - 2 * bo * DOWN;
                    if (cls >= 0 && cls < G::NDC)
                        accW[bo] = mfma(lds_frag(tab, G::IMG_DX + cls, lane), z, accW[bo]);
                }
            }
            if (b < 3) accU = LVG_MFMA_PIPELINE ? accUn : stage_b(b + 1, tpk);
        }
        LVG_TICK(5);
        LVG_MARK("barrier2");
        __syncthreads();                                                    // barrier Y: every wave is done reading XL and WL (and writing ML in WRITE mode)
        LVG_TICK(6);

        // W[ox][v] -> WL[v][ox]: registers 4q .. 4q + 3 are four consecutive ox
        LVG_MARK("wwrite");
        #pragma unroll
        for (int bo = 0; bo < G::OBX; bo++)
            #pragma unroll
            for (int q = 0; q < 4; q++)
            {
                half4 h;
                #pragma unroll
                for (int e = 0; e < 4; e++) h[e] = (_Float16)accW[bo][4 * q + e];
                if (32 * bo + 8 * q + 8 <= G::SW || 32 * bo + 8 * q + 4 * g + 4 <= G::SW)      // (the last 4 of 64 columns are not kept)
                    *reinterpret_cast<half4*>(WL + (32 * w + n) * G::SW + 32 * bo + 8 * q + 4 * g) = h;
            }

        LVG_TICK(7);
        // ---- WRITE mode: mask tile -> global, only the part this tile owns --------------------------------
        LVG_MARK("maskout");
        if (MODE == LVG_SIGNS_WRITE)
        {
            const int uStart = outX0 * DOWN, upY0 = outY0 * DOWN;            // (sign offsets are 0 when writing)
            const int signByte0 = uStart >> 2;
            const int ownRows  = (tileY == p.tilesY - 1) ? kUpT : TH * DOWN;
            const int row = tid >> 1, half = tid & 1;
            const int sy = upY0 + row;
            uint8_t* splane = p.s + (int64_t)cur.plane * ((int64_t)p.sH * p.sWBytes);
            if (row < ownRows && sy < p.sH)
            {
                uint8_t* srow = splane + (uint32_t)(sy * p.sWBytes);
                const uint32_t* m = reinterpret_cast<const uint32_t*>(ML + row * G::SM + 16 * half);
                const uint32_t wds[4] = {m[0], m[1], m[2], m[3]};
                if (tileX != p.tilesX - 1)
                {
                    // interior tile: it owns the first OWN bytes of every row, all of them inside the plane
                    constexpr int OWN = (TW * DOWN) / 4;
                    #pragma unroll
                    for (int hh = 0; hh < 2; hh++)
                    {
                        if (half == hh)
                        {
                            #pragma unroll
                            for (int d = 0; d < 4; d++)
                            {
                                constexpr int dummy = 0; (void)dummy;
                                const int k0 = 16 * hh + 4 * d;                       // compile-time after unrolling
                                if (k0 + 4 <= OWN) *reinterpret_cast<uint32_t*>(srow + signByte0 + k0) = wds[d];
                                else
                                {
                                    #pragma unroll
                                    for (int k = 0; k < 4; k++)
                                        if (k0 + k < OWN) srow[signByte0 + k0 + k] = (uint8_t)(wds[d] >> (8 * k));
                                }
                            }
                        }
                    }
                }
                else
                {
                    #pragma unroll
                    for (int d = 0; d < 4; d++)
                    {
                        const int k0 = 16 * half + 4 * d, bx0 = signByte0 + k0;
                        if (bx0 + 4 <= p.swLimit) *reinterpret_cast<uint32_t*>(srow + bx0) = wds[d];
                        else
                        {
                            #pragma unroll
                            for (int k = 0; k < 4; k++)
                                if (bx0 + k < p.swLimit) srow[bx0 + k] = (uint8_t)(wds[d] >> (8 * k));
                        }
                    }
                }
            }
            // bytes of the 16-pixel row padding carry no pixels: define them as 0
            if (tileX == p.tilesX - 1 && p.sWBytes > p.swLimit)
            {
                const int padBytes = p.sWBytes - p.swLimit;
                for (int idx = tid; idx < ownRows * padBytes; idx += kThreads)
                {
                    const int v = idx / padBytes, k = idx - v * padBytes;
                    const int sy2 = upY0 + v;
                    if (sy2 < p.sH) splane[(uint32_t)(sy2 * p.sWBytes + p.swLimit + k)] = 0;
                }
            }
        }

        LVG_TICK(8);
        // ---- the prefetched next tile -> XL (ML); stage A of this tile is behind barrier Y -----------------
        LVG_MARK("xwrite");
        slotW = ((tile - tileBeg) & 1) ^ 1;
        if (tile + 1 < tileEnd && !(LVG_ABL & 1)) write_tile();

        prv = cur;
        cur = nxt;
        LVG_TICK(9);
    }
#ifdef LVG_TIMING
    if (lane == 0)
    {
        uint32_t* o = g_flreluTiming + ((blockIdx.x & 1023) * 4 + w) * 16;
        for (int i = 0; i < 12; i++) o[i] = tAcc[i];
        o[12] = (uint32_t)(tileEnd - tileBeg);
    }
#endif
    // stage D of the last tile
    if (tileBeg < tileEnd)
    {
        __syncthreads();
        stage_d(prv, true);
    }
}

template <class T, int UP, int DOWN, int FU, int FD, int TW, int TH>
int launch_mfma(FlreluArgs& p, int mode, hipStream_t stream)
{
    typedef MG<UP, DOWN, FU, FD, TW, TH, LVG_SIGNS_READ> GR;
    typedef MG<UP, DOWN, FU, FD, TW, TH, LVG_SIGNS_WRITE> GW;
    typedef MG<UP, DOWN, FU, FD, TW, TH, LVG_SIGNS_NONE> GN;
    p.tilesX = (p.yw + TW - 1) / TW;
    p.tilesY = (p.yh + TH - 1) / TH;
    const int64_t tiles = (int64_t)p.tilesX * p.tilesY * p.n * p.c;
    LVG_REQUIRE(tiles <= 0x7fffffffLL, "filtered_lrelu: too many tiles for one launch");
    // lane offsets inside a plane are 32-bit byte offsets
    if (((int64_t)p.xh * p.xs[2] + (int64_t)p.xw * p.xs[3]) * 2 >= 0x7fffffffLL || ((int64_t)p.yh * p.ys[2] + (int64_t)p.yw * p.ys[3]) * 2 >= 0x7fffffffLL ||
        (int64_t)p.sH * p.sWBytes >= 0x7fffffffLL || p.xs[2] < 0 || p.xs[3] < 0 || p.ys[2] < 0 || p.ys[3] < 0)
        return LVG_ERR_UNSUPPORTED;
    // Persistent workgroups: 3 per CU fit (LDS), each walks over tiles with stride gridDim.
    static int cus[64] = {0};
    int dev = 0; (void)hipGetDevice(&dev);
    int ncu = cus[dev & 63];
    if (ncu == 0)
    {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev & 63] = ncu;
    }
    const int64_t maxGrid = (int64_t)ncu * (mode == LVG_SIGNS_READ ? GR::CU_WGS : (mode == LVG_SIGNS_WRITE ? GW::CU_WGS : GN::CU_WGS));
    const unsigned grid = (unsigned)(tiles < maxGrid ? tiles : maxGrid);
    const size_t lds = mode == LVG_SIGNS_READ ? GR::LDS_BYTES : (mode == LVG_SIGNS_WRITE ? GW::LDS_BYTES : GN::LDS_BYTES);
    // Pairs of input columns are fetched as one dword when every pair is dword aligned and never straddles the
    // image edge: unit x stride, even row / plane strides and width, even first input column of every tile.
    const int rOff = (mode == LVG_SIGNS_READ) ? (p.sOfsX & 3) : 0;
    const int inX00 = lvg_floor_div(UP - 1 - p.px0 - rOff, UP);
    const bool fast = p.xs[3] == 1 && (p.xs[2] & 1) == 0 && (p.xs[1] & 1) == 0 && (p.xs[0] & 1) == 0 && (p.xw & 1) == 0 &&
                      (inX00 & 1) == 0 && ((TW * DOWN / UP) & 1) == 0 && (((uintptr_t)p.x) & 3u) == 0;
    #define LVG_MFMA_LAUNCH(M, F) hipLaunchKernelGGL((filtered_lrelu_mfma_kernel<T, UP, DOWN, FU, FD, TW, TH, M, F>), dim3(grid), dim3(kThreads), lds, stream, p, (int)tiles)
    if (mode == LVG_SIGNS_WRITE)     { if (fast) LVG_MFMA_LAUNCH(LVG_SIGNS_WRITE, true); else LVG_MFMA_LAUNCH(LVG_SIGNS_WRITE, false); }
    else if (mode == LVG_SIGNS_READ) { if (fast) LVG_MFMA_LAUNCH(LVG_SIGNS_READ, true);  else LVG_MFMA_LAUNCH(LVG_SIGNS_READ, false); }
    else                             { if (fast) LVG_MFMA_LAUNCH(LVG_SIGNS_NONE, true);  else LVG_MFMA_LAUNCH(LVG_SIGNS_NONE, false); }
    #undef LVG_MFMA_LAUNCH
    return lvg_check_launch("filtered_lrelu_mfma_kernel");
}

template <class T>
int run_mfma(FlreluArgs& p, int cfg, int mode, hipStream_t stream)
{
    switch (cfg)
    {
        case LVG_FLRELU_CFG_U2D2: return launch_mfma<T, 2, 2, 12, 12, 56, 58>(p, mode, stream);
        case LVG_FLRELU_CFG_U4D2: return launch_mfma<T, 4, 2, 24, 12, 56, 58>(p, mode, stream);
        case LVG_FLRELU_CFG_U2D4: return launch_mfma<T, 2, 4, 12, 24, 26, 27>(p, mode, stream);
    }
    return LVG_ERR_UNSUPPORTED;
}

} // namespace

int lvg_flrelu_mfma_launch(FlreluArgs& p, int cfg, int mode, int dtype, hipStream_t stream)
{
    if (dtype == LVG_F16)  return run_mfma<f16_t>(p, cfg, mode, stream);
    if (dtype == LVG_BF16) return run_mfma<bf16_t>(p, cfg, mode, stream);
    return LVG_ERR_UNSUPPORTED;
}
