// filtered_lrelu_wave.hip -- fused bias -> up-FIR -> gain -> leaky ReLU -> clamp -> down-FIR for float16 / bfloat16
// tensors on gfx950 (round 4): the banded-matrix formulation of filtered_lrelu_mfma.hip re-decomposed so that ONE
// WAVE owns one tile from the first load to the last store. No workgroup barrier in the tile loop.
//
// Semantics: exactly those of filtered_lrelu.hip / filtered_lrelu_mfma.hip (reference torch_utils/ops/filtered_lrelu.cu:139-1099,
// filtered_lrelu.cpp:16-210), including the 2-bit sign / clamp mask (write, and read with offsets).
//
// Why a re-decomposition: the round-2 kernel shared a 128 x 128 up-sampled tile between four waves
// through LDS and two barriers per tile; its measured bound was not a pipe but the serial latency of a 4-wave
// rendezvous (every added piece of work added its full time). Here
//   * a wave's tile is 128 (u) x 32 VB (v) up-sampled pixels; the wave walks over its VB row blocks, keeps the
//     chain  A (vertical up) -> B (horizontal up) -> activation -> C (horizontal down)  in registers exactly like
//     the round-2 kernel, parks W[v][ox] in a wave-private LDS region and finishes with stage D (vertical down);
//   * LDS operations of one wave execute in order, so the input tile, W and the mask staging rows are reused from
//     tile to tile without any synchronisation; waves never wait for each other;
//   * the constant band fragments of stages A, B and C live in REGISTERS (two waves per SIMD leave 256 of them), so
//     the K loop issues no LDS reads for constants;
//   * W is stored with its columns de-interleaved (even ox | odd ox): stage D then produces, in one lane, the two
//     neighbouring output pixels of a row, and a store instruction writes 128 contiguous bytes per row instead of
//     32 scattered 8-byte pieces (the round-3 ablation put the scattered stores at a quarter of the time);
//   * the mask leaves / enters the registers as whole dwords: bytes are assembled with v_perm / v_sad_u8 and one
//     v_permlane32_swap per block instead of byte-wide LDS traffic; READ mode turns a mask byte into the four packed
//     gradient factors of its pixels with ONE 8-byte LDS table read (the kernel is bound by the vector-instruction port).
//
// Arithmetic (unchanged): f16 operands (bfloat16 tensors are converted: their 8-bit mantissa is exact in f16), f32
// accumulation; T', Z and W are rounded to f16 between stages; filter taps are rounded to f16.
// Algorithmic HBM bytes: (N_in + N_out) * 2 + mask bytes; see DESIGN.md.

#include "lvg_common.h"
#include "filtered_lrelu_args.h"
#include <atomic>
#include <stdlib.h>

#ifndef LVG_WABL
#define LVG_WABL 0           // ablation builds only (results are WRONG): 1 no next-tile loads, 2 no y stores, 4 no activation math, 8 no stage D, 16 no mask stores, 32 no matrix products (memory traffic only)
#endif

// Analysis builds only (-DLVG_TIMING; tools/flrelu_check prints the table): shader-clock cycles each wave spends per region of the
// tile loop, read with s_memtime at the region boundaries and left in g_flwTiming[(workgroup * 4 + wave) * 16 + region].
#ifdef LVG_TIMING
__device__ uint32_t g_flwTiming[4096 * 16];
#define LVG_TICK(idx) { const uint32_t now_ = (uint32_t)__builtin_readcyclecounter(); tAcc[idx] += now_ - tLast; tLast = now_; }
extern "C" int lvg_flrelu_wave_timing_read(uint32_t* host, int count)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_flwTiming), (size_t)count * 4, 0, hipMemcpyDeviceToHost);
}
#else
#define LVG_TICK(idx)
#endif

namespace {

#include "flrelu_mfma_common.h"

#ifndef LVG_WAVES_PER_WG
#define LVG_WAVES_PER_WG 4
#endif
constexpr int kWaves = LVG_WAVES_PER_WG;   // waves per workgroup (they share only the taps and the D_y fragment table)
constexpr int kThreads = 64 * kWaves;


template <int UP, int DOWN, int FU, int FD, int VB, int TW, int TH, int MODE>
struct WGeo
{
    static constexpr int KU     = FU / UP;                                  // taps per output of an up stage
    static constexpr int V      = 32 * VB;                                  // up-sampled rows of a tile
    static constexpr int IN_NX  = (UP - 1 + kU - 1) / UP + KU;              // input columns / rows a tile touches
    static constexpr int IN_NY  = (UP - 1 + V - 1) / UP + KU;
    static constexpr int IN_BLK = wdiv_up(IN_NX, 32);                       // 32-blocks of input columns (stage A's M)
    static constexpr int CH_X   = ((((kU - 1 + UP - 1) / UP) + KU - 1) >> 4) + 1;   // 16-chunks of input columns stage B reads
    static constexpr int CH_Y   = ((((V - 1 + UP - 1) / UP) + KU - 1) >> 4) + 1;    // 16-chunks of input rows stage A reads
    static constexpr int SX     = CH_X * 16;                                // X row stride (halves): every column stage B multiplies lies inside the row
    static constexpr int X_ROWS = CH_Y * 16;
    static constexpr int OBX    = wdiv_up(TW, 32);                          // 32-blocks of output columns (stage C's M)
    static constexpr int SW     = 32 * OBX + 4;                             // W row stride (halves): 8-byte column writes of 16 rows hit distinct banks
    static constexpr int NUC    = (UP == 2) ? 2 : (UP == 4 ? 3 : 1);        // distinct band offsets of an up stage
    static constexpr int NDC    = ((31 * DOWN + FD - 1 + 3) >> 4) + 1;      // ... of a down stage (+3: READ-mode column shift)
    static constexpr int NDX    = NDC < 8 ? NDC : 8;                        // classes stage C can meet inside 8 chunks of u
    static constexpr int NDYC   = ((31 * DOWN + FD - 1) >> 4) + 1;
    static constexpr int NDY    = (V / 16) < NDYC ? (V / 16) : NDYC;        // classes (= chunks of v) stage D uses
    // global <-> LDS traffic moves as 16-byte vectors (8 pixels): the vector-memory pipe is paid per instruction
    static constexpr int LPR    = wdiv_up(IN_NX, 8);                        // input: vectors per row
    static constexpr int NVEC   = IN_NY * LPR;
    static constexpr int NLOAD  = wdiv_up(NVEC, 64);
    static constexpr int NVY    = wdiv_up(TW, 8);                           // output: vectors per row
    static constexpr int YP     = (NVY | 1) * 16 + (NVY % 2 == 0 ? 0 : 32); // output staging row pitch (bytes): an odd multiple of 16 >= 64 OBX (blocks are written whole)
    static constexpr int NSTORE = wdiv_up(TH * NVY, 64);
    static constexpr bool HAS_M = MODE != LVG_SIGNS_NONE;
    static constexpr int SM     = 40;                                       // mask staging row stride (bytes): 32 used
    static constexpr int TAPS   = (FU + FD + 3) / 4 * 4;
    // LDS map (bytes)
    static constexpr int OFF_TAPS  = 0;
    static constexpr int OFF_LUT   = TAPS * 4;                              // READ mode: mask byte -> four packed gradient factors (8 bytes per entry)
    static constexpr int WAVES_LDS = X_ROWS * SX * 2 + 64 + V * SW * 2 + 16 + (HAS_M ? 32 * SM : 0);
    // 256 entries where the two workgroups of a CU leave room, else the 171 a writer can produce (codes 0..2 per pixel: <= 0xAA)
    static constexpr int LUTN      = MODE != LVG_SIGNS_READ ? 0 : ((2 * (OFF_LUT + 2048 + NDY * 1024 + kWaves * WAVES_LDS) <= 160 * 1024) ? 256 : 171);
    static constexpr int OFF_TAB   = OFF_LUT + (LUTN * 8 + 15) / 16 * 16;
    static constexpr int OFF_WAVE  = OFF_TAB + NDY * 1024;
    static constexpr int X_BYTES   = X_ROWS * SX * 2 + 64;                  // (+64: stage A's last transpose read runs past the last row; its result is never used)
    static constexpr int W_BYTES   = V * SW * 2 + 16;                       // (the output staging rows alias W: stage D has read W when it writes them)
    static constexpr int M_BYTES   = HAS_M ? 32 * SM : 0;
    static constexpr int WAVE_BYTES = X_BYTES + W_BYTES + M_BYTES;
    static constexpr int LDS_BYTES = OFF_WAVE + kWaves * WAVE_BYTES;
    static_assert(FU % UP == 0 && FD % DOWN == 0, "filter sizes must be multiples of the rates");
    static_assert(IN_NX % 2 == 0 && IN_NX <= SX && LPR * 8 <= SX, "input tile geometry");
    static_assert((TW * DOWN) % 4 == 0 && (TW * DOWN) % UP == 0 && (TH * DOWN) % UP == 0, "tile origin must keep the mask byte and the up-sampling phase fixed");
    static_assert((TW - 1) * DOWN + FD - 1 + 3 < kU && (TH - 1) * DOWN + FD - 1 < V, "tile does not fit its up-sampled block");
    static_assert(TH <= 32 && TW <= 64 && TW % 2 == 0, "stage D: one block of output rows");
    static_assert(YP >= 64 * OBX && YP % 16 == 0 && 32 * YP <= W_BYTES, "output staging rows");
    static_assert((8 / kWaves) * LDS_BYTES <= 160 * 1024, "eight waves per CU");
    static_assert(OFF_TAB % 16 == 0 && OFF_WAVE % 16 == 0 && X_BYTES % 16 == 0 && W_BYTES % 16 == 0 && WAVE_BYTES % 16 == 0, "alignment");
};

struct TileCoord { int tileX, tileY, ch, nb, plane; };   // plane = nb * channels + ch

typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <bool B> struct BoolC { static constexpr bool value = B; };

template <class T, int UP, int DOWN, int FU, int FD, int VB, int TW, int TH, int MODE, bool FASTLOAD>
__global__ __launch_bounds__(kThreads, 2) void filtered_lrelu_wave_kernel(FlreluArgs p, int totalTiles)
{
    typedef WGeo<UP, DOWN, FU, FD, VB, TW, TH, MODE> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float*    taps  = reinterpret_cast<float*>(smem + G::OFF_TAPS);         // [0, FU): up taps, [FU, FU + FD): down taps (flipped)
    _Float16* tabDy = reinterpret_cast<_Float16*>(smem + G::OFF_TAB);       // D_y fragment images (natural k order), 512 halves each, lane-major
    const int tid = threadIdx.x, lane = tid & 63, w = sgpr(tid >> 6);
    const int n = lane & 31, g = lane >> 5;
    unsigned char* wv = smem + G::OFF_WAVE + w * G::WAVE_BYTES;             // this wave's private region
    _Float16* XL = reinterpret_cast<_Float16*>(wv);                         // input tile + bias [X_ROWS][SX]
    _Float16* WL = reinterpret_cast<_Float16*>(wv + G::X_BYTES);            // W [V][SW]
    unsigned char* YL = wv + G::X_BYTES;                                    // output staging rows [32][YP bytes] (alias W)
    unsigned char* ML = wv + G::X_BYTES + G::W_BYTES;                       // mask rows of one row block [32][SM] (not with MODE NONE)

    // ---- once per workgroup: taps, D_y table; once per wave: zero the input tile (its padding is never written) ----
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
    for (int i = lane; i < G::X_BYTES / 4; i += 64) reinterpret_cast<uint32_t*>(XL)[i] = 0u;
    if (MODE == LVG_SIGNS_READ)
    {
        // mask byte (codes of four pixels, 2 bits each: 0 pass, 1 negative, 2 / 3 clamped) -> (f0, f1), (f2, f3) packed f16
        uint32_t* lut = reinterpret_cast<uint32_t*>(smem + G::OFF_LUT);
        const _Float16 one = (_Float16)1.0f, sl = (_Float16)p.slope, zero = (_Float16)0.0f;
        for (int e = tid; e < G::LUTN; e += kThreads)
        {
            half2v lo, hi;
            #pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int c = (e >> (2 * j)) & 3;
                const _Float16 f = c == 0 ? one : (c == 1 ? sl : zero);
                if (j < 2) lo[j] = f; else hi[j - 2] = f;
            }
            lut[2 * e] = h2_bits(lo); lut[2 * e + 1] = h2_bits(hi);
        }
    }
    __syncthreads();

    // Launch-constant geometry: the column shift that aligns the tile with the mask bytes in READ mode and the
    // zero-insertion phases (tile origins are multiples of 4 and of UP in the up-sampled plane).
    const int rOff = (MODE == LVG_SIGNS_READ) ? (p.sOfsX & 3) : 0;
    const int phX = ((UP - 1 - p.px0 - rOff) % UP + UP) % UP;
    const int phY = ((UP - 1 - p.py0) % UP + UP) % UP;
    const float scale = (float)(UP * UP) * p.gain;
    for (int e = tid; e < G::NDY * 512; e += kThreads)
        tabDy[e] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 3, e >> 9, (e & 511) >> 3, e & 7, phX, phY, rOff, scale);
    __syncthreads();

    // ---- band fragments of stages A, B, C in registers ------------------------------------------------------
    half8 fAy[G::NUC], fAx[G::NUC], fDx[G::NDX];
    #pragma unroll
    for (int c = 0; c < G::NUC; c++)
        #pragma unroll
        for (int j = 0; j < 8; j++)
        {
            fAy[c][j] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 0, c, lane, j, phX, phY, rOff, scale);
            fAx[c][j] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 1, c, lane, j, phX, phY, rOff, scale);
        }
    #pragma unroll
    for (int c = 0; c < G::NDX; c++)
        #pragma unroll
        for (int j = 0; j < 8; j++) fDx[c][j] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 2, c, lane, j, phX, phY, rOff, scale);

    // ---- tiles: a WORKGROUP owns a contiguous range and its waves take them round-robin (wave w: first + w, + kWaves, ...), so
    //      the tiles in flight on a CU are neighbours: the halo rows / columns they share come out of L1 / L2 instead of HBM and
    //      the partial cache lines of x-neighbours meet in L2 (with one contiguous range per WAVE the planes in flight on an
    //      XCD were several times its L2 and the input was fetched 2.3 times). tile -> (tileX, tileY, channel, sample). -----
    const int wgBeg = (int)((int64_t)totalTiles * blockIdx.x / gridDim.x);
    const int tileEnd = (int)((int64_t)totalTiles * (blockIdx.x + 1) / gridDim.x);
    const int tileBeg = wgBeg + w;
    TileCoord cur;
    {
        int bid = tileBeg;
        cur.tileX = bid % p.tilesX; bid /= p.tilesX;
        cur.tileY = bid % p.tilesY; bid /= p.tilesY;
        cur.plane = bid;
        cur.ch = bid % p.c; cur.nb = bid / p.c;
        cur.tileX = sgpr(cur.tileX); cur.tileY = sgpr(cur.tileY); cur.ch = sgpr(cur.ch); cur.nb = sgpr(cur.nb); cur.plane = sgpr(cur.plane);
    }

    // ---- activation constants (packed f16) -----------------------------------------------------------------
    ActConst K;
    {
        const _Float16 slope_h = (_Float16)p.slope;
        K.slope2[0] = slope_h; K.slope2[1] = slope_h;
        const _Float16 clamp_h = (_Float16)(p.clamp < 65504.0f ? p.clamp : 65504.0f);    // no clamp = the largest finite f16
        K.clampP[0] = clamp_h; K.clampP[1] = clamp_h; K.clampN[0] = -clamp_h; K.clampN[1] = -clamp_h;
        K.clampBits = h2_bits(K.clampP);
        K.shEven = 8u * (uint32_t)g;
        K.shOdd = 16u + 8u * (uint32_t)g;
        typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
        K.lutBase = (uint32_t)(uintptr_t)(lds_bytes)(smem + G::OFF_LUT);     // LDS byte address of the table
    }
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
            for (int t = ph; t < FU; t += UP) a += fabsf(taps[t]);
            l1 = fmaxf(l1, a);
        }
        const float lim = p.clamp / ((float)(UP * UP) * p.gain * l1 * l1 * 1.05f + 1e-30f);
        const _Float16 lh = (_Float16)fminf(lim, 60000.0f);
        uint16_t lb; __builtin_memcpy(&lb, &lh, 2);
        xLimitBits = ((float)lh <= lim && lb > 0) ? lb : (lb > 0 ? lb - 1u : 0u);      // round down
        if (!(p.slope <= 1.0f)) xLimitBits = 0;
        xLimitBits = (uint32_t)sgpr((int)xLimitBits);
    }

    // ---- input loader: the tile's IN_NY x IN_NX pixels as 16-byte vectors (8 pixels of one row), vector lane + 64 i in
    //      pass i. Unit pixel stride: ONE buffer load per vector from "tile origin + lane offset" -- offsets outside the
    //      plane (rows above / below the image, and the wrap of negative ones) return zeros without faulting, whatever
    //      they fetch is replaced when the tile is written to LDS. ---------------------------------------------------
    const uint32_t pitchB = (uint32_t)((int)p.xs[2]) * 2u, colB = (uint32_t)((int)p.xs[3]) * 2u;
    // this lane's vector of pass i: (row, first column) = (idx / LPR, 8 (idx % LPR)), idx = lane + 64 i; recomputed where needed (a
    // few integer operations) instead of held in registers; only the byte offset from the tile's first pixel is kept.
    auto ld_row = [&](int i) __attribute__((always_inline)) { return div_small<G::LPR>(lane + 64 * i); };
    auto ld_col = [&](int i, int row) __attribute__((always_inline)) { return (lane + 64 * i < G::NVEC) ? 8 * (lane + 64 * i - row * G::LPR) : -1; };
    uint32_t ldOff[G::NLOAD];
    #pragma unroll
    for (int i = 0; i < G::NLOAD; i++)
    {
        const int r = ld_row(i), c = ld_col(i, r);
        ldOff[i] = (c >= 0) ? (uint32_t)r * pitchB + (uint32_t)c * colB : 0u;
    }
    v4u raw[G::NLOAD];                                                      // prefetched vectors of the NEXT tile (storage bits)
    float biasN = 0.0f;
    uint32_t negbN = 0;                                                     // next tile: (-bias, -bias) in storage bits
    int ldInY0N = 0, ldInX0N = 0;                                           // next tile: first input row / column (may be negative)
    const uint32_t planeSpanB = (uint32_t)(((int64_t)(p.xh - 1) * p.xs[2] + (int64_t)(p.xw - 1) * p.xs[3] + 1) * 2);

    auto issue_loads = [&](const TileCoord& tc) __attribute__((always_inline))
    {
        const int uStart = tc.tileX * (TW * DOWN) - rOff, upY0 = tc.tileY * (TH * DOWN);
        const int inX0 = lvg_floor_div(uStart + UP - 1 - p.px0, UP);
        const int inY0 = lvg_floor_div(upY0 + UP - 1 - p.py0, UP);
        ldInY0N = inY0; ldInX0N = inX0;
        const char* xpl = (const char*)((const T*)p.x + ((int64_t)tc.nb * p.xs[0] + (int64_t)tc.ch * p.xs[1]));
        const uint32_t bb = scalar_load_u16((const uint16_t*)p.b + sgpr(tc.ch));
        { T bt; const uint16_t b16 = (uint16_t)bb; __builtin_memcpy(&bt, &b16, 2); biasN = (float)to_acc(bt); }
        negbN = (bb ^ 0x8000u) * 0x10001u;                                    // (-bias, -bias): + bias = 0 outside the image
        const uint32_t base = (uint32_t)(inY0 * (int)p.xs[2] + inX0 * (int)p.xs[3]) * 2u;
        if (FASTLOAD)
        {
            // The buffer covers the plane plus up to 16 bytes of the tensor on either side: a vector that starts before the
            // plane's first pixel (or ends behind its last) still delivers the pixels it shares with the plane.
            const int64_t planeOffB = ((int64_t)tc.nb * p.xs[0] + (int64_t)tc.ch * p.xs[1]) * 2;
            const int lo = (int)min((int64_t)16, p.xLoB + planeOffB), hi = (int)min((int64_t)16, p.xHiB - planeOffB - (int64_t)planeSpanB);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(xpl - lo), 0, (int)planeSpanB + lo + hi, 0x00020000);
            #pragma unroll
            for (int i = 0; i < G::NLOAD; i++)
                raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + (uint32_t)lo + ldOff[i]), 0, 0);
            if (lo < 16 || hi < 16)
            {
                // first / last plane of the tensor: a vector that straddles the tensor's edge came back as zeros; fetch its pixels one by one
                const uint32_t lastEl = planeSpanB - 2u;
                #pragma unroll
                for (int i = 0; i < G::NLOAD; i++)
                {
                    const int o = (int)(base + ldOff[i]);
                    if ((o < 0 && o + 16 > 0) || (o < (int)planeSpanB && o + 16 > (int)planeSpanB))
                    {
                        #pragma unroll
                        for (int d = 0; d < 4; d++)
                        {
                            const uint32_t lo16 = *reinterpret_cast<const uint16_t*>(xpl + min((uint32_t)(o + 4 * d), lastEl));
                            const uint32_t hi16 = *reinterpret_cast<const uint16_t*>(xpl + min((uint32_t)(o + 4 * d + 2), lastEl));
                            raw[i][d] = lo16 | (hi16 << 16);
                        }
                    }
                }
            }
        }
        else
        {
            // any strides: element loads from offsets clamped into the plane
            const uint32_t lastEl = planeSpanB - 2u;
            #pragma unroll
            for (int i = 0; i < G::NLOAD; i++)
            {
                const uint32_t o = base + ldOff[i];
                #pragma unroll
                for (int d = 0; d < 4; d++)
                {
                    const uint32_t lo = *reinterpret_cast<const uint16_t*>(xpl + min(o + (uint32_t)(2 * d) * colB, lastEl));
                    const uint32_t hi = *reinterpret_cast<const uint16_t*>(xpl + min(o + (uint32_t)(2 * d + 1) * colB, lastEl));
                    raw[i][d] = lo | (hi << 16);
                }
            }
        }
    };
    // raw -> XL (+ bias; -bias where the pixel lies outside the image or beyond the tile's IN_NX columns, so that + bias gives
    // the zero padding). Returns the tile's max |x + bias| as f16 bits (WRITE / NONE modes: the no-clamp proof), wave-uniform.
    auto write_tile = [&]() __attribute__((always_inline)) -> uint32_t
    {
        const _Float16 bh = (_Float16)biasN;
        const half2v bias2 = {bh, bh};
        uint32_t mx2 = 0;
        // the tile's rows [yLo, yHi) and columns [xLo, xHi) (tile coordinates) are pixels; everything else is padding (wave-uniform)
        const int yLo = max(0, -ldInY0N), yHi = min(G::IN_NY, p.xh - ldInY0N);
        const int xLo = max(0, -ldInX0N), xHi = min(G::IN_NX, p.xw - ldInX0N);
        const bool interior = yLo == 0 && yHi == G::IN_NY && xLo == 0 && xHi == G::IN_NX && G::LPR * 8 <= p.xw - ldInX0N;
        #pragma unroll
        for (int i = 0; i < G::NLOAD; i++)
        {
            const int lrow = ld_row(i), lcol = ld_col(i, lrow);
            const bool active = lcol >= 0;
            // bit e of `valid`: pixel e of this lane's vector is a pixel of the image (integer arithmetic only: the compare / lane-mask
            // form of these tests cost more scalar and vector instructions than the rest of the loader)
            uint32_t valid = 0xffu;
            if (!interior)
            {
                const int lo = min(max(xLo - lcol, 0), 8), hi = min(max(xHi - lcol, 0), 8);
                valid = (0xffu >> (8 - hi)) & (0xffu << lo);
                valid = ((uint32_t)(lrow - yLo) < (uint32_t)(yHi - yLo)) ? valid : 0u;
            }
            else if (G::LPR * 8 > G::IN_NX) valid = 0xffu >> max(lcol + 8 - G::IN_NX, 0);   // (the columns of a row's last vector beyond the tile)
            v4u hvv;
            #pragma unroll
            for (int d = 0; d < 4; d++)
            {
                uint32_t v = raw[i][d];
                if (!interior || (G::LPR * 8 > G::IN_NX && 8 * (G::LPR - 1) + 2 * d + 2 > G::IN_NX))
                {
                    const uint32_t k0 = (uint32_t)__builtin_amdgcn_sbfe((int)valid, 2 * d, 1), k1 = (uint32_t)__builtin_amdgcn_sbfe((int)valid, 2 * d + 1, 1);   // 0 / all ones
                    const uint32_t keep = (k0 & 0xffffu) | (k1 & 0xffff0000u);
                    v = (v & keep) | (negbN & ~keep);
                }
                const half2v hv = pair_plus_bias<T>(v, bias2, biasN);
                hvv[d] = h2_bits(hv);
                if (MODE != LVG_SIGNS_READ)
                {
                    const uint32_t ab = h2_bits(hv) & 0x7fff7fffu;           // (idle lanes hold + bias of zeros or of stale pixels: finite, and harmless for a bound)
                    ushort2v am, cm; __builtin_memcpy(&am, &ab, 4); __builtin_memcpy(&cm, &mx2, 4);
                    cm = __builtin_elementwise_max(cm, am);
                    __builtin_memcpy(&mx2, &cm, 4);
                }
            }
            if (active) *reinterpret_cast<v4u*>(XL + lrow * G::SX + lcol) = hvv;
        }
        uint32_t m = 0;
        if (MODE != LVG_SIGNS_READ)
        {
            // wave maximum (DPP butterfly inside rows of 16, then row broadcasts); lane 63 holds it
            m = max(mx2 & 0xffffu, mx2 >> 16);
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, false));    // row_half_mirror
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, false));    // row_mirror
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x142, 0xa, 0xf, false));    // row_bcast15 -> rows 1, 3
            m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x143, 0xc, 0xf, false));    // row_bcast31 -> rows 2, 3
            m = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
        }
        return m;
    };

    // ---- READ mode: 16 bytes of one mask row per lane (row = lane >> 1 of a row block, half = lane & 1), fetched one row
    //      block ahead as the 5 aligned dwords that cover them (rows of the mask plane are dword aligned, the tile's first
    //      byte is not); stage_mask() shifts them into place and hands every lane its row's 32 bytes. --------------------
    uint32_t mraw[5];
    int mshiftN = 0, mvalidN = 0;
    uint32_t mokN = 0;
    auto issue_mask_loads = [&](const TileCoord& tc, int vb) __attribute__((always_inline))
    {
        const int uStart = tc.tileX * (TW * DOWN) - rOff, upY0 = tc.tileY * (TH * DOWN) + 32 * vb;
        const int row = lane >> 1, half = lane & 1;
        const int signByte0 = (uStart + p.sOfsX) >> 2;
        const int sy = upY0 + p.sOfsY + row;
        const bool rowOk = (uint32_t)sy < (uint32_t)p.sH;
        const uint8_t* spl = p.s + (int64_t)tc.plane * ((int64_t)p.sH * p.sWBytes);
        const int b0 = signByte0 + 16 * half, a0 = b0 & ~3;
        const uint32_t rowOff = (uint32_t)(sy * p.sWBytes);
        mshiftN = signByte0 & 3;
        mvalidN = p.swLimit - b0;                                            // bytes of this lane's 16 that carry pixels (may be <= 0 or >= 16)
        mokN = 0;
        #pragma unroll
        for (int j = 0; j < 5; j++)
        {
            const int bx = a0 + 4 * j;
            const bool ok = rowOk && bx >= 0 && bx + 4 <= p.sWBytes;
            mokN |= ok ? (1u << j) : 0u;
            mraw[j] = *reinterpret_cast<const uint32_t*>(spl + (ok ? rowOff + (uint32_t)bx : 0u));
        }
    };
    auto stage_mask = [&](uint32_t (&M8)[8]) __attribute__((always_inline))
    {
        uint32_t* m = reinterpret_cast<uint32_t*>(ML + (lane >> 1) * G::SM + 16 * (lane & 1));
        #pragma unroll
        for (int d = 0; d < 4; d++)
        {
            const uint32_t lo = (mokN >> d) & 1u ? mraw[d] : 0u, hi = (mokN >> (d + 1)) & 1u ? mraw[d + 1] : 0u;
            uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)mshiftN);
            const int nv = mvalidN - 4 * d;                                  // bytes at and beyond swLimit carry no pixels
            if (nv < 4) v = nv <= 0 ? 0u : (v & ((1u << (8 * nv)) - 1u));
            m[d] = v;
        }
        const uint32_t* r = reinterpret_cast<const uint32_t*>(ML + n * G::SM);
        #pragma unroll
        for (int d = 0; d < 8; d++) M8[d] = r[d];
    };

    // ---- output: stage D leaves output row n in lane n; the tile goes through LDS once (the W region is free by then) and
    //      leaves as 16-byte vectors, vector lane + 64 i in pass i. -----------------------------------------------------
    const uint32_t yPitchB = (uint32_t)((int)p.ys[2]) * 2u, yColB = (uint32_t)((int)p.ys[3]) * 2u;
    auto st_row = [&](int i) __attribute__((always_inline)) { return div_small<G::NVY>(lane + 64 * i); };
    auto st_col = [&](int i, int row) __attribute__((always_inline)) { return (row < TH) ? 8 * (lane + 64 * i - row * G::NVY) : -1; };
    uint32_t stOff[G::NSTORE];
    #pragma unroll
    for (int i = 0; i < G::NSTORE; i++)
    {
        const int r = st_row(i), c = st_col(i, r);
        stOff[i] = (c >= 0) ? (uint32_t)r * yPitchB + (uint32_t)c * yColB : 0u;
    }
    const bool fastStore = p.ys[3] == 1;                                    // (16-byte stores need no alignment beyond the element's)

    // Software pipeline over tiles: while tile t computes, tile t + 1 sits in the prefetch registers until t's last row block has
    // read the input tile, then moves to LDS, and the loads of tile t + 2 go out at once: a tile's loads have a whole tile
    // period to land and the wave has loads in flight all the time (the kernel is bound by bytes in flight per CU).
    auto advance = [&](TileCoord& tc) __attribute__((always_inline))     // this wave's next tile: kWaves further
    {
        #pragma unroll
        for (int i = 0; i < kWaves; i++)
            if (++tc.tileX == p.tilesX) { tc.tileX = 0; if (++tc.tileY == p.tilesY) { tc.tileY = 0; ++tc.plane; if (++tc.ch == p.c) { tc.ch = 0; ++tc.nb; } } }
    };
    uint32_t tmaxCur = 0;
    TileCoord nxt = cur;
    if (tileBeg < tileEnd)
    {
        issue_loads(cur);
        tmaxCur = write_tile();
        if (MODE == LVG_SIGNS_READ) issue_mask_loads(cur, 0);
        if (tileBeg + kWaves < tileEnd && !(LVG_WABL & 1)) { advance(nxt); issue_loads(nxt); }
    }

#ifdef LVG_TIMING
    uint32_t tAcc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t tLast = (uint32_t)__builtin_readcyclecounter();
#endif
    for (int tile = tileBeg; tile < tileEnd; tile += kWaves)
    {
        LVG_TICK(0);
        const int tileX = cur.tileX, tileY = cur.tileY;
        const int outX0 = tileX * TW, outY0 = tileY * TH;
        const bool noClamp = MODE != LVG_SIGNS_READ && tmaxCur < xLimitBits;

        const bool hasNext = tile + kWaves < tileEnd, hasNext2 = tile + 2 * kWaves < tileEnd;
        TileCoord nxt2 = nxt;
        uint32_t tmaxNext = 0;
        LVG_TICK(1);

        #pragma unroll
        for (int vb = 0; vb < VB; vb++)
        {
            // ---- READ: this row block's mask bytes; the next block's (or the next tile's first) loads go out ----------
            uint32_t M8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (MODE == LVG_SIGNS_READ)
            {
                stage_mask(M8);
                if (vb + 1 < VB) issue_mask_loads(cur, vb + 1);
                else if (hasNext) issue_mask_loads(nxt, 0);
            }
            LVG_TICK(2);

            // ---- stage A: T'[ic][v] for the 32 rows v of this block --------------------------------------------
            half8 tpk[G::IN_BLK][2];
            {
                f32x16 accA[G::IN_BLK];
                #pragma unroll
                for (int m = 0; m < G::IN_BLK; m++) accA[m] = zero16();
                const int c0 = UpChunks<UP>::first(vb), cnt = UpChunks<UP>::count(vb), cls0 = UpChunks<UP>::cls0(vb);
                #pragma unroll
                for (int t = 0; t < 2; t++)
                {
                    if (t < cnt)
                    {
                        #pragma unroll
                        for (int m = 0; m < G::IN_BLK; m++)
                        {
                            const half8 xt = lds_tr_operand(XL, G::SX, 16 * (c0 + t), 32 * m, lane);
                            accA[m] = mfma(xt, fAy[cls0 + t * UpChunks<UP>::step()], accA[m]);
                        }
                    }
                }
                #pragma unroll
                for (int m = 0; m < G::IN_BLK; m++) { tpk[m][0] = pack_chunk(accA[m], 0); tpk[m][1] = pack_chunk(accA[m], 1); }
            }
            LVG_TICK(3);
            // the last row block has read the input tile: the prefetched next tile can replace it (LDS operations of a wave
            // execute in order)
            if (vb == VB - 1 && !(LVG_WABL & 1))
            {
                if (hasNext) tmaxNext = write_tile();
                if (hasNext2) { advance(nxt2); issue_loads(nxt2); }
            }
            LVG_TICK(4);

            // Stages B, activation, C, the W rows and the mask of this row block: instantiated per activation variant and selected once
            // per row block, so that the four column blocks are one straight-line stretch the compiler can software-pipeline (a
            // per-block branch on the variant cut it into pieces). Loader and stage A above are shared by the variants.
            uint32_t mdw[4] = {0, 0, 0, 0};
            auto row_block = [&](auto slopeMaxC, auto clampC) __attribute__((always_inline))
            {
                constexpr bool SLOPEMAX = decltype(slopeMaxC)::value, CLAMP = decltype(clampC)::value;
                // One 32 x 32 block of U^T = A_x * T' (stage B) for column block b of the up-sampled tile.
                auto stage_b = [&](int b) __attribute__((always_inline)) -> f32x16
                {
                    f32x16 acc = zero16();
                    #pragma unroll
                    for (int t = 0; t < 2; t++)
                    {
                        if (t < UpChunks<UP>::count(b))
                        {
                            const int c = UpChunks<UP>::first(b) + t;
                            acc = mfma(fAx[UpChunks<UP>::cls0(b) + t * UpChunks<UP>::step()], tpk[c >> 1][c & 1], acc);
                        }
                    }
                    return acc;
                };

                // ---- stages B, activation, C over the four 32-column blocks of u, software-pipelined: the MFMAs of block
                //      b + 1 are issued before the (vector-pipe) activation of block b, stage C of block b after it --------
                f32x16 accW[G::OBX];
                #pragma unroll
                for (int bo = 0; bo < G::OBX; bo++) accW[bo] = zero16();
                f32x16 accU = stage_b(0);
                #pragma unroll
                for (int b = 0; b < 4; b++)
                {
                    f32x16 accUn;
                    if (b < 3) accUn = stage_b(b + 1);
                    uint32_t zp[8];
                    if (LVG_WABL & 4) { for (int i = 0; i < 8; i++) { half2v t; t[0] = (_Float16)accU[2 * i]; t[1] = (_Float16)accU[2 * i + 1]; zp[i] = h2_bits(t); } }
                    else act_block<MODE, SLOPEMAX, CLAMP, G::LUTN>(accU, zp, mdw[b], M8[2 * b], M8[2 * b + 1], K);
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
                            if (cls >= 0 && cls < G::NDX) accW[bo] = mfma(fDx[cls], z, accW[bo]);
                        }
                    }
                    if (b < 3) accU = accUn;
                }
                LVG_TICK(5);

                // ---- W[ox][v] -> WL[v][ox]: registers 4q .. 4q + 3 are four consecutive ox -------------------------------
                #pragma unroll
                for (int bo = 0; bo < G::OBX; bo++)
                    #pragma unroll
                    for (int q = 0; q < 4; q++)
                    {
                        half4 h;
                        #pragma unroll
                        for (int e = 0; e < 4; e++) h[e] = (_Float16)accW[bo][4 * q + e];
                        *reinterpret_cast<half4*>(WL + (32 * vb + n) * G::SW + 32 * bo + 8 * q + 4 * g) = h;
                    }
                LVG_TICK(6);

            };
            if (MODE == LVG_SIGNS_READ) row_block(BoolC<true>(), BoolC<false>());
            else if (noClamp)           row_block(BoolC<true>(), BoolC<false>());
            else                        row_block(BoolC<true>(), BoolC<true>());   // (slope > 1: the launcher hands the call to the round-2 kernel)
            // ---- WRITE mode: this row block's mask -> global, only the part this tile owns ----------------------------
            if (MODE == LVG_SIGNS_WRITE)
            {
                // lanes (n, g = 0) / (n, g = 1) hold the even / odd bytes of the 8 mask bytes of a block: exchange, interleave,
                // and every lane owns 4 consecutive bytes (8 b + 4 g ..) of row n
                const uint32_t selIl = g ? 0x07030602u : 0x05010400u;
                #pragma unroll
                for (int b = 0; b < 4; b++)
                {
                    const uint2v sw = __builtin_amdgcn_permlane32_swap(mdw[b], mdw[b], false, false);    // [0] = even bytes, [1] = odd bytes of the row
                    *reinterpret_cast<uint32_t*>(ML + n * G::SM + 8 * b + 4 * g) = __builtin_amdgcn_perm(sw[1], sw[0], selIl);
                }
                const int row = lane >> 1, half = lane & 1;
                const uint32_t* m = reinterpret_cast<const uint32_t*>(ML + row * G::SM + 16 * half);
                uint32_t wds[4] = {m[0], m[1], m[2], m[3]};
                const int uStart = outX0 * DOWN, upY0 = outY0 * DOWN + 32 * vb;   // (sign offsets are 0 when writing)
                const int signByte0 = uStart >> 2;
                const int ownRows = ((tileY == p.tilesY - 1) ? G::V : TH * DOWN) - 32 * vb;
                const int sy = upY0 + row;
                const bool lastX = tileX == p.tilesX - 1;
                // bytes of this lane's 16 the tile owns and the plane has; bytes at and beyond swLimit carry no pixels: 0
                const int b0 = signByte0 + 16 * half;
                const int nOwn = min(16, (lastX ? p.sWBytes : signByte0 + (TW * DOWN) / 4) - b0);
                const int nPix = p.swLimit - b0;
                if (lastX)
                {
                    #pragma unroll
                    for (int d = 0; d < 4; d++)
                    {
                        const int nv = nPix - 4 * d;
                        if (nv < 4) wds[d] = nv <= 0 ? 0u : (wds[d] & ((1u << (8 * nv)) - 1u));
                    }
                }
                const bool dwAligned = (signByte0 & 3) == 0;                 // (rows of the mask plane are dword aligned)
                if (row < ownRows && sy < p.sH && !(LVG_WABL & 16))
                {
                    uint8_t* srow = p.s + (int64_t)cur.plane * ((int64_t)p.sH * p.sWBytes) + (uint32_t)(sy * p.sWBytes) + b0;
                    if (nOwn == 16 && dwAligned) *reinterpret_cast<uint4*>(srow) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
                    else
                    {
                        #pragma unroll
                        for (int d = 0; d < 4; d++)
                        {
                            if (4 * d + 4 <= nOwn && dwAligned) *reinterpret_cast<uint32_t*>(srow + 4 * d) = wds[d];
                            else
                            {
                                #pragma unroll
                                for (int kb = 0; kb < 4; kb++)
                                    if (4 * d + kb < nOwn) srow[4 * d + kb] = (uint8_t)(wds[d] >> (8 * kb));
                            }
                        }
                    }
                    // the row padding can reach beyond the 32 bytes of the tile: define it as 0 too
                    if (lastX && half == 1)
                    {
                        #pragma unroll 1
                        for (int kb = 16; kb < 24 && b0 + kb < p.sWBytes; kb++) srow[kb] = 0;
                    }
                }
            }
            LVG_TICK(7);
        }

        // ---- stage D: Y^T[ox][oy] = W^T * D_y^T: lanes = output rows, registers 4q .. 4q + 3 = four consecutive ox ------
        if (!(LVG_WABL & 8))
        {
            f32x16 accY[G::OBX];
            #pragma unroll
            for (int bo = 0; bo < G::OBX; bo++) accY[bo] = zero16();
            #pragma unroll
            for (int c = 0; c < G::NDY; c++)
            {
                const half8 fdy = *reinterpret_cast<const half8*>(tabDy + c * 512 + lane * 8);
                #pragma unroll
                for (int bo = 0; bo < G::OBX; bo++)
                    accY[bo] = mfma(lds_tr_operand(WL, G::SW, 16 * c, 32 * bo, lane), fdy, accY[bo]);
            }
            LVG_TICK(8);
            // output rows -> LDS (row n = lane n; W has been read) -> 16-byte vectors -> global
            #pragma unroll
            for (int bo = 0; bo < G::OBX; bo++)
                #pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    uint2 v;
                    v.x = pack_pair<T>(accY[bo][4 * q], accY[bo][4 * q + 1]);
                    v.y = pack_pair<T>(accY[bo][4 * q + 2], accY[bo][4 * q + 3]);
                    *reinterpret_cast<uint2*>(YL + n * G::YP + (32 * bo + 8 * q + 4 * g) * 2) = v;
                }
            const int rowsHere = min(TH, p.yh - outY0);                      // output rows / columns of this tile that exist (uniform)
            const int colsHere = min(TW, p.yw - outX0);
            char* ypl = (char*)((T*)p.y + ((int64_t)cur.nb * p.ys[0] + (int64_t)cur.ch * p.ys[1]));
            const uint32_t ybase = (uint32_t)(outY0 * (int)p.ys[2] + outX0 * (int)p.ys[3]) * 2u;
            #pragma unroll
            for (int i = 0; i < G::NSTORE; i++)
            {
                const int srow = st_row(i), scol = st_col(i, srow);
                const v4u v = *reinterpret_cast<const v4u*>(YL + (scol >= 0 ? srow * G::YP + scol * 2 : 0));
                const uint32_t yoff = ybase + stOff[i];
                if (scol >= 0 && srow < rowsHere && !(LVG_WABL & 2))
                {
                    if (fastStore && scol + 8 <= colsHere) *reinterpret_cast<v4u*>(ypl + yoff) = v;
                    else
                    {
                        #pragma unroll
                        for (int e = 0; e < 8; e++)
                            if (scol + e < colsHere) *reinterpret_cast<uint16_t*>(ypl + yoff + (uint32_t)e * yColB) = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
        }
        LVG_TICK(9);
        cur = nxt; nxt = nxt2; tmaxCur = tmaxNext;
    }
#ifdef LVG_TIMING
    if (lane == 0)
    {
        uint32_t* o = g_flwTiming + (((int)blockIdx.x * kWaves + w) & 4095) * 16;
        for (int i = 0; i < 12; i++) o[i] = tAcc[i];
        o[12] = (uint32_t)(tileEnd > tileBeg ? (tileEnd - tileBeg + kWaves - 1) / kWaves : 0);
    }
#endif
}

template <class T, int UP, int DOWN, int FU, int FD, int VB, int TW, int TH>
int launch_wave(FlreluArgs& p, int mode, hipStream_t stream)
{
    typedef WGeo<UP, DOWN, FU, FD, VB, TW, TH, LVG_SIGNS_READ> GR;
    typedef WGeo<UP, DOWN, FU, FD, VB, TW, TH, LVG_SIGNS_WRITE> GW;
    typedef WGeo<UP, DOWN, FU, FD, VB, TW, TH, LVG_SIGNS_NONE> GN;
    p.tilesX = (p.yw + TW - 1) / TW;
    p.tilesY = (p.yh + TH - 1) / TH;
    const int64_t tiles = (int64_t)p.tilesX * p.tilesY * p.n * p.c;
    LVG_REQUIRE(tiles <= 0x7fffffffLL, "filtered_lrelu: too many tiles for one launch");
    // leaky ReLU as max(x, slope * x) needs slope <= 1 (forward modes; the READ mode multiplies by a looked-up factor)
    if (mode != LVG_SIGNS_READ && !(p.slope <= 1.0f)) return LVG_ERR_UNSUPPORTED;
    // lane offsets inside a plane are 32-bit byte offsets
    if (((int64_t)p.xh * p.xs[2] + (int64_t)p.xw * p.xs[3]) * 2 >= 0x7fffffffLL || ((int64_t)p.yh * p.ys[2] + (int64_t)p.yw * p.ys[3]) * 2 >= 0x7fffffffLL ||
        (int64_t)p.sH * p.sWBytes >= 0x7fffffffLL || p.xs[2] < 0 || p.xs[3] < 0 || p.ys[2] < 0 || p.ys[3] < 0)
        return LVG_ERR_UNSUPPORTED;
    static int cus[64] = {0};
    int dev = 0; (void)hipGetDevice(&dev);
    int ncu = cus[dev & 63];
    if (ncu == 0)
    {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev & 63] = ncu;
    }
    // Persistent waves: two workgroups of four waves per CU, every wave walks over a contiguous range of tiles.
    static const int wgPerCu = []() { const char* e = getenv("LVG_FLRELU_WG_PER_CU"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8 / kWaves; }();   // (measurements)
    const int64_t maxGrid = (int64_t)ncu * wgPerCu, wantGrid = (tiles + kWaves - 1) / kWaves;
    const unsigned grid = (unsigned)(wantGrid < maxGrid ? wantGrid : maxGrid);
    const size_t lds = mode == LVG_SIGNS_READ ? GR::LDS_BYTES : (mode == LVG_SIGNS_WRITE ? GW::LDS_BYTES : GN::LDS_BYTES);
    // Unit pixel stride: the input moves as 16-byte vectors (any alignment); other strides: element loads.
    const bool fast = p.xs[3] == 1;
    #define LVG_WAVE_LAUNCH(M, F) do { \
        static std::atomic<uint64_t> attr_done{0};          /* one bit per device (the attribute is per device) */ \
        const uint64_t bit_ = 1ull << (dev & 63); \
        if (!(attr_done.load(std::memory_order_acquire) & bit_)) { \
            hipError_t e = hipFuncSetAttribute((const void*)filtered_lrelu_wave_kernel<T, UP, DOWN, FU, FD, VB, TW, TH, M, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) { lvg_set_error("filtered_lrelu: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return LVG_ERR_LAUNCH; } \
            attr_done.fetch_or(bit_, std::memory_order_release); } \
        if (getenv("LVG_FLRELU_DEBUG")) { int nb = -1; hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)filtered_lrelu_wave_kernel<T, UP, DOWN, FU, FD, VB, TW, TH, M, F>, kThreads, lds); \
            fprintf(stderr, "filtered_lrelu_wave: up %d down %d mode %d fast %d: grid %u x %d threads, lds %zu, tiles %lld, occupancy query %d blocks/CU (%s)\n", UP, DOWN, M, (int)F, grid, kThreads, lds, (long long)tiles, nb, hipGetErrorString(e2)); } \
        hipLaunchKernelGGL((filtered_lrelu_wave_kernel<T, UP, DOWN, FU, FD, VB, TW, TH, M, F>), dim3(grid), dim3(kThreads), lds, stream, p, (int)tiles); } while (0)
    if (mode == LVG_SIGNS_WRITE)     { if (fast) LVG_WAVE_LAUNCH(LVG_SIGNS_WRITE, true); else LVG_WAVE_LAUNCH(LVG_SIGNS_WRITE, false); }
    else if (mode == LVG_SIGNS_READ) { if (fast) LVG_WAVE_LAUNCH(LVG_SIGNS_READ, true);  else LVG_WAVE_LAUNCH(LVG_SIGNS_READ, false); }
    else                             { if (fast) LVG_WAVE_LAUNCH(LVG_SIGNS_NONE, true);  else LVG_WAVE_LAUNCH(LVG_SIGNS_NONE, false); }
    #undef LVG_WAVE_LAUNCH
    return lvg_check_launch("filtered_lrelu_wave_kernel");
}

template <class T>
int run_wave(FlreluArgs& p, int cfg, int mode, hipStream_t stream)
{
    switch (cfg)
    {
        case LVG_FLRELU_CFG_U2D2: return launch_wave<T, 2, 2, 12, 12, 2, 56, 26>(p, mode, stream);
        case LVG_FLRELU_CFG_U4D2: return launch_wave<T, 4, 2, 24, 12, 2, 56, 26>(p, mode, stream);
        case LVG_FLRELU_CFG_U2D4: return launch_wave<T, 2, 4, 12, 24, 3, 26, 19>(p, mode, stream);
    }
    return LVG_ERR_UNSUPPORTED;
}

} // namespace

int lvg_flrelu_wave_launch(FlreluArgs& p, int cfg, int mode, int dtype, hipStream_t stream)
{
    if (dtype == LVG_F16)  return run_wave<f16_t>(p, cfg, mode, stream);
    if (dtype == LVG_BF16) return run_wave<bf16_t>(p, cfg, mode, stream);
    return LVG_ERR_UNSUPPORTED;
}
