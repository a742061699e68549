// filtered_lrelu_band.hip -- fused bias -> up-FIR -> gain -> leaky ReLU -> clamp -> down-FIR for float16 tensors on gfx950
// (round 5): the banded-matrix arithmetic of filtered_lrelu_wave.hip with the MEMORY STRUCTURE turned around. Where the
// wave kernel cut a plane into 56 x 26 output tiles (112-byte row pieces at a 296-byte pitch, 70 x 38 input halos, the loader's
// validity / bias / conversion arithmetic on the vector pipe), here
//   * one WORKGROUP owns one plane at a time (persistent over a contiguous range of planes) and walks down it in steps of
//     32 up-sampled rows ("v-blocks"); its waves are the plane's column strips (TW outputs = 128 up-sampled columns each);
//   * whole rows of the plane enter a shared LDS ring through LDS-DMA (buffer_load_dwordx4 ... lds), 16 input rows = one
//     K-chunk of the vertical up stage per slot: every global read is a contiguous run of the plane, no halo is fetched
//     twice from HBM or L2, and the loader costs no vector-pipe instruction (rows outside the plane come back as zeros from
//     the buffer bounds check; the next row's first pixels behind a row's end are zeroed in LDS by the lanes that fetched them);
//   * the bias rides in a spare K slot: row 15 of the last K-chunk of a v-block never meets a non-zero tap, so the lanes that
//     fetch it read a per-plane bias row (b inside the image, 0 outside) instead and the A_y fragment carries sum(taps over
//     rows inside the image) there -- (x + b) inside, 0 in the padding, exact at every edge, no extra instruction;
//   * the vertical down stage streams: W of one v-block (32 rows) lives in wave-private LDS, stage D accumulates output
//     blocks of 32 rows across v-blocks in registers; no vertical halo is recomputed (the wave kernel recomputed 24 %);
//   * the two down stages share ONE fragment table (W's rows are stored in the k order of an MFMA result, so D_y takes the
//     images of D_x); the mask offsets of the READ mode are a bit shift of the mask stream, not a shift of the geometry;
//   * one barrier per v-block keeps the ring consistent. A wave's DMA pieces are an iteration old when it waits for them, its
//     stores are issued at the END of an iteration and are NOT waited for (s_waitcnt vmcnt(number of stores issued since)).
// Semantics: exactly those of filtered_lrelu.hip / filtered_lrelu_wave.hip (reference torch_utils/ops/filtered_lrelu.cu:139-1099,
// filtered_lrelu.cpp:16-210), including the 2-bit sign / clamp mask (write, and read with offsets).
// Arithmetic: f16 operands, f32 accumulation; T', Z and W are rounded to f16 between stages, taps are rounded to f16; the bias
// is added in the f32 accumulator of stage A (the wave kernel rounded x + b to f16 first).
// What it does not take (the launcher answers LVG_ERR_UNSUPPORTED and the caller falls back to the wave kernel): bfloat16,
// non-contiguous input planes, odd plane widths, strided output rows, slope > 1 in the forward modes.
// Algorithmic HBM bytes: (N_in + N_out) * 2 + mask bytes; see DESIGN.md.

#include "lvg_common.h"
#include "filtered_lrelu_args.h"
#include <atomic>
#include <stdlib.h>

#ifndef LVG_BABL
#define LVG_BABL 0           // ablation builds only (results are WRONG): 2 no y stores, 4 no activation math, 16 no mask stores, 32 no matrix products, 64 no input DMA, 128 no barriers, 256 no wait for the DMA pieces
#endif
#define LVG_WABL (LVG_BABL & 32)
#ifndef LVG_BAND_SCHED_FENCE
#define LVG_BAND_SCHED_FENCE 1
#endif
#ifndef LVG_BAND_SLOTS
#define LVG_BAND_SLOTS 3     // ring slots (K-chunks of 16 input rows): two in use, the rest in flight
#endif
#ifndef LVG_BAND_SWP
#define LVG_BAND_SWP 0       // round 6 experiment: software pipeline over the column blocks with the matrix products spread between the vector instructions -- measured no faster (profiles/r06_band_swp1.log): off
#endif
#ifndef LVG_BAND_SWP_GROUPS
#define LVG_BAND_SWP_GROUPS 6
#endif
#ifndef LVG_BAND_SWP_VALU
#define LVG_BAND_SWP_VALU 8
#endif
#ifndef LVG_BAND_PIPE
#define LVG_BAND_PIPE 1      // stage B of column block b + 1 issued before the activation of block b (16 more registers)
#endif

namespace {

#include "flrelu_mfma_common.h"

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <bool B> struct BoolC { static constexpr bool value = B; };
template <int V> struct IntC { static constexpr int value = V; };

constexpr int kSlots = LVG_BAND_SLOTS;          // ring slots (K-chunks of 16 input rows): two in use, one in flight
constexpr int kAyRows = 12;        // rows of the bias-coefficient table: v-blocks that touch rows above the image, one for the interior, those that touch rows below
constexpr int kMaxPieces = 4;      // DMA instructions of one chunk a wave may have to issue
constexpr int kRingPad = 512;      // zeros behind the ring: transpose reads of the last slot's last rows run past the row's end

struct BandArgs
{
    FlreluArgs a;
    int planes;
    int ns;                // strips = waves per workgroup
    int nvb, nch;          // v-blocks and K-chunks (16 input rows) per plane
    int LR;                // 16-byte pieces per ring row
    int np;                // DMA instructions per chunk = LR / 4
    int mp;                // margin pieces in front of every ring row (out of range: zeros): ring column of image column x = x + 8 mp
    int ef;                // strips start ef columns (0..3) before their first input column, so that their ring column is a multiple of 4
    int ldsCol0;           // ring column (halves) of strip 0's first column
    int ayTop, ayBot;      // v-blocks < ayTop / >= ayBot touch rows outside the image: own rows of the bias-coefficient table
    int inX0, inY0;        // first input column of strip 0 / first input row of chunk 0 (may be negative)
    int phX, phY;
};

template <int UP, int DOWN, int FU, int FD, int TW, int MODE>
struct BGeo
{
    static constexpr int KU     = FU / UP;
    static constexpr int IN_NX  = (UP - 1 + kU - 1) / UP + KU + 3;         // input columns a strip touches (+3: fragment shift)
    static constexpr int CH_X   = ((IN_NX - 1) >> 4) + 1;                   // 16-chunks of input columns stage B can read
    static constexpr int IN_BLK = wdiv_up(CH_X, 2);                         // 32-blocks of input columns (stage A's M)
    static constexpr int SPITCH = TW * DOWN / UP;                           // input columns between neighbouring strips
    static constexpr int OBX    = wdiv_up(TW, 32);
    static constexpr int SW     = ((TW + 3) / 8) * 8 + 4;                   // W row stride (halves): a multiple of 4 that is 4 mod 8 (8-byte column pieces of 16 rows on distinct banks)
    static constexpr int NUC    = (UP == 2) ? 2 : (UP == 4 ? 3 : 1);
    static constexpr int CPB    = 2 * DOWN;                                 // K-chunks of v per block of 32 output rows
    static constexpr int NDC    = ((31 * DOWN + FD - 1) >> 4) + 1;          // classes of a down stage (shared by D_x and D_y)
    static constexpr int SPILL  = NDC - CPB;                                // leading chunks of a block that still feed the block above
    static constexpr int NVY    = wdiv_up(TW, 8);                           // output: 16-byte vectors per row piece
    static constexpr int YP0    = (2 * TW + 15) / 16 * 16;
    static constexpr int YP     = YP0 + (((YP0 / 4) % 8 == 4) ? 0 : (((YP0 + 16) / 4) % 8 == 4 ? 16 : 32));   // staging row pitch (bytes): a multiple of 16, 4 mod 8 dwords
    static constexpr int NSTORE = wdiv_up(32 * NVY, 64);
    static constexpr bool HAS_M = MODE != LVG_SIGNS_NONE;
    static constexpr int SM     = 40;
    static constexpr int TAPS   = (FU + FD + 3) / 4 * 4;
    static constexpr int LUTN   = MODE != LVG_SIGNS_READ ? 0 : 171;         // codes 0..2 per pixel: bytes <= 0xAA
    // LDS map (bytes): taps | READ look-up table | down-stage fragment images | bias-coefficient table | bias rows | ring | pad | per-wave regions
    static constexpr int OFF_TAPS = 0;
    static constexpr int OFF_LUT  = TAPS * 4;
    static constexpr int OFF_TAB  = OFF_LUT + (LUTN * 8 + 15) / 16 * 16;    // D fragment images (k permuted like an MFMA result), 1 KiB per class
    static constexpr int OFF_AY   = OFF_TAB + NDC * 1024;                   // [kAyRows][64] halves: element (j = 7) of the last chunk's A_y fragment
    static constexpr int OFF_XB   = OFF_AY + kAyRows * 128;                 // two bias rows (one per plane parity), row pitch bytes each
    static constexpr int W_BYTES  = (32 * SW * 2 > 32 * YP ? 32 * SW * 2 : 32 * YP);   // W rows of one v-block; the output staging rows alias them
    static constexpr int M_BYTES  = HAS_M ? 32 * SM : 0;
    static constexpr int WAVE_BYTES = W_BYTES + M_BYTES;
    static_assert(FU % UP == 0 && FD % DOWN == 0, "filter sizes must be multiples of the rates");
    static_assert((TW * DOWN) % 4 == 0 && (TW * DOWN) % UP == 0 && SPITCH % 4 == 0, "strip origins must keep the mask byte, the up-sampling phase and the 8-byte LDS alignment");
    static_assert((TW - 1) * DOWN + FD - 1 < kU, "strip does not fit its up-sampled block");
    static_assert(TW <= 64 && TW % 2 == 0, "stage D: one or two blocks of output columns");
    static_assert(SPILL >= 0 && SPILL <= 2 && CPB % 2 == 0, "streaming stage D: the spill chunks of a block lie in one v-block");
    static_assert(SW % 8 == 4 && SW >= TW && YP >= 2 * TW && YP % 16 == 0 && (YP / 4) % 8 == 4, "wave-private rows");
    static_assert(OFF_TAB % 16 == 0 && OFF_AY % 16 == 0 && OFF_XB % 16 == 0 && W_BYTES % 16 == 0 && WAVE_BYTES % 16 == 0, "alignment");
    static_assert(NSTORE <= 4, "store instructions of a block");
};

// s_waitcnt vmcnt(k): everything this wave issued except its k youngest vector-memory operations is complete. k is what the
// caller KNOWS to be younger (a lower bound): a smaller immediate waits for more, never for less.
__device__ __forceinline__ void wait_vm_all_but(int k)
{
    if (k <= 0)      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (k == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (k == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (k == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (k == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (k == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (k < 8)  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else             asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ void lds_barrier()
{
    if (LVG_BABL & 128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS-DMA of 16 bytes per lane through a raw buffer (offsets outside [0, num_records) deliver zeros): LDS address = ldsPiece
// (wave-uniform, via M0) + lane * 16. Inline assembly: the compiler would order every later LDS read behind a DMA builtin with
// s_waitcnt vmcnt(0), i.e. wait for the piece it has just requested.
__device__ __forceinline__ void dma16_buf(v4i rsrc, uint32_t laneOff, uint32_t ldsPiece)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(laneOff), "s"(rsrc), "s"(ldsPiece) : "memory");
}

template <int UP, int DOWN, int FU, int FD, int TW, int MODE, int WPS>
__global__ __launch_bounds__(512, WPS) void filtered_lrelu_band_kernel(BandArgs q)
{
    typedef BGeo<UP, DOWN, FU, FD, TW, MODE> G;
    const FlreluArgs& p = q.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float*    taps  = reinterpret_cast<float*>(smem + G::OFF_TAPS);
    _Float16* tabD  = reinterpret_cast<_Float16*>(smem + G::OFF_TAB);
    _Float16* tabAy = reinterpret_cast<_Float16*>(smem + G::OFF_AY);
    const int tid = threadIdx.x, lane = tid & 63, w = sgpr(tid >> 6);
    const int n = lane & 31, g = lane >> 5;
    const int pitchB = q.LR * 16, slotBytes = 16 * pitchB;
    const int offRing = G::OFF_XB + 2 * pitchB;
    const int offWave = offRing + kSlots * slotBytes + kRingPad;
    unsigned char* xbRows = smem + G::OFF_XB;
    unsigned char* ring = smem + offRing;
    unsigned char* wv = smem + offWave + w * G::WAVE_BYTES;
    _Float16* WL = reinterpret_cast<_Float16*>(wv);                         // W [32][SW], rows in the k order of an MFMA result
    unsigned char* YL = wv;                                                 // output staging rows [32][YP] (alias W)
    unsigned char* ML = wv + G::W_BYTES;                                    // mask rows of one v-block [32][SM]
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- once per workgroup: taps, zeroed ring / bias rows / wave regions, tables -------------------------------------
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
    for (int i = tid; i < (offWave + q.ns * G::WAVE_BYTES - G::OFF_XB) / 4; i += (int)blockDim.x) reinterpret_cast<uint32_t*>(smem + G::OFF_XB)[i] = 0u;
    if (MODE == LVG_SIGNS_READ)
    {
        uint32_t* lut = reinterpret_cast<uint32_t*>(smem + G::OFF_LUT);
        const _Float16 one = (_Float16)1.0f, sl = (_Float16)p.slope, zero = (_Float16)0.0f;
        for (int e = tid; e < G::LUTN; e += (int)blockDim.x)
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

    const int phX = q.phX, phY = q.phY;
    const float scale = (float)(UP * UP) * p.gain;
    for (int e = tid; e < G::NDC * 512; e += (int)blockDim.x)
        tabD[e] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 2, e >> 9, (e & 511) >> 3, e & 7, phX, phY, 0, scale);
    // Element j = 7 of the A_y fragment of every v-block's LAST K-chunk: lanes g = 0 keep the band's coefficient (k = 7), lanes g = 1
    // (k = 15: never inside the band) get the sum of the taps that meet rows inside the image -- the factor of the bias row.
    const int ayRows = q.ayTop + 1 + (q.nvb - q.ayBot);
    for (int e = tid; e < ayRows * 64; e += (int)blockDim.x)
    {
        const int r = e >> 6, l = e & 63;
        const int b = r < q.ayTop ? r : (r == q.ayTop ? q.ayTop : q.ayBot + (r - q.ayTop - 1));
        const int last = UpChunks<UP>::count(b) - 1, cls = UpChunks<UP>::cls0(b) + last * UpChunks<UP>::step();
        float v;
        if ((l >> 5) == 0) v = frag_elem<UP, DOWN, FU, FD>(taps, 0, cls, l, 7, phX, phY, 0, scale);
        else
        {
            // row l & 31 of v-block b: taps t of phase (UP - 1 - m % UP) meet input rows  inY0 + 32 b / UP + m / UP + t
            const int m = (l & 31) + phY, i0 = m / UP;
            v = 0.0f;
            for (int t = 0; t < G::KU; t++)
            {
                const int row = q.inY0 + (32 * b) / UP + i0 + t;
                if (row >= 0 && row < p.xh) v += taps[(UP - 1 - m % UP) + t * UP];
            }
        }
        tabAy[e] = (_Float16)v;
    }
    __syncthreads();

    // ---- band fragments of the up stages in registers: A_y (vertical), A_x (horizontal, shifted by ef columns) ------------
    half8 fAy[G::NUC], fAx[G::NUC];
    #pragma unroll
    for (int c = 0; c < G::NUC; c++)
        #pragma unroll
        for (int j = 0; j < 8; j++)
        {
            fAy[c][j] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 0, c, lane, j, phX, phY, 0, scale);
            fAx[c][j] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 1, c, lane, j, phX, phY, 0, scale, q.ef);
        }
    const uint32_t tabLane = ldsBase + (uint32_t)G::OFF_TAB + (uint32_t)lane * 16u;      // this lane's piece of a D fragment image
    typedef __attribute__((address_space(3))) const half8* lds_h8;
    auto frag_d = [&](int cls) __attribute__((always_inline)) -> half8 { return *(lds_h8)(uintptr_t)(tabLane + (uint32_t)cls * 1024u); };

    // ---- activation constants ----------------------------------------------------------------------------------------
    ActConst K;
    {
        const _Float16 slope_h = (_Float16)p.slope;
        K.slope2[0] = slope_h; K.slope2[1] = slope_h;
        const _Float16 clamp_h = (_Float16)(p.clamp < 65504.0f ? p.clamp : 65504.0f);
        K.clampP[0] = clamp_h; K.clampP[1] = clamp_h; K.clampN[0] = -clamp_h; K.clampN[1] = -clamp_h;
        K.clampBits = h2_bits(K.clampP);
        K.shEven = 8u * (uint32_t)g;
        K.shOdd = 16u + 8u * (uint32_t)g;
        K.lutBase = ldsBase + (uint32_t)G::OFF_LUT;
    }

    // No pre-activation of a v-block can exceed  scale * l1(horizontal up taps per phase) * max |T'|  in magnitude (T' = the
    // vertically up-sampled rows stage A leaves in its accumulators; leaky ReLU with slope <= 1 only shrinks it): v-blocks whose
    // |T'| stays below clamp / that factor (5 % margin for the f16 roundings) skip the clamp and the "clamped" flag arithmetic.
    float tLimit = 0.0f;
    {
        float l1 = 0.0f;
        for (int ph = 0; ph < UP; ph++)
        {
            float a = 0.0f;
            for (int t = ph; t < FU; t += UP) a += fabsf(taps[t]);
            l1 = fmaxf(l1, a);
        }
        tLimit = p.clamp / (scale * l1 * 1.05f + 1e-30f);
        if (!(p.slope <= 1.0f) || !(tLimit > 0.0f)) tLimit = 0.0f;         // (never proven)
    }

    // ---- planes of this workgroup (x, y and the mask are whole contiguous planes: pointers advance plane by plane) ----------
    const int planeBeg = (int)((int64_t)q.planes * blockIdx.x / gridDim.x), planeEnd = (int)((int64_t)q.planes * (blockIdx.x + 1) / gridDim.x);
    const int nPlanes = planeEnd - planeBeg;
    if (nPlanes <= 0) return;
    const int totalChunks = nPlanes * q.nch;
    const uint32_t rowBytes = (uint32_t)p.xw * 2u, planeBytes = (uint32_t)p.xh * rowBytes;
    const uint32_t yRowB = (uint32_t)p.yw * 2u;
    const uint64_t yPlaneB = (uint64_t)p.yh * yRowB, sPlaneB = (uint64_t)p.sH * (uint64_t)p.sWBytes;

    // ---- input DMA: chunk = 16 rows x LR pieces of 16 bytes; this wave issues pieces w, w + ns, ... (1 KiB each). Lane ->
    //      (row, piece): margin pieces and pieces behind the row are out of range (zeros). ------------------------------------
    uint32_t dOff[kMaxPieces];              // byte offset from the chunk's first row, or "out of range"
    uint32_t patchR = 0;                    // bit i: this lane's piece of DMA instruction i ends with the next row's first pixels
    int nMine = 0;
    const int lastPiece = q.mp + (((int)p.xw - 1) >> 3);                    // piece that holds the row's last pixel
    const int nGarbR = (8 * (lastPiece - q.mp + 1) - (int)p.xw) >> 1;       // dwords at the end of that piece that belong to the next row (0..3)
    #pragma unroll
    for (int i = 0; i < kMaxPieces; i++)
    {
        const int piece = w + i * q.ns;
        dOff[i] = 0xfffffff0u;
        if (piece < q.np)
        {
            nMine = i + 1;
            const int idx = piece * 64 + lane, row = idx / q.LR, col = idx - row * q.LR;
            const bool any = col >= q.mp && col <= lastPiece;
            dOff[i] = any ? (uint32_t)(row * (int)rowBytes + 16 * (col - q.mp)) : 0xfffffff0u;
            if (col == lastPiece && nGarbR > 0) patchR |= 1u << i;
        }
    }
    nMine = sgpr(nMine);
    // next chunk to issue: flat index, chunk of its plane, ring slot, byte offset of its first row, the plane's buffer descriptor
    int gIssue = 0, issChunk = 0, issSlot = 0;
    uint32_t issRowBase = (uint32_t)(q.inY0 * (int)rowBytes);               // (negative rows wrap: out of range)
    uint64_t issPlanePtr = (uint64_t)(uintptr_t)p.x + (uint64_t)planeBeg * planeBytes;
    const uint32_t myPieceLds = ldsBase + (uint32_t)offRing + (uint32_t)w * 1024u;
    auto issue_chunk = [&]() __attribute__((always_inline))
    {
        v4i rsrc;
        rsrc[0] = sgpr((int)(uint32_t)issPlanePtr); rsrc[1] = sgpr((int)(uint32_t)((issPlanePtr >> 32) & 0xffffu)); rsrc[2] = sgpr((int)planeBytes); rsrc[3] = 0x00020000;
        const uint32_t slotLds = myPieceLds + (uint32_t)issSlot * (uint32_t)slotBytes;
        if (!(LVG_BABL & 64))
        {
            #pragma unroll
            for (int i = 0; i < kMaxPieces; i++)
                if (i < nMine)
                {
                    const uint32_t vo = dOff[i] == 0xfffffff0u ? 0xfffffff0u : issRowBase + dOff[i];
                    dma16_buf(rsrc, vo, (uint32_t)sgpr((int)(slotLds + (uint32_t)(i * q.ns) * 1024u)));
                }
        }
        ++gIssue;
        issSlot = issSlot == kSlots - 1 ? 0 : issSlot + 1;
        issRowBase += 16u * rowBytes;
        if (++issChunk == q.nch) { issChunk = 0; issRowBase = (uint32_t)(q.inY0 * (int)rowBytes); issPlanePtr += planeBytes; }
    };
    // the next row's pixels at the end of the pieces this wave fetched into `count` slots from `slot0` on (landed) -> zeros
    auto patch_slots = [&](int slot0, int count) __attribute__((always_inline))
    {
        if (nGarbR == 0) return;
        int sl = slot0;
        for (int k = 0; k < count; k++)
        {
            unsigned char* slot = ring + sl * slotBytes + w * 1024 + lane * 16;
            #pragma unroll
            for (int i = 0; i < kMaxPieces; i++)
                if (i < nMine && ((patchR >> i) & 1u))
                {
                    unsigned char* pc = slot + (i * q.ns) * 1024;
                    *reinterpret_cast<uint32_t*>(pc + 12) = 0u;
                    if (nGarbR >= 2) *reinterpret_cast<uint32_t*>(pc + 8) = 0u;
                    if (nGarbR >= 3) *reinterpret_cast<uint32_t*>(pc + 4) = 0u;
                }
            sl = sl == kSlots - 1 ? 0 : sl + 1;
        }
    };

    // ---- this wave's strip: transpose-read lane offsets into a ring slot; the lanes that would fetch row 15 read the bias row ----
    const int colOrigin = q.ldsCol0 + w * G::SPITCH;                        // ring column (halves) of the strip's first column; multiple of 4
    const int hgrp = (lane >> 4) & 1, s16 = lane & 15;
    const uint32_t laneA = ldsBase + (uint32_t)offRing + (uint32_t)((8 * g + (s16 >> 2)) * pitchB + (colOrigin + 16 * hgrp + 4 * (s16 & 3)) * 2);
    const bool row15 = g == 1 && (s16 >> 2) == 3;
    const uint32_t laneXb = ldsBase + (uint32_t)G::OFF_XB + (uint32_t)((colOrigin + 16 * hgrp + 4 * (s16 & 3)) * 2);
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    auto tr_read = [&](uint32_t ldsAddr) __attribute__((always_inline)) -> short4v { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)ldsAddr); };
    // W rows in the k order of an MFMA result: logical row v (0..15 of a 16-chunk) sits at 8 ((v >> 2) & 1) + (v & 3) + 4 (v >> 3)
    const int nPhys = (n & 16) + 8 * ((n >> 2) & 1) + (n & 3) + 4 * ((n >> 3) & 1);

    // ---- bias row of a plane: b in the columns of the image, 0 elsewhere (written by wave 0, one plane ahead) ---------------
    int biasCh = planeBeg % p.c;                                            // channel of the plane whose bias row is written next
    auto write_bias_row = [&](int parity) __attribute__((always_inline))
    {
        const uint32_t bb = scalar_load_u16((const uint16_t*)p.b + sgpr(biasCh));
        biasCh = biasCh + 1 == p.c ? 0 : biasCh + 1;
        const uint32_t b2 = bb * 0x10001u;
        unsigned char* row = xbRows + parity * pitchB;
        for (int pc = lane; pc < q.LR; pc += 64)
        {
            v4u v;
            #pragma unroll
            for (int d = 0; d < 4; d++)
            {
                const int x0 = 8 * (pc - q.mp) + 2 * d;                      // image column of the dword's low half
                const bool in0 = x0 >= 0 && x0 < p.xw, in1 = x0 + 1 >= 0 && x0 + 1 < p.xw;
                v[d] = (in0 ? (b2 & 0xffffu) : 0u) | (in1 ? (b2 & 0xffff0000u) : 0u);
            }
            *reinterpret_cast<v4u*>(row + pc * 16) = v;
        }
    };

    // ---- output: the staged rows of a block of 32 output rows -> global, every lane ONE 16-byte store per instruction: vector k
    //      of a row holds columns 8 k .. 8 k + 7 of the strip; the row's last vector is moved left so that it ends with the strip's
    //      last column (it overlaps its neighbour: same values). Returns the number of store instructions certainly issued. ---------
    const int outX0 = w * TW;
    const int colsHere = min(TW, p.yw - outX0);                             // (even, >= 8: the launcher checks)
    char* yPlane = (char*)p.y + (uint64_t)planeBeg * yPlaneB + (uint32_t)outX0 * 2u;      // this strip's first column of the current plane
    auto store_block = [&](int oy0, int rows) __attribute__((always_inline)) -> int
    {
        int laneV = lane;
        asm volatile("" : "+v"(laneV));                                      // (keeps the per-lane store geometry out of the loop-invariant registers)
        int issued = 0;
        #pragma unroll
        for (int i = 0; i < G::NSTORE; i++)
        {
            const int vi = laneV + 64 * i, srow = div_small<G::NVY>(vi), k8 = 8 * (vi - srow * G::NVY);
            const int scol = min(k8, colsHere - 8);
            const bool act = srow < rows && k8 < colsHere;
            const unsigned char* src = YL + (srow < 32 ? srow * G::YP + scol * 2 : 0);
            const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
            if ((64 * i) / G::NVY < rows) ++issued;                          // (the instruction's first row exists: at least one lane stores)
            if (act && !(LVG_BABL & 2))
                *reinterpret_cast<uint4*>(yPlane + (uint32_t)(oy0 + srow) * yRowB + (uint32_t)scol * 2u) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        return (LVG_BABL & 2) ? 0 : issued;
    };
    auto stage_rows = [&](const uint32_t (&ypk)[G::OBX * 8]) __attribute__((always_inline))
    {
        #pragma unroll
        for (int bo = 0; bo < G::OBX; bo++)
            #pragma unroll
            for (int qd = 0; qd < 4; qd++)
                if (32 * bo + 8 * qd < TW)
                    *reinterpret_cast<uint2*>(YL + n * G::YP + (32 * bo + 8 * qd + 4 * g) * 2) = make_uint2(ypk[bo * 8 + 2 * qd], ypk[bo * 8 + 2 * qd + 1]);
    };

    // ---- READ mode: 128 bits of one mask row per lane (row = lane >> 1 of the v-block, half = lane & 1), fetched during the
    //      previous v-block as the 5 aligned dwords that cover them. The sign offsets enter as a BIT shift of the stream
    //      (2 bits per pixel): pixel u of the strip has mask coordinate  w TW DOWN + u + sOfsX. ------------------------------------
    uint32_t mraw[5];
    const uint8_t* sPlane = p.s + (uint64_t)planeBeg * sPlaneB;             // mask plane of the current plane
    const int maskX0 = w * (TW * DOWN) + p.sOfsX;
    const int mshift = 8 * ((maskX0 >> 2) & 3) + 2 * (maskX0 & 3);          // <= 30
    auto issue_mask_loads = [&](const uint8_t* spl, int b) __attribute__((always_inline))
    {
        int laneV = lane;
        asm volatile("" : "+v"(laneV));
        const int row = laneV >> 1, half = laneV & 1;
        const int a0 = ((maskX0 >> 2) + 16 * half) & ~3;                     // first aligned dword of this lane's 128 bits (floor: maskX0 may be negative)
        const int sy = 32 * b + p.sOfsY + row;
        const bool rowOk = (uint32_t)sy < (uint32_t)p.sH;
        const uint32_t rowOffB = (uint32_t)(sy * p.sWBytes);
        #pragma unroll
        for (int j = 0; j < 5; j++)
        {
            const int bx = a0 + 4 * j;
            const bool ok = rowOk && bx >= 0 && bx + 4 <= p.sWBytes;
            uint32_t v = *reinterpret_cast<const uint32_t*>(spl + (ok ? rowOffB + (uint32_t)bx : 0u));
            const int nv = p.swLimit - bx;                                   // bytes at and beyond swLimit carry no pixels
            v = !ok || nv <= 0 ? 0u : (nv < 4 ? (v & ((1u << (8 * nv)) - 1u)) : v);
            mraw[j] = v;
        }
    };
    auto stage_mask = [&]() __attribute__((always_inline))
    {
        uint32_t* m = reinterpret_cast<uint32_t*>(ML + (lane >> 1) * G::SM + 16 * (lane & 1));
        #pragma unroll
        for (int d = 0; d < 4; d++) m[d] = __builtin_amdgcn_alignbit(mraw[d + 1], mraw[d], (uint32_t)mshift);
    };

    // ---- the walk: iteration = one v-block of one plane ---------------------------------------------------------------
    if (w == 0) write_bias_row(0);
    if (MODE == LVG_SIGNS_READ) issue_mask_loads(sPlane, 0);
    f32x16 accY[G::OBX];
    #pragma unroll
    for (int bo = 0; bo < G::OBX; bo++) accY[bo] = zero16();
    int curBlock = 0;                                                       // output block (32 rows) accY is accumulating
    int young = 0;                                                          // vector-memory operations issued after the last DMA piece / mask load
    int gFirst = 0, slotFirst = 0;                                          // first K-chunk of the current v-block: flat index, ring slot
    int patchSlot = 0, patchCount = 0;                                      // slots whose pieces were requested during the previous iteration
    int planeChunk0 = 0;                                                    // flat index of the current plane's chunk 0

    for (int pl = 0; pl < nPlanes; pl++)
    {
        #pragma unroll 1
        for (int b = 0; b < q.nvb; b++)
        {
            const int cFirst = (UP == 4) ? (b >> 1) : b, cCount = (UP == 4) ? 1 + (b & 1) : 2;
            while (gFirst < planeChunk0 + cFirst) { ++gFirst; slotFirst = slotFirst == kSlots - 1 ? 0 : slotFirst + 1; }
            // The DMA pieces (and mask dwords) this wave requested during the previous iteration have landed once all but its
            // `young` youngest operations (the stores of that iteration's end) are complete. Zero the next-row pixels in the pieces,
            // meet the other waves: chunks <= gFirst + 1 complete and visible; nobody reads chunk gFirst - 1 any more.
            if (LVG_BABL & 256) {}
            else if (young >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (young >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            patch_slots(patchSlot, patchCount);
            lds_barrier();
            const int issuedBefore = gIssue;
            patchSlot = issSlot;
            while (gIssue <= gFirst + kSlots - 1 && gIssue < totalChunks) issue_chunk();
            patchCount = gIssue - issuedBefore;
            if (min(gFirst + cCount - 1, planeChunk0 + q.nch - 1) >= issuedBefore)
            {
                // (first iteration, or a plane whose last v-block needed its own extra chunk: the data was requested just now)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                patch_slots(patchSlot, patchCount);
                lds_barrier();
                patchCount = 0;
            }
            young = 0;
            if (w == 0 && b == 0 && pl + 1 < nPlanes) write_bias_row((pl + 1) & 1);   // (read from the next plane's first v-block on: >= 1 barrier away)
            if (MODE == LVG_SIGNS_READ) stage_mask();

            // ---- stage A: T'[ic][v] for the 32 rows v of this v-block; K-chunks = ring slots; bias through row 15 of the last ----
            half8 tpk[G::CH_X];
            bool noClamp = false;
            {
                f32x16 accA[G::IN_BLK];
                #pragma unroll
                for (int m = 0; m < G::IN_BLK; m++) accA[m] = zero16();
                const uint32_t xbAddr = laneXb + (uint32_t)((pl & 1) * pitchB);
                const int ayRow = b < q.ayTop ? b : (b < q.ayBot ? q.ayTop : q.ayTop + 1 + (b - q.ayBot));
                const uint32_t ayDw = (uint32_t)reinterpret_cast<const uint16_t*>(tabAy)[ayRow * 64 + lane];
                const int slotSecond = slotFirst == kSlots - 1 ? 0 : slotFirst + 1;
                auto chunk_a = [&](int slot, auto clsC, bool last) __attribute__((always_inline))
                {
                    constexpr int cls = decltype(clsC)::value;
                    const uint32_t slotA = laneA + (uint32_t)slot * (uint32_t)slotBytes;
                    const uint32_t hiA = (last && row15) ? xbAddr : slotA + 4u * (uint32_t)pitchB;
                    half8 fa = fAy[cls];
                    if (last)
                    {
                        uint32_t w3; __builtin_memcpy(&w3, reinterpret_cast<const char*>(&fa) + 12, 4);
                        w3 = (w3 & 0xffffu) | (ayDw << 16);
                        __builtin_memcpy(reinterpret_cast<char*>(&fa) + 12, &w3, 4);
                    }
                    #pragma unroll
                    for (int m = 0; m < G::IN_BLK; m++)
                    {
                        const short4v lo = tr_read(slotA + 64u * m), hi = tr_read(hiA + 64u * m);
                        half8 xt;
                        __builtin_memcpy(&xt, &lo, 8);
                        __builtin_memcpy(reinterpret_cast<char*>(&xt) + 8, &hi, 8);
                        accA[m] = mfma(xt, fa, accA[m]);
                    }
                };
                if (UP == 2) { chunk_a(slotFirst, IntC<0>(), false); chunk_a(slotSecond, IntC<1>(), true); }
                else if (b & 1) { chunk_a(slotFirst, IntC<0>(), false); chunk_a(slotSecond, IntC<(UP == 4 ? 2 : 0)>(), true); }
                else chunk_a(slotFirst, IntC<(UP == 4 ? 1 : 0)>(), true);
                #pragma unroll
                for (int c = 0; c < G::CH_X; c++) tpk[c] = pack_chunk(accA[c >> 1], c & 1);
                if (MODE != LVG_SIGNS_READ)
                {
                    float mx = 0.0f;
                    #pragma unroll
                    for (int c = 0; c < G::CH_X; c++)
                        #pragma unroll
                        for (int r = 0; r < 8; r += 2)
                            mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(accA[c >> 1][8 * (c & 1) + r]), __builtin_fabsf(accA[c >> 1][8 * (c & 1) + r + 1])));
                    noClamp = __builtin_amdgcn_ballot_w64(!(mx < tLimit)) == 0;      // (NaN / inf count as "not below")
                }
            }

            // ---- stages B, activation, C over the four 32-column blocks of u ---------------------------------------------
            uint32_t mdw[4] = {0, 0, 0, 0};
            auto row_block = [&](auto slopeMaxC, auto clampC) __attribute__((always_inline))
            {
                constexpr bool SLOPEMAX = decltype(slopeMaxC)::value, CLAMP = decltype(clampC)::value;
                auto stage_b = [&](int bc) __attribute__((always_inline)) -> f32x16
                {
                    f32x16 acc = zero16();
                    #pragma unroll
                    for (int t = 0; t < 2; t++)
                    {
                        if (t < UpChunks<UP>::count(bc))
                        {
                            const int c = UpChunks<UP>::first(bc) + t;
                            if (c < G::CH_X) acc = mfma(fAx[UpChunks<UP>::cls0(bc) + t * UpChunks<UP>::step()], tpk[c], acc);
                        }
                    }
                    return acc;
                };
                f32x16 accW[G::OBX];
                #pragma unroll
                for (int bo = 0; bo < G::OBX; bo++) accW[bo] = zero16();
                f32x16 accU = stage_b(0);
#if LVG_BAND_SWP
                // Software pipeline over the four column blocks: the scheduling region of block bc holds the vector work of
                // act(bc) next to the matrix products of stage B of block bc + 1 and of stage C of block bc - 1 (neither depends
                // on it), and the group directives at its end ask for ONE product per run of vector instructions -- a wave
                // that issues its products in clumps leaves the matrix pipe idle through every activation (the products of a
                // wave issue in order, 32 cycles apart; tools/probe_issue.hip: product + ~8 vector instructions share 35 cycles).
                auto stage_c = [&](int bc, const uint32_t (&zq)[8]) __attribute__((always_inline))
                {
                    #pragma unroll
                    for (int h = 0; h < 2; h++)
                    {
                        half8 z;
                        __builtin_memcpy(&z, &zq[4 * h], 16);
                        const int c = 2 * bc + h;
                        #pragma unroll
                        for (int bo = 0; bo < G::OBX; bo++)
                        {
                            const int cls = c - 2 * bo * DOWN;
                            if (cls >= 0 && cls < G::NDC) accW[bo] = mfma(frag_d(cls), z, accW[bo]);
                        }
                    }
                };
                uint32_t zq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                #pragma unroll
                for (int bc = 0; bc < 4; bc++)
                {
                    f32x16 accUn;
                    __builtin_amdgcn_sched_barrier(0);
                    if (bc < 3) accUn = stage_b(bc + 1);
                    if (bc > 0) stage_c(bc - 1, zq);
                    uint32_t zp[8];
                    uint32_t mlo = 0, mhi = 0;
                    if (MODE == LVG_SIGNS_READ) { const uint32_t* r = reinterpret_cast<const uint32_t*>(ML + n * G::SM + 8 * bc); mlo = r[0]; mhi = r[1]; }
                    if (LVG_BABL & 4) { for (int i = 0; i < 8; i++) { half2v t; t[0] = (_Float16)accU[2 * i]; t[1] = (_Float16)accU[2 * i + 1]; zp[i] = h2_bits(t); } }
                    else act_block<MODE, SLOPEMAX, CLAMP, G::LUTN>(accU, zp, mdw[bc], mlo, mhi, K);
                    #pragma unroll
                    for (int i = 0; i < LVG_BAND_SWP_GROUPS; i++)
                    {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, LVG_BAND_SWP_VALU, 0);
                    }
                    #pragma unroll
                    for (int i = 0; i < 8; i++) zq[i] = zp[i];
                    if (bc < 3) accU = accUn;
                }
                __builtin_amdgcn_sched_barrier(0);
                stage_c(3, zq);
#else
                #pragma unroll
                for (int bc = 0; bc < 4; bc++)
                {
                    f32x16 accUn;
                    if (LVG_BAND_SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);      // (keeps the look-ups / mask arithmetic of later blocks from being hoisted into this one's registers)
                    if (LVG_BAND_PIPE && bc < 3) accUn = stage_b(bc + 1);
                    uint32_t zp[8];
                    uint32_t mlo = 0, mhi = 0;
                    if (MODE == LVG_SIGNS_READ) { const uint32_t* r = reinterpret_cast<const uint32_t*>(ML + n * G::SM + 8 * bc); mlo = r[0]; mhi = r[1]; }
                    if (LVG_BABL & 4) { for (int i = 0; i < 8; i++) { half2v t; t[0] = (_Float16)accU[2 * i]; t[1] = (_Float16)accU[2 * i + 1]; zp[i] = h2_bits(t); } }
                    else act_block<MODE, SLOPEMAX, CLAMP, G::LUTN>(accU, zp, mdw[bc], mlo, mhi, K);
                    #pragma unroll
                    for (int h = 0; h < 2; h++)
                    {
                        half8 z;
                        __builtin_memcpy(&z, &zp[4 * h], 16);
                        const int c = 2 * bc + h;
                        #pragma unroll
                        for (int bo = 0; bo < G::OBX; bo++)
                        {
                            const int cls = c - 2 * bo * DOWN;
                            if (cls >= 0 && cls < G::NDC) accW[bo] = mfma(frag_d(cls), z, accW[bo]);
                        }
                    }
                    if (bc < 3) accU = LVG_BAND_PIPE ? accUn : stage_b(bc + 1);
                }
#endif
                // ---- W[ox][v] -> WL[row of v in result order][ox] ------------------------------------------------------------
                #pragma unroll
                for (int bo = 0; bo < G::OBX; bo++)
                    #pragma unroll
                    for (int qd = 0; qd < 4; qd++)
                    {
                        if (32 * bo + 8 * qd >= TW) continue;                     // (columns no output of the strip has)
                        half4 h;
                        #pragma unroll
                        for (int e = 0; e < 4; e++) h[e] = (_Float16)accW[bo][4 * qd + e];
                        if (32 * bo + 8 * qd + 8 <= G::SW || 32 * bo + 8 * qd + 4 * g + 4 <= G::SW)      // (the row ends inside this quad pair: only the pieces that fit)
                            *reinterpret_cast<half4*>(WL + nPhys * G::SW + 32 * bo + 8 * qd + 4 * g) = h;
                    }
            };
            if (MODE == LVG_SIGNS_READ) row_block(BoolC<true>(), BoolC<false>());
            else if (noClamp)           row_block(BoolC<true>(), BoolC<false>());
            else                        row_block(BoolC<true>(), BoolC<true>());

            // ---- READ: the next v-block's mask dwords are requested now (they are consumed at the next iteration start) --------
            if (MODE == LVG_SIGNS_READ)
            {
                if (b + 1 < q.nvb) issue_mask_loads(sPlane, b + 1);
                else if (pl + 1 < nPlanes) issue_mask_loads(sPlane + sPlaneB, 0);
            }
            // ---- WRITE: this v-block's mask rows -> wave-private LDS rows (stored at the end of the iteration) ------------------
            if (MODE == LVG_SIGNS_WRITE)
            {
                const uint32_t selIl = g ? 0x07030602u : 0x05010400u;
                #pragma unroll
                for (int bc = 0; bc < 4; bc++)
                {
                    const uint2v sw = __builtin_amdgcn_permlane32_swap(mdw[bc], mdw[bc], false, false);
                    *reinterpret_cast<uint32_t*>(ML + n * G::SM + 8 * bc + 4 * g) = __builtin_amdgcn_perm(sw[1], sw[0], selIl);
                }
            }

            // ---- stage D, streaming: the two K-chunks of this v-block feed output block (2 b + cc) / CPB with class
            //      (2 b + cc) % CPB; the first SPILL chunks of a block also finish the block above it (classes CPB ..). --------
            int storesNow = 0;
            {
                const int c0 = (2 * b) % G::CPB;                              // class of this v-block's first chunk (even)
                const bool boundary = G::SPILL > 0 && c0 == 0 && b > 0;
                auto d_chunk = [&](int cc, int cls, bool fresh) __attribute__((always_inline))
                {
                    const half8 fdy = frag_d(cls);
                    #pragma unroll
                    for (int bo = 0; bo < G::OBX; bo++)
                        accY[bo] = mfma(lds_tr_operand(WL, G::SW, 16 * cc, 32 * bo, lane), fdy, fresh ? zero16() : accY[bo]);
                };
                uint32_t ypk[G::OBX * 8];
                if (boundary)
                {
                    #pragma unroll
                    for (int cc = 0; cc < 2; cc++) if (cc < G::SPILL) d_chunk(cc, G::CPB + cc, false);
                    // block curBlock is complete: keep it packed in registers until W has been read, start the next one
                    #pragma unroll
                    for (int bo = 0; bo < G::OBX; bo++)
                    {
                        #pragma unroll
                        for (int i = 0; i < 8; i++) ypk[bo * 8 + i] = pack_pair<f16_t>(accY[bo][2 * i], accY[bo][2 * i + 1]);
                    }
                    d_chunk(0, c0, true);
                }
                else if (b == 0) d_chunk(0, c0, true);
                else d_chunk(0, c0, false);
                d_chunk(1, c0 + 1, false);
                if (boundary)
                {
                    stage_rows(ypk);
                    storesNow += store_block(32 * curBlock, min(32, p.yh - 32 * curBlock));
                    ++curBlock;
                }
                // ---- end of the plane: the block in the accumulators (if it has rows) ---------------------------------------------
                if (b == q.nvb - 1)
                {
                    if (32 * curBlock < p.yh)
                    {
                        #pragma unroll
                        for (int bo = 0; bo < G::OBX; bo++)
                            #pragma unroll
                            for (int i = 0; i < 8; i++) ypk[bo * 8 + i] = pack_pair<f16_t>(accY[bo][2 * i], accY[bo][2 * i + 1]);
                        stage_rows(ypk);
                        storesNow += store_block(32 * curBlock, min(32, p.yh - 32 * curBlock));
                    }
                    curBlock = 0;
                }
            }

            // ---- WRITE: the mask rows of this v-block -> global: ONE 16-byte store per lane. A strip owns TW DOWN / 4 bytes of each
            //      of its rows (the last strip: up to the row's end); the lane's piece starts at min(16 half, owned - 16). ------------
            if (MODE == LVG_SIGNS_WRITE)
            {
                int laneV = lane;
                asm volatile("" : "+v"(laneV));
                const int row = laneV >> 1, half = laneV & 1;
                const int signByte0 = (outX0 * DOWN) >> 2;
                const bool lastX = w == q.ns - 1;
                const int nOwn = (lastX ? p.sWBytes - signByte0 : (TW * DOWN) / 4);
                const int sy = 32 * b + row;
                uint8_t* srow = const_cast<uint8_t*>(sPlane) + (uint32_t)(sy * p.sWBytes) + signByte0;
                if (nOwn >= 16)
                {
                    const int o = min(16 * half, nOwn - 16) & ~3;                // (rows of the mask plane and the strips' first bytes are dword aligned)
                    const uint32_t* m = reinterpret_cast<const uint32_t*>(ML + row * G::SM + o);
                    uint32_t wds[4] = {m[0], m[1], m[2], m[3]};
                    // bytes at and beyond swLimit carry no pixels (and the staging row holds 32 bytes): zeros
                    #pragma unroll
                    for (int d = 0; d < 4; d++)
                    {
                        const int nv = min(p.swLimit - (signByte0 + o + 4 * d), 32 - (o + 4 * d));
                        if (nv < 4) wds[d] = nv <= 0 ? 0u : (wds[d] & ((1u << (8 * nv)) - 1u));
                    }
                    if (sy < p.sH && !(LVG_BABL & 16)) *reinterpret_cast<uint4*>(srow + o) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
                    if (32 * b < p.sH && !(LVG_BABL & 16)) storesNow += 1;
                    // the last strip's rows can be longer than 32 bytes + its first byte: the rest of the row is padding, defined as 0
                    if (lastX && nOwn > 32 && sy < p.sH && half == 1)
                    {
                        #pragma unroll 1
                        for (int kb = 32; kb < nOwn; kb++) srow[kb] = 0;
                    }
                }
                else if (sy < p.sH && !(LVG_BABL & 16))
                {
                    // (a last strip narrower than 16 mask bytes: byte stores)
                    #pragma unroll 1
                    for (int kb = half; kb < nOwn; kb += 2)
                        srow[kb] = (signByte0 + kb < p.swLimit && kb < 32) ? ML[row * G::SM + kb] : (uint8_t)0;
                }
            }
            young = storesNow;
        }
        planeChunk0 += q.nch;
        yPlane += yPlaneB;
        sPlane += sPlaneB;
    }
}

template <int UP, int DOWN, int FU, int FD, int TW>
int launch_band(FlreluArgs& a, int mode, hipStream_t stream)
{
    typedef BGeo<UP, DOWN, FU, FD, TW, LVG_SIGNS_READ> GR;
    typedef BGeo<UP, DOWN, FU, FD, TW, LVG_SIGNS_WRITE> GW;
    typedef BGeo<UP, DOWN, FU, FD, TW, LVG_SIGNS_NONE> GN;
    constexpr int KU = FU / UP;
    BandArgs q;
    q.a = a;
    const FlreluArgs& p = q.a;
    if (mode != LVG_SIGNS_READ && !(p.slope <= 1.0f)) return LVG_ERR_UNSUPPORTED;
    // whole contiguous input planes of even width (rows and planes start on dword boundaries); output rows of unit pixel stride
    if (p.xs[3] != 1 || p.xs[2] != p.xw || (p.xw & 1) || (p.xs[0] & 1) || (p.xs[1] & 1) || (((uintptr_t)p.x) & 3)) return LVG_ERR_UNSUPPORTED;
    if (p.xs[1] != (int64_t)p.xh * p.xw || p.xs[0] != (int64_t)p.c * p.xh * p.xw) return LVG_ERR_UNSUPPORTED;
    if (p.ys[3] != 1 || p.ys[2] != p.yw || p.ys[1] != (int64_t)p.yh * p.yw || p.ys[0] != (int64_t)p.c * p.yh * p.yw || (p.yw & 1) || (((uintptr_t)p.y) & 3)) return LVG_ERR_UNSUPPORTED;
    if ((int64_t)p.yh * p.yw * 2 >= 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    if ((int64_t)p.xh * p.xw * 2 >= 0x7fffffffLL || (int64_t)p.sH * p.sWBytes >= 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    if (mode != LVG_SIGNS_NONE && (p.sWBytes & 3)) return LVG_ERR_UNSUPPORTED;
    const int64_t planes = (int64_t)p.n * p.c;
    if (planes > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    q.planes = (int)planes;
    q.ns = (p.yw + TW - 1) / TW;
    if (q.ns > 8 || p.yw - (q.ns - 1) * TW < 8) return LVG_ERR_UNSUPPORTED;     // (the last strip's rows are stored as 16-byte vectors)
    q.phX = ((UP - 1 - p.px0) % UP + UP) % UP;
    q.phY = ((UP - 1 - p.py0) % UP + UP) % UP;
    q.inX0 = lvg_floor_div(UP - 1 - p.px0, UP);
    q.inY0 = lvg_floor_div(UP - 1 - p.py0, UP);
    // A strip's first ring column must be a multiple of 4 (8-byte transpose reads): it starts ef columns early and the A_x
    // fragments are shifted by as many (the shifted band must stay inside the K-chunks it is cut into: ef <= 2 for UP = 4)
    q.ef = ((q.inX0 % 4) + 4) % 4;
    if (UP == 4 && q.ef > 2) return LVG_ERR_UNSUPPORTED;
    // ring row: margin pieces (zeros: the padding left of the image that strip 0 reads) + data pieces, rounded up to 4 mod 8
    // pieces (row pitch = 16 mod 32 dwords: the 16 rows of a transpose read fall on distinct banks)
    q.mp = 1;
    while (q.inX0 - q.ef + 8 * q.mp < 0) q.mp++;
    q.ldsCol0 = q.inX0 - q.ef + 8 * q.mp;
    int LR = q.mp + ((p.xw - 1) >> 3) + 1;
    while (LR % 8 != 4) LR++;
    // zeros right of the image that a valid output can multiply: up to image column xMax. Available: the rest of the row's last
    // piece (zeroed after it lands), the out-of-range pieces behind it, and the next row's margin.
    {
        const int uMax = (p.yw - 1) * DOWN + FD - 1;
        const int xMax = lvg_floor_div(uMax + UP - 1 - p.px0, UP) + KU - 1;
        while (8 * LR - (8 * q.mp + p.xw) + 8 * q.mp < xMax - p.xw + 1) LR += 8;
    }
    q.LR = LR; q.np = LR / 4;
    if (q.np > kMaxPieces * q.ns) return LVG_ERR_UNSUPPORTED;
    const int vNeeded = (p.yh - 1) * DOWN + FD;
    q.nvb = (vNeeded + 31) / 32;
    // v-blocks whose K-chunks reach rows outside the image get their own bias coefficients
    {
        q.ayTop = 0; q.ayBot = q.nvb;
        for (int b = 0; b < q.nvb; b++)
        {
            const int r0 = q.inY0 + (32 * b) / UP, r1 = r0 + (31 + UP - 1) / UP + KU;      // input rows [r0, r1) can meet a tap of this v-block
            if (r0 < 0) q.ayTop = b + 1;
            if (r1 > p.xh && b < q.ayBot) q.ayBot = b;
        }
        if (q.ayBot < q.ayTop) q.ayBot = q.ayTop;
        if (UP == 4) { q.ayTop = q.nvb; q.ayBot = q.nvb; }                  // (odd and even v-blocks use different fragment classes: one row each)
        if (q.ayTop + 1 + (q.nvb - q.ayBot) > kAyRows) return LVG_ERR_UNSUPPORTED;
    }
    {
        const int m = vNeeded - 1 + q.phY, rel = m / UP + KU - 1;
        const int lastB = q.nvb - 1;
        const int byBlocks = UpChunks<UP>::first(lastB) + UpChunks<UP>::count(lastB);
        q.nch = rel / 16 + 1;
        if (q.nch > byBlocks) q.nch = byBlocks;
        if (q.nch < 1) q.nch = 1;
    }

    static int cus[64] = {0};
    int dev = 0; (void)hipGetDevice(&dev);
    int ncu = cus[dev & 63];
    if (ncu == 0)
    {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev & 63] = ncu;
    }
    const int pitchB = LR * 16;
    const int waveBytes = mode == LVG_SIGNS_READ ? GR::WAVE_BYTES : (mode == LVG_SIGNS_WRITE ? GW::WAVE_BYTES : GN::WAVE_BYTES);
    const int offXb = mode == LVG_SIGNS_READ ? GR::OFF_XB : (mode == LVG_SIGNS_WRITE ? GW::OFF_XB : GN::OFF_XB);
    const size_t lds = (size_t)offXb + 2 * pitchB + (size_t)kSlots * 16 * pitchB + kRingPad + (size_t)q.ns * waveBytes;
    if (lds > 160 * 1024) return LVG_ERR_UNSUPPORTED;
    static const int wpsEnv = []() { const char* ev = getenv("LVG_FLRELU_BAND_WPS"); return ev ? atoi(ev) : 0; }();     // (measurements: waves per SIMD the kernel is compiled for)
    static const int gridEnv = []() { const char* ev = getenv("LVG_FLRELU_BAND_MAXGRID"); return ev ? atoi(ev) : 0; }();  // (tests: several planes per workgroup on small tensors)
    const int threads = 64 * q.ns;
    #define LVG_BAND_LAUNCH(M, WPS) do { \
        auto kern = filtered_lrelu_band_kernel<UP, DOWN, FU, FD, TW, M, WPS>; \
        static std::atomic<uint64_t> attr_done{0}; \
        const uint64_t bit_ = 1ull << (dev & 63); \
        if (!(attr_done.load(std::memory_order_acquire) & bit_)) { \
            hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e1 != hipSuccess) { lvg_set_error("filtered_lrelu: cannot reserve LDS for the band kernel: %s", hipGetErrorString(e1)); return LVG_ERR_LAUNCH; } \
            attr_done.fetch_or(bit_, std::memory_order_release); } \
        /* workgroups per CU: the occupancy query, remembered per (device, threads, LDS bytes) */ \
        static std::atomic<uint64_t> occ_key{0}; static std::atomic<int> occ_val{0}; \
        const uint64_t key_ = ((uint64_t)(dev & 63) << 56) | ((uint64_t)threads << 32) | (uint64_t)lds; \
        int perCu = occ_key.load(std::memory_order_acquire) == key_ ? occ_val.load(std::memory_order_relaxed) : 0; \
        if (perCu < 1) { \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, (const void*)kern, threads, lds) != hipSuccess || perCu < 1) perCu = 1; \
            occ_val.store(perCu, std::memory_order_relaxed); occ_key.store(key_, std::memory_order_release); } \
        int64_t maxGrid = (int64_t)ncu * perCu; \
        if (gridEnv > 0 && gridEnv < maxGrid) maxGrid = gridEnv; \
        const unsigned grid = (unsigned)(planes < maxGrid ? planes : maxGrid); \
        if (getenv("LVG_FLRELU_DEBUG")) fprintf(stderr, "filtered_lrelu_band: up %d down %d mode %d: planes %lld, strips %d, v-blocks %d, chunks %d, LR %d, mp %d ef %d col0 %d, lds %zu, %d workgroups/CU, grid %u\n", \
            UP, DOWN, M, (long long)planes, q.ns, q.nvb, q.nch, q.LR, q.mp, q.ef, q.ldsCol0, lds, perCu, grid); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, q); } while (0)
    #define LVG_BAND_MODES(WPS) do { \
        if (mode == LVG_SIGNS_WRITE)     LVG_BAND_LAUNCH(LVG_SIGNS_WRITE, WPS); \
        else if (mode == LVG_SIGNS_READ) LVG_BAND_LAUNCH(LVG_SIGNS_READ, WPS); \
        else                             LVG_BAND_LAUNCH(LVG_SIGNS_NONE, WPS); } while (0)
    if (wpsEnv == 2) LVG_BAND_MODES(2); else LVG_BAND_MODES(3);
    #undef LVG_BAND_MODES
    #undef LVG_BAND_LAUNCH
    return lvg_check_launch("filtered_lrelu_band_kernel");
}

} // namespace

int lvg_flrelu_band_launch(FlreluArgs& p, int cfg, int mode, int dtype, int all, hipStream_t stream)
{
    if (dtype != LVG_F16) return LVG_ERR_UNSUPPORTED;
    if (!all)
    {
        // Measured against the wave kernel at the batch of the training step (16 frames: beyond the Infinity Cache;
        // profiles/r05_band_cold.log, r05_band_small_layers_time.log, r05_sres_ab_perlayer.log): faster where the strips of a plane give
        // workgroups of three or four waves -- 12 waves per CU, three per SIMD -- in the forward modes (148-wide outputs: -16 .. -18 %) and in
        // the up 2 / down 4 backward (86-wide: 436 -> 356 us); on a par or behind at one strip (7 waves per CU), at five (10 waves, uneven over
        // the SIMDs), at six, and in the up 2 / down 2 backward (its mask reads miss the caches one v-block at a time).
        const int tw = cfg == LVG_FLRELU_CFG_U2D4 ? 26 : 56, ns = (p.yw + tw - 1) / tw;
        bool take = false;
        if (cfg == LVG_FLRELU_CFG_U2D2) take = mode != LVG_SIGNS_READ && (ns == 2 || ns == 3);
        if (cfg == LVG_FLRELU_CFG_U4D2) take = mode != LVG_SIGNS_READ && ns == 3;
        if (cfg == LVG_FLRELU_CFG_U2D4) take = ns == 4;
        if (!take) return LVG_ERR_UNSUPPORTED;
    }
    switch (cfg)
    {
        case LVG_FLRELU_CFG_U2D2: return launch_band<2, 2, 12, 12, 56>(p, mode, stream);
        case LVG_FLRELU_CFG_U4D2: return launch_band<4, 2, 24, 12, 56>(p, mode, stream);
        case LVG_FLRELU_CFG_U2D4: return launch_band<2, 4, 12, 24, 26>(p, mode, stream);
    }
    return LVG_ERR_UNSUPPORTED;
}
