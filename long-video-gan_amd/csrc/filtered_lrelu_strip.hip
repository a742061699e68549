// filtered_lrelu_strip.hip -- fused bias -> up-FIR -> gain -> leaky ReLU -> clamp -> down-FIR for float16 tensors on gfx950
// (round 6): the banded-matrix arithmetic of filtered_lrelu_band.hip with NOTHING shared between waves.
// What the round-5 kernels measured as (profiles/r06_abl_a.log, r06_probe_issue.log): with the matrix products compiled out the
// row-band kernel runs at the issue rate of its vector instructions (46 us for L8), with them at 76 us, with memory and the
// workgroup barrier at 103 us -- every wave is a serial chain (LDS round trips, product -> conversion dependencies, the barrier
// of its workgroup) and three waves per SIMD do not cover it; going from two to three waves gave 1.33x. So here
//   * a WAVE owns a column strip of 24 outputs of one plane (64 up-sampled columns for up 2 / down 2 and up 4 / down 2, 128 for
//     up 2 / down 4: one block of output columns, half the accumulators of the row-band kernel) -- <= 128 registers, 12 to 16 waves
//     per CU -- and walks down it in steps of 32 up-sampled rows; work items (plane, strip) are dealt out round robin over the waves;
//   * its input rows enter a wave-PRIVATE LDS ring (three K-chunks of 16 rows) through LDS-DMA; no barrier after the set-up, the
//     chunk stream runs across item boundaries (the next strip's first rows are requested while the current strip finishes);
//   * DMA origins are even columns (dword-aligned addresses); the odd residue, and the padding left of the image in a plane's first
//     strip, are absorbed by shifting the horizontal up fragments (two fragment images in LDS, fetched per item);
//   * output rows and mask bytes leave straight from the accumulators (8-byte / 4-byte stores, no staging round trip); in READ mode
//     the mask dwords of a row come through a buffer descriptor (rows outside the mask plane read as zeros) one block ahead;
//   * the bias rides in the spare K slot of the vertical up stage, the vertical down stage streams over blocks, the mask offsets are
//     a bit shift of the mask stream, the no-clamp proof comes from max |T'| -- all as in filtered_lrelu_band.hip.
// Semantics: reference torch_utils/ops/filtered_lrelu.cu:139-1099, filtered_lrelu.cpp:16-210 (2-bit sign / clamp mask in write and
// read mode with offsets). Arithmetic: f16 operands, f32 accumulation; T', Z and W rounded to f16 between stages.
// Not taken (LVG_ERR_UNSUPPORTED, the caller falls back): bfloat16, non-contiguous planes, odd widths, slope > 1 forward.
// Algorithmic HBM bytes: (N_in + N_out) * 2 + mask bytes; see DESIGN.md.

#include "lvg_common.h"
#include "filtered_lrelu_args.h"
#include <atomic>
#include <stdlib.h>

#ifndef LVG_SABL
#define LVG_SABL 0           // ablation builds only (results are WRONG): 2 no y stores, 4 no activation math, 16 no mask stores / loads, 32 no matrix products, 64 no input DMA, 256 no wait for the DMA
#endif
#define LVG_WABL (LVG_SABL & 32)
#ifndef LVG_SDUMMY_V
#define LVG_SDUMMY_V 0       // measurement builds: extra independent packed multiplies per column block
#endif
#ifndef LVG_SDUMMY_M
#define LVG_SDUMMY_M 0       // measurement builds: extra independent matrix products per column block
#endif

namespace {

#include "flrelu_mfma_common.h"

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <bool B> struct BoolC { static constexpr bool value = B; };
template <int V> struct IntC { static constexpr int value = V; };

constexpr int kSlots = 3;          // ring slots (K-chunks of 16 input rows): two in use, one in flight
constexpr int kAyRows = 12;        // rows of the bias-coefficient table
constexpr int kTW = 24;            // output columns of a strip

struct StripArgs
{
    FlreluArgs a;
    int planes, ns;        // planes, strips per plane
    int items;             // planes * ns
    int nvb, nch;          // v-blocks and K-chunks (16 input rows) per item
    int ayTop, ayBot;
    int inX0, inY0;
    int phX, phY;
    int org0, base0, ef0;  // first strip of a plane: DMA origin column (multiple of 8, <= 0 when the padding reaches left of the image), ring column of its first column (0 or 4), fragment shift
    int ef1;               // other strips: DMA origin = first column rounded down to even, fragment shift = its parity
    int waves;             // waves of the launch: wave g takes items g, g + waves, g + 2 waves, ... (neighbouring strips of a plane run side by side on
                           // neighbouring waves: their overlapping input columns and adjacent output pieces meet in the L2 -- dealt out in contiguous runs the
                           // strips of a plane followed each other on ONE wave, 8 us apart, and the counters showed twice the input bytes fetched from HBM)
    int stepP, stepS;      // waves / ns, waves % ns: what one step of `waves` items adds to (plane, strip); (0, 1) when items are dealt out in contiguous runs
    int rr;                // 1 = round robin
};

template <int UP, int DOWN, int FU, int FD, int NBC, int MODE>
struct SGeo
{
    static constexpr int KU     = FU / UP;
    static constexpr int UW     = 32 * NBC;                                 // up-sampled columns of a strip
    static constexpr int IN_NX  = (UP - 1 + UW - 1) / UP + KU + 3;          // input columns a strip touches (+3: fragment shift)
    static constexpr int CH_X   = ((IN_NX - 1) >> 4) + 1;                   // 16-chunks of input columns stage B can read
    static constexpr int IN_BLK = wdiv_up(CH_X, 2);                         // 32-blocks of input columns (stage A's M)
    static constexpr int PPR    = 4 * IN_BLK;                               // 16-byte pieces per ring row
    static constexpr int PITCH  = 16 * PPR;                                 // ring row pitch (bytes): 16 rows = IN_BLK KiB = IN_BLK DMA instructions
    static constexpr int SLOT   = 16 * PITCH;
    static constexpr int SPITCH = kTW * DOWN / UP;                          // input columns between neighbouring strips
    static constexpr int SW     = 28;                                       // W row stride (halves): 4 mod 8
    static constexpr int NUC    = (UP == 2) ? 2 : (UP == 4 ? 3 : 1);
    static constexpr int CPB    = 2 * DOWN;
    static constexpr int NDC    = ((31 * DOWN + FD - 1) >> 4) + 1;
    static constexpr int SPILL  = NDC - CPB;
    static constexpr bool HAS_M = MODE != LVG_SIGNS_NONE;
    static constexpr int TAPS   = (FU + FD + 3) / 4 * 4;
    static constexpr int LUTN   = MODE != LVG_SIGNS_READ ? 0 : 171;
    static constexpr int MDW    = 2 * NBC + 1;                              // READ: aligned mask dwords of a row that cover the strip's UW pixels at any bit offset
    // LDS map (bytes), shared part: taps | READ look-up table | down-stage fragment images | bias-coefficient table | horizontal up fragment images (two shifts)
    static constexpr int OFF_TAPS = 0;
    static constexpr int OFF_LUT  = TAPS * 4;
    static constexpr int OFF_TAB  = OFF_LUT + (LUTN * 8 + 15) / 16 * 16;
    static constexpr int OFF_AY   = OFF_TAB + NDC * 1024;
    static constexpr int OFF_AX   = OFF_AY + kAyRows * 128;                 // [2][NUC] fragment images of 1 KiB
    static constexpr int OFF_WAVE = OFF_AX + 2 * NUC * 1024;
    // per wave: ring | bias row | W rows of one v-block | pad (transpose reads of the last rows run past their end)
    static constexpr int W_OFF    = kSlots * SLOT + PITCH;
    static constexpr int WAVE_BYTES = W_OFF + 32 * SW * 2 + 64;
    static_assert(FU % UP == 0 && FD % DOWN == 0, "filter sizes must be multiples of the rates");
    static_assert((kTW * DOWN) % 16 == 0 && (kTW * DOWN) % UP == 0 && SPITCH % 2 == 0, "strip origins keep the mask dword, the up-sampling phase and the parity of the first column");
    static_assert((kTW - 1) * DOWN + FD <= UW, "strip does not fit its up-sampled block");
    static_assert(SPILL >= 0 && SPILL <= 2 && CPB % 2 == 0, "streaming stage D: the spill chunks of a block lie in one v-block");
    static_assert(2 * NBC <= NDC, "every 16-chunk of u meets the one block of output columns");
    static_assert(OFF_TAB % 16 == 0 && OFF_AY % 16 == 0 && OFF_AX % 16 == 0 && OFF_WAVE % 16 == 0 && WAVE_BYTES % 16 == 0, "alignment");
    static_assert(4 + 16 * CH_X <= 8 * PPR + 8 * PPR, "stage A reads stay within two ring rows");
};

// LDS-DMA of 16 bytes per lane through a raw buffer (offsets outside [0, num_records) deliver zeros): LDS address = ldsPiece
// (wave-uniform, via M0) + lane * 16. Inline assembly: see filtered_lrelu_band.hip.
__device__ __forceinline__ void dma16_buf(v4i rsrc, uint32_t laneOff, uint32_t ldsPiece)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(laneOff), "s"(rsrc), "s"(ldsPiece) : "memory");
}

// s_waitcnt vmcnt(k): everything this wave issued except its k youngest vector-memory operations is complete (k a lower bound of
// what the caller knows to be younger: a smaller immediate waits for more, never for less).
__device__ __forceinline__ void wait_vm_all_but(int k)
{
    if (LVG_SABL & 256) return;
    if (k <= 0)      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (k == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (k == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (k == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (k == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (k == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (k < 8)  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else             asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <int UP, int DOWN, int FU, int FD, int NBC, int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB) void filtered_lrelu_strip_kernel(StripArgs q)
{
    typedef SGeo<UP, DOWN, FU, FD, NBC, MODE> G;
    const FlreluArgs& p = q.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float*    taps  = reinterpret_cast<float*>(smem + G::OFF_TAPS);
    _Float16* tabD  = reinterpret_cast<_Float16*>(smem + G::OFF_TAB);
    _Float16* tabAy = reinterpret_cast<_Float16*>(smem + G::OFF_AY);
    _Float16* tabAx = reinterpret_cast<_Float16*>(smem + G::OFF_AX);
    const int tid = threadIdx.x, lane = tid & 63, w = sgpr(tid >> 6);
    const int n = lane & 31, g = lane >> 5;
    unsigned char* wv = smem + G::OFF_WAVE + w * G::WAVE_BYTES;
    unsigned char* ring = wv;
    unsigned char* xbRow = wv + kSlots * G::SLOT;                           // bias row of the current item
    _Float16* WL = reinterpret_cast<_Float16*>(wv + G::W_OFF);              // W [32][SW], rows in the k order of an MFMA result
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t ldsWave = ldsBase + (uint32_t)(G::OFF_WAVE + w * G::WAVE_BYTES);

    // ---- once per workgroup: taps, zeroed wave regions, tables ---------------------------------------------------------------
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
    for (int i = tid; i < (WPB * G::WAVE_BYTES) / 4; i += (int)blockDim.x) reinterpret_cast<uint32_t*>(smem + G::OFF_WAVE)[i] = 0u;
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
    for (int e = tid; e < 2 * G::NUC * 512; e += (int)blockDim.x)
    {
        const int var = e / (G::NUC * 512), r = e - var * (G::NUC * 512);
        tabAx[e] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 1, r >> 9, (r & 511) >> 3, r & 7, phX, phY, 0, scale, var == 0 ? q.ef0 : q.ef1);
    }
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
    // ---- from here on the waves of the workgroup share nothing but the read-only tables ------------------------------------------

    half8 fAy[G::NUC];
    #pragma unroll
    for (int c = 0; c < G::NUC; c++)
        #pragma unroll
        for (int j = 0; j < 8; j++)
            fAy[c][j] = (_Float16)frag_elem<UP, DOWN, FU, FD>(taps, 0, c, lane, j, phX, phY, 0, scale);
    typedef __attribute__((address_space(3))) const half8* lds_h8;
    const uint32_t tabLane = ldsBase + (uint32_t)G::OFF_TAB + (uint32_t)lane * 16u;
    auto frag_d = [&](int cls) __attribute__((always_inline)) -> half8 { return *(lds_h8)(uintptr_t)(tabLane + (uint32_t)cls * 1024u); };
    const uint32_t axLane = ldsBase + (uint32_t)G::OFF_AX + (uint32_t)lane * 16u;

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
        if (!(p.slope <= 1.0f) || !(tLimit > 0.0f)) tLimit = 0.0f;
    }

    // ---- this wave's items: (plane, strip) pairs gw, gw + waves, gw + 2 waves, ... ---------------------------------------------------
    const int gw = (int)blockIdx.x * WPB + w;
    // round robin (stepP / stepS = one step of `waves` items), or -- up 2 / down 4, measured faster that way -- contiguous runs (step = one strip)
    const int itemBeg = q.rr ? gw : (int)((int64_t)q.items * gw / q.waves);
    const int nItems = q.rr ? (gw < q.items ? (q.items - gw + q.waves - 1) / q.waves : 0) : (int)((int64_t)q.items * (gw + 1) / q.waves) - itemBeg;
    if (nItems <= 0) return;
    const int totalChunks = nItems * q.nch;
    const uint32_t rowBytes = (uint32_t)p.xw * 2u, planeBytes = (uint32_t)p.xh * rowBytes;
    const uint32_t yRowB = (uint32_t)p.yw * 2u;
    const uint64_t yPlaneB = (uint64_t)p.yh * yRowB, sPlaneB = (uint64_t)p.sH * (uint64_t)p.sWBytes;

    // ---- input DMA: the chunk stream of this wave's items; lane -> (row of the chunk, piece of the row) of DMA instruction i ------
    int issPlane = itemBeg / q.ns, issStrip = itemBeg - issPlane * q.ns;
    int gIssue = 0, issChunk = 0, issSlot = 0;
    uint32_t issRowBase = (uint32_t)(q.inY0 * (int)rowBytes);               // (negative rows wrap: out of range)
    uint64_t issPlanePtr = (uint64_t)(uintptr_t)p.x + (uint64_t)issPlane * planeBytes;
    uint32_t dOff[G::IN_BLK];
    auto strip_origin = [&](int strip) __attribute__((always_inline)) -> int { return strip == 0 ? q.org0 : ((q.inX0 + strip * G::SPITCH) & ~1); };
    auto set_issue_lanes = [&]() __attribute__((always_inline))
    {
        const int org = strip_origin(issStrip);
        #pragma unroll
        for (int i = 0; i < G::IN_BLK; i++)
        {
            const int idx = 64 * i + lane, row = idx / G::PPR, pc = idx - row * G::PPR;
            const int col0 = org + 8 * pc;
            const bool any = col0 >= 0 && col0 < p.xw;                      // (col0 < 0: whole pieces left of the image -- the first strip's origin is a multiple of 8, the others' are >= 0)
            dOff[i] = any ? (uint32_t)(row * (int)rowBytes + 2 * col0) : 0xfffffff0u;
        }
    };
    set_issue_lanes();
    auto issue_chunk = [&]() __attribute__((always_inline))
    {
        v4i rsrc;
        rsrc[0] = sgpr((int)(uint32_t)issPlanePtr); rsrc[1] = sgpr((int)(uint32_t)((issPlanePtr >> 32) & 0xffffu)); rsrc[2] = sgpr((int)planeBytes); rsrc[3] = 0x00020000;
        const uint32_t slotLds = ldsWave + (uint32_t)issSlot * (uint32_t)G::SLOT;
        if (!(LVG_SABL & 64))
        {
            #pragma unroll
            for (int i = 0; i < G::IN_BLK; i++)
            {
                const uint32_t vo = dOff[i] == 0xfffffff0u ? 0xfffffff0u : issRowBase + dOff[i];
                dma16_buf(rsrc, vo, (uint32_t)sgpr((int)(slotLds + (uint32_t)i * 1024u)));
            }
        }
        ++gIssue;
        issSlot = issSlot == kSlots - 1 ? 0 : issSlot + 1;
        issRowBase += 16u * rowBytes;
        if (++issChunk == q.nch)
        {
            issChunk = 0; issRowBase = (uint32_t)(q.inY0 * (int)rowBytes);
            issStrip += q.stepS;
            issPlanePtr += (uint64_t)q.stepP * planeBytes;
            if (issStrip >= q.ns) { issStrip -= q.ns; issPlanePtr += planeBytes; }
            set_issue_lanes();                                               // (the origin moves with the strip)
        }
    };

    const int hgrp = (lane >> 4) & 1, s16 = lane & 15;
    const bool row15 = g == 1 && (s16 >> 2) == 3;
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    auto tr_read = [&](uint32_t ldsAddr) __attribute__((always_inline)) -> short4v { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)ldsAddr); };
    const int nPhys = (n & 16) + 8 * ((n >> 2) & 1) + (n & 3) + 4 * ((n >> 3) & 1);

    // ---- the walk ---------------------------------------------------------------------------------------------------------------
    f32x16 accY = zero16();
    half2v dumV[8]; f32x16 dumAcc = zero16();
    for (int i = 0; i < 8; i++) { dumV[i][0] = (_Float16)(1.0f + 0.001f * lane); dumV[i][1] = (_Float16)1.0f; }
    int curBlock = 0;
    int young = 0;                                                          // vector-memory operations issued after the last DMA piece / mask load
    int gFirst = 0, slotFirst = 0;
    int itemChunk0 = 0;
    int plane = itemBeg / q.ns, strip = itemBeg - plane * q.ns;
    uint32_t mraw[G::MDW];
    #pragma unroll
    for (int j = 0; j < G::MDW; j++) mraw[j] = 0u;

    for (int it = 0; it < nItems; it++)
    {
        // ---- per item: geometry of the strip ---------------------------------------------------------------------------------
        const int org = strip_origin(strip);
        const int trBase = strip == 0 ? q.base0 : 0;                         // ring column of the strip's first column (minus the fragment shift)
        const uint32_t laneA = ldsWave + (uint32_t)((8 * g + (s16 >> 2)) * G::PITCH + (trBase + 16 * hgrp + 4 * (s16 & 3)) * 2);
        const uint32_t laneXb = ldsWave + (uint32_t)(kSlots * G::SLOT) + (uint32_t)((trBase + 16 * hgrp + 4 * (s16 & 3)) * 2);
        half8 fAx[G::NUC];
        #pragma unroll
        for (int c = 0; c < G::NUC; c++) fAx[c] = *(lds_h8)(uintptr_t)(axLane + (uint32_t)((strip == 0 ? 0 : G::NUC) + c) * 1024u);
        const bool lastStrip = strip == q.ns - 1;
        const int outX0 = strip * kTW;
        const int colsHere = min(kTW, p.yw - outX0);
        char* yPlane = (char*)p.y + (uint64_t)plane * yPlaneB + (uint32_t)outX0 * 2u;
        uint8_t* sPlane = const_cast<uint8_t*>(p.s) + (uint64_t)plane * sPlaneB;
        // bias row: b in the columns of the image, 0 elsewhere (ring column j = image column org + j)
        {
            const uint32_t bb = scalar_load_u16((const uint16_t*)p.b + sgpr(plane % p.c));
            for (int j = lane; j < G::PITCH / 2; j += 64)
            {
                const int x = org + j;
                reinterpret_cast<uint16_t*>(xbRow)[j] = (x >= 0 && x < p.xw) ? (uint16_t)bb : (uint16_t)0;
            }
        }
        // right edge: the piece that straddles the end of a row carries the next row's first pixels -> zeros after it lands
        const int straddlePc = (p.xw - org) >> 3;                            // piece that holds column xw (any strip whose ring row reaches the row's end)
        const int nGarb = (org + 8 * straddlePc + 8 - p.xw) & 7;             // halves to clear at the end of that piece (0: xw falls on a piece boundary)
        const bool doPatch = nGarb > 0 && straddlePc < G::PPR;
        auto patch_slot = [&](int slot) __attribute__((always_inline))
        {
            // lane -> (row = lane >> 2, two of the piece's halves)
            uint16_t* pc = reinterpret_cast<uint16_t*>(ring + slot * G::SLOT + (lane >> 2) * G::PITCH + straddlePc * 16);
            const int h = 8 - nGarb + (lane & 3);
            if (h < 8) pc[h] = 0;
            if (h + 4 < 8) pc[h + 4] = 0;
        };
        // READ mode: aligned dwords of this lane's mask row; the sign offsets enter as a BIT shift of the stream
        const int maskX0 = strip * (kTW * DOWN) + p.sOfsX;
        const int mshift = 8 * ((maskX0 >> 2) & 3) + 2 * (maskX0 & 3);       // <= 30
        const int a0 = (maskX0 >> 2) & ~3;                                   // first aligned byte of the row's stream (floor: maskX0 may be negative)
        const bool maskFast = a0 >= 0 && a0 + 4 * G::MDW <= p.swLimit;       // every dword inside the row's pixel bytes
        auto issue_mask_loads = [&](int b) __attribute__((always_inline))
        {
            if (LVG_SABL & 16) return;
            const int sy = 32 * b + p.sOfsY + n;
            const bool rowOk = (uint32_t)sy < (uint32_t)p.sH;
            const uint8_t* rowp = sPlane + (rowOk ? (uint32_t)(sy * p.sWBytes) : 0u);
            if (maskFast)
            {
                #pragma unroll
                for (int j = 0; j < G::MDW; j++)
                {
                    const uint32_t v = *reinterpret_cast<const uint32_t*>(rowp + a0 + 4 * j);
                    mraw[j] = rowOk ? v : 0u;
                }
            }
            else
            {
                #pragma unroll
                for (int j = 0; j < G::MDW; j++)
                {
                    const int bx = a0 + 4 * j;
                    const bool ok = rowOk && bx >= 0 && bx + 4 <= p.sWBytes;
                    uint32_t v = *reinterpret_cast<const uint32_t*>(rowp + (ok ? bx : 0));
                    const int nv = p.swLimit - bx;                           // bytes at and beyond swLimit carry no pixels
                    v = !ok || nv <= 0 ? 0u : (nv < 4 ? (v & ((1u << (8 * nv)) - 1u)) : v);
                    mraw[j] = v;
                }
            }
        };
        if (MODE == LVG_SIGNS_READ) { issue_mask_loads(0); young = 0; }     // (younger than the previous item's stores: the next wait is for everything)

        #pragma unroll 1
        for (int b = 0; b < q.nvb; b++)
        {
            const int cFirst = (UP == 4) ? (b >> 1) : b, cCount = (UP == 4) ? 1 + (b & 1) : 2;
            while (gFirst < itemChunk0 + cFirst) { ++gFirst; slotFirst = slotFirst == kSlots - 1 ? 0 : slotFirst + 1; }
            // The chunks this v-block reads were requested during earlier iterations: they have landed once all but the `young`
            // youngest operations (the stores of the previous iteration's end) are complete. Then request what the ring has room for.
            const int lastNeeded = min(gFirst + cCount - 1, itemChunk0 + q.nch - 1);
            if (lastNeeded < gIssue) wait_vm_all_but(young);
            const int issuedBefore = gIssue;
            while (gIssue <= gFirst + kSlots - 1 && gIssue < totalChunks) issue_chunk();
            if (lastNeeded >= issuedBefore) wait_vm_all_but(0);              // (first iteration of the wave, or the ring was behind: the data was requested just now)
            young = 0;
            const int slotSecond = slotFirst == kSlots - 1 ? 0 : slotFirst + 1;
            if (doPatch)
            {
                patch_slot(slotFirst);
                if (gFirst + 1 <= itemChunk0 + q.nch - 1) patch_slot(slotSecond);
            }
            // READ: this v-block's mask dwords (requested one block ago) -> the two dwords per column block this lane's row needs
            uint32_t mrow[2 * NBC];
            if (MODE == LVG_SIGNS_READ)
            {
                #pragma unroll
                for (int d = 0; d < 2 * NBC; d++) mrow[d] = __builtin_amdgcn_alignbit(mraw[d + 1], mraw[d], (uint32_t)mshift);
            }

            // ---- stage A: T'[ic][v] for the 32 rows v of this v-block; K-chunks = ring slots; bias through row 15 of the last ----
            half8 tpk[G::CH_X];
            bool noClamp = false;
            {
                f32x16 accA[G::IN_BLK];
                #pragma unroll
                for (int m = 0; m < G::IN_BLK; m++) accA[m] = zero16();
                const int ayRow = b < q.ayTop ? b : (b < q.ayBot ? q.ayTop : q.ayTop + 1 + (b - q.ayBot));
                const uint32_t ayDw = (uint32_t)reinterpret_cast<const uint16_t*>(tabAy)[ayRow * 64 + lane];
                auto chunk_a = [&](int slot, auto clsC, bool last) __attribute__((always_inline))
                {
                    constexpr int cls = decltype(clsC)::value;
                    const uint32_t slotA = laneA + (uint32_t)slot * (uint32_t)G::SLOT;
                    const uint32_t hiA = (last && row15) ? laneXb : slotA + 4u * (uint32_t)G::PITCH;
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
                    noClamp = __builtin_amdgcn_ballot_w64(!(mx < tLimit)) == 0;
                }
            }

            // ---- stages B, activation, C over the NBC 32-column blocks of u -----------------------------------------------------
            uint32_t mdw[NBC];
            #pragma unroll
            for (int i = 0; i < NBC; i++) mdw[i] = 0u;
            f32x16 accW = zero16();
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
                f32x16 accU = stage_b(0);
                #pragma unroll
                for (int bc = 0; bc < NBC; bc++)
                {
                    f32x16 accUn;
                    __builtin_amdgcn_sched_barrier(0);
                    if (bc < NBC - 1) accUn = stage_b(bc + 1);
                    uint32_t zp[8];
                    uint32_t mlo = 0, mhi = 0;
                    if (MODE == LVG_SIGNS_READ) { mlo = mrow[2 * bc]; mhi = mrow[2 * bc + 1]; }
                    if (LVG_SABL & 4) { for (int i = 0; i < 8; i++) { half2v t; t[0] = (_Float16)accU[2 * i]; t[1] = (_Float16)accU[2 * i + 1]; zp[i] = h2_bits(t); } }
                    else act_block<MODE, SLOPEMAX, CLAMP, G::LUTN>(accU, zp, mdw[bc], mlo, mhi, K);
                    #pragma unroll
                    for (int h = 0; h < 2; h++)
                    {
                        half8 z;
                        __builtin_memcpy(&z, &zp[4 * h], 16);
                        accW = mfma(frag_d(2 * bc + h), z, accW);
                    }
                    if (LVG_SDUMMY_V)
                    {
                        #pragma unroll
                        for (int i = 0; i < LVG_SDUMMY_V; i++) { dumV[i & 7] = dumV[i & 7] * K.slope2; asm volatile("" : "+v"(dumV[i & 7])); }
                    }
                    if (LVG_SDUMMY_M)
                    {
                        #pragma unroll
                        for (int i = 0; i < LVG_SDUMMY_M; i++) dumAcc = mfma(fAx[0], tpk[0], dumAcc);
                    }
                    if (bc < NBC - 1) accU = accUn;
                }
                // ---- W[ox][v] -> WL[row of v in result order][ox] ------------------------------------------------------------
                #pragma unroll
                for (int qd = 0; qd < kTW / 8; qd++)
                {
                    half4 h;
                    #pragma unroll
                    for (int e = 0; e < 4; e++) h[e] = (_Float16)accW[4 * qd + e];
                    *reinterpret_cast<half4*>(WL + nPhys * G::SW + 8 * qd + 4 * g) = h;
                }
            };
            if (MODE == LVG_SIGNS_READ) row_block(BoolC<true>(), BoolC<false>());
            else if (noClamp)           row_block(BoolC<true>(), BoolC<false>());
            else                        row_block(BoolC<true>(), BoolC<true>());

            // ---- READ: the next v-block's mask dwords are requested now ---------------------------------------------------------
            if (MODE == LVG_SIGNS_READ && b + 1 < q.nvb) issue_mask_loads(b + 1);

            // ---- WRITE: this v-block's mask: after the half-wave exchange lane (n, g) holds bytes 8 bc + 4 g .. + 3 of its row ----
            int storesNow = 0;
            if (MODE == LVG_SIGNS_WRITE)
            {
                const uint32_t selIl = g ? 0x07030602u : 0x05010400u;
                const int signByte0 = (outX0 * DOWN) >> 2;
                const int nOwn = lastStrip ? p.sWBytes - signByte0 : (kTW * DOWN) / 4;
                const int sy = 32 * b + n;
                uint8_t* srow = sPlane + (uint32_t)(sy * p.sWBytes) + signByte0;
                uint32_t mv[NBC], pv[NBC];                                  // this lane's dword of every column block, its partner lane's (other half wave, same row)
                #pragma unroll
                for (int bc = 0; bc < NBC; bc++)
                {
                    const uint2v sw = __builtin_amdgcn_permlane32_swap(mdw[bc], mdw[bc], false, false);
                    uint32_t v = __builtin_amdgcn_perm(sw[1], sw[0], selIl);
                    const int o = 8 * bc + 4 * g;
                    const int nv = p.swLimit - (signByte0 + o);               // bytes at and beyond swLimit carry no pixels: zeros
                    if (nv < 4) v = nv <= 0 ? 0u : (v & ((1u << (8 * nv)) - 1u));
                    mv[bc] = v;
                    const uint2v ex = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                    pv[bc] = g ? ex[0] : ex[1];
                }
                // One store per lane: the lanes of half wave g own bytes 16 g .. 16 g + 15 of their row (NBC = 2: the lower half wave stores the whole
                // 12- / 16-byte piece of the row, the upper one nothing) -- 12 or 16 contiguous bytes per row instead of three 4-byte pieces.
                if (sy < p.sH && !(LVG_SABL & 16))
                {
                    if (NBC == 2)
                    {
                        const int k = nOwn >> 2;                              // dwords of the row this strip owns (3; the last strip: up to 4)
                        if (g == 0)
                        {
                            if (k >= 4)      *reinterpret_cast<uint4*>(srow) = make_uint4(mv[0], pv[0], mv[1], pv[1]);
                            else if (k == 3) { *reinterpret_cast<uint2*>(srow) = make_uint2(mv[0], pv[0]); *reinterpret_cast<uint32_t*>(srow + 8) = mv[1]; }
                            else if (k == 2) *reinterpret_cast<uint2*>(srow) = make_uint2(mv[0], pv[0]);
                            else if (k == 1) *reinterpret_cast<uint32_t*>(srow) = mv[0];
                        }
                    }
                    else
                    {
                        const int k = min(4, max(0, (nOwn >> 2) - 4 * g));
                        const uint32_t d0 = g ? pv[2] : mv[0], d1 = g ? mv[2] : pv[0], d2 = g ? pv[NBC - 1] : mv[1], d3 = g ? mv[NBC - 1] : pv[1];
                        uint8_t* dst = srow + 16 * g;
                        if (k >= 4)      *reinterpret_cast<uint4*>(dst) = make_uint4(d0, d1, d2, d3);
                        else if (k == 3) { *reinterpret_cast<uint2*>(dst) = make_uint2(d0, d1); *reinterpret_cast<uint32_t*>(dst + 8) = d2; }
                        else if (k == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(d0, d1);
                        else if (k == 1) *reinterpret_cast<uint32_t*>(dst) = d0;
                    }
                }
                if (32 * b < p.sH && nOwn >= 4 && !(LVG_SABL & 16)) storesNow += 1;      // (lower bound of the store instructions issued)
            }

            // ---- stage D, streaming (see filtered_lrelu_band.hip): the two K-chunks of this v-block feed output block
            //      (2 b + cc) / CPB with class (2 b + cc) % CPB; the first SPILL chunks of a block also finish the block above. ----
            {
                const int c0 = (2 * b) % G::CPB;
                const bool boundary = G::SPILL > 0 && c0 == 0 && b > 0;
                auto d_chunk = [&](int cc, int cls, bool fresh) __attribute__((always_inline))
                {
                    const half8 fdy = frag_d(cls);
                    accY = mfma(lds_tr_operand(WL, G::SW, 16 * cc, 0, lane), fdy, fresh ? zero16() : accY);
                };
                auto store_block = [&](const uint32_t (&ypk)[8], int oy0) __attribute__((always_inline)) -> int
                {
                    // lane (n, g): output row oy0 + n. Full strips: the two half waves exchange their first / second quad so that lane (n, g) holds
                    // columns 8 g .. 8 g + 7 (one 16-byte store) and its own four of columns 16 .. 23 (8 bytes); ragged strips: 8- / 4-byte pieces.
                    const int oy = oy0 + n;
                    if (LVG_SABL & 2) return 0;
                    char* rowp = yPlane + (uint32_t)oy * yRowB;
                    if (colsHere == kTW)
                    {
                        const uint2v e0 = __builtin_amdgcn_permlane32_swap(ypk[0], ypk[2], false, false);
                        const uint2v e1 = __builtin_amdgcn_permlane32_swap(ypk[1], ypk[3], false, false);
                        if (oy < p.yh)
                        {
                            *reinterpret_cast<uint4*>(rowp + 16 * g) = make_uint4(e0[0], e1[0], e0[1], e1[1]);
                            *reinterpret_cast<uint2*>(rowp + 32 + 8 * g) = make_uint2(ypk[4], ypk[5]);
                        }
                        return 2;
                    }
                    int issued = 0;
                    #pragma unroll
                    for (int qd = 0; qd < kTW / 8; qd++)
                    {
                        const int c = 8 * qd + 4 * g;
                        char* dst = rowp + (uint32_t)c * 2u;
                        if (8 * qd < colsHere)
                        {
                            if (oy < p.yh)
                            {
                                if (c + 4 <= colsHere) *reinterpret_cast<uint2*>(dst) = make_uint2(ypk[2 * qd], ypk[2 * qd + 1]);
                                else if (c + 2 <= colsHere) *reinterpret_cast<uint32_t*>(dst) = ypk[2 * qd];
                            }
                            issued += 1;                                     // (lower bound: a ragged quad pair may take both store forms)
                        }
                    }
                    return issued;
                };
                uint32_t ypk[8];
                if (boundary)
                {
                    #pragma unroll
                    for (int cc = 0; cc < 2; cc++) if (cc < G::SPILL) d_chunk(cc, G::CPB + cc, false);
                    #pragma unroll
                    for (int i = 0; i < 8; i++) ypk[i] = pack_pair<f16_t>(accY[2 * i], accY[2 * i + 1]);
                    d_chunk(0, c0, true);
                }
                else if (b == 0) d_chunk(0, c0, true);
                else d_chunk(0, c0, false);
                d_chunk(1, c0 + 1, false);
                if (boundary)
                {
                    storesNow += store_block(ypk, 32 * curBlock);
                    ++curBlock;
                }
                if (b == q.nvb - 1)
                {
                    if (32 * curBlock < p.yh)
                    {
                        #pragma unroll
                        for (int i = 0; i < 8; i++) ypk[i] = pack_pair<f16_t>(accY[2 * i], accY[2 * i + 1]);
                        storesNow += store_block(ypk, 32 * curBlock);
                    }
                    curBlock = 0;
                }
            }
            young = storesNow;
        }
        itemChunk0 += q.nch;
        strip += q.stepS; plane += q.stepP;
        if (strip >= q.ns) { strip -= q.ns; ++plane; }
    }
    if (LVG_SDUMMY_V || LVG_SDUMMY_M)
    {
        float r = dumAcc[0] + dumAcc[5];
        for (int i = 0; i < 8; i++) r += (float)dumV[i][0];
        if (r == 12345.678f) *(float*)p.y = r;                               // (keeps the dummy work alive)
    }
}

template <int UP, int DOWN, int FU, int FD, int NBC, int WPB>
int launch_strip(FlreluArgs& a, int mode, hipStream_t stream)
{
    typedef SGeo<UP, DOWN, FU, FD, NBC, LVG_SIGNS_READ> GR;
    typedef SGeo<UP, DOWN, FU, FD, NBC, LVG_SIGNS_WRITE> GW;
    constexpr int KU = FU / UP;
    StripArgs q;
    q.a = a;
    const FlreluArgs& p = q.a;
    if (mode != LVG_SIGNS_READ && !(p.slope <= 1.0f)) return LVG_ERR_UNSUPPORTED;
    // whole contiguous planes of even width (rows and planes start on dword boundaries)
    if (p.xs[3] != 1 || p.xs[2] != p.xw || (p.xw & 1) || (((uintptr_t)p.x) & 3)) return LVG_ERR_UNSUPPORTED;
    if (p.xs[1] != (int64_t)p.xh * p.xw || p.xs[0] != (int64_t)p.c * p.xh * p.xw) return LVG_ERR_UNSUPPORTED;
    if (p.ys[3] != 1 || p.ys[2] != p.yw || p.ys[1] != (int64_t)p.yh * p.yw || p.ys[0] != (int64_t)p.c * p.yh * p.yw || (p.yw & 1) || (((uintptr_t)p.y) & 3)) return LVG_ERR_UNSUPPORTED;
    if ((int64_t)p.yh * p.yw * 2 >= 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    if ((int64_t)p.xh * p.xw * 2 >= 0x7fffffffLL || (int64_t)p.sH * p.sWBytes >= 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    if (mode != LVG_SIGNS_NONE && ((p.sWBytes & 3) || (((uintptr_t)p.s) & 3))) return LVG_ERR_UNSUPPORTED;
    const int64_t planes = (int64_t)p.n * p.c;
    q.ns = (p.yw + kTW - 1) / kTW;
    if (planes * q.ns > 0x3fffffffLL) return LVG_ERR_UNSUPPORTED;
    q.planes = (int)planes;
    q.items = q.planes * q.ns;
    q.phX = ((UP - 1 - p.px0) % UP + UP) % UP;
    q.phY = ((UP - 1 - p.py0) % UP + UP) % UP;
    q.inX0 = lvg_floor_div(UP - 1 - p.px0, UP);
    q.inY0 = lvg_floor_div(UP - 1 - p.py0, UP);
    // DMA origins: the first strip of a plane starts at a multiple of 8 columns that is <= its first column and <= 0 when the
    // padding reaches left of the image (pieces are whole inside or whole outside at the left edge); the ring column of its first
    // column splits into a transpose-read base (0 or 4) and a fragment shift (0..3). The other strips start at their first column
    // rounded down to even (dword-aligned addresses) and shift their fragments by its parity.
    q.org0 = 8 * lvg_floor_div(q.inX0, 8);
    if (q.inX0 >= 0) q.org0 = q.inX0 & ~7;
    q.base0 = (q.inX0 - q.org0) & 4;
    q.ef0 = (q.inX0 - q.org0) & 3;
    q.ef1 = q.inX0 & 1;
    if (UP == 4 && (q.ef0 > 2 || q.ef1 > 2)) return LVG_ERR_UNSUPPORTED;
    if (q.inX0 + GW::SPITCH < 0) return LVG_ERR_UNSUPPORTED;                 // (only a plane's first strip may start left of the image)
    const int vNeeded = (p.yh - 1) * DOWN + FD;
    q.nvb = (vNeeded + 31) / 32;
    {
        q.ayTop = 0; q.ayBot = q.nvb;
        for (int b = 0; b < q.nvb; b++)
        {
            const int r0 = q.inY0 + (32 * b) / UP, r1 = r0 + (31 + UP - 1) / UP + KU;
            if (r0 < 0) q.ayTop = b + 1;
            if (r1 > p.xh && b < q.ayBot) q.ayBot = b;
        }
        if (q.ayBot < q.ayTop) q.ayBot = q.ayTop;
        if (UP == 4) { q.ayTop = q.nvb; q.ayBot = q.nvb; }
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
    // every column a valid output can multiply must be inside the ring row: first column + IN_NX - 3 + shift <= 8 PPR
    {
        const int need0 = q.base0 + q.ef0 + GW::IN_NX - 3, need1 = q.ef1 + GW::IN_NX - 3;
        if (need0 > 8 * GW::PPR || need1 > 8 * GW::PPR) return LVG_ERR_UNSUPPORTED;
    }

    static int cus[64] = {0};
    int dev = 0; (void)hipGetDevice(&dev);
    int ncu = cus[dev & 63];
    if (ncu == 0)
    {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev & 63] = ncu;
    }
    const int offWave = mode == LVG_SIGNS_READ ? GR::OFF_WAVE : GW::OFF_WAVE;
    const size_t lds = (size_t)offWave + (size_t)WPB * GW::WAVE_BYTES;
    if (lds > 160 * 1024) return LVG_ERR_UNSUPPORTED;
    static const int gridEnv = []() { const char* ev = getenv("LVG_FLRELU_STRIP_MAXGRID"); return ev ? atoi(ev) : 0; }();   // (tests: several items per wave on small tensors)
    const int threads = 64 * WPB;
    #define LVG_STRIP_LAUNCH(M) do { \
        auto kern = filtered_lrelu_strip_kernel<UP, DOWN, FU, FD, NBC, M, WPB>; \
        static std::atomic<uint64_t> attr_done{0}; \
        const uint64_t bit_ = 1ull << (dev & 63); \
        if (!(attr_done.load(std::memory_order_acquire) & bit_)) { \
            hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e1 != hipSuccess) { lvg_set_error("filtered_lrelu: cannot reserve LDS for the strip kernel: %s", hipGetErrorString(e1)); return LVG_ERR_LAUNCH; } \
            attr_done.fetch_or(bit_, std::memory_order_release); } \
        static std::atomic<uint64_t> occ_key{0}; static std::atomic<int> occ_val{0}; \
        const uint64_t key_ = ((uint64_t)(dev & 63) << 56) | ((uint64_t)threads << 32) | (uint64_t)lds; \
        int perCu = occ_key.load(std::memory_order_acquire) == key_ ? occ_val.load(std::memory_order_relaxed) : 0; \
        if (perCu < 1) { \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, (const void*)kern, threads, lds) != hipSuccess || perCu < 1) perCu = 1; \
            occ_val.store(perCu, std::memory_order_relaxed); occ_key.store(key_, std::memory_order_release); } \
        int64_t maxGrid = (int64_t)ncu * perCu; \
        if (gridEnv > 0 && gridEnv < maxGrid) maxGrid = gridEnv; \
        int64_t wantWg = ((int64_t)q.items + WPB - 1) / WPB; \
        const unsigned grid = (unsigned)(wantWg < maxGrid ? wantWg : maxGrid); \
        q.waves = (int)grid * WPB; q.rr = DOWN == 4 ? 0 : 1; \
        { static const char* ev_ = getenv("LVG_FLRELU_STRIP_RR"); if (ev_) q.rr = atoi(ev_) ? 1 : 0; }      /* (A/B measurements) */ \
        q.stepP = q.rr ? q.waves / q.ns : 0; q.stepS = q.rr ? q.waves % q.ns : 1; \
        if (getenv("LVG_FLRELU_DEBUG")) fprintf(stderr, "filtered_lrelu_strip: up %d down %d mode %d: planes %d, strips %d, v-blocks %d, chunks %d, org0 %d base0 %d ef %d/%d, lds %zu, %d workgroups/CU, grid %u x %d waves\n", \
            UP, DOWN, M, q.planes, q.ns, q.nvb, q.nch, q.org0, q.base0, q.ef0, q.ef1, lds, perCu, grid, WPB); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, q); } while (0)
    if (mode == LVG_SIGNS_WRITE)     LVG_STRIP_LAUNCH(LVG_SIGNS_WRITE);
    else if (mode == LVG_SIGNS_READ) LVG_STRIP_LAUNCH(LVG_SIGNS_READ);
    else                             LVG_STRIP_LAUNCH(LVG_SIGNS_NONE);
    #undef LVG_STRIP_LAUNCH
    return lvg_check_launch("filtered_lrelu_strip_kernel");
}

} // namespace

int lvg_flrelu_strip_launch(FlreluArgs& p, int cfg, int mode, int dtype, int all, hipStream_t stream)
{
    if (dtype != LVG_F16) return LVG_ERR_UNSUPPORTED;
    if (!all)
    {
        // Measured against the band / wave kernels on the launches of the sres step (16 frames, cold operands; profiles/r06_sres_ab_routed.log,
        // r06_sres_ab_all.log): faster on planes of up to four strips -- up 2 / down 4 backward at output widths 38, 54 and 86 (120 -> 109, 249 -> 176,
        // 330 -> 309 us; items in contiguous runs there, round robin measured 356), up 4 / down 2 forward at width 84 (134 -> 96 us),
        // up 2 / down 2 at widths 84 / 86 forward (158 -> 102 us) and backward (138 -> 118 us) -- and slower on the wide planes, where its 96-byte
        // row pieces and 8-byte stores cost more than the row-band kernel's barrier.
        bool take = false;
        if (cfg == LVG_FLRELU_CFG_U2D4) take = p.yw <= 96;
        else take = p.yw > 72 && p.yw <= 96;          // (width 54 = 24 + 24 + 6: slower, 67 against 56 us)
        if (!take) return LVG_ERR_UNSUPPORTED;
    }
    switch (cfg)
    {
        case LVG_FLRELU_CFG_U2D2: return launch_strip<2, 2, 12, 12, 2, 16>(p, mode, stream);
        case LVG_FLRELU_CFG_U4D2: return launch_strip<4, 2, 24, 12, 2, 16>(p, mode, stream);
        case LVG_FLRELU_CFG_U2D4: return launch_strip<2, 4, 12, 24, 4, 12>(p, mode, stream);
    }
    return LVG_ERR_UNSUPPORTED;
}
