// conv2d_wgrad.hip -- weight gradient of the 3 x 3 'valid' frames convolution (conv2d_igemm.hip) on the gfx950 matrix cores:
// `lvg_conv2d_frames_wgrad`.
//
//   gw[dh][dw][co][ci] = sum over frames n and pixels (oy, ox) of  dy[n][oy][ox][co] * x[n][oy + dh][ox + dw][ci]
//
// what autograd derives for the convolution inside the reference's `modulated_conv2d` (model/generator_sres.py:63-66,
// conv2d_gradfix.conv2d = F.conv2d on torch >= 1.11, torch_utils/ops/conv2d_gradfix.py:37-45) once the zero border of the padded
// convolution is written explicitly. Both tensors are channels-last frames whose zero padding lives in MEMORY:
//   dy [n][Hd][Wd][Co]   Hd = 4 * patchesY, Wd = 16 * patchesX; rows / columns past the true gradient are zeros (written by the
//                        layout kernel that produces dy, csrc/modconv2d_layout.hip)
//   x  [n][Hx][Wx][Ci]   Hx >= Hd + 2, Wx >= Wd + 2 (the padded input frame of the forward pass; finite everywhere)
// so the kernel has no masks and no frame-edge cases at all.
//
// GEMM view as in conv3d_wgrad.hip: D[co][ci] = A[co][k] * B[k][ci] with k = the pixel index, operands staged as they lie
// (LDS-DMA, rows = pixels) and read through the transpose read `ds_read_b64_tr_b16`; one workgroup (4 waves) owns a 64 x 64
// (co, ci) tile for all nine taps (9 x 16 accumulator registers per wave) over a range of K-steps (split K, partial sums added
// by the caller in a fixed order). What differs is the K-step: frames here are up to 290 pixels wide, so a K-step is a
// 4 x 16 PIXEL PATCH of dy (64 pixels) and the x band is its 6 x 18 input patch: 6 LDS slots (one per patch row) of pitch 20 rows
// (a multiple of 4: the three vertical taps are immediate offsets of one lane address, like the slots of conv3d_wgrad.hip). Every
// K-step has the same geometry, so the per-lane DMA source offsets are constants relative to a scalar patch origin.
// Roofline: MFMA-bound, 2 * n * Hd * Wd * Co * Ci * 9 FLOP per launch.

#include "wgrad_common.h"

#ifndef LVG_WGRAD_XCD
#define LVG_WGRAD_XCD 1
#endif

namespace {

struct Wgrad2DArgs
{
    const void* x;
    const void* dy;
    float*      part;         // [splits][9][Co][Ci]
    int         Hx, Wx, Hd, Wd;
    int         patchesX, patchesY;
    int         Ci, Co;
    int         xStride, dyStride;     // elements between consecutive pixels
    int64_t     groups;       // K-steps in total: n * patchesY * patchesX
    int64_t     groupsPerSplit;
    int         nct, nit;     // Co / 64, Ci / 64
};

constexpr int kRow2 = 128;                // bytes per LDS row: 64 channels
constexpr int kDyBytes2 = 64 * kRow2;     // dy tile: 64 pixels
constexpr int kPW = 16, kPH = 4;          // patch of dy
constexpr int kSlotPitch = 20;            // LDS rows per band slot: 18 patch columns + 2 spare
constexpr int kSlots = kPH + 2;
constexpr int kBandRows2 = kSlots * kSlotPitch;       // 120 rows = 15 DMA pieces
constexpr int kBandBytes2 = kBandRows2 * kRow2;
constexpr int kBandPieces = kBandRows2 / 8;

template <class T>
__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(Wgrad2DArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int slotBytes = kSlotPitch * kRow2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave >> 1, ib = wave & 1;                        // this wave's 32 x 32 quadrant of the 64 x 64 tile

    // workgroup -> (split, co tile, ci tile); tiles fastest: the workgroups sharing a pixel range run together
#if LVG_WGRAD_XCD
    // XCD-aware order (the dispatcher puts workgroup b on XCD b % 8): every XCD gets a CONTIGUOUS range of (range of K-steps, tile) pairs,
    // tiles fastest, so the workgroups that walk the same pixels -- and re-read each other's operand lines -- share one L2 (measured
    // before: 1.94 GB fetched per launch against 0.54 GB of operands, profiles/r03_traffic_sres.json)
    int bid;
    {
        const int nwg = gridDim.x, b = blockIdx.x, q = nwg >> 3, rm = nwg & 7, xcd = b & 7;
        bid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (b >> 3);
    }
#else
    int bid = blockIdx.x;
#endif
    const int it = bid % p.nit; bid /= p.nit;
    const int ct = bid % p.nct;
    const int split = bid / p.nct;
    const int64_t g0 = (int64_t)split * p.groupsPerSplit;
    const int64_t g1 = min(g0 + p.groupsPerSplit, p.groups);

    // LDS: [dy tile 0 | dy tile 1 | band 0 | band 1]
    const uint32_t ldsBase = (uint32_t)(uintptr_t)smem;
    const uint32_t dyBase = ldsBase, bandBase = ldsBase + 2u * kDyBytes2;

    const unsigned char* const xb = static_cast<const unsigned char*>(p.x) + (size_t)it * 64 * 2;
    const unsigned char* const dyb = static_cast<const unsigned char*>(p.dy) + (size_t)ct * 64 * 2;
    const uint32_t xRowB = (uint32_t)p.xStride * 2, dyRowB = (uint32_t)p.dyStride * 2;

    // Per-lane source offsets of this wave's DMA pieces relative to the patch origin (the same for every K-step).
    // dy: piece i = LDS rows 8 i .. 8 i + 7 = patch row i >> 1, columns 8 (i & 1) .. + 7; wave w brings in pieces w and w + 4.
    // band: piece q = LDS rows 8 q .. 8 q + 7 of the band; LDS row r = slot r / 20 (patch row), column r % 20 (columns 18, 19 are
    // spare rows nobody reads: clamped to column 17); wave w brings in pieces w, w + 4, w + 8, w + 12 (< 15).
    const uint32_t pieceRow = (uint32_t)(lane >> 3);
    uint32_t dyOff[2], xOff[4];
    #pragma unroll
    for (int k = 0; k < 2; k++)
    {
        const uint32_t i = (uint32_t)(wave + 4 * k);
        const uint32_t ldsRow = i * 8 + pieceRow;
        const uint32_t c = (uint32_t)(lane & 7) ^ swz(ldsRow);
        dyOff[k] = ((i >> 1) * (uint32_t)p.Wd + (i & 1u) * 8u + pieceRow) * dyRowB + c * 16;
    }
    #pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const uint32_t q = (uint32_t)(wave + 4 * k);
        const uint32_t ldsRow = q * 8 + pieceRow;
        const uint32_t slot = ldsRow / kSlotPitch;
        uint32_t col = ldsRow - slot * kSlotPitch;
        col = col > (uint32_t)(kPW + 1) ? (uint32_t)(kPW + 1) : col;
        const uint32_t c = (uint32_t)(lane & 7) ^ swz(ldsRow);
        xOff[k] = (slot * (uint32_t)p.Wx + col) * xRowB + c * 16;
    }
    // Stage the operands of the K-step whose patch is (n, py, px) into buffer set `buf`.
    auto stage = [&](int n, int py, int px, int buf)
    {
        const unsigned char* dyp = dyb + ((size_t)((int64_t)n * p.Hd + py * kPH) * p.Wd + (size_t)px * kPW) * dyRowB;
        const unsigned char* xp = xb + ((size_t)((int64_t)n * p.Hx + py * kPH) * p.Wx + (size_t)px * kPW) * xRowB;
        #pragma unroll
        for (int k = 0; k < 2; k++)
            wdma16(dyp, dyOff[k], dyBase + buf * kDyBytes2 + (wave + 4 * k) * 8 * kRow2);
        #pragma unroll
        for (int k = 0; k < 4; k++)
            if (wave + 4 * k < kBandPieces)
                wdma16(xp, xOff[k], bandBase + buf * kBandBytes2 + (wave + 4 * k) * 8 * kRow2);
    };

    // ---- per-lane fragment addresses (relative to the buffer of the step): compile-time geometry of a 16-pixel-wide patch ----------
    const int g = lane >> 5, s16 = lane & 15, hgrp = (lane >> 4) & 1;
    const uint32_t colA = (uint32_t)(cb * 32 + 16 * hgrp + 4 * (s16 & 3)), colB = (uint32_t)(ib * 32 + 16 * hgrp + 4 * (s16 & 3));
    uint32_t aAddr[4], bAddr[4][3];
    #pragma unroll
    for (int ks = 0; ks < 4; ks++)
    {
        const int px = 16 * ks + 8 * g;                              // first pixel of this lane's chunk: patch row ks, column 8 g
        aAddr[ks] = tr_addr((uint32_t)(px + (s16 >> 2)), colA);
        #pragma unroll
        for (int dw = 0; dw < 3; dw++)                               // slot ks = the patch row of tap dh = 0; column of tap dw = pixel column + dw
            bAddr[ks][dw] = tr_addr((uint32_t)(ks * kSlotPitch + 8 * g + dw + (s16 >> 2)), colB);
    }

    f32x16 acc[9];
    #pragma unroll
    for (int t = 0; t < 9; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // patch of the first K-step
    const int perFrame = p.patchesX * p.patchesY;
    int n = (int)(g0 / perFrame);
    int rem = (int)(g0 - (int64_t)n * perFrame);
    int py = rem / p.patchesX, px = rem - py * p.patchesX;
    if (g0 < g1) stage(n, py, px, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int64_t grp = g0; grp < g1; grp++)
    {
        const int buf = (int)((grp - g0) & 1);
        if (++px == p.patchesX) { px = 0; if (++py == p.patchesY) { py = 0; n++; } }
        if (grp + 1 < g1) stage(n, py, px, buf ^ 1);

        const uint32_t dyBuf = dyBase + buf * kDyBytes2, bandBuf = bandBase + buf * kBandBytes2;
        if constexpr (kWgradPrio) __builtin_amdgcn_s_setprio(1);      // the arithmetic outranks the other workgroup's staging on the SIMD
        #pragma unroll
        for (int ks = 0; ks < 4; ks++)
        {
            const uint4 a = tr_read8(dyBuf + aAddr[ks]);
            #pragma unroll
            for (int dw = 0; dw < 3; dw++)
            {
                const uint32_t col = bandBuf + bAddr[ks][dw];
                #pragma unroll
                for (int dh = 0; dh < 3; dh++)
                {
                    const uint4 b = tr_read8(col + dh * slotBytes);
                    acc[dh * 3 + dw] = MmaW<T>::run(a, b, acc[dh * 3 + dw]);
                }
            }
        }
        if constexpr (kWgradPrio) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // partial sums of this range of K-steps: part[split][tap][co][ci]
    float* out = p.part + ((int64_t)split * 9) * ((int64_t)p.Co * p.Ci);
    const int ci = it * 64 + ib * 32 + (lane & 31);
    #pragma unroll
    for (int t = 0; t < 9; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const int co = ct * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(int64_t)t * p.Co * p.Ci + (int64_t)co * p.Ci + ci] = acc[t][r];
        }
}

struct WPlan2D { int patchesX, patchesY, ldsBytes; int64_t groups; };

bool wgrad2d_plan(int64_t n, int hx, int wx, int hd, int wd, int ci, int co, int kh, int kw, WPlan2D& pl)
{
    if (kh != 3 || kw != 3 || n < 1) return false;
    if (ci <= 0 || co <= 0 || ci % 64 != 0 || co % 64 != 0) return false;
    if (hd < kPH || wd < kPW || hd % kPH != 0 || wd % kPW != 0 || hx < hd + 2 || wx < wd + 2) return false;
    pl.patchesX = wd / kPW;
    pl.patchesY = hd / kPH;
    pl.groups = n * pl.patchesX * pl.patchesY;
    pl.ldsBytes = 2 * kDyBytes2 + 2 * kBandBytes2;
    return true;
}

int compute_units2d()
{
    static int cache[64] = {0};                                        // per device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }   // MI355X (also what a box without a GPU plans for)
    int& slot = cache[dev & 63];
    if (slot == 0)
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        {
            (void)hipGetLastError();
            cus = 256;
        }
        slot = cus;
    }
    return slot;
}

// Split K for ONE full round of workgroups, rounded down (the rule measured for conv3d_wgrad.hip: a partial second round costs 1.3x,
// excess splits only partial-sum traffic): two workgroups per CU by registers. LVG_WGRAD2D_SPLITS overrides (measurements).
int wgrad2d_splits(const WPlan2D& pl, int ci, int co)
{
    static const char* const f = getenv("LVG_WGRAD2D_SPLITS");         // measurement override: read once per process
    const int64_t tiles = (int64_t)(ci / 64) * (co / 64);
    const int64_t slots = (int64_t)compute_units2d() * 2;
    int64_t s = slots / tiles;
    if (f && *f) s = atoi(f);
    s = std::max<int64_t>(1, std::min<int64_t>(s, lvg_ceil_div(pl.groups, 8)));   // at least 8 K-steps per workgroup
    return (int)s;
}

} // namespace

extern "C" int lvg_conv2d_frames_wgrad_splits(int64_t n, int hx, int wx, int hd, int wd, int ci, int co, int kh, int kw)
{
    WPlan2D pl;
    if (!wgrad2d_plan(n, hx, wx, hd, wd, ci, co, kh, kw, pl)) return 0;
    return wgrad2d_splits(pl, ci, co);
}

extern "C" int lvg_conv2d_frames_wgrad(const void* x, const void* dy, float* part,
                                       int64_t n, int hx, int wx, int hd, int wd, int ci, int co, int kh, int kw,
                                       int64_t x_pixel_stride, int64_t dy_pixel_stride, int splits, int dtype, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "conv2d_frames_wgrad: float16 / bfloat16 only (dtype %d)", dtype);
    WPlan2D pl;
    if (!wgrad2d_plan(n, hx, wx, hd, wd, ci, co, kh, kw, pl))
    {
        lvg_set_error("conv2d_frames_wgrad: no kernel for Ci=%d Co=%d taps=%dx%d, gradient frames %dx%d, input frames %dx%d (3 x 3 taps, channels %% 64, "
                      "gradient frames of 4 x 16 pixel patches, input frames at least 2 larger)", ci, co, kh, kw, hd, wd, hx, wx);
        return LVG_ERR_UNSUPPORTED;
    }
    if (x_pixel_stride == 0) x_pixel_stride = ci;
    if (dy_pixel_stride == 0) dy_pixel_stride = co;
    LVG_REQUIRE(x_pixel_stride >= ci && dy_pixel_stride >= co && x_pixel_stride % 8 == 0 && dy_pixel_stride % 8 == 0, "conv2d_frames_wgrad: bad pixel strides");
    LVG_REQUIRE(x && dy && part && lvg_aligned16(x) && lvg_aligned16(dy) && lvg_aligned16(part), "conv2d_frames_wgrad: pointers must be 16-byte aligned");
    LVG_REQUIRE(splits == wgrad2d_splits(pl, ci, co), "conv2d_frames_wgrad: splits must be lvg_conv2d_frames_wgrad_splits(...) (= %d)", wgrad2d_splits(pl, ci, co));
    // per-lane offsets inside a patch are 32-bit
    LVG_REQUIRE((int64_t)(kSlots + 1) * wx * x_pixel_stride * 2 < ((int64_t)1 << 31) && (int64_t)(kPH + 1) * wd * dy_pixel_stride * 2 < ((int64_t)1 << 31),
                "conv2d_frames_wgrad: frame rows too long for 32-bit patch offsets");
    Wgrad2DArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.part = part;
    a.Hx = hx; a.Wx = wx; a.Hd = hd; a.Wd = wd;
    a.patchesX = pl.patchesX; a.patchesY = pl.patchesY;
    a.Ci = ci; a.Co = co;
    a.xStride = (int)x_pixel_stride; a.dyStride = (int)dy_pixel_stride;
    a.groups = pl.groups;
    a.groupsPerSplit = lvg_ceil_div(pl.groups, splits);
    a.nct = co / 64; a.nit = ci / 64;
    const int64_t blocks = (int64_t)a.nct * a.nit * splits;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LVG_BF16) hipLaunchKernelGGL(conv2d_wgrad_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), pl.ldsBytes, s, a);
    else                   hipLaunchKernelGGL(conv2d_wgrad_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), pl.ldsBytes, s, a);
    return lvg_check_launch("conv2d_frames_wgrad");
}
