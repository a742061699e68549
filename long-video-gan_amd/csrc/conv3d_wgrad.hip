// conv3d_wgrad.hip -- weight gradient of the frames convolution (conv3d_igemm.hip) on the gfx950 matrix cores.
//
//   gw[dt][dh][dw][co][ci] = sum over pixels m of  dy[m][co] * x[m + (dt-kt/2)*tShift + (dh-1)*W + (dw-1)][ci]
//   (3 x 3 spatial taps, 'same' zero padding in time / rows / columns: out-of-frame sources contribute nothing)
//
// what autograd derives for F.conv3d of the reference's temporal_modulated_conv3d (model/generator_lres.py:119) and
// Conv3dLayer (discriminator_lres.py:169). GEMM view: D[co][ci] = A[co][k] * B[k][ci] with k = the PIXEL index: both
// operands are needed transposed with respect to channels-last memory ([pixel][channel]); they are staged as they lie
// (LDS-DMA, rows = pixels) and read through the gfx950 transpose read `ds_read_b64_tr_b16`.
//
// One workgroup (4 waves) owns a 64 (co) x 64 (ci) tile of ONE temporal tap for ALL nine spatial taps (9 x 16
// accumulator registers per wave: the dy fragment is shared by the nine products) over a range of pixels (split K);
// the partial sums of the ranges go to `part[split]` and are added in a fixed order by the caller (reproducible; no
// atomics). A K-step is a group of 64 pixels = 64 / W whole image rows:
//   * dy tile [64 pixels][64 co];
//   * x band: the image rows of the group plus one neighbour row above and below, laid out in LDS with a ZERO row
//     between image rows and a zero slot wherever the neighbour is outside the frame (or the source frame outside the
//     clip): a spatial tap is then a plain address offset, no masks anywhere in the arithmetic. The slot pitch is W + 4
//     rows (a multiple of 4: the swizzle phase of an address survives a move by whole slots), so the three vertical taps are
//     immediate offsets of one lane address and the addresses themselves are per-lane constants plus a scalar per step.
// Bank layout: 16-byte chunk c of LDS row r lives at chunk c ^ (4 * ((r >> 1) & 1)) (applied to the DMA source
// address): the four rows x 32 bytes a 16-lane group of the transpose read touches are spread over all 64 banks.

#include "wgrad_common.h"

#ifndef LVG_WGRAD_XCD
#define LVG_WGRAD_XCD 1
#endif

namespace {

struct WgradArgs
{
    const void* x;
    const void* dy;
    float*      part;         // [splits][kt][9][Co][Ci]
    const void* zeros;        // >= 128 bytes of zeros in device memory
    int64_t     frames;       // frames of the clip batch (time-major: frame f = t * clips + n)
    int64_t     frameShift;   // frames between consecutive time steps (= clips)
    int         H, W, Ci, Co, kt;
    int         xStride, dyStride;     // elements between consecutive pixels
    int         maxSlots;     // LDS band slots (image rows incl. neighbours and zero slots)
    int64_t     groups;       // K-steps in total: ceil(frames * H / R)
    int64_t     groupsPerSplit;
    int         nct, nit;     // Co / 64, Ci / 64
};

// K-step = 64 pixels = R = 64 / W whole image rows. LDS band: slot s (an image row, a neighbour row or zeros) occupies rows
// [s * P, (s + 1) * P) with P = W + 4: row 0 zero, rows 1 .. W the pixels, rows W + 1 .. W + 3 zero. P is a multiple of
// 4, so moving by whole slots keeps the swizzle phase of a lane address: the three vertical taps are IMMEDIATE offsets of
// one address, and the frame-change shift of a step is a scalar added to per-lane constants.
template <class T, int W>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(WgradArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int R = 64 / W, P = W + 4, wq = W / 8;
    constexpr int slotBytes = P * kRow;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave >> 1, ib = wave & 1;                        // this wave's 32 x 32 quadrant of the 64 x 64 tile
    const int H = p.H;

    // workgroup -> (split, temporal tap, co tile, ci tile); tiles fastest: the workgroups sharing a pixel range run together
#if LVG_WGRAD_XCD
    // XCD-aware order (the dispatcher puts workgroup b on XCD b % 8): every XCD gets a CONTIGUOUS range of (range of K-steps, tile) pairs,
    // tiles fastest, so the workgroups that walk the same pixels -- and re-read each other's operand lines -- share one L2 (measured
    // before: 1.94 GB fetched per launch against 0.54 GB of operands, the 2-D kernel, profiles/r03_traffic_sres.json)
    int bid;
    {
        const int nwg = gridDim.x, b = blockIdx.x, q = nwg >> 3, rm = nwg & 7, xcd = b & 7;
        bid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (b >> 3);
    }
#else
    int bid = blockIdx.x;
#endif
    const int it = bid % p.nit; bid /= p.nit;
    const int ct = bid % p.nct; bid /= p.nct;
    const int dt = bid % p.kt;
    const int split = bid / p.kt;
    const int64_t g0 = (int64_t)split * p.groupsPerSplit;
    const int64_t g1 = min(g0 + p.groupsPerSplit, p.groups);
    const int64_t df = (int64_t)(dt - (p.kt >> 1)) * p.frameShift;  // source frame = frame + df
    const int64_t rowsTotal = p.frames * H;

    // LDS: [dy tile 0 | dy tile 1 | band 0 | band 1]
    const int bandBytes = p.maxSlots * slotBytes;
    const uint32_t ldsBase = (uint32_t)(uintptr_t)smem;
    const uint32_t dyBase = ldsBase, bandBase = ldsBase + 2u * kDyBytes;

    // zero rows of both bands (written once: the DMA pieces never touch them): rows 0 and W + 1 .. W + 3 of every slot
    for (int i = tid; i < 2 * p.maxSlots * 4 * 8; i += 256)
    {
        const int c = i & 7, zr = (i >> 3) & 3, slot = i >> 5;                       // slot over both bands (contiguous)
        const int row = slot * P + (zr == 0 ? 0 : W + zr);
        *reinterpret_cast<uint4*>(smem + 2 * kDyBytes + row * kRow + c * 16) = make_uint4(0, 0, 0, 0);
    }

    const unsigned char* const xb = static_cast<const unsigned char*>(p.x) + (size_t)it * 64 * 2;
    const unsigned char* const dyb = static_cast<const unsigned char*>(p.dy) + (size_t)ct * 64 * 2;
    const unsigned char* const zb = static_cast<const unsigned char*>(p.zeros);
    const uint32_t xRowB = (uint32_t)p.xStride * 2, dyRowB = (uint32_t)p.dyStride * 2;
    const int64_t pixels = rowsTotal * W;

    // per-lane pieces of a DMA: row inside the piece, logical chunk for an even / odd-pair row phase
    const uint32_t pieceRow = (uint32_t)(lane >> 3);
    auto dma_piece = [&](const unsigned char* base, int64_t firstPixel, uint32_t rowStep, uint32_t ldsAddr, uint32_t ldsRow)
    {
        const uint32_t c = (uint32_t)(lane & 7) ^ swz(ldsRow + pieceRow);            // logical chunk this lane must fetch
        if (firstPixel >= 0) wdma16(base + (size_t)firstPixel * rowStep, pieceRow * rowStep + c * 16, ldsAddr);
        else                 wdma16(zb, (uint32_t)(lane & 7) * 16, ldsAddr);
    };
    // Stage the operands of K-step `g` into buffer set `buf`. (f0, y0) = frame and row inside it of the step's first image row.
    // Wave w brings in dy pieces w and w + 4 and the band slots s = w, w + 4, ...: slot s holds row r = s - 1 - (frame changes
    // before it) of the step (r = -1 / R: the neighbour above / below), or zeros at a frame change, outside the frame or
    // outside the clip.
    auto stage = [&](int64_t g, int64_t f0, int y0, int buf)
    {
        #pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const int i = wave + 4 * k;
            int64_t first = g * 64 + i * 8;
            if (first >= pixels) first = -1;                         // (whole pieces: the pixel count is a multiple of 8)
            dma_piece(dyb, first, dyRowB, dyBase + buf * kDyBytes + i * 8 * kRow, (uint32_t)(i * 8));
        }
        const int t1 = H - y0, t2 = 2 * H - y0;                      // rows r >= t1 / t2 of the step lie one / two frames further
        for (int slot = wave; slot < p.maxSlots; slot += 4)
        {
            const int q = slot - 1;
            int r, fc;                                               // row of the step, frame changes before it
            bool real;
            if (q < t1)           { r = q;     fc = 0; real = q >= 0 || y0 > 0; }
            else if (q == t1)     { r = 0;     fc = 0; real = false; }
            else if (q <= t2)     { r = q - 1; fc = 1; real = true; }
            else if (q == t2 + 1) { r = 0;     fc = 0; real = false; }
            else                  { r = q - 2; fc = 2; real = true; }
            if (r > R) continue;                                     // spare slot: nothing reads it
            int64_t rowPixel = -1;
            const int64_t fsrc = f0 + fc + df;                       // source frame
            if (real && f0 + fc < p.frames && fsrc >= 0 && fsrc < p.frames)
                rowPixel = ((fsrc * H) + (y0 + r - fc * H)) * W;
            #pragma unroll
            for (int j = 0; j < wq; j++)
            {
                const uint32_t ldsRow = (uint32_t)(slot * P + 1 + j * 8);
                dma_piece(xb, rowPixel >= 0 ? rowPixel + j * 8 : -1, xRowB, bandBase + buf * bandBytes + ldsRow * kRow, ldsRow);
            }
        }
    };

    // ---- per-lane address constants (relative to the buffer of the step) -------------------------------------------------
    const int g = lane >> 5, s16 = lane & 15, hgrp = (lane >> 4) & 1;
    const uint32_t colA = (uint32_t)(cb * 32 + 16 * hgrp + 4 * (s16 & 3)), colB = (uint32_t)(ib * 32 + 16 * hgrp + 4 * (s16 & 3));
    uint32_t aAddr[4], bAddr[4][3];
    int rowOfChunk[4];                                               // image row (inside the step) of this lane's chunk per sub-step
    #pragma unroll
    for (int ks = 0; ks < 4; ks++)
    {
        const int px = 16 * ks + 8 * g;
        const int r = px / W, x0 = px - r * W;
        rowOfChunk[ks] = r;
        aAddr[ks] = tr_addr((uint32_t)(px + (s16 >> 2)), colA);
        #pragma unroll
        for (int dw = 0; dw < 3; dw++)                               // slot r = the row ABOVE the chunk's row (dh = 0), before frame changes
            bAddr[ks][dw] = tr_addr((uint32_t)(r * P + 1 + x0 + (dw - 1) + (s16 >> 2)), colB);
    }

    f32x16 acc[9];
    #pragma unroll
    for (int t = 0; t < 9; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    int64_t f0 = (g0 * R) / H;                                       // frame / row inside it of the first image row of the step
    int y0 = (int)((g0 * R) - f0 * H);
    if (g0 < g1) stage(g0, f0, y0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int64_t grp = g0; grp < g1; grp++)
    {
        const int buf = (int)((grp - g0) & 1);
        int y1 = y0 + R;
        int64_t f1 = f0;
        while (y1 >= H) { y1 -= H; f1++; }
        if (grp + 1 < g1) stage(grp + 1, f1, y1, buf ^ 1);

        const uint32_t dyBuf = dyBase + buf * kDyBytes, bandBuf = bandBase + buf * bandBytes;
        if constexpr (kWgradPrio) __builtin_amdgcn_s_setprio(1);      // the arithmetic outranks the other workgroup's staging on the SIMD
        #pragma unroll
        for (int ks = 0; ks < 4; ks++)
        {
            // frame changes before this chunk's row shift it down by whole slots
            const int yr = y0 + rowOfChunk[ks];
            const int wraps = (yr >= H) + (yr >= 2 * H);
            const uint32_t bBase = bandBuf + (uint32_t)(wraps * slotBytes);
            const uint4 a = tr_read8(dyBuf + aAddr[ks]);
            #pragma unroll
            for (int dw = 0; dw < 3; dw++)
            {
                const uint32_t col = bBase + bAddr[ks][dw];
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
        y0 = y1; f0 = f1;
    }

    // partial sums of this pixel range: part[split][dt][tap][co][ci]
    float* out = p.part + (((int64_t)split * p.kt + dt) * 9) * ((int64_t)p.Co * p.Ci);
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

struct WPlan { int R, maxSlots, ldsBytes; int64_t groups; };

bool wgrad_plan(int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw, WPlan& pl)
{
    if (kh != 3 || kw != 3 || kt < 1 || kt > 7 || !(kt & 1)) return false;
    if (ci <= 0 || co <= 0 || ci % 64 != 0 || co % 64 != 0) return false;
    if (!(w == 8 || w == 16 || w == 32 || w == 64) || h < 1 || frames < 1) return false;
    pl.R = 64 / w;
    pl.maxSlots = pl.R + 2 + (pl.R - 1) / h + 1;
    pl.groups = lvg_ceil_div(frames * h, pl.R);
    pl.ldsBytes = 2 * kDyBytes + 2 * pl.maxSlots * (w + 4) * kRow;
    return pl.ldsBytes <= 160 * 1024 && frames * h * (int64_t)w * std::max(ci, co) * 2 < ((int64_t)1 << 40);
}

int compute_units()
{
    static int cus[64] = {0};                                          // per device ordinal (a process may drive devices of different sizes)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }   // MI355X (also what a box without a GPU plans for)
    int& slot = cus[dev & 63];
    if (slot == 0)
    {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        {
            (void)hipGetLastError();
            n = 256;
        }
        slot = n;
    }
    return slot;
}

// Split K so that the launch is ONE full round of workgroups, rounded DOWN: 184 registers allow two workgroups per CU, the LDS band
// of 64-pixel-wide frames only one. Measured (profiles/r02_wgrad_splits.log): a launch of 520 workgroups on
// 512 slots takes 1.3x the time of one of 510 (the 8 stragglers run alone), and more splits than slots only add partial-sum traffic
// (1024 splits of the 64-channel layer wrote and re-read 151 MB). LVG_WGRAD_SPLITS / LVG_WGRAD_TARGET override (measurements).
int wgrad_splits(const WPlan& pl, int ci, int co, int kt)
{
    static const char* const f = getenv("LVG_WGRAD_SPLITS");           // measurement overrides: read once per process
    static const char* const tg = getenv("LVG_WGRAD_TARGET");
    const int64_t tiles = (int64_t)(ci / 64) * (co / 64) * kt;
    const int perCU = std::max(1, std::min(2, (160 * 1024) / pl.ldsBytes));
    const int64_t slots = (int64_t)compute_units() * perCU;
    int64_t s = slots / tiles;
    if (tg && *tg) s = lvg_ceil_div(atoi(tg), tiles);
    if (f && *f) s = atoi(f);
    s = std::max<int64_t>(1, std::min<int64_t>(s, lvg_ceil_div(pl.groups, 8)));   // at least 8 K-steps per workgroup
    return (int)s;
}

} // namespace

extern "C" int lvg_conv3d_frames_wgrad_splits(int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw)
{
    WPlan pl;
    if (!wgrad_plan(frames, h, w, ci, co, kt, kh, kw, pl)) return 0;
    return wgrad_splits(pl, ci, co, kt);
}

extern "C" int lvg_conv3d_frames_wgrad(const void* x, const void* dy, float* part, const void* zeros,
                                       int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw, int64_t frame_shift,
                                       int64_t x_pixel_stride, int64_t dy_pixel_stride, int splits, int dtype, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "conv3d_frames_wgrad: float16 / bfloat16 only (dtype %d)", dtype);
    WPlan pl;
    if (!wgrad_plan(frames, h, w, ci, co, kt, kh, kw, pl))
    {
        lvg_set_error("conv3d_frames_wgrad: no kernel for Ci=%d Co=%d taps=%dx%dx%d frames %dx%d (3x3 spatial taps, channels %% 64, width 8 / 16 / 32 / 64)",
                      ci, co, kt, kh, kw, h, w);
        return LVG_ERR_UNSUPPORTED;
    }
    if (x_pixel_stride == 0) x_pixel_stride = ci;
    if (dy_pixel_stride == 0) dy_pixel_stride = co;
    LVG_REQUIRE(x_pixel_stride >= ci && dy_pixel_stride >= co && x_pixel_stride % 8 == 0 && dy_pixel_stride % 8 == 0, "conv3d_frames_wgrad: bad pixel strides");
    LVG_REQUIRE(lvg_aligned16(x) && lvg_aligned16(dy) && lvg_aligned16(part) && lvg_aligned16(zeros) && zeros != nullptr, "conv3d_frames_wgrad: pointers must be 16-byte aligned");
    LVG_REQUIRE(frame_shift > 0 && frames % frame_shift == 0, "conv3d_frames_wgrad: frames must be a multiple of frame_shift");
    LVG_REQUIRE(splits == wgrad_splits(pl, ci, co, kt), "conv3d_frames_wgrad: splits must be lvg_conv3d_frames_wgrad_splits(...) (= %d)", wgrad_splits(pl, ci, co, kt));
    LVG_REQUIRE((int64_t)frames * h * w * std::max(x_pixel_stride, dy_pixel_stride) * 2 < ((int64_t)1 << 40), "conv3d_frames_wgrad: tensor too large");
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.part = part; a.zeros = zeros;
    a.frames = frames; a.frameShift = frame_shift;
    a.H = h; a.W = w; a.Ci = ci; a.Co = co; a.kt = kt;
    a.xStride = (int)x_pixel_stride; a.dyStride = (int)dy_pixel_stride;
    a.maxSlots = pl.maxSlots; a.groups = pl.groups;
    a.groupsPerSplit = lvg_ceil_div(pl.groups, splits);
    a.nct = co / 64; a.nit = ci / 64;
    const int64_t blocks = (int64_t)a.nct * a.nit * kt * splits;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto launch = [&](auto kern) -> int
    {
        if (pl.ldsBytes > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        {
            (void)hipGetLastError();
            lvg_set_error("conv3d_frames_wgrad: cannot opt in to %d bytes of LDS", pl.ldsBytes);
            return LVG_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), pl.ldsBytes, s, a);
        return lvg_check_launch("conv3d_frames_wgrad");
    };
    const bool bf = dtype == LVG_BF16;
    switch (w)
    {
    case 8:  return bf ? launch(conv3d_wgrad_kernel<bf16_t, 8>)  : launch(conv3d_wgrad_kernel<f16_t, 8>);
    case 16: return bf ? launch(conv3d_wgrad_kernel<bf16_t, 16>) : launch(conv3d_wgrad_kernel<f16_t, 16>);
    case 32: return bf ? launch(conv3d_wgrad_kernel<bf16_t, 32>) : launch(conv3d_wgrad_kernel<f16_t, 32>);
    default: return bf ? launch(conv3d_wgrad_kernel<bf16_t, 64>) : launch(conv3d_wgrad_kernel<f16_t, 64>);
    }
}
