// upfirdn2d.hip -- zero-insert upsample -> pad/crop -> FIR -> decimate, per channel plane.
//
// Semantics follow the reference plugin (torch_utils/ops/upfirdn2d.cpp:16-98 host side,
// upfirdn2d.cu:29-200 kernels): for output pixel (ox, oy)
//     mid  = o * down + up - 1 - pad0
//     in0  = floor(mid / up)                       first contributing input sample
//     tap0 = (in0 + 1) * up - mid - 1              its tap in the FLIPPED filter
//     y    = gain * sum_k x[in0 + k] * fflip[tap0 + k * up]        (per axis)
// (upfirdn2d.cu:176-193). fflip is f reversed unless `flip` (true convolution by default).
//
// MI355X design (differs from the reference's 94 CUDA specialisations):
//   * one LDS-tiled kernel template <T, UPX, UPY, DOWNX, DOWNY> with RUNTIME tile extents and
//     tap counts: the host picks the output tile to fit the plane (small planes = one tile per
//     plane, no ragged second tile), LDS is sized per launch (160 KiB/CU available);
//   * a separable filter (1-D fx and/or fy) runs BOTH axes in the one launch with the
//     row-filtered intermediate kept in LDS -- the reference makes two launches with an HBM
//     round trip in between (upfirdn2d.py:241-245). HBM traffic = N_in + N_out elements;
//   * loads are coalesced along W (the contiguous axis of NCHW / of the n c t (hw) views the
//     lres models use), arbitrary element strides are honoured so permuted latent views and
//     cropped views need no .contiguous() copy;
//   * a thread-per-output gather kernel covers everything else (up/down not in {1,2,4},
//     huge filters, tiles that would not fit in LDS): same role as upfirdn2d_kernel_large.
//
// Roofline: HBM stream, (N_in + N_out) * sizeof(T) algorithmic bytes per call.

#include "lvg_common.h"

namespace {

struct UpfirdnArgs
{
    const void*  x;
    void*        y;
    const float* f2d;    // dense 2-D taps or NULL
    const float* fx;     // separable taps along W or NULL
    const float* fy;     // separable taps along H or NULL
    int64_t xs[4];       // element strides n, c, h, w
    int64_t ys[4];
    int64_t fsx, fsy;    // strides of f2d
    int n, c, ih, iw, oh, ow;
    int fw, fh;
    int upx, upy, downx, downy;
    int padx0, pady0;
    int flip;
    float gain;
    // tiling (tiled kernel only)
    int tileW, tileH;       // output tile
    int tilesX, tilesY;
    int inTW, inTH;         // input tile incl. halo
    int midTH;              // rows of the row-filtered intermediate (= inTH)
};

constexpr int kGatherThreads = 256;

// ---------------------------------------------------------------------------------------------
// Gather kernel: one output element per thread, everything runtime.

template <class T>
__global__ __launch_bounds__(kGatherThreads) void upfirdn2d_gather_kernel(UpfirdnArgs p)
{
    typedef typename Elem<T>::acc_t A;
    const int64_t total = (int64_t)p.n * p.c * p.oh * p.ow;
    int64_t idx = (int64_t)blockIdx.x * kGatherThreads + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % p.ow); idx /= p.ow;
    const int oy = (int)(idx % p.oh); idx /= p.oh;
    const int ch = (int)(idx % p.c);
    const int nb = (int)(idx / p.c);

    const int midX = ox * p.downx + p.upx - 1 - p.padx0;
    const int midY = oy * p.downy + p.upy - 1 - p.pady0;
    const int inX0 = lvg_floor_div(midX, p.upx);
    const int inY0 = lvg_floor_div(midY, p.upy);
    const int tapX0 = (inX0 + 1) * p.upx - midX - 1;
    const int tapY0 = (inY0 + 1) * p.upy - midY - 1;

    const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1];
    A acc = (A)0;
    for (int ty = tapY0, iy = inY0; ty < p.fh; ty += p.upy, iy++)
    {
        if (iy < 0 || iy >= p.ih) continue;
        const int fyi = p.flip ? ty : p.fh - 1 - ty; // index into the un-flipped filter
        const A wy = p.f2d ? (A)1 : (p.fy ? (A)p.fy[fyi] : (A)1);
        A row = (A)0;
        for (int tx = tapX0, ix = inX0; tx < p.fw; tx += p.upx, ix++)
        {
            if (ix < 0 || ix >= p.iw) continue;
            const int fxi = p.flip ? tx : p.fw - 1 - tx;
            const A w = p.f2d ? (A)p.f2d[fyi * p.fsy + fxi * p.fsx] : (p.fx ? (A)p.fx[fxi] : (A)1);
            row += (A)to_acc(xp[(int64_t)iy * p.xs[2] + (int64_t)ix * p.xs[3]]) * w;
        }
        acc += row * wy;
    }
    acc *= (A)p.gain;
    ((T*)p.y)[(int64_t)nb * p.ys[0] + (int64_t)ch * p.ys[1] + (int64_t)oy * p.ys[2] + (int64_t)ox * p.ys[3]] = from_acc<T>(acc);
}

// ---------------------------------------------------------------------------------------------
// Tiled kernel. Block = 256 threads as 64 (x) x 4 (y). One block = one output tile of one plane.
// LDS layout (floats): [taps X: fwPad][taps Y: fhPad][input tile inTH x inTW]
//                      [separable only: row-filtered tile inTH x tileW]
// 2-D filters are stored as fh x fw flipped taps in the tap area.

constexpr int kTX = 64, kTY = 4;

template <class T, int UPX, int UPY, int DOWNX, int DOWNY, bool SEP>
__global__ __launch_bounds__(kTX * kTY) void upfirdn2d_tiled_kernel(UpfirdnArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * kTX + tx;

    const int nTapX = SEP ? p.fw : p.fw * p.fh;
    const int tapPadX = (nTapX + 3) & ~3;
    const int tapPadY = SEP ? ((p.fh + 3) & ~3) : 0;
    float* sfx = smem;
    float* sfy = smem + tapPadX;
    float* sin = sfy + tapPadY;
    float* smid = sin + p.inTH * p.inTW; // SEP only

    // Flipped taps into LDS.
    if (SEP)
    {
        for (int k = tid; k < p.fw; k += kTX * kTY)
            sfx[k] = p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f;
        for (int k = tid; k < p.fh; k += kTX * kTY)
            sfy[k] = p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f;
    }
    else
    {
        for (int k = tid; k < p.fw * p.fh; k += kTX * kTY)
        {
            const int ky = k / p.fw, kx = k - ky * p.fw;
            const int sy = p.flip ? ky : p.fh - 1 - ky, sx = p.flip ? kx : p.fw - 1 - kx;
            sfx[k] = p.f2d[sy * p.fsy + sx * p.fsx];
        }
    }

    // Which tile / plane.
    int b = blockIdx.x;
    const int tileX = b % p.tilesX; b /= p.tilesX;
    const int tileY = b % p.tilesY; b /= p.tilesY;
    const int ch = b % p.c;
    const int nb = b / p.c;

    const int outX0 = tileX * p.tileW, outY0 = tileY * p.tileH;
    const int midX0 = outX0 * DOWNX + UPX - 1 - p.padx0;
    const int midY0 = outY0 * DOWNY + UPY - 1 - p.pady0;
    const int inX0 = lvg_floor_div(midX0, UPX);
    const int inY0 = lvg_floor_div(midY0, UPY);

    // Load the input tile (zero outside the plane), coalesced along W.
    const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1];
    for (int r = ty; r < p.inTH; r += kTY)
    {
        const int iy = inY0 + r;
        const bool rowOk = (iy >= 0) && (iy < p.ih);
        const T* rowp = xp + (int64_t)iy * p.xs[2];
        for (int q = tx; q < p.inTW; q += kTX)
        {
            const int ix = inX0 + q;
            float v = 0.0f;
            if (rowOk && ix >= 0 && ix < p.iw) v = (float)to_acc(rowp[(int64_t)ix * p.xs[3]]);
            sin[r * p.inTW + q] = v;
        }
    }
    __syncthreads();

    T* yp = (T*)p.y + (int64_t)nb * p.ys[0] + (int64_t)ch * p.ys[1];

    if (SEP)
    {
        // Pass 1: rows. smid[r][ox] = sum_k sin[r][relX + k] * sfx[tap0 + k*UPX]
        for (int r = ty; r < p.inTH; r += kTY)
        {
            for (int ox = tx; ox < p.tileW; ox += kTX)
            {
                const int midX = midX0 + ox * DOWNX;
                const int inX = lvg_floor_div(midX, UPX);
                int tap = (inX + 1) * UPX - midX - 1;
                const float* src = sin + r * p.inTW + (inX - inX0);
                float acc = 0.0f;
                for (; tap < p.fw; tap += UPX) acc += (*src++) * sfx[tap];
                smid[r * p.tileW + ox] = acc;
            }
        }
        __syncthreads();
        // Pass 2: columns.
        for (int oy = ty; oy < p.tileH; oy += kTY)
        {
            const int gy = outY0 + oy;
            if (gy >= p.oh) break;
            const int midY = midY0 + oy * DOWNY;
            const int inY = lvg_floor_div(midY, UPY);
            const int tapY0 = (inY + 1) * UPY - midY - 1;
            for (int ox = tx; ox < p.tileW; ox += kTX)
            {
                const int gx = outX0 + ox;
                if (gx >= p.ow) break;
                const float* src = smid + (inY - inY0) * p.tileW + ox;
                float acc = 0.0f;
                for (int tap = tapY0; tap < p.fh; tap += UPY, src += p.tileW) acc += (*src) * sfy[tap];
                yp[(int64_t)gy * p.ys[2] + (int64_t)gx * p.ys[3]] = from_acc<T>(acc * p.gain);
            }
        }
    }
    else
    {
        for (int oy = ty; oy < p.tileH; oy += kTY)
        {
            const int gy = outY0 + oy;
            if (gy >= p.oh) break;
            const int midY = midY0 + oy * DOWNY;
            const int inY = lvg_floor_div(midY, UPY);
            const int tapY0 = (inY + 1) * UPY - midY - 1;
            for (int ox = tx; ox < p.tileW; ox += kTX)
            {
                const int gx = outX0 + ox;
                if (gx >= p.ow) break;
                const int midX = midX0 + ox * DOWNX;
                const int inX = lvg_floor_div(midX, UPX);
                const int tapX0 = (inX + 1) * UPX - midX - 1;
                const float* srow = sin + (inY - inY0) * p.inTW + (inX - inX0);
                float acc = 0.0f;
                for (int tyy = tapY0; tyy < p.fh; tyy += UPY, srow += p.inTW)
                {
                    const float* src = srow;
                    const float* frow = sfx + tyy * p.fw;
                    for (int txx = tapX0; txx < p.fw; txx += UPX) acc += (*src++) * frow[txx];
                }
                yp[(int64_t)gy * p.ys[2] + (int64_t)gx * p.ys[3]] = from_acc<T>(acc * p.gain);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host side.

constexpr int kMaxLdsBytes = 64 * 1024; // per block: keeps >= 2 blocks per CU resident (160 KiB LDS)

inline int in_extent(int outExtent, int up, int down, int taps)
{
    return ((outExtent - 1) * down + taps - 1) / up + 1 + 1; // +1: floor_div phase slack
}

template <class T, int UPX, int UPY, int DOWNX, int DOWNY>
int launch_tiled(UpfirdnArgs& p, bool sep, hipStream_t stream)
{
    // Output tile: as much of the plane as fits, W first (coalescing), bounded by LDS.
    // Balanced split so the last tile of a row/column is not a sliver.
    int maxW = 128, maxH = 64;
    for (;;)
    {
        const int tileW = (p.ow + (p.ow + maxW - 1) / maxW - 1) / ((p.ow + maxW - 1) / maxW);
        const int tileH = (p.oh + (p.oh + maxH - 1) / maxH - 1) / ((p.oh + maxH - 1) / maxH);
        p.tileW = tileW; p.tileH = tileH;
        p.inTW = in_extent(tileW, UPX, DOWNX, p.fw);
        p.inTH = in_extent(tileH, UPY, DOWNY, p.fh);
        const int64_t taps = sep ? (((p.fw + 3) & ~3) + ((p.fh + 3) & ~3)) : ((p.fw * p.fh + 3) & ~3);
        const int64_t words = taps + (int64_t)p.inTH * p.inTW + (sep ? (int64_t)p.inTH * tileW : 0);
        if (words * 4 <= kMaxLdsBytes) break;
        if (maxH > 8) maxH /= 2;
        else if (maxW > 32) maxW /= 2;
        else return LVG_ERR_UNSUPPORTED; // filter too large for a tile: gather kernel
    }
    p.tilesX = (p.ow + p.tileW - 1) / p.tileW;
    p.tilesY = (p.oh + p.tileH - 1) / p.tileH;
    const int64_t blocks = (int64_t)p.tilesX * p.tilesY * p.n * p.c;
    if (blocks > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    const int64_t taps = sep ? (((p.fw + 3) & ~3) + ((p.fh + 3) & ~3)) : ((p.fw * p.fh + 3) & ~3);
    const size_t lds = (size_t)(taps + (int64_t)p.inTH * p.inTW + (sep ? (int64_t)p.inTH * p.tileW : 0)) * 4;
    if (sep)
        hipLaunchKernelGGL((upfirdn2d_tiled_kernel<T, UPX, UPY, DOWNX, DOWNY, true>), dim3((unsigned)blocks), dim3(kTX, kTY), lds, stream, p);
    else
        hipLaunchKernelGGL((upfirdn2d_tiled_kernel<T, UPX, UPY, DOWNX, DOWNY, false>), dim3((unsigned)blocks), dim3(kTX, kTY), lds, stream, p);
    return lvg_check_launch("upfirdn2d_tiled_kernel");
}

template <class T>
int launch_gather(UpfirdnArgs& p, hipStream_t stream)
{
    const int64_t total = (int64_t)p.n * p.c * p.oh * p.ow;
    const int64_t blocks = lvg_ceil_div(total, kGatherThreads);
    LVG_REQUIRE(blocks <= 0x7fffffffLL, "upfirdn2d: output too large for one launch");
    hipLaunchKernelGGL((upfirdn2d_gather_kernel<T>), dim3((unsigned)blocks), dim3(kGatherThreads), 0, stream, p);
    return lvg_check_launch("upfirdn2d_gather_kernel");
}

#define LVG_UPFIRDN_CASE(ux, uy, dx, dy) \
    if (p.upx == ux && p.upy == uy && p.downx == dx && p.downy == dy) return launch_tiled<T, ux, uy, dx, dy>(p, sep, stream);

template <class T>
int dispatch_tiled(UpfirdnArgs& p, bool sep, hipStream_t stream)
{
    // isotropic resampling (spatial), and H-only resampling (the models' time axis: [n, c, t, hw])
    LVG_UPFIRDN_CASE(1, 1, 1, 1)
    LVG_UPFIRDN_CASE(2, 2, 1, 1)
    LVG_UPFIRDN_CASE(1, 1, 2, 2)
    LVG_UPFIRDN_CASE(4, 4, 1, 1)
    LVG_UPFIRDN_CASE(1, 1, 4, 4)
    LVG_UPFIRDN_CASE(1, 2, 1, 1)
    LVG_UPFIRDN_CASE(1, 1, 1, 2)
    LVG_UPFIRDN_CASE(2, 1, 1, 1)
    LVG_UPFIRDN_CASE(1, 1, 2, 1)
    return LVG_ERR_UNSUPPORTED;
}

template <class T>
int run(UpfirdnArgs& p, hipStream_t stream)
{
    const bool sep = (p.f2d == nullptr);
    // The tiled kernel wants W to be the fast axis of the input; otherwise gather.
    // (float64 always gathers: the tiles are staged in LDS as float32.)
    const bool wFast = (p.xs[3] == 1) || p.iw == 1;
    if (wFast && sizeof(T) <= 4)
    {
        int rc = dispatch_tiled<T>(p, sep, stream);
        if (rc != LVG_ERR_UNSUPPORTED) return rc;
    }
    return launch_gather<T>(p, stream);
}

} // namespace

extern "C" int lvg_upfirdn2d(const void* x, void* y, const float* f2d, const float* fx, const float* fy,
                             const int64_t xshape[4], const int64_t xstride[4],
                             const int64_t yshape[4], const int64_t ystride[4],
                             int fw, int fh, int64_t fstride_x, int64_t fstride_y,
                             int upx, int upy, int downx, int downy,
                             int padx0, int pady0, int flip, float gain, int dtype, void* stream)
{
    LVG_REQUIRE(x && y, "upfirdn2d: x and y must not be NULL");
    LVG_REQUIRE(dtype >= LVG_F32 && dtype <= LVG_F64, "upfirdn2d: unknown dtype %d", dtype);
    LVG_REQUIRE(upx >= 1 && upy >= 1, "upfirdn2d: upsampling factor must be at least 1");
    LVG_REQUIRE(downx >= 1 && downy >= 1, "upfirdn2d: downsampling factor must be at least 1");
    LVG_REQUIRE(fw >= 1 && fh >= 1, "upfirdn2d: f must be at least 1x1");
    LVG_REQUIRE(!(f2d && (fx || fy)), "upfirdn2d: pass either f2d or fx/fy, not both");
    for (int i = 0; i < 4; i++)
    {
        LVG_REQUIRE(xshape[i] >= 1 && xshape[i] <= 0x7fffffffLL, "upfirdn2d: x has zero size or is too large");
        LVG_REQUIRE(yshape[i] >= 1 && yshape[i] <= 0x7fffffffLL, "upfirdn2d: output must be at least 1x1");
    }
    LVG_REQUIRE(xshape[0] == yshape[0] && xshape[1] == yshape[1], "upfirdn2d: batch/channel mismatch between x and y");
    if (!f2d) { if (!fx) LVG_REQUIRE(fw == 1, "upfirdn2d: fx is NULL but fw != 1"); if (!fy) LVG_REQUIRE(fh == 1, "upfirdn2d: fy is NULL but fh != 1"); }

    UpfirdnArgs p;
    p.x = x; p.y = y; p.f2d = f2d; p.fx = fx; p.fy = fy;
    for (int i = 0; i < 4; i++) { p.xs[i] = xstride[i]; p.ys[i] = ystride[i]; }
    p.fsx = fstride_x; p.fsy = fstride_y;
    p.n = (int)xshape[0]; p.c = (int)xshape[1]; p.ih = (int)xshape[2]; p.iw = (int)xshape[3];
    p.oh = (int)yshape[2]; p.ow = (int)yshape[3];
    p.fw = fw; p.fh = fh;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy;
    p.padx0 = padx0; p.pady0 = pady0; p.flip = flip ? 1 : 0; p.gain = gain;
    p.tileW = p.tileH = p.tilesX = p.tilesY = p.inTW = p.inTH = p.midTH = 0;

    hipStream_t s = (hipStream_t)stream;
    switch (dtype)
    {
        case LVG_F32:  return run<float>(p, s);
        case LVG_F16:  return run<f16_t>(p, s);
        case LVG_BF16: return run<bf16_t>(p, s);
        default:       return run<double>(p, s);
    }
}
