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
#include <stdlib.h>

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
    int planesPerBlock;     // planes batched into one block (tiny planes)
    int laneWLog, laneWInLog; // log2 of lanes along x for the output / input tile (fast kernel)
    int rowChunks, chunkRows; // wave kernels: output rows are split into rowChunks chunks of chunkRows (grid y)
    int uniformPlanes;      // xs[0] == C * xs[1] (and same for y): plane index * stride addresses a plane
    int64_t totalPlanes;
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
// Tiled kernel. Block = 256 threads as 64 (x) x 4 (y). One block = one output tile of P
// consecutive planes (P > 1 only when a whole plane is one tile: the many tiny planes of the
// low-resolution layers are batched so that a block still moves a few thousand elements).
// LDS layout (floats): [taps X][taps Y][input tiles P x inTH x inTW][SEP: row-filtered P x inTH x tileW]
// MODE: 0 dense 2-D taps, 1 separable (rows then columns), 2 columns only (fw == 1, no resampling in x).
// Global loads are issued kLd at a time per thread before any LDS write (bytes in flight).

constexpr int kTX = 64, kTY = 4, kLd = 8;

template <class T, int UPX, int UPY, int DOWNX, int DOWNY, int MODE>
__global__ __launch_bounds__(kTX * kTY) void upfirdn2d_tiled_kernel(UpfirdnArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * kTX + tx;
    constexpr bool SEP = (MODE == 1);

    const int nTapX = (MODE == 0) ? p.fw * p.fh : p.fw;
    const int tapPadX = (nTapX + 3) & ~3;
    const int tapPadY = (MODE == 0) ? 0 : ((p.fh + 3) & ~3);
    float* sfx = smem;
    float* sfy = smem + tapPadX;
    float* sin = sfy + tapPadY;
    float* smid = sin + p.planesPerBlock * p.inTH * p.inTW; // SEP only

    // Flipped taps into LDS.
    if (MODE == 0)
    {
        for (int k = tid; k < p.fw * p.fh; k += kTX * kTY)
        {
            const int ky = k / p.fw, kx = k - ky * p.fw;
            const int sy = p.flip ? ky : p.fh - 1 - ky, sx = p.flip ? kx : p.fw - 1 - kx;
            sfx[k] = p.f2d[sy * p.fsy + sx * p.fsx];
        }
    }
    else
    {
        for (int k = tid; k < p.fw; k += kTX * kTY) sfx[k] = p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f;
        for (int k = tid; k < p.fh; k += kTX * kTY) sfy[k] = p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f;
    }

    // Which tile / planes.
    int b = blockIdx.x;
    const int tileX = b % p.tilesX; b /= p.tilesX;
    const int tileY = b % p.tilesY; b /= p.tilesY;
    const int64_t plane0 = (int64_t)b * p.planesPerBlock;
    const int nPlanes = (int)((p.totalPlanes - plane0) < p.planesPerBlock ? (p.totalPlanes - plane0) : p.planesPerBlock);

    const int outX0 = tileX * p.tileW, outY0 = tileY * p.tileH;
    const int midX0 = outX0 * DOWNX + UPX - 1 - p.padx0;
    const int midY0 = outY0 * DOWNY + UPY - 1 - p.pady0;
    const int inX0 = lvg_floor_div(midX0, UPX);
    const int inY0 = lvg_floor_div(midY0, UPY);

    // Plane base offsets: planes are (n, c) pairs; with xs[0] == C * xs[1] they are equidistant.
    const int64_t planeStrideX = p.xs[1], planeStrideY = p.ys[1];
    const int64_t xBase = p.uniformPlanes ? plane0 * planeStrideX : (plane0 / p.c) * p.xs[0] + (plane0 % p.c) * p.xs[1];
    const int64_t yBase = p.uniformPlanes ? plane0 * planeStrideY : (plane0 / p.c) * p.ys[0] + (plane0 % p.c) * p.ys[1];
    const T* xp = (const T*)p.x + xBase;
    T* yp = (T*)p.y + yBase;

    // Load the input tiles (zero outside the plane). Rows R = plane * inTH + r run over ty.
    const int totalRows = nPlanes * p.inTH;
    for (int q0 = 0; q0 < p.inTW; q0 += kTX)
    {
        const int q = q0 + tx;
        const int ix = inX0 + q;
        const bool colOk = (q < p.inTW) && ix >= 0 && ix < p.iw;
        for (int R0 = 0; R0 < totalRows; R0 += kTY * kLd)
        {
            float v[kLd];
            #pragma unroll
            for (int k = 0; k < kLd; k++)
            {
                const int R = R0 + ty + k * kTY;
                const int pl = R / p.inTH, r = R - pl * p.inTH;
                const int iy = inY0 + r;
                v[k] = 0.0f;
                if (colOk && R < totalRows && iy >= 0 && iy < p.ih)
                    v[k] = (float)to_acc(xp[(int64_t)pl * planeStrideX + (int64_t)iy * p.xs[2] + (int64_t)ix * p.xs[3]]);
            }
            #pragma unroll
            for (int k = 0; k < kLd; k++)
            {
                const int R = R0 + ty + k * kTY;
                if (q < p.inTW && R < totalRows) sin[R * p.inTW + q] = v[k];
            }
        }
    }
    __syncthreads();

    if (SEP)
    {
        // Pass 1: rows. smid[R][ox] = sum_k sin[R][relX + k] * sfx[tap0 + k*UPX]
        for (int ox = tx; ox < p.tileW; ox += kTX)
        {
            const int midX = midX0 + ox * DOWNX;
            const int inX = lvg_floor_div(midX, UPX);
            const int tap0 = (inX + 1) * UPX - midX - 1;
            for (int R = ty; R < totalRows; R += kTY)
            {
                const float* src = sin + R * p.inTW + (inX - inX0);
                float acc = 0.0f;
                for (int tap = tap0; tap < p.fw; tap += UPX) acc = fmaf(*src++, sfx[tap], acc);
                smid[R * p.tileW + ox] = acc;
            }
        }
        __syncthreads();
    }

    const float* colSrc = SEP ? smid : sin;
    const int colStride = SEP ? p.tileW : p.inTW;
    const int outRows = nPlanes * p.tileH;
    for (int ox = tx; ox < p.tileW; ox += kTX)
    {
        const int gx = outX0 + ox;
        if (gx >= p.ow) break;
        int relX = ox, tapX0 = 0;
        if (MODE != 1)
        {
            const int midX = midX0 + ox * DOWNX;
            const int inX = lvg_floor_div(midX, UPX);
            tapX0 = (inX + 1) * UPX - midX - 1;
            relX = inX - inX0;
        }
        for (int RO = ty; RO < outRows; RO += kTY)
        {
            const int pl = RO / p.tileH, oy = RO - pl * p.tileH;
            const int gy = outY0 + oy;
            if (gy >= p.oh) continue;
            const int midY = midY0 + oy * DOWNY;
            const int inY = lvg_floor_div(midY, UPY);
            const int tapY0 = (inY + 1) * UPY - midY - 1;
            const float* src = colSrc + (pl * p.inTH + (inY - inY0)) * colStride + relX;
            float acc = 0.0f;
            if (MODE == 0)
            {
                for (int tyy = tapY0; tyy < p.fh; tyy += UPY, src += colStride)
                {
                    const float* s2 = src;
                    const float* frow = sfx + tyy * p.fw;
                    for (int txx = tapX0; txx < p.fw; txx += UPX) acc = fmaf(*s2++, frow[txx], acc);
                }
            }
            else
            {
                for (int tap = tapY0; tap < p.fh; tap += UPY, src += colStride) acc = fmaf(*src, sfy[tap], acc);
                if (MODE == 2) acc *= sfx[0];
            }
            yp[(int64_t)pl * planeStrideY + (int64_t)gy * p.ys[2] + (int64_t)gx * p.ys[3]] = from_acc<T>(acc * p.gain);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host side.

constexpr int kMaxLdsBytes = 40 * 1024; // per block: >= 4 blocks per CU resident (160 KiB LDS)

inline int in_extent(int outExtent, int up, int down, int taps)
{
    return ((outExtent - 1) * down + taps - 1) / up + 1 + 1; // +1: floor_div phase slack
}

template <class T, int UPX, int UPY, int DOWNX, int DOWNY>
int launch_tiled(UpfirdnArgs& p, bool sep, hipStream_t stream)
{
    const int mode = !sep ? 0 : ((p.fw == 1 && UPX == 1 && DOWNX == 1) ? 2 : 1);
    const int64_t tapWords = (mode == 0) ? ((p.fw * p.fh + 3) & ~3) : (((p.fw + 3) & ~3) + ((p.fh + 3) & ~3));
    // Balanced split so the last tile of a row/column is not a sliver.
    int maxW = 128, maxH = 64;
    for (;;)
    {
        const int nx = (p.ow + maxW - 1) / maxW, ny = (p.oh + maxH - 1) / maxH;
        p.tileW = (p.ow + nx - 1) / nx; p.tileH = (p.oh + ny - 1) / ny;
        p.inTW = in_extent(p.tileW, UPX, DOWNX, p.fw);
        p.inTH = in_extent(p.tileH, UPY, DOWNY, p.fh);
        const int64_t words = tapWords + (int64_t)p.inTH * p.inTW + (mode == 1 ? (int64_t)p.inTH * p.tileW : 0);
        if (words * 4 <= kMaxLdsBytes) break;
        if (maxH > 8) maxH /= 2;
        else if (maxW > 32) maxW /= 2;
        else return LVG_ERR_UNSUPPORTED; // filter too large for a tile: gather kernel
    }
    p.tilesX = (p.ow + p.tileW - 1) / p.tileW;
    p.tilesY = (p.oh + p.tileH - 1) / p.tileH;
    p.totalPlanes = (int64_t)p.n * p.c;
    p.uniformPlanes = (p.n == 1) || (p.xs[0] == (int64_t)p.c * p.xs[1] && p.ys[0] == (int64_t)p.c * p.ys[1]);
    // Batch small planes: aim for >= 4096 outputs per block within the LDS budget.
    p.planesPerBlock = 1;
    if (p.tilesX == 1 && p.tilesY == 1 && p.uniformPlanes)
    {
        const int64_t perPlaneWords = (int64_t)p.inTH * p.inTW + (mode == 1 ? (int64_t)p.inTH * p.tileW : 0);
        int64_t want = 4096 / ((int64_t)p.oh * p.ow);
        const int64_t fit = (kMaxLdsBytes / 4 - tapWords) / perPlaneWords;
        if (want > fit) want = fit;
        if (want > 64) want = 64;
        if (want > p.totalPlanes) want = p.totalPlanes;
        if (want > 1) p.planesPerBlock = (int)want;
    }
    const int64_t planeBlocks = (p.totalPlanes + p.planesPerBlock - 1) / p.planesPerBlock;
    const int64_t blocks = (int64_t)p.tilesX * p.tilesY * planeBlocks;
    if (blocks > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(tapWords + (int64_t)p.planesPerBlock * ((int64_t)p.inTH * p.inTW + (mode == 1 ? (int64_t)p.inTH * p.tileW : 0))) * 4;
    const dim3 grid((unsigned)blocks), block(kTX, kTY);
    if (mode == 0)      hipLaunchKernelGGL((upfirdn2d_tiled_kernel<T, UPX, UPY, DOWNX, DOWNY, 0>), grid, block, lds, stream, p);
    else if (mode == 1) hipLaunchKernelGGL((upfirdn2d_tiled_kernel<T, UPX, UPY, DOWNX, DOWNY, 1>), grid, block, lds, stream, p);
    else                hipLaunchKernelGGL((upfirdn2d_tiled_kernel<T, UPX, UPY, DOWNX, DOWNY, 2>), grid, block, lds, stream, p);
    return lvg_check_launch("upfirdn2d_tiled_kernel");
}

// ---------------------------------------------------------------------------------------------
// Fast kernel for short filters (<= FPX x FPY taps, the 4-tap [1,3,3,1] family that carries the
// low-resolution generator/discriminator): same tiling as above, but tap counts are compile-time,
// taps live in registers, loops are fully unrolled, planes are an outer (wave-uniform) loop so no
// per-element integer division is left, and offsets inside a plane are 32-bit.
// MODE 0 (dense 2-D) is only instantiated without upsampling.

template <int UP, int FP>
__device__ __forceinline__ float tap_for_phase(const float (&f)[FP], int ph, int k)
{
    // f[(UP - 1 - ph) + k * UP] with a compile-time k and a runtime phase in [0, UP)
    float r = f[(UP - 1) + k * UP];
    #pragma unroll
    for (int q = 1; q < UP; q++) r = (ph == q) ? f[(UP - 1 - q) + k * UP] : r;
    return r;
}

template <class T, int UPX, int UPY, int DOWNX, int DOWNY, int MODE, int FPX, int FPY>
__global__ __launch_bounds__(kTX * kTY) void upfirdn2d_fast_kernel(UpfirdnArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NKX = FPX / UPX, NKY = FPY / UPY;
    constexpr bool SEP = (MODE == 1);
    // Thread mapping: LW lanes along x (power of two <= 64 covering the tile width), the rest along
    // rows -- narrow planes (W = 4..32) keep every lane of a wave busy.
    const int tid = threadIdx.y * kTX + threadIdx.x;
    const int lwLog = p.laneWLog, lwInLog = p.laneWInLog;
    const int LW = 1 << lwLog, LH = (kTX * kTY) >> lwLog;
    const int tx = tid & (LW - 1), ty = tid >> lwLog;
    const int LWI = 1 << lwInLog, LHI = (kTX * kTY) >> lwInLog;
    const int txi = tid & (LWI - 1), tyi = tid >> lwInLog;
    float* sin = smem;
    float* smid = sin + p.planesPerBlock * p.inTH * p.inTW;

    // Flipped, zero-padded taps in registers (wave-uniform loads).
    float fxr[FPX], fyr[FPY], f2[(MODE == 0) ? FPY * FPX : 1];
    if (MODE == 0)
    {
        #pragma unroll
        for (int ky = 0; ky < FPY; ky++)
            #pragma unroll
            for (int kx = 0; kx < FPX; kx++)
            {
                const int sy = p.flip ? ky : p.fh - 1 - ky, sx = p.flip ? kx : p.fw - 1 - kx;
                f2[ky * FPX + kx] = (ky < p.fh && kx < p.fw) ? p.f2d[sy * p.fsy + sx * p.fsx] : 0.0f;
            }
    }
    else
    {
        #pragma unroll
        for (int k = 0; k < FPX; k++) fxr[k] = (k < p.fw) ? (p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f) : 0.0f;
        #pragma unroll
        for (int k = 0; k < FPY; k++) fyr[k] = (k < p.fh) ? (p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f) : 0.0f;
    }

    int b = blockIdx.x;
    const int tileX = b % p.tilesX; b /= p.tilesX;
    const int tileY = b % p.tilesY; b /= p.tilesY;
    const int64_t plane0 = (int64_t)b * p.planesPerBlock;
    const int nPlanes = (int)((p.totalPlanes - plane0) < p.planesPerBlock ? (p.totalPlanes - plane0) : p.planesPerBlock);

    const int outX0 = tileX * p.tileW, outY0 = tileY * p.tileH;
    const int midX0 = outX0 * DOWNX + UPX - 1 - p.padx0;
    const int midY0 = outY0 * DOWNY + UPY - 1 - p.pady0;
    const int inX0 = lvg_floor_div(midX0, UPX);
    const int inY0 = lvg_floor_div(midY0, UPY);

    const int64_t xBase = p.uniformPlanes ? plane0 * p.xs[1] : (plane0 / p.c) * p.xs[0] + (plane0 % p.c) * p.xs[1];
    const int64_t yBase = p.uniformPlanes ? plane0 * p.ys[1] : (plane0 / p.c) * p.ys[0] + (plane0 % p.c) * p.ys[1];
    const int xs2 = (int)p.xs[2], xs3 = (int)p.xs[3], ys2 = (int)p.ys[2], ys3 = (int)p.ys[3];
    const int inTW = p.inTW, inTH = p.inTH, tileW = p.tileW, tileH = p.tileH;

    // ---- load (zero outside the plane), kLd loads in flight per thread ----
    for (int pl = 0; pl < nPlanes; pl++)
    {
        const T* xp = (const T*)p.x + xBase + (int64_t)pl * p.xs[1];
        float* dstPlane = sin + pl * inTH * inTW;
        for (int q0 = 0; q0 < inTW; q0 += LWI)
        {
            const int q = q0 + txi;
            const int ix = inX0 + q;
            const bool colOk = (q < inTW) && ix >= 0 && ix < p.iw;
            const int colOff = ix * xs3;
            for (int r0 = 0; r0 < inTH; r0 += LHI * kLd)
            {
                float v[kLd];
                #pragma unroll
                for (int k = 0; k < kLd; k++)
                {
                    const int r = r0 + tyi + k * LHI;
                    const int iy = inY0 + r;
                    v[k] = 0.0f;
                    if (colOk && r < inTH && iy >= 0 && iy < p.ih) v[k] = (float)to_acc(xp[iy * xs2 + colOff]);
                }
                #pragma unroll
                for (int k = 0; k < kLd; k++)
                {
                    const int r = r0 + tyi + k * LHI;
                    if (q < inTW && r < inTH) dstPlane[r * inTW + q] = v[k];
                }
            }
        }
    }
    __syncthreads();

    const int totalRows = nPlanes * inTH;
    if (SEP)
    {
        for (int ox = tx; ox < tileW; ox += LW)
        {
            const int midX = midX0 + ox * DOWNX;
            const int inX = lvg_floor_div(midX, UPX);
            const int ph = midX - inX * UPX;
            float tv[NKX];
            #pragma unroll
            for (int k = 0; k < NKX; k++) tv[k] = tap_for_phase<UPX, FPX>(fxr, ph, k);
            const float* src0 = sin + (inX - inX0);
            for (int R = ty; R < totalRows; R += LH)
            {
                const float* src = src0 + R * inTW;
                float acc = 0.0f;
                #pragma unroll
                for (int k = 0; k < NKX; k++) acc = fmaf(src[k], tv[k], acc);
                smid[R * tileW + ox] = acc;
            }
        }
        __syncthreads();
    }

    const float* colSrc = SEP ? smid : sin;
    const int colStride = SEP ? tileW : inTW;
    for (int ox = tx; ox < tileW; ox += LW)
    {
        const int gx = outX0 + ox;
        if (gx >= p.ow) break;
        int relX = ox;
        if (MODE != 1) relX = lvg_floor_div(midX0 + ox * DOWNX, UPX) - inX0;   // UPX == 1 for MODE 0 / 2
        const float scaleX = (MODE == 2) ? fxr[0] * p.gain : p.gain;
        for (int pl = 0; pl < nPlanes; pl++)
        {
            T* yp = (T*)p.y + yBase + (int64_t)pl * p.ys[1];
            const float* planeSrc = colSrc + pl * inTH * colStride + relX;
            for (int oy = ty; oy < tileH; oy += LH)
            {
                const int gy = outY0 + oy;
                if (gy >= p.oh) break;
                const int midY = midY0 + oy * DOWNY;
                const int inY = lvg_floor_div(midY, UPY);
                const int phY = midY - inY * UPY;
                const float* src = planeSrc + (inY - inY0) * colStride;
                float acc = 0.0f;
                if (MODE == 0)
                {
                    #pragma unroll
                    for (int ky = 0; ky < FPY; ky++)
                        #pragma unroll
                        for (int kx = 0; kx < FPX; kx++) acc = fmaf(src[ky * colStride + kx], f2[ky * FPX + kx], acc);
                }
                else
                {
                    #pragma unroll
                    for (int k = 0; k < NKY; k++) acc = fmaf(src[k * colStride], tap_for_phase<UPY, FPY>(fyr, phY, k), acc);
                }
                yp[gy * ys2 + gx * ys3] = from_acc<T>(acc * scaleX);
            }
        }
    }
}

template <class T, int UPX, int UPY, int DOWNX, int DOWNY, int FP>
int launch_fast(UpfirdnArgs& p, bool sep, hipStream_t stream)
{
    const int mode = !sep ? 0 : ((p.fw == 1 && UPX == 1 && DOWNX == 1) ? 2 : 1);
    if (mode == 0 && (UPX != 1 || UPY != 1)) return LVG_ERR_UNSUPPORTED;
    // Tile: the runtime geometry code is shared with the generic tiled kernel, but extents use the
    // PADDED tap counts because the unrolled loops read FP / UP samples per output.
    const int fwReal = p.fw, fhReal = p.fh;
    const int fwPad = (mode == 2) ? 1 : FP, fhPad = FP;
    int maxW = 128, maxH = 64;
    int64_t perPlaneWords = 0;
    for (;;)
    {
        const int nx = (p.ow + maxW - 1) / maxW, ny = (p.oh + maxH - 1) / maxH;
        p.tileW = (p.ow + nx - 1) / nx; p.tileH = (p.oh + ny - 1) / ny;
        p.inTW = in_extent(p.tileW, UPX, DOWNX, fwPad);
        p.inTH = in_extent(p.tileH, UPY, DOWNY, fhPad);
        perPlaneWords = (int64_t)p.inTH * p.inTW + (mode == 1 ? (int64_t)p.inTH * p.tileW : 0);
        if (perPlaneWords * 4 <= kMaxLdsBytes) break;
        if (maxH > 8) maxH /= 2;
        else if (maxW > 32) maxW /= 2;
        else return LVG_ERR_UNSUPPORTED;
    }
    (void)fwReal; (void)fhReal;
    p.tilesX = (p.ow + p.tileW - 1) / p.tileW;
    p.tilesY = (p.oh + p.tileH - 1) / p.tileH;
    p.totalPlanes = (int64_t)p.n * p.c;
    p.uniformPlanes = (p.n == 1) || (p.xs[0] == (int64_t)p.c * p.xs[1] && p.ys[0] == (int64_t)p.c * p.ys[1]);
    p.planesPerBlock = 1;
    if (p.tilesX == 1 && p.tilesY == 1 && p.uniformPlanes)
    {
        int64_t want = 4096 / ((int64_t)p.oh * p.ow);
        const int64_t fit = (kMaxLdsBytes / 4) / perPlaneWords;
        if (want > fit) want = fit;
        if (want > 64) want = 64;
        if (want > p.totalPlanes) want = p.totalPlanes;
        if (want > 1) p.planesPerBlock = (int)want;
    }
    // 32-bit offsets inside a plane
    const int64_t xExtent = (int64_t)(p.ih - 1) * (p.xs[2] < 0 ? -p.xs[2] : p.xs[2]) + (int64_t)(p.iw - 1) * (p.xs[3] < 0 ? -p.xs[3] : p.xs[3]);
    const int64_t yExtent = (int64_t)(p.oh - 1) * p.ys[2] + (int64_t)(p.ow - 1) * p.ys[3];
    if (xExtent >= 0x7fffffffLL || yExtent >= 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    const int64_t planeBlocks = (p.totalPlanes + p.planesPerBlock - 1) / p.planesPerBlock;
    const int64_t blocks = (int64_t)p.tilesX * p.tilesY * planeBlocks;
    if (blocks > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    p.laneWLog = 0; while ((1 << p.laneWLog) < p.tileW && p.laneWLog < 6) p.laneWLog++;
    p.laneWInLog = 0; while ((1 << p.laneWInLog) < p.inTW && p.laneWInLog < 6) p.laneWInLog++;
    const size_t lds = (size_t)((int64_t)p.planesPerBlock * perPlaneWords) * 4;
    const dim3 grid((unsigned)blocks), block(kTX, kTY);
    if (mode == 0)      { if constexpr (UPX == 1 && UPY == 1) hipLaunchKernelGGL((upfirdn2d_fast_kernel<T, UPX, UPY, DOWNX, DOWNY, 0, FP, FP>), grid, block, lds, stream, p); }
    else if (mode == 1) hipLaunchKernelGGL((upfirdn2d_fast_kernel<T, UPX, UPY, DOWNX, DOWNY, 1, FP, FP>), grid, block, lds, stream, p);
    else                { if constexpr (UPX == 1 && DOWNX == 1) hipLaunchKernelGGL((upfirdn2d_fast_kernel<T, UPX, UPY, DOWNX, DOWNY, 2, 1, FP>), grid, block, lds, stream, p); }
    return lvg_check_launch("upfirdn2d_fast_kernel");
}

// ---------------------------------------------------------------------------------------------
// Wave-streaming kernels for the workhorse of the low-resolution networks: separable <= 4-tap
// filters with x2 up- or down-sampling on NARROW planes (W <= 64: every [N, (C T), H, W] view of
// generator_lres / discriminator_lres). No LDS memory and no barriers:
//   * a row of a plane lives across the lanes of a wave (one lane per column; planes narrower than
//     64 are packed side by side, 64 / GW planes per wave);
//   * the row (x) taps are gathered from neighbouring lanes with wave shuffles;
//   * the column (y) taps run over a sliding window of row results kept in registers while the wave
//     streams down the plane, kRows input rows in flight per lane.
// Every input element is loaded exactly once and every output element stored once.

constexpr int kWaveThreads = 256;

struct WaveGeom { int gwLog; };

template <class T>
__device__ __forceinline__ float wave_load(const T* rowp, int x, bool ok) { return ok ? (float)to_acc(rowp[x]) : 0.0f; }

// Down: y[oy][ox] = gain * sum_ky sum_kx x[2 oy - pady0 + ky][2 ox - padx0 + kx] * ffy[ky] * ffx[kx]
template <class T>
__global__ __launch_bounds__(kWaveThreads) void upfirdn2d_wave_down2_kernel(UpfirdnArgs p)
{
    constexpr int kRows = 8;                       // input rows per batch -> 4 output rows
    const int gw = 1 << p.laneWLog;                // lanes per plane row (>= iw)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> p.laneWLog, x = lane & (gw - 1);
    const int planesPerWave = 64 >> p.laneWLog;
    const int64_t plane = ((int64_t)blockIdx.x * (kWaveThreads / 64) + wave) * planesPerWave + grp;
    const bool planeOk = plane < p.totalPlanes;
    const int64_t pc = planeOk ? plane : 0;
    const int nb = (int)(pc / p.c), ch = (int)(pc - (int64_t)nb * p.c);
    const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1];
    T* yp = (T*)p.y + (int64_t)nb * p.ys[0] + (int64_t)ch * p.ys[1];
    const int xs2 = (int)p.xs[2], ys2 = (int)p.ys[2];

    // flipped, zero-padded taps (uniform)
    float fx[4], fy[4];
    #pragma unroll
    for (int k = 0; k < 4; k++)
    {
        fx[k] = (k < p.fw) ? (p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f) : 0.0f;
        fy[k] = (k < p.fh) ? (p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f) : 0.0f;
    }
    // Source lanes of the 4 row taps of output column ox = x.
    const int laneBase = lane - x;
    int src[4]; bool srcOk[4];
    #pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int ix = 2 * x - p.padx0 + k;
        srcOk[k] = ix >= 0 && ix < p.iw;
        src[k] = laneBase + (srcOk[k] ? ix : 0);
    }
    const bool colLoad = planeOk && x < p.iw;
    const bool colOut = planeOk && x < p.ow;

    // Row-filtered value of input row iy for this lane's output column.
    auto hrow = [&](float v) -> float {
        float acc = 0.0f;
        #pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const float t = __shfl(v, src[k]);
            acc = fmaf(srcOk[k] ? t : 0.0f, fx[k], acc);
        }
        return acc;
    };

    // prologue: the two rows above the first batch
    const int oyBeg = blockIdx.y * p.chunkRows;
    const int oyEnd = (oyBeg + p.chunkRows < p.oh) ? oyBeg + p.chunkRows : p.oh;
    const int base0 = 2 * oyBeg - p.pady0;
    float c0, c1;
    {
        const int iy0 = base0, iy1 = base0 + 1;
        const float v0 = wave_load(xp + (int64_t)iy0 * xs2, x, colLoad && iy0 >= 0 && iy0 < p.ih);
        const float v1 = wave_load(xp + (int64_t)iy1 * xs2, x, colLoad && iy1 >= 0 && iy1 < p.ih);
        c0 = hrow(v0); c1 = hrow(v1);
    }
    for (int oy0 = oyBeg; oy0 < oyEnd; oy0 += kRows / 2)
    {
        const int rbase = 2 * oy0 - p.pady0 + 2;      // first new input row of this batch
        float v[kRows];
        #pragma unroll
        for (int r = 0; r < kRows; r++)
        {
            const int iy = rbase + r;
            v[r] = wave_load(xp + (int64_t)iy * xs2, x, colLoad && iy >= 0 && iy < p.ih);
        }
        float h[kRows + 2];
        h[0] = c0; h[1] = c1;
        #pragma unroll
        for (int r = 0; r < kRows; r++) h[r + 2] = hrow(v[r]);
        #pragma unroll
        for (int j = 0; j < kRows / 2; j++)
        {
            const int oy = oy0 + j;
            float acc = 0.0f;
            #pragma unroll
            for (int k = 0; k < 4; k++) acc = fmaf(h[2 * j + k], fy[k], acc);
            if (colOut && oy < oyEnd) yp[(int64_t)oy * ys2 + x] = from_acc<T>(acc * p.gain);
        }
        c0 = h[kRows]; c1 = h[kRows + 1];
    }
}

// Up: zero insertion x2. Row taps: out_h[ox] = in[i0] * ffx[t0] + in[i0 + 1] * ffx[t0 + 2],
// m = ox + 1 - padx0, i0 = floor(m / 2), t0 = 1 - (m mod 2); columns alike.
template <class T>
__global__ __launch_bounds__(kWaveThreads) void upfirdn2d_wave_up2_kernel(UpfirdnArgs p)
{
    constexpr int kRows = 4;                       // input rows per batch -> 8 output rows
    const int gw = 1 << p.laneWLog;                // lanes per plane row (>= ow >= iw)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> p.laneWLog, x = lane & (gw - 1);
    const int planesPerWave = 64 >> p.laneWLog;
    const int64_t plane = ((int64_t)blockIdx.x * (kWaveThreads / 64) + wave) * planesPerWave + grp;
    const bool planeOk = plane < p.totalPlanes;
    const int64_t pc = planeOk ? plane : 0;
    const int nb = (int)(pc / p.c), ch = (int)(pc - (int64_t)nb * p.c);
    const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1];
    T* yp = (T*)p.y + (int64_t)nb * p.ys[0] + (int64_t)ch * p.ys[1];
    const int xs2 = (int)p.xs[2], ys2 = (int)p.ys[2];

    float fx[4], fy[4];
    #pragma unroll
    for (int k = 0; k < 4; k++)
    {
        fx[k] = (k < p.fw) ? (p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f) : 0.0f;
        fy[k] = (k < p.fh) ? (p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f) : 0.0f;
    }
    const int laneBase = lane - x;
    const int mX = x + 1 - p.padx0;
    const int i0 = lvg_floor_div(mX, 2);
    const int phX = mX - 2 * i0;                    // 0 -> taps 1,3 ; 1 -> taps 0,2
    const float tA = phX ? fx[0] : fx[1], tB = phX ? fx[2] : fx[3];
    const bool okA = i0 >= 0 && i0 < p.iw, okB = (i0 + 1) >= 0 && (i0 + 1) < p.iw;
    const int srcA = laneBase + (okA ? i0 : 0), srcB = laneBase + (okB ? i0 + 1 : 0);
    const bool colLoad = planeOk && x < p.iw;
    const bool colOut = planeOk && x < p.ow;
    const float g = p.gain;

    auto hrow = [&](float v) -> float {
        const float a = __shfl(v, srcA), b = __shfl(v, srcB);
        return fmaf(okA ? a : 0.0f, tA, (okB ? b : 0.0f) * tB);
    };

    // Output rows produced by the input-row pair (j, j + 1): oyA = 2 j - 1 + pady0 (taps 1, 3) and oyA + 1 (taps 0, 2).
    const int oyBeg = blockIdx.y * p.chunkRows;
    const int oyEnd = (oyBeg + p.chunkRows < p.oh) ? oyBeg + p.chunkRows : p.oh;
    const int jMin = lvg_floor_div(oyBeg + 1 - p.pady0, 2);    // j of the chunk's first output row
    const int jMax = lvg_floor_div(oyEnd - p.pady0, 2);        // j of its last output row
    float hPrev;
    {
        const int iy = jMin;
        hPrev = hrow(wave_load(xp + (int64_t)iy * xs2, x, colLoad && iy >= 0 && iy < p.ih));
    }
    for (int j0 = jMin; j0 <= jMax; j0 += kRows)
    {
        float v[kRows];
        #pragma unroll
        for (int r = 0; r < kRows; r++)
        {
            const int iy = j0 + 1 + r;
            v[r] = wave_load(xp + (int64_t)iy * xs2, x, colLoad && iy >= 0 && iy < p.ih);
        }
        #pragma unroll
        for (int r = 0; r < kRows; r++)
        {
            const float hCur = hrow(v[r]);
            const int j = j0 + r;
            const int oyA = 2 * j - 1 + p.pady0;
            const float oa = fmaf(hPrev, fy[1], hCur * fy[3]) * g;
            const float ob = fmaf(hPrev, fy[0], hCur * fy[2]) * g;
            if (colOut && j <= jMax)
            {
                if (oyA >= oyBeg && oyA < oyEnd) yp[(int64_t)oyA * ys2 + x] = from_acc<T>(oa);
                if (oyA + 1 >= oyBeg && oyA + 1 < oyEnd) yp[(int64_t)(oyA + 1) * ys2 + x] = from_acc<T>(ob);
            }
            hPrev = hCur;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Column-only streaming kernel: <= 4 taps, x2 up or down along H, nothing along W (fw == 1) -- the
// time-axis resamplers on [N, C, T, (H W)] views. A lane owns VB bytes (one vector) of a row and
// walks down a chunk of rows with the same sliding window as the wave kernels; lanes of a wave
// cover consecutive vectors, so every load/store is a full coalesced line.

template <class T, int VB> struct VecIO
{
    static constexpr int V = VB / (int)sizeof(T);
    struct alignas(VB) Raw { T v[V]; };
    static __device__ __forceinline__ void load(const T* p, bool ok, float (&o)[V])
    {
        if (ok) { const Raw r = *reinterpret_cast<const Raw*>(p);
                  #pragma unroll
                  for (int i = 0; i < V; i++) o[i] = (float)to_acc(r.v[i]); }
        else    {
                  #pragma unroll
                  for (int i = 0; i < V; i++) o[i] = 0.0f; }
    }
    static __device__ __forceinline__ void store(T* p, const float (&o)[V], float g)
    {
        Raw r;
        #pragma unroll
        for (int i = 0; i < V; i++) r.v[i] = from_acc<T>(o[i] * g);
        *reinterpret_cast<Raw*>(p) = r;
    }
};

constexpr int kColChunk = 8;    // output rows per thread: short chains (one or two load batches), many threads

template <class T, int VB, bool UP2>
__global__ __launch_bounds__(256) void upfirdn2d_col_kernel(UpfirdnArgs p)
{
    typedef VecIO<T, VB> IO;
    constexpr int V = IO::V;
    const int vecsPerRow = p.iw / V;                        // ow == iw
    const int chunks = (p.oh + kColChunk - 1) / kColChunk;
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int xv = (int)(t % vecsPerRow); t /= vecsPerRow;
    const int chunk = (int)(t % chunks); t /= chunks;
    if (t >= p.totalPlanes) return;
    const int nb = (int)(t / p.c), ch = (int)(t - (int64_t)nb * p.c);
    const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1] + xv * V;
    T* yp = (T*)p.y + (int64_t)nb * p.ys[0] + (int64_t)ch * p.ys[1] + xv * V;
    const int xs2 = (int)p.xs[2], ys2 = (int)p.ys[2];
    float fy[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) fy[k] = (k < p.fh) ? (p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f) : 0.0f;
    const float g = p.gain * (p.fx ? p.fx[0] : 1.0f);
    const int oyBeg = chunk * kColChunk;
    const int oyEnd = (oyBeg + kColChunk < p.oh) ? oyBeg + kColChunk : p.oh;

    if (!UP2)
    {
        // out[oy] = sum_k in[2 oy - pady0 + k] * fy[k]
        constexpr int kRows = 8;
        float c0[V], c1[V];
        {
            const int iy0 = 2 * oyBeg - p.pady0, iy1 = iy0 + 1;
            IO::load(xp + (int64_t)iy0 * xs2, iy0 >= 0 && iy0 < p.ih, c0);
            IO::load(xp + (int64_t)iy1 * xs2, iy1 >= 0 && iy1 < p.ih, c1);
        }
        for (int oy0 = oyBeg; oy0 < oyEnd; oy0 += kRows / 2)
        {
            const int rbase = 2 * oy0 - p.pady0 + 2;
            float h[kRows + 2][V];
            #pragma unroll
            for (int i = 0; i < V; i++) { h[0][i] = c0[i]; h[1][i] = c1[i]; }
            #pragma unroll
            for (int r = 0; r < kRows; r++)
            {
                const int iy = rbase + r;
                IO::load(xp + (int64_t)iy * xs2, iy >= 0 && iy < p.ih && (oy0 + r / 2) < oyEnd + 1, h[r + 2]);
            }
            #pragma unroll
            for (int j = 0; j < kRows / 2; j++)
            {
                const int oy = oy0 + j;
                if (oy < oyEnd)
                {
                    float acc[V];
                    #pragma unroll
                    for (int i = 0; i < V; i++)
                    {
                        float a = 0.0f;
                        #pragma unroll
                        for (int k = 0; k < 4; k++) a = fmaf(h[2 * j + k][i], fy[k], a);
                        acc[i] = a;
                    }
                    IO::store(yp + (int64_t)oy * ys2, acc, g);
                }
            }
            #pragma unroll
            for (int i = 0; i < V; i++) { c0[i] = h[kRows][i]; c1[i] = h[kRows + 1][i]; }
        }
    }
    else
    {
        // pair (j, j+1) -> rows oyA = 2 j - 1 + pady0 (taps 1, 3) and oyA + 1 (taps 0, 2)
        constexpr int kRows = 4;
        const int jMin = lvg_floor_div(oyBeg + 1 - p.pady0, 2);
        const int jMax = lvg_floor_div(oyEnd - p.pady0, 2);
        float hPrev[V];
        IO::load(xp + (int64_t)jMin * xs2, jMin >= 0 && jMin < p.ih, hPrev);
        for (int j0 = jMin; j0 <= jMax; j0 += kRows)
        {
            float v[kRows][V];
            #pragma unroll
            for (int r = 0; r < kRows; r++)
            {
                const int iy = j0 + 1 + r;
                IO::load(xp + (int64_t)iy * xs2, iy >= 0 && iy < p.ih && (j0 + r) <= jMax, v[r]);
            }
            #pragma unroll
            for (int r = 0; r < kRows; r++)
            {
                const int j = j0 + r;
                const int oyA = 2 * j - 1 + p.pady0;
                if (j <= jMax)
                {
                    float oa[V], ob[V];
                    #pragma unroll
                    for (int i = 0; i < V; i++)
                    {
                        oa[i] = fmaf(hPrev[i], fy[1], v[r][i] * fy[3]);
                        ob[i] = fmaf(hPrev[i], fy[0], v[r][i] * fy[2]);
                    }
                    if (oyA >= oyBeg && oyA < oyEnd) IO::store(yp + (int64_t)oyA * ys2, oa, g);
                    if (oyA + 1 >= oyBeg && oyA + 1 < oyEnd) IO::store(yp + (int64_t)(oyA + 1) * ys2, ob, g);
                }
                #pragma unroll
                for (int i = 0; i < V; i++) hPrev[i] = v[r][i];
            }
        }
    }
}

template <class T, int VB>
int launch_col_vb(UpfirdnArgs& p, bool up2, hipStream_t stream)
{
    constexpr int V = VB / (int)sizeof(T);
    const int64_t chunks = (p.oh + kColChunk - 1) / kColChunk;
    const int64_t threads = p.totalPlanes * chunks * (p.iw / V);
    const int64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    if (up2) hipLaunchKernelGGL((upfirdn2d_col_kernel<T, VB, true>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else     hipLaunchKernelGGL((upfirdn2d_col_kernel<T, VB, false>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return lvg_check_launch("upfirdn2d_col_kernel");
}

template <class T>
int launch_col(UpfirdnArgs& p, hipStream_t stream)
{
    if (p.f2d || p.fw != 1 || p.fh > 4 || p.upx != 1 || p.downx != 1 || p.iw != p.ow) return LVG_ERR_UNSUPPORTED;
    const bool up2 = p.upy == 2 && p.downy == 1, down2 = p.upy == 1 && p.downy == 2;
    if (!up2 && !down2) return LVG_ERR_UNSUPPORTED;
    if (p.xs[3] != 1 || p.ys[3] != 1) return LVG_ERR_UNSUPPORTED;
    if ((int64_t)(p.ih + 16) * (p.xs[2] < 0 ? -p.xs[2] : p.xs[2]) >= 0x7fffffffLL || (int64_t)(p.oh + 16) * p.ys[2] >= 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    p.totalPlanes = (int64_t)p.n * p.c;
    // widest vector such that every row of every plane starts on a vector boundary
    auto fits = [&](int vb) {
        const int v = vb / (int)sizeof(T);
        if (v < 1 || p.iw % v) return false;
        if (((uintptr_t)p.x % vb) || ((uintptr_t)p.y % vb)) return false;
        for (int i = 0; i < 3; i++) if ((p.xs[i] % v) || (p.ys[i] % v)) return false;
        return true;
    };
    if (fits(16)) return launch_col_vb<T, 16>(p, up2, stream);
    if (fits(8))  return launch_col_vb<T, 8>(p, up2, stream);
    if (fits(4))  return launch_col_vb<T, 4>(p, up2, stream);
    return LVG_ERR_UNSUPPORTED;
}

// 16-bit element types: one lane owns a PAIR of adjacent columns (one dword per row), so a wave row
// covers 128 input columns' worth of planes with 4-byte loads/stores and every lane produces output.

template <class T>
__device__ __forceinline__ float half_of(uint32_t raw, int shift, uint32_t okMask)
{
    // shift = 0 / 16 selects the low / high element; okMask = 0 turns the value into +0.0 (branch-free)
    T t; t.bits = (uint16_t)(((raw & okMask) >> shift) & 0xffffu);
    return (float)to_acc(t);
}
template <class T>
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    return (uint32_t)from_acc<T>(a).bits | ((uint32_t)from_acc<T>(b).bits << 16);
}

template <class T>
__global__ __launch_bounds__(kWaveThreads) void upfirdn2d_wave_down2_pair_kernel(UpfirdnArgs p)
{
    constexpr int kRows = 8;
    const int gw = 1 << p.laneWLog;                // lanes per plane row: >= iw / 2 and >= ow
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> p.laneWLog, x = lane & (gw - 1);
    const int planesPerWave = 64 >> p.laneWLog;
    const int64_t plane = ((int64_t)blockIdx.x * (kWaveThreads / 64) + wave) * planesPerWave + grp;
    const bool planeOk = plane < p.totalPlanes;
    const int64_t pc = planeOk ? plane : 0;
    const int nb = (int)(pc / p.c), ch = (int)(pc - (int64_t)nb * p.c);
    const uint32_t* xp = (const uint32_t*)((const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1]);
    T* yp = (T*)p.y + (int64_t)nb * p.ys[0] + (int64_t)ch * p.ys[1];
    const int xs2w = (int)(p.xs[2] >> 1), ys2 = (int)p.ys[2];     // input row stride in dwords
    const int iwPairs = p.iw >> 1;

    float fx[4], fy[4];
    #pragma unroll
    for (int k = 0; k < 4; k++)
    {
        fx[k] = (k < p.fw) ? (p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f) : 0.0f;
        fy[k] = (k < p.fh) ? (p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f) : 0.0f;
    }
    const int laneBase = lane - x;
    int src[4], sh[4]; uint32_t okm[4];
    #pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int e = 2 * x - p.padx0 + k;                    // input column of tap k for output column x
        const bool ok = e >= 0 && e < p.iw;
        const int ec = ok ? e : 0;
        okm[k] = ok ? 0xffffffffu : 0u;
        src[k] = laneBase + (ec >> 1);
        sh[k] = (ec & 1) * 16;
    }
    const bool colLoad = planeOk && x < iwPairs;
    const bool colOut = planeOk && x < p.ow;

    auto hrow = [&](uint32_t raw) -> float {
        float acc = 0.0f;
        #pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const uint32_t t = (uint32_t)__shfl((int)raw, src[k]);
            acc = fmaf(half_of<T>(t, sh[k], okm[k]), fx[k], acc);
        }
        return acc;
    };
    auto ld = [&](int iy) -> uint32_t { return (colLoad && iy >= 0 && iy < p.ih) ? xp[(int64_t)iy * xs2w + x] : 0u; };

    const int oyBeg = blockIdx.y * p.chunkRows;
    const int oyEnd = (oyBeg + p.chunkRows < p.oh) ? oyBeg + p.chunkRows : p.oh;
    float c0 = hrow(ld(2 * oyBeg - p.pady0)), c1 = hrow(ld(2 * oyBeg - p.pady0 + 1));
    for (int oy0 = oyBeg; oy0 < oyEnd; oy0 += kRows / 2)
    {
        const int rbase = 2 * oy0 - p.pady0 + 2;
        uint32_t v[kRows];
        #pragma unroll
        for (int r = 0; r < kRows; r++) v[r] = ld(rbase + r);
        float h[kRows + 2];
        h[0] = c0; h[1] = c1;
        #pragma unroll
        for (int r = 0; r < kRows; r++) h[r + 2] = hrow(v[r]);
        #pragma unroll
        for (int j = 0; j < kRows / 2; j++)
        {
            const int oy = oy0 + j;
            float acc = 0.0f;
            #pragma unroll
            for (int k = 0; k < 4; k++) acc = fmaf(h[2 * j + k], fy[k], acc);
            if (colOut && oy < oyEnd) yp[(int64_t)oy * ys2 + x] = from_acc<T>(acc * p.gain);
        }
        c0 = h[kRows]; c1 = h[kRows + 1];
    }
}

// Returns LVG_ERR_UNSUPPORTED when the shape is not one the wave kernels take.
template <class T>
int launch_wave(UpfirdnArgs& p, hipStream_t stream)
{
    if (p.f2d || p.fw > 4 || p.fh > 4) return LVG_ERR_UNSUPPORTED;
    if (p.xs[3] != 1 || p.ys[3] != 1) return LVG_ERR_UNSUPPORTED;
    const bool up2 = p.upx == 2 && p.upy == 2 && p.downx == 1 && p.downy == 1;
    const bool down2 = p.upx == 1 && p.upy == 1 && p.downx == 2 && p.downy == 2;
    if (!up2 && !down2) return LVG_ERR_UNSUPPORTED;
    const int wmax = (p.iw > p.ow) ? p.iw : p.ow;
    if (wmax > 64) return LVG_ERR_UNSUPPORTED;
    // rows must be addressable with 32-bit offsets inside a plane
    if ((int64_t)(p.ih + 16) * (p.xs[2] < 0 ? -p.xs[2] : p.xs[2]) >= 0x7fffffffLL || (int64_t)(p.oh + 16) * p.ys[2] >= 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    p.totalPlanes = (int64_t)p.n * p.c;
    // Row chunking (grid y) is wired in but measured slower than one wave per plane on MI355X (the halo
    // rows cost more than the shorter dependent chains save), so a plane is one chunk.
    p.chunkRows = 1 << 30;
    p.rowChunks = 1;
    if constexpr (sizeof(T) == 2)
    {
        // paired-lane variant: rows must start on dword boundaries
        const bool even = !(p.iw & 1) && !(p.xs[0] & 1) && !(p.xs[1] & 1) && !(p.xs[2] & 1) && !((uintptr_t)p.x & 3);
        // (a paired-lane up2 variant measured slower than the plain one: fixed ~14 us overhead; not used)
        if (even && down2)
        {
            const int need = down2 ? ((p.iw / 2 > p.ow) ? p.iw / 2 : p.ow) : ((p.ow / 2 > p.iw / 2) ? p.ow / 2 : p.iw / 2);
            p.laneWLog = 0; while ((1 << p.laneWLog) < need) p.laneWLog++;
            const int ppb = (kWaveThreads / 64) * (64 >> p.laneWLog);
            const int64_t nblk = (p.totalPlanes + ppb - 1) / ppb;
            if (nblk > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
            hipLaunchKernelGGL((upfirdn2d_wave_down2_pair_kernel<T>), dim3((unsigned)nblk, (unsigned)p.rowChunks), dim3(kWaveThreads), 0, stream, p);
            return lvg_check_launch("upfirdn2d_wave_pair_kernel");
        }
    }
    p.laneWLog = 0; while ((1 << p.laneWLog) < wmax) p.laneWLog++;
    const int planesPerBlock = (kWaveThreads / 64) * (64 >> p.laneWLog);
    const int64_t blocks = (p.totalPlanes + planesPerBlock - 1) / planesPerBlock;
    if (blocks > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    if (up2) hipLaunchKernelGGL((upfirdn2d_wave_up2_kernel<T>), dim3((unsigned)blocks, (unsigned)p.rowChunks), dim3(kWaveThreads), 0, stream, p);
    else     hipLaunchKernelGGL((upfirdn2d_wave_down2_kernel<T>), dim3((unsigned)blocks, (unsigned)p.rowChunks), dim3(kWaveThreads), 0, stream, p);
    return lvg_check_launch("upfirdn2d_wave_kernel");
}

// ---------------------------------------------------------------------------------------------
// Channels-last (NHWC) kernel: separable <= 4 taps, up/down in {1, 2}. The frames-layout networks
// keep activations channels-last (MIOpen's MFMA implicit-GEMM convs are NHWC-native; with NCHW tensors
// every conv is wrapped in two layout-transpose kernels). A lane owns one VB-byte vector of channels
// of one output pixel; lanes of a wave cover consecutive channel vectors, then consecutive pixels, so
// every load and store is a fully coalesced 16-byte access. Neighbouring pixels re-read the same input
// vectors from L1/L2; HBM sees each input once.

template <class T, int UPX, int UPY, int DOWNX, int DOWNY, int VB>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_kernel(UpfirdnArgs p)
{
    typedef VecIO<T, VB> IO;
    constexpr int V = IO::V;
    constexpr int FP = 4;
    constexpr int NKX = FP / UPX, NKY = FP / UPY;
    const int cvecs = p.c / V;
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)p.n * p.oh * p.ow * cvecs;
    if (t >= total) return;
    const int cv = (int)(t % cvecs); t /= cvecs;
    const int ox = (int)(t % p.ow); t /= p.ow;
    const int oy = (int)(t % p.oh);
    const int nb = (int)(t / p.oh);

    float fx[FP], fy[FP];
    #pragma unroll
    for (int k = 0; k < FP; k++)
    {
        fx[k] = (k < p.fw) ? (p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f) : 0.0f;
        fy[k] = (k < p.fh) ? (p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f) : 0.0f;
    }
    const int midX = ox * DOWNX + UPX - 1 - p.padx0, midY = oy * DOWNY + UPY - 1 - p.pady0;
    const int inX = lvg_floor_div(midX, UPX), inY = lvg_floor_div(midY, UPY);
    const int phX = midX - inX * UPX, phY = midY - inY * UPY;

    const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)cv * V;     // channel stride is 1
    float acc[V];
    #pragma unroll
    for (int i = 0; i < V; i++) acc[i] = 0.0f;
    #pragma unroll
    for (int ky = 0; ky < NKY; ky++)
    {
        const int iy = inY + ky;
        const float wy = tap_for_phase<UPY, FP>(fy, phY, ky);
        const bool rowOk = iy >= 0 && iy < p.ih;
        float row[V];
        #pragma unroll
        for (int i = 0; i < V; i++) row[i] = 0.0f;
        #pragma unroll
        for (int kx = 0; kx < NKX; kx++)
        {
            const int ix = inX + kx;
            const float wx = tap_for_phase<UPX, FP>(fx, phX, kx);
            float v[V];
            IO::load(xp + (int64_t)iy * p.xs[2] + (int64_t)ix * p.xs[3], rowOk && ix >= 0 && ix < p.iw, v);
            #pragma unroll
            for (int i = 0; i < V; i++) row[i] = fmaf(v[i], wx, row[i]);
        }
        #pragma unroll
        for (int i = 0; i < V; i++) acc[i] = fmaf(row[i], wy, acc[i]);
    }
    T* yp = (T*)p.y + (int64_t)nb * p.ys[0] + (int64_t)oy * p.ys[2] + (int64_t)ox * p.ys[3] + (int64_t)cv * V;
    IO::store(yp, acc, p.gain);
}

// Streaming NHWC x2 kernels: a lane owns one channel vector of one OUTPUT COLUMN and walks down the rows
// of a frame with the sliding window of the wave kernels; the row (x) taps are plain neighbouring-pixel
// loads (+-C elements: coalesced, L1-resident). Per input row a lane issues 4 (down) / 2 (up) vector loads
// instead of the 16 / 4 per output pixel of the gather form above.

template <class T, int VB, bool UP2>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_stream_kernel(UpfirdnArgs p)
{
    typedef VecIO<T, VB> IO;
    constexpr int V = IO::V;
    const int cvecs = p.c / V;
    const int chunks = p.rowChunks;
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)p.n * chunks * p.ow * cvecs;
    if (t >= total) return;
    const int cv = (int)(t % cvecs); t /= cvecs;
    const int ox = (int)(t % p.ow); t /= p.ow;
    const int chunk = (int)(t % chunks);
    const int nb = (int)(t / chunks);
    const int oyBeg = chunk * p.chunkRows;
    const int oyEnd = (oyBeg + p.chunkRows < p.oh) ? oyBeg + p.chunkRows : p.oh;

    float fx[4], fy[4];
    #pragma unroll
    for (int k = 0; k < 4; k++)
    {
        fx[k] = (k < p.fw) ? (p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f) : 0.0f;
        fy[k] = (k < p.fh) ? (p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f) : 0.0f;
    }
    const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)cv * V;
    T* yp = (T*)p.y + (int64_t)nb * p.ys[0] + (int64_t)ox * p.ys[3] + (int64_t)cv * V;
    const int64_t xs2 = p.xs[2], xs3 = p.xs[3], ys2 = p.ys[2];
    const float g = p.gain;

    if (!UP2)
    {
        // row-filtered value of input row iy at this output column: sum_k in[iy][2 ox - padx0 + k] * fx[k]
        const int ix0 = 2 * ox - p.padx0;
        auto hrow = [&](int iy, float (&h)[V]) {
            #pragma unroll
            for (int i = 0; i < V; i++) h[i] = 0.0f;
            const bool rowOk = iy >= 0 && iy < p.ih;
            #pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int ix = ix0 + k;
                float v[V];
                IO::load(xp + (int64_t)iy * xs2 + (int64_t)ix * xs3, rowOk && ix >= 0 && ix < p.iw, v);
                #pragma unroll
                for (int i = 0; i < V; i++) h[i] = fmaf(v[i], fx[k], h[i]);
            }
        };
        constexpr int kRows = (V >= 8) ? 2 : 4;      // input rows per batch (register budget: (kRows+2) x V floats)
        float w[kRows + 2][V];
        hrow(2 * oyBeg - p.pady0, w[0]);
        hrow(2 * oyBeg - p.pady0 + 1, w[1]);
        for (int oy0 = oyBeg; oy0 < oyEnd; oy0 += kRows / 2)
        {
            const int rbase = 2 * oy0 - p.pady0 + 2;
            #pragma unroll
            for (int r = 0; r < kRows; r++) hrow(rbase + r, w[r + 2]);
            #pragma unroll
            for (int j = 0; j < kRows / 2; j++)
            {
                const int oy = oy0 + j;
                if (oy < oyEnd)
                {
                    float acc[V];
                    #pragma unroll
                    for (int i = 0; i < V; i++)
                    {
                        float a = 0.0f;
                        #pragma unroll
                        for (int k = 0; k < 4; k++) a = fmaf(w[2 * j + k][i], fy[k], a);
                        acc[i] = a;
                    }
                    IO::store(yp + (int64_t)oy * ys2, acc, g);
                }
            }
            #pragma unroll
            for (int i = 0; i < V; i++) { w[0][i] = w[kRows][i]; w[1][i] = w[kRows + 1][i]; }
        }
    }
    else
    {
        const int mX = ox + 1 - p.padx0;
        const int i0 = lvg_floor_div(mX, 2);
        const int phX = mX - 2 * i0;
        const float tA = phX ? fx[0] : fx[1], tB = phX ? fx[2] : fx[3];
        auto hrow = [&](int iy, float (&h)[V]) {
            const bool rowOk = iy >= 0 && iy < p.ih;
            float a[V], b[V];
            IO::load(xp + (int64_t)iy * xs2 + (int64_t)i0 * xs3, rowOk && i0 >= 0 && i0 < p.iw, a);
            IO::load(xp + (int64_t)iy * xs2 + (int64_t)(i0 + 1) * xs3, rowOk && i0 + 1 >= 0 && i0 + 1 < p.iw, b);
            #pragma unroll
            for (int i = 0; i < V; i++) h[i] = fmaf(a[i], tA, b[i] * tB);
        };
        const int jMin = lvg_floor_div(oyBeg + 1 - p.pady0, 2);
        const int jMax = lvg_floor_div(oyEnd - p.pady0, 2);
        float hPrev[V], hCur[V];
        hrow(jMin, hPrev);
        for (int j = jMin; j <= jMax; j++)
        {
            hrow(j + 1, hCur);
            const int oyA = 2 * j - 1 + p.pady0;
            float oa[V], ob[V];
            #pragma unroll
            for (int i = 0; i < V; i++)
            {
                oa[i] = fmaf(hPrev[i], fy[1], hCur[i] * fy[3]);
                ob[i] = fmaf(hPrev[i], fy[0], hCur[i] * fy[2]);
                hPrev[i] = hCur[i];
            }
            if (oyA >= oyBeg && oyA < oyEnd) IO::store(yp + (int64_t)oyA * ys2, oa, g);
            if (oyA + 1 >= oyBeg && oyA + 1 < oyEnd) IO::store(yp + (int64_t)(oyA + 1) * ys2, ob, g);
        }
    }
}

// NHWC x2 DOWN-sampler through LDS (16-byte vectors): one workgroup walks a chunk of output rows of one frame. Every input row is
// fetched from global memory ONCE, with fully coalesced 16-byte loads two rows ahead of its use, and staged in LDS; a thread owns
// NO output vectors (column, channel vector) of the row, takes their four x taps out of LDS and keeps the y window in registers.
// (The streaming form above issues 4 global loads per input row and output vector -- every input vector twice, from L1 / L2 --
// and needs 134 VGPRs: 3 waves per SIMD.)
template <class T, int NV, int NO>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_down2_lds_kernel(UpfirdnArgs p)
{
    typedef VecIO<T, 16> IO;
    typedef uint4 Raw;                                                   // a 16-byte vector as stored
    constexpr int V = IO::V;
    extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
    Raw* lds = reinterpret_cast<Raw*>(smemRaw);                         // [2][rowVecs]
    const int tid = threadIdx.x;
    const int cvecs = p.c / V;
    const int rowVecs = p.iw * cvecs;
    const int chunk = blockIdx.x % p.rowChunks, nb = blockIdx.x / p.rowChunks;
    const int oyBeg = chunk * p.chunkRows;
    const int oyEnd = (oyBeg + p.chunkRows < p.oh) ? oyBeg + p.chunkRows : p.oh;

    float fx[4], fy[4];
    #pragma unroll
    for (int k = 0; k < 4; k++)
    {
        fx[k] = (k < p.fw) ? (p.fx ? p.fx[p.flip ? k : p.fw - 1 - k] : 1.0f) : 0.0f;
        fy[k] = (k < p.fh) ? (p.fy ? p.fy[p.flip ? k : p.fh - 1 - k] : 1.0f) : 0.0f;
    }
    const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0];
    T* yp = (T*)p.y + (int64_t)nb * p.ys[0];
    const int xs2 = (int)p.xs[2], ys2 = (int)p.ys[2];

    // this thread's output vectors: LDS index of the first x tap, which taps are inside the row
    int tap0[NO]; unsigned tapOk[NO]; int outOff[NO];
    #pragma unroll
    for (int j = 0; j < NO; j++)
    {
        const int ov = tid + 256 * j;
        const int ox = ov / cvecs, cv = ov - ox * cvecs;
        const int ix0 = 2 * ox - p.padx0;
        tap0[j] = ix0 * cvecs + cv;
        tapOk[j] = 0;
        #pragma unroll
        for (int k = 0; k < 4; k++) if (ix0 + k >= 0 && ix0 + k < p.iw) tapOk[j] |= 1u << k;
        outOff[j] = ov * V;
    }
    Raw pre[2][NV];
    // No execution-mask branches in the loop: rows outside the image are fetched from a clamped row and zeroed by a select, x taps
    // outside the row are read from a clamped LDS index and zeroed the same way (a zero WEIGHT would turn an Inf next door into NaN).
    auto fetch = [&](int iy, Raw (&dst)[NV])                           // global -> registers
    {
        const bool rowOk = iy >= 0 && iy < p.ih;
        const T* row = xp + (int64_t)(rowOk ? iy : 0) * xs2;
        #pragma unroll
        for (int i = 0; i < NV; i++)
        {
            const Raw r = *reinterpret_cast<const Raw*>(row + (int64_t)(tid + 256 * i) * V);
            dst[i] = rowOk ? r : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stage = [&](int buf, const Raw (&src)[NV])                    // registers -> LDS
    {
        #pragma unroll
        for (int i = 0; i < NV; i++) lds[buf * rowVecs + tid + 256 * i] = src[i];
    };
    auto hrow = [&](int buf, float (&h)[NO][V])                        // x taps of one staged row
    {
        #pragma unroll
        for (int j = 0; j < NO; j++)
        {
            #pragma unroll
            for (int i = 0; i < V; i++) h[j][i] = 0.0f;
            #pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const bool ok = (tapOk[j] >> k) & 1u;
                Raw r = lds[buf * rowVecs + (ok ? tap0[j] + k * cvecs : 0)];
                r = ok ? r : make_uint4(0u, 0u, 0u, 0u);
                typename IO::Raw t;
                __builtin_memcpy(&t, &r, 16);
                #pragma unroll
                for (int i = 0; i < V; i++) h[j][i] = fmaf((float)to_acc(t.v[i]), fx[k], h[j][i]);
            }
        }
    };
    // rows come in pairs (a, a + 1): pair t = rows iy0 + 2 t, iy0 + 2 t + 1; pair 0 only fills the window, pair t >= 1 completes output row oyBeg + t - 1
    const int iy0 = 2 * oyBeg - p.pady0;
    const int pairs = oyEnd - oyBeg + 1;
    float w[4][NO][V];
    fetch(iy0, pre[0]); fetch(iy0 + 1, pre[1]);
    for (int t = 0; t < pairs; t++)
    {
        const int a = iy0 + 2 * t;
        stage(0, pre[0]); stage(1, pre[1]);
        if (t + 1 < pairs) { fetch(a + 2, pre[0]); fetch(a + 3, pre[1]); }      // lands while this pair is filtered
        __syncthreads();
        hrow(0, w[2]); hrow(1, w[3]);
        if (t > 0)
        {
            const int oy = oyBeg + t - 1;
            #pragma unroll
            for (int j = 0; j < NO; j++)
            {
                float acc[V];
                #pragma unroll
                for (int i = 0; i < V; i++)
                {
                    float s = 0.0f;
                    #pragma unroll
                    for (int k = 0; k < 4; k++) s = fmaf(w[k][j][i], fy[k], s);
                    acc[i] = s;
                }
                IO::store(yp + (int64_t)oy * ys2 + outOff[j], acc, p.gain);
            }
        }
        #pragma unroll
        for (int j = 0; j < NO; j++)
            #pragma unroll
            for (int i = 0; i < V; i++) { w[0][j][i] = w[2][j][i]; w[1][j][i] = w[3][j][i]; }
        __syncthreads();                                                // the next pair overwrites both buffers
    }
}

template <class T>
int launch_nhwc_down2_lds(UpfirdnArgs& p, hipStream_t stream)
{
    constexpr int V = 16 / (int)sizeof(T);
    const int cvecs = p.c / V;
    const int64_t rowVecs = (int64_t)p.iw * cvecs, outVecs = (int64_t)p.ow * cvecs;
    if (rowVecs > 1024 || outVecs > 512 || rowVecs % 256 || outVecs % 256) return LVG_ERR_UNSUPPORTED;      // whole rows per pass of the 256 threads
    if (p.ys[3] != p.c || p.xs[3] != p.c || p.xs[2] != (int64_t)p.iw * p.c || p.ys[2] != (int64_t)p.ow * p.c) return LVG_ERR_UNSUPPORTED;   // dense rows
    if ((int64_t)p.ih * p.xs[2] >= 0x7fffffffLL || (int64_t)p.oh * p.ys[2] >= 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    // row chunks: enough workgroups to fill the chip twice over, at least 8 output rows each (two halo rows per chunk)
    int chunks = 1;
    while ((int64_t)p.n * chunks < 2048 && (p.oh + chunks) / (chunks + 1) >= 8) chunks++;
    p.chunkRows = (p.oh + chunks - 1) / chunks;
    p.rowChunks = (p.oh + p.chunkRows - 1) / p.chunkRows;
    const int64_t blocks = (int64_t)p.n * p.rowChunks;
    if (blocks > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    const size_t lds = 2 * (size_t)rowVecs * 16;
    const int nv = (int)((rowVecs + 255) / 256), no = (int)((outVecs + 255) / 256);
    #define LVG_DOWN2_LDS(NVv, NOv) hipLaunchKernelGGL((upfirdn2d_nhwc_down2_lds_kernel<T, NVv, NOv>), dim3((unsigned)blocks), dim3(256), lds, stream, p)
    if (nv == 1 && no == 1)      LVG_DOWN2_LDS(1, 1);
    else if (nv == 2 && no == 1) LVG_DOWN2_LDS(2, 1);
    else if (nv == 4 && no == 1) LVG_DOWN2_LDS(4, 1);
    else if (nv == 4 && no == 2) LVG_DOWN2_LDS(4, 2);
    else return LVG_ERR_UNSUPPORTED;
    #undef LVG_DOWN2_LDS
    return lvg_check_launch("upfirdn2d_nhwc_down2_lds_kernel");
}

template <class T, int VB>
int launch_nhwc_vb(UpfirdnArgs& p, hipStream_t stream)
{
    constexpr int V = VB / (int)sizeof(T);
    const int64_t total = (int64_t)p.n * p.oh * p.ow * (p.c / V);
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return LVG_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)blocks), block(256);
    #define LVG_NHWC_CASE(ux, uy, dx, dy) if (p.upx == ux && p.upy == uy && p.downx == dx && p.downy == dy) { \
        hipLaunchKernelGGL((upfirdn2d_nhwc_kernel<T, ux, uy, dx, dy, VB>), grid, block, 0, stream, p); return lvg_check_launch("upfirdn2d_nhwc_kernel"); }
    const bool up2 = p.upx == 2 && p.upy == 2 && p.downx == 1 && p.downy == 1;
    const bool down2 = p.upx == 1 && p.upy == 1 && p.downx == 2 && p.downy == 2;
    if (down2 && VB == 16 && !getenv("LVG_UPFIRDN_NO_LDS"))
    {
        const int rc = launch_nhwc_down2_lds<T>(p, stream);
        if (rc != LVG_ERR_UNSUPPORTED) return rc;
    }
    if (up2 || down2)
    {
        // streaming form: enough lanes even for short frames thanks to row chunks of 16 output rows
        // (equal chunks: 18 output rows are 2 x 10/8, not 16 + 2 -- the short tail chunk measured 37 % slower)
        p.rowChunks = (p.oh + 15) / 16;
        if (const char* e = getenv("LVG_UPFIRDN_CHUNKS")) { if (*e && atoi(e) > 0) p.rowChunks = std::min(atoi(e), (p.oh + 1) / 2); }   // measurement override
        p.chunkRows = (((p.oh + p.rowChunks - 1) / p.rowChunks) + 1) & ~1;
        p.rowChunks = (p.oh + p.chunkRows - 1) / p.chunkRows;
        const int64_t threads = (int64_t)p.n * p.rowChunks * p.ow * (p.c / V);
        const int64_t nb = (threads + 255) / 256;
        if (nb <= 0x7fffffffLL)
        {
            if (up2) hipLaunchKernelGGL((upfirdn2d_nhwc_stream_kernel<T, VB, true>), dim3((unsigned)nb), block, 0, stream, p);
            else     hipLaunchKernelGGL((upfirdn2d_nhwc_stream_kernel<T, VB, false>), dim3((unsigned)nb), block, 0, stream, p);
            return lvg_check_launch("upfirdn2d_nhwc_stream_kernel");
        }
    }
    LVG_NHWC_CASE(2, 2, 1, 1)
    LVG_NHWC_CASE(1, 1, 2, 2)
    LVG_NHWC_CASE(1, 1, 1, 1)
    #undef LVG_NHWC_CASE
    return LVG_ERR_UNSUPPORTED;
}

template <class T>
int launch_nhwc(UpfirdnArgs& p, hipStream_t stream)
{
    // channels-last: unit channel stride, pixels C apart, both tensors
    if (p.f2d || p.fw > 4 || p.fh > 4 || p.c < 2) return LVG_ERR_UNSUPPORTED;
    if (p.xs[1] != 1 || p.ys[1] != 1 || p.xs[3] != p.c || p.ys[3] != p.c) return LVG_ERR_UNSUPPORTED;
    auto fits = [&](int vb) {
        const int v = vb / (int)sizeof(T);
        if (v < 1 || p.c % v) return false;
        if (((uintptr_t)p.x % vb) || ((uintptr_t)p.y % vb)) return false;
        if ((p.xs[0] % v) || (p.xs[2] % v) || (p.ys[0] % v) || (p.ys[2] % v)) return false;
        return true;
    };
    if (fits(16)) return launch_nhwc_vb<T, 16>(p, stream);
    if (fits(8))  return launch_nhwc_vb<T, 8>(p, stream);
    return LVG_ERR_UNSUPPORTED;
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
    if (p.upx == ux && p.upy == uy && p.downx == dx && p.downy == dy) { \
        if (p.fw <= 4 && p.fh <= 4 && 4 % ux == 0 && 4 % uy == 0) { \
            int rcf = launch_fast<T, ux, uy, dx, dy, 4>(p, sep, stream); \
            if (rcf != LVG_ERR_UNSUPPORTED) return rcf; } \
        return launch_tiled<T, ux, uy, dx, dy>(p, sep, stream); }

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
    if constexpr (sizeof(T) <= 4)
    {
        int rcn = launch_nhwc<T>(p, stream);
        if (rcn != LVG_ERR_UNSUPPORTED) return rcn;
    }
    if constexpr (sizeof(T) <= 4) if (wFast)
    {
        int rcw = launch_wave<T>(p, stream);
        if (rcw != LVG_ERR_UNSUPPORTED) return rcw;
        rcw = launch_col<T>(p, stream);
        if (rcw != LVG_ERR_UNSUPPORTED) return rcw;
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
    p.tileW = p.tileH = p.tilesX = p.tilesY = p.inTW = p.inTH = 0;
    p.planesPerBlock = 1; p.uniformPlanes = 0; p.totalPlanes = (int64_t)p.n * p.c; p.laneWLog = p.laneWInLog = 6; p.rowChunks = 1; p.chunkRows = 1 << 30;

    hipStream_t s = (hipStream_t)stream;
    switch (dtype)
    {
        case LVG_F32:  return run<float>(p, s);
        case LVG_F16:  return run<f16_t>(p, s);
        case LVG_BF16: return run<bf16_t>(p, s);
        default:       return run<double>(p, s);
    }
}
