// 1 x 1 convolutions with a THIN side (3 image channels) on channels-last frames: the generator's ToRGB (64 -> 3, reference
// model/generator_lres.py:600-640 ToRGB.forward -> temporal_modulated_conv3d with a [3, C, 1, 1, 1] weight) and the discriminator's first layer
// (3 -> 32, model/discriminator_lres.py:169 Conv3dLayer with kernel 1), and their gradients. These are pure HBM streams over the WIDE tensor
// ([pixels, W] channels-last, W = 8 .. 128 channels): (W + T) * s bytes per pixel for T <= 4 thin channels. The library's implicit-GEMM kernels
// run them at 1/2 .. 1/5 of that (ToRGB forward 147 us, backward 339 us at 1024 frames of 36 x 64 against 60 / 120 us of traffic).
//
//   thin_out:  y[m][t] = sum_c x[m][c] w[t][c]          (ToRGB forward; data gradient of the 3 -> 32 layer)
//   thin_in:   y[m][c] = sum_t x[m][t] w[t][c]          (3 -> 32 forward; data gradient of ToRGB)
//   wgrad:     g[t][c] = sum_m thin[m][t] wide[m][c]    (both weight gradients), per-workgroup partial sums in a fixed order
//
// A pixel's W channels are spread over LP = W / 8 lanes, one 16-byte vector each, so a wave instruction moves 1 KiB of contiguous memory; the thin side is
// 2-byte accesses (6 bytes per pixel: 2 % of the traffic). Float32 accumulation, one rounding on store; weights arrive as float32 [T, W].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lvg_common.h"

namespace {

constexpr int kMaxThin = 4;

struct ThinArgs
{
    const void* wide;        // [M, W] (thin_out: input; wgrad: the wide operand)
    const void* thin;        // [M, T] (thin_in: input; wgrad: the thin operand)
    const float* w;          // [T, W] float32
    void* out;               // thin_out: [M, T]; thin_in: [M, W]; wgrad: float [blocks, T, W]
    int64_t M;
    int W, T, lpShift;       // lanes per pixel = 1 << lpShift = W / 8
};

template <class T> __device__ __forceinline__ void load8(const T* p, float (&v)[8])
{
    const Vec16<T> r = load_vec16(p);
    #pragma unroll
    for (int i = 0; i < 8; i++) v[i] = to_acc(r.v[i]);
}

// ---- y[m][t] = sum_c x[m][c] w[t][c] ---------------------------------------------------------------------------------------------
template <class T> __global__ __launch_bounds__(256) void thin_out_kernel(ThinArgs q)
{
    const int lp = 1 << q.lpShift;
    const int sub = threadIdx.x & (lp - 1);                      // which 8-channel vector of the pixel
    float wv[kMaxThin][8];
    #pragma unroll
    for (int t = 0; t < kMaxThin; t++)
        #pragma unroll
        for (int i = 0; i < 8; i++) wv[t][i] = t < q.T ? q.w[t * q.W + sub * 8 + i] : 0.f;
    const T* x = static_cast<const T*>(q.wide);
    T* y = static_cast<T*>(q.out);
    const int64_t ppb = 256 >> q.lpShift;                        // pixels per workgroup and pass
    const int64_t stride = (int64_t)gridDim.x * ppb;
    for (int64_t m = (int64_t)blockIdx.x * ppb + (threadIdx.x >> q.lpShift); m < q.M; m += stride)
    {
        float v[8];
        load8(x + m * q.W + sub * 8, v);
        float acc[kMaxThin];
        #pragma unroll
        for (int t = 0; t < kMaxThin; t++)
        {
            float s = 0.f;
            #pragma unroll
            for (int i = 0; i < 8; i++) s = fmaf(v[i], wv[t][i], s);
            acc[t] = s;
        }
        // sum over the pixel's lanes (a fixed butterfly: every lane ends with the same total)
        for (int d = 1; d < lp; d <<= 1)
            #pragma unroll
            for (int t = 0; t < kMaxThin; t++) acc[t] += __shfl_xor(acc[t], d, 64);
        #pragma unroll
        for (int t = 0; t < kMaxThin; t++)                       // output t leaves through lane t mod lp of the pixel
            if (t < q.T && (t & (lp - 1)) == sub) y[m * q.T + t] = from_acc<T>(acc[t]);
    }
}

// ---- y[m][c] = sum_t x[m][t] w[t][c] ---------------------------------------------------------------------------------------------
template <class T> __global__ __launch_bounds__(256) void thin_in_kernel(ThinArgs q)
{
    const int lp = 1 << q.lpShift;
    const int sub = threadIdx.x & (lp - 1);
    float wv[kMaxThin][8];
    #pragma unroll
    for (int t = 0; t < kMaxThin; t++)
        #pragma unroll
        for (int i = 0; i < 8; i++) wv[t][i] = t < q.T ? q.w[t * q.W + sub * 8 + i] : 0.f;
    const T* x = static_cast<const T*>(q.thin);
    T* y = static_cast<T*>(q.out);
    const int64_t ppb = 256 >> q.lpShift;
    const int64_t stride = (int64_t)gridDim.x * ppb;
    for (int64_t m = (int64_t)blockIdx.x * ppb + (threadIdx.x >> q.lpShift); m < q.M; m += stride)
    {
        float xt[kMaxThin];
        #pragma unroll
        for (int t = 0; t < kMaxThin; t++) xt[t] = t < q.T ? to_acc(x[m * q.T + t]) : 0.f;
        Vec16<T> r;
        #pragma unroll
        for (int i = 0; i < 8; i++)
        {
            float s = 0.f;
            #pragma unroll
            for (int t = 0; t < kMaxThin; t++) s = fmaf(xt[t], wv[t][i], s);
            r.v[i] = from_acc<T>(s);
        }
        store_vec16(y + m * q.W + sub * 8, r);
    }
}

// ---- g[t][c] = sum_m thin[m][t] wide[m][c]: one partial [T, W] per workgroup ---------------------------------------------------------
template <class T> __global__ __launch_bounds__(256) void thin_wgrad_kernel(ThinArgs q)
{
    __shared__ float red[4][kMaxThin][128];                      // per wave: [t][channel]
    const int lp = 1 << q.lpShift;
    const int sub = threadIdx.x & (lp - 1), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* a = static_cast<const T*>(q.wide);
    const T* b = static_cast<const T*>(q.thin);
    float acc[kMaxThin][8];
    #pragma unroll
    for (int t = 0; t < kMaxThin; t++)
        #pragma unroll
        for (int i = 0; i < 8; i++) acc[t][i] = 0.f;
    const int64_t ppb = 256 >> q.lpShift;
    // a workgroup owns a CONTIGUOUS range of pixels (the partial sums do not depend on the grid's interleaving; the caller adds them in block order)
    const int64_t per = ((q.M + gridDim.x - 1) / gridDim.x + ppb - 1) / ppb * ppb;
    const int64_t beg = (int64_t)blockIdx.x * per, end = beg + per < q.M ? beg + per : q.M;
    for (int64_t m = beg + (threadIdx.x >> q.lpShift); m < end; m += ppb)
    {
        float v[8], xt[kMaxThin];
        load8(a + m * q.W + sub * 8, v);
        #pragma unroll
        for (int t = 0; t < kMaxThin; t++) xt[t] = t < q.T ? to_acc(b[m * q.T + t]) : 0.f;
        #pragma unroll
        for (int t = 0; t < kMaxThin; t++)
            #pragma unroll
            for (int i = 0; i < 8; i++) acc[t][i] = fmaf(xt[t], v[i], acc[t][i]);
    }
    // lanes that hold the same channel vector (lane = sub mod lp): butterfly over the wave, then the four waves through LDS
    for (int d = lp; d < 64; d <<= 1)
        #pragma unroll
        for (int t = 0; t < kMaxThin; t++)
            #pragma unroll
            for (int i = 0; i < 8; i++) acc[t][i] += __shfl_xor(acc[t][i], d, 64);
    if (lane < lp)
        #pragma unroll
        for (int t = 0; t < kMaxThin; t++)
            #pragma unroll
            for (int i = 0; i < 8; i++) red[wave][t][sub * 8 + i] = acc[t][i];
    __syncthreads();
    float* out = static_cast<float*>(q.out) + (int64_t)blockIdx.x * q.T * q.W;
    for (int i = threadIdx.x; i < q.T * q.W; i += 256)
    {
        const int t = i / q.W, c = i - t * q.W;
        out[i] = (red[0][t][c] + red[1][t][c]) + (red[2][t][c] + red[3][t][c]);
    }
}

int lp_shift(int wide)
{
    switch (wide) { case 8: return 0; case 16: return 1; case 32: return 2; case 64: return 3; case 128: return 4; default: return -1; }
}

int grid_for(int64_t pixels, int lpShift)
{
    const int64_t ppb = 256 >> lpShift;
    const int64_t need = (pixels + ppb - 1) / ppb;
    const int64_t cap = 256 * 8;                                 // eight workgroups per CU, grid-stride beyond
    return (int)(need < cap ? (need > 0 ? need : 1) : cap);
}

} // namespace

#define LVG_THIN_CHECK(name) \
    const int sh = lp_shift(wide); \
    LVG_REQUIRE(sh >= 0 && thin >= 1 && thin <= kMaxThin, name ": wide channels must be 8 / 16 / 32 / 64 / 128 and thin channels 1 .. 4 (got %d, %d)", wide, thin); \
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, name ": float16 / bfloat16 tensors only"); \
    LVG_REQUIRE(pixels >= 0, name ": negative size"); \
    if (pixels == 0) return LVG_OK;

extern "C" int lvg_pointwise_thin_out(const void* x, const float* w, void* y, int64_t pixels, int wide, int thin, int dtype, void* stream)
{
    LVG_THIN_CHECK("lvg_pointwise_thin_out")
    LVG_REQUIRE(x && w && y && lvg_aligned16(x), "lvg_pointwise_thin_out: null or misaligned pointer");
    ThinArgs q{x, nullptr, w, y, pixels, wide, thin, sh};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LVG_F16) hipLaunchKernelGGL(thin_out_kernel<f16_t>, dim3(grid_for(pixels, sh)), dim3(256), 0, s, q);
    else                  hipLaunchKernelGGL(thin_out_kernel<bf16_t>, dim3(grid_for(pixels, sh)), dim3(256), 0, s, q);
    return lvg_check_launch("lvg_pointwise_thin_out");
}

extern "C" int lvg_pointwise_thin_in(const void* x, const float* w, void* y, int64_t pixels, int wide, int thin, int dtype, void* stream)
{
    LVG_THIN_CHECK("lvg_pointwise_thin_in")
    LVG_REQUIRE(x && w && y && lvg_aligned16(y), "lvg_pointwise_thin_in: null or misaligned pointer");
    ThinArgs q{nullptr, x, w, y, pixels, wide, thin, sh};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LVG_F16) hipLaunchKernelGGL(thin_in_kernel<f16_t>, dim3(grid_for(pixels, sh)), dim3(256), 0, s, q);
    else                  hipLaunchKernelGGL(thin_in_kernel<bf16_t>, dim3(grid_for(pixels, sh)), dim3(256), 0, s, q);
    return lvg_check_launch("lvg_pointwise_thin_in");
}

extern "C" int lvg_pointwise_thin_wgrad_blocks(int64_t pixels, int wide)
{
    const int sh = lp_shift(wide);
    if (sh < 0 || pixels <= 0) return 0;
    const int64_t ppb = 256 >> sh;
    const int64_t need = (pixels + ppb * 16 - 1) / (ppb * 16);   // at least 16 passes per workgroup
    return (int)(need < 1024 ? (need > 0 ? need : 1) : 1024);
}

extern "C" int lvg_pointwise_thin_wgrad(const void* wideT, const void* thinT, float* partial, int64_t pixels, int wide, int thin, int dtype, int blocks, void* stream)
{
    LVG_THIN_CHECK("lvg_pointwise_thin_wgrad")
    LVG_REQUIRE(wideT && thinT && partial && lvg_aligned16(wideT), "lvg_pointwise_thin_wgrad: null or misaligned pointer");
    LVG_REQUIRE(blocks == lvg_pointwise_thin_wgrad_blocks(pixels, wide), "lvg_pointwise_thin_wgrad: blocks must come from lvg_pointwise_thin_wgrad_blocks");
    ThinArgs q{wideT, thinT, nullptr, partial, pixels, wide, thin, sh};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LVG_F16) hipLaunchKernelGGL(thin_wgrad_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, q);
    else                  hipLaunchKernelGGL(thin_wgrad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, q);
    return lvg_check_launch("lvg_pointwise_thin_wgrad");
}
