// weight_prep.hip -- the weight side of a modulated convolution in ONE pass per direction (gfx950).
//
// The reference normalises and scales the float32 master weight, derives the demodulation term from it and casts it
// to the compute dtype on every forward pass (model/generator_lres.py:97-112):
//     w = w / max|w| over (ci, taps) per output channel        (:98, only when demodulating)
//     w = w / sqrt(ci * taps)                                   (:102-103)
//     w2[co, ci] = sum over the taps of w^2                     (:107 einsum "oizyx,nit->not", weight.square() ...)
//     conv(..., w.type(input.dtype))                            (:119)
// eight tensor passes over the weight forward and about twice that backward in PyTorch. Here:
//   lvg_weight_prep:          w [Co, Ci, taps] f32 -> wp [taps, Co, Ci] 16-bit (the layout conv3d_igemm.hip consumes),
//                             w2 [Co, Ci] f32, amax [Co] f32 (kept for the backward pass)
//   lvg_weight_prep_backward: g (gradient of the 16-bit weight, any strides), g_w2 [Co, Ci] f32 -> dw [Co, Ci, taps] f32
// One workgroup per output channel; the backward pass keeps the combined gradient of the channel in LDS between its two
// sweeps (the max-normalisation couples all elements of a channel). Ties in max|w| share the gradient equally, as
// torch.amax does.

#include "lvg_common.h"

namespace {

constexpr int kPrepThreads = 256;

struct PrepArgs
{
    const float* w;        // [Co][Ci][taps]
    void*        wp;       // [taps][Co][Ci] 16-bit
    float*       w2;       // [Co][Ci] or NULL
    float*       amax;     // [Co] (written forward, read backward)
    const void*  g;        // backward: gradient of the 16-bit weight, element strides below
    const float* gw2;      // backward: gradient of w2 [Co][Ci], or NULL
    float*       dw;       // backward: [Co][Ci][taps]
    int64_t      gStrideCo, gStrideCi, gStrideTap;
    int          Co, Ci, taps;
    float        scale;    // 1 / sqrt(Ci * taps)
    int          normalize;
};

__device__ __forceinline__ float block_max(float v, float* red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < kPrepThreads / 64; i++) r = fmaxf(r, red[i]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ float block_sum(float v, float* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < kPrepThreads / 64; i++) r += red[i];        // fixed order: reproducible
    __syncthreads();
    return r;
}

constexpr int kMaxCiPerThread = 4;      // Ci <= 1024

template <class T>
__global__ __launch_bounds__(kPrepThreads) void weight_prep_kernel(PrepArgs p)
{
    extern __shared__ float lds[];                                   // this output channel's weights, [Ci * taps]
    __shared__ float red[kPrepThreads / 64];
    const int co = blockIdx.x;
    const int n = p.Ci * p.taps;
    const float* w = p.w + (int64_t)co * n;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += kPrepThreads)             // coalesced: the channel's weights are contiguous
    {
        const float v = w[i];
        lds[i] = v;
        m = fmaxf(m, fabsf(v));
    }
    float a = 1.f;
    if (p.normalize) a = block_max(m, red); else __syncthreads();
    if (threadIdx.x == 0) p.amax[co] = a;
    T* wp = static_cast<T*>(p.wp);
    float sq[kMaxCiPerThread] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < p.taps; t++)                                 // tap-major output: consecutive lanes = consecutive ci
    {
        T* row = wp + ((int64_t)t * p.Co + co) * p.Ci;
        #pragma unroll
        for (int k = 0; k < kMaxCiPerThread; k++)
        {
            const int ci = threadIdx.x + k * kPrepThreads;
            if (ci < p.Ci)
            {
                float v = __fmul_rn(__fdiv_rn(lds[ci * p.taps + t], a), p.scale);   // IEEE division, same operation order as the tensor expressions
                asm volatile("" : "+v"(v));       // keep the float32 product: hipcc would fuse multiply + f16 conversion (v_fma_mixlo_f16: ONE rounding)
                sq[k] = fmaf(v, v, sq[k]);
                row[ci] = from_acc<T>(v);
            }
        }
    }
    if (p.w2)
    {
        #pragma unroll
        for (int k = 0; k < kMaxCiPerThread; k++)
        {
            const int ci = threadIdx.x + k * kPrepThreads;
            if (ci < p.Ci) p.w2[(int64_t)co * p.Ci + ci] = sq[k];
        }
    }
}

template <class T>
__global__ __launch_bounds__(kPrepThreads) void weight_prep_backward_kernel(PrepArgs p)
{
    extern __shared__ float lds[];                                   // [0, n): the channel's weights; [n, 2n): its combined gradient
    __shared__ float red[kPrepThreads / 64];
    const int co = blockIdx.x;
    const int n = p.Ci * p.taps;
    float* wbuf = lds;
    float* gbuf = lds + n;
    const float* w = p.w + (int64_t)co * n;
    for (int i = threadIdx.x; i < n; i += kPrepThreads) wbuf[i] = w[i];
    __syncthreads();
    const T* g = static_cast<const T*>(p.g) + (int64_t)co * p.gStrideCo;
    const float a = p.normalize ? p.amax[co] : 1.f;
    const float inv = 1.f / a;
    // sweep 1: G = dL/d(scaled weight) = g + 2 * ws * g_w2 (ws = wn * scale, wn = w / a);  dot = sum_j scale * G_j * wn_j
    float dot = 0.f, ties = 0.f;
    for (int t = 0; t < p.taps; t++)
    {
        #pragma unroll
        for (int k = 0; k < kMaxCiPerThread; k++)
        {
            const int ci = threadIdx.x + k * kPrepThreads;
            if (ci < p.Ci)
            {
                const int i = ci * p.taps + t;
                const float wv = wbuf[i];
                const float wn = wv * inv;
                const float gq = p.gw2 ? 2.f * p.gw2[(int64_t)co * p.Ci + ci] : 0.f;
                const float G = to_acc(g[ci * p.gStrideCi + t * p.gStrideTap]) + gq * (wn * p.scale);
                gbuf[i] = G;
                dot = fmaf(G * p.scale, wn, dot);
                if (p.normalize && fabsf(wv) == a) ties += 1.f;
            }
        }
    }
    float corr = 0.f;
    if (p.normalize)
    {
        dot = block_sum(dot, red);
        ties = block_sum(ties, red);
        corr = dot * inv / fmaxf(ties, 1.f);
    }
    else __syncthreads();
    // sweep 2: d w_i = scale * G_i / a  -  (sum_j scale * G_j * wn_j / a) * d a / d w_i,   d a / d w_i = sign(w_i) / ties on the maxima
    float* dw = p.dw + (int64_t)co * n;
    for (int i = threadIdx.x; i < n; i += kPrepThreads)             // coalesced
    {
        const float wv = wbuf[i];
        float d = gbuf[i] * p.scale * inv;
        if (p.normalize && fabsf(wv) == a) d -= corr * (wv > 0.f ? 1.f : (wv < 0.f ? -1.f : 0.f));
        dw[i] = d;
    }
}

// ---- RMS-normalised weights of the 2-D modulated convolution (reference model/generator_sres.py:50-58) ------------------------------
//     w' = w * rsqrt(mean over (ci, taps) of w^2) * scale          scale = 1 / sqrt(fan_in) on the 16-bit layers
//     w2[co, ci] = sum over the taps of w'^2                        (the weight half of the demodulation term)
//     wp = w' in the compute dtype, [taps][co_pad][ci_pad] with zero-filled padding: what lvg_conv2d_frames consumes
// One workgroup per (padded) output channel, the channel's weights in LDS between the statistic and the output sweep.
struct Prep2dArgs
{
    const float* w;        // [Co][Ci][taps]
    void*        wp;       // [taps][coPad][ciPad] 16-bit
    float*       w2;       // [Co][Ci]
    float*       stat;     // [Co]: rsqrt(mean w^2)
    const float* g;        // backward: gradient of wp's elements, float32 [taps][coPad][ciPad], or NULL
    const float* gw2;      // backward: gradient of w2 [Co][Ci], or NULL
    float*       dw;       // backward: [Co][Ci][taps]
    int          Co, Ci, taps, coPad, ciPad;
    float        scale;
};

template <class T>
__global__ __launch_bounds__(kPrepThreads) void weight_prep2d_kernel(Prep2dArgs p)
{
    extern __shared__ float lds[];
    __shared__ float red[kPrepThreads / 64];
    const int co = blockIdx.x;
    T* wp = static_cast<T*>(p.wp);
    if (co >= p.Co)                                                   // padding rows
    {
        for (int t = 0; t < p.taps; t++)
            for (int ci = threadIdx.x; ci < p.ciPad; ci += kPrepThreads) wp[((int64_t)t * p.coPad + co) * p.ciPad + ci] = from_acc<T>(0.0f);
        return;
    }
    const int n = p.Ci * p.taps;
    const float* w = p.w + (int64_t)co * n;
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += kPrepThreads)
    {
        const float v = w[i];
        lds[i] = v;
        ss = fmaf(v, v, ss);
    }
    ss = block_sum(ss, red);                                          // (also publishes lds)
    const float a = 1.0f / sqrtf(ss / (float)n);
    if (threadIdx.x == 0) p.stat[co] = a;
    for (int ci = threadIdx.x; ci < p.ciPad; ci += kPrepThreads)
    {
        float sq = 0.f;
        for (int t = 0; t < p.taps; t++)
        {
            float v = 0.f;
            if (ci < p.Ci)
            {
                v = __fmul_rn(__fmul_rn(lds[ci * p.taps + t], a), p.scale);
                asm volatile("" : "+v"(v));       // keep the float32 product (no fused multiply + conversion: one rounding each, as the tensor expressions)
                sq = fmaf(v, v, sq);
            }
            wp[((int64_t)t * p.coPad + co) * p.ciPad + ci] = from_acc<T>(v);
        }
        if (ci < p.Ci) p.w2[(int64_t)co * p.Ci + ci] = sq;
    }
}

// d w from G = g + 2 w' g_w2 (either part may be absent):  d w_k = scale a G_k - (scale a^3 / n) (sum_i G_i w_i) w_k,  a = rsqrt(mean w^2)
__global__ __launch_bounds__(kPrepThreads) void weight_prep2d_backward_kernel(Prep2dArgs p)
{
    extern __shared__ float lds[];                                   // [0, n): the channel's weights; [n, 2n): its combined gradient
    __shared__ float red[kPrepThreads / 64];
    const int co = blockIdx.x;
    const int n = p.Ci * p.taps;
    float* wbuf = lds;
    float* gbuf = lds + n;
    const float* w = p.w + (int64_t)co * n;
    for (int i = threadIdx.x; i < n; i += kPrepThreads) wbuf[i] = w[i];
    __syncthreads();
    const float a = p.stat[co];
    float dot = 0.f;
    for (int ci = threadIdx.x; ci < p.Ci; ci += kPrepThreads)
    {
        const float gq = p.gw2 ? 2.f * p.gw2[(int64_t)co * p.Ci + ci] * a * p.scale : 0.f;
        for (int t = 0; t < p.taps; t++)
        {
            const int i = ci * p.taps + t;
            const float wv = wbuf[i];
            const float G = (p.g ? p.g[((int64_t)t * p.coPad + co) * p.ciPad + ci] : 0.f) + gq * wv;
            gbuf[i] = G;
            dot = fmaf(G, wv, dot);
        }
    }
    dot = block_sum(dot, red);
    const float k1 = p.scale * a, k2 = p.scale * a * a * a * dot / (float)n;
    float* dw = p.dw + (int64_t)co * n;
    for (int i = threadIdx.x; i < n; i += kPrepThreads) dw[i] = k1 * gbuf[i] - k2 * wbuf[i];      // coalesced
}

// wp [taps][Co][Ci] -> wt [taps][Ci][Co] with the tap order reversed: the weight of the data-gradient convolution (mirrored taps,
// channel roles swapped) in the layout conv3d_igemm.hip consumes. 64 x 64 tiles through LDS, 16-bit elements moved as raw words.
__global__ __launch_bounds__(256) void weight_dgrad_pack_kernel(const uint16_t* __restrict__ wp, uint16_t* __restrict__ wt, int taps, int Co, int Ci)
{
    __shared__ uint16_t tile[64][66];
    const int t = blockIdx.z, co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
    const uint16_t* src = wp + (int64_t)t * Co * Ci;
    uint16_t* dst = wt + (int64_t)(taps - 1 - t) * Ci * Co;
    for (int i = threadIdx.x; i < 64 * 64; i += 256)
    {
        const int r = i >> 6, c = i & 63;                              // r: co, c: ci (coalesced along ci)
        tile[r][c] = (co0 + r < Co && ci0 + c < Ci) ? src[(int64_t)(co0 + r) * Ci + ci0 + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256)
    {
        const int r = i >> 6, c = i & 63;                              // r: ci, c: co (coalesced along co)
        if (ci0 + r < Ci && co0 + c < Co) dst[(int64_t)(ci0 + r) * Co + co0 + c] = tile[c][r];
    }
}

} // namespace

extern "C" int lvg_weight_dgrad_pack(const void* wp, void* wt, int taps, int co, int ci, void* stream)
{
    LVG_REQUIRE(wp && wt && taps > 0 && co > 0 && ci > 0 && taps <= 65535, "weight_dgrad_pack: empty input");
    hipLaunchKernelGGL(weight_dgrad_pack_kernel, dim3((unsigned)lvg_ceil_div(ci, 64), (unsigned)lvg_ceil_div(co, 64), (unsigned)taps), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint16_t*>(wp), static_cast<uint16_t*>(wt), taps, co, ci);
    return lvg_check_launch("weight_dgrad_pack");
}

extern "C" int lvg_weight_prep(const float* w, void* wp, float* w2, float* amax, int co, int ci, int taps, float scale, int normalize,
                               int dtype, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "weight_prep: float16 / bfloat16 output only (dtype %d)", dtype);
    LVG_REQUIRE(co > 0 && ci > 0 && taps > 0 && w && wp && amax, "weight_prep: empty input");
    PrepArgs a = {};
    a.w = w; a.wp = wp; a.w2 = w2; a.amax = amax; a.Co = co; a.Ci = ci; a.taps = taps; a.scale = scale; a.normalize = normalize;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)ci * taps * sizeof(float);
    LVG_REQUIRE(ci <= kPrepThreads * kMaxCiPerThread && lds <= 150 * 1024, "weight_prep: %d x %d elements per output channel: no kernel", ci, taps);
    auto launch = [&](auto kern) -> int
    {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        {
            (void)hipGetLastError();
            lvg_set_error("weight_prep: cannot opt in to %zu bytes of LDS", lds);
            return LVG_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3(co), dim3(kPrepThreads), lds, s, a);
        return lvg_check_launch("weight_prep");
    };
    return dtype == LVG_BF16 ? launch(weight_prep_kernel<bf16_t>) : launch(weight_prep_kernel<f16_t>);
}

extern "C" int lvg_weight_prep_backward(const float* w, const float* amax, const void* g, const int64_t* g_strides, const float* g_w2,
                                        float* dw, int co, int ci, int taps, float scale, int normalize, int dtype, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "weight_prep_backward: float16 / bfloat16 gradient only (dtype %d)", dtype);
    LVG_REQUIRE(co > 0 && ci > 0 && taps > 0 && w && g && dw && amax && g_strides, "weight_prep_backward: empty input");
    const size_t lds = 2 * (size_t)ci * taps * sizeof(float);
    LVG_REQUIRE(ci <= kPrepThreads * kMaxCiPerThread && lds <= 150 * 1024, "weight_prep_backward: %d x %d elements per output channel: no kernel", ci, taps);
    PrepArgs a = {};
    a.w = w; a.amax = const_cast<float*>(amax); a.g = g; a.gw2 = g_w2; a.dw = dw;
    a.gStrideCo = g_strides[0]; a.gStrideCi = g_strides[1]; a.gStrideTap = g_strides[2];
    a.Co = co; a.Ci = ci; a.taps = taps; a.scale = scale; a.normalize = normalize;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto launch = [&](auto kern) -> int
    {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        {
            (void)hipGetLastError();
            lvg_set_error("weight_prep_backward: cannot opt in to %zu bytes of LDS", lds);
            return LVG_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3(co), dim3(kPrepThreads), lds, s, a);
        return lvg_check_launch("weight_prep_backward");
    };
    return dtype == LVG_BF16 ? launch(weight_prep_backward_kernel<bf16_t>) : launch(weight_prep_backward_kernel<f16_t>);
}

extern "C" int lvg_weight_prep2d(const float* w, void* wp, void* wt, float* w2, float* stat, int co, int ci, int taps, int co_pad, int ci_pad,
                                 float scale, int dtype, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "weight_prep2d: float16 / bfloat16 output only (dtype %d)", dtype);
    LVG_REQUIRE(co > 0 && ci > 0 && taps > 0 && taps <= 65535 && co_pad >= co && ci_pad >= ci && w && wp && w2 && stat, "weight_prep2d: empty input");
    const size_t lds = (size_t)ci * taps * sizeof(float);
    LVG_REQUIRE(lds <= 150 * 1024, "weight_prep2d: %d x %d elements per output channel: no kernel", ci, taps);
    Prep2dArgs a = {};
    a.w = w; a.wp = wp; a.w2 = w2; a.stat = stat; a.Co = co; a.Ci = ci; a.taps = taps; a.coPad = co_pad; a.ciPad = ci_pad; a.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto launch = [&](auto kern) -> int
    {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        {
            (void)hipGetLastError();
            lvg_set_error("weight_prep2d: cannot opt in to %zu bytes of LDS", lds);
            return LVG_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3(co_pad), dim3(kPrepThreads), lds, s, a);
        return lvg_check_launch("weight_prep2d");
    };
    if (int rc = dtype == LVG_BF16 ? launch(weight_prep2d_kernel<bf16_t>) : launch(weight_prep2d_kernel<f16_t>)) return rc;
    if (wt) return lvg_weight_dgrad_pack(wp, wt, taps, co_pad, ci_pad, stream);
    return LVG_OK;
}

extern "C" int lvg_weight_prep2d_backward(const float* w, const float* stat, const float* g, const float* g_w2, float* dw,
                                          int co, int ci, int taps, int co_pad, int ci_pad, float scale, void* stream)
{
    LVG_REQUIRE(co > 0 && ci > 0 && taps > 0 && co_pad >= co && ci_pad >= ci && w && stat && dw && (g || g_w2), "weight_prep2d_backward: empty input");
    const size_t lds = 2 * (size_t)ci * taps * sizeof(float);
    LVG_REQUIRE(lds <= 150 * 1024, "weight_prep2d_backward: %d x %d elements per output channel: no kernel", ci, taps);
    Prep2dArgs a = {};
    a.w = w; a.stat = const_cast<float*>(stat); a.g = g; a.gw2 = g_w2; a.dw = dw;
    a.Co = co; a.Ci = ci; a.taps = taps; a.coPad = co_pad; a.ciPad = ci_pad; a.scale = scale;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(weight_prep2d_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    {
        (void)hipGetLastError();
        lvg_set_error("weight_prep2d_backward: cannot opt in to %zu bytes of LDS", lds);
        return LVG_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(weight_prep2d_backward_kernel, dim3(co), dim3(kPrepThreads), lds, static_cast<hipStream_t>(stream), a);
    return lvg_check_launch("weight_prep2d_backward");
}
