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

} // namespace

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
