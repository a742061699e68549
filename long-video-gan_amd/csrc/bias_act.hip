// bias_act.hip -- fused bias + activation + gain + clamp, forward and 1st/2nd-order
// backward, as one HBM-streaming kernel for gfx950.
//
// Behaviour follows the reference plugin (torch_utils/ops/bias_act.cu:23-147, host side
// bias_act.cpp:32-90); the kernel itself is organised for CDNA4: every lane moves 16 bytes
// per access (4 x f32 / 8 x f16|bf16 / 2 x f64), four independent accesses are in flight per
// lane, and the bias index is resolved once per 16-byte vector instead of once per element.
//
// Roofline: pure stream. Algorithmic bytes per element: grad 0: 2*s (+bias), grad 1: 3*s
// (4*s when xref is needed), grad 2: 4-5*s  (SURVEY.md 8d).

#include "lvg_common.h"

namespace {

struct BiasActArgs
{
    const void* x;      // grad 0: input; grad 1: dy; grad 2: d_dx
    const void* b;
    const void* xref;
    const void* yref;
    const void* dy;
    void*       y;
    int64_t     n;
    int64_t     start;  // first element handled by the scalar kernel
    int64_t     stepB;
    int         sizeB;
    int         grad;
    int         biasMode; // 0 none, 1 one bias per 16-byte vector, 2 contiguous biases (stepB == 1)
    float       alpha;
    float       gain;
    float       clamp;
};

constexpr int kThreads = 256;
constexpr int kUnroll  = 4;

template <class A> __device__ __forceinline__ A lvg_exp(A v);
template <> __device__ __forceinline__ float  lvg_exp<float>(float v)   { return expf(v); }
template <> __device__ __forceinline__ double lvg_exp<double>(double v) { return exp(v); }
template <class A> __device__ __forceinline__ A lvg_log(A v);
template <> __device__ __forceinline__ float  lvg_log<float>(float v)   { return logf(v); }
template <> __device__ __forceinline__ double lvg_log<double>(double v) { return log(v); }
template <class A> __device__ __forceinline__ A lvg_tanh(A v);
template <> __device__ __forceinline__ float  lvg_tanh<float>(float v)   { return tanhf(v); }
template <> __device__ __forceinline__ double lvg_tanh<double>(double v) { return tanh(v); }

// One element. `in` is x (grad 0), dy (grad 1) or d_dx (grad 2); `bias` is already resolved.
template <class A, int ACT>
__device__ __forceinline__ A bias_act_elem(int G, A in, A bias, A xr, A yr, A dyv, A alpha, A gain, A clamp)
{
    const A one = (A)1, two = (A)2, zero = (A)0;
    const A kExpRange = (A)80, kHalfExpRange = (A)40;
    const A kSeluScale = (A)1.0507009873554804934193349852946;
    const A kSeluAlpha = (A)1.6732632423543772848170429916717;

    A v = in;
    if (G == 0) v += bias; else xr += bias;
    const A yy = (gain != zero) ? yr / gain : zero; // forward activation value recovered from the saved output
    A r = zero;

    if (ACT == LVG_ACT_LINEAR)
    {
        if (G <= 1) r = v;
    }
    else if (ACT == LVG_ACT_RELU)
    {
        if (G == 0) r = (v > zero) ? v : zero;
        else if (G == 1) r = (yy > zero) ? v : zero;
    }
    else if (ACT == LVG_ACT_LRELU)
    {
        if (G == 0) r = (v > zero) ? v : v * alpha;
        else if (G == 1) r = (yy > zero) ? v : v * alpha;
    }
    else if (ACT == LVG_ACT_TANH)
    {
        if (G == 0) r = lvg_tanh<A>(v);
        else if (G == 1) r = v * (one - yy * yy);
        else r = v * (one - yy * yy) * (-two * yy);
    }
    else if (ACT == LVG_ACT_SIGMOID)
    {
        if (G == 0) r = (v < -kExpRange) ? zero : one / (lvg_exp<A>(-v) + one);
        else if (G == 1) r = v * yy * (one - yy);
        else r = v * yy * (one - yy) * (one - two * yy);
    }
    else if (ACT == LVG_ACT_ELU)
    {
        if (G == 0) r = (v >= zero) ? v : lvg_exp<A>(v) - one;
        else if (G == 1) r = (yy >= zero) ? v : v * (yy + one);
        else r = (yy >= zero) ? zero : v * (yy + one);
    }
    else if (ACT == LVG_ACT_SELU)
    {
        if (G == 0) r = (v >= zero) ? kSeluScale * v : (kSeluScale * kSeluAlpha) * (lvg_exp<A>(v) - one);
        else if (G == 1) r = (yy >= zero) ? v * kSeluScale : v * (yy + kSeluScale * kSeluAlpha);
        else r = (yy >= zero) ? zero : v * (yy + kSeluScale * kSeluAlpha);
    }
    else if (ACT == LVG_ACT_SOFTPLUS)
    {
        if (G == 0) r = (v > kExpRange) ? v : lvg_log<A>(lvg_exp<A>(v) + one);
        else if (G == 1) r = v * (one - lvg_exp<A>(-yy));
        else { A c = lvg_exp<A>(-yy); r = v * c * (one - c); }
    }
    else if (ACT == LVG_ACT_SWISH)
    {
        if (G == 0)
            r = (v < -kExpRange) ? zero : v / (lvg_exp<A>(-v) + one);
        else
        {
            A c = lvg_exp<A>(xr);
            A d = c + one;
            if (G == 1) r = (xr > kHalfExpRange) ? v : v * c * (xr + d) / (d * d);
            else        r = (xr > kHalfExpRange) ? zero : v * c * (xr * (two - d) + two * d) / (d * d * d);
            yr = (xr < -kExpRange) ? zero : xr / (lvg_exp<A>(-xr) + one) * gain; // forward output, for the clamp mask
        }
    }

    r *= gain * dyv;

    if (clamp >= zero)
    {
        if (G == 0) r = (r > -clamp && r < clamp) ? r : ((r >= zero) ? clamp : -clamp);
        else        r = (yr > -clamp && yr < clamp) ? r : zero;
    }
    return r;
}

// 16 bytes per lane per access, kUnroll accesses per lane.
template <class T, int ACT>
__global__ __launch_bounds__(kThreads) void bias_act_vec_kernel(BiasActArgs p)
{
    typedef typename Elem<T>::acc_t A;
    constexpr int V = Elem<T>::kVec;
    const int64_t nvec = p.n / V;
    const A alpha = (A)p.alpha, gain = (A)p.gain, clamp = (A)p.clamp;
    const int G = p.grad;

    const T* __restrict__ xp  = (const T*)p.x;
    const T* __restrict__ bp  = (const T*)p.b;
    const T* __restrict__ xrp = (const T*)p.xref;
    const T* __restrict__ yrp = (const T*)p.yref;
    const T* __restrict__ dyp = (const T*)p.dy;
    T* __restrict__ yp = (T*)p.y;

    const int64_t base = (int64_t)blockIdx.x * (kUnroll * kThreads) + threadIdx.x;

    Vec16<T> vx[kUnroll], vxr[kUnroll], vyr[kUnroll], vdy[kUnroll];
    #pragma unroll
    for (int u = 0; u < kUnroll; u++)
    {
        const int64_t iv = base + (int64_t)u * kThreads;
        if (iv < nvec)
        {
            vx[u] = load_vec16<T>(xp + iv * V);
            if (xrp) vxr[u] = load_vec16<T>(xrp + iv * V);
            if (yrp) vyr[u] = load_vec16<T>(yrp + iv * V);
            if (dyp) vdy[u] = load_vec16<T>(dyp + iv * V);
        }
    }

    #pragma unroll
    for (int u = 0; u < kUnroll; u++)
    {
        const int64_t iv = base + (int64_t)u * kThreads;
        if (iv >= nvec) continue;
        const int64_t i0 = iv * V;

        A bias[V];
        if (p.biasMode == 1)
        {
            int c;
            if (p.n <= 0x7fffffffLL) c = (int)(((uint32_t)i0 / (uint32_t)p.stepB) % (uint32_t)p.sizeB);
            else                     c = (int)((i0 / p.stepB) % p.sizeB);
            const A bv = (A)to_acc(bp[c]);
            #pragma unroll
            for (int k = 0; k < V; k++) bias[k] = bv;
        }
        else if (p.biasMode == 2)
        {
            int c0;
            if (p.n <= 0x7fffffffLL) c0 = (int)((uint32_t)i0 % (uint32_t)p.sizeB);
            else                     c0 = (int)(i0 % p.sizeB);
            Vec16<T> vb = load_vec16<T>(bp + c0);
            #pragma unroll
            for (int k = 0; k < V; k++) bias[k] = (A)to_acc(vb.v[k]);
        }
        else
        {
            #pragma unroll
            for (int k = 0; k < V; k++) bias[k] = (A)0;
        }

        Vec16<T> out;
        #pragma unroll
        for (int k = 0; k < V; k++)
        {
            const A in  = (A)to_acc(vx[u].v[k]);
            const A xr  = xrp ? (A)to_acc(vxr[u].v[k]) : (A)0;
            const A yr  = yrp ? (A)to_acc(vyr[u].v[k]) : (A)0;
            const A dyv = dyp ? (A)to_acc(vdy[u].v[k]) : (A)1;
            out.v[k] = from_acc<T>(bias_act_elem<A, ACT>(G, in, bias[k], xr, yr, dyv, alpha, gain, clamp));
        }
        store_vec16<T>(yp + i0, out);
    }
}

// One element per lane: tails, unaligned views and bias layouts the vector kernel does not take.
template <class T, int ACT>
__global__ __launch_bounds__(kThreads) void bias_act_scalar_kernel(BiasActArgs p)
{
    typedef typename Elem<T>::acc_t A;
    const int64_t i = p.start + (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= p.n) return;
    const T* bp = (const T*)p.b;
    A bias = (A)0;
    if (bp) bias = (A)to_acc(bp[(i / p.stepB) % p.sizeB]);
    const A in  = (A)to_acc(((const T*)p.x)[i]);
    const A xr  = p.xref ? (A)to_acc(((const T*)p.xref)[i]) : (A)0;
    const A yr  = p.yref ? (A)to_acc(((const T*)p.yref)[i]) : (A)0;
    const A dyv = p.dy   ? (A)to_acc(((const T*)p.dy)[i])   : (A)1;
    ((T*)p.y)[i] = from_acc<T>(bias_act_elem<A, ACT>(p.grad, in, bias, xr, yr, dyv, (A)p.alpha, (A)p.gain, (A)p.clamp));
}

template <class T, int ACT>
int launch_act(BiasActArgs& p, bool vecOk, hipStream_t stream)
{
    constexpr int V = Elem<T>::kVec;
    int64_t done = 0;
    if (vecOk)
    {
        const int64_t nvec = p.n / V;
        if (nvec > 0)
        {
            const int64_t blocks = lvg_ceil_div(nvec, (int64_t)kUnroll * kThreads);
            LVG_REQUIRE(blocks <= 0x7fffffffLL, "bias_act: tensor too large for one launch");
            hipLaunchKernelGGL((bias_act_vec_kernel<T, ACT>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p);
            int rc = lvg_check_launch("bias_act_vec_kernel");
            if (rc) return rc;
        }
        done = nvec * V;
    }
    if (done < p.n)
    {
        p.start = done;
        const int64_t blocks = lvg_ceil_div(p.n - done, (int64_t)kThreads);
        LVG_REQUIRE(blocks <= 0x7fffffffLL, "bias_act: tensor too large for one launch");
        hipLaunchKernelGGL((bias_act_scalar_kernel<T, ACT>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p);
        return lvg_check_launch("bias_act_scalar_kernel");
    }
    return LVG_OK;
}

template <class T>
int launch_dtype(BiasActArgs& p, int act, bool vecOk, hipStream_t stream)
{
    switch (act)
    {
        case LVG_ACT_LINEAR:   return launch_act<T, LVG_ACT_LINEAR>(p, vecOk, stream);
        case LVG_ACT_RELU:     return launch_act<T, LVG_ACT_RELU>(p, vecOk, stream);
        case LVG_ACT_LRELU:    return launch_act<T, LVG_ACT_LRELU>(p, vecOk, stream);
        case LVG_ACT_TANH:     return launch_act<T, LVG_ACT_TANH>(p, vecOk, stream);
        case LVG_ACT_SIGMOID:  return launch_act<T, LVG_ACT_SIGMOID>(p, vecOk, stream);
        case LVG_ACT_ELU:      return launch_act<T, LVG_ACT_ELU>(p, vecOk, stream);
        case LVG_ACT_SELU:     return launch_act<T, LVG_ACT_SELU>(p, vecOk, stream);
        case LVG_ACT_SOFTPLUS: return launch_act<T, LVG_ACT_SOFTPLUS>(p, vecOk, stream);
        case LVG_ACT_SWISH:    return launch_act<T, LVG_ACT_SWISH>(p, vecOk, stream);
    }
    lvg_set_error("bias_act: unknown activation id %d", act);
    return LVG_ERR_INVALID;
}

} // namespace

extern "C" int lvg_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                            void* y, int64_t n, int64_t sizeB, int64_t stepB, int dtype, int grad, int act,
                            float alpha, float gain, float clamp, void* stream)
{
    LVG_REQUIRE(n >= 0, "bias_act: negative element count");
    if (n == 0) return LVG_OK;
    LVG_REQUIRE(x && y, "bias_act: x and y must not be NULL");
    LVG_REQUIRE(grad >= 0 && grad <= 2, "bias_act: grad must be 0, 1 or 2 (got %d)", grad);
    LVG_REQUIRE(dtype >= LVG_F32 && dtype <= LVG_F64, "bias_act: unknown dtype %d", dtype);
    if (b)
    {
        LVG_REQUIRE(sizeB >= 1 && sizeB <= 0x7fffffffLL, "bias_act: b has %lld elements", (long long)sizeB);
        LVG_REQUIRE(stepB >= 1, "bias_act: bias step must be positive");
    }

    BiasActArgs p;
    p.x = x; p.b = b; p.xref = xref; p.yref = yref; p.dy = dy; p.y = y;
    p.n = n; p.start = 0;
    p.stepB = b ? stepB : 1;
    p.sizeB = b ? (int)sizeB : 1;
    p.grad = grad;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;

    const int V = (dtype == LVG_F32) ? 4 : (dtype == LVG_F64) ? 2 : 8;
    bool vecOk = lvg_aligned16(x) && lvg_aligned16(y) && (!xref || lvg_aligned16(xref)) &&
                 (!yref || lvg_aligned16(yref)) && (!dy || lvg_aligned16(dy));
    p.biasMode = 0;
    if (b)
    {
        if (p.stepB % V == 0) p.biasMode = 1;
        else if (p.stepB == 1 && p.sizeB % V == 0 && lvg_aligned16(b)) p.biasMode = 2;
        else vecOk = false;
    }

    hipStream_t s = (hipStream_t)stream;
    switch (dtype)
    {
        case LVG_F32:  return launch_dtype<float>(p, act, vecOk, s);
        case LVG_F16:  return launch_dtype<f16_t>(p, act, vecOk, s);
        case LVG_BF16: return launch_dtype<bf16_t>(p, act, vecOk, s);
        default:       return launch_dtype<double>(p, act, vecOk, s);
    }
}
