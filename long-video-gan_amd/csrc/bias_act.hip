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
    int         biasMode; // 0 none, 1 one bias per row, 2 contiguous biases along the row (stepB == 1)
    int64_t     rows;     // vector-kernel view: rows x rowVecs 16-byte vectors
    int64_t     rowVecs;
    float       alpha;
    float       gain;
    float       clamp;
    float*      dbPartial; // grad 1, channels-last stream: per-workgroup sums of the result over its pixels, [gridDim.x][chanVecs * V], or NULL
    int         chanVecs;  // ... channels / V (divides kThreads)
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
template <class A, int ACT, int G>
__device__ __forceinline__ A bias_act_elem(A in, A bias, A xr, A yr, A dyv, A alpha, A gain, A clamp)
{
    const A one = (A)1, two = (A)2, zero = (A)0;
    const A kExpRange = (A)80, kHalfExpRange = (A)40;
    const A kSeluScale = (A)1.0507009873554804934193349852946;
    const A kSeluAlpha = (A)1.6732632423543772848170429916717;

    A v = in;
    if (G == 0) v += bias; else xr += bias;
    // Forward activation value recovered from the saved output. relu / lrelu only need its sign,
    // which spares the IEEE division on the hot path.
    A yy = zero;
    bool yyPos = false;
    if (G > 0)
    {
        if (ACT == LVG_ACT_RELU || ACT == LVG_ACT_LRELU)
            yyPos = (gain > zero) ? (yr > zero) : ((gain < zero) ? (yr < zero) : false);
        else
            yy = (gain != zero) ? yr / gain : zero;
    }
    A r = zero;

    if (ACT == LVG_ACT_LINEAR)
    {
        if (G <= 1) r = v;
    }
    else if (ACT == LVG_ACT_RELU)
    {
        if (G == 0) r = (v > zero) ? v : zero;
        else if (G == 1) r = yyPos ? v : zero;
    }
    else if (ACT == LVG_ACT_LRELU)
    {
        if (G == 0) r = (v > zero) ? v : v * alpha;
        else if (G == 1) r = yyPos ? v : v * alpha;
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

// Vector kernel: 16 bytes per lane per access, UNROLL independent accesses per lane.
// The tensor is viewed as `rows` x `rowVecs` 16-byte vectors such that the bias is resolved
// without any per-element integer division:
//   biasMode 1 (stepB % V == 0): a row = one bias segment of stepB elements, bias = b[row % sizeB];
//   biasMode 2 (stepB == 1, sizeB % V == 0): a row = sizeB elements, bias = b[col*V .. col*V+V-1];
//   biasMode 0: a single row.
// grid = (ceil(rowVecs / (UNROLL*kThreads)), rowsY, rowsZ), row = z * rowsY + y.
template <class T, int ACT, int G, int UNROLL>
__global__ __launch_bounds__(kThreads) void bias_act_vec_kernel(BiasActArgs p)
{
    typedef typename Elem<T>::acc_t A;
    constexpr int V = Elem<T>::kVec;
    const A alpha = (A)p.alpha, gain = (A)p.gain, clamp = (A)p.clamp;

    const int64_t row = (int64_t)blockIdx.z * gridDim.y + blockIdx.y;
    if (row >= p.rows) return;
    const int64_t rowBase = row * p.rowVecs * V; // first element of this row

    const T* __restrict__ xp  = (const T*)p.x + rowBase;
    const T* __restrict__ bp  = (const T*)p.b;
    const T* __restrict__ xrp = p.xref ? (const T*)p.xref + rowBase : nullptr;
    const T* __restrict__ yrp = p.yref ? (const T*)p.yref + rowBase : nullptr;
    const T* __restrict__ dyp = p.dy ? (const T*)p.dy + rowBase : nullptr;
    T* __restrict__ yp = (T*)p.y + rowBase;

    const int64_t col0 = (int64_t)blockIdx.x * (UNROLL * kThreads) + threadIdx.x;

    Vec16<T> vx[UNROLL], vxr[UNROLL], vyr[UNROLL], vdy[UNROLL];
    #pragma unroll
    for (int u = 0; u < UNROLL; u++)
    {
        const int64_t col = col0 + (int64_t)u * kThreads;
        if (col < p.rowVecs)
        {
            vx[u] = load_vec16<T>(xp + col * V);
            if (G > 0 && ACT == LVG_ACT_SWISH) vxr[u] = load_vec16<T>(xrp + col * V);
            if (G > 0 && ACT != LVG_ACT_SWISH && ACT != LVG_ACT_LINEAR) vyr[u] = load_vec16<T>(yrp + col * V);
            if (G > 0 && ACT == LVG_ACT_LINEAR && yrp) vyr[u] = load_vec16<T>(yrp + col * V);
            if (G == 2) vdy[u] = load_vec16<T>(dyp + col * V);
        }
    }

    A rowBias = (A)0;
    if (p.biasMode == 1) rowBias = (A)to_acc(bp[row % p.sizeB]);
    float dbSum[V];
    #pragma unroll
    for (int k = 0; k < V; k++) dbSum[k] = 0.f;

    #pragma unroll
    for (int u = 0; u < UNROLL; u++)
    {
        const int64_t col = col0 + (int64_t)u * kThreads;
        if (col >= p.rowVecs) continue;

        A bias[V];
        if (p.biasMode == 2)
        {
            Vec16<T> vb = load_vec16<T>(bp + col * V);
            #pragma unroll
            for (int k = 0; k < V; k++) bias[k] = (A)to_acc(vb.v[k]);
        }
        else
        {
            #pragma unroll
            for (int k = 0; k < V; k++) bias[k] = rowBias;
        }

        Vec16<T> out;
        #pragma unroll
        for (int k = 0; k < V; k++)
        {
            const A in  = (A)to_acc(vx[u].v[k]);
            const A xr  = (G > 0 && ACT == LVG_ACT_SWISH) ? (A)to_acc(vxr[u].v[k]) : (A)0;
            const A yr  = (G > 0 && ACT != LVG_ACT_SWISH && (ACT != LVG_ACT_LINEAR || yrp)) ? (A)to_acc(vyr[u].v[k]) : (A)0;
            const A dyv = (G == 2) ? (A)to_acc(vdy[u].v[k]) : (A)1;
            out.v[k] = from_acc<T>(bias_act_elem<A, ACT, G>(in, bias[k], xr, yr, dyv, alpha, gain, clamp));
        }
        store_vec16<T>(yp + col * V, out);
        if (G == 1 && UNROLL == kUnroll && p.dbPartial)
        {
            #pragma unroll
            for (int k = 0; k < V; k++) dbSum[k] += (float)to_acc(out.v[k]);    // the ROUNDED result, like a reduction of the stored tensor
        }
    }
    if (G == 1 && UNROLL == kUnroll && p.dbPartial)
    {
        // Bias gradient of the layer in the same pass (the separate reduction re-read the whole gradient: 72 us per 300 MB tensor):
        // the tensor is a channels-last stream, kThreads and the workgroup's first vector are multiples of chanVecs, so a lane sees
        // ONE channel vector in all its UNROLL accesses. Lanes with the same channel vector are added in a fixed order.
        __shared__ float red[kThreads][V + 1];
        #pragma unroll
        for (int k = 0; k < V; k++) red[threadIdx.x][k] = dbSum[k];
        __syncthreads();
        const int C = p.chanVecs * V;
        for (int c = threadIdx.x; c < C; c += kThreads)
        {
            const int cg = c / V, k = c - cg * V;
            float tot = 0.f;
            for (int j = cg; j < kThreads; j += p.chanVecs) tot += red[j][k];
            p.dbPartial[(int64_t)blockIdx.x * C + c] = tot;
        }
    }
}

// Flat variant for SHORT rows (frames layout: a bias segment is one H*W plane, a few dozen vectors):
// one 1-D grid over all vectors; a lane finds (row, col) of its first vector with one 32-bit division
// and steps to its other UNROLL-1 vectors incrementally, so short rows still fill every lane.
template <class T, int ACT, int G>
__global__ __launch_bounds__(kThreads) void bias_act_flat_kernel(BiasActArgs p)
{
    typedef typename Elem<T>::acc_t A;
    constexpr int V = Elem<T>::kVec;
    const A alpha = (A)p.alpha, gain = (A)p.gain, clamp = (A)p.clamp;
    const uint32_t nvec = (uint32_t)(p.rows * p.rowVecs);
    const uint32_t rowVecs = (uint32_t)p.rowVecs;
    const uint32_t idx0 = blockIdx.x * (uint32_t)(kUnroll * kThreads) + threadIdx.x;
    uint32_t row = idx0 / rowVecs, col = idx0 - row * rowVecs;
    const uint32_t drow = (uint32_t)kThreads / rowVecs, dcol = (uint32_t)kThreads - drow * rowVecs;

    const T* __restrict__ xp  = (const T*)p.x;
    const T* __restrict__ bp  = (const T*)p.b;
    const T* __restrict__ xrp = (const T*)p.xref;
    const T* __restrict__ yrp = (const T*)p.yref;
    const T* __restrict__ dyp = (const T*)p.dy;
    T* __restrict__ yp = (T*)p.y;

    Vec16<T> vx[kUnroll], vxr[kUnroll], vyr[kUnroll], vdy[kUnroll];
    uint32_t rows[kUnroll], cols[kUnroll];
    #pragma unroll
    for (int u = 0; u < kUnroll; u++)
    {
        rows[u] = row; cols[u] = col;
        const uint32_t iv = idx0 + (uint32_t)u * kThreads;
        if (iv < nvec)
        {
            const int64_t e = (int64_t)iv * V;
            vx[u] = load_vec16<T>(xp + e);
            if (G > 0 && ACT == LVG_ACT_SWISH) vxr[u] = load_vec16<T>(xrp + e);
            if (G > 0 && ACT != LVG_ACT_SWISH && (ACT != LVG_ACT_LINEAR || yrp)) vyr[u] = load_vec16<T>(yrp + e);
            if (G == 2) vdy[u] = load_vec16<T>(dyp + e);
        }
        col += dcol; row += drow;
        if (col >= rowVecs) { col -= rowVecs; row++; }
    }
    #pragma unroll
    for (int u = 0; u < kUnroll; u++)
    {
        const uint32_t iv = idx0 + (uint32_t)u * kThreads;
        if (iv >= nvec) continue;
        A bias[V];
        if (p.biasMode == 2)
        {
            Vec16<T> vb = load_vec16<T>(bp + (int64_t)cols[u] * V);
            #pragma unroll
            for (int k = 0; k < V; k++) bias[k] = (A)to_acc(vb.v[k]);
        }
        else
        {
            const A rb = (p.biasMode == 1) ? (A)to_acc(bp[rows[u] % (uint32_t)p.sizeB]) : (A)0;
            #pragma unroll
            for (int k = 0; k < V; k++) bias[k] = rb;
        }
        Vec16<T> out;
        #pragma unroll
        for (int k = 0; k < V; k++)
        {
            const A in  = (A)to_acc(vx[u].v[k]);
            const A xr  = (G > 0 && ACT == LVG_ACT_SWISH) ? (A)to_acc(vxr[u].v[k]) : (A)0;
            const A yr  = (G > 0 && ACT != LVG_ACT_SWISH && (ACT != LVG_ACT_LINEAR || yrp)) ? (A)to_acc(vyr[u].v[k]) : (A)0;
            const A dyv = (G == 2) ? (A)to_acc(vdy[u].v[k]) : (A)1;
            out.v[k] = from_acc<T>(bias_act_elem<A, ACT, G>(in, bias[k], xr, yr, dyv, alpha, gain, clamp));
        }
        store_vec16<T>(yp + (int64_t)iv * V, out);
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
    const A alpha = (A)p.alpha, gain = (A)p.gain, clamp = (A)p.clamp;
    A r;
    if (p.grad == 0)      r = bias_act_elem<A, ACT, 0>(in, bias, xr, yr, dyv, alpha, gain, clamp);
    else if (p.grad == 1) r = bias_act_elem<A, ACT, 1>(in, bias, xr, yr, dyv, alpha, gain, clamp);
    else                  r = bias_act_elem<A, ACT, 2>(in, bias, xr, yr, dyv, alpha, gain, clamp);
    ((T*)p.y)[i] = from_acc<T>(r);
}

template <class T, int ACT, int G>
int launch_vec(const BiasActArgs& p, hipStream_t stream)
{
    // rows are folded over grid y/z (each <= 65535)
    const int64_t rowsY = p.rows < 65535 ? p.rows : 65535;
    const int64_t rowsZ = lvg_ceil_div(p.rows, rowsY);
    LVG_REQUIRE(rowsZ <= 65535, "bias_act: too many bias segments for one launch");
    if (p.rowVecs < 4 * kThreads && p.rows * p.rowVecs < 0xffffffffLL - kUnroll * kThreads)
    {
        // short rows: flat grid (every lane busy)
        const int64_t bx = lvg_ceil_div(p.rows * p.rowVecs, (int64_t)kUnroll * kThreads);
        hipLaunchKernelGGL((bias_act_flat_kernel<T, ACT, G>), dim3((unsigned)bx), dim3(kThreads), 0, stream, p);
        return lvg_check_launch("bias_act_flat_kernel");
    }
    if (p.rowVecs <= kThreads)
    {
        hipLaunchKernelGGL((bias_act_vec_kernel<T, ACT, G, 1>), dim3(1, (unsigned)rowsY, (unsigned)rowsZ), dim3(kThreads), 0, stream, p);
    }
    else
    {
        const int64_t bx = lvg_ceil_div(p.rowVecs, (int64_t)kUnroll * kThreads);
        LVG_REQUIRE(bx <= 0x7fffffffLL, "bias_act: tensor too large for one launch");
        hipLaunchKernelGGL((bias_act_vec_kernel<T, ACT, G, kUnroll>), dim3((unsigned)bx, (unsigned)rowsY, (unsigned)rowsZ), dim3(kThreads), 0, stream, p);
    }
    return lvg_check_launch("bias_act_vec_kernel");
}

template <class T, int ACT>
int launch_act(BiasActArgs& p, bool vecOk, hipStream_t stream)
{
    int64_t done = 0;
    if (vecOk && p.rows > 0 && p.rowVecs > 0)
    {
        int rc;
        if (p.grad == 0)      rc = launch_vec<T, ACT, 0>(p, stream);
        else if (p.grad == 1) rc = launch_vec<T, ACT, 1>(p, stream);
        else                  rc = launch_vec<T, ACT, 2>(p, stream);
        if (rc) return rc;
        done = p.rows * p.rowVecs * Elem<T>::kVec;
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
    p.dbPartial = nullptr; p.chanVecs = 0;

    // Operands each grad form needs (bias_act.py:150,179,198 of the reference pass exactly these).
    if (grad >= 1 && act == LVG_ACT_SWISH) LVG_REQUIRE(xref, "bias_act: swish backward needs xref");
    if (grad >= 1 && act != LVG_ACT_SWISH && act != LVG_ACT_LINEAR) LVG_REQUIRE(yref, "bias_act: backward needs yref");
    if (grad >= 1 && clamp >= 0 && act != LVG_ACT_SWISH) LVG_REQUIRE(yref, "bias_act: clamped backward needs yref");
    if (grad == 2) LVG_REQUIRE(dy, "bias_act: grad 2 needs dy");

    const int V = (dtype == LVG_F32) ? 4 : (dtype == LVG_F64) ? 2 : 8;
    bool vecOk = lvg_aligned16(x) && lvg_aligned16(y) && (!xref || lvg_aligned16(xref)) &&
                 (!yref || lvg_aligned16(yref)) && (!dy || lvg_aligned16(dy));
    p.biasMode = 0;
    p.rows = 1;
    p.rowVecs = n / V;
    if (b)
    {
        if (p.stepB % V == 0 && n % p.stepB == 0) { p.biasMode = 1; p.rows = n / p.stepB; p.rowVecs = p.stepB / V; }
        else if (p.stepB == 1 && p.sizeB % V == 0 && n % p.sizeB == 0 && lvg_aligned16(b)) { p.biasMode = 2; p.rows = n / p.sizeB; p.rowVecs = p.sizeB / V; }
        else vecOk = false;
    }
    else if (p.rowVecs > (int64_t)0x7fffffff * kUnroll)
    {
        // no bias: split a huge stream into rows so the x grid stays in range
        p.rowVecs = 1 << 24; p.rows = (n / V) / p.rowVecs;
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

// Workgroups (= partial-sum rows) of lvg_bias_act_grad_bias for n elements of a channels-last stream; 0: the form does not apply
extern "C" int64_t lvg_bias_act_grad_bias_slots(int64_t n, int channels, int dtype)
{
    if (dtype < LVG_F32 || dtype > LVG_F64 || n <= 0 || channels <= 0) return 0;
    const int V = (dtype == LVG_F32) ? 4 : (dtype == LVG_F64) ? 2 : 8;
    if (channels % V != 0 || kThreads % (channels / V) != 0 || n % channels != 0) return 0;
    const int64_t vecs = n / V;
    if (vecs < 4 * kThreads || vecs > (int64_t)0x7fffffff * kUnroll) return 0;    // short tensors take the plain path + a tensor reduction
    return lvg_ceil_div(vecs, (int64_t)kUnroll * kThreads);
}

extern "C" int lvg_bias_act_grad_bias(const void* dy, const void* xref, const void* yref, void* dx, float* db_partial,
                                      int64_t n, int channels, int dtype, int act, float alpha, float gain, float clamp, void* stream)
{
    LVG_REQUIRE(dy && dx && db_partial, "bias_act_grad_bias: null pointer");
    LVG_REQUIRE(lvg_bias_act_grad_bias_slots(n, channels, dtype) > 0, "bias_act_grad_bias: %lld elements, %d channels, dtype %d: no kernel", (long long)n, channels, dtype);
    LVG_REQUIRE(act != LVG_ACT_SWISH, "bias_act_grad_bias: swish needs the bias in its backward pass -- use lvg_bias_act");
    if (act != LVG_ACT_LINEAR || clamp >= 0) LVG_REQUIRE(yref, "bias_act_grad_bias: backward needs yref");
    LVG_REQUIRE(lvg_aligned16(dy) && lvg_aligned16(dx) && lvg_aligned16(xref) && lvg_aligned16(yref), "bias_act_grad_bias: pointers must be 16-byte aligned");
    const int V = (dtype == LVG_F32) ? 4 : (dtype == LVG_F64) ? 2 : 8;
    BiasActArgs p;
    p.x = dy; p.b = nullptr; p.xref = xref; p.yref = yref; p.dy = nullptr; p.y = dx;
    p.n = n; p.start = 0; p.stepB = 1; p.sizeB = 1; p.grad = 1;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.biasMode = 0; p.rows = 1; p.rowVecs = n / V;
    p.dbPartial = db_partial; p.chanVecs = channels / V;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype)
    {
        case LVG_F32:  return launch_dtype<float>(p, act, true, s);
        case LVG_F16:  return launch_dtype<f16_t>(p, act, true, s);
        case LVG_BF16: return launch_dtype<bf16_t>(p, act, true, s);
        default:       return launch_dtype<double>(p, act, true, s);
    }
}

// ---- plane sums of a contiguous NCHW tensor: out[plane] = sum over the plane's hw elements (float32) --------------------------------
// The bias gradient of filtered_lrelu's backward pass is dx.sum([0, 2, 3]) in the reference (filtered_lrelu.py:254): a tensor
// reduction over channel planes. One workgroup per plane, 16-byte loads, fixed summation order (reproducible); the caller adds the
// N sums of a channel.
namespace {

template <class T, int SQ>
__global__ __launch_bounds__(256) void plane_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int hw, int vec)
{
    // SQ = 1: sum of squares (the mean-square statistic of the generators' input-magnitude EMAs); SQ = 2: max |x| (the scale of a split-precision operand)
    auto term = [](float v) __attribute__((always_inline)) -> float { return SQ == 1 ? v * v : (SQ == 2 ? fabsf(v) : v); };
    auto acc = [](float a_, float b_) __attribute__((always_inline)) -> float { return SQ == 2 ? fmaxf(a_, b_) : a_ + b_; };
    __shared__ float red[4];
    const T* row = x + (int64_t)blockIdx.x * hw;
    float s = 0.f;
    constexpr int V = Elem<T>::kVec;
    if (vec)
    {
        // 16-byte loads over the aligned interior of the plane, element loads for the (< V) elements in front of it and behind it: planes
        // of 94 x 150 or 166 x 278 half-precision pixels are not multiples of 16 bytes, so every second plane starts 8 bytes off -- the
        // all-or-nothing test of the first version sent exactly the large sres planes down the 2-byte path (1.4-1.7 TB/s)
        const uintptr_t addr = reinterpret_cast<uintptr_t>(row);
        int head = (int)(((16 - (addr & 15)) & 15) / sizeof(T));
        if (head > hw) head = hw;
        const T* body = row + head;
        const int nv = (hw - head) / V;
        float s2 = 0.f;
        int i = threadIdx.x;
        for (; i + 256 < nv; i += 512)                                // two independent 16-byte loads in flight per trip
        {
            const Vec16<T> a = load_vec16<T>(body + (int64_t)i * V), b = load_vec16<T>(body + (int64_t)(i + 256) * V);
            #pragma unroll
            for (int e = 0; e < V; e++) { s = acc(s, term((float)to_acc(a.v[e]))); s2 = acc(s2, term((float)to_acc(b.v[e]))); }
        }
        if (i < nv)
        {
            const Vec16<T> a = load_vec16<T>(body + (int64_t)i * V);
            #pragma unroll
            for (int e = 0; e < V; e++) s = acc(s, term((float)to_acc(a.v[e])));
        }
        s = acc(s, s2);
        if ((int)threadIdx.x < head) s = acc(s, term((float)to_acc(row[threadIdx.x])));
        for (int j = head + nv * V + threadIdx.x; j < hw; j += 256) s = acc(s, term((float)to_acc(row[j])));
    }
    else
        for (int j = threadIdx.x; j < hw; j += 256) s = acc(s, term((float)to_acc(row[j])));
    for (int o = 32; o > 0; o >>= 1) s = acc(s, __shfl_xor(s, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc(acc(red[0], red[1]), acc(red[2], red[3]));
}

} // namespace

template <int SQ>
static int plane_sum_launch(const void* x, float* out, int64_t planes, int64_t hw, int dtype, void* stream, const char* what)
{
    LVG_REQUIRE(x && out && planes >= 1 && planes <= 0x7fffffffLL && hw >= 1 && hw <= 0x7fffffffLL, "%s: bad sizes", what);
    LVG_REQUIRE(dtype == LVG_F32 || dtype == LVG_F16 || dtype == LVG_BF16, "%s: float32 / float16 / bfloat16 only (dtype %d)", what, dtype);
    const int esz = dtype == LVG_F32 ? 4 : 2;
    const int vec = (((uintptr_t)x) % esz) == 0 && hw >= 64;          // (element-aligned base: every plane then has a 16-byte-aligned interior)
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LVG_F32)       hipLaunchKernelGGL((plane_sum_kernel<float, SQ>), dim3((unsigned)planes), dim3(256), 0, s, (const float*)x, out, (int)hw, vec);
    else if (dtype == LVG_F16)  hipLaunchKernelGGL((plane_sum_kernel<f16_t, SQ>), dim3((unsigned)planes), dim3(256), 0, s, (const f16_t*)x, out, (int)hw, vec);
    else                        hipLaunchKernelGGL((plane_sum_kernel<bf16_t, SQ>), dim3((unsigned)planes), dim3(256), 0, s, (const bf16_t*)x, out, (int)hw, vec);
    return lvg_check_launch(what);
}

extern "C" int lvg_plane_sum(const void* x, float* out, int64_t planes, int64_t hw, int dtype, void* stream)
{
    return plane_sum_launch<0>(x, out, planes, hw, dtype, stream, "plane_sum");
}

extern "C" int lvg_plane_sum_sq(const void* x, float* out, int64_t planes, int64_t hw, int dtype, void* stream)
{
    return plane_sum_launch<1>(x, out, planes, hw, dtype, stream, "plane_sum_sq");
}

extern "C" int lvg_plane_absmax(const void* x, float* out, int64_t planes, int64_t hw, int dtype, void* stream)
{
    return plane_sum_launch<2>(x, out, planes, hw, dtype, stream, "plane_absmax");
}
