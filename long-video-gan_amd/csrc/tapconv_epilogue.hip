// tapconv_epilogue.hip -- temporal-tap gather fused with the modulated-conv epilogue (gfx950).
//
// A kt x kh x kw convolution over time-major channels-last frames [(T N), H, W, C] is run as ONE 2-D
// convolution whose output channels stack the kt temporal taps, z [(T N), H, W, kt*C] (tap-major), instead
// of kt convolutions whose outputs are accumulated with kt-1 elementwise adds. This kernel performs the
// remaining temporal sum while it applies the epilogue, so the sum never makes an extra HBM round trip:
//
//     ysum[f,p,c] = sum_k z[f + (k - kc) * N, p, k*C + c]            (frames outside the clip contribute 0)
//     out[f,p,c]  = clamp(act(ysum * pre[f,c] + b[c] + res[f,p,c]) * gain) * post[f,c]      (+ msq[f])
//
// The backward kernel consumes dout and the saved ysum and writes the gradient already scattered into the
// tap-stacked layout, dz[f + (k - kc) * N, p, k*C + c] = dy[f,p,c] (zeros where the source frame is outside
// the clip), so the dense backward pass is again ONE convolution. It also emits the per-(frame, channel)
// reductions of modconv_epilogue.hip.
//
// Traffic per output element (s bytes): forward (kt + 1 [+1 ysum] [+1 res])*s, backward (2 + kt)*s, against
// 3*(kt-1)*s for the adds plus 2*s / 3*s for the separate epilogue it replaces. Channels-last only.

#include "epilogue_common.h"

namespace {

template <class T, int ACT, int TAPS>
__global__ __launch_bounds__(kThreads) void tapconv_fwd_kernel(EpilogueArgs p)
{
    constexpr int V = Elem<T>::kVec;
    const int     cv = p.channels / V;                       // power of two, divides kThreads
    const int     cvLog = __ffs(cv) - 1;
    const int64_t f  = blockIdx.y;
    const int64_t zFrame = p.frameVecs * TAPS * V;           // elements per frame of z
    const T* z   = static_cast<const T*>(p.y);
    T*       out = static_cast<T*>(p.out) + f * p.frameVecs * V;
    T*       ys  = p.ysum ? static_cast<T*>(p.ysum) + f * p.frameVecs * V : nullptr;
    const T* res = p.res ? static_cast<const T*>(p.res) + f * p.frameVecs * V : nullptr;
    const int c = threadIdx.x & (cv - 1);
    ChanVec<T> k;
    load_chan<T>(p, f, c * V, k);

    // per tap: base pointer of the source frame (or NULL when it lies outside the clip)
    const T* src[TAPS];
    #pragma unroll
    for (int t = 0; t < TAPS; t++)
    {
        const int64_t fs = f + (int64_t)(t - p.tapCenter) * p.tapShift;
        src[t] = (fs >= 0 && fs < p.frames) ? z + fs * zFrame + ((int64_t)t * cv + c) * V : nullptr;
    }

    const int64_t first = (int64_t)blockIdx.x * p.chunkVecs;
    const int64_t last  = min(first + p.chunkVecs, p.frameVecs);
    float sq = 0.f;
    constexpr int U = 2;                                     // U x TAPS independent loads in flight per lane
    auto finish = [&](float (&acc)[V], int64_t i)
    {
        Vec16<T> o, ysv, rv;
        if (res) rv = load_vec16<T>(res + i * V);
        #pragma unroll
        for (int e = 0; e < V; e++)
        {
            const float bias = k.b[e] + (res ? to_acc(rv.v[e]) : 0.f);
            bool inside;
            const float g = epi_value<ACT>(acc[e], k.pre[e], bias, p.alpha, p.gain, p.clamp, inside);
            sq = fmaf(g, g, sq);
            o.v[e] = from_acc<T>(g * k.post[e]);
            ysv.v[e] = from_acc<T>(acc[e]);
        }
        store_vec16<T>(out + i * V, o);
        if (ys) store_vec16<T>(ys + i * V, ysv);
    };
    auto gather = [&](int64_t i, Vec16<T> (&in)[TAPS])
    {
        const int64_t pix = i >> cvLog;                      // i = pix * cv + c
        #pragma unroll
        for (int t = 0; t < TAPS; t++)
            if (src[t]) in[t] = load_vec16<T>(src[t] + pix * (int64_t)(TAPS * cv) * V);
    };
    auto reduce = [&](const Vec16<T> (&in)[TAPS], float (&acc)[V])
    {
        #pragma unroll
        for (int e = 0; e < V; e++) acc[e] = 0.f;
        #pragma unroll
        for (int t = 0; t < TAPS; t++)
            if (src[t])
            {
                #pragma unroll
                for (int e = 0; e < V; e++) acc[e] += to_acc(in[t].v[e]);
            }
    };
    int64_t i = first + threadIdx.x;
    for (; i + (U - 1) * kThreads < last; i += U * kThreads)
    {
        Vec16<T> in[U][TAPS];
        #pragma unroll
        for (int u = 0; u < U; u++) gather(i + u * kThreads, in[u]);
        #pragma unroll
        for (int u = 0; u < U; u++)
        {
            float acc[V];
            reduce(in[u], acc);
            finish(acc, i + u * kThreads);
        }
    }
    for (; i < last; i += kThreads)
    {
        Vec16<T> in[TAPS];
        float acc[V];
        gather(i, in);
        reduce(in, acc);
        finish(acc, i);
    }
    if (p.msq)
    {
        __shared__ float part[kThreads / 64];
        sq = wave_sum(sq);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) p.msq[(int64_t)blockIdx.x * p.frames + f] = (part[0] + part[1]) + (part[2] + part[3]);   // slot = chunk: no atomics
    }
}

// grid.y runs over the SOURCE frames f in [-(TAPS-1-kc)*N, frames + kc*N): in-range f compute dy[f] and store
// it to its (up to TAPS) destinations; out-of-range f only write the zeros their destinations are owed.
template <class T, int ACT, int TAPS>
__global__ __launch_bounds__(kThreads) void tapconv_bwd_kernel(EpilogueArgs p)
{
    constexpr int V = Elem<T>::kVec;
    const int     cv = p.channels / V;
    const int     cvLog = __ffs(cv) - 1;
    const int64_t f  = (int64_t)blockIdx.y - (int64_t)(TAPS - 1 - p.tapCenter) * p.tapShift;
    const bool    live = (f >= 0 && f < p.frames);
    const int64_t zFrame = p.frameVecs * TAPS * V;
    T* dz = static_cast<T*>(p.dy);
    const int c = threadIdx.x & (cv - 1);

    T* dst[TAPS];
    #pragma unroll
    for (int t = 0; t < TAPS; t++)
    {
        const int64_t fd = f + (int64_t)(t - p.tapCenter) * p.tapShift;
        dst[t] = (fd >= 0 && fd < p.frames) ? dz + fd * zFrame + ((int64_t)t * cv + c) * V : nullptr;
    }
    const int64_t first = (int64_t)blockIdx.x * p.chunkVecs;
    const int64_t last  = min(first + p.chunkVecs, p.frameVecs);
    auto scatter = [&](int64_t i, const Vec16<T>& v)
    {
        const int64_t pix = i >> cvLog;
        #pragma unroll
        for (int t = 0; t < TAPS; t++)
            if (dst[t]) store_vec16<T>(dst[t] + pix * (int64_t)(TAPS * cv) * V, v);
    };

    if (!live)
    {
        Vec16<T> zero;
        #pragma unroll
        for (int e = 0; e < V; e++) zero.v[e] = from_acc<T>(0.f);
        for (int64_t i = first + threadIdx.x; i < last; i += kThreads) scatter(i, zero);
        return;                                              // whole block: no barrier below is skipped partially
    }

    const T* y    = static_cast<const T*>(p.y)    + f * p.frameVecs * V;       // the saved ysum
    const T* dout = static_cast<const T*>(p.dout) + f * p.frameVecs * V;
    const T* res  = p.res ? static_cast<const T*>(p.res) + f * p.frameVecs * V : nullptr;
    ChanVec<T> k;
    load_chan<T>(p, f, c * V, k);
    float aPre[V], aPost[V], aSum[V];
    #pragma unroll
    for (int e = 0; e < V; e++) { aPre[e] = 0.f; aPost[e] = 0.f; aSum[e] = 0.f; }

    auto one = [&](const Vec16<T>& in, const Vec16<T>& go, int64_t i)
    {
        Vec16<T> o, rv;
        if (res) rv = load_vec16<T>(res + i * V);
        #pragma unroll
        for (int e = 0; e < V; e++)
        {
            const float yv = to_acc(in.v[e]), gv = to_acc(go.v[e]);
            const float bias = k.b[e] + (res ? to_acc(rv.v[e]) : 0.f);
            const float u = fmaf(yv, k.pre[e], bias);
            bool inside;
            const float g = epi_value<ACT>(yv, k.pre[e], bias, p.alpha, p.gain, p.clamp, inside);
            const float du = inside ? gv * k.post[e] * p.gain * act_slope<ACT>(u, p.alpha) : 0.f;
            aPost[e] = fmaf(gv, g, aPost[e]);
            aPre[e]  = fmaf(du, yv, aPre[e]);
            aSum[e] += du;
            o.v[e] = from_acc<T>(du * k.pre[e]);
        }
        scatter(i, o);
    };
    constexpr int U = 2;
    int64_t i = first + threadIdx.x;
    for (; i + (U - 1) * kThreads < last; i += U * kThreads)
    {
        Vec16<T> in[U], go[U];
        #pragma unroll
        for (int u = 0; u < U; u++)
        {
            in[u] = load_vec16<T>(y + (i + u * kThreads) * V);
            go[u] = load_vec16<T>(dout + (i + u * kThreads) * V);
        }
        #pragma unroll
        for (int u = 0; u < U; u++) one(in[u], go[u], i + u * kThreads);
    }
    for (; i < last; i += kThreads) one(load_vec16<T>(y + i * V), load_vec16<T>(dout + i * V), i);

    __shared__ float red[3][V][kThreads];
    #pragma unroll
    for (int e = 0; e < V; e++)
    {
        red[0][e][threadIdx.x] = aPre[e];
        red[1][e][threadIdx.x] = aPost[e];
        red[2][e][threadIdx.x] = aSum[e];
    }
    __syncthreads();
    const int rows = kThreads / cv;
    for (int j = threadIdx.x; j < p.channels; j += kThreads)
    {
        const int cc = j & (cv - 1), e = j / cv;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < rows; r++)
        {
            s0 += red[0][e][cc + r * cv];
            s1 += red[1][e][cc + r * cv];
            s2 += red[2][e][cc + r * cv];
        }
        const int64_t o = ((int64_t)blockIdx.x * p.frames + f) * p.channels + cc * V + e;     // slot = chunk: the caller adds the chunks
        if (p.pre)  p.d_pre[o] = s0;
        if (p.post) p.d_post[o] = s1;
        p.d_sum[o] = s2;
    }
}

template <class T, int ACT, int TAPS>
int launch_taps(EpilogueArgs& p, bool backward, hipStream_t stream)
{
    const int64_t gridFrames = backward ? p.frames + (int64_t)(TAPS - 1) * p.tapShift : p.frames;
    if (gridFrames > 65535) { lvg_set_error("tapconv_epilogue: %lld frames exceed grid.y", (long long)gridFrames); return LVG_ERR_UNSUPPORTED; }
    p.chunkVecs = epilogue_chunk_vecs(p.frameVecs, p.frames);
    dim3 grid((unsigned)lvg_ceil_div(p.frameVecs, p.chunkVecs), (unsigned)gridFrames);
    if (backward) hipLaunchKernelGGL((tapconv_bwd_kernel<T, ACT, TAPS>), grid, dim3(kThreads), 0, stream, p);
    else          hipLaunchKernelGGL((tapconv_fwd_kernel<T, ACT, TAPS>), grid, dim3(kThreads), 0, stream, p);
    return lvg_check_launch("tapconv_epilogue");
}

template <class T, int ACT>
int dispatch_taps(EpilogueArgs& p, bool backward, hipStream_t stream)
{
    switch (p.taps)
    {
    case 1: return launch_taps<T, ACT, 1>(p, backward, stream);
    case 3: return launch_taps<T, ACT, 3>(p, backward, stream);
    case 5: return launch_taps<T, ACT, 5>(p, backward, stream);
    default:
        lvg_set_error("tapconv_epilogue: %d temporal taps have no kernel (1, 3, 5)", p.taps);
        return LVG_ERR_UNSUPPORTED;
    }
}

template <class T>
int dispatch_act_taps(EpilogueArgs& p, int act, bool backward, hipStream_t stream)
{
    switch (act)
    {
    case LVG_ACT_LINEAR: return dispatch_taps<T, LVG_ACT_LINEAR>(p, backward, stream);
    case LVG_ACT_RELU:   return dispatch_taps<T, LVG_ACT_RELU>(p, backward, stream);
    case LVG_ACT_LRELU:  return dispatch_taps<T, LVG_ACT_LRELU>(p, backward, stream);
    default:
        lvg_set_error("tapconv_epilogue: activation %d has no fused kernel (linear, relu, lrelu only)", act);
        return LVG_ERR_UNSUPPORTED;
    }
}

int run_taps(EpilogueArgs& p, int dtype, int act, bool backward, void* stream)
{
    LVG_REQUIRE(p.frames >= 0 && p.channels >= 0 && p.pixels >= 0 && p.taps >= 1 && p.tapShift >= 1, "tapconv_epilogue: bad extent");
    if (p.frames == 0 || p.channels == 0 || p.pixels == 0) return LVG_OK;
    LVG_REQUIRE(p.y && (backward ? (p.dout && p.dy && p.d_sum) : p.out != nullptr), "tapconv_epilogue: NULL tensor");
    LVG_REQUIRE(!backward || ((!p.pre || p.d_pre) && (!p.post || p.d_post)), "tapconv_epilogue: d_pre/d_post missing");
    LVG_REQUIRE((int64_t)p.pixels * p.channels * p.taps <= 0x7fffffffLL, "tapconv_epilogue: frame too large");
    p.tapCenter = p.taps / 2;
    const int V = (dtype == LVG_F32) ? 4 : 8;
    const int cv = p.channels / V;
    const bool ok = p.channels % V == 0 && cv >= 1 && cv <= kThreads && (cv & (cv - 1)) == 0 &&
                    lvg_aligned16(p.y) && lvg_aligned16(backward ? p.dy : p.out) && (!backward || lvg_aligned16(p.dout)) &&
                    (!p.res || lvg_aligned16(p.res)) && (!p.ysum || lvg_aligned16(p.ysum));
    if (!ok) { lvg_set_error("tapconv_epilogue: needs channels %% %d == 0, channels/%d a power of two <= 256, 16-byte aligned tensors", V, V); return LVG_ERR_UNSUPPORTED; }
    p.frameVecs = (int64_t)p.pixels * cv;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype)
    {
    case LVG_F32:  return dispatch_act_taps<float>(p, act, backward, s);
    case LVG_F16:  return dispatch_act_taps<f16_t>(p, act, backward, s);
    case LVG_BF16: return dispatch_act_taps<bf16_t>(p, act, backward, s);
    default:
        lvg_set_error("tapconv_epilogue: dtype %d not supported (f32, f16, bf16)", dtype);
        return LVG_ERR_UNSUPPORTED;
    }
}

} // namespace

extern "C" int lvg_tapconv_epilogue_slots(int64_t frames, int channels, int pixels, int dtype)
{
    if (frames <= 0 || channels <= 0 || pixels <= 0) return 1;
    const int V = dtype == LVG_F32 ? 4 : 8;
    const int64_t frameVecs = (int64_t)pixels * (channels / V);
    return (int)lvg_ceil_div(frameVecs, epilogue_chunk_vecs(frameVecs, frames));
}

extern "C" int lvg_tapconv_epilogue(const void* z, const float* pre, const void* b, const void* res, const float* post,
                                    void* out, void* ysum, float* msq,
                                    int64_t frames, int channels, int pixels, int taps, int64_t tap_shift,
                                    int dtype, int act, float alpha, float gain, float clamp, void* stream)
{
    EpilogueArgs p = {};
    p.y = z; p.pre = pre; p.b = b; p.res = res; p.post = post; p.out = out; p.ysum = ysum; p.msq = msq;
    p.frames = frames; p.channels = channels; p.pixels = pixels; p.taps = taps; p.tapShift = tap_shift;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    return run_taps(p, dtype, act, false, stream);
}

extern "C" int lvg_tapconv_epilogue_backward(const void* dout, const void* ysum, const float* pre, const void* b, const void* res,
                                             const float* post, void* dz, float* d_pre, float* d_post, float* d_sum,
                                             int64_t frames, int channels, int pixels, int taps, int64_t tap_shift,
                                             int dtype, int act, float alpha, float gain, float clamp, void* stream)
{
    EpilogueArgs p = {};
    p.y = ysum; p.pre = pre; p.b = b; p.res = res; p.post = post; p.dout = dout; p.dy = dz;
    p.d_pre = d_pre; p.d_post = d_post; p.d_sum = d_sum;
    p.frames = frames; p.channels = channels; p.pixels = pixels; p.taps = taps; p.tapShift = tap_shift;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    return run_taps(p, dtype, act, true, stream);
}
