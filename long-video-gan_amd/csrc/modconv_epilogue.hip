// modconv_epilogue.hip -- epilogue of a style-modulated convolution fused with the prologue of the
// next one, as ONE HBM-streaming pass over the activations (gfx950):
//
//     out[f,c,p] = clamp( act( y[f,c,p] * pre[f,c] + b[c] ) * gain, +-clamp ) * post[f,c]
//     msq[f]    += sum_{c,p} clamp(...)^2                   (optional; the input-magnitude statistic)
//
// f = frame (sample x time), c = channel, p = pixel. `pre` is the demodulation coefficient of the
// convolution that produced y, `post` the style modulation of the convolution that consumes out.
// The reference spells this as three elementwise passes and a reduction (model/generator_lres.py:
// 122-123 `output * demodulation`, :570 bias_act, :101-103 `input * style`, :574 magnitude EMA) --
// 7 tensor streams where this kernel moves 2.
//
// The backward kernel recomputes the activation from y and returns, next to dy, the three
// per-(frame, channel) reductions the small tensors need:
//     d_pre[f,c] = sum_p du*y     d_post[f,c] = sum_p dout*clamped     d_sum[f,c] = sum_p du   (db = sum_f d_sum)
// with du = dout * post * [|g| < clamp] * gain * act'(u) and dy = du * pre.
//
// Two kernel families:
//   * channels-last ([f][p][c] in memory, c % vector == 0, c/vector a power of two <= 256): 16-byte
//     lanes; every thread keeps ONE channel-vector for its whole loop, so pre/post/b live in registers
//     and the (f,c) reductions are a block-level LDS transpose plus one atomic per channel;
//   * strided planes (any layout, e.g. NCHW or 3-channel RGB): one wavefront per (f,c) plane.
//
// Roofline: pure stream. Algorithmic bytes per element: forward 2*s, backward 3*s (s = sizeof(T)).

#include "epilogue_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Channels-last kernels. grid = (chunks per frame, frames).

template <class T, int ACT>
__global__ __launch_bounds__(kThreads) void epilogue_cl_fwd_kernel(EpilogueArgs p)
{
    constexpr int V = Elem<T>::kVec;
    const int     cv = p.channels / V;                       // power of two, divides kThreads
    const int64_t f  = blockIdx.y;
    const T* y   = static_cast<const T*>(p.y)   + f * p.frameVecs * V;
    T*       out = static_cast<T*>(p.out)       + f * p.frameVecs * V;
    T*       mid = p.mid ? static_cast<T*>(p.mid) + f * p.frameVecs * V : nullptr;
    ChanVec<T> k;
    load_chan<T>(p, f, (threadIdx.x & (cv - 1)) * V, k);

    const int64_t first = (int64_t)blockIdx.x * p.chunkVecs;
    const int64_t last  = min(first + p.chunkVecs, p.frameVecs);
    float sq = 0.f;
    auto one = [&](const Vec16<T>& in, int64_t i)
    {
        Vec16<T> o, m;
        #pragma unroll
        for (int e = 0; e < V; e++)
        {
            bool inside;
            float g = epi_value<ACT>(to_acc(in.v[e]), k.pre[e], k.b[e], p.alpha, p.gain, p.clamp, inside);
            sq = fmaf(g, g, sq);
            o.v[e] = from_acc<T>(g * k.post[e]);
            m.v[e] = from_acc<T>(g);
        }
        store_vec16<T>(out + i * V, o);
        if (mid) store_vec16<T>(mid + i * V, m);
    };
    // U independent 16-byte loads are issued before any of them is consumed (a per-iteration bounds
    // check would serialise them: one load in flight per lane measured 3.2 TB/s).
    constexpr int U = 4;
    int64_t i = first + threadIdx.x;
    for (; i + (U - 1) * kThreads < last; i += U * kThreads)
    {
        Vec16<T> in[U];
        #pragma unroll
        for (int u = 0; u < U; u++) in[u] = load_vec16<T>(y + (i + u * kThreads) * V);
        #pragma unroll
        for (int u = 0; u < U; u++) one(in[u], i + u * kThreads);
    }
    for (; i < last; i += kThreads) one(load_vec16<T>(y + i * V), i);
    if (p.msq)
    {
        __shared__ float part[kThreads / 64];
        sq = wave_sum(sq);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) p.msq[(int64_t)blockIdx.x * p.frames + f] = (part[0] + part[1]) + (part[2] + part[3]);   // slot = chunk: no atomics
    }
}

template <class T, int ACT>
__global__ __launch_bounds__(kThreads) void epilogue_cl_bwd_kernel(EpilogueArgs p)
{
    constexpr int V = Elem<T>::kVec;
    const int     cv = p.channels / V;
    const int64_t f  = blockIdx.y;
    const T* y    = static_cast<const T*>(p.y)    + f * p.frameVecs * V;
    const T* dout = static_cast<const T*>(p.dout) + f * p.frameVecs * V;
    T*       dy   = static_cast<T*>(p.dy)         + f * p.frameVecs * V;
    const T* dmid = p.dmid ? static_cast<const T*>(p.dmid) + f * p.frameVecs * V : nullptr;
    ChanVec<T> k;
    load_chan<T>(p, f, (threadIdx.x & (cv - 1)) * V, k);

    float aPre[V], aPost[V], aSum[V];
    #pragma unroll
    for (int e = 0; e < V; e++) { aPre[e] = 0.f; aPost[e] = 0.f; aSum[e] = 0.f; }

    const int64_t first = (int64_t)blockIdx.x * p.chunkVecs;
    const int64_t last  = min(first + p.chunkVecs, p.frameVecs);
    auto one = [&](const Vec16<T>& in, const Vec16<T>& go, const Vec16<T>& gm, int64_t i)
    {
        Vec16<T> o;
        #pragma unroll
        for (int e = 0; e < V; e++)
        {
            const float yv = to_acc(in.v[e]), gv = to_acc(go.v[e]);
            const float u = fmaf(yv, k.pre[e], k.b[e]);
            bool inside;
            const float g = epi_value<ACT>(yv, k.pre[e], k.b[e], p.alpha, p.gain, p.clamp, inside);
            // gradient of the value before `post`: through `out` (x post) and, in the dual form, directly through `mid`
            const float gg = dmid ? fmaf(gv, k.post[e], to_acc(gm.v[e])) : gv * k.post[e];
            const float du = inside ? gg * p.gain * act_slope<ACT>(u, p.alpha) : 0.f;
            aPost[e] = fmaf(gv, g, aPost[e]);
            aPre[e]  = fmaf(du, yv, aPre[e]);
            aSum[e] += du;
            o.v[e] = from_acc<T>(du * k.pre[e]);
        }
        store_vec16<T>(dy + i * V, o);
    };
    constexpr int U = 2;                                     // 2 streams x 2 = 4 loads in flight per lane
    int64_t i = first + threadIdx.x;
    for (; i + (U - 1) * kThreads < last; i += U * kThreads)
    {
        Vec16<T> in[U], go[U], gm[U];
        #pragma unroll
        for (int u = 0; u < U; u++)
        {
            in[u] = load_vec16<T>(y + (i + u * kThreads) * V);
            go[u] = load_vec16<T>(dout + (i + u * kThreads) * V);
            if (dmid) gm[u] = load_vec16<T>(dmid + (i + u * kThreads) * V);
        }
        #pragma unroll
        for (int u = 0; u < U; u++) one(in[u], go[u], gm[u], i + u * kThreads);
    }
    for (; i < last; i += kThreads)
    {
        Vec16<T> gm1;
        if (dmid) gm1 = load_vec16<T>(dmid + i * V);
        one(load_vec16<T>(y + i * V), load_vec16<T>(dout + i * V), gm1, i);
    }

    // Block reduction over the threads that share a channel-vector (t, t + cv, t + 2cv, ...):
    // red[q][e][t] is written conflict-free (t fastest); reader j owns channel (j % cv) * V + j / cv.
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
        const int c = j & (cv - 1), e = j / cv;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < rows; r++)
        {
            s0 += red[0][e][c + r * cv];
            s1 += red[1][e][c + r * cv];
            s2 += red[2][e][c + r * cv];
        }
        const int64_t o = ((int64_t)blockIdx.x * p.frames + f) * p.channels + c * V + e;      // slot = chunk: the caller adds the chunks
        if (p.pre)  p.d_pre[o] = s0;
        if (p.post) p.d_post[o] = s1;
        p.d_sum[o] = s2;
    }
}

// ------------------------------------------------------------------------------------------------
// Strided-plane kernels: one wavefront per (frame, channel) plane, 4 planes per block.

template <class T, int ACT>
__global__ __launch_bounds__(kThreads) void epilogue_plane_fwd_kernel(EpilogueArgs p)
{
    const int64_t plane = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float sq = 0.f;
    int64_t f = 0;
    if (plane < p.frames * p.channels)
    {
        f = plane / p.channels;
        const int     c = (int)(plane - f * p.channels);
        const float pre  = p.pre  ? p.pre[plane]  : 1.f;
        const float post = p.post ? p.post[plane] : 1.f;
        const float b    = p.b ? to_acc(static_cast<const T*>(p.b)[c]) : 0.f;
        const int64_t base = f * p.strideF + c * p.strideC;
        const T* y   = static_cast<const T*>(p.y) + base;
        T*       out = static_cast<T*>(p.out) + base;
        for (int i = lane; i < p.pixels; i += 64)
        {
            bool inside;
            const float g = epi_value<ACT>(to_acc(y[i * p.strideP]), pre, b, p.alpha, p.gain, p.clamp, inside);
            sq = fmaf(g, g, sq);
            out[i * p.strideP] = from_acc<T>(g * post);
        }
    }
    if (p.msq)
    {
        sq = wave_sum(sq);
        if (lane == 0 && plane < p.frames * p.channels) p.msq[(plane - f * p.channels) * p.frames + f] = sq;   // slot = channel
    }
}

template <class T, int ACT>
__global__ __launch_bounds__(kThreads) void epilogue_plane_bwd_kernel(EpilogueArgs p)
{
    const int64_t plane = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (plane >= p.frames * p.channels) return;
    const int64_t f = plane / p.channels;
    const int     c = (int)(plane - f * p.channels);
    const float pre  = p.pre  ? p.pre[plane]  : 1.f;
    const float post = p.post ? p.post[plane] : 1.f;
    const float b    = p.b ? to_acc(static_cast<const T*>(p.b)[c]) : 0.f;
    const int64_t base = f * p.strideF + c * p.strideC;
    const T* y    = static_cast<const T*>(p.y) + base;
    const T* dout = static_cast<const T*>(p.dout) + base;
    T*       dy   = static_cast<T*>(p.dy) + base;
    float aPre = 0.f, aPost = 0.f, aSum = 0.f;
    for (int i = lane; i < p.pixels; i += 64)
    {
        const float yv = to_acc(y[i * p.strideP]), gv = to_acc(dout[i * p.strideP]);
        const float u = fmaf(yv, pre, b);
        bool inside;
        const float g = epi_value<ACT>(yv, pre, b, p.alpha, p.gain, p.clamp, inside);
        const float du = inside ? gv * post * p.gain * act_slope<ACT>(u, p.alpha) : 0.f;
        aPost = fmaf(gv, g, aPost);
        aPre  = fmaf(du, yv, aPre);
        aSum += du;
        dy[i * p.strideP] = from_acc<T>(du * pre);
    }
    aPre = wave_sum(aPre); aPost = wave_sum(aPost); aSum = wave_sum(aSum);
    if (lane == 0)
    {
        if (p.pre)  p.d_pre[plane]  = aPre;
        if (p.post) p.d_post[plane] = aPost;
        p.d_sum[plane] = aSum;
    }
}

// ------------------------------------------------------------------------------------------------

template <class T, int ACT>
int launch(EpilogueArgs& p, bool backward, bool channelsLast, hipStream_t stream)
{
    constexpr int V = Elem<T>::kVec;
    const int cv = p.channels / V;
    const bool vec = channelsLast && p.channels % V == 0 && cv <= kThreads && (cv & (cv - 1)) == 0 &&
                     lvg_aligned16(p.y) && lvg_aligned16(backward ? p.dy : p.out) && (!backward || lvg_aligned16(p.dout)) &&
                     lvg_aligned16(p.mid) && lvg_aligned16(p.dmid) && p.frames <= 65535;
    if (vec)
    {
        p.frameVecs = (int64_t)p.pixels * cv;
        p.chunkVecs = epilogue_chunk_vecs(p.frameVecs, p.frames);
        dim3 grid((unsigned)lvg_ceil_div(p.frameVecs, p.chunkVecs), (unsigned)p.frames);
        if (backward) hipLaunchKernelGGL((epilogue_cl_bwd_kernel<T, ACT>), grid, dim3(kThreads), 0, stream, p);
        else          hipLaunchKernelGGL((epilogue_cl_fwd_kernel<T, ACT>), grid, dim3(kThreads), 0, stream, p);
        return lvg_check_launch("modconv_epilogue (channels-last)");
    }
    if (p.mid || p.dmid)
    {
        lvg_set_error("modconv_epilogue: the dual form needs channels-last tensors with a power-of-two number of 16-byte channel vectors");
        return LVG_ERR_UNSUPPORTED;
    }
    const int64_t planes = p.frames * p.channels;
    const int64_t blocks = lvg_ceil_div(planes, kThreads / 64);
    if (blocks > 0x7fffffffLL) { lvg_set_error("modconv_epilogue: %lld planes exceed the grid limit", (long long)planes); return LVG_ERR_UNSUPPORTED; }
    if (backward) hipLaunchKernelGGL((epilogue_plane_bwd_kernel<T, ACT>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p);
    else          hipLaunchKernelGGL((epilogue_plane_fwd_kernel<T, ACT>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p);
    return lvg_check_launch("modconv_epilogue (planes)");
}

template <class T>
int dispatch_act(EpilogueArgs& p, int act, bool backward, bool channelsLast, hipStream_t stream)
{
    switch (act)
    {
    case LVG_ACT_LINEAR: return launch<T, LVG_ACT_LINEAR>(p, backward, channelsLast, stream);
    case LVG_ACT_RELU:   return launch<T, LVG_ACT_RELU>(p, backward, channelsLast, stream);
    case LVG_ACT_LRELU:  return launch<T, LVG_ACT_LRELU>(p, backward, channelsLast, stream);
    default:
        lvg_set_error("modconv_epilogue: activation %d has no fused kernel (linear, relu, lrelu only)", act);
        return LVG_ERR_UNSUPPORTED;
    }
}

int run(EpilogueArgs& p, int dtype, int act, bool backward, int channels_last, void* stream)
{
    LVG_REQUIRE(p.frames >= 0 && p.channels >= 0 && p.pixels >= 0, "modconv_epilogue: negative extent");
    if (p.frames == 0 || p.channels == 0 || p.pixels == 0) return LVG_OK;
    LVG_REQUIRE(p.y && (backward ? (p.dout && p.dy && p.d_sum) : p.out != nullptr), "modconv_epilogue: NULL tensor");
    LVG_REQUIRE(!backward || ((!p.pre || p.d_pre) && (!p.post || p.d_post)), "modconv_epilogue: d_pre/d_post missing");
    LVG_REQUIRE((int64_t)p.pixels * p.channels <= 0x7fffffffLL, "modconv_epilogue: frame too large");
    if (channels_last) { p.strideP = p.channels; p.strideC = 1; }
    else               { p.strideP = 1;          p.strideC = p.pixels; }
    p.strideF = (int64_t)p.pixels * p.channels;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype)
    {
    case LVG_F32:  return dispatch_act<float>(p, act, backward, channels_last != 0, s);
    case LVG_F16:  return dispatch_act<f16_t>(p, act, backward, channels_last != 0, s);
    case LVG_BF16: return dispatch_act<bf16_t>(p, act, backward, channels_last != 0, s);
    default:
        lvg_set_error("modconv_epilogue: dtype %d not supported (f32, f16, bf16)", dtype);
        return LVG_ERR_UNSUPPORTED;
    }
}

} // namespace

// Partial-sum slots of the reductions (see include/lvg_ops.h): the channels-last kernels write one partial per chunk of a frame, the
// plane kernels one mean-square partial per channel and final gradient sums.
extern "C" int lvg_modconv_epilogue_slots(int64_t frames, int channels, int pixels, int channels_last, int dtype, int backward)
{
    if (frames <= 0 || channels <= 0 || pixels <= 0) return 1;
    const int V = dtype == LVG_F32 ? 4 : 8;
    const int cv = channels / V;
    const bool vec = channels_last && channels % V == 0 && cv <= kThreads && (cv & (cv - 1)) == 0 && frames <= 65535;
    if (vec)
    {
        const int64_t frameVecs = (int64_t)pixels * cv;
        return (int)lvg_ceil_div(frameVecs, epilogue_chunk_vecs(frameVecs, frames));
    }
    return backward ? 1 : channels;
}

extern "C" int lvg_modconv_epilogue(const void* y, const float* pre, const void* b, const float* post, void* out, float* msq,
                                    int64_t frames, int channels, int pixels, int channels_last, int dtype, int act,
                                    float alpha, float gain, float clamp, void* stream)
{
    EpilogueArgs p = {};
    p.y = y; p.pre = pre; p.b = b; p.post = post; p.out = out; p.msq = msq;
    p.frames = frames; p.channels = channels; p.pixels = pixels;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    return run(p, dtype, act, false, channels_last, stream);
}

extern "C" int lvg_modconv_epilogue_backward(const void* dout, const void* y, const float* pre, const void* b, const float* post,
                                             void* dy, float* d_pre, float* d_post, float* d_sum,
                                             int64_t frames, int channels, int pixels, int channels_last, int dtype, int act,
                                             float alpha, float gain, float clamp, void* stream)
{
    EpilogueArgs p = {};
    p.y = y; p.pre = pre; p.b = b; p.post = post; p.dout = dout; p.dy = dy;
    p.d_pre = d_pre; p.d_post = d_post; p.d_sum = d_sum;
    p.frames = frames; p.channels = channels; p.pixels = pixels;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    return run(p, dtype, act, true, channels_last, stream);
}

// Dual form: `mid` = the value before `post` as a second output (forward), `dmid` = its gradient as a second input (backward).
// Channels-last tensors only (LVG_ERR_UNSUPPORTED otherwise); mid / dmid may be NULL (= the plain form).
extern "C" int lvg_modconv_epilogue_dual(const void* y, const float* pre, const void* b, const float* post, void* out, void* mid, float* msq,
                                         int64_t frames, int channels, int pixels, int dtype, int act,
                                         float alpha, float gain, float clamp, void* stream)
{
    EpilogueArgs p = {};
    p.y = y; p.pre = pre; p.b = b; p.post = post; p.out = out; p.mid = mid; p.msq = msq;
    p.frames = frames; p.channels = channels; p.pixels = pixels;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    return run(p, dtype, act, false, 1, stream);
}

extern "C" int lvg_modconv_epilogue_dual_backward(const void* dout, const void* dmid, const void* y, const float* pre, const void* b, const float* post,
                                                  void* dy, float* d_pre, float* d_post, float* d_sum,
                                                  int64_t frames, int channels, int pixels, int dtype, int act,
                                                  float alpha, float gain, float clamp, void* stream)
{
    EpilogueArgs p = {};
    p.y = y; p.pre = pre; p.b = b; p.post = post; p.dout = dout; p.dmid = dmid; p.dy = dy;
    p.d_pre = d_pre; p.d_post = d_post; p.d_sum = d_sum;
    p.frames = frames; p.channels = channels; p.pixels = pixels;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    return run(p, dtype, act, true, 1, stream);
}
