// epilogue_common.h -- pieces shared by the modulated-conv epilogue kernels
// (modconv_epilogue.hip: plain epilogue; tapconv_epilogue.hip: temporal-tap gather + epilogue).
#pragma once
#include "lvg_common.h"
#include <algorithm>

namespace {

struct EpilogueArgs
{
    const void*  y;
    const float* pre;      // [frames, channels] or NULL (= 1)
    const void*  b;        // [channels] in T or NULL (= 0)
    const float* post;     // [frames, channels] or NULL (= 1)
    void*        out;
    float*       msq;      // forward: [frames] floats, atomically accumulated (zeroed by the caller); or NULL
    const void*  dout;     // backward only
    void*        dy;
    float*       d_pre;    // [frames, channels], zero-initialised by the caller
    float*       d_post;
    float*       d_sum;
    int64_t      frames;
    int          channels;
    int          pixels;
    int64_t      strideF;  // element strides (plane kernels)
    int64_t      strideC;
    int64_t      strideP;
    int64_t      frameVecs;  // channels-last kernels: 16-byte vectors per frame
    int          chunkVecs;  //                       vectors per block
    float        alpha, gain, clamp;
    // temporal-tap gather (tapconv_epilogue.hip): `y` is then z [frames, pixels, taps*channels]
    const void*  res;       // optional residual added before the activation, layout of `out`
    void*        ysum;      // forward: the gathered sum, saved for the backward pass (or NULL)
    int          taps;      // temporal taps stacked along z's channels (tap-major)
    int          tapCenter; // tap that reads the frame itself
    int64_t      tapShift;  // frames between consecutive time steps (= clips per batch in time-major layout)
    // dual form (modconv_epilogue.hip, channels-last only): the value BEFORE `post` is a second output (the activated tensor that a
    // skip connection reads next to the modulated one), and its gradient a second input of the backward pass
    void*        mid;       // forward: clamp(act(y * pre + b) * gain) in T, or NULL
    const void*  dmid;      // backward: gradient with respect to `mid`, or NULL
};

constexpr int kThreads = 256;

template <int ACT> __device__ __forceinline__ float act_fwd(float u, float alpha)
{
    if (ACT == LVG_ACT_RELU)  return u > 0.f ? u : 0.f;
    if (ACT == LVG_ACT_LRELU) return u > 0.f ? u : u * alpha;
    return u;
}
template <int ACT> __device__ __forceinline__ float act_slope(float u, float alpha)
{
    if (ACT == LVG_ACT_RELU)  return u > 0.f ? 1.f : 0.f;
    if (ACT == LVG_ACT_LRELU) return u > 0.f ? 1.f : alpha;
    return 1.f;
}

// Forward value before `post`, and whether the clamp was hit.
template <int ACT> __device__ __forceinline__ float epi_value(float y, float pre, float b, float alpha, float gain, float clamp, bool& inside)
{
    float g = act_fwd<ACT>(fmaf(y, pre, b), alpha) * gain;
    inside = true;
    if (clamp >= 0.f)
    {
        inside = (g > -clamp && g < clamp);
        if (!inside) g = (g >= 0.f) ? clamp : -clamp;
    }
    return g;
}

__device__ __forceinline__ float wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Per-thread channel vector of the channels-last kernels: one 16-byte vector of channels, fixed for the
// whole loop of a thread, so demodulation / modulation / bias live in registers.

template <class T> struct ChanVec
{
    static constexpr int V = Elem<T>::kVec;
    float pre[V], post[V], b[V];
};

template <class T> __device__ __forceinline__ void load_chan(const EpilogueArgs& p, int64_t f, int c0, ChanVec<T>& k)
{
    constexpr int V = Elem<T>::kVec;
    const int64_t fc = f * p.channels + c0;
    #pragma unroll
    for (int i = 0; i < V; i++)
    {
        k.pre[i]  = p.pre  ? p.pre[fc + i]  : 1.f;
        k.post[i] = p.post ? p.post[fc + i] : 1.f;
        k.b[i]    = p.b    ? to_acc(static_cast<const T*>(p.b)[c0 + i]) : 0.f;
    }
}

// Equal chunks of a frame (multiples of the block size, so a thread keeps its channel vector); few small
// frames are split further so that the grid fills the chip.
static inline int epilogue_chunk_vecs(int64_t frameVecs, int64_t frames)
{
    int64_t chunks = lvg_ceil_div(frameVecs, kThreads * 32);
    if (frames * chunks < 2048)
        chunks = std::max<int64_t>(chunks, std::min<int64_t>(lvg_ceil_div(2048, frames), lvg_ceil_div(frameVecs, kThreads)));
    return (int)(lvg_ceil_div(lvg_ceil_div(frameVecs, chunks), kThreads) * kThreads);
}

} // namespace
