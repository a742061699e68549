// video_io.hip -- the two ends of the pixel path, fused into one HBM pass each (gfx950).
//
//   lvg_video_to_uint8:   network output [N, C, T, H, W] (float32 / float16 / bfloat16, values in [-1, 1])
//                         -> display bytes [N, T, H, W, C] uint8,  (x * 127.5 + 128).clamp(0, 255).to(uint8)
//                         (reference utils.py:163 write_video_grid / :203 save_image_grid, followed by the
//                         "c h w -> h w c" rearrangement of :171 / :209 -- three tensor passes there)
//   lvg_video_from_uint8: decoded frames [N, T, H, W, C] uint8 -> network input [N, C, T, H, W],
//                         2 * float(x) / 255 - 1, optional horizontal flip per sample
//                         (reference dataset.py:81-83 read_frame and :93-94 x_flip -- per-frame CPU work there)
//
// Both are pure streams: one thread owns 4 consecutive pixels of one row, reads / writes the C planes with 16-byte
// (f32) or 8-byte (16-bit) accesses and the interleaved bytes as 4 * C contiguous bytes. Arithmetic is done exactly as
// the reference spells it in float32 (separate multiply and add, no fused multiply-add; IEEE division), so the
// bytes / floats are bit-identical to the reference's.

#include "lvg_common.h"
#include <algorithm>

namespace {

struct VideoArgs
{
    const void* src;
    void*       dst;
    const uint8_t* flip;      // from_uint8: per-sample flags (1 = mirror x) or NULL
    int64_t     planes;       // N * T
    int         C, T, H, W;
    int64_t     quads;        // N * T * H * (W / 4)
};

constexpr int kVThreads = 256;

template <class T, int C>
__global__ __launch_bounds__(kVThreads) void video_to_uint8_kernel(VideoArgs p)
{
    const int wq = p.W >> 2;
    const int64_t hw = (int64_t)p.H * p.W;
    for (int64_t q = (int64_t)blockIdx.x * kVThreads + threadIdx.x; q < p.quads; q += (int64_t)gridDim.x * kVThreads)
    {
        const int64_t row = q / wq;                       // (n, t, h)
        const int x0 = (int)(q - row * wq) * 4;
        const int64_t nt = row / p.H;
        const int h = (int)(row - nt * p.H);
        const int64_t n = nt / p.T;
        const int t = (int)(nt - n * p.T);
        const T* src = static_cast<const T*>(p.src) + ((n * C) * p.T + t) * hw + (int64_t)h * p.W + x0;
        uint8_t* dst = static_cast<uint8_t*>(p.dst) + ((nt * p.H + h) * p.W + x0) * C;
        // 4 pixels x C channels = C dwords of interleaved bytes (byte k of the group = pixel k / C, channel k % C)
        uint32_t words[4] = {0, 0, 0, 0};
        #pragma unroll
        for (int c = 0; c < C; c++)
        {
            const T* plane = src + (int64_t)c * p.T * hw;
            #pragma unroll
            for (int i = 0; i < 4; i++)
            {
                float v = __fadd_rn(__fmul_rn(to_acc(plane[i]), 127.5f), 128.0f);
                v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);          // a NaN passes both tests and becomes 0 below
                const uint32_t b = v == v ? (uint32_t)(int)v : 0u;
                const int k = i * C + c;
                words[k >> 2] |= b << ((k & 3) * 8);
            }
        }
        uint32_t* dw = reinterpret_cast<uint32_t*>(dst);              // (x0 * C) is a multiple of 4: dword aligned
        #pragma unroll
        for (int c = 0; c < C; c++) dw[c] = words[c];
    }
}

template <class T, int C>
__global__ __launch_bounds__(kVThreads) void video_from_uint8_kernel(VideoArgs p)
{
    const int wq = p.W >> 2;
    const int64_t hw = (int64_t)p.H * p.W;
    for (int64_t q = (int64_t)blockIdx.x * kVThreads + threadIdx.x; q < p.quads; q += (int64_t)gridDim.x * kVThreads)
    {
        const int64_t row = q / wq;
        const int x0 = (int)(q - row * wq) * 4;
        const int64_t nt = row / p.H;
        const int h = (int)(row - nt * p.H);
        const int64_t n = nt / p.T;
        const int t = (int)(nt - n * p.T);
        const bool mirror = p.flip && p.flip[n];
        const uint8_t* src = static_cast<const uint8_t*>(p.src) + (nt * p.H + h) * (int64_t)p.W * C;
        T* dst = static_cast<T*>(p.dst) + ((n * C) * p.T + t) * hw + (int64_t)h * p.W + x0;
        // the 4 source pixels are contiguous either way (mirrored: the group W - 4 - x0 .. W - 1 - x0, read backwards)
        const int xg = mirror ? p.W - 4 - x0 : x0;
        uint32_t words[4];
        const uint32_t* sw = reinterpret_cast<const uint32_t*>(src + (int64_t)xg * C);
        #pragma unroll
        for (int c = 0; c < C; c++) words[c] = sw[c];
        #pragma unroll
        for (int c = 0; c < C; c++)
        {
            T* plane = dst + (int64_t)c * p.T * hw;
            #pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int k = (mirror ? 3 - i : i) * C + c;
                const float b = (float)((words[k >> 2] >> ((k & 3) * 8)) & 0xffu);
                plane[i] = from_acc<T>(__fsub_rn(__fdiv_rn(__fmul_rn(2.0f, b), 255.0f), 1.0f));
            }
        }
    }
}

template <class T, int C> int launch_video_c(bool to_bytes, const VideoArgs& a, hipStream_t stream)
{
    const int64_t blocks = std::min<int64_t>(lvg_ceil_div(a.quads, kVThreads), 256 * 32);
    if (to_bytes) hipLaunchKernelGGL((video_to_uint8_kernel<T, C>), dim3((unsigned)blocks), dim3(kVThreads), 0, stream, a);
    else          hipLaunchKernelGGL((video_from_uint8_kernel<T, C>), dim3((unsigned)blocks), dim3(kVThreads), 0, stream, a);
    return lvg_check_launch(to_bytes ? "video_to_uint8" : "video_from_uint8");
}

template <class T> int launch_video(bool to_bytes, const VideoArgs& a, hipStream_t stream)
{
    switch (a.C)
    {
    case 1:  return launch_video_c<T, 1>(to_bytes, a, stream);
    case 2:  return launch_video_c<T, 2>(to_bytes, a, stream);
    case 3:  return launch_video_c<T, 3>(to_bytes, a, stream);
    default: return launch_video_c<T, 4>(to_bytes, a, stream);
    }
}

int video_dispatch(bool to_bytes, const void* src, void* dst, const uint8_t* flip, int64_t n, int c, int t, int h, int w, int dtype, void* stream)
{
    const char* what = to_bytes ? "video_to_uint8" : "video_from_uint8";
    LVG_REQUIRE(n >= 0 && c >= 1 && c <= 4 && t >= 1 && h >= 1 && w >= 4 && w % 4 == 0, "%s: need 1..4 channels and a width that is a multiple of 4 (got C=%d W=%d)", what, c, w);
    LVG_REQUIRE(dtype == LVG_F32 || dtype == LVG_F16 || dtype == LVG_BF16, "%s: float32 / float16 / bfloat16 only", what);
    if (n == 0) return LVG_OK;
    VideoArgs a;
    a.src = src; a.dst = dst; a.flip = flip;
    a.planes = n * t; a.C = c; a.T = t; a.H = h; a.W = w;
    a.quads = n * t * (int64_t)h * (w / 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype)
    {
    case LVG_F32:  return launch_video<float>(to_bytes, a, s);
    case LVG_F16:  return launch_video<f16_t>(to_bytes, a, s);
    default:       return launch_video<bf16_t>(to_bytes, a, s);
    }
}

} // namespace

extern "C" int lvg_video_to_uint8(const void* video, void* bytes, int64_t n, int c, int t, int h, int w, int dtype, void* stream)
{
    return video_dispatch(true, video, bytes, nullptr, n, c, t, h, w, dtype, stream);
}

extern "C" int lvg_video_from_uint8(const void* bytes, void* video, const uint8_t* flip, int64_t n, int c, int t, int h, int w, int dtype, void* stream)
{
    return video_dispatch(false, bytes, video, flip, n, c, t, h, w, dtype, stream);
}
