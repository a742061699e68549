// modconv2d_layout.hip -- prologue / epilogue of the 2-D style-modulated convolution of the super-resolution
// generator (reference model/generator_sres.py:24-67, modulated_conv2d), fused with the layout change that puts
// the dense contraction on MIOpen's channels-last MFMA implicit-GEMM kernels.
//
// The reference multiplies the style into per-sample weights and runs a grouped convolution (groups = batch).
// Here the style multiplies the ACTIVATIONS and the demodulation the convolution OUTPUT, around one dense
// convolution with the shared weight -- algebraically the same -- and because the fused filtered_lrelu kernels
// work on NCHW planes while the fast convolution kernels want NHWC, both multiplications are done by the two
// transposing kernels below, one pass each, instead of  x * style  ->  cat  ->  to(channels_last)  and
// conv  ->  y * demod  ->  contiguous():
//
//   nchw_to_nhwc:  dst[n, p, c] = src[n, c, p] * scale[n, c]      src = two NCHW tensors concatenated along c
//                  (previous layer's output and the conditioning frames), channels >= cA + cB zero-filled
//                  (the channel count is padded to a multiple of 8 for the 16-byte NHWC vectors of the conv);
//                  optional  partial[n, tile, c] = sum_{p in tile} src[n, c, p] * oth[n, p, c]   (backward of the
//                  epilogue: d demod)
//   nhwc_to_nchw:  dst[n, c, p] = src[n, p, c] * scale[n, c]      c < cDst (padding channels dropped);
//                  optional  partial[n, tile, c] = sum_{p in tile} src[n, p, c] * oth[n, c, p], oth = two NCHW
//                  tensors concatenated along c (backward of the prologue: d style)
//
// p = y * W + x. 16-bit element types only (the float32 layers of the network keep the NCHW path). One workgroup
// = one 64 (channels) x 64 (pixels) tile staged through LDS: 16-byte global accesses on both sides. The
// reductions are written as per-tile partial sums in a fixed order (no atomics): results are reproducible run
// to run; the host sums the tiles.
// Roofline: HBM stream, 2 * N * C * P * 2 bytes (+ the same again for `oth` when reducing).

#include "lvg_common.h"

namespace {

constexpr int kTile = 64;
constexpr int kLdsStride = 72;        // halves per LDS row (16-byte aligned rows)
// Element (row r, column q) lives at column  (q & 7) | 8 * ((q >> 3) ^ ((r >> 3) & 7)):  the 8-element segments of a row are
// XOR-permuted by the row's group index, so a column walk down 8 row groups (phase 2) hits 8 different banks while the
// 16-byte row vectors of phase 1 stay contiguous and aligned.
__device__ __forceinline__ int swz(int r, int q) { return (q & 7) | (((q >> 3) ^ ((r >> 3) & 7)) << 3); }

struct LayoutArgs
{
    const void* srcA; const void* srcB;   // nchw_to_nhwc: NCHW sources; nhwc_to_nchw: the NHWC source in srcA
    const void* othA; const void* othB;   // reduction partner (layout opposite to the source), or null
    void*       dst;
    const float* scale;                    // [n, cA + cB] (nchw_to_nhwc) / [n, cDst] (nhwc_to_nchw) or null = 1
    float*      partial;                   // [n, tiles, cRed] or null
    int n, hw;
    int cA, cB;                            // channels of the two NCHW tensors (sources or reduction partners); cB may be 0
    int cNhwc;                             // channels (= pixel stride) of the NHWC tensor (source or destination)
    int cDst;                              // nhwc_to_nchw: channels written
    int cOth;                              // nchw_to_nhwc: channels (= pixel stride) of the NHWC reduction partner
    int vecP;                              // 1: pixel rows of the NCHW tensors may be accessed as 16-byte vectors
    // nchw_to_nhwc into a LARGER frame (the explicitly zero-padded frames of conv2d_igemm.hip / conv2d_wgrad.hip): source pixel
    // p = y * srcW + x goes to pixel (y + offY) * dstW + (x + offX) of a frame of dstHW pixels; srcW == 0: same frame (pixel p).
    // The border is NOT written here (the caller zero-fills the tensor once).
    int srcW, dstW, dstHW, offY, offX;
};

template <class T> __device__ __forceinline__ float ld1(const T* p) { return (float)to_acc(*p); }

// element (c, p) of the channel-concatenated NCHW pair, 0 outside
template <class T>
__device__ __forceinline__ const T* nchw_ptr(const void* a, const void* b, int cA, int cB, int64_t n, int c, int hw)
{
    if (c < cA) return (const T*)a + (n * cA + c) * (int64_t)hw;
    if (c < cA + cB) return (const T*)b + (n * cB + (c - cA)) * (int64_t)hw;
    return nullptr;
}

template <class T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(LayoutArgs a)
{
    __shared__ __attribute__((aligned(16))) T tile[kTile * kLdsStride];     // [c][p], values BEFORE scaling
    __shared__ float red[32 * kTile];
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * kTile, c0 = blockIdx.y * kTile;
    const int64_t n = blockIdx.z;
    const int cSrc = a.cA + a.cB;
    // phase 1: rows = channels, 8 threads x 8 pixels per row
    #pragma unroll
    for (int v = 0; v < 2; v++)
    {
        const int idx = tid + 256 * v, c = idx >> 3, seg = idx & 7;
        const int cc = c0 + c, p = p0 + 8 * seg;
        Vec16<T> val;
        #pragma unroll
        for (int e = 0; e < 8; e++) val.v[e] = from_acc<T>(0.0f);
        const T* row = nchw_ptr<T>(a.srcA, a.srcB, a.cA, a.cB, n, cc, a.hw);
        if (row)
        {
            if (a.vecP && p + 8 <= a.hw) val = load_vec16<T>(row + p);
            else
            {
                #pragma unroll
                for (int e = 0; e < 8; e++) if (p + e < a.hw) val.v[e] = row[p + e];
            }
        }
        store_vec16<T>(tile + c * kLdsStride + swz(c, 8 * seg), val);
    }
    __syncthreads();
    // phase 2: 8 threads x 8 channels per pixel
    float dot[8];
    #pragma unroll
    for (int j = 0; j < 8; j++) dot[j] = 0.0f;
    #pragma unroll
    for (int v = 0; v < 2; v++)
    {
        const int idx = tid + 256 * v, pl = idx >> 3, cg = idx & 7;
        const int p = p0 + pl, cc = c0 + 8 * cg;
        if (p < a.hw && cc < a.cNhwc)
        {
            float x[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) x[j] = (float)to_acc(tile[(8 * cg + j) * kLdsStride + swz(8 * cg + j, pl)]);
            Vec16<T> out;
            #pragma unroll
            for (int j = 0; j < 8; j++)
            {
                const float s = (a.scale && cc + j < cSrc) ? a.scale[n * cSrc + cc + j] : 1.0f;
                out.v[j] = from_acc<T>(x[j] * s);
            }
            int64_t dp = n * a.hw + p;
            if (a.srcW > 0)
            {
                const int y = p / a.srcW, x = p - y * a.srcW;
                dp = n * a.dstHW + (int64_t)(y + a.offY) * a.dstW + (x + a.offX);
            }
            store_vec16<T>((T*)a.dst + dp * (int64_t)a.cNhwc + cc, out);
            if (a.partial && cc < a.cOth)
            {
                const Vec16<T> o = load_vec16<T>((const T*)a.othA + (n * a.hw + p) * (int64_t)a.cOth + cc);
                #pragma unroll
                for (int j = 0; j < 8; j++) dot[j] += x[j] * (float)to_acc(o.v[j]);
            }
        }
    }
    if (a.partial)
    {
        // this thread: channels 8 cg .. 8 cg + 7, pixels (tid >> 3) and (tid >> 3) + 32 -> red[pixel slot][channel]
        const int cg = tid & 7, slot = tid >> 3;
        #pragma unroll
        for (int j = 0; j < 8; j++) red[slot * kTile + 8 * cg + j] = dot[j];
        __syncthreads();
        if (tid < kTile && c0 + tid < cSrc)
        {
            float s = 0.0f;
            for (int k = 0; k < 32; k++) s += red[k * kTile + tid];            // fixed order
            a.partial[(n * gridDim.x + blockIdx.x) * (int64_t)cSrc + c0 + tid] = s;
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(LayoutArgs a)
{
    __shared__ __attribute__((aligned(16))) T tile[kTile * kLdsStride];     // [p][c], values BEFORE scaling
    __shared__ float red[8 * kTile];
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * kTile, c0 = blockIdx.y * kTile;
    const int64_t n = blockIdx.z;
    const int cRed = a.cA + a.cB;
    // phase 1: rows = pixels, 8 threads x 8 channels per pixel
    #pragma unroll
    for (int v = 0; v < 2; v++)
    {
        const int idx = tid + 256 * v, pl = idx >> 3, cg = idx & 7;
        const int p = p0 + pl, cc = c0 + 8 * cg;
        Vec16<T> val;
        #pragma unroll
        for (int e = 0; e < 8; e++) val.v[e] = from_acc<T>(0.0f);
        if (p < a.hw && cc < a.cNhwc) val = load_vec16<T>((const T*)a.srcA + (n * a.hw + p) * (int64_t)a.cNhwc + cc);
        store_vec16<T>(tile + pl * kLdsStride + swz(pl, 8 * cg), val);
    }
    __syncthreads();
    // phase 2: 8 threads x 8 pixels per channel
    #pragma unroll
    for (int v = 0; v < 2; v++)
    {
        const int idx = tid + 256 * v, c = idx >> 3, seg = idx & 7;
        const int cc = c0 + c, p = p0 + 8 * seg;
        float x[8];
        #pragma unroll
        for (int j = 0; j < 8; j++) x[j] = (float)to_acc(tile[(8 * seg + j) * kLdsStride + swz(8 * seg + j, c)]);
        if (cc < a.cDst && p < a.hw)
        {
            const float s = a.scale ? a.scale[n * a.cDst + cc] : 1.0f;
            T* row = (T*)a.dst + (n * a.cDst + cc) * (int64_t)a.hw;
            if (a.vecP && p + 8 <= a.hw)
            {
                Vec16<T> out;
                #pragma unroll
                for (int j = 0; j < 8; j++) out.v[j] = from_acc<T>(x[j] * s);
                store_vec16<T>(row + p, out);
            }
            else
            {
                #pragma unroll
                for (int j = 0; j < 8; j++) if (p + j < a.hw) row[p + j] = from_acc<T>(x[j] * s);
            }
        }
        if (a.partial)
        {
            float d = 0.0f;
            const T* orow = nchw_ptr<T>(a.othA, a.othB, a.cA, a.cB, n, cc, a.hw);
            if (orow && p < a.hw)
            {
                if (a.vecP && p + 8 <= a.hw)
                {
                    const Vec16<T> o = load_vec16<T>(orow + p);
                    #pragma unroll
                    for (int j = 0; j < 8; j++) d += x[j] * (float)to_acc(o.v[j]);
                }
                else
                {
                    #pragma unroll
                    for (int j = 0; j < 8; j++) if (p + j < a.hw) d += x[j] * (float)to_acc(orow[p + j]);
                }
            }
            // v = 0: channels 0..31 of the tile, v = 1: channels 32..63; 8 pixel segments per channel
            red[seg * kTile + c] = d;
        }
    }
    if (a.partial)
    {
        __syncthreads();
        if (tid < kTile && c0 + tid < cRed)
        {
            float s = 0.0f;
            for (int k = 0; k < 8; k++) s += red[k * kTile + tid];             // fixed order
            a.partial[(n * gridDim.x + blockIdx.x) * (int64_t)cRed + c0 + tid] = s;
        }
    }
}

int check_common(const char* what, int64_t n, int64_t hw, int dtype)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "%s: float16 / bfloat16 only (dtype %d)", what, dtype);
    LVG_REQUIRE(n >= 1 && n <= 65535 && hw >= 1 && hw <= 0x3fffffffLL, "%s: batch must be 1..65535 and the plane at most 2^30 pixels", what);
    return LVG_OK;
}

} // namespace

static int nchw_to_nhwc_launch(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                               int64_t n, int64_t hw, int c_a, int c_b, int c_dst, int c_oth, int dtype, void* stream,
                               int src_w, int dst_h, int dst_w, int off_y, int off_x);

extern "C" int lvg_modconv2d_nchw_to_nhwc(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                                          int64_t n, int64_t hw, int c_a, int c_b, int c_dst, int c_oth, int dtype, void* stream)
{
    return nchw_to_nhwc_launch(src_a, src_b, scale, oth, dst, partial, n, hw, c_a, c_b, c_dst, c_oth, dtype, stream, 0, 0, 0, 0, 0);
}

// The same pass writing into the interior of a larger channels-last frame [n][dst_h][dst_w][c_dst] at (off_y, off_x): source planes are
// src_h x src_w (hw = src_h * src_w). The caller zero-fills `dst` beforehand (border pixels and nothing else keep that zero). `oth` /
// `partial` (the reduction partner, dense frames of hw pixels) as in lvg_modconv2d_nchw_to_nhwc.
extern "C" int lvg_modconv2d_nchw_to_nhwc_padded(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                                                 int64_t n, int src_h, int src_w, int c_a, int c_b, int c_dst, int c_oth,
                                                 int dst_h, int dst_w, int off_y, int off_x, int dtype, void* stream)
{
    LVG_REQUIRE(src_h >= 1 && src_w >= 1 && off_y >= 0 && off_x >= 0 && dst_h >= src_h + off_y && dst_w >= src_w + off_x,
                "modconv2d_nchw_to_nhwc_padded: the source plane must fit inside the destination frame");
    LVG_REQUIRE((int64_t)dst_h * dst_w <= 0x3fffffffLL, "modconv2d_nchw_to_nhwc_padded: destination frame too large");
    return nchw_to_nhwc_launch(src_a, src_b, scale, oth, dst, partial, n, (int64_t)src_h * src_w, c_a, c_b, c_dst, c_oth, dtype, stream,
                               src_w, dst_h, dst_w, off_y, off_x);
}

static int nchw_to_nhwc_launch(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                               int64_t n, int64_t hw, int c_a, int c_b, int c_dst, int c_oth, int dtype, void* stream,
                               int src_w, int dst_h, int dst_w, int off_y, int off_x)
{
    if (int rc = check_common("modconv2d_nchw_to_nhwc", n, hw, dtype)) return rc;
    LVG_REQUIRE(src_a && dst && c_a >= 1 && c_b >= 0 && (c_b == 0 || src_b), "modconv2d_nchw_to_nhwc: bad sources");
    LVG_REQUIRE(c_dst >= c_a + c_b && c_dst % 8 == 0, "modconv2d_nchw_to_nhwc: the NHWC channel count must be a multiple of 8 covering the sources");
    LVG_REQUIRE((partial == nullptr) == (oth == nullptr), "modconv2d_nchw_to_nhwc: partial and oth go together");
    LVG_REQUIRE(!oth || (c_oth % 8 == 0 && c_oth >= c_a + c_b), "modconv2d_nchw_to_nhwc: bad reduction partner");
    LVG_REQUIRE(lvg_aligned16(dst) && (!oth || lvg_aligned16(oth)), "modconv2d_nchw_to_nhwc: NHWC tensors must be 16-byte aligned");
    LayoutArgs a = {};
    a.srcA = src_a; a.srcB = src_b; a.othA = oth; a.dst = dst; a.scale = scale; a.partial = partial;
    a.n = (int)n; a.hw = (int)hw; a.cA = c_a; a.cB = c_b; a.cNhwc = c_dst; a.cOth = c_oth;
    a.srcW = src_w; a.dstW = dst_w; a.dstHW = dst_h * dst_w; a.offY = off_y; a.offX = off_x;
    a.vecP = (hw % 8 == 0) && lvg_aligned16(src_a) && (!src_b || lvg_aligned16(src_b));
    dim3 grid((unsigned)lvg_ceil_div(hw, kTile), (unsigned)lvg_ceil_div(c_dst, kTile), (unsigned)n);
    if (dtype == LVG_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else                  hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("modconv2d_nchw_to_nhwc");
}

extern "C" int lvg_modconv2d_nhwc_to_nchw(const void* src, const float* scale, const void* oth_a, const void* oth_b, void* dst, float* partial,
                                          int64_t n, int64_t hw, int c_src, int c_dst, int c_a, int c_b, int dtype, void* stream)
{
    if (int rc = check_common("modconv2d_nhwc_to_nchw", n, hw, dtype)) return rc;
    LVG_REQUIRE(src && dst && c_src % 8 == 0 && c_dst >= 1 && c_dst <= c_src, "modconv2d_nhwc_to_nchw: bad channel counts");
    LVG_REQUIRE((partial == nullptr) == (oth_a == nullptr), "modconv2d_nhwc_to_nchw: partial and oth go together");
    LVG_REQUIRE(!oth_a || (c_a >= 1 && c_b >= 0 && (c_b == 0 || oth_b) && c_a + c_b <= c_src), "modconv2d_nhwc_to_nchw: bad reduction partner");
    LVG_REQUIRE(lvg_aligned16(src), "modconv2d_nhwc_to_nchw: the NHWC tensor must be 16-byte aligned");
    LayoutArgs a = {};
    a.srcA = src; a.othA = oth_a; a.othB = oth_b; a.dst = dst; a.scale = scale; a.partial = partial;
    a.n = (int)n; a.hw = (int)hw; a.cA = oth_a ? c_a : 0; a.cB = oth_a ? c_b : 0; a.cNhwc = c_src; a.cDst = c_dst;
    a.vecP = (hw % 8 == 0) && lvg_aligned16(dst) && (!oth_a || lvg_aligned16(oth_a)) && (!oth_b || lvg_aligned16(oth_b));
    const int cCover = (oth_a && c_a + c_b > c_dst) ? c_a + c_b : c_dst;
    dim3 grid((unsigned)lvg_ceil_div(hw, kTile), (unsigned)lvg_ceil_div(cCover, kTile), (unsigned)n);
    if (dtype == LVG_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else                  hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("modconv2d_nhwc_to_nchw");
}
