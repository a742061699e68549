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
    int vecP;                              // pixel rows of the NCHW tensors: 2 = 16-byte vectors, 1 = dwords (even plane size), 0 = elements (load_row8)
    int othPlanar;                         // nchw_to_nhwc: the reduction partner is NCHW like the source ([n, cOth, hw]) instead of NHWC
    // nchw_to_nhwc into a LARGER frame (the explicitly zero-padded frames of conv2d_igemm.hip / conv2d_wgrad.hip): source pixel
    // p = y * srcW + x goes to pixel (y + offY) * dstW + (x + offX) of a frame of dstHW pixels; srcW == 0: same frame (pixel p).
    // The border is NOT written here (the caller zero-fills the tensor once).
    int srcW, dstW, dstHW, offY, offX;
    float invW;                            // 1 / srcW (0: divide)
};

template <class T> __device__ __forceinline__ float ld1(const T* p) { return (float)to_acc(*p); }

// 8 consecutive pixels of an NCHW row from pixel p (a multiple of 8). level 2: one 16-byte access (hw % 8 == 0, aligned bases); level 1: four 4-byte accesses
// (hw even: every row starts on a dword -- all plane sizes of the super-resolution networks, 38 x 38 .. 278 x 166, are even but not multiples of 8: round 6);
// level 0: element by element. Pixels at or past hw read as 0 / are not written.
template <class T> __device__ __forceinline__ Vec16<T> load_row8(const T* row, int p, int hw, int level)
{
    Vec16<T> val;
    if (level == 2 && p + 8 <= hw) return load_vec16<T>(row + p);
    #pragma unroll
    for (int e = 0; e < 8; e++) val.v[e] = from_acc<T>(0.0f);
    if (level >= 1)
    {
        #pragma unroll
        for (int e = 0; e < 8; e += 2)
            if (p + e + 2 <= hw)
            {
                uint32_t w;
                w = *reinterpret_cast<const uint32_t*>(row + p + e);
                __builtin_memcpy(&val.v[e], &w, 4);
            }
    }
    else
    {
        #pragma unroll
        for (int e = 0; e < 8; e++) if (p + e < hw) val.v[e] = row[p + e];
    }
    return val;
}
template <class T> __device__ __forceinline__ void store_row8(T* row, int p, int hw, int level, const Vec16<T>& val)
{
    if (level == 2 && p + 8 <= hw) { store_vec16<T>(row + p, val); return; }
    if (level >= 1)
    {
        #pragma unroll
        for (int e = 0; e < 8; e += 2)
            if (p + e + 2 <= hw)
            {
                uint32_t w;
                __builtin_memcpy(&w, &val.v[e], 4);
                *reinterpret_cast<uint32_t*>(row + p + e) = w;
            }
    }
    else
    {
        #pragma unroll
        for (int e = 0; e < 8; e++) if (p + e < hw) row[p + e] = val.v[e];
    }
}

// element (c, p) of the channel-concatenated NCHW pair, 0 outside
template <class T>
__device__ __forceinline__ const T* nchw_ptr(const void* a, const void* b, int cA, int cB, int64_t n, int c, int hw)
{
    if (c < cA) return (const T*)a + (n * cA + c) * (int64_t)hw;
    if (c < cA + cB) return (const T*)b + (n * cB + (c - cA)) * (int64_t)hw;
    return nullptr;
}

// 4 (rows) x 16 (columns) block of a row-major 16-bit LDS matrix through the gfx950 transpose read: the 16 lanes of a group supply
// the addresses of row (s >> 2), columns 4 (s & 3) .. + 3, and lane s receives column s of the four rows.
template <class T> __device__ __forceinline__ void lds_tr4(const T* p, T* out4)
{
    typedef short short4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    __builtin_memcpy(out4, &v, 8);
}

// pixel p = y * w + x of a plane -> (y, x) without an integer division (exact for p < 2^22; the caller passes invW = 0 beyond)
__device__ __forceinline__ void split_pixel(int p, int w, float invW, int& y, int& x)
{
    if (invW == 0.0f) { y = p / w; x = p - y * w; return; }
    y = (int)((float)p * invW);
    x = p - y * w;
    if (x < 0) { y--; x += w; } else if (x >= w) { y++; x -= w; }
}

// One workgroup = 64 channels x 64 pixels. Phase 1: channel rows of the NCHW source -> LDS (16-byte vectors, 8 lanes per row).
// Phase 2: each 16-lane group takes 16 pixels x 8 channels out of LDS with two transpose reads (lane = pixel, its 8 channels =
// the 16-byte unit of the NHWC destination); the four groups of a wave write 64 contiguous bytes of 16 pixels per instruction.
template <class T, bool RED>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(LayoutArgs a)
{
    __shared__ __attribute__((aligned(16))) T tile[kTile * kLdsStride];     // [c][p], values BEFORE scaling; pitch 72: the four rows of a
                                                                            // transpose read start 36 dwords apart = disjoint banks
    __shared__ __attribute__((aligned(16))) float sc[kTile];
    __shared__ float red[RED ? kTile * 65 : 1];                             // [c][p] products for the fixed-order sum
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
        if (row && p < a.hw) val = load_row8<T>(row, p, a.hw, a.vecP);
        store_vec16<T>(tile + c * kLdsStride + 8 * seg, val);
        if (RED && a.othPlanar)
        {
            // planar partner: the product is formed here, where a thread holds 8 pixels of one channel row of both tensors
            float dsum = 0.0f;
            if (row && cc < a.cOth)
            {
                const T* orow = (const T*)a.othA + (n * a.cOth + cc) * (int64_t)a.hw;
                if (p < a.hw)
                {
                    const Vec16<T> o = load_row8<T>(orow, p, a.hw, a.vecP);          // (pixels past hw: 0 in both operands)
                    #pragma unroll
                    for (int e = 0; e < 8; e++) dsum += (float)to_acc(val.v[e]) * (float)to_acc(o.v[e]);
                }
            }
            red[c * 65 + seg] = dsum;
        }
    }
    if (tid < kTile) sc[tid] = (a.scale && c0 + tid < cSrc) ? a.scale[n * cSrc + c0 + tid] : 1.0f;
    __syncthreads();
    // phase 2: wave = 16 pixels, lane group g and trip `it` = channels 8 (4 it + g) .. + 7
    const int lane = tid & 63, wv = tid >> 6, g = lane >> 4, s = lane & 15;
    const int pl = 16 * wv + s, p = p0 + pl;
    int64_t dp = n * a.hw + p;
    if (a.srcW > 0)
    {
        int y, x;
        split_pixel(p, a.srcW, a.invW, y, x);
        dp = n * a.dstHW + (int64_t)(y + a.offY) * a.dstW + (x + a.offX);
    }
    #pragma unroll
    for (int it = 0; it < 2; it++)
    {
        const int cg = 4 * it + g, cc = c0 + 8 * cg;
        Vec16<T> xin;
        const T* q = tile + (8 * cg + (s >> 2)) * kLdsStride + 16 * wv + 4 * (s & 3);
        lds_tr4<T>(q, xin.v);
        lds_tr4<T>(q + 4 * kLdsStride, xin.v + 4);
        float x[8];
        #pragma unroll
        for (int j = 0; j < 8; j++) x[j] = (float)to_acc(xin.v[j]);
        const bool live = p < a.hw && cc < a.cNhwc;
        if (live)
        {
            const float4 s0 = *reinterpret_cast<const float4*>(sc + 8 * cg), s1 = *reinterpret_cast<const float4*>(sc + 8 * cg + 4);
            const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            Vec16<T> out;
            #pragma unroll
            for (int j = 0; j < 8; j++) out.v[j] = from_acc<T>(x[j] * sv[j]);
            store_vec16<T>((T*)a.dst + dp * (int64_t)a.cNhwc + cc, out);
        }
        if (RED && !a.othPlanar)
        {
            float d[8];
            #pragma unroll
            for (int j = 0; j < 8; j++) d[j] = 0.0f;
            if (live && cc < a.cOth)
            {
                const Vec16<T> o = load_vec16<T>((const T*)a.othA + (n * a.hw + p) * (int64_t)a.cOth + cc);
                #pragma unroll
                for (int j = 0; j < 8; j++) d[j] = x[j] * (float)to_acc(o.v[j]);
            }
            #pragma unroll
            for (int j = 0; j < 8; j++) red[(8 * cg + j) * 65 + pl] = d[j];
        }
    }
    if (RED)
    {
        __syncthreads();
        if (tid < kTile && c0 + tid < cSrc)
        {
            float t = 0.0f;
            const int terms = a.othPlanar ? 8 : kTile;
            for (int k = 0; k < terms; k++) t += red[tid * 65 + k];                // fixed order
            a.partial[(n * gridDim.x + blockIdx.x) * (int64_t)cSrc + c0 + tid] = t;
        }
    }
}

// The border of the larger frames lvg_modconv2d_nchw_to_nhwc_padded writes into (everything but the src_h x src_w interior at
// (off_y, off_x)): one thread per 16 bytes.
template <class T>
__global__ __launch_bounds__(256) void frame_border_zero_kernel(T* dst, int64_t total, int chunks, int border, int dstW, int dstHW, int offY, int offX,
                                                                int srcH, int srcW, int cNhwc)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % chunks);
    const int64_t r = idx / chunks;
    int k = (int)(r % border);
    const int64_t f = r / border;
    const int top = offY * dstW, bot = dstHW - (offY + srcH) * dstW, side = dstW - srcW;
    int pix;
    if (k < top) pix = k;
    else if (k - top < bot) pix = (offY + srcH) * dstW + (k - top);
    else
    {
        k -= top + bot;
        const int row = k / side, j = k - row * side;
        pix = (offY + row) * dstW + (j < offX ? j : srcW + j);
    }
    Vec16<T> z;
    #pragma unroll
    for (int e = 0; e < 8; e++) z.v[e] = from_acc<T>(0.0f);
    store_vec16<T>(dst + (f * dstHW + pix) * (int64_t)cNhwc + 8 * ch, z);
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
            Vec16<T> out;
            #pragma unroll
            for (int j = 0; j < 8; j++) out.v[j] = from_acc<T>(x[j] * s);
            store_row8<T>(row, p, a.hw, a.vecP, out);
        }
        if (a.partial)
        {
            float d = 0.0f;
            const T* orow = nchw_ptr<T>(a.othA, a.othB, a.cA, a.cB, n, cc, a.hw);
            if (orow && p < a.hw)
            {
                const Vec16<T> o = load_row8<T>(orow, p, a.hw, a.vecP);              // (pixels past hw read as 0)
                #pragma unroll
                for (int j = 0; j < 8; j++) d += x[j] * (float)to_acc(o.v[j]);
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


// ---------------------------------------------------------------------------------------------------------------------------------
// float32 layers of the super-resolution generator (the three lowest resolutions): the operand of their split-precision
// contraction in ONE pass. dst[n, y + offY, x + offX, blk * cPad + c] = part_blk(src[n, c, y, x] * mul[n, c] * scale) for the
// stacked blocks blk = 0 .. nBlk - 1, part 0 = the float16 rounding of the scaled value, part 1 = the float16 rounding of what
// that left (torch_utils/ops/conv2d_frames.py::split16; `pattern` bit blk selects the part). Replaces cat / float / mul / permute /
// half / sub / half / zeros / three strided copies (~15 passes over a 38 MB tensor per call). The frame's border and its padding
// channels are NOT written (the caller zero-fills the frame once).
struct SplitArgs
{
    const float* src; const float* mul; const float* scale;
    uint16_t* dst;
    int n, c, hw, srcW, dstW, dstHW, offY, offX, cPad, cTot, nBlk, pattern;
    float invW;
};

__global__ __launch_bounds__(256) void split16_frames_kernel(SplitArgs a)
{
    __shared__ float tile[kTile][kTile + 1];                               // [c][p], already multiplied by mul and scale
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * kTile, c0 = blockIdx.y * kTile;
    const int64_t n = blockIdx.z;
    const float sc = a.scale ? a.scale[0] : 1.0f;
    #pragma unroll
    for (int i = 0; i < 16; i++)
    {
        const int idx = tid + 256 * i, c = idx >> 6, pl = idx & 63;
        float v = 0.0f;
        if (c0 + c < a.c && p0 + pl < a.hw)
        {
            v = a.src[(n * a.c + c0 + c) * (int64_t)a.hw + p0 + pl];
            if (a.mul) v *= a.mul[n * a.c + c0 + c];
            v *= sc;                                                        // (a power of two: exact)
        }
        tile[c][pl] = v;
    }
    __syncthreads();
    const int pl = tid & 63, cg0 = tid >> 6, p = p0 + pl;
    if (p >= a.hw) return;
    int y, x;
    split_pixel(p, a.srcW, a.invW, y, x);
    uint16_t* row = a.dst + (n * a.dstHW + (int64_t)(y + a.offY) * a.dstW + (x + a.offX)) * a.cTot;
    #pragma unroll
    for (int it = 0; it < 2; it++)
    {
        const int cl = 8 * (cg0 + 4 * it), cc = c0 + cl;
        if (cc >= a.c) continue;
        uint16_t hi[8], lo[8];
        #pragma unroll
        for (int e = 0; e < 8; e++)
        {
            const float t = tile[cl + e][pl];
            const _Float16 h = (_Float16)t;
            const _Float16 l = (_Float16)(t - (float)h);
            __builtin_memcpy(&hi[e], &h, 2); __builtin_memcpy(&lo[e], &l, 2);
        }
        const int nValid = min(8, a.c - cc);
        for (int blk = 0; blk < a.nBlk; blk++)
        {
            const uint16_t* part = ((a.pattern >> blk) & 1) ? lo : hi;
            uint16_t* d = row + (int64_t)blk * a.cPad + cc;
            if (nValid == 8) *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(part);
            else for (int e = 0; e < nValid; e++) d[e] = part[e];
        }
    }
}


// float32 epilogue of the same layers: dst[n, c, p] = src[n, p, c] * scale[n, c] * factor[0] for c < cDst; src = the float32 NHWC result of
// the split-precision contraction (pixel stride cSrc >= cDst). Replaces mul / permute / mul / contiguous.
struct UnsplitArgs
{
    const float* src; const float* scale; const float* factor;
    float* dst;
    int n, hw, cSrc, cDst;
};

__global__ __launch_bounds__(256) void nhwc_f32_to_nchw_kernel(UnsplitArgs a)
{
    __shared__ float tile[kTile][kTile + 1];                               // [p][c]
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * kTile, c0 = blockIdx.y * kTile;
    const int64_t n = blockIdx.z;
    #pragma unroll
    for (int i = 0; i < 16; i++)
    {
        const int idx = tid + 256 * i, pl = idx >> 6, c = idx & 63;         // consecutive threads: consecutive channels of one pixel
        float v = 0.0f;
        if (p0 + pl < a.hw && c0 + c < a.cDst) v = a.src[(n * a.hw + p0 + pl) * (int64_t)a.cSrc + c0 + c];
        tile[pl][c] = v;
    }
    __syncthreads();
    const float f = a.factor ? a.factor[0] : 1.0f;
    #pragma unroll
    for (int i = 0; i < 16; i++)
    {
        const int idx = tid + 256 * i, c = idx >> 6, pl = idx & 63;         // consecutive threads: consecutive pixels of one channel
        if (c0 + c < a.cDst && p0 + pl < a.hw)
        {
            float v = tile[pl][c] * f;
            if (a.scale) v *= a.scale[n * a.cDst + c0 + c];
            a.dst[(n * a.cDst + c0 + c) * (int64_t)a.hw + p0 + pl] = v;
        }
    }
}

} // namespace

static int nchw_to_nhwc_launch(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                               int64_t n, int64_t hw, int c_a, int c_b, int c_dst, int c_oth, int dtype, void* stream,
                               int src_w, int dst_h, int dst_w, int off_y, int off_x, int zero_border, int oth_planar);

extern "C" int lvg_modconv2d_nchw_to_nhwc(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                                          int64_t n, int64_t hw, int c_a, int c_b, int c_dst, int c_oth, int dtype, void* stream)
{
    return nchw_to_nhwc_launch(src_a, src_b, scale, oth, dst, partial, n, hw, c_a, c_b, c_dst, c_oth, dtype, stream, 0, 0, 0, 0, 0, 0, 0);
}

// The same pass writing into the interior of a larger channels-last frame [n][dst_h][dst_w][c_dst] at (off_y, off_x): source planes are
// src_h x src_w (hw = src_h * src_w). zero_border != 0: the border pixels are zero-filled here (every byte of dst is then written:
// the caller may pass uninitialised memory); 0: the border is not touched (the caller zero-filled dst). `oth` / `partial` (the
// reduction partner, dense frames of hw pixels) as in lvg_modconv2d_nchw_to_nhwc.
extern "C" int lvg_modconv2d_nchw_to_nhwc_padded(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                                                 int64_t n, int src_h, int src_w, int c_a, int c_b, int c_dst, int c_oth,
                                                 int dst_h, int dst_w, int off_y, int off_x, int zero_border, int dtype, void* stream)
{
    LVG_REQUIRE(src_h >= 1 && src_w >= 1 && off_y >= 0 && off_x >= 0 && dst_h >= src_h + off_y && dst_w >= src_w + off_x,
                "modconv2d_nchw_to_nhwc_padded: the source plane must fit inside the destination frame");
    LVG_REQUIRE((int64_t)dst_h * dst_w <= 0x3fffffffLL, "modconv2d_nchw_to_nhwc_padded: destination frame too large");
    return nchw_to_nhwc_launch(src_a, src_b, scale, oth, dst, partial, n, (int64_t)src_h * src_w, c_a, c_b, c_dst, c_oth, dtype, stream,
                               src_w, dst_h, dst_w, off_y, off_x, zero_border, 0);
}

// ... with a PLANAR reduction partner: oth [n][c_oth][src_h][src_w] in the source's own layout (c_oth <= c_a + c_b channels take part), 16-byte aligned:
// partial[n][tile][c] = sum over the tile's pixels of src * oth. (The convolution that stores planes itself, lvg_conv2d_frames_planes, keeps its OUTPUT for
// the backward pass -- y * demod as NCHW planes -- where the two-pass route kept the channels-last y.)
extern "C" int lvg_modconv2d_nchw_to_nhwc_padded_planar(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                                                        int64_t n, int src_h, int src_w, int c_a, int c_b, int c_dst, int c_oth,
                                                        int dst_h, int dst_w, int off_y, int off_x, int zero_border, int dtype, void* stream)
{
    LVG_REQUIRE(src_h >= 1 && src_w >= 1 && off_y >= 0 && off_x >= 0 && dst_h >= src_h + off_y && dst_w >= src_w + off_x,
                "modconv2d_nchw_to_nhwc_padded_planar: the source plane must fit inside the destination frame");
    LVG_REQUIRE((int64_t)dst_h * dst_w <= 0x3fffffffLL, "modconv2d_nchw_to_nhwc_padded_planar: destination frame too large");
    return nchw_to_nhwc_launch(src_a, src_b, scale, oth, dst, partial, n, (int64_t)src_h * src_w, c_a, c_b, c_dst, c_oth, dtype, stream,
                               src_w, dst_h, dst_w, off_y, off_x, zero_border, 1);
}

static int nchw_to_nhwc_launch(const void* src_a, const void* src_b, const float* scale, const void* oth, void* dst, float* partial,
                               int64_t n, int64_t hw, int c_a, int c_b, int c_dst, int c_oth, int dtype, void* stream,
                               int src_w, int dst_h, int dst_w, int off_y, int off_x, int zero_border, int oth_planar)
{
    if (int rc = check_common("modconv2d_nchw_to_nhwc", n, hw, dtype)) return rc;
    LVG_REQUIRE(src_a && dst && c_a >= 1 && c_b >= 0 && (c_b == 0 || src_b), "modconv2d_nchw_to_nhwc: bad sources");
    LVG_REQUIRE(c_dst >= c_a + c_b && c_dst % 8 == 0, "modconv2d_nchw_to_nhwc: the NHWC channel count must be a multiple of 8 covering the sources");
    LVG_REQUIRE((partial == nullptr) == (oth == nullptr), "modconv2d_nchw_to_nhwc: partial and oth go together");
    LVG_REQUIRE(!oth || oth_planar || (c_oth % 8 == 0 && c_oth >= c_a + c_b), "modconv2d_nchw_to_nhwc: bad reduction partner");
    LVG_REQUIRE(!oth || !oth_planar || (c_oth >= 1 && c_oth <= c_a + c_b), "modconv2d_nchw_to_nhwc: a planar reduction partner has at most the source's channels");
    LVG_REQUIRE(lvg_aligned16(dst) && (!oth || lvg_aligned16(oth)), "modconv2d_nchw_to_nhwc: NHWC tensors must be 16-byte aligned");
    LayoutArgs a = {};
    a.srcA = src_a; a.srcB = src_b; a.othA = oth; a.dst = dst; a.scale = scale; a.partial = partial;
    a.n = (int)n; a.hw = (int)hw; a.cA = c_a; a.cB = c_b; a.cNhwc = c_dst; a.cOth = c_oth;
    a.srcW = src_w; a.dstW = dst_w; a.dstHW = dst_h * dst_w; a.offY = off_y; a.offX = off_x;
    a.vecP = ((hw % 8 == 0) && lvg_aligned16(src_a) && (!src_b || lvg_aligned16(src_b)) && (!(oth && oth_planar) || lvg_aligned16(oth))) ? 2
             : ((hw % 2 == 0 && ((uintptr_t)src_a % 4) == 0 && (!src_b || ((uintptr_t)src_b % 4) == 0) && (!(oth && oth_planar) || ((uintptr_t)oth % 4) == 0)) ? 1 : 0);
    a.othPlanar = (oth && oth_planar) ? 1 : 0;
    dim3 grid((unsigned)lvg_ceil_div(hw, kTile), (unsigned)lvg_ceil_div(c_dst, kTile), (unsigned)n);
    a.invW = (src_w > 0 && hw < (1 << 22)) ? 1.0f / (float)src_w : 0.0f;
    if (zero_border)
    {
        const int border = dst_h * dst_w - (int)hw;
        const int chunks = c_dst / 8;
        const int64_t total = (int64_t)n * border * chunks;
        if (total > 0)
        {
            const unsigned blocks = (unsigned)lvg_ceil_div(total, 256);
            const int src_h = (int)(hw / src_w);
            if (dtype == LVG_F16) hipLaunchKernelGGL(frame_border_zero_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (f16_t*)dst, total, chunks, border, dst_w, dst_h * dst_w, off_y, off_x, src_h, src_w, c_dst);
            else                  hipLaunchKernelGGL(frame_border_zero_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dst, total, chunks, border, dst_w, dst_h * dst_w, off_y, off_x, src_h, src_w, c_dst);
        }
    }
    if (dtype == LVG_F16) { if (partial) hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16_t, true>), grid, dim3(256), 0, (hipStream_t)stream, a); else hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, a); }
    else                  { if (partial) hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t, true>), grid, dim3(256), 0, (hipStream_t)stream, a); else hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, a); }
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
    a.vecP = ((hw % 8 == 0) && lvg_aligned16(dst) && (!oth_a || lvg_aligned16(oth_a)) && (!oth_b || lvg_aligned16(oth_b))) ? 2
             : ((hw % 2 == 0 && ((uintptr_t)dst % 4) == 0 && (!oth_a || ((uintptr_t)oth_a % 4) == 0) && (!oth_b || ((uintptr_t)oth_b % 4) == 0)) ? 1 : 0);
    const int cCover = (oth_a && c_a + c_b > c_dst) ? c_a + c_b : c_dst;
    dim3 grid((unsigned)lvg_ceil_div(hw, kTile), (unsigned)lvg_ceil_div(cCover, kTile), (unsigned)n);
    if (dtype == LVG_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else                  hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("modconv2d_nhwc_to_nchw");
}

extern "C" int lvg_split16_frames(const float* src, const float* mul, const float* scale, void* dst, int64_t n, int c, int src_h, int src_w,
                                  int dst_h, int dst_w, int off_y, int off_x, int c_pad, int n_blocks, int pattern, void* stream)
{
    LVG_REQUIRE(src && dst && n >= 1 && c >= 1 && src_h >= 1 && src_w >= 1, "split16_frames: bad sizes");
    LVG_REQUIRE(off_y >= 0 && off_x >= 0 && dst_h >= src_h + off_y && dst_w >= src_w + off_x, "split16_frames: the source plane must fit inside the destination frame");
    LVG_REQUIRE(c_pad >= c && c_pad % 8 == 0 && n_blocks >= 1 && n_blocks <= 8, "split16_frames: c_pad must be a multiple of 8 that holds the channels; 1..8 blocks");
    LVG_REQUIRE((int64_t)dst_h * dst_w <= 0x3fffffffLL && (int64_t)src_h * src_w <= 0x3fffffffLL && n <= 65535, "split16_frames: frame too large");
    LVG_REQUIRE((((uintptr_t)dst) & 15) == 0, "split16_frames: dst must be 16-byte aligned");
    SplitArgs a;
    a.src = src; a.mul = mul; a.scale = scale; a.dst = (uint16_t*)dst;
    a.n = (int)n; a.c = c; a.hw = src_h * src_w; a.srcW = src_w; a.dstW = dst_w; a.dstHW = dst_h * dst_w; a.offY = off_y; a.offX = off_x;
    a.cPad = c_pad; a.cTot = c_pad * n_blocks; a.nBlk = n_blocks; a.pattern = pattern;
    a.invW = a.hw < (1 << 22) ? 1.0f / (float)src_w : 0.0f;
    const dim3 grid((unsigned)((a.hw + kTile - 1) / kTile), (unsigned)((c + kTile - 1) / kTile), (unsigned)n);
    hipLaunchKernelGGL(split16_frames_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("split16_frames");
}

extern "C" int lvg_nhwc_f32_to_nchw(const float* src, const float* scale, const float* factor, float* dst, int64_t n, int64_t hw, int c_src, int c_dst, void* stream)
{
    LVG_REQUIRE(src && dst && n >= 1 && n <= 65535 && hw >= 1 && hw <= 0x3fffffffLL && c_dst >= 1 && c_src >= c_dst, "nhwc_f32_to_nchw: bad sizes");
    UnsplitArgs a;
    a.src = src; a.scale = scale; a.factor = factor; a.dst = dst; a.n = (int)n; a.hw = (int)hw; a.cSrc = c_src; a.cDst = c_dst;
    const dim3 grid((unsigned)((hw + kTile - 1) / kTile), (unsigned)((c_dst + kTile - 1) / kTile), (unsigned)n);
    hipLaunchKernelGGL(nhwc_f32_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("nhwc_f32_to_nchw");
}
