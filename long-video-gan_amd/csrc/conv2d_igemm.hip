// conv2d_igemm.hip -- dense 3 x 3 contraction of the super-resolution networks as a hand-written implicit GEMM on the gfx950
// matrix cores: `lvg_conv2d_frames`.
//
// Replaces, for 16-bit channels-last frames, the library convolution inside the reference's `modulated_conv2d`
// (model/generator_sres.py:28-67: `conv2d_gradfix.conv2d(..., padding = kernel - 1, groups = batch)`, :63-66) and -- with the
// weight mirrored and its channel roles exchanged -- its data gradient (what autograd derives for that call,
// torch_utils/ops/conv2d_gradfix.py:37-45 = F.conv2d on torch >= 1.11):
//
//   x   [n][Hi][Wi][Ci]      channels-last frames; the caller has written the zero border of the padded convolution explicitly
//                            (the layout prologue csrc/modconv2d_layout.hip writes the padded frame anyway), so the kernel is a
//                            'valid' correlation and needs no masks
//   w   [kh][kw][Co][Ci]     tap-major, input channel fastest (kh = kw = 3)
//   out[n][oy][ox][co] = pre[n][co] * sum_{dh, dw, ci} x[n][oy + offY + dh][ox + offX + dw][ci] * w[dh][dw][co][ci]
//                            oy < Ho <= Hi - offY - 2, ox < Wo <= Wi - offX - 2
//
// The kernel is igemm_kernel.h with T2D = true: the K loop (LDS-DMA staging with the bank swizzle on the source address, weight
// ring, role-specialised waves, pinned fragment order, LDS-staged 16-byte stores) is the one of conv3d_igemm.hip; what differs
// is the tile. The frames of these networks are up to 290 pixels wide: a contiguous band of 128 pixels plus a row above and below is
// 88 KB there, so a workgroup owns an 8 x 16 PIXEL TILE of one output frame instead and stages the tile's 10 x 18 input patch
// (23.5 KB per 64-channel chunk: two bands + the weight ring = 79 KB, two workgroups per CU). An LDS-DMA lane supplies its own
// source address, so the patch costs no more staging instructions than a contiguous band; the bank layout is keyed by the patch
// column (igemm_kernel.h).
// Roofline: MFMA-bound, 2 * n * Ho * Wo * Co * Ci * 9 FLOP per launch (Ci, Co as padded to multiples of 64).

#include "igemm_kernel.h"

namespace {

struct Plan2D
{
    int bm, bn, bandRows, nABuf, ldsBytes, tilesX, tilesY;
    int64_t mTiles;
    int coMain;      // output channels covered by the bn-channel tiles; the remaining 64 (Co = 128 k + 64) take a second launch of 64-channel tiles
};

int env_int2(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// 8 x 16 tiles on 4 waves, two workgroups per CU (the measured optimum of the time-major kernel: two independent 4-wave
// workgroups de-synchronise their barriers); 16 x 16 tiles on 8 waves through LVG_CONV2D_BM=256 (A/B measurements).
int make_plan2d(int64_t n, int ho, int wo, int ci, int co, Plan2D& pl, bool outF32 = false)
{
    static const int fbm = env_int2("LVG_CONV2D_BM", 0), fbn = env_int2("LVG_CONV2D_BN", 0);                              // read once per process
    pl.bm = (fbm == 256 && !outF32) ? 256 : 128;
    // Co = 128 k + 64 (k >= 1): k tiles of 128 channels + ONE of 64 (two launches) instead of 2 k + 1 tiles of 64 -- the 64-channel
    // tile does half the MFMAs per fragment read and per staged band (measured: 576 output channels as 9 x 64 ran at 0.78 of 4 x 128 + 64)
    pl.bn = co >= 128 ? 128 : 64;
    if (fbn == 64) pl.bn = 64;
    pl.coMain = co / pl.bn * pl.bn;
    const int th = pl.bm / kTileW;
    pl.bandRows = (int)lvg_ceil_div((th + 2) * kPatchPitch, 8) * 8;
    pl.nABuf = ci / kBK > 1 ? 2 : 1;
    pl.tilesX = (int)lvg_ceil_div(wo, kTileW);
    pl.tilesY = (int)lvg_ceil_div(ho, th);
    pl.mTiles = n * pl.tilesX * pl.tilesY;
    pl.ldsBytes = kZeroBytes + pl.nABuf * pl.bandRows * kRowBytes + 2 * pl.bn * kRowBytes;
    const int epilogue = (pl.bm / 64 * 2) * 64 * (pl.bn + 16);
    if (pl.ldsBytes < epilogue) pl.ldsBytes = epilogue;
    if (pl.ldsBytes > 160 * 1024) return -1;
    const int nw = pl.bm / 64 * 2;
    if (lvg_ceil_div(pl.bandRows / 8, (nw / 4) * 9) > 6) return -1;      // MAXAI band pieces per band wave and K-step
    return 0;
}

bool shape_ok2d(int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int kh, int kw, int64_t xstride, int64_t ostride, int offy = 0, int offx = 0)
{
    if (kh != 3 || kw != 3 || ci <= 0 || co <= 0 || ci % kBK != 0 || co % 64 != 0) return false;
    if (n < 1 || ho < 1 || wo < 1 || offy < 0 || offx < 0 || ho > hi - offy - 2 || wo > wi - offx - 2) return false;
    if (xstride < ci || xstride % 8 != 0 || ostride < co || ostride % 8 != 0) return false;
    // 32-bit pixel indices and 32-bit byte offsets into x
    return n * hi * wi < ((int64_t)1 << 31) && n * hi * wi * xstride * 2 < ((int64_t)1 << 32) && n * ho * wo < ((int64_t)1 << 31);
}

template <class T, int BM, int BN, bool OUTF = false, bool PLANES = false>
int launch2d(const ConvArgs2D& a, const Plan2D& pl, hipStream_t stream)
{
    auto kern = conv3d_igemm_kernel<T, BM, BN, 2, 2, true, OUTF, false, false, PLANES>;
    if (pl.ldsBytes > 64 * 1024)
    {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        {
            (void)hipGetLastError();
            lvg_set_error("conv2d_frames: cannot opt in to %d bytes of LDS", pl.ldsBytes);
            return LVG_ERR_LAUNCH;
        }
    }
    const int64_t blocks = pl.mTiles * a.nTiles;
    const int lds = std::max(kZeroBytes + pl.nABuf * pl.bandRows * kRowBytes + 2 * BN * kRowBytes, (BM / 64 * 2) * 64 * (BN + 16));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BM / 64 * 128), lds, stream, a);
    return lvg_check_launch("conv2d_frames");
}

template <class T>
int launch2d_tile(const ConvArgs2D& a0, const Plan2D& pl, hipStream_t s, bool outF32, bool planes = false)
{
    ConvArgs2D a = a0;
    a.coBase = 0;
    a.nTiles = pl.coMain / pl.bn;
    int rc = LVG_OK;
    if (planes)
    {
        // NCHW plane output (8 x 16 tiles): the same two launches -- tiles of 128 channels, then the last 64
        rc = pl.bn == 128 ? launch2d<T, 128, 128, false, true>(a, pl, s) : launch2d<T, 128, 64, false, true>(a, pl, s);
        if (rc != LVG_OK || pl.coMain == a.Co) return rc;
        a.coBase = pl.coMain;
        a.nTiles = 1;
        return launch2d<T, 128, 64, false, true>(a, pl, s);
    }
    if (outF32)             rc = pl.bn == 128 ? launch2d<T, 128, 128, true>(a, pl, s) : launch2d<T, 128, 64, true>(a, pl, s);     // (8 x 16 tiles only)
    else if (pl.bm == 256)  rc = pl.bn == 128 ? launch2d<T, 256, 128>(a, pl, s) : launch2d<T, 256, 64>(a, pl, s);
    else                    rc = pl.bn == 128 ? launch2d<T, 128, 128>(a, pl, s) : launch2d<T, 128, 64>(a, pl, s);
    if (rc != LVG_OK || pl.coMain == a.Co) return rc;
    a.coBase = pl.coMain;                                                // the last 64 channels
    a.nTiles = 1;
    if (outF32) return launch2d<T, 128, 64, true>(a, pl, s);
    return pl.bm == 256 ? launch2d<T, 256, 64>(a, pl, s) : launch2d<T, 128, 64>(a, pl, s);
}

} // namespace

extern "C" int64_t lvg_conv2d_frames_workgroups(int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int kh, int kw)
{
    Plan2D pl;
    if (!shape_ok2d(n, hi, wi, ho, wo, ci, co, kh, kw, ci, co)) return 0;
    if (make_plan2d(n, ho, wo, ci, co, pl) != 0) return 0;
    return pl.mTiles * (pl.coMain / pl.bn + (pl.coMain != co ? 1 : 0));
}

extern "C" int lvg_conv2d_frames(const void* x, const void* w, const float* pre, void* out,
                                 int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int kh, int kw, int in_off_y, int in_off_x,
                                 int64_t x_pixel_stride, int64_t out_pixel_stride, int dtype, int out_dtype, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "conv2d_frames: float16 / bfloat16 operands only (dtype %d)", dtype);
    LVG_REQUIRE(out_dtype == dtype || out_dtype == LVG_F32, "conv2d_frames: the output is the operand type or float32 (out_dtype %d)", out_dtype);
    const bool outF32 = out_dtype == LVG_F32;
    LVG_REQUIRE(x && w && out, "conv2d_frames: null tensor");
    LVG_REQUIRE(lvg_aligned16(x) && lvg_aligned16(w) && lvg_aligned16(out) && lvg_aligned16(pre), "conv2d_frames: pointers must be 16-byte aligned");
    if (x_pixel_stride == 0) x_pixel_stride = ci;
    if (out_pixel_stride == 0) out_pixel_stride = co;
    if (!shape_ok2d(n, hi, wi, ho, wo, ci, co, kh, kw, x_pixel_stride, out_pixel_stride, in_off_y, in_off_x))
    {
        lvg_set_error("conv2d_frames: no kernel for Ci=%d Co=%d taps=%dx%d, %lld frames %dx%d -> %dx%d at (%d, %d) (3 x 3 taps, channels %% 64, output inside the "
                      "valid region, pixel strides %% 8, 32-bit offsets)", ci, co, kh, kw, (long long)n, hi, wi, ho, wo, in_off_y, in_off_x);
        return LVG_ERR_UNSUPPORTED;
    }
    Plan2D pl;
    if (make_plan2d(n, ho, wo, ci, co, pl, outF32) != 0)
    {
        lvg_set_error("conv2d_frames: no tile plan");
        return LVG_ERR_UNSUPPORTED;
    }
    ConvArgs2D a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.pre = pre; a.out = out;
    a.M = n * ho * wo;
    a.tShift = 0;
    a.H = ho; a.W = wo; a.Ci = ci; a.Co = co; a.kt = 1; a.kh = kh; a.kw = kw;
    a.xStride = (int)x_pixel_stride;
    a.reach = 0;
    a.bandRows = pl.bandRows;
    a.nABuf = pl.nABuf;
    a.nBBuf = 2;
    a.nTiles = pl.coMain / pl.bn;
    a.slopeNeg = 1.f;
    a.gain = 1.f;
    a.clamp = -1.f;
    a.Hi = hi; a.Wi = wi;
    a.tilesX = pl.tilesX; a.tilesY = pl.tilesY;
    a.oStride = (int)out_pixel_stride;
    a.offY = in_off_y; a.offX = in_off_x;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == LVG_BF16 ? launch2d_tile<bf16_t>(a, pl, s, outF32) : launch2d_tile<f16_t>(a, pl, s, outF32);
}

// The same contraction with the result stored as NCHW planes: out [n][co_out][ho][wo] = pre[n][co_out] * acc for the first co_out <= co channels
// (co: the padded count of the weight). wo even (the kernel stores pixel pairs at least); 16-bit output.
static int conv2d_planes_launch(const void* x, const void* w, const float* pre, void* out, const void* dot_a, const void* dot_b, float* dot_partial, int c_dot_a, int c_dot_b,
                                int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int co_out, int kh, int kw, int in_off_y, int in_off_x,
                                int64_t x_pixel_stride, int dtype, void* stream);

extern "C" int lvg_conv2d_frames_planes(const void* x, const void* w, const float* pre, void* out,
                                        int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int co_out, int kh, int kw, int in_off_y, int in_off_x,
                                        int64_t x_pixel_stride, int dtype, void* stream)
{
    return conv2d_planes_launch(x, w, pre, out, nullptr, nullptr, nullptr, 0, 0, n, hi, wi, ho, wo, ci, co, co_out, kh, kw, in_off_y, in_off_x, x_pixel_stride, dtype, stream);
}

// Rows of dot_partial per frame for lvg_conv2d_frames_planes_dot (0: no kernel for the shape): two per 8 x 16 pixel tile.
extern "C" int64_t lvg_conv2d_frames_planes_dot_rows(int ho, int wo)
{
    if (ho < 1 || wo < 1) return 0;
    return 2 * lvg_ceil_div(ho, 8) * lvg_ceil_div(wo, kTileW);
}

// ... and, from the same accumulators (before `pre`), dot_partial[n][row][c] = sum over the pixels of half tile `row` of acc[n][pixel][c] * oth[n][c][pixel] for
// c < c_dot_a + c_dot_b <= co, oth = the channel concatenation of dot_a [n][c_dot_a][ho][wo] and dot_b [n][c_dot_b][ho][wo] (x's dtype; dot_b may be NULL with
// c_dot_b = 0): the caller adds the rows in order (reproducible). The data gradient of the modulated convolution uses it for d styles = sum dx * x
// (model/generator_sres.py:61 `x * styles` differentiated) while dx * styles leaves as planes.
extern "C" int lvg_conv2d_frames_planes_dot(const void* x, const void* w, const float* pre, void* out, const void* dot_a, const void* dot_b, float* dot_partial,
                                            int c_dot_a, int c_dot_b,
                                            int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int co_out, int kh, int kw, int in_off_y, int in_off_x,
                                            int64_t x_pixel_stride, int dtype, void* stream)
{
    LVG_REQUIRE(dot_a && dot_partial && c_dot_a >= 1 && c_dot_b >= 0 && (c_dot_b == 0 || dot_b) && c_dot_a + c_dot_b <= co,
                "conv2d_frames_planes_dot: bad reduction partner (channels %d + %d of %d)", c_dot_a, c_dot_b, co);
    LVG_REQUIRE(((uintptr_t)dot_a % 4) == 0 && ((uintptr_t)dot_b % 4) == 0 && lvg_aligned16(dot_partial), "conv2d_frames_planes_dot: partner planes must be 4-byte aligned");
    return conv2d_planes_launch(x, w, pre, out, dot_a, dot_b, dot_partial, c_dot_a, c_dot_b, n, hi, wi, ho, wo, ci, co, co_out, kh, kw, in_off_y, in_off_x, x_pixel_stride, dtype, stream);
}

static int conv2d_planes_launch(const void* x, const void* w, const float* pre, void* out, const void* dot_a, const void* dot_b, float* dot_partial, int c_dot_a, int c_dot_b,
                                int64_t n, int hi, int wi, int ho, int wo, int ci, int co, int co_out, int kh, int kw, int in_off_y, int in_off_x,
                                int64_t x_pixel_stride, int dtype, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "conv2d_frames_planes: float16 / bfloat16 only (dtype %d)", dtype);
    LVG_REQUIRE(x && w && out, "conv2d_frames_planes: null tensor");
    LVG_REQUIRE(lvg_aligned16(x) && lvg_aligned16(w) && lvg_aligned16(out) && lvg_aligned16(pre), "conv2d_frames_planes: pointers must be 16-byte aligned");
    LVG_REQUIRE(co_out >= 1 && co_out <= co && wo % 2 == 0, "conv2d_frames_planes: 1 <= co_out <= co, even output width (got co_out %d, co %d, width %d)", co_out, co, wo);
    if (x_pixel_stride == 0) x_pixel_stride = ci;
    if (!shape_ok2d(n, hi, wi, ho, wo, ci, co, kh, kw, x_pixel_stride, co, in_off_y, in_off_x) || n * (int64_t)co_out * ho * wo >= ((int64_t)1 << 40))
    {
        lvg_set_error("conv2d_frames_planes: no kernel for Ci=%d Co=%d taps=%dx%d, %lld frames %dx%d -> %dx%d at (%d, %d)", ci, co, kh, kw, (long long)n, hi, wi, ho, wo, in_off_y, in_off_x);
        return LVG_ERR_UNSUPPORTED;
    }
    Plan2D pl;
    if (make_plan2d(n, ho, wo, ci, co, pl, true) != 0)                   // (8 x 16 tiles, as for the float32-output kernels)
    {
        lvg_set_error("conv2d_frames_planes: no tile plan");
        return LVG_ERR_UNSUPPORTED;
    }
    ConvArgs2D a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.pre = pre; a.out = out;
    a.M = n * ho * wo;
    a.H = ho; a.W = wo; a.Ci = ci; a.Co = co; a.kt = 1; a.kh = kh; a.kw = kw;
    a.xStride = (int)x_pixel_stride;
    a.bandRows = pl.bandRows;
    a.nABuf = pl.nABuf;
    a.nBBuf = 2;
    a.nTiles = pl.coMain / pl.bn;
    a.slopeNeg = 1.f;
    a.gain = 1.f;
    a.clamp = -1.f;
    a.Hi = hi; a.Wi = wi;
    a.tilesX = pl.tilesX; a.tilesY = pl.tilesY;
    a.oStride = co;
    a.offY = in_off_y; a.offX = in_off_x;
    a.coOut = co_out;
    a.dotA = dot_a; a.dotB = dot_b; a.dotPartial = dot_partial; a.cDotA = c_dot_a; a.cDotB = c_dot_b;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == LVG_BF16 ? launch2d_tile<bf16_t>(a, pl, s, false, true) : launch2d_tile<f16_t>(a, pl, s, false, true);
}
