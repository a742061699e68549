// conv3d_igemm.hip -- the dense contraction of the modulated convolutions as a hand-written implicit GEMM
// on the gfx950 matrix cores, with the temporal-tap sum and the modulated-conv epilogue fused on store.
//
// Replaces, for 16-bit channels-last frames, the pair (MIOpen igemm convolution over tap-stacked output
// channels -> tapconv_epilogue / modconv_epilogue kernel) of the reference's `temporal_modulated_conv3d`
// (model/generator_lres.py:83-125) and its bias_act (:570): the convolution output never makes a round trip
// through HBM before the epilogue, and the kt temporal taps are part of the K loop instead of kt x the output.
//
//   x   [M = frames*H*W][Ci]            time-major channels-last frames (frame f = t * clips + n)
//   w   [kt][kh][kw][Co][Ci]            tap-major, input channel fastest
//   acc[m][co] = sum_{dt,dh,dw,ci} x[m + (dt-kt/2)*tShift + (dh-kh/2)*W + (dw-kw/2)][ci] * w[dt][dh][dw][co][ci]
//               (terms whose source pixel leaves the frame -- rows, columns or time -- are zero: 'same' padding)
//   out[m][co] = clamp(act(acc * pre[f][co] + b[co] + res[m][co]) * gain) * post[f][co]      ysum = acc (saved)
//
// GEMM view: D[co][pixel] = W[co][k] * X[k][pixel], k = (dt, dh, dw, ci). One workgroup owns BM (128 / 256)
// consecutive pixels x BN (64 / 128) output channels; a wave owns 64 pixels x BN/2 channels as 2 x (BN/64) MFMA
// blocks (v_mfma_f32_32x32x16, weights = A operand, pixels = B operand: a lane of the result holds ONE pixel and 4
// consecutive output channels per register quad -> 8-byte channels-last stores).
//
// What makes it implicit: in channels-last memory the pixels a tile needs for ALL kh x kw spatial taps are one
// contiguous band of BM + 2 * (kh/2 * W + kw/2) pixel rows (a tap is a constant shift of the flattened pixel index;
// lanes whose source pixel wraps around a row / frame edge read a zero row of LDS instead). The band of one
// (temporal tap, 64-channel chunk) is brought into LDS ONCE and feeds kh*kw K-steps; only the weight tile
// (BN x 64) is loaded per K-step. Staging is LDS-DMA (`global_load_lds_dwordx4`: no staging registers, no
// ds_write pass): the LDS image is lane-linear, so the bank swizzle (16-byte chunk c of row r lives at chunk
// c ^ ((r >> 1) & 7): conflict-free `ds_read_b128` for any tap shift, since the 16-lane read groups cover 16
// rows that are distinct modulo 16 -- SQ_LDS_BANK_CONFLICT = 0 measured) is applied to the per-lane SOURCE address.
//
// Measured facts the schedule is built on (profiles/r02_conv3d_igemm_*.csv; the round-2 script is in the git history):
//  * the K loop is bound by INSTRUCTION ISSUE, not by the matrix pipe or by bytes: the first version spent 160
//    scalar + 90 vector instructions per K-step and wave next to its 16 MFMAs (31 % MFMA busy). Everything per-step
//    is therefore strength-reduced: running 64-bit tile pointers in SGPRs with the SGPR-base form of the LDS-DMA
//    instruction (no 64-bit lane addresses), masks through ONE address select per pixel block, fragment addresses
//    as (e ^ const) + base;
//  * vmcnt retires in order and the two operands have different latencies (weight tiles are shared by every
//    workgroup and come out of L2 / Infinity Cache, a band is first-touch HBM data): with both on the same waves
//    every K-step paid an HBM round trip. The last BM/128 waves therefore stage ONLY bands (waited for once per
//    band), the others ONLY weight tiles (counted vmcnt per K-step, ring of 2 or 3 tiles);
//  * a register-staged variant of the same schedule (global_load -> ds_write) ran at half the rate and was dropped;
//  * so was a form without the LDS weight ring (every wave loading its own MFMA A-operand fragments from L2 into registers,
//    one barrier per band instead of per K-step, 2 or 3 register sets): bit-identical results, 0.53-0.54 PFLOP/s against 0.99
//    on the 512-channel layers (profiles/r02_conv_direct_weights.log) -- fragment-shaped loads (32 rows x 32 bytes per
//    instruction, every weight byte fetched by two waves) saturate the texture-address path long before the matrix pipe.

#include <atomic>
#include "igemm_kernel.h"

namespace {

struct Plan
{
    int bm, bn, pb, bandRows, nABuf, nBBuf, ldsBytes;
    int64_t mTiles;
    int persistGrid;            // > 0: the persistent kernel with this many workgroups (igemm_kernel.h, PERSIST)
    bool static1;               // the one-workgroup-per-CU static-tap form of the 256 x 64 tile (igemm_kernel.h, STATIC1)
};

int device_cus()
{
    // per device (one process may drive devices of different sizes)
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    int n = cus[dev & 63].load(std::memory_order_relaxed);
    if (n == 0)
    {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
        cus[dev & 63].store(n, std::memory_order_relaxed);
    }
    return n;
}

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// Tile choice (measured on MI355X, tools/conv_bench.py, profiles/r02_conv_variants.log): 128 pixels x 128 channels on
// 4 waves with two independent workgroups per CU is the fastest form on every 512- / 256-channel layer (two
// workgroups de-synchronise their barriers; 256-pixel tiles on 8 waves ran 20 % slower there). Wide frames
// (W >= 32: the halo is half a 128-pixel tile) take 256-pixel tiles.
// LVG_CONV_BM / _BN / _NB / _PERSIST (environment, read once) or lvg_conv3d_frames_set_plan (at run time) override -- A/B measurements and
// the test that all forms compute the same bits.
struct Overrides { int bm, bn, nb, persist; };
Overrides& overrides()
{
    // process-wide on purpose: a backward pass runs on autograd's own thread and has to see what the test's thread set (a
    // thread-local override would silently test the default tile there). Test / measurement hook: include/lvg_test_hooks.h.
    static Overrides o = {env_int("LVG_CONV_BM", 0), env_int("LVG_CONV_BN", 0), env_int("LVG_CONV_NB", 0), env_int("LVG_CONV_PERSIST", 0)};
    return o;
}
int make_plan(int64_t M, int W, int Ci, int Co, int kt, int kh, int kw, Plan& pl, bool outF32 = false)
{
    const int reach = (kh / 2) * W + kw / 2;
    const int ntap = kh * kw;
    auto fill = [&](int bm, int bn, int nb)
    {
        pl.bm = bm; pl.bn = bn; pl.pb = 2;
        pl.bandRows = (int)lvg_ceil_div(bm + 2 * reach, 8) * 8;
        pl.nABuf = (kt * (Ci / kBK) > 1) ? 2 : 1;
        pl.nBBuf = ntap > 1 ? nb : 2;                                  // no spatial taps: everything one K-step ahead
        pl.mTiles = lvg_ceil_div(M, bm);
        pl.ldsBytes = kZeroBytes + pl.nABuf * pl.bandRows * kRowBytes + pl.nBBuf * bn * kRowBytes;
        const int epilogue = (bm / 64 * 2) * 64 * (bn + 16);             // the epilogue stages every wave's 64-pixel x bn/2-channel tile, pitch bn + 16 bytes
        if (pl.ldsBytes < epilogue) pl.ldsBytes = epilogue;
    };
    const int fbm = overrides().bm, fbn = overrides().bn, fnb = overrides().nb;
    int bn = (Co % 128 == 0) ? 128 : 64;
    if (fbn == 64 || (fbn == 128 && Co % 128 == 0)) bn = fbn;
    // (round 4: 64-channel tiles of wide frames too -- the 64 -> 64 layers at 36 x 64 stream 0.9 GB per launch and re-read weights and
    // halo per tile: 513 -> 460 us with 256-pixel tiles, tools/conv_bench.py)
    int bm = (2 * reach >= 64 && lvg_ceil_div(M, 256) * (Co / bn) >= 512) ? 256 : 128;
    if (fbm == 128 || fbm == 256) bm = fbm;
    int nb = 2;
    if (fnb == 2 || fnb == 3) nb = fnb;
    if (outF32) { bm = 128; nb = 2; }                                  // (the float32-output instantiations)
    fill(bm, bn, nb);
    if (pl.ldsBytes > 160 * 1024 && nb == 3) fill(bm, bn, 2);
    if (pl.ldsBytes > 160 * 1024 && bm == 256) fill(128, bn, 2);
    if (pl.ldsBytes > 160 * 1024) return -1;
    // Persistent workgroups for the 64-channel tiles (round 4): a tile of these layers has 9 .. 45 K-steps, and one workgroup per tile ran
    // prologue (band from HBM) -> K loop -> stores one after the other (memory operations alone 254 us, arithmetic alone ~160 us, together
    // 410 us on 64 -> 64 @ 36 x 64: profiles/r04_conv_abl64.log). LVG_CONV_PERSIST=0 switches it off (A/B).
    pl.persistGrid = 0;
    // 256 x 64 tiles with two bands (more than one 64-channel chunk or temporal tap) need > 80 KB of LDS: one workgroup per CU whatever the
    // registers, so they take the static-tap loop (LVG_CONV_STATIC1=0: the generic loop, A/B)
    static const int static1On = env_int("LVG_CONV_STATIC1", 1);
    pl.static1 = static1On && !outF32 && pl.bm == 256 && pl.bn == 64 && nb == 2 && kh == 3 && kw == 3 && (pl.ldsBytes > 80 * 1024 || static1On == 2)   // (2: also the single-band tiles, measurement)
                 && pl.bandRows / 8 <= 2 * 9 * 3;                       // (band pieces per band wave and K-step: the static loop's 3 slots)
    const int persistOn = overrides().persist;
    if (persistOn && !outF32 && ntap > 1 && pl.bn == 64 && nb == 2)
    {
        const int aBytes = pl.bandRows * kRowBytes;
        const int lds = kZeroBytes + 2 * aBytes + pl.nBBuf * pl.bn * kRowBytes;
        const int staging = (pl.bm / 64 * 2) * 64 * (pl.bn + 16);     // every wave's 64 pixels x 32 channels, pitch 64 + 16 bytes: must fit the finished band
        const int64_t tiles = pl.mTiles * (Co / pl.bn);
        int perCu = (160 * 1024) / lds;
        if (perCu > 2) perCu = 2;
        if (staging <= aBytes && perCu >= 1)
        {
            int grid = device_cus() * perCu;
            if (grid > 8) grid -= grid % 8;
            if (tiles >= 3 * (int64_t)grid && tiles < ((int64_t)1 << 30)) { pl.persistGrid = grid; pl.nABuf = 2; pl.ldsBytes = lds; }
        }
    }
    const int nw = pl.bm / (32 * pl.pb) * 2;
    const int aw = ntap > 1 ? nw / 4 : nw;                             // band-staging waves (see `split` in the kernel)
    if (lvg_ceil_div(pl.bandRows / 8, aw * ntap) > 6) return -1;       // MAXAI band pieces per wave and K-step
    return 0;
}

bool shape_ok(int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw, int64_t xstride)
{
    if (ci <= 0 || co <= 0 || ci % kBK != 0 || co % 64 != 0 || kh * kw > 25 || kt > 7) return false;
    const int64_t M = frames * h * w;
    // 32-bit pixel indices (incl. the temporal halo) and 32-bit byte offsets into x
    return xstride >= ci && xstride % 8 == 0 && (kt / 2 + 1) * M < ((int64_t)1 << 31) && M * xstride * 2 < ((int64_t)1 << 32);
}

template <class T, int BM, int BN, int PB, int NB, bool OUTF = false, bool PERSIST = false, bool STATIC1 = false>
int launch(const ConvArgs& a, const Plan& pl, hipStream_t stream)
{
    auto kern = conv3d_igemm_kernel<T, BM, BN, PB, NB, false, OUTF, PERSIST && !STATIC1, STATIC1>;
    if (pl.ldsBytes > 64 * 1024)
    {
        // opt in to > 64 KiB of dynamic LDS; the attribute is per device, setting it again is cheap
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        {
            (void)hipGetLastError();
            lvg_set_error("conv3d_frames: cannot opt in to %d bytes of LDS", pl.ldsBytes);
            return LVG_ERR_LAUNCH;
        }
    }
    const int64_t blocks = PERSIST ? (int64_t)pl.persistGrid : pl.mTiles * a.nTiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BM / (32 * PB) * 128), pl.ldsBytes, stream, a);
    return lvg_check_launch("conv3d_frames");
}

template <class T, int NB>
int launch_ring(const ConvArgs& a, const Plan& pl, hipStream_t s)
{
    if constexpr (NB == 2)
    {
        if (pl.static1 && pl.persistGrid == 0) return launch<T, 256, 64, 2, 2, false, false, true>(a, pl, s);
        if (pl.persistGrid > 0) return pl.bm == 256 ? launch<T, 256, 64, 2, 2, false, true>(a, pl, s) : launch<T, 128, 64, 2, 2, false, true>(a, pl, s);
    }
    if (pl.bm == 256) return pl.bn == 128 ? launch<T, 256, 128, 2, NB>(a, pl, s) : launch<T, 256, 64, 2, NB>(a, pl, s);
    return pl.bn == 128 ? launch<T, 128, 128, 2, NB>(a, pl, s) : launch<T, 128, 64, 2, NB>(a, pl, s);
}

// float32 output (the float32-accurate contraction from split operands): 128-pixel tiles, ring of 2 only
template <class T>
int launch_f32out(const ConvArgs& a, const Plan& pl, hipStream_t s)
{
    return pl.bn == 128 ? launch<T, 128, 128, 2, 2, true>(a, pl, s) : launch<T, 128, 64, 2, 2, true>(a, pl, s);
}

template <class T>
int launch_tile(const ConvArgs& a, const Plan& pl, hipStream_t s)
{
    // kernels without spatial taps run the generic loop, which reads the ring depth (2) from the arguments
    return pl.nBBuf == 3 ? launch_ring<T, 3>(a, pl, s) : launch_ring<T, 2>(a, pl, s);
}

} // namespace

extern "C" int lvg_conv3d_frames_set_plan(int bm, int bn, int nb, int persist)
{
    LVG_REQUIRE((bm == 0 || bm == 128 || bm == 256) && (bn == 0 || bn == 64 || bn == 128) && (nb == 0 || nb == 2 || nb == 3) && (persist == 0 || persist == 1),
                "conv3d_frames_set_plan: bm in {0, 128, 256}, bn in {0, 64, 128}, nb in {0, 2, 3} (0 = the kernel's own choice), persist in {0, 1}");
    overrides().bm = bm; overrides().bn = bn; overrides().nb = nb; overrides().persist = persist;
    return LVG_OK;
}

extern "C" int64_t lvg_conv3d_frames_workgroups(int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw)
{
    Plan pl;
    if (frames <= 0 || h <= 0 || w <= 0 || !shape_ok(frames, h, w, ci, co, kt, kh, kw, ci)) return 0;
    if (make_plan(frames * h * w, w, ci, co, kt, kh, kw, pl) != 0) return 0;
    return pl.mTiles * (co / pl.bn);
}

// ... of lvg_conv3d_frames_ex with out_dtype = LVG_F32
extern "C" int64_t lvg_conv3d_frames_workgroups_f32out(int64_t frames, int h, int w, int ci, int co, int kt, int kh, int kw)
{
    Plan pl;
    if (frames <= 0 || h <= 0 || w <= 0 || !shape_ok(frames, h, w, ci, co, kt, kh, kw, ci)) return 0;
    if (make_plan(frames * h * w, w, ci, co, kt, kh, kw, pl, true) != 0) return 0;
    return pl.mTiles * (co / pl.bn);
}

extern "C" int lvg_conv3d_frames_ex(const void* x, const void* w, const float* pre, const void* b, const void* res, const float* post,
                                    void* out, void* ysum, float* msq_partial,
                                    int64_t frames, int h, int wd, int ci, int co, int kt, int kh, int kw, int64_t frame_shift,
                                    int64_t x_pixel_stride, int dtype, int out_dtype, int act, float alpha, float gain, float clamp, void* stream);

extern "C" int lvg_conv3d_frames(const void* x, const void* w, const float* pre, const void* b, const void* res, const float* post,
                                 void* out, void* ysum, float* msq_partial,
                                 int64_t frames, int h, int wd, int ci, int co, int kt, int kh, int kw, int64_t frame_shift,
                                 int64_t x_pixel_stride, int dtype, int act, float alpha, float gain, float clamp, void* stream)
{
    return lvg_conv3d_frames_ex(x, w, pre, b, res, post, out, ysum, msq_partial, frames, h, wd, ci, co, kt, kh, kw, frame_shift, x_pixel_stride,
                                dtype, dtype, act, alpha, gain, clamp, stream);
}

extern "C" int lvg_conv3d_frames_ex(const void* x, const void* w, const float* pre, const void* b, const void* res, const float* post,
                                    void* out, void* ysum, float* msq_partial,
                                    int64_t frames, int h, int wd, int ci, int co, int kt, int kh, int kw, int64_t frame_shift,
                                    int64_t x_pixel_stride, int dtype, int out_dtype, int act, float alpha, float gain, float clamp, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "conv3d_frames: float16 / bfloat16 operands only (dtype %d)", dtype);
    LVG_REQUIRE(out_dtype == dtype || out_dtype == LVG_F32, "conv3d_frames: the output is the operand type or float32 (out_dtype %d)", out_dtype);
    LVG_REQUIRE(frames > 0 && h > 0 && wd > 0, "conv3d_frames: empty input");
    LVG_REQUIRE((kt & 1) && (kh & 1) && (kw & 1) && kt >= 1 && kh >= 1 && kw >= 1, "conv3d_frames: odd kernel sizes only");
    LVG_REQUIRE(act == LVG_ACT_LINEAR || act == LVG_ACT_RELU || act == LVG_ACT_LRELU, "conv3d_frames: linear / relu / lrelu only");
    LVG_REQUIRE(lvg_aligned16(x) && lvg_aligned16(w) && lvg_aligned16(out) && lvg_aligned16(ysum) && lvg_aligned16(res)
                && lvg_aligned16(pre) && lvg_aligned16(post) && lvg_aligned16(b), "conv3d_frames: pointers must be 16-byte aligned");
    LVG_REQUIRE(frame_shift > 0 && frames % frame_shift == 0, "conv3d_frames: frames must be a multiple of frame_shift");
    if (x_pixel_stride == 0) x_pixel_stride = ci;
    if (!shape_ok(frames, h, wd, ci, co, kt, kh, kw, x_pixel_stride))
    {
        lvg_set_error("conv3d_frames: no kernel for Ci=%d Co=%d taps=%dx%dx%d on %lld pixels (Ci %% 64, Co %% 64, <= 7 x 25 taps, pixel stride %% 8, 32-bit offsets)",
                      ci, co, kt, kh, kw, (long long)(frames * h * wd));
        return LVG_ERR_UNSUPPORTED;
    }
    Plan pl;
    if (make_plan(frames * h * wd, wd, ci, co, kt, kh, kw, pl, out_dtype == LVG_F32) != 0)
    {
        lvg_set_error("conv3d_frames: frame width %d needs a band the tile cannot hold: no kernel", wd);
        return LVG_ERR_UNSUPPORTED;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.pre = pre; a.b = b; a.res = res; a.post = post;
    a.out = out; a.ysum = ysum; a.msqPartial = msq_partial;
    a.M = frames * h * wd;
    a.tShift = frame_shift * h * wd;
    a.H = h; a.W = wd; a.Ci = ci; a.Co = co; a.kt = kt; a.kh = kh; a.kw = kw;
    a.xStride = (int)x_pixel_stride;
    a.reach = (kh / 2) * wd + kw / 2;
    a.bandRows = pl.bandRows;
    a.nABuf = pl.nABuf;
    a.nBBuf = pl.nBBuf;
    a.nTiles = co / pl.bn;
    a.totalTiles = (int)(pl.mTiles * (co / pl.bn));
    a.slopeNeg = act == LVG_ACT_LINEAR ? 1.f : (act == LVG_ACT_RELU ? 0.f : alpha);
    a.gain = gain;
    a.clamp = clamp;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == LVG_F32) return dtype == LVG_BF16 ? launch_f32out<bf16_t>(a, pl, s) : launch_f32out<f16_t>(a, pl, s);
    return dtype == LVG_BF16 ? launch_tile<bf16_t>(a, pl, s) : launch_tile<f16_t>(a, pl, s);
}
