// flrelu_check.cpp -- standalone (no torch) parity + timing driver for lvg_filtered_lrelu through the C ABI.
// TEST / MEASUREMENT TOOL: compares the MFMA kernel (impl 2) and the fp32-VALU kernel (impl 1) against the C oracle
// (oracle/_build/liblvg_oracle.so, float64) on seeded inputs, forward with mask write, then the backward-shaped
// call reading that mask; then times full-size layers with HIP events.
//   make -C oracle && make -C long-video-gan_amd/csrc
//   hipcc --offload-arch=gfx950 -O2 tools/flrelu_check.cpp -o tools/bin/flrelu_check -ldl
//   tools/bin/flrelu_check [check|time|all]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

typedef int (*flrelu_fn)(const void*, void*, const void*, uint8_t*, const float*, const float*, const int64_t*, const int64_t*,
                         const int64_t*, const int64_t*, int, int, int, int, int, int, const int64_t*, int, int, int,
                         float, float, float, int, int, int, void*);
typedef int (*setimpl_fn)(int);
typedef const char* (*err_fn)(void);
typedef int (*orc_fn)(const double*, const double*, const double*, const double*, double*, uint8_t*, int64_t, int64_t, int, int,
                      int, int, int, int, int, int, int, int, int, int, int, int, int, int, double, double, double, int, int, int, int);

static flrelu_fn g_flrelu; static setimpl_fn g_setimpl; static err_fn g_err; static orc_fn g_orc;
typedef int (*timing_fn)(uint32_t*, int); static timing_fn g_timing, g_wtiming;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t b; memcpy(&b, &h, 2); return b; }
static float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

static uint32_t g_seed = 1;
static float rnd_normal()
{
    float s = 0;
    for (int i = 0; i < 12; i++) { g_seed = g_seed * 1664525u + 1013904223u; s += (float)(g_seed >> 8) / 16777216.0f; }
    return s - 6.0f;
}

// windowed-sinc low-pass (stand-in for scipy.signal.firwin; only needs to be a plausible filter)
static std::vector<float> lowpass(int taps, double cutoff)
{
    std::vector<float> f(taps);
    double sum = 0;
    for (int i = 0; i < taps; i++)
    {
        const double t = i - (taps - 1) / 2.0;
        const double sinc = t == 0 ? 1.0 : sin(M_PI * cutoff * t) / (M_PI * cutoff * t);
        const double win = 0.54 - 0.46 * cos(2 * M_PI * i / (taps - 1));
        f[i] = (float)(sinc * win); sum += f[i];
    }
    for (auto& v : f) v = (float)(v / sum);
    return f;
}

struct Case { const char* name; int n, c, h, w, up, down, nu, nd, px0, px1, py0, py1; };

struct Dev
{
    void* p = nullptr; size_t bytes = 0;
    void alloc(size_t b) { bytes = b; HIPCHK(hipMalloc(&p, b ? b : 16)); }
    ~Dev() { if (p) (void)hipFree(p); }
};

static int call(int impl, int dtype, const void* x, void* y, const void* b, uint8_t* s, const float* fu, const float* fd,
                int n, int c, int xh, int xw, int yh, int yw, int nu, int nd, int up, int down, int px0, int py0,
                int swb, int sh, int sofx, int sofy, int sw_active, float gain, float slope, float clamp, int flip, int mode)
{
    g_setimpl(impl);
    int64_t xs[4] = {n, c, xh, xw}, xst[4] = {(int64_t)c * xh * xw, (int64_t)xh * xw, xw, 1};
    int64_t ys[4] = {n, c, yh, yw}, yst[4] = {(int64_t)c * yh * yw, (int64_t)yh * yw, yw, 1};
    int64_t ss[2] = {swb, sh};
    int rc = g_flrelu(x, y, b, s, fu, fd, xs, xst, ys, yst, nu, nd, up, down, px0, py0, ss, sofx, sofy, sw_active, gain, slope, clamp, flip, mode, dtype, nullptr);
    if (rc != 0) printf("  lvg_filtered_lrelu rc=%d: %s\n", rc, g_err());
    return rc;
}

static double max_err(const std::vector<uint16_t>& got, const std::vector<double>& want, int dtype, double* mean_out, double* worst_rel)
{
    double m = 0, sum = 0, wr = 0;
    for (size_t i = 0; i < got.size(); i++)
    {
        const double g = dtype == 1 ? h2f(got[i]) : bf2f(got[i]);
        const double e = fabs(g - want[i]);
        if (!(e == e)) { m = 1e30; break; }
        sum += e; if (e > m) m = e;
        const double rel = e / (fabs(want[i]) + 1.0);
        if (rel > wr) wr = rel;
    }
    *mean_out = sum / got.size(); *worst_rel = wr;
    return m;
}

static const char* impl_name(int impl) { return impl == 5 ? "STRIP" : impl == 4 ? "BAND" : impl == 3 ? "WAVE" : impl == 2 ? "MFMA" : "VALU"; }
static int g_impl_mask = 0xe;     // bit i: run impl i (FLRELU_IMPLS=3,2,1)

static int run_check(const Case& cs, int dtype)
{
    static const float clamp_env = getenv("FLRELU_CLAMP") ? (float)atof(getenv("FLRELU_CLAMP")) : 2.5f;      // (256: no pixel reaches the clamp, the kernels' no-clamp paths run)
    const float gain = sqrtf(2.0f), slope = 0.2f, clamp = clamp_env;
    const int n = cs.n, c = cs.c, xh = cs.h, xw = cs.w, up = cs.up, down = cs.down, nu = cs.nu, nd = cs.nd;
    std::vector<float> fu = lowpass(nu, 0.9 / up), fd = lowpass(nd, 0.9 / down);
    const int cw = xw * up + cs.px0 + cs.px1 - (nu - 1), chh = xh * up + cs.py0 + cs.py1 - (nu - 1);
    const int yw = (cw - (nd - 1) + (down - 1)) / down, yh = (chh - (nd - 1) + (down - 1)) / down;
    const int sw_active = yw * down - (down - 1) + (nd - 1), sh = yh * down - (down - 1) + (nd - 1);
    const int swb = ((sw_active + 15) & ~15) >> 2;
    const size_t nx = (size_t)n * c * xh * xw, ny = (size_t)n * c * yh * yw, ns = (size_t)n * c * sh * swb;
    g_seed = 77 + cs.h * 3 + cs.up;
    std::vector<uint16_t> hx(nx), hb(c), hdy(ny);
    std::vector<double> x64(nx), b64(c), dy64(ny);
    for (size_t i = 0; i < nx; i++) { float v = rnd_normal(); hx[i] = dtype == 1 ? f2h(v) : f2bf(v); x64[i] = dtype == 1 ? h2f(hx[i]) : bf2f(hx[i]); }
    for (int i = 0; i < c; i++) { float v = 0.3f * rnd_normal(); hb[i] = dtype == 1 ? f2h(v) : f2bf(v); b64[i] = dtype == 1 ? h2f(hb[i]) : bf2f(hb[i]); }
    for (size_t i = 0; i < ny; i++) { float v = rnd_normal(); hdy[i] = dtype == 1 ? f2h(v) : f2bf(v); dy64[i] = dtype == 1 ? h2f(hdy[i]) : bf2f(hdy[i]); }
    // oracle forward (separable filter as outer product)
    std::vector<double> fu2((size_t)nu * nu), fd2((size_t)nd * nd), yref(ny), dxref(nx);
    for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++) fu2[i * nu + j] = (double)fu[i] * fu[j];
    for (int i = 0; i < nd; i++) for (int j = 0; j < nd; j++) fd2[i * nd + j] = (double)fd[i] * fd[j];
    std::vector<uint8_t> sref(ns, 0);
    int orc = g_orc(x64.data(), fu2.data(), fd2.data(), b64.data(), yref.data(), sref.data(), n, c, xh, xw, nu, nu, nd, nd, up, down,
                    cs.px0, cs.px1, cs.py0, cs.py1, 0, 0, sh, swb, gain, slope, clamp, 0, 1, yh, yw);
    if (orc) { printf("oracle fwd rc=%d\n", orc); return 1; }

    Dev dx, dy, db, ds, dfu, dfd, ddy, ddx, dzb;
    dx.alloc(nx * 2); dy.alloc(ny * 2); db.alloc(c * 2); ds.alloc(ns); dfu.alloc(nu * 4); dfd.alloc(nd * 4); ddy.alloc(ny * 2); ddx.alloc(nx * 2); dzb.alloc(c * 2);
    HIPCHK(hipMemcpy(dx.p, hx.data(), nx * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, hb.data(), c * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dfu.p, fu.data(), nu * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dfd.p, fd.data(), nd * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ddy.p, hdy.data(), ny * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dzb.p, 0, c * 2));
    const double tol = dtype == 1 ? 5e-3 : 3e-2;
    int fails = 0;
    for (int impl = 5; impl >= 1; impl--)
    {
        if (!((g_impl_mask >> impl) & 1)) continue;
        HIPCHK(hipMemset(dy.p, 0xff, ny * 2)); HIPCHK(hipMemset(ds.p, 0xee, ns)); HIPCHK(hipMemset(ddx.p, 0xff, nx * 2));
        if (call(impl, dtype, dx.p, dy.p, db.p, (uint8_t*)ds.p, (float*)dfu.p, (float*)dfd.p, n, c, xh, xw, yh, yw, nu, nd, up, down, cs.px0, cs.py0,
                 swb, sh, 0, 0, sw_active, gain, slope, clamp, 0, 1)) return 1;
        HIPCHK(hipDeviceSynchronize());
        std::vector<uint16_t> gy(ny); std::vector<uint8_t> gs(ns);
        HIPCHK(hipMemcpy(gy.data(), dy.p, ny * 2, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(gs.data(), ds.p, ns, hipMemcpyDeviceToHost));
        double mean, wrel;
        const double me = max_err(gy, yref, dtype, &mean, &wrel);
        // mask: compare 2-bit codes over the active width; pad bytes must be 0
        size_t diff = 0, total = 0, padbad = 0;
        for (size_t pl = 0; pl < (size_t)n * c; pl++)
            for (int v = 0; v < sh; v++)
            {
                const uint8_t* a = &gs[(pl * sh + v) * swb]; const uint8_t* r = &sref[(pl * sh + v) * swb];
                for (int u = 0; u < sw_active; u++) { total++; if (((a[u >> 2] >> ((u & 3) * 2)) & 3) != ((r[u >> 2] >> ((u & 3) * 2)) & 3)) { diff++;
                    if (getenv("FLRELU_DEBUG_MASK") && diff <= 12) printf("    mask mismatch plane %zu v %d u %d (byte %d): got %d want %d\n", pl, v, u, u >> 2, (a[u >> 2] >> ((u & 3) * 2)) & 3, (r[u >> 2] >> ((u & 3) * 2)) & 3); } }
                for (int k = (sw_active + 3) >> 2; k < swb; k++) if (a[k]) padbad++;
            }
        // backward-shaped call reading the GPU's own mask; oracle in READ mode on the same mask
        const int pp0 = (nu - 1) + (nd - 1) - cs.px0, pp1 = xw * up - yw * down + cs.px0 - (up - 1);
        const int pq0 = (nu - 1) + (nd - 1) - cs.py0, pq1 = xh * up - yh * down + cs.py0 - (up - 1);
        const double gg = (double)gain * up * up / (down * down);
        std::vector<double> zb(c, 0.0);
        orc = g_orc(dy64.data(), fd2.data(), fu2.data(), zb.data(), dxref.data(), gs.data(), n, c, yh, yw, nd, nd, nu, nu, down, up,
                    pp0, pp1, pq0, pq1, -(nu - 1) + cs.px0, -(nu - 1) + cs.py0, sh, swb, gg, slope, INFINITY, 1, 2, xh, xw);
        if (orc) { printf("oracle bwd rc=%d\n", orc); return 1; }
        if (call(impl, dtype, ddy.p, ddx.p, dzb.p, (uint8_t*)ds.p, (float*)dfd.p, (float*)dfu.p, n, c, yh, yw, xh, xw, nd, nu, down, up, pp0, pq0,
                 swb, sh, -(nu - 1) + cs.px0, -(nu - 1) + cs.py0, swb * 4, (float)gg, slope, INFINITY, 1, 2)) return 1;
        HIPCHK(hipDeviceSynchronize());
        std::vector<uint16_t> gdx(nx);
        HIPCHK(hipMemcpy(gdx.data(), ddx.p, nx * 2, hipMemcpyDeviceToHost));
        double bmean, bwrel;
        const double bme = max_err(gdx, dxref, dtype, &bmean, &bwrel);
        // no-mask forward must equal the mask-writing forward bit for bit
        HIPCHK(hipMemset(ddy.p, 0xff, ny * 2));
        if (call(impl, dtype, dx.p, ddy.p, db.p, nullptr, (float*)dfu.p, (float*)dfd.p, n, c, xh, xw, yh, yw, nu, nd, up, down, cs.px0, cs.py0,
                 0, 0, 0, 0, 0, gain, slope, clamp, 0, 0)) return 1;
        HIPCHK(hipDeviceSynchronize());
        std::vector<uint16_t> gy2(ny);
        HIPCHK(hipMemcpy(gy2.data(), ddy.p, ny * 2, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ddy.p, hdy.data(), ny * 2, hipMemcpyHostToDevice));
        size_t nomask_diff = 0;
        for (size_t i = 0; i < ny; i++) if (gy[i] != gy2[i]) nomask_diff++;
        const bool ok = wrel <= tol && bwrel <= tol && (double)diff / total <= 3e-4 && padbad == 0 && nomask_diff == 0;
        printf("%-14s %s impl=%s  y[%d,%d,%d,%d] fwd max %.2e mean %.2e rel %.2e | mask diff %.1e pad %zu | bwd max %.2e mean %.2e rel %.2e | nomask diff %zu  %s\n",
               cs.name, dtype == 1 ? "f16 " : "bf16", impl_name(impl), n, c, yh, yw, me, mean, wrel, (double)diff / total, padbad, bme, bmean, bwrel, nomask_diff, ok ? "OK" : "FAIL");
        if (!ok) fails++;
    }
    return fails;
}

static int g_only_impl = 0, g_reps = 10;

static void run_time(const Case& cs, int dtype, int mode /*1 = fwd write signs, 0 = fwd no signs, 2 = backward-shaped read*/)
{
    const float gain = sqrtf(2.0f), slope = 0.2f, clamp = 256.0f;
    const int n = cs.n, c = cs.c, xh = cs.h, xw = cs.w, up = cs.up, down = cs.down, nu = cs.nu, nd = cs.nd;
    std::vector<float> fu = lowpass(nu, 0.9 / up), fd = lowpass(nd, 0.9 / down);
    const int cw = xw * up + cs.px0 + cs.px1 - (nu - 1), chh = xh * up + cs.py0 + cs.py1 - (nu - 1);
    const int yw = (cw - (nd - 1) + (down - 1)) / down, yh = (chh - (nd - 1) + (down - 1)) / down;
    const int sw_active = yw * down - (down - 1) + (nd - 1), sh = yh * down - (down - 1) + (nd - 1);
    const int swb = ((sw_active + 15) & ~15) >> 2;
    const size_t nx = (size_t)n * c * xh * xw, ny = (size_t)n * c * yh * yw, ns = (size_t)n * c * sh * swb;
    std::vector<uint16_t> hx(nx), hb(c), hdy(ny);
    g_seed = 5;
    for (auto& v : hx) { float f = rnd_normal(); v = dtype == 1 ? f2h(f) : f2bf(f); }
    for (auto& v : hb) { float f = 0.3f * rnd_normal(); v = dtype == 1 ? f2h(f) : f2bf(f); }
    for (auto& v : hdy) { float f = rnd_normal(); v = dtype == 1 ? f2h(f) : f2bf(f); }
    Dev dx, dy, db, ds, dfu, dfd, ddy, ddx, dzb;
    dx.alloc(nx * 2); dy.alloc(ny * 2); db.alloc(c * 2); ds.alloc(ns); dfu.alloc(nu * 4); dfd.alloc(nd * 4); ddy.alloc(ny * 2); ddx.alloc(nx * 2); dzb.alloc(c * 2);
    HIPCHK(hipMemcpy(dx.p, hx.data(), nx * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, hb.data(), c * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dfu.p, fu.data(), nu * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dfd.p, fd.data(), nd * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ddy.p, hdy.data(), ny * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dzb.p, 0, c * 2));
    const int pp0 = (nu - 1) + (nd - 1) - cs.px0, pq0 = (nu - 1) + (nd - 1) - cs.py0;
    const double gg = (double)gain * up * up / (down * down);
    for (int impl = 5; impl >= 1; impl--)
    {
        if (g_only_impl && impl != g_only_impl) continue;
        if (!g_only_impl && !((g_impl_mask >> impl) & 1)) continue;
        auto once = [&]() {
            if (mode == 2)
                return call(impl, dtype, ddy.p, ddx.p, dzb.p, (uint8_t*)ds.p, (float*)dfd.p, (float*)dfu.p, n, c, yh, yw, xh, xw, nd, nu, down, up, pp0, pq0,
                            swb, sh, -(nu - 1) + cs.px0, -(nu - 1) + cs.py0, swb * 4, (float)gg, slope, INFINITY, 1, 2);
            return call(impl, dtype, dx.p, dy.p, db.p, mode == 1 ? (uint8_t*)ds.p : nullptr, (float*)dfu.p, (float*)dfd.p, n, c, xh, xw, yh, yw, nu, nd, up, down,
                        cs.px0, cs.py0, mode == 1 ? swb : 0, mode == 1 ? sh : 0, 0, 0, mode == 1 ? sw_active : 0, gain, slope, clamp, 0, mode == 1 ? 1 : 0);
        };
        // the mask must exist before a READ-mode run
        if (mode == 2) { call(impl, dtype, dx.p, dy.p, db.p, (uint8_t*)ds.p, (float*)dfu.p, (float*)dfd.p, n, c, xh, xw, yh, yw, nu, nd, up, down, cs.px0, cs.py0, swb, sh, 0, 0, sw_active, gain, slope, clamp, 0, 1); }
        if (once()) return;
        HIPCHK(hipDeviceSynchronize());
        hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
        const int reps = g_reps;
        float best = 1e30f;
        for (int t = 0; t < 3; t++)
        {
            HIPCHK(hipEventRecord(a, nullptr));
            for (int i = 0; i < reps; i++) once();
            HIPCHK(hipEventRecord(b, nullptr));
            HIPCHK(hipEventSynchronize(b));
            float ms; HIPCHK(hipEventElapsedTime(&ms, a, b));
            if (ms / reps < best) best = ms / reps;
        }
        const double bytes = (double)(nx + ny) * 2 + (mode ? (double)ns : 0.0);
        printf("%-10s %s %-5s impl=%s  %8.1f us  %7.1f GB/s  (%.3f of 8 TB/s; algorithmic bytes %.1f MB)\n", cs.name, dtype == 1 ? "f16 " : "bf16",
               mode == 1 ? "fwd+s" : mode == 0 ? "fwd" : "bwd", impl_name(impl), best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12, bytes / 1e6);
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        if (g_wtiming && impl == 3)     // -DLVG_TIMING build of the wave kernel
        {
            std::vector<uint32_t> t(4096 * 16);
            if (g_wtiming(t.data(), 4096 * 16) == 0)
            {
                static const char* names[10] = {"loop", "loads", "mask-in", "stageA", "xwrite", "stageBC", "wwrite", "mask-out", "stageD", "ystore"};
                double sum[12] = {0}; double tiles = 0; int waves = 0;
                for (int wv = 0; wv < 4096; wv++)
                {
                    if (!t[wv * 16 + 12]) continue;
                    waves++; tiles += t[wv * 16 + 12];
                    for (int r = 0; r < 10; r++) sum[r] += t[wv * 16 + r];
                }
                double tot = 0; for (int r = 0; r < 10; r++) tot += sum[r];
                printf("  wave timing: %d waves, %.1f tiles each; cycles per tile per wave:", waves, tiles / waves);
                for (int r = 0; r < 10; r++) printf(" %s %.0f", names[r], sum[r] / tiles);
                printf(" | total %.0f\n", tot / tiles);
            }
        }
        if (g_timing && impl == 2)      // -DLVG_TIMING build of the library: cycles per region, mean over waves, per tile
        {
            std::vector<uint32_t> t(4096 * 16);
            if (g_timing(t.data(), 4096 * 16) == 0)
            {
                static const char* names[10] = {"loop-top", "barrier1", "prefetch", "stageA", "stageD", "stageBC", "barrier2", "wwrite", "maskout", "xwrite"};
                double sum[12] = {0}; double tiles = 0; int waves = 0;
                for (int wv = 0; wv < 4096; wv++)
                {
                    if (!t[wv * 16 + 12]) continue;
                    waves++; tiles += t[wv * 16 + 12];
                    for (int r = 0; r < 10; r++) sum[r] += t[wv * 16 + r];
                }
                double tot = 0; for (int r = 0; r < 10; r++) tot += sum[r];
                printf("  timing: %d waves, %.1f tiles each; cycles per tile per wave:", waves, tiles / waves);
                for (int r = 0; r < 10; r++) printf(" %s %.0f", names[r], sum[r] / tiles);
                printf(" | total %.0f\n", tot / tiles);
            }
        }
    }
}

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const std::string what = argc > 1 ? argv[1] : "all";
    const char* root = getenv("GRAFT_REPO_ROOT"); std::string r = root ? root : ".";
    const char* libenv = getenv("LVG_LIB");
    void* lib = dlopen(libenv ? libenv : (r + "/long-video-gan_amd/lib/liblvg_hip.so").c_str(), RTLD_NOW);
    void* orc = dlopen((r + "/oracle/_build/liblvg_oracle.so").c_str(), RTLD_NOW);
    if (!lib || !orc) { printf("dlopen failed: %s\n", dlerror()); return 2; }
    g_flrelu = (flrelu_fn)dlsym(lib, "lvg_filtered_lrelu"); g_setimpl = (setimpl_fn)dlsym(lib, "lvg_filtered_lrelu_set_impl");
    g_err = (err_fn)dlsym(lib, "lvg_last_error"); g_orc = (orc_fn)dlsym(orc, "orc_filtered_lrelu");
    if (!g_flrelu || !g_setimpl || !g_err || !g_orc) { printf("missing symbol\n"); return 2; }
    g_timing = (timing_fn)dlsym(lib, "lvg_flrelu_timing_read");
    g_wtiming = (timing_fn)dlsym(lib, "lvg_flrelu_wave_timing_read");
    if (const char* im = getenv("FLRELU_IMPLS")) { g_impl_mask = 0; for (const char* c = im; *c; c++) if (*c >= '1' && *c <= '5') g_impl_mask |= 1 << (*c - '0'); }
    int fails = 0;
    if (what == "check" || what == "all")
    {
        const Case cases[] = {
            {"tiny_u2d2", 1, 2, 17, 23, 2, 2, 12, 12, 9, 8, 9, 8},
            {"L4_u2d2", 2, 3, 40, 54, 2, 2, 12, 12, 9, 8, 9, 8},
            {"L8_u2d2", 1, 4, 94, 150, 2, 2, 12, 12, 9, 8, 9, 8},
            {"L13_crop", 1, 3, 166, 278, 2, 2, 12, 12, -11, -12, -11, -12},
            {"L5_u4d2", 2, 3, 40, 54, 4, 2, 24, 12, -6, -9, -6, -9},
            {"L10_u4d2", 1, 3, 94, 150, 4, 2, 24, 12, -6, -9, -6, -9},
            {"tap4_u2d2", 2, 3, 30, 41, 2, 2, 4, 4, 3, 2, 3, 2},
            {"L3_u4d2", 2, 5, 31, 38, 4, 2, 24, 12, -6, -9, -6, -9},
            {"L6_u2d2", 3, 2, 58, 86, 2, 2, 12, 12, 9, 8, 9, 8},
            {"odd_u2d2", 2, 2, 33, 47, 2, 2, 12, 12, 8, 9, 8, 9},
            {"odd_u4d2", 1, 3, 21, 27, 4, 2, 24, 12, -5, -10, -5, -10},
            {"wide_u2d2", 1, 2, 20, 300, 2, 2, 12, 12, 9, 8, 9, 8},
            {"tall_u2d2", 1, 2, 300, 20, 2, 2, 12, 12, 9, 8, 9, 8},
            {"many_u2d2", 3, 7, 40, 54, 2, 2, 12, 12, 9, 8, 9, 8},          // (with LVG_FLRELU_BAND_MAXGRID=2: ten planes per workgroup)
            {"many_u4d2", 3, 5, 24, 38, 4, 2, 24, 12, -6, -9, -6, -9},
            {"exact32_u2d2", 1, 3, 34, 62, 2, 2, 12, 12, 9, 8, 9, 8},       // 32 output rows: the block boundary is the last v-block
            {"rows37_u2d2", 1, 3, 39, 118, 2, 2, 12, 12, 9, 8, 9, 8},       // 37 output rows: two blocks finish in the last v-block
        };
        for (const Case& cs : cases)
            for (int dtype = 1; dtype <= 2; dtype++) fails += run_check(cs, dtype);
        printf("check: %d failure(s)\n", fails);
    }
    if (what == "time" || what == "all")
    {
        const Case big[] = {
            {"L8", 8, 512, 94, 150, 2, 2, 12, 12, 9, 8, 9, 8},
            {"L10", 8, 256, 94, 150, 4, 2, 24, 12, -6, -9, -6, -9},
            {"L13", 8, 128, 166, 278, 2, 2, 12, 12, -11, -12, -11, -12},
        };
        for (const Case& cs : big)
            for (int mode = 0; mode <= 2; mode++) run_time(cs, 1, mode);
        run_time(big[0], 2, 1);
    }
    if (what == "timesmall")
    {
        const Case small[] = {
            {"L4", 8, 512, 40, 54, 2, 2, 12, 12, 9, 8, 9, 8},
            {"L6", 8, 512, 58, 86, 2, 2, 12, 12, 9, 8, 9, 8},
            {"L5", 8, 512, 40, 54, 4, 2, 24, 12, -6, -9, -6, -9},
            {"L7", 8, 512, 58, 86, 4, 2, 24, 12, -6, -9, -6, -9},
            {"L9", 8, 362, 94, 150, 2, 2, 12, 12, 9, 8, 9, 8},
        };
        for (const Case& cs : small)
            for (int mode = 0; mode <= 2; mode++) run_time(cs, 1, mode);
    }
    if (what == "timecold")     // the bench's batch (16 frames): 450 .. 680 MB per call, beyond the 256 MiB Infinity Cache
    {
        const Case cold[] = {
            {"L8x16", 16, 512, 94, 150, 2, 2, 12, 12, 9, 8, 9, 8},
            {"L7x16", 16, 512, 58, 86, 4, 2, 24, 12, -6, -9, -6, -9},
            {"L13x16", 16, 128, 166, 278, 2, 2, 12, 12, -11, -12, -11, -12},
        };
        for (const Case& cs : cold)
            for (int mode = 0; mode <= 2; mode++) run_time(cs, 1, mode);
    }
    if (what == "one")      // one <L8|L10|L13> <dtype 1|2> <mode 0|1|2> <impl 1|2> [reps]: for rocprofv3
    {
        const Case big[] = {
            {"L8", 8, 512, 94, 150, 2, 2, 12, 12, 9, 8, 9, 8},
            {"L10", 8, 256, 94, 150, 4, 2, 24, 12, -6, -9, -6, -9},
            {"L13", 8, 128, 166, 278, 2, 2, 12, 12, -11, -12, -11, -12},
        };
        g_only_impl = atoi(argv[5]);
        g_reps = argc > 6 ? atoi(argv[6]) : 2;
        for (const Case& cs : big)
            if (std::string(cs.name) == argv[2]) run_time(cs, atoi(argv[3]), atoi(argv[4]));
    }
    return fails ? 1 : 0;
}
