// style_prep.hip -- the style side of a modulated convolution in two launches forward, three backward (gfx950).
//
// The reference normalises the per-frame styles by their maximum per sample and derives the demodulation term from them and the
// squared weights on every forward pass (model/generator_lres.py:97-112):
//     s = s / max|s| over (ci, t) per sample n                         (:99, only when demodulating)
//     demod[n, co, t] = rsqrt( sum_ci w2[co, ci] * s[n, ci, t]^2 + 1e-8 )   (:107-108; w2 = sum over the taps of w^2: weight_prep.hip)
// about 8 tensor passes forward and 25 backward per layer in PyTorch, each a launch of a few microseconds: ~700 launches per step
// for the 20 demodulated layers of the low-resolution generator. Here, on styles in frames order s [T, N, Ci] (row r = t * N + n):
//   lvg_style_prep:          mod [R, Ci] = s / amax[n],  demod [R, Co] = rsqrt(mod^2 . w2^T + 1e-8),  amax [N]
//   lvg_style_prep_backward: (g_mod, g_demod) -> ds [T, N, Ci], dw2 [Co, Ci]
// The three small matrix products (forward: [R, Ci] x [Ci, Co]; backward: [R, Co] x [Co, Ci] and [Co, R] x [R, Ci]) are float32
// tiles on the vector ALUs (64 x 64 per workgroup, 4 x 4 per lane) with the elementwise terms applied on load / on store: they are
// 0.1 - 0.6 GFLOP each, the point is the launch count, not the rate. Reductions run in a fixed order (reproducible); ties in max|s|
// share the gradient equally, as torch.amax does.

#include "lvg_common.h"

namespace {

constexpr int kThreads = 256;

struct StyleArgs
{
    const float* s;        // [T][N][Ci]
    const float* w2;       // [Co][Ci]
    float*       mod;      // [R][Ci]
    float*       demod;    // [R][Co]
    float*       amax;     // [N]
    const float* gMod;     // backward: [R][Ci] or NULL
    const float* gDemod;   // backward: [R][Co]
    float*       gm;       // backward scratch: gradient with respect to mod, [R][Ci]
    float*       ds;       // backward: [T][N][Ci]
    float*       dw2;      // backward: [Co][Ci]
    int          T, N, Ci, Co, R;
};

__device__ __forceinline__ float block_reduce(float v, float* red, bool isMax)
{
    for (int o = 32; o > 0; o >>= 1)
    {
        const float u = __shfl_xor(v, o, 64);
        v = isMax ? fmaxf(v, u) : v + u;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < kThreads / 64; i++) r = isMax ? fmaxf(r, red[i]) : r + red[i];      // fixed order
    __syncthreads();
    return r;
}

// grid (N, splits): every workgroup of a sample scans the whole sample for the maximum (T * Ci floats out of L2) and writes its share of
// the sample's rows.
__global__ __launch_bounds__(kThreads) void style_norm_kernel(StyleArgs p)
{
    __shared__ float red[kThreads / 64];
    const int n = blockIdx.x;
    const int per = p.Ci >> 2;                                        // float4 per row
    const int total = p.T * per;
    float m = 0.f;
    for (int i = threadIdx.x; i < total; i += kThreads)
    {
        const int t = i / per, c = i - t * per;
        const float4 v = *reinterpret_cast<const float4*>(p.s + ((int64_t)t * p.N + n) * p.Ci + c * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    const float a = block_reduce(m, red, true);
    if (blockIdx.y == 0 && threadIdx.x == 0) p.amax[n] = a;
    const int t0 = (int)((int64_t)p.T * blockIdx.y / gridDim.y), t1 = (int)((int64_t)p.T * (blockIdx.y + 1) / gridDim.y);
    for (int i = t0 * per + threadIdx.x; i < t1 * per; i += kThreads)
    {
        const int t = i / per, c = i - t * per;
        const int64_t off = ((int64_t)t * p.N + n) * p.Ci + c * 4;
        const float4 v = *reinterpret_cast<const float4*>(p.s + off);
        *reinterpret_cast<float4*>(p.mod + off) = make_float4(__fdiv_rn(v.x, a), __fdiv_rn(v.y, a), __fdiv_rn(v.z, a), __fdiv_rn(v.w, a));
    }
}

// C[M x N] = A[M x K] . B[K x N] on 64 x 64 tiles, the operands defined per mode:
//   0  demod [R x Co]  = rsqrt(mod^2 . w2^T + 1e-8)                 A(m,k) = mod[m][k]^2        B(k,n) = w2[n][k]
//   1  gm    [R x Ci]  = g_mod + 2 mod (gq . w2)                    A(m,k) = gq[m][k]           B(k,n) = w2[k][n]
//   2  dw2   [Co x Ci] = gq^T . mod^2                               A(m,k) = gq[k][m]           B(k,n) = mod[k][n]^2
// with gq = d loss / d (sum + 1e-8) = -1/2 g_demod demod^3.
constexpr int kBM = 64, kBN = 64, kBK = 32, kPad = 4;

// One workgroup per 64 x 64 tile; 32-deep K steps staged through LDS with the NEXT step's operands already in flight in registers
// (these products are small -- 0.5 GFLOP, 64 .. 128 tiles on 256 CUs, one wave per SIMD -- so the exposed latency of a K step's
// global loads, not arithmetic, was the time: 31 / 33 / 48 us per launch with unprefetched 16-deep steps).
template <int MODE>
__global__ __launch_bounds__(kThreads) void style_gemm_kernel(StyleArgs p)
{
    __shared__ __attribute__((aligned(16))) float As[kBK][kBM + kPad];
    __shared__ __attribute__((aligned(16))) float Bs[kBK][kBN + kPad];
    const int M = MODE == 2 ? p.Co : p.R, N = MODE == 0 ? p.Co : p.Ci, K = MODE == 0 ? p.Ci : (MODE == 1 ? p.Co : p.R);
    const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    auto gq4 = [&](int64_t off) -> float4
    {
        const float4 g = *reinterpret_cast<const float4*>(p.gDemod + off), d = *reinterpret_cast<const float4*>(p.demod + off);
        return make_float4(-0.5f * g.x * d.x * d.x * d.x, -0.5f * g.y * d.y * d.y * d.y, -0.5f * g.z * d.z * d.z * d.z, -0.5f * g.w * d.w * d.w * d.w);
    };
    auto sq4 = [](float4 v) { return make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w); };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // two float4 of each operand tile per thread and K step. "row form": 64 rows x 8 quads along k (source rows contiguous in k, stored
    // transposed); "k form": 32 k rows x 16 quads along the tile's m / n (stored as they come)
    auto fetch = [&](int k0, float4 (&ra)[2], float4 (&rb)[2])
    {
        #pragma unroll
        for (int i = 0; i < 2; i++)
        {
            const int idx = tid + kThreads * i;
            if (MODE == 2)
            {
                const int k = k0 + (idx >> 4), m = m0 + (idx & 15) * 4;                   // gq[k][m .. m + 3]
                ra[i] = (k < K && m < M) ? gq4((int64_t)k * p.Co + m) : zero4;
            }
            else
            {
                const int m = m0 + (idx >> 3), k = k0 + (idx & 7) * 4;                    // row m, k .. k + 3
                float4 v = zero4;
                if (m < M && k < K) v = MODE == 0 ? sq4(*reinterpret_cast<const float4*>(p.mod + (int64_t)m * p.Ci + k)) : gq4((int64_t)m * p.Co + k);
                ra[i] = v;
            }
            if (MODE == 0)
            {
                const int n = n0 + (idx >> 3), k = k0 + (idx & 7) * 4;                    // w2[n][k .. k + 3]
                rb[i] = (n < N && k < K) ? *reinterpret_cast<const float4*>(p.w2 + (int64_t)n * p.Ci + k) : zero4;
            }
            else
            {
                const int k = k0 + (idx >> 4), n = n0 + (idx & 15) * 4;
                float4 v = zero4;
                if (k < K && n < N)
                {
                    v = *reinterpret_cast<const float4*>((MODE == 1 ? p.w2 : p.mod) + (int64_t)k * p.Ci + n);
                    if (MODE == 2) v = sq4(v);
                }
                rb[i] = v;
            }
        }
    };
    auto stage = [&](const float4 (&ra)[2], const float4 (&rb)[2])
    {
        #pragma unroll
        for (int i = 0; i < 2; i++)
        {
            const int idx = tid + kThreads * i;
            if (MODE == 2) *reinterpret_cast<float4*>(&As[idx >> 4][(idx & 15) * 4]) = ra[i];
            else
            {
                const int r = idx >> 3, kq = (idx & 7) * 4;
                As[kq + 0][r] = ra[i].x; As[kq + 1][r] = ra[i].y; As[kq + 2][r] = ra[i].z; As[kq + 3][r] = ra[i].w;
            }
            if (MODE == 0)
            {
                const int r = idx >> 3, kq = (idx & 7) * 4;
                Bs[kq + 0][r] = rb[i].x; Bs[kq + 1][r] = rb[i].y; Bs[kq + 2][r] = rb[i].z; Bs[kq + 3][r] = rb[i].w;
            }
            else *reinterpret_cast<float4*>(&Bs[idx >> 4][(idx & 15) * 4]) = rb[i];
        }
    };

    float acc[4][4] = {};
    float4 ra[2], rb[2];
    fetch(0, ra, rb);
    for (int k0 = 0; k0 < K; k0 += kBK)
    {
        stage(ra, rb);
        __syncthreads();
        if (k0 + kBK < K) fetch(k0 + kBK, ra, rb);                                        // lands while this step is multiplied
        #pragma unroll
        for (int k = 0; k < kBK; k++)
        {
            const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
            #pragma unroll
            for (int i = 0; i < 4; i++)
                #pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    const int n = n0 + tx * 4;
    if (n >= N) return;                                                   // N % 4 == 0 (host check)
    #pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
        float4 o;
        if (MODE == 0)
        {
            o = make_float4(rsqrtf(acc[i][0] + 1e-8f), rsqrtf(acc[i][1] + 1e-8f), rsqrtf(acc[i][2] + 1e-8f), rsqrtf(acc[i][3] + 1e-8f));
            *reinterpret_cast<float4*>(p.demod + (int64_t)m * p.Co + n) = o;
        }
        else if (MODE == 1)
        {
            const int64_t off = (int64_t)m * p.Ci + n;
            const float4 md = *reinterpret_cast<const float4*>(p.mod + off);
            const float4 g = p.gMod ? *reinterpret_cast<const float4*>(p.gMod + off) : zero4;
            o = make_float4(fmaf(2.f * md.x, acc[i][0], g.x), fmaf(2.f * md.y, acc[i][1], g.y), fmaf(2.f * md.z, acc[i][2], g.z), fmaf(2.f * md.w, acc[i][3], g.w));
            *reinterpret_cast<float4*>(p.gm + off) = o;
        }
        else
        {
            o = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            *reinterpret_cast<float4*>(p.dw2 + (int64_t)m * p.Ci + n) = o;
        }
    }
}

// d s = gm / a - [|s| == a] sign(s) (sum over the sample of gm * mod) / (a * ties): grid (N, splits), like the forward pass.
__global__ __launch_bounds__(kThreads) void style_norm_backward_kernel(StyleArgs p)
{
    __shared__ float red[kThreads / 64];
    const int n = blockIdx.x;
    const int per = p.Ci >> 2;
    const int total = p.T * per;
    const float a = p.amax[n];
    float dot = 0.f, ties = 0.f;
    for (int i = threadIdx.x; i < total; i += kThreads)
    {
        const int t = i / per, c = i - t * per;
        const int64_t off = ((int64_t)t * p.N + n) * p.Ci + c * 4;
        const float4 g = *reinterpret_cast<const float4*>(p.gm + off), md = *reinterpret_cast<const float4*>(p.mod + off);
        const float4 sv = *reinterpret_cast<const float4*>(p.s + off);
        dot = fmaf(g.x, md.x, fmaf(g.y, md.y, fmaf(g.z, md.z, fmaf(g.w, md.w, dot))));
        ties += (fabsf(sv.x) == a ? 1.f : 0.f) + (fabsf(sv.y) == a ? 1.f : 0.f) + (fabsf(sv.z) == a ? 1.f : 0.f) + (fabsf(sv.w) == a ? 1.f : 0.f);
    }
    dot = block_reduce(dot, red, false);
    ties = block_reduce(ties, red, false);
    const float inv = 1.f / a;
    const float corr = dot * inv / fmaxf(ties, 1.f);
    auto one = [&](float g, float sv) -> float
    {
        float d = g * inv;
        if (fabsf(sv) == a) d -= corr * (sv > 0.f ? 1.f : (sv < 0.f ? -1.f : 0.f));
        return d;
    };
    const int t0 = (int)((int64_t)p.T * blockIdx.y / gridDim.y), t1 = (int)((int64_t)p.T * (blockIdx.y + 1) / gridDim.y);
    for (int i = t0 * per + threadIdx.x; i < t1 * per; i += kThreads)
    {
        const int t = i / per, c = i - t * per;
        const int64_t off = ((int64_t)t * p.N + n) * p.Ci + c * 4;
        const float4 g = *reinterpret_cast<const float4*>(p.gm + off), sv = *reinterpret_cast<const float4*>(p.s + off);
        *reinterpret_cast<float4*>(p.ds + off) = make_float4(one(g.x, sv.x), one(g.y, sv.y), one(g.z, sv.z), one(g.w, sv.w));
    }
}

int norm_splits(int t)
{
    return t < 8 ? (t < 1 ? 1 : t) : 8;
}

bool sizes_ok(int t, int n, int ci, int co)
{
    return t > 0 && n > 0 && ci > 0 && co > 0 && ci % 4 == 0 && co % 4 == 0 && (int64_t)t * n < (1 << 24) && n <= 65535;
}

} // namespace

extern "C" int lvg_style_prep(const float* s, const float* w2, float* mod, float* demod, float* amax, int t, int n, int ci, int co, void* stream)
{
    LVG_REQUIRE(s && w2 && mod && demod && amax, "style_prep: null pointer");
    LVG_REQUIRE(sizes_ok(t, n, ci, co), "style_prep: T=%d N=%d Ci=%d Co=%d: channel counts must be positive multiples of 4", t, n, ci, co);
    LVG_REQUIRE(lvg_aligned16(s) && lvg_aligned16(w2) && lvg_aligned16(mod) && lvg_aligned16(demod), "style_prep: pointers must be 16-byte aligned");
    StyleArgs a = {};
    a.s = s; a.w2 = w2; a.mod = mod; a.demod = demod; a.amax = amax; a.T = t; a.N = n; a.Ci = ci; a.Co = co; a.R = t * n;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(style_norm_kernel, dim3(n, norm_splits(t)), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(style_gemm_kernel<0>, dim3((unsigned)lvg_ceil_div(co, kBN), (unsigned)lvg_ceil_div(a.R, kBM)), dim3(kThreads), 0, st, a);
    return lvg_check_launch("style_prep");
}

extern "C" int lvg_style_prep_backward(const float* s, const float* amax, const float* w2, const float* mod, const float* demod,
                                       const float* g_mod, const float* g_demod, float* gm_scratch, float* ds, float* dw2,
                                       int t, int n, int ci, int co, void* stream)
{
    LVG_REQUIRE(s && amax && w2 && mod && demod && g_demod && gm_scratch && ds && dw2, "style_prep_backward: null pointer");
    LVG_REQUIRE(sizes_ok(t, n, ci, co), "style_prep_backward: T=%d N=%d Ci=%d Co=%d: channel counts must be positive multiples of 4", t, n, ci, co);
    LVG_REQUIRE(lvg_aligned16(s) && lvg_aligned16(w2) && lvg_aligned16(mod) && lvg_aligned16(demod) && lvg_aligned16(g_mod) && lvg_aligned16(g_demod)
                && lvg_aligned16(gm_scratch) && lvg_aligned16(ds) && lvg_aligned16(dw2), "style_prep_backward: pointers must be 16-byte aligned");
    StyleArgs a = {};
    a.s = s; a.w2 = w2; a.mod = const_cast<float*>(mod); a.demod = const_cast<float*>(demod); a.amax = const_cast<float*>(amax);
    a.gMod = g_mod; a.gDemod = g_demod; a.gm = gm_scratch; a.ds = ds; a.dw2 = dw2;
    a.T = t; a.N = n; a.Ci = ci; a.Co = co; a.R = t * n;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(style_gemm_kernel<1>, dim3((unsigned)lvg_ceil_div(ci, kBN), (unsigned)lvg_ceil_div(a.R, kBM)), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(style_gemm_kernel<2>, dim3((unsigned)lvg_ceil_div(ci, kBN), (unsigned)lvg_ceil_div(co, kBM)), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(style_norm_backward_kernel, dim3(n, norm_splits(t)), dim3(kThreads), 0, st, a);
    return lvg_check_launch("style_prep_backward");
}
