// optim.hip -- Adam step (+ optional exponential moving average of the weights) over one flat float32 range,
// one pass: the optimizer side of the training loop (reference model/video_gan_lres.py:83-90 torch.optim.Adam with
// betas (0, 0.99), :208-214 update_G_ema; model/video_gan_sres.py likewise). The parameters, their gradients
// (lvg.ddp.FlatGradSync) and both moments live in flat buffers, so the ~130 M-element update of both networks is a
// handful of streaming launches instead of per-tensor / multi-tensor kernels plus a separate lerp pass.
//
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
//   p_ema += (p - p_ema) * ema_w                                  (when p_ema is given)
// -- torch.optim.Adam's update (amsgrad off, no weight decay) in the same operation order, so results agree
// with it to float32 rounding. HBM stream: 7 floats per element (9 with the EMA).

#include "lvg_common.h"

namespace {

struct AdamArgs
{
    float* p; const float* g; float* m; float* v; float* ema;
    int64_t n;
    float lr_over_bc1, b1, b2, eps, inv_sqrt_bc2, ema_w;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a)
{
    m = m + (g - m) * (1.0f - a.b1);                    // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.b2 + g * g * (1.0f - a.b2);               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p = p - a.lr_over_bc1 * (m / denom);
}

template <bool EMA>
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a)
{
    const int64_t nvec = a.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256)
    {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        const float4 g = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        adam1(p.x, g.x, m.x, v.x, a); adam1(p.y, g.y, m.y, v.y, a); adam1(p.z, g.z, m.z, v.z, a); adam1(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (EMA)
        {
            float4 e = reinterpret_cast<float4*>(a.ema)[i];
            e.x += (p.x - e.x) * a.ema_w; e.y += (p.y - e.y) * a.ema_w; e.z += (p.z - e.z) * a.ema_w; e.w += (p.w - e.w) * a.ema_w;
            reinterpret_cast<float4*>(a.ema)[i] = e;
        }
    }
    // tail (n % 4 elements)
    const int64_t t = (nvec << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && t < a.n)
    {
        float p = a.p[t], m = a.m[t], v = a.v[t];
        adam1(p, a.g[t], m, v, a);
        a.p[t] = p; a.m[t] = m; a.v[t] = v;
        if (EMA) a.ema[t] += (p - a.ema[t]) * a.ema_w;
    }
}

} // namespace

extern "C" int lvg_adam_step(float* p, const float* g, float* m, float* v, float* p_ema, int64_t n,
                             float lr, float beta1, float beta2, float eps, int64_t step, float ema_weight, void* stream)
{
    LVG_REQUIRE(p && g && m && v && n >= 1 && step >= 1, "adam_step: bad arguments");
    LVG_REQUIRE(lvg_aligned16(p) && lvg_aligned16(g) && lvg_aligned16(m) && lvg_aligned16(v) && (!p_ema || lvg_aligned16(p_ema)),
                "adam_step: ranges must start on 16-byte boundaries");
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.ema = p_ema; a.n = n;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.lr_over_bc1 = (float)((double)lr / bc1); a.b1 = beta1; a.b2 = beta2; a.eps = eps;
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)); a.ema_w = ema_weight;
    int64_t blocks = lvg_ceil_div(lvg_ceil_div(n, 4), 256 * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (p_ema) hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else       hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("adam_step");
}
