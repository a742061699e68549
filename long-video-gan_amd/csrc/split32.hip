// split32.hip -- operand preparation of the float32 route (torch_utils/ops/conv3d_frames.py: float32 tensors on the 16-bit matrix cores as THREE
// bfloat16 parts, t = t1 + t2 + t3 to 24 bits; the reference runs its low-resolution networks in float32 with TF32 off, train_lres.py:267-269).
// One pass over a channels-last float32 tensor writes the stacked bfloat16 operand [pixels, blocks * C] whose channel block k holds part pattern[k]
// of the tensor: [x1 | x1 | x2 | x1 | x2 | x3] for a convolution input, [x1 | x2 | x3] for the weight-gradient operands. 4 + 2 * blocks bytes per
// element; the tensor expressions it replaces (cast, subtract, cast, subtract, cast, concatenate) moved ~50.
//   t1 = bf16(t), r = t - t1 (exact), t2 = bf16(r), t3 = bf16(r - t2)      (round to nearest even, the rounding of torch's float -> bfloat16)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lvg_common.h"

namespace {

struct SplitArgs
{
    const float* x;
    bf16_t* out;
    int64_t pixels, xStride;     // elements between consecutive pixels of x
    int C, blocks, pattern;      // block k holds part (pattern >> 2 k) & 3
};

__global__ __launch_bounds__(256) void split32_stack_kernel(SplitArgs q)
{
    const int vpp = q.C >> 3;                                        // 8-channel vectors per pixel
    const int64_t total = q.pixels * vpp;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t outRow = (int64_t)q.blocks * q.C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride)
    {
        const int64_t m = i / vpp;
        const int v = (int)(i - m * vpp);
        const float* src = q.x + m * q.xStride + v * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        const float t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        Vec16<bf16_t> part[3];
        #pragma unroll
        for (int e = 0; e < 8; e++)
        {
            const bf16_t p1 = from_acc<bf16_t>(t[e]);
            const float r1 = t[e] - to_acc(p1);
            const bf16_t p2 = from_acc<bf16_t>(r1);
            const float r2 = r1 - to_acc(p2);
            part[0].v[e] = p1; part[1].v[e] = p2; part[2].v[e] = from_acc<bf16_t>(r2);
        }
        bf16_t* dst = q.out + m * outRow + v * 8;
        for (int k = 0; k < q.blocks; k++)
        {
            const int which = (q.pattern >> (2 * k)) & 3;
            store_vec16(dst + (int64_t)k * q.C, which == 0 ? part[0] : (which == 1 ? part[1] : part[2]));
        }
    }
}

} // namespace

extern "C" int lvg_split32_stack(const float* x, void* out, int64_t pixels, int channels, int64_t x_pixel_stride, int blocks, int pattern, void* stream)
{
    LVG_REQUIRE(x && out && lvg_aligned16(x) && lvg_aligned16(out), "lvg_split32_stack: null or misaligned pointer");
    LVG_REQUIRE(channels > 0 && channels % 8 == 0 && blocks >= 1 && blocks <= 8 && pixels >= 0, "lvg_split32_stack: channels must be a multiple of 8, 1 .. 8 blocks");
    if (x_pixel_stride == 0) x_pixel_stride = channels;
    LVG_REQUIRE(x_pixel_stride >= channels && x_pixel_stride % 4 == 0, "lvg_split32_stack: bad pixel stride");
    for (int k = 0; k < blocks; k++) LVG_REQUIRE(((pattern >> (2 * k)) & 3) < 3, "lvg_split32_stack: parts are 0 .. 2");
    if (pixels == 0) return LVG_OK;
    SplitArgs q{x, static_cast<bf16_t*>(out), pixels, x_pixel_stride, channels, blocks, pattern};
    const int64_t total = pixels * (channels / 8);
    const int64_t need = (total + 255) / 256;
    const int grid = (int)(need < 256 * 16 ? need : 256 * 16);
    hipLaunchKernelGGL(split32_stack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), q);
    return lvg_check_launch("lvg_split32_stack");
}
