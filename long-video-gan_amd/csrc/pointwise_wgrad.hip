// pointwise_wgrad.hip -- weight gradient of the 1 x 1 (skip / residual) convolutions of the low-resolution networks on the gfx950 matrix cores:
//
//   gw[co][ci] = sum over pixels m of dy[m][co] * x[m][ci]
//
// what autograd derives for the 1 x 1 convolutions of reference model/generator_lres.py:558-577 (the skip branch of a generator block) and
// model/discriminator_lres.py:169 (Conv3dLayer with kernel size 1: conv_skip). It is the K-loop of conv3d_wgrad.hip without taps: a GEMM whose K
// dimension is the PIXEL index, both operands lying [pixel][channel] in channels-last memory, staged as they lie by LDS-DMA and read through the
// transpose read `ds_read_b64_tr_b16`. Arithmetic intensity is Co Ci / (Co + Ci) FLOP per byte (32 for 64 x 64 channels): an HBM stream over x and
// dy for the large frames, a few microseconds of matrix work for the small ones. The library's split-K kernels spend as much again on zero-fill /
// cast helper launches (35 launches of 15 us per step, profiles/r05_window_main.csv).
//
// One workgroup (4 waves, 32 x 32 quadrants) owns a 64 (co) x 64 (ci) tile over a contiguous range of pixels; K-step = 64 pixels, two buffer sets;
// the partial sums of the ranges go to part[split][co][ci] and are added by the caller in split order (reproducible, no atomics).
#include "wgrad_common.h"

namespace {

struct PwArgs
{
    const void* x;
    const void* dy;
    float*      part;         // [splits][Co][Ci]
    const void* zeros;        // >= 128 bytes of zeros in device memory
    int64_t     pixels;       // a multiple of 8
    int64_t     steps;        // K-steps in total: ceil(pixels / 64)
    int64_t     stepsPerSplit;
    int         Ci, Co, xStride, dyStride, nct, nit;
};

constexpr int kTile = 64 * kRow;          // one operand tile: 64 pixels x 64 channels

template <class T>
__global__ __launch_bounds__(256) void pointwise_wgrad_kernel(PwArgs p)
{
    __shared__ __attribute__((aligned(256))) unsigned char smem[4 * kTile];     // [dy 0 | x 0 | dy 1 | x 1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave >> 1, ib = wave & 1;

    int bid = blockIdx.x;
    const int it = bid % p.nit; bid /= p.nit;
    const int ct = bid % p.nct;
    const int split = bid / p.nct;
    const int64_t s0 = (int64_t)split * p.stepsPerSplit;
    const int64_t s1 = min(s0 + p.stepsPerSplit, p.steps);

    const uint32_t ldsBase = (uint32_t)(uintptr_t)smem;
    const unsigned char* const xb = static_cast<const unsigned char*>(p.x) + (size_t)it * 64 * 2;
    const unsigned char* const dyb = static_cast<const unsigned char*>(p.dy) + (size_t)ct * 64 * 2;
    const unsigned char* const zb = static_cast<const unsigned char*>(p.zeros);
    const uint32_t xRowB = (uint32_t)p.xStride * 2, dyRowB = (uint32_t)p.dyStride * 2;
    const uint32_t pieceRow = (uint32_t)(lane >> 3);

    // 1-KiB piece i (8 pixel rows) of one operand tile of K-step `s`; pieces past the last pixel are zeros
    auto piece = [&](const unsigned char* base, uint32_t rowStep, int64_t s, int i, uint32_t ldsTile)
    {
        const uint32_t ldsRow = (uint32_t)(i * 8);
        const uint32_t c = (uint32_t)(lane & 7) ^ swz(ldsRow + pieceRow);
        const int64_t first = s * 64 + i * 8;
        if (first < p.pixels) wdma16(base + (size_t)first * rowStep, pieceRow * rowStep + c * 16, ldsTile + ldsRow * kRow);
        else                  wdma16(zb, (uint32_t)(lane & 7) * 16, ldsTile + ldsRow * kRow);
    };
    auto stage = [&](int64_t s, int buf)
    {
        const uint32_t dyT = ldsBase + (uint32_t)(buf * 2) * kTile, xT = dyT + kTile;
        #pragma unroll
        for (int k = 0; k < 2; k++)
        {
            piece(dyb, dyRowB, s, wave + 4 * k, dyT);
            piece(xb, xRowB, s, wave + 4 * k, xT);
        }
    };

    // per-lane transpose-read addresses (the same pixel rows of both tiles)
    const int g = lane >> 5, s16 = lane & 15, hgrp = (lane >> 4) & 1;
    const uint32_t colA = (uint32_t)(cb * 32 + 16 * hgrp + 4 * (s16 & 3)), colB = (uint32_t)(ib * 32 + 16 * hgrp + 4 * (s16 & 3));
    uint32_t aAddr[4], bAddr[4];
    #pragma unroll
    for (int ks = 0; ks < 4; ks++)
    {
        const uint32_t row = (uint32_t)(16 * ks + 8 * g + (s16 >> 2));
        aAddr[ks] = tr_addr(row, colA);
        bAddr[ks] = tr_addr(row, colB);
    }

    f32x16 acc;
    #pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    if (s0 < s1) stage(s0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t s = s0; s < s1; s++)
    {
        const int buf = (int)((s - s0) & 1);
        if (s + 1 < s1) stage(s + 1, buf ^ 1);
        const uint32_t dyT = ldsBase + (uint32_t)(buf * 2) * kTile, xT = dyT + kTile;
        #pragma unroll
        for (int ks = 0; ks < 4; ks++)
        {
            const uint4 a = tr_read8(dyT + aAddr[ks]);
            const uint4 b = tr_read8(xT + bAddr[ks]);
            acc = MmaW<T>::run(a, b, acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    float* out = p.part + (int64_t)split * p.Co * p.Ci;
    const int ci = it * 64 + ib * 32 + (lane & 31);
    #pragma unroll
    for (int r = 0; r < 16; r++)
    {
        const int co = ct * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(int64_t)co * p.Ci + ci] = acc[r];
    }
}

bool pw_ok(int64_t pixels, int ci, int co)
{
    return pixels > 0 && pixels % 8 == 0 && ci > 0 && co > 0 && ci % 64 == 0 && co % 64 == 0 && pixels * (int64_t)std::max(ci, co) * 2 < ((int64_t)1 << 40);
}

// Splits: about four workgroups per compute unit in total (32 KiB of LDS and ~60 registers each: they are all resident), at least 4 K-steps each.
int pw_splits(int64_t pixels, int ci, int co)
{
    const int64_t tiles = (int64_t)(ci / 64) * (co / 64), steps = lvg_ceil_div(pixels, 64);
    int64_t s = (int64_t)1024 / tiles;
    s = std::max<int64_t>(1, std::min<int64_t>(s, lvg_ceil_div(steps, 4)));
    return (int)s;
}

} // namespace

extern "C" int lvg_pointwise_wgrad_splits(int64_t pixels, int ci, int co)
{
    return pw_ok(pixels, ci, co) ? pw_splits(pixels, ci, co) : 0;
}

extern "C" int lvg_pointwise_wgrad(const void* x, const void* dy, float* part, const void* zeros, int64_t pixels, int ci, int co,
                                   int64_t x_pixel_stride, int64_t dy_pixel_stride, int splits, int dtype, void* stream)
{
    LVG_REQUIRE(dtype == LVG_F16 || dtype == LVG_BF16, "lvg_pointwise_wgrad: float16 / bfloat16 only (dtype %d)", dtype);
    if (!pw_ok(pixels, ci, co))
    {
        lvg_set_error("lvg_pointwise_wgrad: no kernel for Ci=%d Co=%d over %lld pixels (channels %% 64, pixels %% 8)", ci, co, (long long)pixels);
        return LVG_ERR_UNSUPPORTED;
    }
    if (x_pixel_stride == 0) x_pixel_stride = ci;
    if (dy_pixel_stride == 0) dy_pixel_stride = co;
    LVG_REQUIRE(x_pixel_stride >= ci && dy_pixel_stride >= co && x_pixel_stride % 8 == 0 && dy_pixel_stride % 8 == 0, "lvg_pointwise_wgrad: bad pixel strides");
    LVG_REQUIRE(x && dy && part && zeros && lvg_aligned16(x) && lvg_aligned16(dy) && lvg_aligned16(part) && lvg_aligned16(zeros), "lvg_pointwise_wgrad: pointers must be 16-byte aligned");
    LVG_REQUIRE(splits == pw_splits(pixels, ci, co), "lvg_pointwise_wgrad: splits must be lvg_pointwise_wgrad_splits(...) (= %d)", pw_splits(pixels, ci, co));
    PwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.part = part; a.zeros = zeros;
    a.pixels = pixels; a.steps = lvg_ceil_div(pixels, 64); a.stepsPerSplit = lvg_ceil_div(a.steps, splits);
    a.Ci = ci; a.Co = co; a.xStride = (int)x_pixel_stride; a.dyStride = (int)dy_pixel_stride;
    a.nct = co / 64; a.nit = ci / 64;
    const int64_t blocks = (int64_t)a.nct * a.nit * splits;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == LVG_BF16) hipLaunchKernelGGL(pointwise_wgrad_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else                   hipLaunchKernelGGL(pointwise_wgrad_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return lvg_check_launch("lvg_pointwise_wgrad");
}
