// Temporal noise filter bank of the low-resolution generator (reference model/generator_lres.py:378-388, BlurredNoise.blur):
//   y[r][f][t] = scale[f] * sum_k noise[r][t + k] * bank[f][k],   r = (sample, noise channel), f = one of F low-pass filters, k < K taps
// i.e. a 'valid' correlation of every noise row with every filter (the reference spells it as a grouped conv1d over F copies of the row).
// The bank is a staircase: filter f holds its taps right-aligned in a row of K (5000) entries, 125 .. 5000 of them non-zero, so the dense
// [rows * T, K] x [K, F] product does ~3x the necessary work, and as a library GEMM it needs the [rows, T, K] window matrix materialised
// (819 MB for 64 rows x 640 frames) because a Toeplitz operand has no BLAS layout.
//
// Here the product runs on the float32 matrix cores (v_mfma_f32_32x32x2_f32: exact float32 products, float32 accumulation) with the Toeplitz
// operand read straight out of the noise row in LDS: lane (t, kk) of a K-pair p takes noise[t0 + t + k0 + 2 p + kk] -- consecutive lanes,
// consecutive addresses. Filters are processed in GROUPS of 32 that share a tap count (the longest of the group, rounded up): group g only
// walks its last 2 * pairs[g] taps. The bank is consumed in a packed form, bankP[pairOff[g] + p][lane] = bank[32 g + lane % 32][k0_g + 2 p +
// lane / 32] (one coalesced 256-byte line per MFMA, L2-resident; packing is done once per bank by the caller).
//
// Workgroup = 4 waves = ONE 32-frame x 32-filter tile for TWO noise rows; the four waves split the K range of the group (balanced whatever the
// group's length) and are summed through LDS in a fixed order (reproducible). Grid: longest groups first.
// Bound: float32 MFMA (157 TFLOP/s): sum_g pairs[g] * 2 rows * 4096 FLOP per tile; 13.2 GFLOP for the shipped bank at 64 rows x 640 frames against 52 GFLOP dense.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lvg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct NoiseBankArgs
{
    const float* noise;      // [R, L]
    const float* bankP;      // packed bank, see above
    const int* pairOff;      // [G + 1] first pair of every group in bankP
    const float* scale;      // [F] or null
    float* out;              // [R, F, T]
    int R, L, T, F, K, G;
    int tBlocks, rowPairs;
};

constexpr int kWaves = 4;
constexpr int kBlock = 8;       // K-pairs per loop iteration of a wave (pairs per group: multiples of 2 * kWaves * kBlock)

__global__ __launch_bounds__(256) void noise_bank_kernel(NoiseBankArgs q)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // heaviest groups (the last ones of an ascending bank) first
    const int perGroup = q.tBlocks * q.rowPairs;
    const int g = q.G - 1 - (int)(blockIdx.x / (unsigned)perGroup);
    const int rem = (int)(blockIdx.x % (unsigned)perGroup);
    const int rp = rem / q.tBlocks, tb = rem - rp * q.tBlocks;
    const int r0 = 2 * rp, t0 = 32 * tb;
    const int p0 = q.pairOff[g], pairs = q.pairOff[g + 1] - p0;      // pairs: a multiple of 2 * kBlock * kWaves
    const int k0 = q.K - 2 * pairs;                                  // first tap of the group (may be negative: the packed bank holds zeros there)
    const int seg = 32 + 2 * pairs;                                  // noise samples a row of the tile needs: [t0 + k0, t0 + k0 + seg)

    // the two noise rows -> LDS (zeros outside the row)
    for (int i = tid; i < 2 * seg; i += 256)
    {
        const int row = i >= seg, j = i - row * seg;
        const int src = t0 + k0 + j, r = r0 + row;
        smem[i] = (r < q.R && src >= 0 && src < q.L) ? q.noise[(int64_t)r * q.L + src] : 0.f;
    }
    __syncthreads();

    f32x16 acc0, acc1;
    #pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int per = pairs / kWaves;                                  // this wave's share of the K range (a multiple of 2 * kBlock)
    const int pBeg = wave * per;
    const float* bp = q.bankP + ((int64_t)(p0 + pBeg) * 64 + lane);
    const float* a0 = smem + (lane & 31) + (lane >> 5) + 2 * pBeg;
    const float* a1 = a0 + seg;
    // Eight lines of the bank per block, two register sets: while one block is multiplied the lines of the next are in flight (~1000 matrix cycles to
    // cover the L2 latency; a single set with a register copy at the end of the iteration made the compiler wait for every load there). The loads are
    // unconditional: the packed bank ends in kBlock spare lines, so the last prefetch stays inside the allocation.
    float b0[kBlock], b1[kBlock];
    auto fetch = [&](float (&b)[kBlock], int p) __attribute__((always_inline))
    {
        #pragma unroll
        for (int j = 0; j < kBlock; j++) b[j] = bp[(p + j) * 64];
    };
    auto multiply = [&](const float (&b)[kBlock], int p) __attribute__((always_inline))
    {
        #pragma unroll
        for (int j = 0; j < kBlock; j++)
        {
            const float x0 = a0[2 * (p + j)], x1 = a1[2 * (p + j)];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, b[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, b[j], acc1, 0, 0, 0);
        }
    };
    fetch(b0, 0);
    for (int p = 0; p < per; p += 2 * kBlock)
    {
        fetch(b1, p + kBlock);
        __builtin_amdgcn_sched_barrier(0);              // (the scheduler otherwise sinks the loads behind the products they are meant to overlap)
        multiply(b0, p);
        __builtin_amdgcn_sched_barrier(0);
        fetch(b0, p + 2 * kBlock);
        __builtin_amdgcn_sched_barrier(0);
        multiply(b1, p + kBlock);
        __builtin_amdgcn_sched_barrier(0);
    }

    // sum of the four K ranges: waves 1 .. 3 park their accumulators in LDS, wave 0 adds them in wave order
    __syncthreads();                                                 // the noise rows are no longer read
    if (wave > 0)
    {
        float* dst = smem + ((wave - 1) * 32) * 64 + lane;
        #pragma unroll
        for (int i = 0; i < 16; i++) { dst[i * 64] = acc0[i]; dst[(16 + i) * 64] = acc1[i]; }
    }
    __syncthreads();
    if (wave == 0)
    {
        #pragma unroll
        for (int w = 0; w < kWaves - 1; w++)
        {
            const float* src = smem + (w * 32) * 64 + lane;
            #pragma unroll
            for (int i = 0; i < 16; i++) { acc0[i] += src[i * 64]; acc1[i] += src[(16 + i) * 64]; }
        }
        // result element i of a lane: frame 8 (i / 4) + 4 (lane / 32) + i % 4, filter lane % 32
        const int f = 32 * g + (lane & 31);
        if (f < q.F)
        {
            const float s = q.scale ? q.scale[f] : 1.f;
            #pragma unroll
            for (int row = 0; row < 2; row++)
            {
                const int r = r0 + row;
                if (r >= q.R) break;
                float* o = q.out + ((int64_t)r * q.F + f) * q.T;
                #pragma unroll
                for (int qd = 0; qd < 4; qd++)
                {
                    const int t = t0 + 8 * qd + 4 * (lane >> 5);
                    float v[4];
                    #pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = (row ? acc1[qd * 4 + j] : acc0[qd * 4 + j]) * s;
                    if (t + 3 < q.T && (q.T & 3) == 0)
                        *reinterpret_cast<float4*>(o + t) = make_float4(v[0], v[1], v[2], v[3]);
                    else
                        for (int j = 0; j < 4; j++)
                            if (t + j < q.T) o[t + j] = v[j];
                }
            }
        }
    }
}

} // namespace

// Layout of the packed bank (what `bankP` must hold), for callers that build it: see the head of this file.
extern "C" int lvg_noise_filter_bank(const float* noise, const float* bankP, const int* pairOff, const float* scale, float* out,
                                     int rows, int length, int frames, int filters, int taps, int groups, int maxPairs, void* stream)
{
    LVG_REQUIRE(noise && bankP && pairOff && out, "lvg_noise_filter_bank: null pointer");
    LVG_REQUIRE(rows >= 0 && frames >= 0 && filters > 0 && taps > 0 && groups > 0 && groups * 32 >= filters && length == frames + taps - 1,
                "lvg_noise_filter_bank: need length == frames + taps - 1 and 32 * groups >= filters (rows %d, length %d, frames %d, filters %d, taps %d, groups %d)",
                rows, length, frames, filters, taps, groups);
    LVG_REQUIRE(maxPairs > 0 && maxPairs % (2 * kBlock * kWaves) == 0, "lvg_noise_filter_bank: pairs per group must be multiples of %d", 2 * kBlock * kWaves);
    if (rows == 0 || frames == 0) return LVG_OK;
    NoiseBankArgs q;
    q.noise = noise; q.bankP = bankP; q.pairOff = pairOff; q.scale = scale; q.out = out;
    q.R = rows; q.L = length; q.T = frames; q.F = filters; q.K = taps; q.G = groups;
    q.tBlocks = (frames + 31) / 32; q.rowPairs = (rows + 1) / 2;
    const size_t noiseBytes = (size_t)2 * (32 + 2 * (size_t)maxPairs) * sizeof(float);
    const size_t sumBytes = (size_t)(kWaves - 1) * 32 * 64 * sizeof(float);
    const size_t lds = noiseBytes > sumBytes ? noiseBytes : sumBytes;
    if (lds > 64 * 1024)
    {
        lvg_set_error("lvg_noise_filter_bank: filters longer than ~8000 taps are not supported (two noise rows of a tile must fit 64 KiB of LDS)");
        return LVG_ERR_UNSUPPORTED;
    }
    const int64_t grid = (int64_t)groups * q.tBlocks * q.rowPairs;
    hipLaunchKernelGGL(noise_bank_kernel, dim3((unsigned)grid), dim3(256), lds, static_cast<hipStream_t>(stream), q);
    return lvg_check_launch("lvg_noise_filter_bank");
}
