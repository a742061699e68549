// probe_issue.hip -- MEASUREMENT TOOL (round 6): how a gfx950 SIMD shares its issue slots between waves.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_issue.hip -o tools/bin/probe_issue && tools/bin/probe_issue
// Question behind it: filtered_lrelu's fused kernels issue ~300 vector + ~150 scalar + ~46 LDS instructions next to 28 MFMAs per
// 32-row block; the matrix pipe is 33 % busy, the vector pipe 45 %, and the kernel still does not speed up. Which limit is it?
// Every experiment runs W waves per SIMD (one or two workgroups per CU, pinned by their LDS size), each wave looping over
//   NM x { one v_mfma_f32_32x32x16_f16 ; NV packed-f16 multiplies ; NS scalar adds ; NL ds_read_b128 }
// with independent accumulators (IND) or as the dependent chain  mfma -> 8 x cvt_pk of its result -> B operand of the next mfma (DEP).
// Printed: shader cycles (s_memtime) per loop iteration of one wave, and the same divided by W = cycles the SIMD spends per
// iteration of any of its waves. If a SIMD issued ONE instruction of any kind per ~4 cycles the second number would follow the
// total instruction count; if the categories issue side by side it follows max(MFMA pipe, vector count x 4, ...).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int NV, int NS, int NL, int DEP>
__global__ __launch_bounds__(1024) void issue_kernel(int iters, float* out, uint64_t* cyc, float seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
    __syncthreads();
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(seed * (float)(lane + j)); b[j] = (_Float16)(seed * (float)(lane - j)); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    half2v x[8];
    for (int i = 0; i < 8; i++) { x[i][0] = (_Float16)(1.0f + seed * i); x[i][1] = (_Float16)(1.0f - seed * i); }
    half2v sc; sc[0] = (_Float16)(1.0f + seed); sc[1] = (_Float16)(1.0f - seed);
    int su; asm volatile("s_mov_b32 %0, 7" : "=s"(su));
    v4u ld = {0, 0, 0, 0};
    const unsigned char* lp = smem + lane * 16;
    const uint64_t t0 = __builtin_readcyclecounter();
    #pragma unroll 1
    for (int it = 0; it < iters; it++)
    {
        #pragma unroll
        for (int h = 0; h < 2; h++)
        {
            if (DEP)
            {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (DEP == 2)
                {
                    // (the good order: independent work in the shadow of the MFMA, the conversions of its result last)
                    #pragma unroll
                    for (int i = 0; i < NV; i++) { x[i & 7] = x[i & 7] * sc; asm volatile("" : "+v"(x[i & 7])); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                #pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const float2v f = {acc0[2 * j + 8 * h], acc0[2 * j + 1 + 8 * h]};
                    const half2v hv = __builtin_convertvector(f, half2v);
                    b[2 * j] = hv[0]; b[2 * j + 1] = hv[1];
                }
            }
            else if (h == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            else             acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            #pragma unroll
            for (int i = 0; i < NL; i++)
            {
                const v4u v = *reinterpret_cast<const v4u*>(lp + 1024 * ((i + 2 * h) & 7));
                ld[0] ^= v[0]; ld[1] ^= v[1]; ld[2] ^= v[2]; ld[3] ^= v[3];
            }
            if (DEP != 2)
            {
                #pragma unroll
                for (int i = 0; i < NV; i++) { x[i & 7] = x[i & 7] * sc; asm volatile("" : "+v"(x[i & 7])); }
            }
            #pragma unroll
            for (int i = 0; i < NS; i++) { int t_; asm volatile("s_add_u32 %0, %1, 3" : "=s"(t_) : "s"(su)); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float r = (float)su + (float)(ld[0] ^ ld[1] ^ ld[2] ^ ld[3]);
    for (int i = 0; i < 16; i++) r += acc0[i] + acc1[i];
    for (int i = 0; i < 8; i++) r += (float)x[i][0] + (float)x[i][1] + (float)b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long*)&cyc[0], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long*)&cyc[1], (unsigned long long)t1); atomicMin((unsigned long long*)&cyc[2], (unsigned long long)t0); }
}

// the same loop without the MFMA (what the vector / scalar / LDS instructions cost alone)
template <int NV, int NS, int NL>
__global__ __launch_bounds__(1024) void filler_kernel(int iters, float* out, uint64_t* cyc, float seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
    __syncthreads();
    half2v x[8];
    for (int i = 0; i < 8; i++) { x[i][0] = (_Float16)(1.0f + seed * i); x[i][1] = (_Float16)(1.0f - seed * i); }
    half2v sc; sc[0] = (_Float16)(1.0f + seed); sc[1] = (_Float16)(1.0f - seed);
    int su; asm volatile("s_mov_b32 %0, 7" : "=s"(su));
    v4u ld = {0, 0, 0, 0};
    const unsigned char* lp = smem + lane * 16;
    const uint64_t t0 = __builtin_readcyclecounter();
    #pragma unroll 1
    for (int it = 0; it < iters; it++)
    {
        #pragma unroll
        for (int h = 0; h < 2; h++)
        {
            #pragma unroll
            for (int i = 0; i < NL; i++)
            {
                const v4u v = *reinterpret_cast<const v4u*>(lp + 1024 * ((i + 2 * h) & 7));
                ld[0] ^= v[0]; ld[1] ^= v[1]; ld[2] ^= v[2]; ld[3] ^= v[3];
            }
            #pragma unroll
            for (int i = 0; i < NV; i++) { x[i & 7] = x[i & 7] * sc; asm volatile("" : "+v"(x[i & 7])); }
            #pragma unroll
            for (int i = 0; i < NS; i++) { int t_; asm volatile("s_add_u32 %0, %1, 3" : "=s"(t_) : "s"(su)); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float r = (float)su + (float)(ld[0] ^ ld[1] ^ ld[2] ^ ld[3]);
    for (int i = 0; i < 8; i++) r += (float)x[i][0] + (float)x[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long*)&cyc[0], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long*)&cyc[1], (unsigned long long)t1); atomicMin((unsigned long long*)&cyc[2], (unsigned long long)t0); }
}

// Groups of NM matrix products issued back to back -- all on ONE accumulator (CH = 1: each waits for its predecessor's result),
// or alternating between two (CH = 0) or four (CH = 2) accumulators -- followed by NV independent packed multiplies.
template <int NM, int CH, int NV>
__global__ __launch_bounds__(1024) void chain_kernel(int iters, float* out, uint64_t* cyc, float seed)
{
    const int lane = threadIdx.x & 63;
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(seed * (float)(lane + j)); b[j] = (_Float16)(seed * (float)(lane - j)); }
    f32x16 acc[4];
    for (int k = 0; k < 4; k++) for (int r = 0; r < 16; r++) acc[k][r] = 0.0f;
    half2v x[8];
    for (int i = 0; i < 8; i++) { x[i][0] = (_Float16)(1.0f + seed * i); x[i][1] = (_Float16)(1.0f - seed * i); }
    half2v sc; sc[0] = (_Float16)(1.0f + seed); sc[1] = (_Float16)(1.0f - seed);
    const uint64_t t0 = __builtin_readcyclecounter();
    #pragma unroll 1
    for (int it = 0; it < iters; it++)
    {
        #pragma unroll
        for (int m = 0; m < NM; m++)
        {
            const int k = CH == 1 ? 0 : (CH == 0 ? (m & 1) : (m & 3));
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        #pragma unroll
        for (int i = 0; i < NV; i++) { x[i & 7] = x[i & 7] * sc; asm volatile("" : "+v"(x[i & 7])); }
        __builtin_amdgcn_sched_barrier(0);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float r = 0.0f;
    for (int k = 0; k < 4; k++) for (int i = 0; i < 16; i++) r += acc[k][i];
    for (int i = 0; i < 8; i++) r += (float)x[i][0] + (float)x[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long*)&cyc[0], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long*)&cyc[1], (unsigned long long)t1); atomicMin((unsigned long long*)&cyc[2], (unsigned long long)t0); }
}

// NV independent packed multiplies and NN "s_nop 7" (8 wait states each) per group: does a wave's s_nop take issue time from the OTHER waves of its SIMD?
template <int NV, int NN>
__global__ __launch_bounds__(1024) void nop_kernel(int iters, float* out, uint64_t* cyc, float seed)
{
    half2v x[8];
    for (int i = 0; i < 8; i++) { x[i][0] = (_Float16)(1.0f + seed * i); x[i][1] = (_Float16)(1.0f - seed * i); }
    half2v sc; sc[0] = (_Float16)(1.0f + seed); sc[1] = (_Float16)(1.0f - seed);
    const uint64_t t0 = __builtin_readcyclecounter();
    #pragma unroll 1
    for (int it = 0; it < iters; it++)
    {
        #pragma unroll
        for (int h = 0; h < 2; h++)
        {
            #pragma unroll
            for (int i = 0; i < NV; i++) { x[i & 7] = x[i & 7] * sc; asm volatile("" : "+v"(x[i & 7])); }
            #pragma unroll
            for (int i = 0; i < NN; i++) asm volatile("s_nop 7");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float r = 0.0f;
    for (int i = 0; i < 8; i++) r += (float)x[i][0] + (float)x[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long*)&cyc[0], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long*)&cyc[1], (unsigned long long)t1); atomicMin((unsigned long long*)&cyc[2], (unsigned long long)t0); }
}

static float* g_out; static uint64_t* g_cyc; static int g_ncu;

template <class K>
static void run(const char* name, K kern, int mfmaPerHalf, int nv, int ns, int nl)
{
    const int Ws[6] = {1, 2, 3, 4, 6, 8};
    printf("%-34s", name);
    for (int wi = 0; wi < 6; wi++)
    {
        const int W = Ws[wi];
        const int wgPerCu = W > 4 ? 2 : 1, wavesPerWg = 4 * W / wgPerCu;
        const size_t lds = wgPerCu == 1 ? 150 * 1024 : 76 * 1024;          // one / two workgroups fit a CU's 160 KiB
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int iters = 4000;
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(g_ncu * wgPerCu), dim3(64 * wavesPerWg), lds, 0, 200, g_out, g_cyc, 0.001f);
        HIPCHK(hipDeviceSynchronize());
        { const uint64_t init[3] = {0, 0, ~0ull}; HIPCHK(hipMemcpy(g_cyc, init, 24, hipMemcpyHostToDevice)); }
        HIPCHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(g_ncu * wgPerCu), dim3(64 * wavesPerWg), lds, 0, iters, g_out, g_cyc, 0.001f);
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipDeviceSynchronize());
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t c3[3] = {0, 0, 0}; HIPCHK(hipMemcpy(c3, g_cyc, 24, hipMemcpyDeviceToHost));
        const uint64_t c = c3[0];                                            // slowest wave of the launch (the oldest wave of a SIMD wins the arbitration: wave 0 alone shows no contention)
        const double perIter = (double)c / iters;                            // cycles of one wave per loop iteration (two halves)
        printf(" | W%d %7.1f %6.1f", W, perIter, perIter / W);
        (void)ms;
    }
    const int instr = 2 * (mfmaPerHalf + nv + ns + nl);
    printf(" | instr/iter %d (vector+mfma %d)\n", instr, 2 * (mfmaPerHalf + nv));
}

int main()
{
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&g_ncu, hipDeviceAttributeMultiprocessorCount, dev));
    HIPCHK(hipMalloc(&g_out, (size_t)g_ncu * 2 * 1024 * sizeof(float)));
    HIPCHK(hipMalloc(&g_cyc, 64));
    printf("CUs %d. Columns: W waves per SIMD: cycles per iteration of one wave, and / W (SIMD cycles per wave-iteration). One iteration = 2 x the named group.\n", g_ncu);
    #define IND(NV, NS, NL) run("mfma + " #NV " valu + " #NS " salu + " #NL " lds", issue_kernel<NV, NS, NL, 0>, 1, NV, NS, NL)
    #define DEPC(NV, NS, NL) run("DEP mfma>4cvt + " #NV " valu + " #NS " salu + " #NL " lds", issue_kernel<NV, NS, NL, 1>, 1, NV + 4, NS, NL)
    #define DEP2(NV, NS, NL) run("DEP2 mfma>" #NV " valu>4cvt + " #NS " salu + " #NL " lds", issue_kernel<NV, NS, NL, 2>, 1, NV + 4, NS, NL)
    #define FIL(NV, NS, NL) run("no mfma: " #NV " valu + " #NS " salu + " #NL " lds", filler_kernel<NV, NS, NL>, 0, NV, NS, NL)
    if (getenv("PROBE_CHAIN"))
    {
        #define CHN(NM, CH, NV) run("chain: " #NM " mfma (mode " #CH ": 1 one acc, 0 two, 2 four) + " #NV " valu", chain_kernel<NM, CH, NV>, NM / 2, NV / 2, 0, 0)
        CHN(2, 1, 0); CHN(2, 0, 0); CHN(4, 1, 0); CHN(4, 0, 0); CHN(4, 2, 0);
        CHN(2, 1, 16); CHN(2, 0, 16); CHN(4, 1, 32); CHN(4, 0, 32); CHN(4, 2, 32); CHN(4, 1, 16); CHN(4, 0, 16);
        CHN(2, 1, 32); CHN(2, 0, 32); CHN(1, 1, 8); CHN(1, 1, 16);
        #define NOPK(NV, NN) run("no mfma: " #NV " valu + " #NN " x s_nop 7", nop_kernel<NV, NN>, 0, NV, NN, 0)
        NOPK(8, 0); NOPK(8, 1); NOPK(8, 2); NOPK(8, 4); NOPK(0, 4); NOPK(16, 2);
        return 0;
    }
    IND(0, 0, 0);
    FIL(8, 0, 0); FIL(16, 0, 0); FIL(0, 8, 0); FIL(8, 8, 0); FIL(0, 0, 4); FIL(8, 4, 2);
    IND(2, 0, 0); IND(4, 0, 0); IND(6, 0, 0); IND(8, 0, 0); IND(12, 0, 0); IND(16, 0, 0);
    IND(0, 8, 0); IND(8, 4, 0); IND(8, 8, 0); IND(8, 4, 2); IND(12, 6, 2);
    DEPC(0, 0, 0); DEPC(4, 0, 0); DEPC(8, 0, 0); DEPC(8, 4, 2); DEPC(12, 6, 2);
    DEP2(4, 0, 0); DEP2(8, 0, 0); DEP2(12, 0, 0); DEP2(8, 4, 2); DEP2(12, 6, 2);
    return 0;
}
