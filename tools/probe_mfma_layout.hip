// probe_mfma_layout.hip -- measures, on the GPU, the facts the MFMA filtered_lrelu kernel relies on:
//   (1) v_mfma_f32_32x32x16_f16 operand / result lane maps (A: i = lane&31, k = 8*(lane>>5)+j;
//       B: n = lane&31, same k; C/D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5));
//   (2) ds_read_b64_tr_b16: which (source lane, sub-element) every (destination lane, element) receives;
//   (3) __builtin_amdgcn_s_bitreplicate availability is a compile-time matter (not used here).
// Build + run:  hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_layout.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

// D = A * B with A[i][k] = a_in[i*16+k], B[k][n] = b_in[k*32+n] under the ASSUMED lane maps; host checks D.
__global__ void mfma_probe(const _Float16* a_in, const _Float16* b_in, float* d_out)
{
    const int lane = threadIdx.x;
    half8 a, b;
    for (int j = 0; j < 8; j++)
    {
        const int k = 8 * (lane >> 5) + j;
        a[j] = a_in[(lane & 31) * 16 + k];
        b[j] = b_in[k * 32 + (lane & 31)];
    }
    float16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++)
    {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        d_out[row * 32 + (lane & 31)] = c[r];
    }
}

// Every lane points at 4 consecutive halves holding the code (lane*4 + e); the result tells who got what.
__global__ void tr_probe(int* out)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[64 * 4];
    const int lane = threadIdx.x;
    for (int e = 0; e < 4; e++) lds[lane * 4 + e] = (_Float16)(float)(lane * 4 + e);
    __syncthreads();
    half4 v;
    const uint32_t addr = (uint32_t)(uintptr_t)(&lds[lane * 4]);   // LDS byte address (generic -> low 32 bits = LDS offset)
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int e = 0; e < 4; e++) out[lane * 4 + e] = (int)(float)v[e];
}

int main()
{
    // ---- MFMA
    std::vector<_Float16> a(32 * 16), b(16 * 32);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)((s >> 16) % 17) - 8); };   // small integers: exact
    for (auto& v : a) v = (_Float16)rnd();
    for (auto& v : b) v = (_Float16)rnd();
    _Float16 *da, *db; float* dd; int* dt;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dd, 32 * 32 * 4); hipMalloc(&dt, 256 * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<float> d(32 * 32);
    hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int n = 0; n < 32; n++)
        {
            float ref = 0;
            for (int k = 0; k < 16; k++) ref += (float)a[i * 16 + k] * (float)b[k * 32 + n];
            if (ref != d[i * 32 + n]) bad++;
        }
    printf("mfma_f32_32x32x16_f16 layout assumption: %s (%d mismatches of 1024)\n", bad ? "WRONG" : "CONFIRMED", bad);

    // ---- transpose read
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dt);
    std::vector<int> t(256);
    hipError_t e = hipMemcpy(t.data(), dt, 256 * 4, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 (%s): dest lane l, elem e <- (source lane, sub-element)\n", hipGetErrorString(e));
    int model_bad = 0;
    for (int l = 0; l < 64; l++)
    {
        printf("  l=%2d:", l);
        for (int el = 0; el < 4; el++)
        {
            const int code = t[l * 4 + el];
            printf(" (%2d,%d)", code >> 2, code & 3);
            // model: within the 16-lane group, dest(l, e) = source lane (4*e + (l&15)/4), sub-element (l&3)
            const int grp = l & ~15, li = l & 15;
            if (code != ((grp + 4 * el + (li >> 2)) * 4 + (li & 3))) model_bad++;
        }
        printf("\n");
    }
    printf("tr model [dest(l,e) <- lane 16*(l/16) + 4e + (l%%16)/4, sub-element l%%4]: %s (%d mismatches)\n", model_bad ? "WRONG" : "CONFIRMED", model_bad);
    return 0;
}
