// probe_unaligned.hip -- MEASUREMENT TOOL: do global dword / dwordx4 loads and stores at byte-misaligned addresses work on this GPU?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
__global__ void k(const uint8_t* src, uint8_t* dst, int ofs)
{
    const int t = threadIdx.x;
    const uint4 v = *reinterpret_cast<const uint4*>(src + ofs + 16 * t);
    *reinterpret_cast<uint4*>(dst + ofs + 16 * t) = v;
    const uint32_t w = *reinterpret_cast<const uint32_t*>(src + 4096 + ofs + 4 * t);
    *reinterpret_cast<uint32_t*>(dst + 4096 + ofs + 4 * t) = w;
}
int main()
{
    uint8_t *s, *d; uint8_t h[8192], o[8192];
    for (int i = 0; i < 8192; i++) h[i] = (uint8_t)(i * 7 + 3);
    hipMalloc(&s, 8192 + 64); hipMalloc(&d, 8192 + 64);
    for (int ofs = 0; ofs < 4; ofs++)
    {
        hipMemcpy(s, h, 8192, hipMemcpyHostToDevice); hipMemset(d, 0, 8192);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, ofs);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o, d, 8192, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; i++) if (o[ofs + i] != h[ofs + i]) bad++;
        for (int i = 0; i < 256; i++) if (o[4096 + ofs + i] != h[4096 + ofs + i]) bad++;
        printf("unaligned probe: byte offset %d: %s, %d mismatches\n", ofs, hipGetErrorString(e), bad);
    }
    return 0;
}
