// probe_band.hip -- MEASUREMENT TOOL (round 5): the memory path of a row-band filtered_lrelu kernel, without its arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_band.hip -o tools/bin/probe_band
//   tools/bin/probe_band
// Part 1 (primitives the design relies on; each prints what the hardware does):
//   * buffer_load_dwordx4 ... lds (LDS-DMA, 16 bytes per lane) from global byte offsets that are only 2- / 4-byte aligned;
//   * what a raw-buffer LDS-DMA returns for offsets outside num_records (negative = wrapped, beyond the end, straddling the end);
//   * ds_read_b64_tr_b16 at LDS addresses that are 4- but not 8-byte aligned.
// Part 2 (the mover): one workgroup of NS waves per plane, persistent over a contiguous range of planes. Whole rows of the plane
//   enter an LDS ring as 16-row chunks through LDS-DMA (every global access a contiguous run of the plane), the waves of the
//   workgroup walk down the plane in lock step (one barrier per 16 output rows), every wave reads its 56-column strip from
//   the ring the way stage A of the kernel will (transpose reads) and stores its output rows as 16-byte vectors. The output is
//   y[oy][ox] = x[oy + ry][ox + rx] (zero outside the plane's rows), checked on the host. Timed against a plain 16-byte copy
//   of the same byte count.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void;

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ------------------------------------------------------------------------------------------------ part 1
__global__ void prim_dma_kernel(const uint8_t* src, int numRecords, const int* laneOff, uint8_t* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 1024 / 4; i += 64) reinterpret_cast<uint32_t*>(smem)[i] = 0xEEEEEEEEu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, numRecords, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void)smem, 16, laneOff[threadIdx.x], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    *reinterpret_cast<v4u*>(out + 16 * threadIdx.x) = *reinterpret_cast<const v4u*>(smem + 16 * threadIdx.x);
}

__global__ void prim_tr_kernel(const uint16_t* src, int byteShift, uint16_t* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 2048; i += 64) reinterpret_cast<uint16_t*>(smem)[i] = src[i];
    __syncthreads();
    // 16 rows x 32 columns block of a [.. x 64] matrix whose first element sits byteShift bytes into the LDS
    const int lane = threadIdx.x, g = lane >> 5, hgrp = (lane >> 4) & 1, s = lane & 15;
    const unsigned char* p = smem + byteShift + ((8 * g + (s >> 2)) * 64 + 16 * hgrp + 4 * (s & 3)) * 2;
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * 64 * 2));
    for (int j = 0; j < 4; j++) { out[lane * 8 + j] = (uint16_t)lo[j]; out[lane * 8 + 4 + j] = (uint16_t)hi[j]; }
}

static void part1()
{
    const int N = 8192;
    std::vector<uint8_t> h(N);
    for (int i = 0; i < N; i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *src, *out; int* offs;
    HIPCHK(hipMalloc(&src, N)); HIPCHK(hipMalloc(&out, 1024)); HIPCHK(hipMalloc(&offs, 64 * 4));
    HIPCHK(hipMemcpy(src, h.data(), N, hipMemcpyHostToDevice));
    std::vector<uint8_t> o(1024);
    // (a) alignment of the global side: every lane at 20 * lane + shift
    for (int shift = 0; shift <= 6; shift += 2)
    {
        std::vector<int> lo(64);
        for (int l = 0; l < 64; l++) lo[l] = 1024 + 20 * l + shift;
        HIPCHK(hipMemcpy(offs, lo.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(prim_dma_kernel, dim3(1), dim3(64), 1024, 0, src + 0, N, offs, out);
        hipError_t e = hipDeviceSynchronize();
        HIPCHK(hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int b = 0; b < 16; b++) if (o[16 * l + b] != h[lo[l] + b]) bad++;
        printf("prim: LDS-DMA x4, global offset = 20 lane + %d (alignment %d): %s, %d bad bytes of 1024\n", shift, shift == 0 ? 4 : (shift & 3 ? 2 : 4), hipGetErrorString(e), bad);
    }
    // (b) out-of-range lanes: num_records = 4096
    {
        const int NR = 4096;
        std::vector<int> lo(64, 0);
        lo[0] = -16; lo[1] = -4; lo[2] = NR; lo[3] = NR - 4; lo[4] = NR - 8; lo[5] = NR - 12; lo[6] = NR - 16; lo[7] = NR + 1000; lo[8] = -12; lo[9] = 0x7ffffff0;
        HIPCHK(hipMemcpy(offs, lo.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(prim_dma_kernel, dim3(1), dim3(64), 1024, 0, src + 1024, NR, offs, out);
        hipError_t e = hipDeviceSynchronize();
        HIPCHK(hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost));
        printf("prim: LDS-DMA x4 out of range (num_records %d, base = buffer + 1024): %s\n", NR, hipGetErrorString(e));
        for (int l = 0; l < 10; l++)
        {
            printf("   lane %d offset %11d:", l, lo[l]);
            for (int d = 0; d < 4; d++)
            {
                uint32_t w; memcpy(&w, &o[16 * l + 4 * d], 4);
                const long long src0 = 1024LL + lo[l] + 4 * d;
                uint32_t want = 0; bool have = src0 >= 0 && src0 + 4 <= N;
                if (have) memcpy(&want, &h[src0], 4);
                printf(" %s", w == 0 ? "zero" : (w == 0xEEEEEEEEu ? "untouched" : (have && w == want ? "data" : "other")));
            }
            printf("\n");
        }
    }
    // (c) transpose read at 4-byte (not 8-byte) aligned LDS addresses
    {
        std::vector<uint16_t> m(2048 + 64), r(512);
        for (int i = 0; i < 2048 + 64; i++) m[i] = (uint16_t)i;
        uint16_t *ms, *mo;
        HIPCHK(hipMalloc(&ms, (2048 + 64) * 2)); HIPCHK(hipMalloc(&mo, 512 * 2));
        HIPCHK(hipMemcpy(ms, m.data(), (2048 + 64) * 2, hipMemcpyHostToDevice));
        for (int shift = 0; shift <= 6; shift += 2)
        {
            hipLaunchKernelGGL(prim_tr_kernel, dim3(1), dim3(64), 4096 + 64, 0, ms, shift, mo);
            hipError_t e = hipDeviceSynchronize();
            HIPCHK(hipMemcpy(r.data(), mo, 1024, hipMemcpyDeviceToHost));
            // expected (tools/probe_mfma_layout.hip): lane (n = lane & 31, g = lane >> 5), element j = M[row 8 g + j][col n]
            int bad = 0;
            for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) if (r[l * 8 + j] != (uint16_t)((8 * (l >> 5) + j) * 64 + (l & 31) + shift / 2)) bad++;
            printf("prim: ds_read_b64_tr_b16 with the matrix %d bytes into the LDS: %s, %d bad of 512\n", shift, hipGetErrorString(e), bad);
        }
    }
}

// ------------------------------------------------------------------------------------------------ part 2
struct MoverArgs
{
    const uint16_t* x; uint16_t* y;
    int planes, xh, xw, yh, yw;
    int ry, rx;           // y[oy][ox] = x[oy + ry][ox + rx]
    int LR;               // 16-byte pieces per ring row (piece 0 = left margin, zero-filled)
    int NP;               // DMA instructions per 16-row chunk = LR / 4
    int nvb;              // 16-row output groups per plane
    int waveLds;          // bytes of LDS per wave (staging + what the real kernel will hold)
};

constexpr int kTW = 56;

template <int NSLOT>
__global__ __launch_bounds__(512, 1) void mover_kernel(MoverArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = sgpr(tid >> 6), NS = sgpr((int)blockDim.x >> 6);
    const int pitch = p.LR * 16, slotBytes = 16 * pitch;
    unsigned char* ring = smem;
    unsigned char* stage = smem + NSLOT * slotBytes + 1024 + w * p.waveLds;      // (+1024: reads past the last row of the last slot stay inside)
    const uint32_t ringLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;

    const int planeBeg = (int)((int64_t)p.planes * blockIdx.x / gridDim.x), planeEnd = (int)((int64_t)p.planes * (blockIdx.x + 1) / gridDim.x);
    if (planeBeg >= planeEnd) return;
    const int nch = p.nvb + 1;                                                  // chunks per plane (the last one only feeds the transpose reads)
    const int totalChunks = (planeEnd - planeBeg) * nch;
    const uint32_t planeBytes = (uint32_t)p.xh * p.xw * 2u, rowBytes = (uint32_t)p.xw * 2u;

    // this wave's DMA instructions of a chunk: pieces w, w + NS, ...; lane -> (row, piece) of the chunk image
    uint32_t dOff[4]; int nMine = 0;
    #pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int piece = w + i * NS;
        dOff[i] = 0xfffffff0u;
        if (piece < p.NP)
        {
            nMine = i + 1;
            const int idx = piece * 64 + lane, row = idx / p.LR, col = idx - row * p.LR;     // col 0 = margin
            const uint32_t b = (uint32_t)(col - 1) * 16u;
            dOff[i] = (col >= 1 && b < rowBytes) ? (uint32_t)row * rowBytes + b : 0xfffffff0u;
        }
    }
    nMine = sgpr(nMine);
    auto issue_chunk = [&](int gchunk) __attribute__((always_inline))
    {
        const int pl = gchunk / nch, c = gchunk - pl * nch;
        const unsigned char* base = (const unsigned char*)(p.x) + (size_t)(planeBeg + pl) * planeBytes;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)planeBytes, 0x00020000);
        const uint32_t rowBase = (uint32_t)((p.ry + 16 * c) * (int)rowBytes);     // (negative rows wrap: out of range -> zeros)
        const uint32_t slotLds = ringLds + (uint32_t)(gchunk % NSLOT) * slotBytes;
        #pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < nMine)
            {
                const uint32_t vo = dOff[i] == 0xfffffff0u ? 0xfffffff0u : rowBase + dOff[i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void)(uintptr_t)(slotLds + (uint32_t)(w + i * NS) * 1024u), 16, vo, 0, 0, 0);
            }
    };

    // prologue: chunks 0 .. NSLOT - 2
    for (int g = 0; g < NSLOT - 1 && g < totalChunks; g++) issue_chunk(g);

    const int stripCol = 8 + w * kTW + p.rx;                                     // ring column (halves) of this strip's first pixel
    const int nvy = kTW / 8;                                                     // 16-byte vectors per output row piece
    uint32_t sink = 0;
    int pendingRows = 0, pendingPlane = 0, pendingOy = 0;
    int pl = 0, c = 0;
    for (int g = 0; g < totalChunks; g++)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // everything this wave issued is an iteration old
        __syncthreads();                                                         // chunks <= g + 1 visible; chunk g - 1 no longer read by anyone
        // deferred stores of the previous group, then the next chunk's DMA
        if (pendingRows > 0)
        {
            uint16_t* ypl = p.y + (size_t)(planeBeg + pendingPlane) * p.yh * p.yw;
            #pragma unroll
            for (int i = 0; i < 2; i++)
            {
                const int v = lane + 64 * i, row = v / nvy, col = (v - row * nvy) * 8;
                if (row < pendingRows)
                {
                    const v4u val = *reinterpret_cast<const v4u*>(stage + row * 144 + col * 2);
                    const int ox = w * kTW + col;
                    if (ox + 8 <= p.yw) *reinterpret_cast<v4u*>(ypl + (size_t)(pendingOy + row) * p.yw + ox) = val;
                    else for (int e = 0; e < 8; e++) if (ox + e < p.yw) ypl[(size_t)(pendingOy + row) * p.yw + ox + e] = (uint16_t)(val[e >> 1] >> (16 * (e & 1)));
                }
            }
            pendingRows = 0;
        }
        if (g + NSLOT - 1 < totalChunks) issue_chunk(g + NSLOT - 1);
        if (c < p.nvb)
        {
            // "stage A": the transpose reads of chunks g, g + 1 (3 column blocks of 32)
            #pragma unroll
            for (int t = 0; t < 2; t++)
            {
                const unsigned char* slot = ring + ((g + t) % NSLOT) * slotBytes;
                #pragma unroll
                for (int m = 0; m < 3; m++)
                {
                    const int gg = lane >> 5, hgrp = (lane >> 4) & 1, s = lane & 15;
                    const unsigned char* q = slot + (8 * gg + (s >> 2)) * pitch + ((8 + w * kTW) + 32 * m + 16 * hgrp + 4 * (s & 3)) * 2;
                    typedef __attribute__((address_space(3))) short4v* lds_ptr;
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(q));
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(q + 4 * pitch));
                    sink += (uint32_t)lo[0] ^ (uint32_t)hi[3];
                }
            }
            // this strip's 16 rows x 56 columns of chunk g -> staging rows
            const unsigned char* slot = ring + (g % NSLOT) * slotBytes;
            #pragma unroll
            for (int i = 0; i < 14; i++)
            {
                const int e = lane + 64 * i, row = e / kTW, col = e - row * kTW;
                *reinterpret_cast<uint16_t*>(stage + row * 144 + col * 2) = *reinterpret_cast<const uint16_t*>(slot + row * pitch + (stripCol + col) * 2);
            }
            pendingRows = min(16, p.yh - 16 * c); pendingPlane = pl; pendingOy = 16 * c;
        }
        if (++c == nch) { c = 0; ++pl; }
    }
    if (pendingRows > 0)
    {
        uint16_t* ypl = p.y + (size_t)(planeBeg + pendingPlane) * p.yh * p.yw;
        for (int i = 0; i < 2; i++)
        {
            const int v = lane + 64 * i, row = v / nvy, col = (v - row * nvy) * 8;
            if (row < pendingRows)
            {
                const v4u val = *reinterpret_cast<const v4u*>(stage + row * 144 + col * 2);
                const int ox = w * kTW + col;
                for (int e = 0; e < 8; e++) if (ox + e < p.yw) ypl[(size_t)(pendingOy + row) * p.yw + ox + e] = (uint16_t)(val[e >> 1] >> (16 * (e & 1)));
            }
        }
    }
    if (sink == 0x12345679u) p.y[0] = 1;
}

__global__ void copy_kernel(const uint4* src, uint4* dst, size_t nIn, size_t nOut)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t k = i; k < nIn; k += stride) { const uint4 v = src[k]; if (k < nOut) dst[k] = v; else { acc.x ^= v.x; acc.y ^= v.y; } }
    if (acc.x == 0x12345u && acc.y == 0x54321u) dst[0] = acc;
}

static void run_mover(const char* name, int planes, int xh, int xw, int yh, int yw, int wgPerCu, int nslot)
{
    const int NS = (yw + kTW - 1) / kTW;
    int LR = 1 + (xw * 2 + 15) / 16;
    while (LR % 8 != 4) LR++;
    MoverArgs p;
    p.planes = planes; p.xh = xh; p.xw = xw; p.yh = yh; p.yw = yw; p.ry = -3; p.rx = 1; p.LR = LR; p.NP = LR / 4; p.nvb = (yh + 15) / 16;
    if (p.NP > 4 * NS) { printf("%s: too many DMA pieces per chunk\n", name); return; }
    const size_t nx = (size_t)planes * xh * xw, ny = (size_t)planes * yh * yw;
    std::vector<uint16_t> hx(nx), hy(ny);
    uint32_t seed = 12345;
    for (size_t i = 0; i < nx; i++) { seed = seed * 1664525u + 1013904223u; hx[i] = (uint16_t)((seed >> 16) | 1); }
    uint16_t *dx, *dy;
    HIPCHK(hipMalloc(&dx, nx * 2 + 64)); HIPCHK(hipMalloc(&dy, ny * 2 + 64));
    HIPCHK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dy, 0xCD, ny * 2));
    int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int ringBytes = nslot * 16 * LR * 16 + 1024;
    // LDS per workgroup so that exactly wgPerCu fit (the real kernel: W + staging + mask rows per wave)
    const int ldsTotal = (160 * 1024) / wgPerCu / 256 * 256;
    p.waveLds = (ldsTotal - ringBytes) / NS / 16 * 16;
    if (p.waveLds < 16 * 144) { printf("%s: ring does not leave room for %d workgroups per CU\n", name, wgPerCu); return; }
    const size_t lds = (size_t)ringBytes + (size_t)NS * p.waveLds;
    const int grid = planes < ncu * wgPerCu ? planes : ncu * wgPerCu;
    p.x = dx; p.y = dy;
    auto launch = [&]()
    {
        if (nslot == 3) hipLaunchKernelGGL(mover_kernel<3>, dim3(grid), dim3(64 * NS), lds, 0, p);
        else            hipLaunchKernelGGL(mover_kernel<4>, dim3(grid), dim3(64 * NS), lds, 0, p);
    };
    if (nslot == 3) HIPCHK(hipFuncSetAttribute((const void*)mover_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    else            HIPCHK(hipFuncSetAttribute((const void*)mover_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = -1;
    if (nslot == 3) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)mover_kernel<3>, 64 * NS, lds);
    else            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)mover_kernel<4>, 64 * NS, lds);
    launch();
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(hy.data(), dy, ny * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    for (int pl = 0; pl < planes; pl += (planes > 64 ? planes / 61 : 1))
        for (int oy = 0; oy < yh; oy++)
            for (int ox = 0; ox < yw; ox++)
            {
                const int iy = oy + p.ry, ix = ox + p.rx;
                if (ix < 0 || ix >= xw) continue;
                const uint16_t want = (iy >= 0 && iy < xh) ? hx[((size_t)pl * xh + iy) * xw + ix] : 0;
                checked++;
                if (hy[((size_t)pl * yh + oy) * yw + ox] != want) { if (bad < 5) printf("   mismatch plane %d oy %d ox %d: got %04x want %04x\n", pl, oy, ox, hy[((size_t)pl * yh + oy) * yw + ox], want); bad++; }
            }
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    const int reps = 20;
    HIPCHK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)(nx + ny) * 2;
    // plain copy of the same byte count
    const size_t nIn = nx * 2 / 16, nOut = ny * 2 / 16;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(copy_kernel, dim3(ncu * 8), dim3(256), 0, 0, (const uint4*)dx, (uint4*)dy, nIn, nOut);
    HIPCHK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(copy_kernel, dim3(ncu * 8), dim3(256), 0, 0, (const uint4*)dx, (uint4*)dy, nIn, nOut);
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    float msc = 0; HIPCHK(hipEventElapsedTime(&msc, e0, e1));
    const double usc = msc * 1e3 / reps;
    printf("mover %-8s NS %d LR %d slots %d wg/CU %d (occupancy query %d) lds %zu: %zu checked, %zu bad | %7.1f us %7.1f GB/s (%.3f of 8 TB/s) | plain copy %7.1f us %7.1f GB/s\n",
           name, NS, LR, nslot, wgPerCu, occ, lds, checked, bad, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0, usc, bytes / usc / 1e3);
    HIPCHK(hipFree(dx)); HIPCHK(hipFree(dy));
}

int main(int argc, char** argv)
{
    part1();
    for (int nslot = 3; nslot <= 4; nslot++)
        for (int wg = 2; wg <= 5; wg++)
        {
            run_mover("L8", 4096, 94, 150, 92, 148, wg, nslot);
            run_mover("L13", 1024, 166, 278, 144, 256, wg, nslot);
        }
    run_mover("L4", 4096, 40, 54, 38, 52, 8, 3);
    run_mover("L6", 4096, 58, 86, 56, 84, 6, 3);
    return 0;
}
