// filtered_lrelu.hip -- fused bias -> up-FIR -> gain -> leaky ReLU -> clamp -> down-FIR
// for gfx950, plus the in-place activation step of the generic path.
//
// Behaviour follows the reference plugin (torch_utils/ops/filtered_lrelu.cu:139-1099 fused
// kernel, :1105-1211 activation kernel, filtered_lrelu.cpp:16-290 host side):
//   up stage   : u in [0, cw):  m = u + up-1-px0, i0 = floor(m/up), t0 = up-1-(m mod up)
//                upval[u] = up^2 * sum_k (x[i0+k] + b) * ffu[t0 + k*up]      (zero outside x)
//   activation : v = upval*gain; sign bit -> v *= slope; |v| > clamp -> +-clamp
//                mask (2 bits/pixel, 4 pixels/byte): 1 = negative, 2 = clamped
//   down stage : y[o] = sum_k act[o*down + k] * ffd[k]
// separably along x and y. ffu/ffd are the filters reversed unless `flip`.
//
// MI355X design (not the reference's 48 KB-shared-memory CUDA tiling):
//   * one workgroup of 512 threads (8 waves, 2 per SIMD) owns one output tile of one plane and
//     keeps all four intermediates in LDS (two ping-pong buffers, <= 70 KiB -> 2 tiles per CU);
//   * the op is VALU-bound, not HBM-bound, at 16-bit I/O (>= 72 fp32 FMAs per output pixel for
//     12/12 taps against ~5 bytes of traffic), so every FIR stage is register-blocked: a thread
//     produces 8 outputs (rows stages) or a 4-column strip (column stages) from values it loads
//     ONCE from LDS with 16-byte ds_read_b128, taps sit in registers with compile-time indices
//     (the tile's up-sampling phase is a wave-uniform template switch);
//   * filters are kernel arguments read through LDS: no global filter buffers, so calls on
//     different streams may overlap (the reference is not stream-safe, filtered_lrelu.py:215);
//   * column origin of the up-sampled tile is aligned to the sign-mask byte (4 pixels), so a
//     thread's 4-column strip is exactly one mask byte per row: no cross-lane shuffles.
//
// Roofline: HBM stream of (N_in + N_out) * sizeof(T) + mask bytes; see DESIGN.md for the VALU
// ceiling that actually binds at 16-bit I/O.

#include "lvg_common.h"
#include "filtered_lrelu_args.h"
#include <atomic>
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------
// In-place activation kernel of the generic path.

struct ActArgs
{
    void*    x;
    uint8_t* s;
    int64_t  xs[4];
    int      n, c, h, w;
    int      sWBytes, sH;   // sign plane: bytes per row, rows
    int      sOfsX, sOfsY;
    float    gain, slope, clamp;
    int      mode;
};

constexpr int kActThreads = 256;

// One thread = 4 horizontally adjacent pixels = one sign byte. Block = 64 (x) x 4 (y).
template <class T, int MODE>
__global__ __launch_bounds__(kActThreads) void filtered_lrelu_act_kernel(ActArgs p)
{
    typedef typename Elem<T>::acc_t A;
    const int xb = blockIdx.x * 64 + threadIdx.x;            // byte column (4 pixels)
    const int y  = blockIdx.y * 4 + threadIdx.y;
    const int wBytes = (MODE == LVG_SIGNS_WRITE) ? p.sWBytes : (p.w + 3) >> 2;
    const int hRows  = (MODE == LVG_SIGNS_WRITE) ? p.sH : p.h;
    if (xb >= wBytes || y >= hRows) return;
    for (int q = blockIdx.z; q < p.n * p.c; q += gridDim.z)
    {
        const int nb = q / p.c, ch = q - nb * p.c;
        T* row = (T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1] + (int64_t)y * p.xs[2];
        uint32_t bits = 0;
        #pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int xx = xb * 4 + k;
            if (xx >= p.w || y >= p.h) continue;
            T* pv = row + (int64_t)xx * p.xs[3];
            A v = (A)to_acc(*pv) * (A)p.gain;
            if (MODE == LVG_SIGNS_READ)
            {
                const int sx = xx + p.sOfsX, sy = y + p.sOfsY;
                if (sx >= 0 && sy >= 0 && (sx >> 2) < p.sWBytes && sy < p.sH)
                {
                    const uint32_t sb = p.s[((int64_t)q * p.sH + sy) * p.sWBytes + (sx >> 2)] >> ((sx & 3) << 1);
                    if (sb & 1) v *= (A)p.slope;
                    if (sb & 2) v = (A)0;
                }
            }
            else
            {
                uint32_t sb = 0;
                if (v < (A)0) { v *= (A)p.slope; sb = 1; }
                if (fabs((double)v) > (double)p.clamp) { v = (v < (A)0) ? (A)(-p.clamp) : (A)p.clamp; sb = 2; }
                bits |= sb << (k * 2);
            }
            *pv = from_acc<T>(v);
        }
        if (MODE == LVG_SIGNS_WRITE)
            p.s[((int64_t)q * p.sH + y) * p.sWBytes + xb] = (uint8_t)bits;
    }
}

template <class T>
int launch_act(ActArgs& p, hipStream_t stream)
{
    const int wBytes = (p.mode == LVG_SIGNS_WRITE) ? p.sWBytes : (p.w + 3) >> 2;
    const int hRows  = (p.mode == LVG_SIGNS_WRITE) ? p.sH : p.h;
    int64_t planes = (int64_t)p.n * p.c;
    dim3 grid((wBytes + 63) / 64, (hRows + 3) / 4, (unsigned)(planes < 65535 ? planes : 65535));
    dim3 block(64, 4);
    if (p.mode == LVG_SIGNS_WRITE)     hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, LVG_SIGNS_WRITE>), grid, block, 0, stream, p);
    else if (p.mode == LVG_SIGNS_READ) hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, LVG_SIGNS_READ>), grid, block, 0, stream, p);
    else                               hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, LVG_SIGNS_NONE>), grid, block, 0, stream, p);
    return lvg_check_launch("filtered_lrelu_act_kernel");
}

// ---------------------------------------------------------------------------------------------
// Fused kernel.


constexpr int kNT = 512; // threads per workgroup

constexpr int cround(int v, int m) { return (v + m - 1) / m * m; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// LDS strides are chosen so that with LANES <-> ROWS every 16-byte access is conflict-free:
// reads (ds_read_b128, 64 banks) need stride/4 odd, writes (32 banks) need stride = 4 (mod 32) or
// consecutive addresses. oddq(v) = smallest multiple of 4 >= v whose quarter is odd.
constexpr int oddq(int v) { int r = cround(v, 4); return ((r / 4) % 2 == 1) ? r : r + 4; }
constexpr int wstride(int v) { int r = cround(v, 4); while (r % 32 != 4 && r % 32 != 12 && r % 32 != 20 && r % 32 != 28) r += 4; return r; }

template <int UP, int DOWN, int FU, int FD, int TW, int TH>
struct Geom
{
    static constexpr int TW_   = TW;
    static constexpr int TH_   = TH;
    static constexpr int KU    = FU / UP;                        // taps per output of the up stage
    static constexpr int QX    = (UP == 4) ? 2 : (UP == 2 ? 4 : 8); // input columns per stage-A item
    static constexpr int OUTA  = QX * UP;                        // outputs per stage-A item (8)
    static constexpr int QY    = (UP == 4) ? 2 : (UP == 2 ? 4 : 8); // input rows advanced per stage-B item
    static constexpr int RB    = QY * UP;                        // output rows per stage-B item (8)
    static constexpr int PC    = 4;                              // outputs per stage-C item
    static constexpr int RD    = 4;                              // output rows per stage-D item
    static constexpr int UPW_N = TW * DOWN + FD - DOWN;          // up-sampled columns the tile needs
    static constexpr int UPW_A = cround(UPW_N + 3, 8);           // columns computed (mask-byte aligned origin)
    static constexpr int UPH   = TH * DOWN + FD - DOWN;          // up-sampled rows
    static constexpr int NRG   = (UPH + RB - 1) / RB;            // stage-B row groups
    static constexpr int INH   = (UP - 1 + UPH - 1) / UP + KU;   // input rows needed
    static constexpr int INH_A = NRG * QY + KU + 1;              // rows of the row-filtered tile stage B may touch
    static constexpr int NINA  = cround(QX + KU, (QX % 4 == 0) ? 4 : 2);
    static constexpr int NINC  = cround(PC * DOWN + FD - DOWN, 4);
    static constexpr int NIND  = cround(RD * DOWN + FD - DOWN, 4);
    // row strides (floats)
    static constexpr int INW_S  = oddq(UPW_A / UP + KU + 4);            // input tile            [INH][INW_S]
    static constexpr int UPX_S  = wstride(UPW_A);                       // row-filtered tile     [INH_A][UPX_S]
    static constexpr int UPW_S  = oddq(cround(UPW_N, 4) + 4);           // activated tile        [UPH][UPW_S]
    static constexpr int DNT_S  = oddq((TH - RD) * DOWN + NIND);        // decimated, TRANSPOSED [TW][DNT_S]
    static constexpr int SZ_IN   = INH * INW_S + 8;
    static constexpr int SZ_UPX  = INH_A * UPX_S;
    static constexpr int SZ_UPXY = UPH * UPW_S + 8;
    static constexpr int SZ_DWNT = TW * DNT_S + 8;
    static constexpr int TAPS  = cround(FU + FD, 4);
    static constexpr int BUF0  = cround(cmax(SZ_IN, SZ_UPXY), 4);
    static constexpr int BUF1  = cround(cmax(SZ_UPX, SZ_DWNT), 4);
    static constexpr int LDS_BYTES = (TAPS + BUF0 + BUF1) * 4;
    static_assert(FU % UP == 0 && FD % DOWN == 0, "filter sizes must be multiples of the rates");
    static_assert(TW % 4 == 0 && (TW * DOWN) % 8 == 0 && TH % RD == 0, "tile alignment");
    static_assert(UPW_A % OUTA == 0, "stage A item width");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS overflow");
};

__device__ __forceinline__ float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Stage A: rows of the up stage. sIn[INH][INW_S] -> sUpX[INH][UPX_S]. PH = (m of column 0) mod UP.
// Lanes run over rows (row fastest): reads and writes are 16-byte, conflict-free by stride choice.
template <class G, int UP, int FU, int PH>
__device__ __forceinline__ void stage_up_rows(const float* __restrict__ sIn, float* __restrict__ sUpX, const float* taps, int tid)
{
    float ft[FU];
    #pragma unroll
    for (int i = 0; i < FU; i++) ft[i] = uniform(taps[i]);
    constexpr int NJ = G::UPW_A / G::OUTA;
    for (int item = tid; item < G::INH * NJ; item += kNT)
    {
        const int jb = item / G::INH, row = item - jb * G::INH;
        const float* src = sIn + row * G::INW_S + jb * G::QX;
        float in[G::NINA];
        if (G::QX % 4 == 0)
        {
            #pragma unroll
            for (int i = 0; i < G::NINA / 4; i++) { const float4 t = *(const float4*)(src + 4 * i); in[4*i] = t.x; in[4*i+1] = t.y; in[4*i+2] = t.z; in[4*i+3] = t.w; }
        }
        else
        {
            #pragma unroll
            for (int i = 0; i < G::NINA / 2; i++) { const float2 t = *(const float2*)(src + 2 * i); in[2*i] = t.x; in[2*i+1] = t.y; }
        }
        float out[G::OUTA];
        #pragma unroll
        for (int r = 0; r < G::OUTA; r++)
        {
            const int irel = (PH + r) / UP;
            const int t0 = UP - 1 - (PH + r) % UP;
            float acc = 0.0f;
            #pragma unroll
            for (int k = 0; k < G::KU; k++) acc = fmaf(in[irel + k], ft[t0 + k * UP], acc);
            out[r] = acc;
        }
        float* dst = sUpX + row * G::UPX_S + jb * G::OUTA;
        #pragma unroll
        for (int i = 0; i < G::OUTA / 4; i++) *(float4*)(dst + 4 * i) = make_float4(out[4*i], out[4*i+1], out[4*i+2], out[4*i+3]);
    }
}

// Stage B: columns of the up stage + activation + mask. sUpX -> sUpXY[UPH][UPW_S] (column 0 = the
// tile's first needed pixel). A thread owns a 4-column strip (= one mask byte per row) and RB rows.
// Lanes run over strips: every access is 16 consecutive bytes per lane.
template <class G, int UP, int FU, int PH, int MODE>
__device__ __forceinline__ void stage_up_cols_act(const float* __restrict__ sUpX, float* __restrict__ sUpXY, const float* taps,
                                                  const FlreluArgs& p, int tid, int rOff, int signByte0, int signY0,
                                                  int64_t signPlane, int ownCols, int ownRows)
{
    float ft[FU];
    const float scale = (float)(UP * UP) * p.gain;
    #pragma unroll
    for (int i = 0; i < FU; i++) ft[i] = uniform(taps[i]) * scale;
    constexpr int NCG = G::UPW_A / 4;
    constexpr int NROWS = G::QY + G::KU;
    const float slope = p.slope, clamp = p.clamp;
    for (int item = tid; item < G::NRG * NCG; item += kNT)
    {
        const int rg = item / NCG, cg = item - rg * NCG;
        const int v0 = rg * G::RB;
        const float* src = sUpX + (v0 / UP) * G::UPX_S + cg * 4;
        float4 in[NROWS];
        #pragma unroll
        for (int i = 0; i < NROWS; i++) in[i] = *(const float4*)(src + i * G::UPX_S);
        const int byteX = signByte0 + cg;
        const bool byteOk = byteX >= 0 && byteX < p.swLimit;
        #pragma unroll
        for (int rr = 0; rr < G::RB; rr++)
        {
            const int v = v0 + rr;
            if (v >= G::UPH) break;
            const int irel = (PH + rr) / UP;
            const int t0 = UP - 1 - (PH + rr) % UP;
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            #pragma unroll
            for (int k = 0; k < G::KU; k++)
            {
                const float w = ft[t0 + k * UP];
                a[0] = fmaf(in[irel + k].x, w, a[0]); a[1] = fmaf(in[irel + k].y, w, a[1]);
                a[2] = fmaf(in[irel + k].z, w, a[2]); a[3] = fmaf(in[irel + k].w, w, a[3]);
            }
            const int sy = signY0 + v;
            if (MODE == LVG_SIGNS_READ)
            {
                if (byteOk && sy >= 0 && sy < p.sH)
                {
                    const uint32_t sb = p.s[signPlane + (int64_t)sy * p.sWBytes + byteX];
                    #pragma unroll
                    for (int c4 = 0; c4 < 4; c4++)
                    {
                        const uint32_t q = sb >> (c4 * 2);
                        if (q & 1) a[c4] *= slope;
                        if (q & 2) a[c4] = 0.0f;
                    }
                }
            }
            else
            {
                uint32_t bits = 0;
                #pragma unroll
                for (int c4 = 0; c4 < 4; c4++)
                {
                    uint32_t sb = __float_as_uint(a[c4]) >> 31;      // IEEE sign bit (-0.0 counts)
                    if (sb) a[c4] *= slope;
                    if (fabsf(a[c4]) > clamp) { sb = 2; a[c4] = (a[c4] < 0.0f) ? -clamp : clamp; }
                    bits |= sb << (c4 * 2);
                }
                if (MODE == LVG_SIGNS_WRITE)
                {
                    if (cg * 4 < ownCols && v < ownRows && byteOk && sy >= 0 && sy < p.sH)
                        p.s[signPlane + (int64_t)sy * p.sWBytes + byteX] = (uint8_t)bits;
                }
            }
            float* dst = sUpXY + v * G::UPW_S + cg * 4 - rOff;
            if (rOff == 0)
            {
                if (cg * 4 + 4 <= G::UPW_S) *(float4*)dst = make_float4(a[0], a[1], a[2], a[3]);
            }
            else
            {
                #pragma unroll
                for (int c4 = 0; c4 < 4; c4++)
                {
                    const int col = cg * 4 + c4 - rOff;
                    if (col >= 0 && col < G::UPW_S) dst[c4] = a[c4];
                }
            }
        }
    }
}

// Stage C: rows of the down stage, decimating. sUpXY[UPH][UPW_S] -> sDownT[TW][DNT_S] (TRANSPOSED, so
// that the column stage also filters along a contiguous axis). Lanes run over rows.
template <class G, int DOWN, int FD>
__device__ __forceinline__ void stage_down_rows(const float* __restrict__ sUpXY, float* __restrict__ sDownT, const float* taps, int tid)
{
    float ft[FD];
    #pragma unroll
    for (int i = 0; i < FD; i++) ft[i] = uniform(taps[i]);
    constexpr int NOG = G::TW_ / G::PC;
    for (int item = tid; item < G::UPH * NOG; item += kNT)
    {
        const int og = item / G::UPH, row = item - og * G::UPH;
        const float* src = sUpXY + row * G::UPW_S + og * G::PC * DOWN;
        float in[G::NINC];
        #pragma unroll
        for (int i = 0; i < G::NINC / 4; i++) { const float4 t = *(const float4*)(src + 4 * i); in[4*i] = t.x; in[4*i+1] = t.y; in[4*i+2] = t.z; in[4*i+3] = t.w; }
        #pragma unroll
        for (int o = 0; o < G::PC; o++)
        {
            float acc = 0.0f;
            #pragma unroll
            for (int k = 0; k < FD; k++) acc = fmaf(in[o * DOWN + k], ft[k], acc);
            sDownT[(og * G::PC + o) * G::DNT_S + row] = acc;
        }
    }
}

// Stage D: columns of the down stage, decimating, store. sDownT[TW][DNT_S] -> y. A thread produces RD
// consecutive output rows of one column; lanes run over columns, so every global store instruction
// writes 64 consecutive pixels of one output row.
template <class T, class G, int DOWN, int FD>
__device__ __forceinline__ void stage_down_cols_store(const float* __restrict__ sDownT, const float* taps, const FlreluArgs& p,
                                                      int tid, T* __restrict__ yp, int outX0, int outY0)
{
    float ft[FD];
    #pragma unroll
    for (int i = 0; i < FD; i++) ft[i] = uniform(taps[i]);
    constexpr int NRGD = G::TH_ / G::RD;
    for (int item = tid; item < G::TW_ * NRGD; item += kNT)
    {
        const int rgd = item / G::TW_, col = item - rgd * G::TW_;
        const int ox = outX0 + col;
        const int oy0 = outY0 + rgd * G::RD;
        if (ox >= p.yw || oy0 >= p.yh) continue;
        const float* src = sDownT + col * G::DNT_S + rgd * G::RD * DOWN;
        float in[G::NIND];
        #pragma unroll
        for (int i = 0; i < G::NIND / 4; i++) { const float4 t = *(const float4*)(src + 4 * i); in[4*i] = t.x; in[4*i+1] = t.y; in[4*i+2] = t.z; in[4*i+3] = t.w; }
        T* dst = yp + (int64_t)oy0 * p.ys[2] + (int64_t)ox * p.ys[3];
        #pragma unroll
        for (int o = 0; o < G::RD; o++)
        {
            float acc = 0.0f;
            #pragma unroll
            for (int k = 0; k < FD; k++) acc = fmaf(in[o * DOWN + k], ft[k], acc);
            if (oy0 + o < p.yh) dst[(int64_t)o * p.ys[2]] = from_acc<T>(acc);
        }
    }
}

template <int UP_, int DOWN_, int FU_, int FD_, int TW__, int TH__>
struct GeomX : Geom<UP_, DOWN_, FU_, FD_, TW__, TH__> {};

template <class T, int UP, int DOWN, int FU, int FD, int TW, int TH, int MODE>
__global__ __launch_bounds__(kNT) void filtered_lrelu_fused_kernel(FlreluArgs p)
{
    typedef GeomX<UP, DOWN, FU, FD, TW, TH> G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* taps = smem;               // [0, FU): flipped up taps, [FU, FU+FD): flipped down taps
    float* buf0 = smem + G::TAPS;
    float* buf1 = buf0 + G::BUF0;
    const int tid = threadIdx.x;

    // Tile / plane of this workgroup (x fastest: neighbours share halos in cache).
    int bid = blockIdx.x;
    const int tileX = bid % p.tilesX; bid /= p.tilesX;
    const int tileY = bid % p.tilesY; bid /= p.tilesY;
    const int ch = bid % p.c;
    const int nb = bid / p.c;
    const int64_t plane = (int64_t)nb * p.c + ch;

    if (tid < FU)
    {
        float v = 0.0f;
        if (tid < p.fuN) v = p.fu ? p.fu[p.flip ? tid : p.fuN - 1 - tid] : 1.0f;
        taps[tid] = v;
    }
    else if (tid < FU + FD)
    {
        const int t = tid - FU;
        float v = 0.0f;
        if (t < p.fdN) v = p.fd ? p.fd[p.flip ? t : p.fdN - 1 - t] : 1.0f;
        taps[FU + t] = v;
    }

    // Geometry (all wave-uniform).
    const int outX0 = tileX * TW, outY0 = tileY * TH;
    const int upX0 = outX0 * DOWN, upY0 = outY0 * DOWN;              // first needed up-sampled pixel
    const int rOff = (MODE == LVG_SIGNS_READ) ? ((upX0 + p.sOfsX) & 3) : 0; // (x & 3) is the non-negative residue
    const int uStart = upX0 - rOff;                                  // column 0 of the computed tile
    const int mX0 = uStart + UP - 1 - p.px0;
    const int mY0 = upY0 + UP - 1 - p.py0;
    const int inX0 = lvg_floor_div(mX0, UP), inY0 = lvg_floor_div(mY0, UP);
    const int phX = mX0 - inX0 * UP, phY = mY0 - inY0 * UP;

    // Input tile + bias into buf0 (zero outside the image; bias only on real pixels).
    {
        const T* xp = (const T*)p.x + (int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1];
        const float bias = (float)to_acc(((const T*)p.b)[ch]);
        constexpr int TOTAL = G::INH * G::INW_S;
        constexpr int PER = (TOTAL + kNT - 1) / kNT;
        float v[PER];
        #pragma unroll
        for (int i = 0; i < PER; i++)
        {
            const int idx = tid + i * kNT;
            const int r = idx / G::INW_S, q = idx - r * G::INW_S;
            const int iy = inY0 + r, ix = inX0 + q;
            v[i] = 0.0f;
            if (idx < TOTAL && iy >= 0 && iy < p.xh && ix >= 0 && ix < p.xw)
                v[i] = (float)to_acc(xp[(int64_t)iy * p.xs[2] + (int64_t)ix * p.xs[3]]) + bias;
        }
        #pragma unroll
        for (int i = 0; i < PER; i++)
        {
            const int idx = tid + i * kNT;
            if (idx < TOTAL) buf0[idx] = v[i];
        }
    }
    __syncthreads();

    // A: rows of the up stage (buf0 -> buf1).
    if (UP == 1)      stage_up_rows<G, UP, FU, 0>(buf0, buf1, taps, tid);
    else if (UP == 2) { if (phX == 0) stage_up_rows<G, UP, FU, 0>(buf0, buf1, taps, tid); else stage_up_rows<G, UP, FU, 1 % UP>(buf0, buf1, taps, tid); }
    else
    {
        if (phX == 0)      stage_up_rows<G, UP, FU, 0>(buf0, buf1, taps, tid);
        else if (phX == 1) stage_up_rows<G, UP, FU, 1 % UP>(buf0, buf1, taps, tid);
        else if (phX == 2) stage_up_rows<G, UP, FU, 2 % UP>(buf0, buf1, taps, tid);
        else               stage_up_rows<G, UP, FU, 3 % UP>(buf0, buf1, taps, tid);
    }
    __syncthreads();

    // B: columns of the up stage + activation + mask (buf1 -> buf0).
    {
        const int signByte0 = (uStart + p.sOfsX) >> 2;   // exact: uStart + sOfsX is a multiple of 4
        const int signY0 = upY0 + p.sOfsY;
        const int64_t signPlane = plane * (int64_t)p.sH * p.sWBytes;
        const int ownCols = (tileX == p.tilesX - 1) ? G::UPW_A : TW * DOWN;
        const int ownRows = (tileY == p.tilesY - 1) ? G::UPH : TH * DOWN;
        if (UP == 1)      stage_up_cols_act<G, UP, FU, 0, MODE>(buf1, buf0, taps, p, tid, rOff, signByte0, signY0, signPlane, ownCols, ownRows);
        else if (UP == 2) { if (phY == 0) stage_up_cols_act<G, UP, FU, 0, MODE>(buf1, buf0, taps, p, tid, rOff, signByte0, signY0, signPlane, ownCols, ownRows);
                            else          stage_up_cols_act<G, UP, FU, 1 % UP, MODE>(buf1, buf0, taps, p, tid, rOff, signByte0, signY0, signPlane, ownCols, ownRows); }
        else
        {
            if (phY == 0)      stage_up_cols_act<G, UP, FU, 0, MODE>(buf1, buf0, taps, p, tid, rOff, signByte0, signY0, signPlane, ownCols, ownRows);
            else if (phY == 1) stage_up_cols_act<G, UP, FU, 1 % UP, MODE>(buf1, buf0, taps, p, tid, rOff, signByte0, signY0, signPlane, ownCols, ownRows);
            else if (phY == 2) stage_up_cols_act<G, UP, FU, 2 % UP, MODE>(buf1, buf0, taps, p, tid, rOff, signByte0, signY0, signPlane, ownCols, ownRows);
            else               stage_up_cols_act<G, UP, FU, 3 % UP, MODE>(buf1, buf0, taps, p, tid, rOff, signByte0, signY0, signPlane, ownCols, ownRows);
        }
        // Bytes of the 16-pixel row padding carry no pixels: define them as 0.
        if (MODE == LVG_SIGNS_WRITE && tileX == p.tilesX - 1 && p.sWBytes > p.swLimit)
        {
            const int padBytes = p.sWBytes - p.swLimit;
            for (int idx = tid; idx < ownRows * padBytes; idx += kNT)
            {
                const int v = idx / padBytes, k = idx - v * padBytes;
                const int sy = signY0 + v;
                if (sy >= 0 && sy < p.sH) p.s[signPlane + (int64_t)sy * p.sWBytes + p.swLimit + k] = 0;
            }
        }
    }
    __syncthreads();

    // C: rows of the down stage (buf0 -> buf1).
    stage_down_rows<G, DOWN, FD>(buf0, buf1, taps + FU, tid);
    __syncthreads();

    // D: columns of the down stage, store.
    T* yp = (T*)p.y + (int64_t)nb * p.ys[0] + (int64_t)ch * p.ys[1];
    stage_down_cols_store<T, G, DOWN, FD>(buf1, taps + FU, p, tid, yp, outX0, outY0);
}

// 1x1 / no-resampling case (ToRGB layers: up = down = 1, one tap each): a plain stream.
template <class T, int MODE>
__global__ __launch_bounds__(256) void filtered_lrelu_pointwise_kernel(FlreluArgs p)
{
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const bool live = ox < p.yw;     // no early exit: the mask byte is assembled with lane shuffles
    const int oy = blockIdx.y;
    const float fu = p.fu ? p.fu[0] : 1.0f, fd = p.fd ? p.fd[0] : 1.0f;
    const float pre = fu * fu * p.gain, post = fd * fd;
    for (int q = blockIdx.z; q < p.n * p.c; q += gridDim.z)
    {
        const int nb = q / p.c, ch = q - nb * p.c;
        const int ix = ox - p.px0, iy = oy - p.py0;
        float v = 0.0f;
        if (live && ix >= 0 && ix < p.xw && iy >= 0 && iy < p.xh)
            v = (float)to_acc(((const T*)p.x)[(int64_t)nb * p.xs[0] + (int64_t)ch * p.xs[1] + (int64_t)iy * p.xs[2] + (int64_t)ix * p.xs[3]])
                + (float)to_acc(((const T*)p.b)[ch]);
        v *= pre;
        const int sx = ox + p.sOfsX, sy = oy + p.sOfsY;
        const bool inMask = live && sx >= 0 && sy >= 0 && (sx >> 2) < p.swLimit && sy < p.sH;
        const int64_t sIdx = ((int64_t)q * p.sH + sy) * p.sWBytes + (sx >> 2);
        if (MODE == LVG_SIGNS_READ)
        {
            if (inMask)
            {
                const uint32_t sb = p.s[sIdx] >> ((sx & 3) << 1);
                if (sb & 1) v *= p.slope;
                if (sb & 2) v = 0.0f;
            }
        }
        else
        {
            uint32_t sb = __float_as_uint(v) >> 31;
            if (sb) v *= p.slope;
            if (fabsf(v) > p.clamp) { sb = 2; v = (v < 0.0f) ? -p.clamp : p.clamp; }
            if (MODE == LVG_SIGNS_WRITE)
            {
                // 4 neighbouring lanes hold the 4 pixels of one byte (sOfs is 0 when writing and the
                // block's x origin is a multiple of 4)
                uint32_t bits = live ? (sb << ((sx & 3) << 1)) : 0u;
                bits |= (uint32_t)__shfl_xor((int)bits, 1);
                bits |= (uint32_t)__shfl_xor((int)bits, 2);
                if ((sx & 3) == 0 && inMask) p.s[sIdx] = (uint8_t)bits;
            }
        }
        if (live)
            ((T*)p.y)[(int64_t)nb * p.ys[0] + (int64_t)ch * p.ys[1] + (int64_t)oy * p.ys[2] + (int64_t)ox * p.ys[3]] = from_acc<T>(v * post);
    }
}

// ---------------------------------------------------------------------------------------------
// Host side: specialisation table.

enum { CFG_NONE = LVG_FLRELU_CFG_NONE, CFG_POINTWISE = LVG_FLRELU_CFG_POINTWISE, CFG_U2D2 = LVG_FLRELU_CFG_U2D2,
       CFG_U4D2 = LVG_FLRELU_CFG_U4D2, CFG_U2D4 = LVG_FLRELU_CFG_U2D4 };

int pick_config(int fuN, int fdN, int up, int down)
{
    if (up == 1 && down == 1 && fuN == 1 && fdN == 1) return CFG_POINTWISE;
    if (up == 2 && down == 2 && fuN <= 12 && fdN <= 12 && fuN >= 2 && fdN >= 2) return CFG_U2D2;
    if (up == 4 && down == 2 && fuN <= 24 && fdN <= 12 && fuN >= 4 && fdN >= 2) return CFG_U4D2;
    if (up == 2 && down == 4 && fuN <= 12 && fdN <= 24 && fuN >= 2 && fdN >= 4) return CFG_U2D4;
    return CFG_NONE;
}

template <class T, int UP, int DOWN, int FU, int FD, int TW, int TH>
int launch_fused(FlreluArgs& p, int mode, hipStream_t stream)
{
    typedef GeomX<UP, DOWN, FU, FD, TW, TH> G;
    p.tilesX = (p.yw + TW - 1) / TW;
    p.tilesY = (p.yh + TH - 1) / TH;
    const int64_t blocks = (int64_t)p.tilesX * p.tilesY * p.n * p.c;
    LVG_REQUIRE(blocks <= 0x7fffffffLL, "filtered_lrelu: too many tiles for one launch");
    const size_t lds = G::LDS_BYTES;
    // Opt in to > 64 KiB of dynamic LDS once per specialisation.
    #define LVG_FLRELU_LAUNCH(M) do { \
        int dev_ = 0; (void)hipGetDevice(&dev_); \
        static std::atomic<uint64_t> attr_done{0};          /* one bit per device (the attribute is per device) */ \
        const uint64_t bit_ = 1ull << (dev_ & 63); \
        if (!(attr_done.load(std::memory_order_acquire) & bit_)) { \
            hipError_t e = hipFuncSetAttribute((const void*)filtered_lrelu_fused_kernel<T, UP, DOWN, FU, FD, TW, TH, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) { lvg_set_error("filtered_lrelu: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return LVG_ERR_LAUNCH; } \
            attr_done.fetch_or(bit_, std::memory_order_release); } \
        hipLaunchKernelGGL((filtered_lrelu_fused_kernel<T, UP, DOWN, FU, FD, TW, TH, M>), dim3((unsigned)blocks), dim3(kNT), lds, stream, p); } while (0)
    if (mode == LVG_SIGNS_WRITE)     LVG_FLRELU_LAUNCH(LVG_SIGNS_WRITE);
    else if (mode == LVG_SIGNS_READ) LVG_FLRELU_LAUNCH(LVG_SIGNS_READ);
    else                             LVG_FLRELU_LAUNCH(LVG_SIGNS_NONE);
    #undef LVG_FLRELU_LAUNCH
    return lvg_check_launch("filtered_lrelu_fused_kernel");
}

template <class T>
int launch_pointwise(FlreluArgs& p, int mode, hipStream_t stream)
{
    const int64_t planes = (int64_t)p.n * p.c;
    dim3 grid((p.yw + 255) / 256, p.yh, (unsigned)(planes < 65535 ? planes : 65535));
    LVG_REQUIRE(p.yh <= 65535, "filtered_lrelu: image too tall for the pointwise kernel");
    if (mode == LVG_SIGNS_WRITE)     hipLaunchKernelGGL((filtered_lrelu_pointwise_kernel<T, LVG_SIGNS_WRITE>), grid, dim3(256), 0, stream, p);
    else if (mode == LVG_SIGNS_READ) hipLaunchKernelGGL((filtered_lrelu_pointwise_kernel<T, LVG_SIGNS_READ>), grid, dim3(256), 0, stream, p);
    else                             hipLaunchKernelGGL((filtered_lrelu_pointwise_kernel<T, LVG_SIGNS_NONE>), grid, dim3(256), 0, stream, p);
    return lvg_check_launch("filtered_lrelu_pointwise_kernel");
}

template <class T>
int run_fused(FlreluArgs& p, int cfg, int mode, hipStream_t stream)
{
    switch (cfg)
    {
        case CFG_POINTWISE: return launch_pointwise<T>(p, mode, stream);
        // (narrow planes -- the float32 layers of the super-resolution generator are 36 x 29 -- take a 40-column tile: one tile
        // per plane either way, two thirds of the up-sampled columns)
        case CFG_U2D2:      return p.yw <= 40 ? launch_fused<T, 2, 2, 12, 12, 40, 32>(p, mode, stream) : launch_fused<T, 2, 2, 12, 12, 64, 32>(p, mode, stream);
        case CFG_U4D2:      return launch_fused<T, 4, 2, 24, 12, 64, 32>(p, mode, stream);
        case CFG_U2D4:      return launch_fused<T, 2, 4, 12, 24, 32, 16>(p, mode, stream);
    }
    return LVG_ERR_UNSUPPORTED;
}

} // namespace

extern "C" int lvg_filtered_lrelu_act(void* x, uint8_t* s, const int64_t xshape[4], const int64_t xstride[4],
                                      const int64_t sshape[2], int sofs_x, int sofs_y,
                                      float gain, float slope, float clamp, int sign_mode,
                                      int dtype, void* stream)
{
    LVG_REQUIRE(x, "filtered_lrelu_act: x must not be NULL");
    LVG_REQUIRE(dtype >= LVG_F32 && dtype <= LVG_F64, "filtered_lrelu_act: unknown dtype %d", dtype);
    LVG_REQUIRE(sign_mode >= LVG_SIGNS_NONE && sign_mode <= LVG_SIGNS_READ, "filtered_lrelu_act: bad sign mode");
    for (int i = 0; i < 4; i++) LVG_REQUIRE(xshape[i] >= 1 && xshape[i] <= 0x7fffffffLL, "filtered_lrelu_act: x is empty or too large");
    ActArgs p;
    p.x = x; p.s = s;
    for (int i = 0; i < 4; i++) p.xs[i] = xstride[i];
    p.n = (int)xshape[0]; p.c = (int)xshape[1]; p.h = (int)xshape[2]; p.w = (int)xshape[3];
    p.sWBytes = 0; p.sH = 0;
    if (sign_mode != LVG_SIGNS_NONE)
    {
        LVG_REQUIRE(s, "filtered_lrelu_act: sign tensor missing");
        LVG_REQUIRE(sshape[0] >= 1 && sshape[1] >= 1 && sshape[0] <= 0x1fffffffLL && sshape[1] <= 0x7fffffffLL, "filtered_lrelu_act: signs tensor is too large");
        p.sWBytes = (int)sshape[0]; p.sH = (int)sshape[1];
        if (sign_mode == LVG_SIGNS_WRITE)
            LVG_REQUIRE(p.sH == p.h && p.sWBytes * 4 >= p.w, "filtered_lrelu_act: sign plane must cover x");
    }
    p.sOfsX = sofs_x; p.sOfsY = sofs_y;
    p.gain = gain; p.slope = slope; p.clamp = clamp; p.mode = sign_mode;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype)
    {
        case LVG_F32:  return launch_act<float>(p, st);
        case LVG_F16:  return launch_act<f16_t>(p, st);
        case LVG_BF16: return launch_act<bf16_t>(p, st);
        default:       return launch_act<double>(p, st);
    }
}

static std::atomic<int> g_flrelu_impl{0};      // process-wide on purpose (include/lvg_test_hooks.h): the backward launch comes from autograd's thread, not the test's

extern "C" int lvg_filtered_lrelu_set_impl(int impl)
{
    if (impl < 0 || impl > 5) return LVG_ERR_INVALID;     // 0 default, 1 fp32-VALU kernel, 2 round-2 MFMA kernel, 3 wave-per-tile MFMA kernel, 4 row-band MFMA kernel (falls back to 3 for what it does not take), 5 strip kernel (round 6; falls back likewise)
    return g_flrelu_impl.exchange(impl);
}

extern "C" int lvg_filtered_lrelu_supported(int fu_n, int fd_n, int up, int down, int dtype)
{
    if (dtype != LVG_F32 && dtype != LVG_F16 && dtype != LVG_BF16) return 0;
    return pick_config(fu_n, fd_n, up, down) != CFG_NONE;
}

extern "C" int lvg_filtered_lrelu(const void* x, void* y, const void* b, uint8_t* s,
                                  const float* fu, const float* fd,
                                  const int64_t xshape[4], const int64_t xstride[4],
                                  const int64_t yshape[4], const int64_t ystride[4],
                                  int fu_n, int fd_n, int up, int down, int px0, int py0,
                                  const int64_t sshape[2], int sofs_x, int sofs_y, int sw_active,
                                  float gain, float slope, float clamp, int flip, int sign_mode,
                                  int dtype, void* stream)
{
    LVG_REQUIRE(x && y && b, "filtered_lrelu: x, y and b must not be NULL");
    LVG_REQUIRE(up >= 1 && down >= 1, "filtered_lrelu: up and down must be at least 1");
    LVG_REQUIRE(fu_n >= 1 && fd_n >= 1, "filtered_lrelu: fu and fd must not be empty");
    LVG_REQUIRE(sign_mode >= LVG_SIGNS_NONE && sign_mode <= LVG_SIGNS_READ, "filtered_lrelu: bad sign mode");
    LVG_REQUIRE(fu || fu_n == 1, "filtered_lrelu: fu is NULL but has %d taps", fu_n);
    LVG_REQUIRE(fd || fd_n == 1, "filtered_lrelu: fd is NULL but has %d taps", fd_n);
    for (int i = 0; i < 4; i++)
    {
        LVG_REQUIRE(xshape[i] >= 1 && xshape[i] <= 0x7fffffffLL, "filtered_lrelu: x is empty or too large");
        LVG_REQUIRE(yshape[i] >= 1 && yshape[i] <= 0x7fffffffLL, "filtered_lrelu: output must be at least 1x1");
    }
    LVG_REQUIRE(xshape[0] == yshape[0] && xshape[1] == yshape[1], "filtered_lrelu: batch/channel mismatch between x and y");
    const int cfg = (dtype == LVG_F32 || dtype == LVG_F16 || dtype == LVG_BF16) ? pick_config(fu_n, fd_n, up, down) : CFG_NONE;
    if (cfg == CFG_NONE)
    {
        lvg_set_error("filtered_lrelu: no fused kernel for up=%d down=%d taps=%d/%d dtype=%d", up, down, fu_n, fd_n, dtype);
        return LVG_ERR_UNSUPPORTED;
    }

    FlreluArgs p;
    p.x = x; p.y = y; p.b = b; p.s = s; p.fu = fu; p.fd = fd;
    for (int i = 0; i < 4; i++) { p.xs[i] = xstride[i]; p.ys[i] = ystride[i]; }
    p.n = (int)xshape[0]; p.c = (int)xshape[1]; p.xh = (int)xshape[2]; p.xw = (int)xshape[3];
    p.yh = (int)yshape[2]; p.yw = (int)yshape[3];
    p.fuN = fu_n; p.fdN = fd_n; p.px0 = px0; p.py0 = py0;
    p.sWBytes = 0; p.sH = 0; p.swLimit = 0;
    if (sign_mode != LVG_SIGNS_NONE)
    {
        LVG_REQUIRE(s, "filtered_lrelu: sign tensor missing");
        LVG_REQUIRE(sshape[0] >= 1 && sshape[1] >= 1 && sshape[0] <= 0x1fffffffLL && sshape[1] <= 0x7fffffffLL, "filtered_lrelu: signs is too large");
        p.sWBytes = (int)sshape[0]; p.sH = (int)sshape[1];
        p.swLimit = (sw_active + 3) >> 2;
        if (p.swLimit > p.sWBytes) p.swLimit = p.sWBytes;
        if (sign_mode == LVG_SIGNS_WRITE) LVG_REQUIRE(sofs_x == 0 && sofs_y == 0, "filtered_lrelu: sign offsets must be 0 when writing the mask");
    }
    p.sOfsX = sofs_x; p.sOfsY = sofs_y;
    p.gain = gain; p.slope = slope; p.clamp = clamp; p.flip = flip ? 1 : 0;
    p.tilesX = p.tilesY = 0;
    {
        // extent of the x tensor around its first element (the wave kernel's 16-byte loads may touch up to 14 bytes outside
        // a plane, never outside the tensor)
        const int64_t es = dtype == LVG_F32 ? 4 : (dtype == LVG_F64 ? 8 : 2);
        int64_t lo = 0, hi = 0;
        for (int i = 0; i < 4; i++) { const int64_t e = (xshape[i] - 1) * xstride[i]; if (e < 0) lo -= e; else hi += e; }
        p.xLoB = lo * es; p.xHiB = (hi + 1) * es;
    }

    hipStream_t st = (hipStream_t)stream;
    // 16-bit I/O: the MFMA kernel (filtered_lrelu_mfma.hip). LVG_FLRELU_MFMA=0 keeps the fp32-VALU kernel
    // for every dtype (A/B measurements, bisecting); float32 I/O always uses it (exact f32 intermediates).
    static const bool env_mfma = []() { const char* e = getenv("LVG_FLRELU_MFMA"); return !(e && e[0] == '0'); }();
    // LVG_FLRELU_WAVE=0 selects the round-2 kernel (four waves per tile) instead of the wave-per-tile kernel (A/B measurements).
    static const bool env_wave = []() { const char* e = getenv("LVG_FLRELU_WAVE"); return !(e && e[0] == '0'); }();
    const int impl = g_flrelu_impl.load(std::memory_order_relaxed);
    const bool use_mfma = impl == 0 ? env_mfma : impl >= 2;
    const bool use_wave = impl == 0 ? env_wave : impl >= 3;
    // The row-band kernel (round 5, float16) takes the planes it measured faster on (its launcher decides: planes of two or three
    // column strips, profiles/r05_band_*.log); LVG_FLRELU_BAND=0 keeps it out of the default route, =2 sends it everything it can take
    // (A/B measurements). impl 4 (lvg_filtered_lrelu_set_impl) = everything it can take.
    static const int env_band = []() { const char* e = getenv("LVG_FLRELU_BAND"); return e ? atoi(e) : 1; }();
    // The strip kernel (round 6, float16) takes the planes it measured faster on (its launcher decides: planes of up to four strips,
    // profiles/r06_sres_ab.log); LVG_FLRELU_STRIP=0 keeps it out of the default route, =2 sends it everything it can take. impl 5 = everything.
    static const int env_strip = []() { const char* e = getenv("LVG_FLRELU_STRIP"); return e ? atoi(e) : 1; }();
    const bool use_strip = impl == 0 ? (env_strip > 0 && env_wave) : impl == 5;
    const bool strip_all = impl == 5 || env_strip >= 2;
    const bool use_band = impl == 0 ? (env_band > 0 && env_wave) : impl == 4;
    const bool band_all = impl == 4 || env_band >= 2;
    if (use_mfma && (dtype == LVG_F16 || dtype == LVG_BF16) && (cfg == CFG_U2D2 || cfg == CFG_U4D2 || cfg == CFG_U2D4))
    {
        int rc = use_strip ? lvg_flrelu_strip_launch(p, cfg, sign_mode, dtype, strip_all ? 1 : 0, st) : LVG_ERR_UNSUPPORTED;
        if (rc == LVG_ERR_UNSUPPORTED && use_band) rc = lvg_flrelu_band_launch(p, cfg, sign_mode, dtype, band_all ? 1 : 0, st);
        if (rc == LVG_ERR_UNSUPPORTED && use_wave) rc = lvg_flrelu_wave_launch(p, cfg, sign_mode, dtype, st);
        if (rc == LVG_ERR_UNSUPPORTED) rc = lvg_flrelu_mfma_launch(p, cfg, sign_mode, dtype, st);      // (slope > 1: the round-2 kernel)
        if (rc != LVG_ERR_UNSUPPORTED) return rc;      // (planes of 2 GiB and more: the VALU kernel below)
    }
    switch (dtype)
    {
        case LVG_F32:  return run_fused<float>(p, cfg, sign_mode, st);
        case LVG_F16:  return run_fused<f16_t>(p, cfg, sign_mode, st);
        default:       return run_fused<bf16_t>(p, cfg, sign_mode, st);
    }
}
