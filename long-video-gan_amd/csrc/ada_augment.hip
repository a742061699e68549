// ada_augment.hip -- fused stages of the ADA augmentation pipeline (reference model/ada_augment.py), gfx950.
//
// lvg_ada_warp: the geometric stage (:271-304) in ONE launch. The reference pads the clip by reflection (:286, margins read back to the
// host), up-samples x2 with the 12-tap low-pass (:290), resamples bilinearly through the inverse affine map (affine_grid + grid_sample,
// zeros outside, align_corners = False; :293-298) on a grid of (h + 6) * 2 x (w + 6) * 2 points and down-samples x2 with the flipped
// filter (:301): four launches with 4x-sized intermediates in memory. Here one workgroup owns a 16 x 16 tile of the OUTPUT of one
// sample and walks over the sample's planes (channels x frames share the map): the 42 x 42 points of the intermediate grid the tile
// needs are mapped into the up-sampled image once (double precision: the composed matrix of :287, :291-292, :296), their bounding box
// decides which patch of the (virtually reflect-padded) source is staged in LDS per plane, every point takes its four bilinear corners
// straight from the separable 6 x 6 poly-phase taps of the up-sampler, and the down-sampler runs row pass / column pass through LDS.
// Nothing of the padded or over-sampled images ever exists in memory; the margins come from a device tensor.
//
// lvg_ada_colour: colour matrix, additive noise, cutout (:376-381, :407-427) in one pass over the pixels (and, transposed, their backward).

#include "lvg_common.h"
#include <type_traits>

namespace {

constexpr int kTile = 16;                  // output tile edge (32 measured: forward 2.5 -> 3.0 ms on 16 x 24 planes of 144 x 256, adjoint unchanged)
constexpr int kMid = 2 * kTile + 10;       // points of the intermediate grid per tile edge (12-tap down-sampler)
constexpr int kMidPitch = kMid + 1;
constexpr int kPatch = 88;                 // edge of the LDS source patch (larger footprints read the source directly)
constexpr int kTaps = 12;

struct WarpArgs
{
    const float* x; const float* g; const int* margins; const float* taps; float* y;
    int n, k, h, w, tilesX, planesPerGroup;
};

__device__ __forceinline__ int reflect(int i, int n)
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__device__ __forceinline__ void mul3(const double* a, const double* b, double* c)
{
    double t[9];
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) t[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    #pragma unroll
    for (int i = 0; i < 9; i++) c[i] = t[i];
}

__global__ __launch_bounds__(256) void ada_warp_kernel(WarpArgs p)
{
    __shared__ float patch[kPatch * kPatch + 8];                       // (+8: the unused seventh column of a window may index one past the last row)
    __shared__ float mid[kMid * kMidPitch];
    __shared__ float rowp[kMid * (kTile + 1)];
    __shared__ float f[kTaps];
    __shared__ int box[4];                                             // x min, y min, x max, y max of the corners inside the up-sampled image
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tyi = tile / p.tilesX, txi = tile - tyi * p.tilesX;
    const int oy0 = tyi * kTile, ox0 = txi * kTile;
    const int sample = blockIdx.z;
    // (clamped like the rule that produces them, :281-282: a reflection reaches at most w - 1 / h - 1 pixels)
    const int mx0 = min(max(p.margins[0], 0), p.w - 1), my0 = min(max(p.margins[1], 0), p.h - 1);
    const int mx1 = min(max(p.margins[2], 0), p.w - 1), my1 = min(max(p.margins[3], 0), p.h - 1);
    const int hp = p.h + my0 + my1, wp = p.w + mx0 + mx1, hu = 2 * hp, wu = 2 * wp;
    const int hm = (p.h + 6) * 2, wm = (p.w + 6) * 2;
    if (tid < kTaps) f[tid] = p.taps[tid];
    if (tid < 4) box[tid] = tid < 2 ? 0x7fffffff : -0x7fffffff;
    // the sample's map: reference :287 (padding shift), :291 (x2), :292 (half-sample shift), :296 (normalised coordinates of both grids)
    double G[9];
    #pragma unroll
    for (int i = 0; i < 9; i++) G[i] = (double)p.g[sample * 9 + i];
    {
        const double T0[9] = {1, 0, (mx0 - mx1) / 2.0, 0, 1, (my0 - my1) / 2.0, 0, 0, 1};
        mul3(T0, G, G);
        const double S2[9] = {2, 0, 0, 0, 2, 0, 0, 0, 1}, Sh[9] = {0.5, 0, 0, 0, 0.5, 0, 0, 0, 1};
        mul3(S2, G, G); mul3(G, Sh, G);
        const double Tm[9] = {1, 0, -0.5, 0, 1, -0.5, 0, 0, 1}, Tp[9] = {1, 0, 0.5, 0, 1, 0.5, 0, 0, 1};
        mul3(Tm, G, G); mul3(G, Tp, G);
        const double Sa[9] = {2.0 / wu, 0, 0, 0, 2.0 / hu, 0, 0, 0, 1}, Sb[9] = {wm / 2.0, 0, 0, 0, hm / 2.0, 0, 0, 0, 1};
        mul3(Sa, G, G); mul3(G, Sb, G);
    }
    __syncthreads();
    // a point of the intermediate grid (rows 2 oy0 + 1 .. + kMid, columns likewise: :301 crops one sample on each side) -> its upper-left
    // corner in the up-sampled image and the bilinear weights; false: no corner can lie inside the image (or non-finite coordinates)
    auto point = [&](int idx, int& x0, int& y0, float& tx, float& ty) -> bool
    {
        const int a = 2 * oy0 + 1 + idx / kMid, b = 2 * ox0 + 1 + idx % kMid;
        const double xn = (2.0 * b + 1.0) / wm - 1.0, yn = (2.0 * a + 1.0) / hm - 1.0;
        const double gx = G[0] * xn + G[1] * yn + G[2], gy = G[3] * xn + G[4] * yn + G[5];
        const double px = ((gx + 1.0) * wu - 1.0) / 2.0, py = ((gy + 1.0) * hu - 1.0) / 2.0;
        const double fx = floor(px), fy = floor(py);
        if (!(fx >= -2.0 && fx <= (double)wu && fy >= -2.0 && fy <= (double)hu)) return false;
        x0 = (int)fx; y0 = (int)fy; tx = (float)(px - fx); ty = (float)(py - fy);
        return true;
    };
    #pragma unroll 1
    for (int idx = tid; idx < kMid * kMid; idx += 256)
    {
        int x0, y0; float tx, ty;
        if (point(idx, x0, y0, tx, ty))
        {
            const int ux0 = max(x0, 0), ux1 = min(x0 + 1, wu - 1), uy0 = max(y0, 0), uy1 = min(y0 + 1, hu - 1);
            if (ux0 <= ux1 && uy0 <= uy1)
            {
                atomicMin(&box[0], ux0); atomicMin(&box[1], uy0); atomicMax(&box[2], ux1); atomicMax(&box[3], uy1);
            }
        }
    }
    __syncthreads();
    // source patch that covers the box: column i of the padded source feeds up-sampled column u with tap 5 + u - 2 i, i in ceil((u - 6) / 2) .. + 5
    const bool any = box[0] <= box[2];
    const int pi0 = (box[0] - 6 + 1) >> 1, pj0 = (box[1] - 6 + 1) >> 1;
    const int pw = any ? ((box[2] - 6 + 1) >> 1) + 6 - pi0 : 0, ph = any ? ((box[3] - 6 + 1) >> 1) + 6 - pj0 : 0;
    const bool staged = pw <= kPatch && ph <= kPatch;
    const int k0 = blockIdx.y * p.planesPerGroup, k1 = min(p.k, k0 + p.planesPerGroup);
    for (int pl = k0; pl < k1; pl++)
    {
        const float* src = p.x + ((int64_t)sample * p.k + pl) * p.h * p.w;
        auto padded = [&](int j, int i) -> float                        // the reflect-padded source, zero outside it (the up-sampler's padding)
        {
            if (i < 0 || i >= wp || j < 0 || j >= hp) return 0.f;
            return src[(int64_t)reflect(j - my0, p.h) * p.w + reflect(i - mx0, p.w)];
        };
        if (staged)
            for (int e = tid; e < pw * ph; e += 256)
            {
                const int j = e / pw, i = e - j * pw;
                patch[j * kPatch + i] = padded(pj0 + j, pi0 + i);
            }
        __syncthreads();
        auto fill = [&](auto stagedTag)
        {
            constexpr bool kStaged = decltype(stagedTag)::value;
            #pragma unroll 1
            for (int idx = tid; idx < kMid * kMid; idx += 256)
            {
                int x0, y0; float tx, ty;
                float v = 0.f;
                if (point(idx, x0, y0, tx, ty))
                {
                    // up-sampled value at (row vv, column uu) = 4 sum_j sum_i P[j][i] f[5 + vv - 2 j] f[5 + uu - 2 i], i = ceil((uu - 6) / 2) .. + 5.
                    // The two columns x0, x0 + 1 draw on 7 consecutive source columns (6 when x0 is odd), the two rows likewise: one sweep
                    // over the 7 x 7 window gives the row sums of both columns, then the four corners.
                    const int ia = (x0 - 6 + 1) >> 1, ib = (x0 + 1 - 6 + 1) >> 1;          // first source column of the left / right corner
                    const int ja = (y0 - 6 + 1) >> 1, jb = (y0 + 1 - 6 + 1) >> 1;
                    const bool okL = x0 >= 0 && x0 < wu, okR = x0 + 1 >= 0 && x0 + 1 < wu;
                    const bool okT = y0 >= 0 && y0 < hu, okB = y0 + 1 >= 0 && y0 + 1 < hu;
                    float cTL = 0.f, cTR = 0.f, cBL = 0.f, cBR = 0.f;
                    #pragma unroll
                    for (int jj = 0; jj < 7; jj++)
                    {
                        const int j = ja + jj;
                        if (jj == 6 && jb == ja) continue;                    // (the seventh row belongs to the lower corners only when they start one row later)
                        float s7[7];
                        #pragma unroll
                        for (int ii = 0; ii < 7; ii++) s7[ii] = kStaged ? patch[(j - pj0) * kPatch + (ia + ii - pi0)] : padded(j, ia + ii);
                        float hl = 0.f, hr = 0.f;
                        #pragma unroll
                        for (int ii = 0; ii < 6; ii++)
                        {
                            hl = fmaf(s7[ii], f[5 + x0 - 2 * (ia + ii)], hl);
                            hr = fmaf(s7[ii + (ib - ia)], f[5 + x0 + 1 - 2 * (ib + ii)], hr);
                        }
                        const int tTop = 5 + y0 - 2 * j, tBot = 5 + y0 + 1 - 2 * j;                   // vertical taps of this source row (if in range)
                        const float wT = (jj < 6) ? f[min(max(tTop, 0), kTaps - 1)] : 0.f;
                        const float wB = (j >= jb && j < jb + 6) ? f[min(max(tBot, 0), kTaps - 1)] : 0.f;
                        cTL = fmaf(hl, wT, cTL); cTR = fmaf(hr, wT, cTR);
                        cBL = fmaf(hl, wB, cBL); cBR = fmaf(hr, wB, cBR);
                    }
                    const float wl = 1.f - tx, wt = 1.f - ty;
                    if (okT && okL) v = fmaf(4.f * cTL, wl * wt, v);
                    if (okT && okR) v = fmaf(4.f * cTR, tx * wt, v);
                    if (okB && okL) v = fmaf(4.f * cBL, wl * ty, v);
                    if (okB && okR) v = fmaf(4.f * cBR, tx * ty, v);
                }
                mid[(idx / kMid) * kMidPitch + idx % kMid] = v;
            }
        };
        if (staged) fill(std::true_type{}); else fill(std::false_type{});
        __syncthreads();
        // down-sampler, flipped filter (:301): out[oy][ox] = sum_ky sum_kx M[2 oy + 1 + ky][2 ox + 1 + kx] f[ky] f[kx]
        for (int e = tid; e < kMid * kTile; e += 256)
        {
            const int r = e / kTile, ox = e - r * kTile;
            float acc = 0.f;
            #pragma unroll
            for (int kx = 0; kx < kTaps; kx++) acc = fmaf(mid[r * kMidPitch + 2 * ox + kx], f[kx], acc);
            rowp[r * (kTile + 1) + ox] = acc;
        }
        __syncthreads();
        for (int e = tid; e < kTile * kTile; e += 256)
        {
            const int oy = e / kTile, ox = e % kTile;
            float acc = 0.f;
            #pragma unroll
            for (int ky = 0; ky < kTaps; ky++) acc = fmaf(rowp[(2 * oy + ky) * (kTile + 1) + ox], f[ky], acc);
            if (oy0 + oy < p.h && ox0 + ox < p.w)
                p.y[(((int64_t)sample * p.k + pl) * p.h + oy0 + oy) * p.w + ox0 + ox] = acc;
        }
        __syncthreads();
    }
}

// The adjoint of ada_warp_kernel (the stage is linear in the clip): d x = P^T U^T B^T D^T d y. Same decomposition -- a workgroup takes
// the 16 x 16 tile of d y that the forward workgroup produced -- run backwards: the tile's share of the gradient of the 42 x 42
// intermediate points (transposed down-sampler, column pass then row pass through LDS), each point's four bilinear weights spread over
// the 7 x 7 source window its corners draw on (transposed poly-phase up-sampler) into an LDS accumulator shaped like the forward's
// source patch, and the patch added to d x through the reflection map. Tiles overlap in the source (and reflected pixels fold onto
// their originals), so the last step uses float atomics: like grid_sample's own backward, the summation order is not fixed -- and
// device-scope atomics are what its time is (21 ms against 2.5 ms for the forward kernel on 16 x 24 planes of 144 x 256; a gather
// form through the inverse map would need the transposed down-sampler's output in memory: not built).
__global__ __launch_bounds__(256) void ada_warp_adjoint_kernel(WarpArgs p)
{
    __shared__ float patch[kPatch * kPatch + 8];
    __shared__ float mid[kMid * kMidPitch];
    __shared__ float rowp[kMid * (kTile + 1)];
    __shared__ float dyt[kTile * (kTile + 1)];
    __shared__ float f[kTaps];
    __shared__ int box[4];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tyi = tile / p.tilesX, txi = tile - tyi * p.tilesX;
    const int oy0 = tyi * kTile, ox0 = txi * kTile;
    const int sample = blockIdx.z;
    const int mx0 = min(max(p.margins[0], 0), p.w - 1), my0 = min(max(p.margins[1], 0), p.h - 1);
    const int mx1 = min(max(p.margins[2], 0), p.w - 1), my1 = min(max(p.margins[3], 0), p.h - 1);
    const int hp = p.h + my0 + my1, wp = p.w + mx0 + mx1, hu = 2 * hp, wu = 2 * wp;
    const int hm = (p.h + 6) * 2, wm = (p.w + 6) * 2;
    if (tid < kTaps) f[tid] = p.taps[tid];
    if (tid < 4) box[tid] = tid < 2 ? 0x7fffffff : -0x7fffffff;
    double G[9];
    #pragma unroll
    for (int i = 0; i < 9; i++) G[i] = (double)p.g[sample * 9 + i];
    {
        const double T0[9] = {1, 0, (mx0 - mx1) / 2.0, 0, 1, (my0 - my1) / 2.0, 0, 0, 1};
        mul3(T0, G, G);
        const double S2[9] = {2, 0, 0, 0, 2, 0, 0, 0, 1}, Sh[9] = {0.5, 0, 0, 0, 0.5, 0, 0, 0, 1};
        mul3(S2, G, G); mul3(G, Sh, G);
        const double Tm[9] = {1, 0, -0.5, 0, 1, -0.5, 0, 0, 1}, Tp[9] = {1, 0, 0.5, 0, 1, 0.5, 0, 0, 1};
        mul3(Tm, G, G); mul3(G, Tp, G);
        const double Sa[9] = {2.0 / wu, 0, 0, 0, 2.0 / hu, 0, 0, 0, 1}, Sb[9] = {wm / 2.0, 0, 0, 0, hm / 2.0, 0, 0, 0, 1};
        mul3(Sa, G, G); mul3(G, Sb, G);
    }
    __syncthreads();
    auto point = [&](int idx, int& x0, int& y0, float& tx, float& ty) -> bool
    {
        const int a = 2 * oy0 + 1 + idx / kMid, b = 2 * ox0 + 1 + idx % kMid;
        const double xn = (2.0 * b + 1.0) / wm - 1.0, yn = (2.0 * a + 1.0) / hm - 1.0;
        const double gx = G[0] * xn + G[1] * yn + G[2], gy = G[3] * xn + G[4] * yn + G[5];
        const double px = ((gx + 1.0) * wu - 1.0) / 2.0, py = ((gy + 1.0) * hu - 1.0) / 2.0;
        const double fx = floor(px), fy = floor(py);
        if (!(fx >= -2.0 && fx <= (double)wu && fy >= -2.0 && fy <= (double)hu)) return false;
        x0 = (int)fx; y0 = (int)fy; tx = (float)(px - fx); ty = (float)(py - fy);
        return true;
    };
    #pragma unroll 1
    for (int idx = tid; idx < kMid * kMid; idx += 256)
    {
        int x0, y0; float tx, ty;
        if (point(idx, x0, y0, tx, ty))
        {
            const int ux0 = max(x0, 0), ux1 = min(x0 + 1, wu - 1), uy0 = max(y0, 0), uy1 = min(y0 + 1, hu - 1);
            if (ux0 <= ux1 && uy0 <= uy1)
            {
                atomicMin(&box[0], ux0); atomicMin(&box[1], uy0); atomicMax(&box[2], ux1); atomicMax(&box[3], uy1);
            }
        }
    }
    __syncthreads();
    const bool any = box[0] <= box[2];
    const int pi0 = (box[0] - 6 + 1) >> 1, pj0 = (box[1] - 6 + 1) >> 1;
    const int pw = any ? ((box[2] - 6 + 1) >> 1) + 6 - pi0 : 0, ph = any ? ((box[3] - 6 + 1) >> 1) + 6 - pj0 : 0;
    const bool staged = pw <= kPatch && ph <= kPatch;
    const int k0 = blockIdx.y * p.planesPerGroup, k1 = min(p.k, k0 + p.planesPerGroup);
    for (int pl = k0; pl < k1; pl++)
    {
        const float* dy = p.x + ((int64_t)sample * p.k + pl) * p.h * p.w;       // (x = the incoming gradient, y = d x, zero-filled by the launcher)
        float* dx = p.y + ((int64_t)sample * p.k + pl) * p.h * p.w;
        auto spread = [&](int j, int i, float v)                               // transposed reflect padding: the padded pixel (j, i) is a source pixel
        {
            if (v == 0.f || i < 0 || i >= wp || j < 0 || j >= hp) return;
            unsafeAtomicAdd(dx + (int64_t)reflect(j - my0, p.h) * p.w + reflect(i - mx0, p.w), v);
        };
        if (staged)
            for (int e = tid; e < pw * ph; e += 256) patch[(e / pw) * kPatch + e % pw] = 0.f;
        for (int e = tid; e < kTile * kTile; e += 256)
        {
            const int oy = e / kTile, ox = e % kTile;
            dyt[oy * (kTile + 1) + ox] = (oy0 + oy < p.h && ox0 + ox < p.w) ? dy[(int64_t)(oy0 + oy) * p.w + ox0 + ox] : 0.f;
        }
        __syncthreads();
        // transposed down-sampler: rowp[r][ox] = sum_{oy: r = 2 oy + ky} d y[oy][ox] f[ky];  mid[r][c] = sum_{ox: c = 2 ox + kx} rowp[r][ox] f[kx]
        for (int e = tid; e < kMid * kTile; e += 256)
        {
            const int r = e / kTile, ox = e - r * kTile;
            float acc = 0.f;
            #pragma unroll
            for (int q = 0; q < 6; q++)
            {
                const int oy = (r >> 1) - q, ky = r - 2 * oy;
                if (oy >= 0 && oy < kTile && ky < kTaps) acc = fmaf(dyt[oy * (kTile + 1) + ox], f[ky], acc);
            }
            rowp[r * (kTile + 1) + ox] = acc;
        }
        __syncthreads();
        for (int e = tid; e < kMid * kMid; e += 256)
        {
            const int r = e / kMid, c = e - r * kMid;
            float acc = 0.f;
            #pragma unroll
            for (int q = 0; q < 6; q++)
            {
                const int ox = (c >> 1) - q, kx = c - 2 * ox;
                if (ox >= 0 && ox < kTile && kx < kTaps) acc = fmaf(rowp[r * (kTile + 1) + ox], f[kx], acc);
            }
            mid[r * kMidPitch + c] = acc;
        }
        __syncthreads();
        // transposed bilinear sampling + transposed up-sampler, point by point
        #pragma unroll 1
        for (int idx = tid; idx < kMid * kMid; idx += 256)
        {
            const float gm = mid[(idx / kMid) * kMidPitch + idx % kMid];
            int x0, y0; float tx, ty;
            if (gm == 0.f || !point(idx, x0, y0, tx, ty)) continue;
            const int ia = (x0 - 6 + 1) >> 1, ib = (x0 + 1 - 6 + 1) >> 1;
            const int ja = (y0 - 6 + 1) >> 1, jb = (y0 + 1 - 6 + 1) >> 1;
            const bool okL = x0 >= 0 && x0 < wu, okR = x0 + 1 >= 0 && x0 + 1 < wu;
            const bool okT = y0 >= 0 && y0 < hu, okB = y0 + 1 >= 0 && y0 + 1 < hu;
            const float g4 = 4.f * gm;
            const float wTL = (okT && okL) ? g4 * (1.f - tx) * (1.f - ty) : 0.f, wTR = (okT && okR) ? g4 * tx * (1.f - ty) : 0.f;
            const float wBL = (okB && okL) ? g4 * (1.f - tx) * ty : 0.f,         wBR = (okB && okR) ? g4 * tx * ty : 0.f;
            float hT[7], hB[7];
            #pragma unroll
            for (int ii = 0; ii < 7; ii++)
            {
                const int i = ia + ii;
                const float cl = (ii < 6) ? f[5 + x0 - 2 * i] : 0.f;
                const int tr = 5 + x0 + 1 - 2 * i;
                const float cr = (i >= ib && i < ib + 6) ? f[min(max(tr, 0), kTaps - 1)] : 0.f;
                hT[ii] = wTL * cl + wTR * cr;
                hB[ii] = wBL * cl + wBR * cr;
            }
            #pragma unroll
            for (int jj = 0; jj < 7; jj++)
            {
                const int j = ja + jj;
                const float rT = (jj < 6) ? f[5 + y0 - 2 * j] : 0.f;
                const int tb = 5 + y0 + 1 - 2 * j;
                const float rB = (j >= jb && j < jb + 6) ? f[min(max(tb, 0), kTaps - 1)] : 0.f;
                if (rT == 0.f && rB == 0.f) continue;
                #pragma unroll
                for (int ii = 0; ii < 7; ii++)
                {
                    const float v = rT * hT[ii] + rB * hB[ii];
                    if (v == 0.f) continue;
                    if (staged)
                    {
                        const int pj = j - pj0, pi = ia + ii - pi0;
                        if (pj >= 0 && pj < ph && pi >= 0 && pi < pw) unsafeAtomicAdd(&patch[pj * kPatch + pi], v);
                    }
                    else spread(j, ia + ii, v);
                }
            }
        }
        __syncthreads();
        if (staged)
            for (int e = tid; e < pw * ph; e += 256)
            {
                const int j = e / pw, i = e - j * pw;
                spread(pj0 + j, pi0 + i, patch[j * kPatch + i]);
            }
        __syncthreads();
    }
}

struct ColourArgs
{
    const float* x; const float* cmat; const float* noise; const float* sigma; const float* cut; float* y;
    int n, t, h, w, transpose;
};

// one thread per pixel of a sample's [t, h, w] volume, the three colours together
__global__ __launch_bounds__(256) void ada_colour_kernel(ColourArgs p)
{
    const int64_t plane = (int64_t)p.t * p.h * p.w;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= plane) return;
    const int px = (int)(i % p.w), py = (int)((i / p.w) % p.h);
    const float* x = p.x + (int64_t)s * 3 * plane + i;
    float v[3] = {x[0], x[plane], x[2 * plane]};
    bool keep = true;
    if (p.cut)
    {
        const float* q = p.cut + s * 4;
        keep = fabsf((px + 0.5f) / p.w - q[0]) >= q[2] / 2 || fabsf((py + 0.5f) / p.h - q[1]) >= q[3] / 2;
    }
    if (p.transpose & 1)
    {
        // backward: d x = C[:3, :3]^T (d y where kept)
        if (!keep) v[0] = v[1] = v[2] = 0.f;
        if (p.cmat)
        {
            const float* m = p.cmat + s * 16;
            const float r = v[0], g = v[1], b = v[2];
            v[0] = m[0] * r + m[4] * g + m[8] * b; v[1] = m[1] * r + m[5] * g + m[9] * b; v[2] = m[2] * r + m[6] * g + m[10] * b;
        }
    }
    else
    {
        if (p.cmat)
        {
            const float* m = p.cmat + s * 16;
            const float r = v[0], g = v[1], b = v[2];
            const float o0 = (p.transpose & 2) ? 0.f : m[3], o1 = (p.transpose & 2) ? 0.f : m[7], o2 = (p.transpose & 2) ? 0.f : m[11];
            v[0] = fmaf(m[0], r, fmaf(m[1], g, fmaf(m[2], b, o0)));
            v[1] = fmaf(m[4], r, fmaf(m[5], g, fmaf(m[6], b, o1)));
            v[2] = fmaf(m[8], r, fmaf(m[9], g, fmaf(m[10], b, o2)));
        }
        if (p.noise)
        {
            const float* nz = p.noise + (int64_t)s * 3 * plane + i;
            const float sg = p.sigma[s];
            v[0] = fmaf(nz[0], sg, v[0]); v[1] = fmaf(nz[plane], sg, v[1]); v[2] = fmaf(nz[2 * plane], sg, v[2]);
        }
        if (!keep) v[0] = v[1] = v[2] = 0.f;
    }
    float* y = p.y + (int64_t)s * 3 * plane + i;
    y[0] = v[0]; y[plane] = v[1]; y[2 * plane] = v[2];
}

} // namespace

extern "C" int lvg_ada_warp(const float* x, const float* g_inv, const int* margins, const float* taps, float* y,
                            int n, int k, int h, int w, void* stream)
{
    LVG_REQUIRE(x && g_inv && margins && taps && y, "ada_warp: null pointer");
    LVG_REQUIRE(n >= 1 && n <= 65535 && k >= 1 && h >= 2 && w >= 2 && (int64_t)n * k * h * w < 0x7fffffffLL, "ada_warp: bad sizes");
    WarpArgs a = {};
    a.x = x; a.g = g_inv; a.margins = margins; a.taps = taps; a.y = y; a.n = n; a.k = k; a.h = h; a.w = w;
    a.tilesX = (w + kTile - 1) / kTile;
    const int tiles = a.tilesX * ((h + kTile - 1) / kTile);
    // planes per workgroup: the geometry of a tile is shared by all planes of the sample; split them only as far as the chip needs workgroups
    int groups = 1;
    while ((int64_t)tiles * n * groups < 1024 && groups < k) groups++;
    a.planesPerGroup = (k + groups - 1) / groups;
    groups = (k + a.planesPerGroup - 1) / a.planesPerGroup;
    LVG_REQUIRE(groups <= 65535, "ada_warp: too many plane groups");
    hipLaunchKernelGGL(ada_warp_kernel, dim3((unsigned)tiles, (unsigned)groups, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("ada_warp");
}

extern "C" int lvg_ada_warp_adjoint(const float* dy, const float* g_inv, const int* margins, const float* taps, float* dx,
                                    int n, int k, int h, int w, void* stream)
{
    LVG_REQUIRE(dy && g_inv && margins && taps && dx, "ada_warp_adjoint: null pointer");
    LVG_REQUIRE(n >= 1 && n <= 65535 && k >= 1 && h >= 2 && w >= 2 && (int64_t)n * k * h * w < 0x7fffffffLL, "ada_warp_adjoint: bad sizes");
    WarpArgs a = {};
    a.x = dy; a.g = g_inv; a.margins = margins; a.taps = taps; a.y = dx; a.n = n; a.k = k; a.h = h; a.w = w;
    a.tilesX = (w + kTile - 1) / kTile;
    const int tiles = a.tilesX * ((h + kTile - 1) / kTile);
    int groups = 1;
    while ((int64_t)tiles * n * groups < 1024 && groups < k) groups++;
    a.planesPerGroup = (k + groups - 1) / groups;
    groups = (k + a.planesPerGroup - 1) / a.planesPerGroup;
    LVG_REQUIRE(groups <= 65535, "ada_warp_adjoint: too many plane groups");
    if (hipMemsetAsync(dx, 0, (size_t)n * k * h * w * sizeof(float), (hipStream_t)stream) != hipSuccess)
    {
        (void)hipGetLastError();
        lvg_set_error("ada_warp_adjoint: cannot clear the output");
        return LVG_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(ada_warp_adjoint_kernel, dim3((unsigned)tiles, (unsigned)groups, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("ada_warp_adjoint");
}

extern "C" int lvg_ada_colour(const float* x, const float* cmat, const float* noise, const float* sigma, const float* cut, float* y,
                              int n, int t, int h, int w, int transpose, void* stream)
{
    LVG_REQUIRE(x && y && n >= 1 && n <= 65535 && t >= 1 && h >= 1 && w >= 1, "ada_colour: bad arguments");
    LVG_REQUIRE((noise == nullptr) == (sigma == nullptr), "ada_colour: noise and sigma go together");
    const int64_t plane = (int64_t)t * h * w;
    LVG_REQUIRE(plane * 3 * n < 0x7fffffffLL, "ada_colour: tensor too large");
    ColourArgs a = {};
    a.x = x; a.cmat = cmat; a.noise = noise; a.sigma = sigma; a.cut = cut; a.y = y; a.n = n; a.t = t; a.h = h; a.w = w; a.transpose = transpose;
    hipLaunchKernelGGL(ada_colour_kernel, dim3((unsigned)lvg_ceil_div(plane, 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    return lvg_check_launch("ada_colour");
}
