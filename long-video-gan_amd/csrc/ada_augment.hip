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

// The adjoint of ada_warp_kernel (the stage is linear in the clip): d x = P^T U^T B^T D^T d y, as GATHERS only -- no atomics, a fixed
// summation order. D^T (transposed down-sampler) is an ordinary x2 up-sampling of d y and is done by lvg_upfirdn2d into a workspace M~
// of the intermediate grid's size; this kernel does the rest for one 16 x 16 tile of d x and the planes of a sample:
//   B^T: the gradient of an up-sampled pixel (v, u) collects, from every point (a, b) of the intermediate grid whose bilinear footprint
//        covers it, M~[a][b] * (1 - |px - u|)(1 - |py - v|). The points are found through the INVERSE of the affine map: they lie in
//        the parallelogram J^-1 ([-1, 1]^2) around J^-1 (u, v), whose bounding box is enumerated (10 - 45 candidates for the zooms ADA draws);
//   U^T: the transposed poly-phase up-sampler is a 12-tap x2 decimation of that gradient: row pass / column pass through LDS;
//   P^T: a source pixel receives its own padded position and up to two mirror images per axis (reflect padding): the tile loops over the
//        <= 9 reflection branches that exist for it and sums.
// (A first version scattered through LDS and device-scope float atomics: 21 ms against 2.5 ms for the forward kernel on 16 x 24 planes of
// 144 x 256, twice the time of the composition's backward; profiles/r03_ada_bench.log.)
constexpr int kStage = 96;                 // edge of the LDS window of M~ (larger footprints read it from memory)

__global__ __launch_bounds__(256) void ada_warp_adjoint_kernel(WarpArgs p)
{
    __shared__ float mst[kStage * kStage];
    __shared__ float ut[kMid * kMidPitch];
    __shared__ float rp[kMid * (kTile + 1)];
    __shared__ float f[kTaps];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tyi = tile / p.tilesX, txi = tile - tyi * p.tilesX;
    const int y0 = tyi * kTile, x0 = txi * kTile;
    const int sample = blockIdx.z;
    const int mx0 = min(max(p.margins[0], 0), p.w - 1), my0 = min(max(p.margins[1], 0), p.h - 1);
    const int mx1 = min(max(p.margins[2], 0), p.w - 1), my1 = min(max(p.margins[3], 0), p.h - 1);
    const int hp = p.h + my0 + my1, wp = p.w + mx0 + mx1, hu = 2 * hp, wu = 2 * wp;
    const int hm = (p.h + 6) * 2, wm = (p.w + 6) * 2;
    if (tid < kTaps) f[tid] = p.taps[tid];
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
    // the forward map of a grid point (row a, column b) as px = cxb b + cxa a + cx0, py = cyb b + cya a + cy0, and its inverse
    const double cxb = G[0] * (2.0 / wm) * wu / 2.0, cxa = G[1] * (2.0 / hm) * wu / 2.0;
    const double cx0 = ((G[0] * (1.0 / wm - 1.0) + G[1] * (1.0 / hm - 1.0) + G[2] + 1.0) * wu - 1.0) / 2.0;
    const double cyb = G[3] * (2.0 / wm) * hu / 2.0, cya = G[4] * (2.0 / hm) * hu / 2.0;
    const double cy0 = ((G[3] * (1.0 / wm - 1.0) + G[4] * (1.0 / hm - 1.0) + G[5] + 1.0) * hu - 1.0) / 2.0;
    const double det = cxb * cya - cxa * cyb;
    // (a singular or non-finite map samples nothing the forward pass could have used consistently: its gradient is left zero)
    const bool usable = isfinite(det) && fabs(det) > 1e-12 && isfinite(cx0) && isfinite(cy0);
    const double i00 = cya / det, i01 = -cxa / det, i10 = -cyb / det, i11 = cxb / det;      // (b, a) = J^-1 (px - cx0, py - cy0)
    const double hb = fabs(i00) + fabs(i01) + 1e-6, ha = fabs(i10) + fabs(i11) + 1e-6;
    const int ty = tid / kTile, tx = tid % kTile;
    const int y = y0 + ty, x = x0 + tx;
    const int yEnd = min(y0 + kTile, p.h) - 1, xEnd = min(x0 + kTile, p.w) - 1;
    __syncthreads();
    const int k0 = blockIdx.y * p.planesPerGroup, k1 = min(p.k, k0 + p.planesPerGroup);
    for (int pl = k0; pl < k1; pl++)
    {
        const float* M = p.x + ((int64_t)sample * p.k + pl) * hm * wm;
        float acc = 0.f;
        #pragma unroll 1
        for (int br = 0; br < (usable ? 9 : 0); br++)
        {
            const int by = br / 3, bx = br - 3 * by;
            // rows / columns of the tile that have an image in this reflection branch, and the padded positions they come from
            int yl, yh, xl, xh;
            if (by == 0)      { yl = y0; yh = yEnd; }
            else if (by == 1) { yl = max(y0, 1); yh = min(yEnd, my0); }
            else              { yl = max(y0, p.h - 1 - my1); yh = min(yEnd, p.h - 2); }
            if (bx == 0)      { xl = x0; xh = xEnd; }
            else if (bx == 1) { xl = max(x0, 1); xh = min(xEnd, mx0); }
            else              { xl = max(x0, p.w - 1 - mx1); xh = min(xEnd, p.w - 2); }
            if (yl > yh || xl > xh) continue;                                   // (uniform over the workgroup)
            auto prow = [&](int yy) { return by == 0 ? my0 + yy : (by == 1 ? my0 - yy : my0 + 2 * (p.h - 1) - yy); };
            auto pcol = [&](int xx) { return bx == 0 ? mx0 + xx : (bx == 1 ? mx0 - xx : mx0 + 2 * (p.w - 1) - xx); };
            const int jmin = min(prow(yl), prow(yh)), jmax = max(prow(yl), prow(yh));
            const int imin = min(pcol(xl), pcol(xh)), imax = max(pcol(xl), pcol(xh));
            const int v0 = 2 * jmin - 5, nv = 2 * (jmax - jmin) + 12, u0 = 2 * imin - 5, nu = 2 * (imax - imin) + 12;
            // window of M~ that the up-sampled region's candidates can fall in
            double bl = 1e300, bh = -1e300, al = 1e300, ah = -1e300;
            #pragma unroll
            for (int c = 0; c < 4; c++)
            {
                const double du = (double)((c & 1) ? u0 + nu - 1 : u0) - cx0, dv = (double)((c & 2) ? v0 + nv - 1 : v0) - cy0;
                const double bs = i00 * du + i01 * dv, as = i10 * du + i11 * dv;
                bl = fmin(bl, bs); bh = fmax(bh, bs); al = fmin(al, as); ah = fmax(ah, as);
            }
            const double bLo = fmax(ceil(bl - hb), 0.0), bHi = fmin(floor(bh + hb), (double)(wm - 1));
            const double aLo = fmax(ceil(al - ha), 0.0), aHi = fmin(floor(ah + ha), (double)(hm - 1));
            if (!(bLo <= bHi && aLo <= aHi)) continue;                          // nothing of the intermediate grid maps here
            const int b_lo = (int)bLo, b_hi = (int)bHi, a_lo = (int)aLo, a_hi = (int)aHi;
            const int sw = b_hi - b_lo + 1, sh = a_hi - a_lo + 1;
            const bool staged = sw <= kStage && sh <= kStage;
            __syncthreads();                                                    // (previous branch / plane done with the LDS tiles)
            if (staged)
                for (int e = tid; e < sw * sh; e += 256)
                {
                    const int r = e / sw, c = e - r * sw;
                    mst[r * kStage + c] = M[(int64_t)(a_lo + r) * wm + b_lo + c];
                }
            __syncthreads();
            // B^T: gradient of the up-sampled pixels of the region
            #pragma unroll 1
            for (int e = tid; e < nv * nu; e += 256)
            {
                const int r = e / nu, c = e - r * nu;
                const int v = v0 + r, u = u0 + c;
                float g = 0.f;
                if (v >= 0 && v < hu && u >= 0 && u < wu)
                {
                    const double du = (double)u - cx0, dv = (double)v - cy0;
                    const double bs = i00 * du + i01 * dv, as = i10 * du + i11 * dv;
                    const int b0 = (int)fmax(ceil(bs - hb), (double)b_lo), b1 = (int)fmin(floor(bs + hb), (double)b_hi);
                    const int a0 = (int)fmax(ceil(as - ha), (double)a_lo), a1 = (int)fmin(floor(as + ha), (double)a_hi);
                    for (int a = a0; a <= a1; a++)
                    {
                        const double pxr = cxa * a + cx0 - (double)u, pyr = cya * a + cy0 - (double)v;
                        for (int b = b0; b <= b1; b++)
                        {
                            const float wx = 1.f - fabsf((float)(cxb * b + pxr)), wy = 1.f - fabsf((float)(cyb * b + pyr));
                            if (wx > 0.f && wy > 0.f)
                                g = fmaf(staged ? mst[(a - a_lo) * kStage + (b - b_lo)] : M[(int64_t)a * wm + b], wx * wy, g);
                        }
                    }
                }
                ut[r * kMidPitch + c] = g;
            }
            __syncthreads();
            // U^T, row pass: rp[r][ii] = sum_t ut[r][2 ii + t] f[t]   (padded column imin + ii draws on up-sampled columns 2 i - 5 + t, tap t)
            const int ni = imax - imin + 1;
            for (int e = tid; e < nv * ni; e += 256)
            {
                const int r = e / ni, ii = e - r * ni;
                float s = 0.f;
                #pragma unroll
                for (int t = 0; t < kTaps; t++) s = fmaf(ut[r * kMidPitch + 2 * ii + t], f[t], s);
                rp[r * (kTile + 1) + ii] = s;
            }
            __syncthreads();
            // column pass at this thread's own pixel, if it has an image in the branch
            if (y >= yl && y <= yh && x >= xl && x <= xh)
            {
                const int jj = prow(y) - jmin, ii = pcol(x) - imin;
                float s = 0.f;
                #pragma unroll
                for (int t = 0; t < kTaps; t++) s = fmaf(rp[(2 * jj + t) * (kTile + 1) + ii], f[t], s);
                acc = fmaf(4.f, s, acc);
            }
        }
        if (y < p.h && x < p.w) p.y[(((int64_t)sample * p.k + pl) * p.h + y) * p.w + x] = acc;
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

extern "C" int lvg_ada_warp_adjoint(const float* dy, const float* g_inv, const int* margins, const float* taps, float* workspace, float* dx,
                                    int n, int k, int h, int w, void* stream)
{
    LVG_REQUIRE(dy && g_inv && margins && taps && workspace && dx, "ada_warp_adjoint: null pointer");
    LVG_REQUIRE(n >= 1 && n <= 65535 && k >= 1 && h >= 2 && w >= 2 && (int64_t)n * k * (h + 6) * (w + 6) * 4 < 0x7fffffffLL, "ada_warp_adjoint: bad sizes");
    // D^T: the transposed x2 down-sampler (flipped filter, one sample cropped on each side) = x2 up-sampling of d y with padding 12 / 11
    const int hm = (h + 6) * 2, wm = (w + 6) * 2;
    const int64_t xs[4] = {n, k, h, w}, xst[4] = {(int64_t)k * h * w, (int64_t)h * w, w, 1};
    const int64_t ys[4] = {n, k, hm, wm}, yst[4] = {(int64_t)k * hm * wm, (int64_t)hm * wm, wm, 1};
    if (int rc = lvg_upfirdn2d(dy, workspace, nullptr, taps, taps, xs, xst, ys, yst, kTaps, kTaps, 0, 0, 2, 2, 1, 1, 12, 12, 0, 1.0f, LVG_F32, stream)) return rc;
    WarpArgs a = {};
    a.x = workspace; a.g = g_inv; a.margins = margins; a.taps = taps; a.y = dx; a.n = n; a.k = k; a.h = h; a.w = w;
    a.tilesX = (w + kTile - 1) / kTile;
    const int tiles = a.tilesX * ((h + kTile - 1) / kTile);
    int groups = 1;
    while ((int64_t)tiles * n * groups < 1024 && groups < k) groups++;
    a.planesPerGroup = (k + groups - 1) / groups;
    groups = (k + a.planesPerGroup - 1) / a.planesPerGroup;
    LVG_REQUIRE(groups <= 65535, "ada_warp_adjoint: too many plane groups");
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
