// igemm_kernel.h -- the implicit-GEMM convolution kernel shared by conv3d_igemm.hip (time-major frames, temporal taps in the
// K loop, 'same' padding through masks: T2D = false) and conv2d_igemm.hip (2-D pixel tiles of explicitly zero-padded frames,
// 'valid' correlation, no masks: T2D = true). Each .hip file is its own translation unit with its own instantiations, so the
// register allocation of one family cannot disturb the other (cdna_hip_programming.md 5.4 rule 19).
#pragma once
#include "epilogue_common.h"
#include <stdlib.h>
#include <string.h>
#include <utility>

namespace {

struct ConvArgs
{
    const void*  x;
    const void*  w;
    const float* pre;
    const void*  b;
    const void*  res;
    const float* post;
    void*        out;
    void*        ysum;
    float*       msqPartial;   // one float per workgroup (sum of squares of the value before `post`), or NULL
    int64_t      M;            // frames * H * W
    int64_t      tShift;       // pixels between consecutive time steps (= clips * H * W)
    int          H, W, Ci, Co, kt, kh, kw;
    int          xStride;      // elements between consecutive pixels of x (>= Ci: x may be a channel slice of a wider tensor)
    int          reach;        // (kh/2) * W + kw/2: pixels of halo on each side of a tile
    int          bandRows;     // BM + 2 * reach, rounded up to a multiple of 8
    int          nABuf;        // 2 when there is more than one band per tile
    int          nBBuf;        // weight-tile ring: 2 (prefetch one K-step ahead) or 3 (two)
    int          nTiles;       // Co / BN
    int          totalTiles;   // pixel tiles x channel tiles (the persistent kernels walk them; the others have one workgroup per tile)
    float        slopeNeg;     // activation as max-free form: u > 0 ? u : u * slopeNeg (linear 1, relu 0, lrelu alpha)
    float        gain, clamp;
};

// T2D (conv2d_igemm.hip): 'valid' correlation of frames [n][Hi][Wi] -> [n][H][W] (H <= Hi - kh + 1, W <= Wi - kw + 1); a workgroup
// owns a (BM / 16) x 16 pixel tile of one output frame; M, tShift, reach are unused. (A separate type: the kernel arguments of the
// time-major kernels keep their layout.)
struct ConvArgs2D : ConvArgs
{
    int          Hi, Wi;       // input frame
    int          tilesX, tilesY;
    int          oStride;      // elements between consecutive pixels of out (>= Co)
    int          offY, offX;   // output pixel (0, 0) reads input pixels (offY + dh, offX + dw)
    int          coBase;       // first output channel of this launch (a launch may cover a channel range of the Co channels)
    int          coOut;        // PLANES kernels: channels of the NCHW output tensor (<= Co; channels past it are computed and dropped)
    // PLANES kernels, optional: dot[(tile, pixel half)][c] = sum over the half tile's pixels of acc * oth[n][c][pixel] for c < cDotA + cDotB, oth = the
    // channel-concatenation of two NCHW tensors of the output's plane size (the data gradient's d mod = sum dx * x, formed on the accumulators)
    const void*  dotA; const void* dotB;
    float*       dotPartial;
    int          cDotA, cDotB;
};

template <bool T2D> struct ConvArgsOf { typedef ConvArgs type; };
template <> struct ConvArgsOf<true> { typedef ConvArgs2D type; };
__device__ __forceinline__ int ConvArgs2DStride(const ConvArgs2D& p) { return p.oStride; }
__device__ __forceinline__ int ConvArgs2DStride(const ConvArgs& p) { return p.Co; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <class T> struct Mma;
template <> struct Mma<bf16_t>
{
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, const f32x16& c)
    {
        bf16x8 av, bv;
        __builtin_memcpy(&av, &a, 16);
        __builtin_memcpy(&bv, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t>
{
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, const f32x16& c)
    {
        f16x8 av, bv;
        __builtin_memcpy(&av, &a, 16);
        __builtin_memcpy(&bv, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    }
};

// Measurement builds only (-DLVG_CONV_ABL=bits): 1 no staging in the K loop, 2 no MFMA, 4 no fragment reads,
// 16 no wait / barrier, 64 no band staging, 128 no weight staging, 256 no output stores, 512 direct 8-byte stores (no LDS transpose).
// The shipped library is built with 0.
#ifndef LVG_CONV_ABL
#define LVG_CONV_ABL 0
#endif
constexpr int kAbl = LVG_CONV_ABL;
#ifndef LVG_CONV_PIN
#define LVG_CONV_PIN 1
#endif
constexpr bool kPinOrder = LVG_CONV_PIN != 0;
#ifndef LVG_CONV_PRIO
#define LVG_CONV_PRIO 1
#endif
constexpr int kPrio = LVG_CONV_PRIO;        // 1: the arithmetic of a K-step (fragment reads + MFMAs) runs at wave priority 1, staging / waits at 0
constexpr int kBK = 64;       // input channels per K-step (one 128-byte LDS row)
constexpr int kRowBytes = kBK * 2;
constexpr int kZeroBytes = 1024;   // LDS [0, 1024): zeros (what masked lanes read); the tiles follow

// One 1-KiB piece (8 LDS rows x 128 B) per wave instruction: lane -> (row in piece, physical chunk).
// Returns the LOGICAL 16-byte chunk this lane must fetch so that the lane-linear LDS image is the swizzled one.
__device__ __forceinline__ int piece_chunk(int piece, int lane)
{
    const int row = piece * 8 + (lane >> 3);
    return (lane & 7) ^ ((row >> 1) & 7);
}

// LDS-DMA of 16 bytes per lane: LDS address = ldsPiece (wave-uniform, via M0) + lane * 16; source = base (SGPR pair)
// + 32-bit lane offset. Inline assembly on purpose: hipcc orders every later ds_read behind a
// `__builtin_amdgcn_global_load_lds` it has seen (s_waitcnt vmcnt(0) before the first fragment read of the K-step,
// i.e. no overlap of the prefetch with the MFMAs of the same wave); an asm statement is outside its bookkeeping, the
// kernel waits (vmcnt) itself before the barrier that publishes the tile. M0 is not used by anything else here.
__device__ __forceinline__ void dma16(const unsigned char* base, uint32_t laneOff, uint32_t ldsPiece)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(laneOff), "s"(base), "s"(ldsPiece) : "memory");
}

// f(integral_constant<int, 0>), ..., f(integral_constant<int, N - 1>): a loop whose counter is a compile-time constant in the body
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

#ifndef LVG_CONV2D_STATIC
#define LVG_CONV2D_STATIC 1
#endif
constexpr bool kStaticTaps = LVG_CONV2D_STATIC != 0;   // T2D: K loop unrolled over the 9 taps of a band with compile-time tap geometry (0: the generic loop; A/B builds)
#ifndef LVG_CONV3D_STATIC
#define LVG_CONV3D_STATIC 1
#endif
constexpr bool kStatic3D = LVG_CONV3D_STATIC != 0;     // time-major kernel, 3 x 3 spatial taps: the same unrolled K loop (masks kept); 0: the generic loop (A/B builds)
#ifndef LVG_CONV3D_STATIC_BN64
#define LVG_CONV3D_STATIC_BN64 0
#endif
constexpr bool kStatic3DBn64 = LVG_CONV3D_STATIC_BN64 != 0;   // ... also on the 64-channel tiles (costs them registers: 116 -> 236, i.e. resident workgroups)
#ifndef LVG_CONV2D_SPREAD
#define LVG_CONV2D_SPREAD 0
#endif
constexpr bool kSpreadDma = LVG_CONV2D_SPREAD != 0;    // T2D static loop: the LDS-DMA pieces of a K-step are issued between its MFMA sub-steps instead of all in front

template <int N> __device__ __forceinline__ void wait_vm_const()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// T2D tiles: TW = 16 output pixels wide, BM / 16 rows; the band is the tile's (BM / 16 + kh - 1) x (16 + kw - 1) input patch, one LDS
// row per patch pixel, patch row pitch kPatchPitch (kw = 3). 16-byte chunk c of the LDS row of patch column hx lives at chunk
// c ^ ((hx >> 1) & 7): the 16 lanes of a ds_read_b128 group are 16 pixels of two tile rows whose patch columns hx are distinct modulo 16
// (lanes 0-3, 12-15 of one row, 4-11 of the next), and the row parity of an LDS row is the parity of hx (even pitch) -- conflict-free
// for every tap, like the keyed-by-row layout of the contiguous band.
constexpr int kTileW = 16;
constexpr int kPatchPitch = kTileW + 2;

// OUTF: the result leaves as float32 straight from the accumulators (one 16-byte store per lane and register quad; `out` is a float
// tensor, `ysum` unused) -- the output side of the float32-accurate contraction built from split 16-bit operands (conv2d_frames.py).
// PERSIST (time-major kernel with spatial taps, generic K loop): a workgroup walks a strided sequence of tiles of its XCD's range. The band of
// the NEXT tile is staged during the K-steps of this one (two band buffers always), the weight ring runs on into the next tile's first
// tiles, and the epilogue stages the output rows in the band buffer the tile has just finished with -- so the HBM reads of tile i + 1, the
// matrix work of tile i and the stores of tile i - 1 overlap inside ONE workgroup (the layers with few K-steps per tile -- 64 channels --
// spent their time in the serial prologue -> K loop -> stores of each workgroup: profiles/r04_conv_abl64.log).
// STATIC1 (time-major kernel, 64-channel tiles): the static-tap K loop for tiles whose LDS footprint (two bands) leaves room for ONE workgroup per
// CU anyway -- the loop's 236 registers, which cost the two-workgroup case its occupancy, are free there (eight waves = two per SIMD = 256
// registers each), and the generic loop's band waves (137-165 scalar + 88 vector instructions per K-step against 8 MFMAs) paced those tiles.
// PLANES (2-D tiles, 16-bit output, round 6): the result leaves as NCHW PLANES, out[n][co][oy][ox] = pre[n][co] * acc, for the consumer that tiles planes
// (filtered_lrelu) -- instead of channels-last rows followed by a transposing pass over the whole tensor (csrc/modconv2d_layout.hip, 6 % of a super-resolution
// training iteration). The two MFMA operands have the same register format, so exchanging them yields the transposed result block: a lane then holds ONE output
// channel and four horizontally adjacent pixels per register quad. Staged as [channel][pixel] rows in LDS, a lane reads eight adjacent pixels of a channel
// (16 bytes) and stores them into the plane: 32-byte runs per (channel, tile row); the neighbouring tile -- same XCD, next in its order -- completes the line in L2.
template <class T, int BM, int BN, int PB, int NB, bool T2D = false, bool OUTF = false, bool PERSIST = false, bool STATIC1 = false, bool PLANES = false>
__global__ __launch_bounds__(BM / (32 * PB) * 128, STATIC1 ? 1 : 2) void conv3d_igemm_kernel(typename ConvArgsOf<T2D>::type p)
{
    static_assert(!PLANES || (T2D && !OUTF && PB == 2), "plane output: 2-D tiles, 16-bit output");
    static_assert(!PERSIST || (!T2D && !OUTF), "persistent form: time-major frames, 16-bit output");
    static_assert(!STATIC1 || (!T2D && !OUTF && !PERSIST && NB == 2), "one-workgroup static form: time-major frames, 16-bit output, ring of two");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW  = BM / (32 * PB) * 2;      // waves: BM / (32 PB) along the pixels x 2 along the output channels
    constexpr int NCB = BN / 64;                 // 32-channel MFMA blocks per wave
    constexpr int NWA = NW / 4;                  // band-staging waves (the last NWA) when the kernel has spatial taps
    constexpr int NWB = NW - NWA;                // weight-staging waves then
    constexpr int NBP = BN / 8;                  // weight pieces per K-step
    constexpr int NBI = (NBP + NWB - 1) / NWB;   // ... per wave, at most
    constexpr int MAXAI = 6;                     // band pieces per wave and K-step (host guarantees)
    constexpr int bBytes = BN * kRowBytes;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware tile order: the dispatcher puts workgroup b on XCD b % 8; give every XCD a contiguous range of
    // tiles (channel tile fastest), so that the workgroups sharing an L2 share bands and walk the weights together.
    const int nwg = PERSIST ? p.totalTiles : (int)gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int xcdFirst = xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q;
    int tile = xcdFirst + (bid >> 3);
    // PERSIST: this workgroup's tiles are tile, tile + tileStep, ... below tileEnd (the workgroups of an XCD walk its range side by side:
    // neighbouring tiles -- shared halo rows, the same weights -- are in flight together)
    const int tileEnd = xcdFirst + q + (xcd < rm ? 1 : 0), tileStep = (int)(gridDim.x >> 3);
    if constexpr (PERSIST) { if (tile >= tileEnd) return; }
    int mt = tile / p.nTiles, nt = tile - mt * p.nTiles;
    int64_t m0 = (int64_t)mt * BM;
    int co0 = nt * BN;
    if constexpr (T2D) co0 += p.coBase;
    int64_t aM0 = m0;                                                  // the tile whose band `issueA` stages (PERSIST: also the next tile's)
    // T2D: spatial tile -> (frame, tile row, tile column)
    int n2 = 0, y0 = 0, x0 = 0;
    if constexpr (T2D)
    {
        const int perFrame = p.tilesX * p.tilesY;
        n2 = mt / perFrame;
        const int rem = mt - n2 * perFrame;
        const int tyI = rem / p.tilesX;
        y0 = tyI * (BM / kTileW);
        x0 = (rem - tyI * p.tilesX) * kTileW;
    }

    // LDS: [zero page | nBBuf weight tiles | nABuf bands]
    const int aBytes = p.bandRows * kRowBytes;
    const uint32_t ldsBase = (uint32_t)(uintptr_t)smem;                 // generic -> LDS byte address (low 32 bits)
    const int bOff = kZeroBytes, aOff = kZeroBytes + p.nBBuf * bBytes;

    const int ntap = p.kh * p.kw;
    const int nchunk = p.Ci / kBK;
    const int nMacro = p.kt * nchunk;
    const int nSteps = nMacro * ntap;
    const int nAI = p.bandRows >> 3;                                   // band pieces in total
    // weight-tile ring: NB slots with spatial taps (tiles staged NB - 1 K-steps ahead), 2 without (host sets nBBuf): `dist` below
    // Without spatial taps (one K-step per band) everything is staged by all waves one K-step ahead (host: nBBuf = 2).
    const bool split = ntap > 1;
    const int bWaves = split ? NWB : NW, aWaves = split ? NWA : NW, aWave0 = split ? NWB : 0;
    const bool isB = wave < bWaves;
    const int aPerStep = (nAI + aWaves * ntap - 1) / (aWaves * ntap);  // per band wave and K-step (<= MAXAI)
    const int aPerStep0 = (nAI + NW * ntap - 1) / (NW * ntap);         // prologue: all waves
    const int pt = p.kt >> 1;

    const unsigned char* const xb = static_cast<const unsigned char*>(p.x);
    const unsigned char* const wb = static_cast<const unsigned char*>(p.w);
    const uint32_t rowStride = (uint32_t)p.Ci * 2;                     // bytes per weight row
    const uint32_t xRowStride = (uint32_t)p.xStride * 2;               // bytes per pixel of x

    if (tid < kZeroBytes / 16) reinterpret_cast<uint4*>(smem)[tid] = make_uint4(0, 0, 0, 0);

    // ---- staging -----------------------------------------------------------------------------------------------
    // Weight tiles: per lane and piece the source offset inside a tile is fixed; the tile base is a running pointer.
    uint32_t bLaneOff[NBI];
    int bPiece[NBI];                                                   // LDS piece of each; a wave with fewer pieces repeats its last
    #pragma unroll
    for (int i = 0; i < NBI; i++)
    {
        int piece = wave + i * bWaves;
        if (piece >= NBP) piece -= bWaves;                             // NBI - 1 pieces exist for every weight wave
        bPiece[i] = piece;
        bLaneOff[i] = (uint32_t)(piece * 8 + (lane >> 3)) * rowStride + piece_chunk(piece, lane) * 16;
    }
    const uint64_t tapStride = (uint64_t)p.Co * rowStride;             // bytes between the tiles of consecutive taps
    const uint64_t macroJump = (uint64_t)ntap * tapStride - (uint64_t)(nchunk - 1) * kRowBytes;   // last chunk -> next temporal tap
    const unsigned char* wMacro = wb + (uint64_t)co0 * rowStride;      // tap 0 of the band the NEXT staged tile belongs to
    const unsigned char* wNext = wMacro;                               // the next tile to stage
    int nTap = 0, nKc = 0;                                             // its (spatial tap, chunk)
    auto issueB = [&](int buf) -> int
    {
        int issued = 0;
        #pragma unroll
        for (int i = 0; i < NBI; i++)
        {
            const int piece = wave + i * bWaves;
            if (isB && piece < NBP)
            {
                dma16(wNext, bLaneOff[i], ldsBase + bOff + buf * bBytes + piece * 1024);
                issued++;
            }
        }
        // advance to the tile of the next K-step: next tap; then next chunk of the same temporal tap; then next temporal tap
        if (++nTap == ntap)
        {
            nTap = 0;
            if (++nKc == nchunk) { nKc = 0; wMacro += macroJump; }
            else wMacro += kRowBytes;
            wNext = wMacro;
        }
        else wNext += tapStride;
        return issued;
    };
    // Band pieces of (dt, kc) that this wave brings in during K-step `tapSlot` of the previous band (waves w0 .. w0 + nw - 1
    // share the band, `per` pieces per wave and K-step). Source rows are clamped into the tensor: clamped rows are only
    // ever read by masked lanes.
    const uint32_t aChunkOff = (uint32_t)(lane & 7);
    auto issueA = [&](int dt, int kc, int tapSlot, int buf, int per, int nw, int w0)
    {
        const int g0 = (int)(aM0 - p.reach + (int64_t)(dt - pt) * p.tShift);    // |.| < 2^31 (host check)
        const int last = (int)p.M - 1;
        #pragma unroll
        for (int i = 0; i < MAXAI; i++)
        {
            const int piece = (tapSlot * per + i) * nw + (wave - w0);
            if (i < per && wave >= w0 && piece < nAI)
            {
                if constexpr (T2D)
                {
                    // LDS row r of the band = patch pixel (hy, hx); rows / columns past the input frame (tiles overhanging the
                    // output frame, spare rows of the last piece) are clamped: only outputs that are never stored read them
                    const int r = piece * 8 + (lane >> 3);
                    const int hy = r / kPatchPitch, hx = r - hy * kPatchPitch;
                    int yy = y0 + p.offY + hy, xx = x0 + p.offX + hx;
                    yy = yy > p.Hi - 1 ? p.Hi - 1 : yy;
                    xx = xx > p.Wi - 1 ? p.Wi - 1 : xx;
                    const uint32_t chunk = aChunkOff ^ (uint32_t)((hx >> 1) & 7);
                    const uint32_t off = (uint32_t)((n2 * p.Hi + yy) * p.Wi + xx) * xRowStride + (uint32_t)kc * kRowBytes + chunk * 16;  // < 2^32 (host check)
                    dma16(xb, off, ldsBase + aOff + buf * aBytes + piece * 1024);
                }
                else
                {
                    int g = g0 + piece * 8 + (lane >> 3);
                    g = g < 0 ? 0 : (g > last ? last : g);
                    const uint32_t chunk = aChunkOff ^ (uint32_t)(((piece * 8 + (lane >> 3)) >> 1) & 7);
                    const uint32_t off = (uint32_t)g * xRowStride + (uint32_t)kc * kRowBytes + chunk * 16;  // < 2^32 (host check)
                    dma16(xb, off, ldsBase + aOff + buf * aBytes + piece * 1024);
                }
            }
        }
    };

    // State that runs on from tile to tile in the persistent form: the weight ring position, the parity of the band buffers.
    const int dist = p.nBBuf - 1;                                     // weight-tile ring: tiles are staged `dist` K-steps ahead
    int bufNext = dist;                                               // ring slot the next staged weight tile goes to
    uint32_t curB = 0;                                                // byte offset of the current weight tile in the ring
    uint32_t stageOffB = (uint32_t)((NB - 1) * bBytes);               // (weight waves of the split loop) ring offset of the slot being staged
    int gbase = 0;                                                    // bands completed in earlier tiles (PERSIST; 0 otherwise)
    bool firstTile = true;
    (void)firstTile; (void)stageOffB;

next_tile:                                                            // PERSIST: the per-tile part starts over from here (a backwards goto: one body, no re-indentation)
    const int tileNext = tile + tileStep;
    const bool haveNext = PERSIST && tileNext < tileEnd;
    const int mtNext = tileNext / p.nTiles;
    const int64_t m0Next = (int64_t)mtNext * BM;
    const int co0Next = (tileNext - mtNext * p.nTiles) * BN;
    (void)haveNext; (void)m0Next; (void)co0Next;

    // ---- which temporal taps and which spatial taps read a real pixel, per lane and pixel block ------------------
    uint32_t vmask[PB];
    int jrow[PB];
    const int txl = l31 & (kTileW - 1);                                // T2D: this lane's column inside the tile
    #pragma unroll
    for (int pb = 0; pb < PB; pb++)
    {
        const int j = wr * (32 * PB) + pb * 32 + l31;
        jrow[pb] = j;
        if constexpr (T2D)
        {
            jrow[pb] = (j / kTileW) * kPatchPitch + txl;               // band row of tap (0, 0); every tap reads a staged pixel: no masks
            vmask[pb] = 0xffffffffu;
            continue;
        }
        const int64_t m = m0 + j;
        uint32_t mask = 0;
        if (m < p.M)
        {
            const uint32_t row = (uint32_t)m / (uint32_t)p.W;         // M < 2^31 (host check)
            const int ww = (int)((uint32_t)m - row * (uint32_t)p.W);
            const int hh = (int)(row % (uint32_t)p.H);
            uint32_t sp = 0;
            for (int dh = 0; dh < p.kh; dh++)
                for (int dw = 0; dw < p.kw; dw++)
                {
                    const int y = hh + dh - (p.kh >> 1), x = ww + dw - (p.kw >> 1);
                    if (y >= 0 && y < p.H && x >= 0 && x < p.W) sp |= 1u << (dh * p.kw + dw);
                }
            mask = sp;                                                // bits 0 .. 24: spatial taps; bits 25 .. 31: temporal taps
            for (int dt = 0; dt < p.kt; dt++)
            {
                const int64_t ms = m + (int64_t)(dt - pt) * p.tShift;
                if (ms >= 0 && ms < p.M) mask |= 1u << (25 + dt);
            }
        }
        vmask[pb] = mask;
    }

    f32x16 acc[NCB][PB];
    #pragma unroll
    for (int cb = 0; cb < NCB; cb++)
        #pragma unroll
        for (int pb = 0; pb < PB; pb++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[cb][pb][r] = 0.f;

    // Fragment addresses. Row r, logical chunk c = 2 ks + hi lives at byte r * 128 + ((c ^ key(r)) << 4), key = (r >> 1) & 7,
    // and (2 ks + hi) ^ key = (hi ^ key) ^ 2 ks: address = rowBase + (e ^ (ks << 5)) with e = (hi ^ key) << 4.
    uint32_t wAddr[NCB][4];                                           // weight fragments: fixed per lane (+ ring position)
    #pragma unroll
    for (int cb = 0; cb < NCB; cb++)
    {
        const int row = wc * (BN / 2) + cb * 32 + l31;
        const uint32_t e = (uint32_t)(hi ^ ((row >> 1) & 7)) << 4;
        #pragma unroll
        for (int ks = 0; ks < 4; ks++) wAddr[cb][ks] = (uint32_t)(bOff + row * kRowBytes) + (e ^ (uint32_t)(ks << 5));
    }

    // ---- prologue: first band (all waves), first `dist` weight tiles ------------------------------------------------
    if (!PERSIST || firstTile)                                         // (later tiles of a persistent workgroup: staged during the previous tile)
    {
        for (int t = 0; t < ntap; t++) issueA(0, 0, t, 0, aPerStep0, NW, 0);
        for (int d = 0; d < dist; d++)
            if (d < nSteps) issueB(d);
        wait_vm_const<0>();
        __syncthreads();
    }

    // ---- K loop: (dt, kc) = band, tap = spatial tap inside it --------------------------------------------------------
    int dt = 0, kc = 0, tap = 0, dh = 0, dw = 0, macro = 0;
    (void)dh;
    uint32_t curA = (uint32_t)aOff + (uint32_t)((gbase & 1) * aBytes);   // byte offset of the current band
    const uint32_t ringBytes = (uint32_t)(p.nBBuf * bBytes);

    // One K-step of arithmetic: fragment addresses of this tap, then 4 x (fragment reads, MFMAs) with the reads of
    // sub-step ks + 1 issued ahead of the MFMAs of sub-step ks (two fragment register sets).
    int shift = 0, tbit = 25;                                         // dh * W + dw and 25 + dt of the current K-step
    const int pitch = T2D ? kPatchPitch : p.W;                        // band rows between vertically adjacent pixels
    auto compute = [&]() __attribute__((always_inline))
    {
        uint32_t xBase[PB], xE[PB];
        #pragma unroll
        for (int pb = 0; pb < PB; pb++)
        {
            const uint32_t rb = (uint32_t)(jrow[pb] + shift);
            if constexpr (T2D)
            {
                xE[pb] = ((uint32_t)hi ^ (((uint32_t)(txl + dw) >> 1) & 7u)) << 4;
                xBase[pb] = curA + (rb << 7);
                continue;
            }
            const bool ok = (vmask[pb] >> tap) & (vmask[pb] >> tbit) & 1u;
            xE[pb] = ((uint32_t)hi ^ ((rb >> 1) & 7u)) << 4;
            // masked lanes read zeros from the 256-byte zero page at the SAME bank position (row parity kept): the
            // conflict-free bank pattern of the group survives
            xBase[pb] = ok ? curA + (rb << 7) : ((rb & 1u) << 7);
        }
        uint4 wf[2][NCB], xf[2][PB];
        auto fetch = [&](int ks, int set) __attribute__((always_inline))
        {
            if constexpr (!(kAbl & 4))
            {
                #pragma unroll
                for (int cb = 0; cb < NCB; cb++)
                    wf[set][cb] = *reinterpret_cast<const uint4*>(smem + (wAddr[cb][ks] + curB));
                #pragma unroll
                for (int pb = 0; pb < PB; pb++)
                    xf[set][pb] = *reinterpret_cast<const uint4*>(smem + (xBase[pb] + (xE[pb] ^ (uint32_t)(ks << 5))));
            }
            else
            {
                #pragma unroll
                for (int cb = 0; cb < NCB; cb++) wf[set][cb] = make_uint4(ks, tap, cb, 1);
                #pragma unroll
                for (int pb = 0; pb < PB; pb++) xf[set][pb] = make_uint4(ks, tap, pb, xBase[pb]);
            }
        };
        if constexpr (kPrio == 1) __builtin_amdgcn_s_setprio(1);
        fetch(0, 0);
        #pragma unroll
        for (int ks = 0; ks < kBK / 16; ks++)
        {
            if (ks + 1 < kBK / 16) fetch(ks + 1, (ks + 1) & 1);
            if constexpr (!(kAbl & 2))
            {
                #pragma unroll
                for (int pb = 0; pb < PB; pb++)
                    #pragma unroll
                    for (int cb = 0; cb < NCB; cb++)
                        acc[cb][pb] = PLANES ? Mma<T>::run(xf[ks & 1][pb], wf[ks & 1][cb], acc[cb][pb]) : Mma<T>::run(wf[ks & 1][cb], xf[ks & 1][pb], acc[cb][pb]);
            }
            else
            {
                #pragma unroll
                for (int cb = 0; cb < NCB; cb++)
                    #pragma unroll
                    for (int pb = 0; pb < PB; pb++)
                        acc[cb][pb][ks] += __uint_as_float(wf[ks & 1][cb].x ^ xf[ks & 1][pb].y ^ wf[ks & 1][cb].w ^ xf[ks & 1][pb].z);
            }
        }
        // Pin the order the source states (hipcc otherwise re-uses ONE fragment register set and issues the reads of sub-step
        // ks + 1 behind the MFMAs of sub-step ks, exposing the LDS latency four times per K-step): reads(0), then
        // 3 x [reads(ks + 1), MFMAs(ks)], MFMAs(3).   masks: 0x100 = DS read, 0x8 = MFMA
        if constexpr (kAbl == 0 && kPinOrder)
        {
            __builtin_amdgcn_sched_group_barrier(0x100, NCB + PB, 0);
            #pragma unroll
            for (int ks = 0; ks < kBK / 16 - 1; ks++)
            {
                __builtin_amdgcn_sched_group_barrier(0x100, NCB + PB, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, NCB * PB, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x8, NCB * PB, 0);
        }
        if constexpr (kPrio == 1) __builtin_amdgcn_s_setprio(0);
    };
    // counters of the next K-step
    auto advance = [&]() __attribute__((always_inline))
    {
        curB += bBytes;
        if (curB == ringBytes) curB = 0;
        if (++bufNext == p.nBBuf) bufNext = 0;
        shift++;
        if (++dw == p.kw) { dw = 0; dh++; shift += pitch - p.kw; }
        if (++tap == ntap)
        {
            tap = 0; dh = 0; dw = 0; shift = 0; macro++;
            if (++kc == nchunk) { kc = 0; dt++; tbit++; }
            curA = (uint32_t)aOff + (uint32_t)(((gbase + macro) & 1) * aBytes);  // nABuf == 1 only when there is one band
        }
    };

    if constexpr (T2D && kStaticTaps)
    {
        // ---- 2-D tiles, 3 x 3 taps (the host guarantees kt = 1, kh = kw = 3): the nine K-steps of a band are unrolled with the tap as
        // a compile-time constant. Measured motivation (instruction mix of the generic loop, tools/isa_loop_count.py): the weight waves
        // issue ~75 scalar + vector instructions next to the 16 MFMAs of a K-step, the BAND wave ~275 (piece -> patch pixel -> clamped
        // source offset for three pieces, tap counters, the generic staging bookkeeping) -- and every K-step ends in a barrier, so the
        // band wave paces the workgroup. Here everything a K-step needs is precomputed: fragment addresses per (tap column, pixel block,
        // sub-step) with the tap row as an immediate offset, the lane offsets of all the band pieces of this wave, and the staging
        // pointers advance by constants.
        constexpr int NTAP = 9;
        constexpr int NPIECE = ((BM / kTileW + 2) * kPatchPitch + 7) / 8;     // 1-KiB pieces of a band
        constexpr int NPW = (NPIECE + NWA - 1) / NWA;                         // ... per band wave
        constexpr int PER = (NPW + NTAP - 1) / NTAP;                          // ... per band wave and K-step
        static_assert(PER <= MAXAI, "band pieces per K-step");
        uint32_t xA[3][PB][4];
        #pragma unroll
        for (int c = 0; c < 3; c++)
            #pragma unroll
            for (int pb = 0; pb < PB; pb++)
                #pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    xA[c][pb][ks] = ((uint32_t)(jrow[pb] + c) << 7) + ((((uint32_t)hi ^ (((uint32_t)(txl + c) >> 1) & 7u)) << 4) ^ (uint32_t)(ks << 5));
        // one K-step of arithmetic on band `bandOff`, weight tile `wOff` (byte offsets in LDS, wave-uniform)
        auto mma_step = [&](auto tapc, uint32_t bandOff, uint32_t wOff, auto&& stage) __attribute__((always_inline))
        {
            constexpr int TAP = decltype(tapc)::value, DH = TAP / 3, DW = TAP % 3;
            uint4 wf[2][NCB], xf[2][PB];
            auto fetch = [&](int ks, int set) __attribute__((always_inline))
            {
                #pragma unroll
                for (int cb = 0; cb < NCB; cb++)
                    wf[set][cb] = *reinterpret_cast<const uint4*>(smem + (wAddr[cb][ks] + wOff));
                #pragma unroll
                for (int pb = 0; pb < PB; pb++)
                    xf[set][pb] = *reinterpret_cast<const uint4*>(smem + (xA[DW][pb][ks] + bandOff) + DH * kPatchPitch * kRowBytes);
            };
            if constexpr (!kSpreadDma) stage(std::integral_constant<int, -1>{});
            if constexpr (kPrio == 1) __builtin_amdgcn_s_setprio(1);
            fetch(0, 0);
            static_for<kBK / 16>([&](auto ksc) __attribute__((always_inline))
            {
                constexpr int ks = decltype(ksc)::value;
                if constexpr (ks + 1 < kBK / 16) fetch(ks + 1, (ks + 1) & 1);
                #pragma unroll
                for (int pb = 0; pb < PB; pb++)
                    #pragma unroll
                    for (int cb = 0; cb < NCB; cb++)
                        acc[cb][pb] = PLANES ? Mma<T>::run(xf[ks & 1][pb], wf[ks & 1][cb], acc[cb][pb]) : Mma<T>::run(wf[ks & 1][cb], xf[ks & 1][pb], acc[cb][pb]);
                // this sub-step's share of the staging: the DMA instructions (volatile asm) sit between the MFMA groups in program order
                if constexpr (kSpreadDma) stage(ksc);
            });
            if constexpr (kPinOrder && !kSpreadDma)
            {
                __builtin_amdgcn_sched_group_barrier(0x100, NCB + PB, 0);
                #pragma unroll
                for (int ks = 0; ks < kBK / 16 - 1; ks++)
                {
                    __builtin_amdgcn_sched_group_barrier(0x100, NCB + PB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, NCB * PB, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x8, NCB * PB, 0);
            }
            if constexpr (kPrio == 1) __builtin_amdgcn_s_setprio(0);
        };
        const uint32_t ldsB0 = ldsBase + bOff;
        if (isB)
        {
            // weight waves: the tile of the NEXT K-step goes into the other ring slot (the prologue staged the first tile; wNext = the second)
            for (macro = 0; macro < nMacro; macro++)
            {
                const uint32_t bandOff = (uint32_t)aOff + (uint32_t)((macro & 1) * aBytes);
                const bool more = macro + 1 < nMacro;
                static_for<NTAP>([&](auto tapc) __attribute__((always_inline))
                {
                    constexpr int TAP = decltype(tapc)::value;
                    const uint32_t stageOff = curB ^ (uint32_t)bBytes;              // two slots: the one not being read
                    const unsigned char* const wStage = wNext;
                    if constexpr (TAP == NTAP - 2)
                    {
                        // the tile after the last tap of this band is the first tap of the next band (on the last band: this band's first
                        // tile once more, into a slot nobody reads again -- keeps the loop free of branches around the staging)
                        if (more) wMacro += kRowBytes;
                        wNext = wMacro;
                    }
                    else wNext += tapStride;
                    // NBI pieces: all in front of the arithmetic (sub-step -1), or spread over the first three sub-steps
                    constexpr int PER_KS = (NBI + 2) / 3;
                    mma_step(tapc, bandOff, curB, [&](auto ksc) __attribute__((always_inline))
                    {
                        constexpr int ks = decltype(ksc)::value;
                        #pragma unroll
                        for (int i = 0; i < NBI; i++)
                            if (ks < 0 || (i / PER_KS == ks))
                                dma16(wStage, bLaneOff[i], ldsB0 + stageOff + bPiece[i] * 1024);
                    });
                    wait_vm_const<0>();
                    __syncthreads();
                    curB ^= (uint32_t)bBytes;
                });
            }
        }
        else
        {
            // band waves: lane offsets of this wave's pieces of a band (piece index = slot * NWA + wave - first band wave), relative to
            // the 64-channel chunk; the pieces of the next band are spread over the nine K-steps of this one and waited for at the last
            const int wsub = NWA == 1 ? 0 : wave - NWB;                               // this wave among the band waves
            uint32_t aLane[NPW];
            #pragma unroll
            for (int sl = 0; sl < NPW; sl++)
            {
                const int piece = sl * NWA + wsub;
                const int r = piece * 8 + (lane >> 3);
                const int hy = r / kPatchPitch, hx = r - hy * kPatchPitch;
                int yy = y0 + p.offY + hy, xx = x0 + p.offX + hx;
                yy = yy > p.Hi - 1 ? p.Hi - 1 : yy;
                xx = xx > p.Wi - 1 ? p.Wi - 1 : xx;
                aLane[sl] = (uint32_t)((n2 * p.Hi + yy) * p.Wi + xx) * xRowStride + (aChunkOff ^ (uint32_t)((hx >> 1) & 7)) * 16;
            }
            const uint32_t pieceLds0 = ldsBase + (uint32_t)aOff + (uint32_t)(wsub * 1024);
            for (macro = 0; macro < nMacro; macro++)
            {
                const uint32_t bandOff = (uint32_t)aOff + (uint32_t)((macro & 1) * aBytes);
                const uint32_t nextLds = pieceLds0 + (uint32_t)(((macro + 1) & 1) * aBytes);
                const bool more = macro + 1 < nMacro;
                const unsigned char* const xNext = xb + (uint32_t)(macro + 1) * kRowBytes;      // next 64-channel chunk
                static_for<NTAP>([&](auto tapc) __attribute__((always_inline))
                {
                    constexpr int TAP = decltype(tapc)::value;
                    mma_step(tapc, bandOff, curB, [&](auto ksc) __attribute__((always_inline))
                    {
                        constexpr int ks = decltype(ksc)::value;
                        if (more)
                        {
                            #pragma unroll
                            for (int i = 0; i < PER; i++)
                            {
                                const int sl = TAP * PER + i;                        // compile-time after unrolling
                                if ((ks < 0 || i == ks) && sl < NPW && sl * NWA + wsub < NPIECE)
                                    dma16(xNext, aLane[sl < NPW ? sl : 0], nextLds + (uint32_t)(sl * NWA * 1024));
                            }
                        }
                    });
                    if constexpr (TAP == NTAP - 1) wait_vm_const<0>();
                    __syncthreads();
                    curB ^= (uint32_t)bBytes;
                });
            }
        }
    }
    else if (!PERSIST && !T2D && kStatic3D && (BN == 128 || kStatic3DBn64 || STATIC1) && NB == 2 && split && p.kh == 3 && p.kw == 3 && nAI <= NWA * 9 * ((BN == 128 || BM == 256) ? 3 : 4))
    {
        // ---- time-major frames, 3 x 3 spatial taps (any number of temporal taps): the static-tap loop of the 2-D kernel with the 'same'
        // padding masks kept. Per (tap, pixel block) the band row and its swizzle phase are precomputed as ONE address word; a K-step
        // selects between it (+ the band's offset) and the zero page by one bit test. The band waves keep the lane offsets of their
        // pieces in registers (rebuilt when the temporal tap changes: the band of tap dt is the band shifted by whole frames).
        // band pieces per band wave and K-step: at most SLOTS (3 where the accumulators leave fewer registers: frames up to 64 pixels wide
        // need 25 pieces per band wave there, 33 on the 64-channel tiles)
        constexpr int NTAP = 9, SLOTS = (BN == 128 || BM == 256) ? 3 : 4;
        const int Wd = p.W;
        uint32_t xA0[NTAP][PB];
        #pragma unroll
        for (int t = 0; t < NTAP; t++)
            #pragma unroll
            for (int pb = 0; pb < PB; pb++)
            {
                const uint32_t rb = (uint32_t)(jrow[pb] + (t / 3) * Wd + (t % 3));
                xA0[t][pb] = (rb << 7) + (((uint32_t)hi ^ ((rb >> 1) & 7u)) << 4);       // row address + chunk of sub-step 0 (sub-step ks: ^ ks << 5)
            }
        uint32_t tm[PB];                                                       // this band's mask word: the spatial bits where the temporal tap is valid, else 0
        auto set_tm = [&]() __attribute__((always_inline))
        {
            #pragma unroll
            for (int pb = 0; pb < PB; pb++) tm[pb] = ((vmask[pb] >> tbit) & 1u) ? vmask[pb] : 0u;
        };
        auto mma_step3 = [&](auto tapc, uint32_t bandOff, uint32_t wOff) __attribute__((always_inline))
        {
            constexpr int TAP = decltype(tapc)::value;
            uint32_t xB[PB];
            #pragma unroll
            for (int pb = 0; pb < PB; pb++)
                xB[pb] = (tm[pb] & (1u << TAP)) ? xA0[TAP][pb] + bandOff : (xA0[TAP][pb] & 0xffu);     // masked lanes: zero page, same bank position
            uint4 wf[2][NCB], xf[2][PB];
            auto fetch = [&](int ks, int set) __attribute__((always_inline))
            {
                #pragma unroll
                for (int cb = 0; cb < NCB; cb++)
                    wf[set][cb] = *reinterpret_cast<const uint4*>(smem + (wAddr[cb][ks] + wOff));
                #pragma unroll
                for (int pb = 0; pb < PB; pb++)
                    xf[set][pb] = *reinterpret_cast<const uint4*>(smem + (xB[pb] ^ (uint32_t)(ks << 5)));
            };
            if constexpr (kPrio == 1) __builtin_amdgcn_s_setprio(1);
            fetch(0, 0);
            #pragma unroll
            for (int ks = 0; ks < kBK / 16; ks++)
            {
                if (ks + 1 < kBK / 16) fetch(ks + 1, (ks + 1) & 1);
                #pragma unroll
                for (int pb = 0; pb < PB; pb++)
                    #pragma unroll
                    for (int cb = 0; cb < NCB; cb++)
                        acc[cb][pb] = Mma<T>::run(wf[ks & 1][cb], xf[ks & 1][pb], acc[cb][pb]);
            }
            if constexpr (kPinOrder)
            {
                __builtin_amdgcn_sched_group_barrier(0x100, NCB + PB, 0);
                #pragma unroll
                for (int ks = 0; ks < kBK / 16 - 1; ks++)
                {
                    __builtin_amdgcn_sched_group_barrier(0x100, NCB + PB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, NCB * PB, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x8, NCB * PB, 0);
            }
            if constexpr (kPrio == 1) __builtin_amdgcn_s_setprio(0);
        };
        const uint32_t ldsB0 = ldsBase + bOff;
        if (isB)
        {
            for (macro = 0; macro < nMacro; macro++)
            {
                const uint32_t bandOff = (uint32_t)aOff + (uint32_t)((macro & 1) * aBytes);
                const bool more = macro + 1 < nMacro;
                set_tm();
                static_for<NTAP>([&](auto tapc) __attribute__((always_inline))
                {
                    constexpr int TAP = decltype(tapc)::value;
                    const uint32_t stageOff = curB ^ (uint32_t)bBytes;
                    #pragma unroll
                    for (int i = 0; i < NBI; i++)
                        dma16(wNext, bLaneOff[i], ldsB0 + stageOff + bPiece[i] * 1024);
                    if constexpr (TAP == NTAP - 2)
                    {
                        if (more) wMacro += (kc + 1 == nchunk) ? macroJump : (uint64_t)kRowBytes;
                        wNext = wMacro;
                    }
                    else wNext += tapStride;
                    mma_step3(tapc, bandOff, curB);
                    wait_vm_const<0>();
                    __syncthreads();
                    curB ^= (uint32_t)bBytes;
                });
                if (++kc == nchunk) { kc = 0; dt++; tbit++; }
            }
        }
        else
        {
            const int wsub = NWA == 1 ? 0 : wave - NWB;
            const int npw = (nAI + NWA - 1) / NWA;                               // pieces of a band per band wave
            const int per = (npw + NTAP - 1) / NTAP;                             // ... per K-step (<= SLOTS)
            const int last = (int)p.M - 1;
            uint32_t aLane[NTAP * SLOTS];                                        // slot TAP * SLOTS + i: lane offset of piece ((TAP * per + i) * NWA + wsub) of the band of `tabDt`
            int tabDt = -1;
            auto build = [&](int bdt) __attribute__((always_inline))
            {
                const int g0 = (int)(m0 - p.reach + (int64_t)(bdt - pt) * p.tShift);
                #pragma unroll
                for (int sl = 0; sl < NTAP * SLOTS; sl++)
                {
                    const int piece = ((sl / SLOTS) * per + (sl % SLOTS)) * NWA + wsub;
                    int g = g0 + piece * 8 + (lane >> 3);
                    g = g < 0 ? 0 : (g > last ? last : g);
                    const uint32_t chunk = aChunkOff ^ (uint32_t)(((piece * 8 + (lane >> 3)) >> 1) & 7);
                    aLane[sl] = (uint32_t)g * xRowStride + chunk * 16;
                }
                tabDt = bdt;
            };
            for (macro = 0; macro < nMacro; macro++)
            {
                const uint32_t bandOff = (uint32_t)aOff + (uint32_t)((macro & 1) * aBytes);
                const bool more = macro + 1 < nMacro;
                const int mkc = (kc + 1 == nchunk) ? 0 : kc + 1;
                const int mdt = (kc + 1 == nchunk) ? dt + 1 : dt;
                if (more && mdt != tabDt) build(mdt);
                const unsigned char* const xNext = xb + (uint32_t)mkc * kRowBytes;
                const uint32_t nextLds = ldsBase + (uint32_t)aOff + (uint32_t)(((macro + 1) & 1) * aBytes) + (uint32_t)(wsub * 1024);
                set_tm();
                static_for<NTAP>([&](auto tapc) __attribute__((always_inline))
                {
                    constexpr int TAP = decltype(tapc)::value;
                    if (more)
                    {
                        #pragma unroll
                        for (int i = 0; i < SLOTS; i++)
                        {
                            const int q = (TAP * per + i) * NWA + wsub;
                            if (i < per && q < nAI) dma16(xNext, aLane[TAP * SLOTS + i], nextLds + (uint32_t)((TAP * per + i) * NWA * 1024));
                        }
                    }
                    mma_step3(tapc, bandOff, curB);
                    if constexpr (TAP == NTAP - 1) wait_vm_const<0>();
                    __syncthreads();
                    curB ^= (uint32_t)bBytes;
                });
                if (++kc == nchunk) { kc = 0; dt++; tbit++; }
            }
        }
    }
    else if (!split)
    {
        // no spatial taps: every wave stages its share of both operands one K-step ahead
        for (int step = 0; step < nSteps; step++)
        {
            const int mkc = (kc + 1 == nchunk) ? 0 : kc + 1;
            const int mdt = (kc + 1 == nchunk) ? dt + 1 : dt;
            if (step + dist < nSteps && !(kAbl & (1 | 128))) issueB(bufNext);
            if (macro + 1 < nMacro && !(kAbl & (1 | 64))) issueA(mdt, mkc, tap, (macro + 1) & 1, aPerStep, aWaves, aWave0);
            compute();
            if constexpr (!(kAbl & 16)) { wait_vm_const<0>(); __syncthreads(); }
            advance();
        }
    }
    else if (isB)
    {
        // Weight waves: NBI pieces per K-step, always (where no tile is left to stage an earlier one is staged again into a ring
        // slot nobody reads any more): no branches between the staging and the MFMAs, counted wait. The taps of a band are two
        // plain loops -- while the staged tile (NB - 1 K-steps ahead) still belongs to this band, and after it moved on to the
        // next band -- so the per-step scalar work is the staging itself and a handful of counters.
        constexpr int D = NB - 1;
        const uint32_t ldsB0 = ldsBase + bOff;
        uint32_t& stageOff = stageOffB;                                // ring offset of the slot being staged (starts at D * bBytes)
        auto kstep = [&]() __attribute__((always_inline))
        {
            if constexpr (!(kAbl & (1 | 128)))
            {
                #pragma unroll
                for (int i = 0; i < NBI; i++)
                    dma16(wNext, bLaneOff[i], ldsB0 + stageOff + bPiece[i] * 1024);
                wNext += tapStride;
            }
            compute();
            if constexpr (!(kAbl & 16))
            {
                wait_vm_const<(D > 1 ? NBI : 0)>();
                __syncthreads();
            }
            stageOff += bBytes;
            if (stageOff == ringBytes) stageOff = 0;
            curB += bBytes;
            if (curB == ringBytes) curB = 0;
            tap++;
            shift++;
            if (++dw == p.kw) { dw = 0; dh++; shift += pitch - p.kw; }
        };
        // the prologue staged the first D tiles of band 0 through the generic path: wNext already points D taps in
        for (macro = 0; macro < nMacro; macro++)
        {
            tap = 0; dh = 0; dw = 0; shift = 0;
            curA = (uint32_t)aOff + (uint32_t)(((gbase + macro) & 1) * aBytes);
            for (int t = 0; t < ntap - D; t++) kstep();
            // the staged tile moves on to the next band (on the last band: back to this band's first tile, never read -- or, in the
            // persistent form, on to the first tile of the workgroup's next tile)
            if (macro + 1 < nMacro) wMacro += (kc + 1 == nchunk) ? macroJump : (uint64_t)kRowBytes;
            else if constexpr (PERSIST) { if (haveNext) wMacro = wb + (uint64_t)co0Next * rowStride; }
            wNext = wMacro;
            for (int t = 0; t < D; t++) kstep();
            if (++kc == nchunk) { kc = 0; dt++; tbit++; }
        }
    }
    else
    {
        // band waves: the pieces of the next band, spread over the taps of this one; waited for at its last tap
        for (int step = 0; step < nSteps; step++)
        {
            const int mkc = (kc + 1 == nchunk) ? 0 : kc + 1;
            const int mdt = (kc + 1 == nchunk) ? dt + 1 : dt;
            if (macro + 1 < nMacro && !(kAbl & (1 | 64))) issueA(mdt, mkc, tap, (gbase + macro + 1) & 1, aPerStep, aWaves, aWave0);
            else if constexpr (PERSIST)
            {
                // last band of the tile: the first band of the next tile goes into the other buffer
                if (haveNext && macro + 1 == nMacro && !(kAbl & (1 | 64))) { aM0 = m0Next; issueA(0, 0, tap, (gbase + macro + 1) & 1, aPerStep, aWaves, aWave0); aM0 = m0; }
            }
            compute();
            if constexpr (!(kAbl & 16))
            {
                if (tap == ntap - 1) wait_vm_const<0>();
                __syncthreads();
            }
            advance();
        }
    }

    // ---- epilogue: registers -> LDS (wave-private rows of this wave's 64 / 32 output channels) -> 16-byte channels-last stores -------
    // A result lane holds ONE pixel and 4 output channels per register quad: stored directly, a wave instruction writes 16 bytes to
    // each of 32 cache lines (measured: 36 % of the kernel time on the 128-channel layers, ablation bit 256). Through LDS a wave
    // instruction writes whole 128-byte (64-byte for 32-channel wave tiles) pixel rows instead.
    // Staging rows have a pitch of row bytes + 16; the two 8-byte halves of a 16-byte chunk are exchanged in rows 16 .. 31 of a pixel
    // block, which makes the 8-byte writes of the 32 lanes of a half wave hit 64 distinct banks (exchanged back, statically, on read).
    const T* bias = static_cast<const T*>(p.b);
    const T* res  = static_cast<const T*>(p.res);
    T* out  = static_cast<T*>(p.out);
    T* ysum = static_cast<T*>(p.ysum);
    const uint32_t hw = (uint32_t)(p.H * p.W);
    constexpr int WCO = NCB * 32;                 // output channels of a wave tile
    constexpr int RB = WCO * 2;                   // bytes of a staged pixel row
    constexpr int PITCH = RB + 16;
    constexpr int ROWS = PB * 32;                 // pixels of a wave tile
    constexpr int CPR = RB / 16;                  // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;                 // rows per wave instruction on the way out
    constexpr int NI = ROWS / RPI;
    constexpr bool kLdsStore = !(kAbl & 512) && !OUTF;
    // (persistent form: in the band buffer this tile has finished with -- the front of the LDS holds the zero page, the next tile's first
    // weight tiles and possibly its band; the host only picks this form when NW x ROWS x PITCH fits a band)
    unsigned char* const stage = smem + (PERSIST ? aOff + ((gbase + nMacro - 1) & 1) * aBytes : 0) + wave * (ROWS * PITCH);
    wait_vm_const<0>();                           // weight waves leave the K loop with re-staged tiles still in flight towards LDS: they must
                                                  // have landed before the staging below reuses that LDS -- and, on the float32-output path that
                                                  // stages nothing, before this wave can end (a workgroup that ends with LDS-DMA in flight lets the
                                                  // late pieces land in the LDS of the NEXT workgroup on the CU: intermittent wrong tiles, measured
                                                  // as a 1e-2 gradient error in one run out of two of the float32 model test)
    if constexpr (kLdsStore) __syncthreads();
    if constexpr (PLANES)
    {
        // acc[cb][pb][4 qd + e]: channel co0 + wc * WCO + cb * 32 + l31, pixel pb * 32 + 8 qd + 4 hi + e of this wave's 64 (tile rows 4 wr .. 4 wr + 3, 16 wide)
        constexpr int PP = ROWS * 2 + 8;              // staged channel row: 64 pixels + 8 bytes (the 32 lanes of a half wave write 8 bytes each to 64 distinct banks)
        constexpr int NSEG = WCO / 8;                 // 8-pixel segments per lane: lane task t = i * 64 + lane -> channel row t >> 3, segment t & 7
        unsigned char* const st = smem + wave * (WCO * PP);
        auto seg_of = [&](int i, int& cl, int& oy, int& ox) __attribute__((always_inline))
        {
            const int t = i * 64 + lane, seg = t & 7;
            cl = t >> 3;
            oy = y0 + wr * (ROWS / kTileW) + (seg >> 1);
            ox = x0 + (seg & 1) * 8;
        };
        // The reduction partner's tile (this wave's channels x 64 pixels of the NCHW tensors) is fetched the way the result is stored -- 16 bytes per lane, 32-byte
        // runs -- at the very start, so that the loads fly during the output path; it goes through the staging area afterwards. (First version: every lane read its
        // own channel's pixels as dwords straight from global memory -- 64 cache lines per load instruction: +127 us on a 375 us data gradient.)
        uint4 partner[NSEG];
        if (p.dotPartial)
        {
            #pragma unroll
            for (int i = 0; i < NSEG; i++)
            {
                int cl, oy, ox;
                seg_of(i, cl, oy, ox);
                const int c = co0 + wc * WCO + cl;
                const T* plane = nullptr;
                if (c < p.cDotA) plane = static_cast<const T*>(p.dotA) + ((int64_t)n2 * p.cDotA + c) * p.H * p.W;
                else if (c < p.cDotA + p.cDotB) plane = static_cast<const T*>(p.dotB) + ((int64_t)n2 * p.cDotB + (c - p.cDotA)) * p.H * p.W;
                uint4 v = make_uint4(0, 0, 0, 0);
                const int left = p.W - ox;
                if (plane && oy < p.H && left >= 2)
                {
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(plane + (int64_t)oy * p.W + ox);      // (even width: dword-aligned)
                    if (left >= 8) { const uint2 a = *reinterpret_cast<const uint2*>(src), b = *reinterpret_cast<const uint2*>(src + 2); v = make_uint4(a.x, a.y, b.x, b.y); }
                    else
                    {
                        v.x = src[0];
                        if (left >= 4) v.y = src[1];
                        if (left >= 6) v.z = src[2];
                    }
                }
                partner[i] = v;
            }
        }
        #pragma unroll
        for (int cb = 0; cb < NCB; cb++)
        {
            const int c = co0 + wc * WCO + cb * 32 + l31;
            const float scale = (p.pre && c < p.coOut) ? p.pre[(int64_t)n2 * p.coOut + c] : 1.f;
            #pragma unroll
            for (int pb = 0; pb < PB; pb++)
                #pragma unroll
                for (int qd = 0; qd < 4; qd++)
                {
                    T o4[4];
                    #pragma unroll
                    for (int e = 0; e < 4; e++) o4[e] = from_acc<T>(acc[cb][pb][qd * 4 + e] * scale);
                    uint2 ov;
                    __builtin_memcpy(&ov, o4, 8);
                    *reinterpret_cast<uint2*>(st + (cb * 32 + l31) * PP + (pb * 32 + 8 * qd + 4 * hi) * 2) = ov;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T* const outp = static_cast<T*>(p.out);
        #pragma unroll
        for (int i = 0; i < NSEG; i++)
        {
            int cl, oy, ox;
            seg_of(i, cl, oy, ox);
            const uint2 lo = *reinterpret_cast<const uint2*>(st + cl * PP + ((i * 64 + lane) & 7) * 16);
            const uint2 hi2 = *reinterpret_cast<const uint2*>(st + cl * PP + ((i * 64 + lane) & 7) * 16 + 8);
            const int c = co0 + wc * WCO + cl;
            if (c < p.coOut && oy < p.H)
            {
                T* const dst = outp + (((int64_t)n2 * p.coOut + c) * p.H + oy) * p.W + ox;
                // W is even (host check): 8, 6, 4 or 2 pixels of the segment are inside the plane; rows are 4-byte aligned (dword-aligned multi-dword stores)
                const int left = p.W - ox;
                if (left >= 8) { *reinterpret_cast<uint2*>(dst) = lo; *reinterpret_cast<uint2*>(dst + 4) = hi2; }
                else if (left >= 6) { *reinterpret_cast<uint2*>(dst) = lo; *reinterpret_cast<uint32_t*>(dst + 4) = hi2.x; }
                else if (left >= 4) *reinterpret_cast<uint2*>(dst) = lo;
                else if (left >= 2) *reinterpret_cast<uint32_t*>(dst) = lo.x;
            }
        }
        if (p.dotPartial)
        {
            // partner tile -> the staging area (the result rows have been read), then each lane's channel row against its accumulators
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            #pragma unroll
            for (int i = 0; i < NSEG; i++)
            {
                const int t = i * 64 + lane;
                *reinterpret_cast<uint2*>(st + (t >> 3) * PP + (t & 7) * 16) = make_uint2(partner[i].x, partner[i].y);
                *reinterpret_cast<uint2*>(st + (t >> 3) * PP + (t & 7) * 16 + 8) = make_uint2(partner[i].z, partner[i].w);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            #pragma unroll
            for (int cb = 0; cb < NCB; cb++)
            {
                const int c = co0 + wc * WCO + cb * 32 + l31;
                float dsum = 0.f;
                #pragma unroll
                for (int pb = 0; pb < PB; pb++)
                    #pragma unroll
                    for (int qd = 0; qd < 4; qd++)
                    {
                        const uint2 raw = *reinterpret_cast<const uint2*>(st + (cb * 32 + l31) * PP + (pb * 32 + 8 * qd + 4 * hi) * 2);
                        T four[4];
                        __builtin_memcpy(four, &raw, 8);
                        #pragma unroll
                        for (int e = 0; e < 4; e++) dsum = fmaf(acc[cb][pb][qd * 4 + e], (float)to_acc(four[e]), dsum);      // (pixels outside the plane: partner 0)
                    }
                dsum += __shfl_xor(dsum, 32, 64);                  // the two half waves hold the same channels, different pixels
                if (hi == 0 && c < p.cDotA + p.cDotB)
                    p.dotPartial[((int64_t)mt * (BM / ROWS) + wr) * (p.cDotA + p.cDotB) + c] = dsum;
            }
        }
        return;
    }
    const int flipW = (l31 >> 4) & 1;
    // 16-byte stores of the staged wave tile to `dst` (rows = pixels m0 + wr * ROWS + ..., this wave's channel range)
    auto flush = [&](T* dst) __attribute__((always_inline))
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        #pragma unroll
        for (int i = 0; i < NI; i++)
        {
            const int R = i * RPI + lane / CPR, c = lane % CPR;
            uint4 v = *reinterpret_cast<const uint4*>(stage + R * PITCH + c * 16);
            if (((i * RPI) >> 4) & 1) v = make_uint4(v.z, v.w, v.x, v.y);
            if constexpr (T2D)
            {
                const int oy = y0 + wr * (ROWS / kTileW) + R / kTileW, ox = x0 + (R & (kTileW - 1));
                const int64_t m = (int64_t)(n2 * p.H + oy) * p.W + ox;
                if (oy < p.H && ox < p.W) *reinterpret_cast<uint4*>(dst + m * p.oStride + (co0 + wc * WCO + c * 8)) = v;
            }
            else
            {
                const int64_t m = m0 + wr * ROWS + R;
                if (m < p.M) *reinterpret_cast<uint4*>(dst + m * p.Co + (co0 + wc * WCO + c * 8)) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    float sq = 0.f;
    // The arithmetic of the epilogue, in three forms. FAST (16-bit output, bias present, `pre` and `post` both present -- the generator's
    // modulated convolutions -- or both absent -- the discriminator's layers --, 0 < negative slope <= 1): no per-quad pointer tests (they cost ~165 register copies and ~150 selects per wave tile as
    // merges of uniform branches), leaky ReLU as max(u, u * slope), two-element float vectors throughout (v_pk_fma / v_pk_mul), clamp and
    // residual as compile-time variants. On the 64- and 128-channel layers a wave tile has 72 .. 288 MFMAs but 2 048 .. 4 096 outputs, and
    // the epilogue was the larger part of the tile's issue cycles (849 vector instructions per 32 outputs of a lane before, r04).
    // PLAIN (no operand but the accumulators, linear: the data gradients): a conversion. The generic form keeps every other combination
    // and the float32-output kernels.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto fast_math = [&](auto clampTag, auto resTag, auto scaleTag) __attribute__((always_inline))
    {
        constexpr bool CLAMP = decltype(clampTag)::value, RES = decltype(resTag)::value, SCALE = decltype(scaleTag)::value;
        const float slope = p.slopeNeg, gainv = p.gain, cl = p.clamp;
        #pragma unroll
        for (int pb = 0; pb < PB; pb++)
        {
            int64_t mt_ = m0 + jrow[pb];
            bool valid = mt_ < p.M;
            if constexpr (T2D)
            {
                const int j = wr * (32 * PB) + pb * 32 + l31;
                const int oy = y0 + j / kTileW, ox = x0 + txl;
                valid = oy < p.H && ox < p.W;
                mt_ = (int64_t)(n2 * p.H + oy) * p.W + ox;
            }
            const int64_t m = valid ? mt_ : (T2D ? (int64_t)n2 * p.H * p.W : p.M - 1);   // pixels past the end: computed on a valid pixel's terms, never stored
            const int64_t f = T2D ? (int64_t)n2 : (int64_t)((uint32_t)m / hw);
            const float* const preRow = SCALE ? p.pre + f * p.Co + (co0 + wc * (BN / 2) + 4 * hi) : nullptr;
            const float* const postRow = SCALE ? p.post + f * p.Co + (co0 + wc * (BN / 2) + 4 * hi) : nullptr;
            const T* const biasRow = bias + (co0 + wc * (BN / 2) + 4 * hi);
            const T* const resRow = RES ? res + m * p.Co + (co0 + wc * (BN / 2) + 4 * hi) : nullptr;
            f32x2 sq2 = {0.f, 0.f};
            #pragma unroll
            for (int cb = 0; cb < NCB; cb++)
                #pragma unroll
                for (int qd = 0; qd < 4; qd++)
                {
                    const int cOff = cb * 32 + 8 * qd;
                    float4 pv = make_float4(1.f, 1.f, 1.f, 1.f), qv = pv;
                    if constexpr (SCALE)
                    {
                        pv = *reinterpret_cast<const float4*>(preRow + cOff);
                        qv = *reinterpret_cast<const float4*>(postRow + cOff);
                    }
                    const uint2 braw = *reinterpret_cast<const uint2*>(biasRow + cOff);
                    T b4[4];
                    __builtin_memcpy(b4, &braw, 8);
                    f32x2 add2[2] = {{to_acc(b4[0]), to_acc(b4[1])}, {to_acc(b4[2]), to_acc(b4[3])}};
                    if constexpr (RES)
                    {
                        const uint2 rraw = *reinterpret_cast<const uint2*>(resRow + cOff);
                        T r4[4];
                        __builtin_memcpy(r4, &rraw, 8);
                        add2[0] += f32x2{to_acc(r4[0]), to_acc(r4[1])};
                        add2[1] += f32x2{to_acc(r4[2]), to_acc(r4[3])};
                    }
                    const f32x2 pre2[2] = {{pv.x, pv.y}, {pv.z, pv.w}}, post2[2] = {{qv.x, qv.y}, {qv.z, qv.w}};
                    T o4[4];
                    #pragma unroll
                    for (int h2 = 0; h2 < 2; h2++)
                    {
                        const f32x2 a = {acc[cb][pb][qd * 4 + 2 * h2], acc[cb][pb][qd * 4 + 2 * h2 + 1]};
                        const f32x2 u = SCALE ? a * pre2[h2] + add2[h2] : a + add2[h2];
                        const f32x2 nn = u * slope;
                        f32x2 g = {fmaxf(u.x, nn.x), fmaxf(u.y, nn.y)};           // u > 0 ? u : u * slope for 0 < slope <= 1 (NaN stays NaN)
                        g = g * gainv;
                        if constexpr (CLAMP)
                        {
                            // one v_med3_f32 per value instead of two compares and two selects; a NaN comes out as -clamp, as in the reference's
                            // kernel (bias_act.cu:139: `(y > -clamp & y < clamp) ? y : (y >= 0) ? clamp : -clamp`)
                            g.x = __builtin_amdgcn_fmed3f(g.x, -cl, cl);
                            g.y = __builtin_amdgcn_fmed3f(g.y, -cl, cl);
                        }
                        sq2 = g * g + sq2;
                        const f32x2 o = SCALE ? g * post2[h2] : g;
                        o4[2 * h2] = from_acc<T>(o.x);
                        o4[2 * h2 + 1] = from_acc<T>(o.y);
                    }
                    uint2 ov;
                    __builtin_memcpy(&ov, o4, 8);
                    if constexpr (kLdsStore)
                        *reinterpret_cast<uint2*>(stage + (pb * 32 + l31) * PITCH + cb * 64 + qd * 16 + ((hi ^ flipW) << 3)) = ov;
                    else if (valid && (!(kAbl & 256) || sq == 12345.f))
                    {
                        T y4[4];
                        #pragma unroll
                        for (int e = 0; e < 4; e++) y4[e] = from_acc<T>(acc[cb][pb][qd * 4 + e]);
                        uint2 yv;
                        __builtin_memcpy(&yv, y4, 8);
                        const int co = co0 + wc * (BN / 2) + cb * 32 + 8 * qd + 4 * hi;
                        *reinterpret_cast<uint2*>(out + m * p.Co + co) = ov;
                        if (ysum) *reinterpret_cast<uint2*>(ysum + m * p.Co + co) = yv;
                    }
                }
            if (valid) sq += sq2.x + sq2.y;
        }
    };
    auto generic_math = [&]() __attribute__((always_inline))
    {
        #pragma unroll
        for (int pb = 0; pb < PB; pb++)
        {
            int64_t mt_ = m0 + jrow[pb];
            bool valid = mt_ < p.M;
            if constexpr (T2D)
            {
                const int j = wr * (32 * PB) + pb * 32 + l31;
                const int oy = y0 + j / kTileW, ox = x0 + txl;
                valid = oy < p.H && ox < p.W;
                mt_ = (int64_t)(n2 * p.H + oy) * p.W + ox;
            }
            if constexpr (!kLdsStore) { if (!valid) continue; }
            const int64_t m = valid ? mt_ : (T2D ? (int64_t)n2 * p.H * p.W : p.M - 1);   // pixels past the end: computed on a valid pixel's terms, never stored
            const int64_t f = T2D ? (int64_t)n2 : (int64_t)((uint32_t)m / hw);
            #pragma unroll
            for (int cb = 0; cb < NCB; cb++)
                #pragma unroll
                for (int qd = 0; qd < 4; qd++)
                {
                    const int co = co0 + wc * (BN / 2) + cb * 32 + 8 * qd + 4 * hi;
                    float pre4[4] = {1.f, 1.f, 1.f, 1.f}, post4[4] = {1.f, 1.f, 1.f, 1.f}, add4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.pre)  { const float4 v = *reinterpret_cast<const float4*>(p.pre + f * p.Co + co);  pre4[0] = v.x; pre4[1] = v.y; pre4[2] = v.z; pre4[3] = v.w; }
                    if (p.post) { const float4 v = *reinterpret_cast<const float4*>(p.post + f * p.Co + co); post4[0] = v.x; post4[1] = v.y; post4[2] = v.z; post4[3] = v.w; }
                    if constexpr (OUTF)
                    {
                        // float32 output: bias and residual are float32 tensors too
                        if (bias) { const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.b) + co); add4[0] += v.x; add4[1] += v.y; add4[2] += v.z; add4[3] += v.w; }
                        if (res)  { const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + m * p.Co + co); add4[0] += v.x; add4[1] += v.y; add4[2] += v.z; add4[3] += v.w; }
                    }
                    else
                    {
                    if (bias)
                    {
                        uint2 raw = *reinterpret_cast<const uint2*>(bias + co);
                        T t4[4];
                        __builtin_memcpy(t4, &raw, 8);
                        #pragma unroll
                        for (int e = 0; e < 4; e++) add4[e] += to_acc(t4[e]);
                    }
                    if (res)
                    {
                        uint2 raw = *reinterpret_cast<const uint2*>(res + m * p.Co + co);
                        T t4[4];
                        __builtin_memcpy(t4, &raw, 8);
                        #pragma unroll
                        for (int e = 0; e < 4; e++) add4[e] += to_acc(t4[e]);
                    }
                    }
                    T o4[4], y4[4];
                    float sqq = 0.f;
                    #pragma unroll
                    for (int e = 0; e < 4; e++)
                    {
                        const float a = acc[cb][pb][qd * 4 + e];
                        const float u = fmaf(a, pre4[e], add4[e]);
                        float g = (u > 0.f ? u : u * p.slopeNeg) * p.gain;
                        if (p.clamp >= 0.f) g = g > p.clamp ? p.clamp : (g < -p.clamp ? -p.clamp : g);
                        if constexpr (kLdsStore) sqq = fmaf(g, g, sqq); else sq = fmaf(g, g, sq);
                        o4[e] = from_acc<T>(g * post4[e]);
                        y4[e] = from_acc<T>(a);
                    }
                    if (kLdsStore && valid) sq += sqq;
                    uint2 ov, yv;
                    __builtin_memcpy(&ov, o4, 8);
                    __builtin_memcpy(&yv, y4, 8);
                    if constexpr (OUTF)
                    {
                        float4 fv;
                        {
                            float o[4];
                            #pragma unroll
                            for (int e = 0; e < 4; e++)
                            {
                                const float a = acc[cb][pb][qd * 4 + e];
                                const float u = fmaf(a, pre4[e], add4[e]);
                                float g = (u > 0.f ? u : u * p.slopeNeg) * p.gain;
                                if (p.clamp >= 0.f) g = g > p.clamp ? p.clamp : (g < -p.clamp ? -p.clamp : g);
                                o[e] = g * post4[e];                 // (the magnitude statistic was accumulated by the common code above)
                            }
                            fv = make_float4(o[0], o[1], o[2], o[3]);
                        }
                        const int64_t ostr = T2D ? (int64_t)ConvArgs2DStride(p) : (int64_t)p.Co;
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + m * ostr + co) = fv;
                        if (p.ysum)
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.ysum) + m * ostr + co) =
                                make_float4(acc[cb][pb][qd * 4], acc[cb][pb][qd * 4 + 1], acc[cb][pb][qd * 4 + 2], acc[cb][pb][qd * 4 + 3]);
                    }
                    else if constexpr (kLdsStore)
                        *reinterpret_cast<uint2*>(stage + (pb * 32 + l31) * PITCH + cb * 64 + qd * 16 + ((hi ^ flipW) << 3)) = ov;
                    else if (!(kAbl & 256) || sq == 12345.f)                 // (ablation 256: no output stores)
                    {
                        *reinterpret_cast<uint2*>(out + m * p.Co + co) = ov;
                        if (ysum) *reinterpret_cast<uint2*>(ysum + m * p.Co + co) = yv;
                    }
                }
        }
    };
    auto plain_math = [&]() __attribute__((always_inline))
    {
        const float gainv = p.gain;
        #pragma unroll
        for (int pb = 0; pb < PB; pb++)
            #pragma unroll
            for (int cb = 0; cb < NCB; cb++)
                #pragma unroll
                for (int qd = 0; qd < 4; qd++)
                {
                    T o4[4];
                    #pragma unroll
                    for (int e = 0; e < 4; e++) o4[e] = from_acc<T>(acc[cb][pb][qd * 4 + e] * gainv);
                    uint2 ov;
                    __builtin_memcpy(&ov, o4, 8);
                    *reinterpret_cast<uint2*>(stage + (pb * 32 + l31) * PITCH + cb * 64 + qd * 16 + ((hi ^ flipW) << 3)) = ov;
                }
    };
    int form = 0;                                                     // 0 generic, 1 fast with scales, 2 fast without, 3 plain
    if constexpr (!OUTF && !T2D && kLdsStore)
    {
        const bool lrelu = p.slopeNeg > 0.f && p.slopeNeg <= 1.f;
        if (bias && lrelu && p.pre && p.post) form = 1;
        else if (bias && lrelu && !p.pre && !p.post) form = 2;
        else if (!bias && !res && !p.pre && !p.post && p.slopeNeg == 1.f && p.clamp < 0.f && !p.msqPartial) form = 3;
    }
    auto pick = [&](auto scaleTag) __attribute__((always_inline))
    {
        if (p.clamp >= 0.f) { if (res) fast_math(std::true_type{}, std::true_type{}, scaleTag); else fast_math(std::true_type{}, std::false_type{}, scaleTag); }
        else                { if (res) fast_math(std::false_type{}, std::true_type{}, scaleTag); else fast_math(std::false_type{}, std::false_type{}, scaleTag); }
    };
    if (form == 1) pick(std::true_type{});
    else if (form == 2) pick(std::false_type{});
    else if (form == 3) plain_math();
    else generic_math();
    if constexpr (kLdsStore)
    {
        if (!(kAbl & 256) || sq == 12345.f)
        {
            flush(out);
            if (ysum)
            {
                #pragma unroll
                for (int pb = 0; pb < PB; pb++)
                    #pragma unroll
                    for (int cb = 0; cb < NCB; cb++)
                        #pragma unroll
                        for (int qd = 0; qd < 4; qd++)
                        {
                            T y4[4];
                            #pragma unroll
                            for (int e = 0; e < 4; e++) y4[e] = from_acc<T>(acc[cb][pb][qd * 4 + e]);
                            uint2 yv;
                            __builtin_memcpy(&yv, y4, 8);
                            *reinterpret_cast<uint2*>(stage + (pb * 32 + l31) * PITCH + cb * 64 + qd * 16 + ((hi ^ flipW) << 3)) = yv;
                        }
                flush(ysum);
            }
        }
        __syncthreads();                          // the statistic below re-uses the front of the LDS
    }
    if (p.msqPartial)
    {
        sq = wave_sum(sq);
        float* red = reinterpret_cast<float*>(smem) + 64;   // the K loop ended with a barrier; LDS past the zero rows in use
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0)
        {
            float tot = 0.f;
            for (int i = 0; i < NW; i++) tot += red[i];
            p.msqPartial[tile] = tot;
        }
    }
    if constexpr (PERSIST)
    {
        if (haveNext)
        {
            // (the barriers above -- after the staged stores, inside the statistic -- already separate this tile's LDS reads from the next tile's staging)
            gbase += nMacro;
            firstTile = false;
            tile = tileNext; mt = mtNext; nt = tile - mt * p.nTiles;
            m0 = m0Next; co0 = co0Next; aM0 = m0;
            goto next_tile;
        }
    }
}

} // namespace
