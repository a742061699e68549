// flrelu_mfma_common.h -- pieces shared by the banded-matrix filtered_lrelu kernels (filtered_lrelu_wave.hip, filtered_lrelu_band.hip):
// band fragments of the FIR matrices as MFMA operands, the LDS transpose read, the packed-f16 activation with its 2-bit mask.
// Semantics: reference torch_utils/ops/filtered_lrelu.cu:139-1099. Include inside an anonymous namespace after defining LVG_WABL.
#pragma once

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kU = 128;           // up-sampled columns of a tile (4 blocks of 32)
constexpr int wdiv_up(int a, int b) { return (a + b - 1) / b; }

// Offset (in input samples, relative to the first input sample of a 32-output block) of up-stage class `cls`.
template <int UP> __host__ __device__ constexpr int up_class_offset(int cls) { return UP == 2 ? 16 * cls : (UP == 4 ? 8 * cls - 8 : 0); }

// 16-chunks of the input that output block b of an up stage needs: first chunk, count, class of the first, class step.
template <int UP> struct UpChunks
{
    __host__ __device__ static constexpr int first(int b)  { return UP == 2 ? b : (UP == 4 ? ((b & 1) ? (b - 1) / 2 : b / 2) : 2 * b); }
    __host__ __device__ static constexpr int count(int b)  { return UP == 2 ? 2 : (UP == 4 ? ((b & 1) ? 2 : 1) : 2); }
    __host__ __device__ static constexpr int cls0(int b)   { return UP == 2 ? 0 : (UP == 4 ? ((b & 1) ? 0 : 1) : 0); }
    __host__ __device__ static constexpr int step()        { return UP == 4 ? 2 : 1; }
};

// One element of a band fragment (a 32 x 16 slice of a banded filter matrix as an MFMA operand: lane = row (& 31) and half of
// k (>> 5), j = the lane's j-th k). kind 0: A_y (up, natural k order 8 gg + j), 1: A_x (up, k permuted like an MFMA result's
// registers, scaled), 2: D_x (down, permuted, READ-mode column shift), 3: D_y (down, natural).
template <int UP, int DOWN, int FU, int FD>
__device__ __forceinline__ float frag_elem(const float* taps, int kind, int cls, int lane, int j, int phX, int phY, int rOff, float scale, int colShift = 0)
{
    constexpr int KU = FU / UP;
    const int row = lane & 31, gg = lane >> 5;
    const bool perm = kind == 1 || kind == 2;
    const int k = perm ? ((j & 3) + 8 * (j >> 2) + 4 * gg) : (8 * gg + j);
    if (kind < 2)
    {
        const bool isX = kind == 1;
        const int ph = isX ? phX : phY;
        const int kk = up_class_offset<UP>(cls) + k;
        const int m = row + ph, i0 = m / UP, t = kk - i0 - (isX ? colShift : 0);   // colShift: the operand's columns start that many samples early (band kernel)
        return (t >= 0 && t < KU) ? taps[(UP - 1 - m % UP) + t * UP] * (isX ? scale : 1.0f) : 0.0f;
    }
    const bool isX = kind == 2;
    const int t = 16 * cls + k - (isX ? rOff : 0) - row * DOWN;
    return (t >= 0 && t < FD) ? taps[FU + t] : 0.0f;
}

// MFMA operand (lane: index = lane & 31 along the COLUMNS of a row-major LDS matrix, k = 8 * (lane >> 5) + j along
// its ROWS) through the gfx950 transpose read (semantics measured by tools/probe_mfma_layout.hip).
__device__ __forceinline__ half8 lds_tr_operand(const _Float16* base, int stride, int row0, int col0, int lane)
{
    const int g = lane >> 5, hgrp = (lane >> 4) & 1, s = lane & 15;
    const _Float16* p = base + (row0 + 8 * g + (s >> 2)) * stride + col0 + 16 * hgrp + 4 * (s & 3);
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * stride));
    half8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &hi, 8);
    return r;
}

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c)
{
#if LVG_WABL & 32
    c[0] += (float)a[0] + (float)b[0]; return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
    #pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.0f;
    return z;
}

// Rows 16 * h .. 16 * h + 15 of a 32x32 result as the B operand of the next MFMA (k order: see frag_elem).
__device__ __forceinline__ half8 pack_chunk(const f32x16& c, int h)
{
    half8 r;
    #pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (_Float16)c[8 * h + j];
    return r;
}

__device__ __forceinline__ uint32_t h2_bits(half2v v) { uint32_t u; __builtin_memcpy(&u, &v, 4); return u; }
__device__ __forceinline__ half2v bits_h2(uint32_t u) { half2v v; __builtin_memcpy(&v, &u, 4); return v; }
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// One 16-bit element at a wave-uniform address through the scalar cache (the aligned dword that holds it).
__device__ __forceinline__ uint32_t scalar_load_u16(const uint16_t* ptr)
{
    const uint64_t a = (uint64_t)(uintptr_t)ptr;
    const uint64_t a4 = a & ~(uint64_t)3;
    uint32_t wd;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wd) : "s"(a4) : "memory");
    return (a & 2) ? (wd >> 16) : (wd & 0xffffu);
}

// Two stored elements (element 0 in the low half of the dword) + bias -> f16 pair. bfloat16 values beyond the
// f16 range saturate instead of turning into inf (inf * a zero tap of the banded matrix would be NaN).
template <class T> __device__ __forceinline__ half2v pair_plus_bias(uint32_t raw, half2v bias2, float bias);
template <> __device__ __forceinline__ half2v pair_plus_bias<f16_t>(uint32_t raw, half2v bias2, float) { return bits_h2(raw) + bias2; }
template <> __device__ __forceinline__ half2v pair_plus_bias<bf16_t>(uint32_t raw, half2v, float bias)
{
    const float a = __uint_as_float(raw << 16) + bias, b = __uint_as_float(raw & 0xffff0000u) + bias;
    half2v r;
    r[0] = (_Float16)__builtin_fminf(__builtin_fmaxf(a, -65504.0f), 65504.0f);
    r[1] = (_Float16)__builtin_fminf(__builtin_fmaxf(b, -65504.0f), 65504.0f);
    return r;
}

// Two f32 -> one dword of T (element 0 in the low half): one v_cvt_pk_* (written as a vector conversion; two scalar conversions
// were paired across the wrong elements and glued back with four more instructions).
typedef float float2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf162v __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ uint32_t pack_pair(float a, float b);
template <> __device__ __forceinline__ uint32_t pack_pair<f16_t>(float a, float b)
{
    const float2v f = {a, b};
    const half2v h = __builtin_convertvector(f, half2v);
    uint32_t u; __builtin_memcpy(&u, &h, 4); return u;
}
template <> __device__ __forceinline__ uint32_t pack_pair<bf16_t>(float a, float b)
{
    const float2v f = {a, b};
    const bf162v h = __builtin_convertvector(f, bf162v);
    uint32_t u; __builtin_memcpy(&u, &h, 4); return u;
}

// idx / D for the small indices of the vector maps (exact for idx < 65536 / D ... checked for the ranges used: tools note in DESIGN)
template <int D> __device__ __forceinline__ int div_small(int idx) { return (int)(((uint32_t)idx * (uint32_t)((65536 + D - 1) / D)) >> 16); }

struct ActConst
{
    half2v slope2, clampP, clampN;
    uint32_t clampBits;
    uint32_t shEven, shOdd;       // READ: bit offset of this lane's mask byte inside its dword for an even / odd q
    uint32_t lutBase;             // READ: LDS byte address of the factor table
};

// Activation of one 32 x 32 block of U^T held as an MFMA result (register r = pixel u = (r & 3) + 8 (r >> 2) + 4 g of this
// lane's row v), in packed f16. Registers 4q .. 4q + 3 are the four pixels of one mask byte. Result: the block as 8 packed
// dwords = the two B-operand chunks of the next MFMA.
//   WRITE: mdw = this lane's four mask bytes (q = 0..3 in bytes 0..3): codes 1 = negative, 2 = clamped, 2 bits per pixel.
//          The sign bits of four packed halves come out of ONE v_perm (selectors 8..11 replicate bit 15 / 31 of its sources),
//          v_and keeps bit 2k of byte k, v_sad_u8 adds the four bytes into one.
//   READ:  mlo / mhi = the dwords holding this lane's mask bytes of q = 0, 1 / q = 2, 3; factors (1, slope, 0) by code through
//          a v_perm look-up: the byte is replicated, shifted per half so that every byte of a dword holds the code of its
//          pixel in bits 0-1, and (code | 4 * byte parity) selects the low / high byte of the f16 factor.
// The file is compiled with -fno-honor-nans (no canonicalisation ops around min / max): a NaN pre-activation comes out as
// -clamp instead of NaN.
template <int MODE, bool SLOPEMAX, bool CLAMP, int LUTN>
__device__ __forceinline__ void act_block(const f32x16& accU, uint32_t (&zp)[8], uint32_t& mdw, uint32_t mlo, uint32_t mhi, const ActConst& k)
{
    uint32_t bA = 0, bB = 0;
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        half2v P[2];
        #pragma unroll
        for (int h = 0; h < 2; h++) { P[h][0] = (_Float16)accU[4 * q + 2 * h]; P[h][1] = (_Float16)accU[4 * q + 2 * h + 1]; }
        if (MODE == LVG_SIGNS_READ)
        {
            // the mask byte of these four pixels indexes a table of their four factors (1, slope, 0 by code): one LDS read replaces the
            // bit arithmetic -- the kernel is bound by the vector-instruction port, the LDS pipe has room
            uint32_t byte = __builtin_amdgcn_ubfe(q < 2 ? mlo : mhi, (q & 1) ? k.shOdd : k.shEven, 8u);
            if (LUTN < 256) byte = min(byte, (uint32_t)(LUTN - 1));
            typedef __attribute__((address_space(3))) const uint2v* lds_u2;
            const uint2v f = *(lds_u2)(uintptr_t)(k.lutBase + byte * 8u);
            zp[2 * q] = h2_bits(P[0] * bits_h2(f[0]));
            zp[2 * q + 1] = h2_bits(P[1] * bits_h2(f[1]));
        }
        else
        {
            half2v L[2];
            #pragma unroll
            for (int h = 0; h < 2; h++)
            {
                const half2v ls = P[h] * k.slope2;
                if (SLOPEMAX) L[h] = __builtin_elementwise_max(P[h], ls);    // 0 <= slope <= 1
                else
                {
                    // all ones in the halves that are negative (sign bit: -0.0 counts)
                    const uint32_t m = __builtin_amdgcn_perm(0u, h2_bits(P[h]), 0x09090808u);
                    L[h] = bits_h2((h2_bits(ls) & m) | (h2_bits(P[h]) & ~m));
                }
            }
            if (MODE == LVG_SIGNS_WRITE)
            {
                uint32_t x = __builtin_amdgcn_perm(h2_bits(P[1]), h2_bits(P[0]), 0x0B0A0908u) & 0x40100401u;   // byte k: bit 2k = pixel k negative
                if (CLAMP)
                {
                    uint32_t Tb[2];
                    #pragma unroll
                    for (int h = 0; h < 2; h++)
                    {
                        const uint32_t a = h2_bits(L[h]) & 0x7fff7fffu;       // "clamped" = sign bit of (clamp - |L|) as 16-bit integers
                        short2v cv, av; __builtin_memcpy(&cv, &k.clampBits, 4); __builtin_memcpy(&av, &a, 4);
                        const short2v d = cv - av;
                        __builtin_memcpy(&Tb[h], &d, 4);
                    }
                    const uint32_t C = __builtin_amdgcn_perm(Tb[1], Tb[0], 0x0B0A0908u);
                    x = (C & 0x80200802u) | (~C & x);                       // code 2 replaces the sign bit
                }
                if (q == 0) bA = __builtin_amdgcn_sad_u8(x, 0u, 0u);
                if (q == 1) bB = __builtin_amdgcn_sad_u8(x, 0u, 0u);
                if (q == 2) bA = __builtin_amdgcn_sad_hi_u8(x, 0u, bA);
                if (q == 3) bB = __builtin_amdgcn_sad_hi_u8(x, 0u, bB);
            }
            #pragma unroll
            for (int h = 0; h < 2; h++)
                zp[2 * q + h] = CLAMP ? h2_bits(__builtin_elementwise_min(__builtin_elementwise_max(L[h], k.clampN), k.clampP)) : h2_bits(L[h]);
        }
    }
    if (MODE == LVG_SIGNS_WRITE) mdw = bA | (bB << 8);
}
