// wgrad_common.h -- pieces shared by the weight-gradient kernels (conv3d_wgrad.hip: time-major frames of width 8 / 16 / 32 / 64;
// conv2d_wgrad.hip: 4 x 16 pixel patches of explicitly zero-padded frames of any width).
#pragma once
#include "lvg_common.h"
#include <algorithm>
#include <stdlib.h>
#include <string.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <class T> struct MmaW;
template <> struct MmaW<bf16_t>
{
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, const f32x16& c)
    {
        bf16x8 av, bv;
        __builtin_memcpy(&av, &a, 16);
        __builtin_memcpy(&bv, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
    }
};
template <> struct MmaW<f16_t>
{
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, const f32x16& c)
    {
        f16x8 av, bv;
        __builtin_memcpy(&av, &a, 16);
        __builtin_memcpy(&bv, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    }
};

#ifndef LVG_WGRAD_PRIO
#define LVG_WGRAD_PRIO 1
#endif
constexpr bool kWgradPrio = LVG_WGRAD_PRIO != 0;
constexpr int kRow = 128;                 // bytes per LDS row: 64 channels
constexpr int kDyBytes = 64 * kRow;       // dy tile: 64 pixels

__device__ __forceinline__ void wdma16(const unsigned char* base, uint32_t laneOff, uint32_t ldsAddr)
{
    // the base pointer and the LDS address are wave-uniform by construction; make that visible to the register allocator
    uint64_t b = (uint64_t)(uintptr_t)base;
    b = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    base = reinterpret_cast<const unsigned char*>((uintptr_t)b);
    ldsAddr = (uint32_t)__builtin_amdgcn_readfirstlane((int)ldsAddr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(laneOff), "s"(base), "s"(ldsAddr) : "memory");
}

// chunk swizzle of LDS row r (absolute row index inside a 256-byte aligned buffer)
__device__ __forceinline__ uint32_t swz(uint32_t r) { return ((r >> 1) & 1u) << 2; }

// Byte offset (inside its buffer) of the 8 bytes a lane hands to the transpose read for LDS row `row`: this lane's 4 source
// elements start at column `col` (channels). The chunk swizzle toggles bit 6 of the offset with bit 1 of the row.
__device__ __forceinline__ uint32_t tr_addr(uint32_t row, uint32_t col)
{
    return (row << 7) + ((col * 2u) ^ ((row & 2u) << 5));
}

typedef __attribute__((address_space(3))) short4v* lds_tr_ptr;

// Transposed MFMA operand from the lane address of its first row: rows +0..3 and +4..7 (same swizzle phase: +512 bytes).
__device__ __forceinline__ uint4 tr_read8(uint32_t addr)
{
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(uintptr_t)addr);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(uintptr_t)(addr + 4 * kRow));
    uint4 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &hi, 8);
    return r;
}

} // namespace
