// lvg_common.h -- shared device/host helpers for liblvg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/lvg_ops.h"

// ---------------------------------------------------------------------------------------------
// Element types. Storage types are raw 16/32/64-bit words; arithmetic is done in `acc_t`
// (float for f32/f16/bf16, double for f64) exactly like the reference kernels' InternalType.

struct f16_t  { uint16_t bits; };
struct bf16_t { uint16_t bits; };

template <class T> struct Elem;
template <> struct Elem<float>  { typedef float  acc_t; static constexpr int kVec = 4; };
template <> struct Elem<double> { typedef double acc_t; static constexpr int kVec = 2; };
template <> struct Elem<f16_t>  { typedef float  acc_t; static constexpr int kVec = 8; };
template <> struct Elem<bf16_t> { typedef float  acc_t; static constexpr int kVec = 8; };

__device__ __forceinline__ float  to_acc(float v)  { return v; }
__device__ __forceinline__ double to_acc(double v) { return v; }
__device__ __forceinline__ float  to_acc(f16_t v)  { _Float16 h; __builtin_memcpy(&h, &v.bits, 2); return (float)h; }
__device__ __forceinline__ float  to_acc(bf16_t v) { return __uint_as_float(((uint32_t)v.bits) << 16); }

template <class T> __device__ __forceinline__ T from_acc(typename Elem<T>::acc_t v);
template <> __device__ __forceinline__ float  from_acc<float>(float v)   { return v; }
template <> __device__ __forceinline__ double from_acc<double>(double v) { return v; }
template <> __device__ __forceinline__ f16_t  from_acc<f16_t>(float v)   { _Float16 h = (_Float16)v; f16_t r; __builtin_memcpy(&r.bits, &h, 2); return r; }
template <> __device__ __forceinline__ bf16_t from_acc<bf16_t>(float v)
{
    // gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN -- the same
    // rounding as torch's float -> bfloat16); the integer emulation cost ~6 VALU ops per element, which
    // showed up in the bf16 streaming kernels.
    __bf16 h = (__bf16)v;
    bf16_t r;
    __builtin_memcpy(&r.bits, &h, 2);
    return r;
}

// 16-byte vector of T (kVec elements).
template <class T> struct alignas(16) Vec16 { T v[Elem<T>::kVec]; };

template <class T> __device__ __forceinline__ Vec16<T> load_vec16(const T* p)
{
    Vec16<T> r;
    *reinterpret_cast<uint4*>(&r) = *reinterpret_cast<const uint4*>(p);
    return r;
}
template <class T> __device__ __forceinline__ void store_vec16(T* p, const Vec16<T>& r)
{
    *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(&r);
}

// ---------------------------------------------------------------------------------------------
// Host-side error plumbing.

void lvg_set_error(const char* fmt, ...);

#define LVG_REQUIRE(cond, ...) do { if (!(cond)) { lvg_set_error(__VA_ARGS__); return LVG_ERR_INVALID; } } while (0)

static inline int lvg_check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { lvg_set_error("%s: %s", what, hipGetErrorString(e)); return LVG_ERR_LAUNCH; }
    return LVG_OK;
}

static inline bool lvg_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int64_t lvg_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// floor division for possibly negative numerators (b > 0)
__host__ __device__ __forceinline__ int lvg_floor_div(int a, int b)
{
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}
