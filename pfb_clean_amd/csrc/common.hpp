// common.hpp -- shared device/host helpers for libpfb_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdint>
#include <cmath>
#include "../../include/pfb_hip.h"

namespace pfb {

// ---------------------------------------------------------------- error plumbing
void set_error(const char* fmt, ...);

#define PFB_HIP_CHECK(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            ::pfb::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,          \
                             hipGetErrorString(_e));                               \
            return PFB_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define PFB_REQUIRE(cond, code, ...)                                               \
    do {                                                                           \
        if (!(cond)) {                                                             \
            ::pfb::set_error(__VA_ARGS__);                                         \
            return (code);                                                         \
        }                                                                          \
    } while (0)

// -------------------------------------------------------------------- complex
template <typename T> struct vec2;
template <> struct vec2<float>  { using type = float2;  };
template <> struct vec2<double> { using type = double2; };

template <typename T>
struct alignas(2 * sizeof(T)) cplx {
    T x, y;
    __host__ __device__ cplx() = default;
    __host__ __device__ cplx(T a, T b) : x(a), y(b) {}
};

// complex64 lives in an aligned VGPR pair and computes with the PACKED fp32 instructions of
// CDNA3/4 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two lanes of a 64-bit operand per
// issue, with per-lane source selection op_sel / op_sel_hi and per-lane negation neg_lo /
// neg_hi).  A complex add is ONE VALU instruction, a complex multiply TWO (instead of 2 / 4):
// the FFT kernels are VALU-issue bound (rocprofv3 SQ_INSTS_VALU x 4 cycles = 55-65 % of the
// kernel time before this), so instruction count is time.  clang does emit v_pk_* for
// ext_vector arithmetic but cannot fold the swizzles and half-negations of complex
// arithmetic into the modifiers (it adds v_xor / v_mov), hence the asm bodies below; its SLP
// vectoriser on scalar code was worse still (see the Makefile).
typedef float v2f __attribute__((ext_vector_type(2)));
template <>
struct alignas(8) cplx<float> {
    union {
        struct { float x, y; };
        v2f v;
    };
    __host__ __device__ cplx() = default;
    __host__ __device__ cplx(float a, float b) : x(a), y(b) {}
    __host__ __device__ explicit cplx(v2f q) : v(q) {}
};

template <typename T> __device__ __forceinline__ cplx<T> operator+(cplx<T> a, cplx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ cplx<T> operator-(cplx<T> a, cplx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> __device__ __forceinline__ cplx<T> operator*(cplx<T> a, cplx<T> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T> __device__ __forceinline__ cplx<T> operator*(T s, cplx<T> a) { return {s * a.x, s * a.y}; }
template <typename T> __device__ __forceinline__ cplx<T> conj(cplx<T> a) { return {a.x, -a.y}; }
// a * conj(b)
template <typename T> __device__ __forceinline__ cplx<T> mulc(cplx<T> a, cplx<T> b) {
    return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
// multiply by +i / -i
template <typename T> __device__ __forceinline__ cplx<T> mul_i(cplx<T> a)  { return {-a.y, a.x}; }
template <typename T> __device__ __forceinline__ cplx<T> mul_mi(cplx<T> a) { return {a.y, -a.x}; }

// a + rot(b), rot = multiplication by -i (INV = false) / +i (INV = true); a +- conj(b)
template <bool INV, typename T> __device__ __forceinline__ cplx<T> addrot(cplx<T> a, cplx<T> b) {
    return INV ? cplx<T>(a.x - b.y, a.y + b.x) : cplx<T>(a.x + b.y, a.y - b.x);
}
template <typename T> __device__ __forceinline__ cplx<T> addc(cplx<T> a, cplx<T> b) { return {a.x + b.x, a.y - b.y}; }
template <typename T> __device__ __forceinline__ cplx<T> subc(cplx<T> a, cplx<T> b) { return {a.x - b.x, a.y + b.y}; }

// ---- complex64: packed overloads (non-template: preferred over the generic ones above)
using cf32 = cplx<float>;
__device__ __forceinline__ cf32 operator+(cf32 a, cf32 b) { return cf32(a.v + b.v); }
__device__ __forceinline__ cf32 operator-(cf32 a, cf32 b) { return cf32(a.v - b.v); }
__device__ __forceinline__ cf32 operator*(float s, cf32 a) { return cf32(a.v * s); }
__device__ __forceinline__ cf32 operator*(cf32 a, cf32 b) {
    v2f t, r;       // t = (ax bx, ay bx);  r = (-ay by + t.lo, ax by + t.hi)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a.v), "v"(b.v));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(a.v), "v"(b.v), "v"(t));
    return cf32(r);
}
__device__ __forceinline__ cf32 mulc(cf32 a, cf32 b) {       // a * conj(b)
    v2f t, r;       // t = (ax bx, ay bx);  r = (ay by + t.lo, -ax by + t.hi)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a.v), "v"(b.v));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(r) : "v"(a.v), "v"(b.v), "v"(t));
    return cf32(r);
}
template <bool INV> __device__ __forceinline__ cf32 addrot(cf32 a, cf32 b) {
    v2f r;          // INV: (ax - by, ay + bx)   else: (ax + by, ay - bx)
    if constexpr (INV)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a.v), "v"(b.v));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a.v), "v"(b.v));
    return cf32(r);
}
__device__ __forceinline__ cf32 addc(cf32 a, cf32 b) {       // a + conj(b)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a.v), "v"(b.v));
    return cf32(r);
}
__device__ __forceinline__ cf32 subc(cf32 a, cf32 b) {       // a - conj(b)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a.v), "v"(b.v));
    return cf32(r);
}
// (x, y) -> (x + y, y - x) = a (1 - i)   and   (x - y, x + y) = a (1 + i): the w_8 rotations
__device__ __forceinline__ cf32 mul_1mi(cf32 a) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a.v));
    return cf32(r);
}
__device__ __forceinline__ cf32 mul_1pi(cf32 a) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a.v));
    return cf32(r);
}
template <typename T> __device__ __forceinline__ cplx<T> mul_1mi(cplx<T> a) { return {a.x + a.y, a.y - a.x}; }
template <typename T> __device__ __forceinline__ cplx<T> mul_1pi(cplx<T> a) { return {a.x - a.y, a.x + a.y}; }

// ------------------------------------------------------------------ reductions
// wave64 shuffle reduction, then LDS across waves.  Result valid in thread 0.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// `red` must hold at least (blockDim.x/64) doubles per quantity.
template <int NQ>
__device__ __forceinline__ void block_sum(double (&v)[NQ], double* red) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double s = wave_sum(v[q]);
        if (lane == 0) red[q * nwave + wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double s = 0.0;
            for (int w = 0; w < nwave; ++w) s += red[q * nwave + w];
            v[q] = s;
        }
    }
    __syncthreads();
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline bool factorable(int n) {
    if (n < 1) return false;
    const int pr[] = {2, 3, 5, 7, 11, 13};
    for (int p : pr) while (n % p == 0) n /= p;
    return n == 1;
}
inline bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

}  // namespace pfb
