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
