// fft_generic.hpp -- runtime-length mixed-radix Stockham FFT executed by one workgroup
// in LDS (ping-pong buffers).  Any N = 2^a 3^b 5^c 7^d 11^e 13^f whose two complex
// buffers fit in the 160 KB LDS of a gfx950 CU.  This is the coverage path (odd
// good_size lengths, PSF grids that are not exactly 2x the image); power-of-two
// grids with 2x oversampling take the register-resident path in fft_pow2.hpp.
//
// Stockham autosort pass of radix R (p = product of the radices already done):
//   for i < N/R:  k = i mod p
//     u[r] = src[i + r N/R] * w_{pR}^{r k}          r < R
//     v    = DFT_R(u)
//     dst[(i-k) R + k + s p] = v[s]                 s < R
// natural order in, natural order out, no bit reversal.
#pragma once
#include "common.hpp"

namespace pfb {

struct FftFactors {
    int n;          // transform length
    int npass;
    int radix[24];
};

inline bool plan_factors(int n, FftFactors* f) {
    f->n = n;
    f->npass = 0;
    int m = n;
    // radix 4 first (fewest LDS round trips for the power-of-two part), then the rest
    while (m % 4 == 0) { f->radix[f->npass++] = 4; m /= 4; }
    const int pr[] = {2, 3, 5, 7, 11, 13};
    for (int p : pr) while (m % p == 0) { f->radix[f->npass++] = p; m /= p; }
    return m == 1 && f->npass <= 24;
}

// tw[n * tws] = exp(-2 pi i n / N) for the transform length N of this call
// (tables are built for the full axis length; half-length transforms use tws = 2).
template <typename T, bool INV>
__device__ __forceinline__ cplx<T> twiddle(const cplx<T>* __restrict__ tw, int idx) {
    cplx<T> w = tw[idx];
    if (INV) w.y = -w.y;
    return w;
}

template <typename T, int R, bool INV>
__device__ void stockham_pass(const cplx<T>* __restrict__ src, cplx<T>* __restrict__ dst,
                              int N, int p, const cplx<T>* __restrict__ tw, int tws) {
    const int S = N / R;
    const int tstep = (N / (p * R)) * tws;      // index step of w_{pR}
    cplx<T> root[R];                             // R-th roots of unity
    if (R != 2 && R != 4) {
#pragma unroll
        for (int m = 0; m < R; ++m) root[m] = twiddle<T, INV>(tw, m * S * tws);
    }
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        const int k = i % p;
        cplx<T> u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = src[i + r * S];
        if (p > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = u[r] * twiddle<T, INV>(tw, r * k * tstep);
        }
        cplx<T> v[R];
        if (R == 2) {
            v[0] = u[0] + u[1];
            v[1] = u[0] - u[1];
        } else if (R == 4) {
            cplx<T> a = u[0] + u[2], b = u[0] - u[2];
            cplx<T> c = u[1] + u[3], d = u[1] - u[3];
            // forward: w_4 = -i ; inverse: +i
            cplx<T> di = INV ? mul_i(d) : mul_mi(d);
            v[0] = a + c;
            v[1] = b + di;
            v[2] = a - c;
            v[3] = b - di;
        } else {
#pragma unroll
            for (int s = 0; s < R; ++s) {
                cplx<T> acc = u[0];
#pragma unroll
                for (int r = 1; r < R; ++r) acc = acc + u[r] * root[(r * s) % R];
                v[s] = acc;
            }
        }
        const int j = (i - k) * R + k;
#pragma unroll
        for (int s = 0; s < R; ++s) dst[j + s * p] = v[s];
    }
}

// Whole FFT by one workgroup.  Data starts in bufA; returns the buffer holding the
// result (bufA or bufB).  Ends with a barrier.  All threads of the block must call.
template <typename T, bool INV>
__device__ cplx<T>* fft_lds_generic(cplx<T>* bufA, cplx<T>* bufB, const FftFactors& f,
                                    const cplx<T>* __restrict__ tw, int tws) {
    cplx<T>* src = bufA;
    cplx<T>* dst = bufB;
    int p = 1;
    __syncthreads();
    for (int s = 0; s < f.npass; ++s) {
        const int R = f.radix[s];
        switch (R) {
            case 2:  stockham_pass<T, 2, INV>(src, dst, f.n, p, tw, tws); break;
            case 3:  stockham_pass<T, 3, INV>(src, dst, f.n, p, tw, tws); break;
            case 4:  stockham_pass<T, 4, INV>(src, dst, f.n, p, tw, tws); break;
            case 5:  stockham_pass<T, 5, INV>(src, dst, f.n, p, tw, tws); break;
            case 7:  stockham_pass<T, 7, INV>(src, dst, f.n, p, tw, tws); break;
            case 11: stockham_pass<T, 11, INV>(src, dst, f.n, p, tw, tws); break;
            default: stockham_pass<T, 13, INV>(src, dst, f.n, p, tw, tws); break;
        }
        p *= R;
        cplx<T>* t = src; src = dst; dst = t;
        __syncthreads();
    }
    return src;
}

}  // namespace pfb
