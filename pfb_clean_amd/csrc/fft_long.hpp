// fft_long.hpp -- plan-time line FFTs of ANY 13-smooth length, in global memory.
//
// The coverage kernels (fft_generic.hpp) keep one line in two LDS buffers, which caps a line at 10240 complex64 /
// 5120 complex128 values.  PSF grids beyond that (nx_psf up to 16384 and more: 6000^2 / 7200^2 MeerKAT images
// with psf-oversize 2) still have to be transformed ONCE per gridding run: psfhat = r2c(ifftshift(psf))
// (gridder.py:712-714) and the re-gridding of a caller's psfhat onto the power-of-two grid of the fast kernels
// (pfb_psfhat_regrid).  Here every Stockham pass (same recurrence as fft_generic.hpp) is its own launch over all
// lines of a batch, ping-ponging between two global buffers: N log N work, radix-sized passes, no length limit.
// Plan time only -- nothing here runs inside the PCG / PD loops.
#pragma once
#include "common.hpp"
#include "fft_generic.hpp"

namespace pfb {

// one Stockham pass of radix R over `nlines` lines of length N.
//   element e of line l lives at base + l * ls + e * es  (same strides in src and dst)
//   tw[n * tws] = exp(-2 pi i n / N)
// line_fast: consecutive threads take consecutive LINES (column transforms of a row-major array: ls = 1)
template <typename T, int R, bool INV>
__global__ void __launch_bounds__(256)
k_long_pass(const cplx<T>* __restrict__ src, cplx<T>* __restrict__ dst, int N, int p,
            const cplx<T>* __restrict__ tw, int tws, size_t nlines, size_t es, size_t ls, int line_fast) {
    const int S = N / R;
    const size_t total = nlines * (size_t)S;
    const int tstep = (N / (p * R)) * tws;
    cplx<T> root[R];
    if (R != 2 && R != 4) {
#pragma unroll
        for (int m = 0; m < R; ++m) root[m] = twiddle<T, INV>(tw, m * S * tws);
    }
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t l;
        int i;
        if (line_fast) { l = idx % nlines; i = (int)(idx / nlines); }
        else           { i = (int)(idx % S); l = idx / S; }
        const int k = i % p;
        const cplx<T>* sp = src + l * ls;
        cplx<T> u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = sp[(size_t)(i + r * S) * es];
        if (p > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = u[r] * twiddle<T, INV>(tw, r * k * tstep);
        }
        cplx<T> v[R];
        if (R == 2) {
            v[0] = u[0] + u[1];
            v[1] = u[0] - u[1];
        } else if (R == 4) {
            const cplx<T> a = u[0] + u[2], b = u[0] - u[2], c = u[1] + u[3], d = u[1] - u[3];
            const cplx<T> di = INV ? mul_i(d) : mul_mi(d);
            v[0] = a + c; v[1] = b + di; v[2] = a - c; v[3] = b - di;
        } else {
#pragma unroll
            for (int s = 0; s < R; ++s) {
                cplx<T> acc = u[0];
#pragma unroll
                for (int r = 1; r < R; ++r) acc = acc + u[r] * root[(r * s) % R];
                v[s] = acc;
            }
        }
        cplx<T>* dp = dst + l * ls;
        const int j = (i - k) * R + k;
#pragma unroll
        for (int s = 0; s < R; ++s) dp[(size_t)(j + s * p) * es] = v[s];
    }
}

// All passes of a batch; the result is left in `data` (`work`: same size / layout).
template <typename T, bool INV>
static int long_fft(cplx<T>* data, cplx<T>* work, size_t total_elems, const FftFactors& f, const cplx<T>* tw, int tws,
                    size_t nlines, size_t es, size_t ls, hipStream_t st) {
    cplx<T>* src = data;
    cplx<T>* dst = work;
    int p = 1;
    const int line_fast = ls < es ? 1 : 0;
    for (int s = 0; s < f.npass; ++s) {
        const int R = f.radix[s];
        const size_t items = nlines * (size_t)(f.n / R);
        size_t g = (items + 255) / 256;
        if (g > 65536) g = 65536;
        if (g < 1) g = 1;
#define PFB_LP(RR) hipLaunchKernelGGL((k_long_pass<T, RR, INV>), dim3((unsigned)g), dim3(256), 0, st, \
                                      (const cplx<T>*)src, dst, f.n, p, tw, tws, nlines, es, ls, line_fast)
        switch (R) {
            case 2:  PFB_LP(2); break;
            case 3:  PFB_LP(3); break;
            case 4:  PFB_LP(4); break;
            case 5:  PFB_LP(5); break;
            case 7:  PFB_LP(7); break;
            case 11: PFB_LP(11); break;
            default: PFB_LP(13); break;
        }
#undef PFB_LP
        p *= R;
        cplx<T>* t = src; src = dst; dst = t;
    }
    if (src != data)
        PFB_HIP_CHECK(hipMemcpyAsync(data, src, sizeof(cplx<T>) * total_elems, hipMemcpyDeviceToDevice, st));
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

// ---- rows of a real array as packed complex transforms (the long-line versions of k_psfhat_rows / k_psf_rows_c2r)
template <typename T>
__global__ void __launch_bounds__(256)
k_long_pack_rows(const T* __restrict__ psf, cplx<T>* __restrict__ z, int P, int Q, int shift) {
    const int M = Q / 2;
    const int u = blockIdx.y, band = blockIdx.z;
    const int su = shift ? P / 2 : 0, sv = shift ? M : 0;
    const T* row = psf + ((size_t)band * P + (u + su) % P) * Q;
    cplx<T>* zr = z + ((size_t)band * P + u) * M;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < M; n += gridDim.x * blockDim.x)
        zr[n] = cplx<T>(row[(2 * n + sv) % Q], row[(2 * n + 1 + sv) % Q]);
}

// X[v] = 1/2 [ (Z[v] + conj Z[M-v]) - i w_Q^v (Z[v] - conj Z[M-v]) ],  v = 0..M
template <typename T>
__global__ void __launch_bounds__(256)
k_long_post_rows(const cplx<T>* __restrict__ z, cplx<T>* __restrict__ out, const cplx<T>* __restrict__ twQ, int P, int Q) {
    const int M = Q / 2;
    const size_t r = (size_t)blockIdx.z * P + blockIdx.y;
    const cplx<T>* zr = z + r * M;
    cplx<T>* orow = out + r * (M + 1);
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v <= M; v += gridDim.x * blockDim.x) {
        const cplx<T> zv = zr[v == M ? 0 : v];
        const cplx<T> zm = conj(zr[v == 0 ? 0 : M - v]);
        orow[v] = T(0.5) * ((zv + zm) + mul_mi(twQ[v] * (zv - zm)));
    }
}

// Z[v] = (Y[v] + conj Y[M-v]) + i conj(w_Q^v) (Y[v] - conj Y[M-v]),  v < M  (DC / Nyquist imaginary parts ignored)
template <typename T>
__global__ void __launch_bounds__(256)
k_long_pre_rows(const cplx<T>* __restrict__ spec, cplx<T>* __restrict__ z, const cplx<T>* __restrict__ twQ, int P, int Q) {
    const int M = Q / 2;
    const size_t r = (size_t)blockIdx.z * P + blockIdx.y;
    const cplx<T>* srow = spec + r * (M + 1);
    cplx<T>* zr = z + r * M;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < M; v += gridDim.x * blockDim.x) {
        cplx<T> yv = srow[v], ym = srow[M - v];
        if (v == 0) { yv.y = 0; ym.y = 0; }
        ym = conj(ym);
        zr[v] = (yv + ym) + mul_i(mulc(yv - ym, twQ[v]));
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_long_unpack_rows(const cplx<T>* __restrict__ z, T* __restrict__ psf, int P, int Q) {
    const int M = Q / 2;
    const size_t r = (size_t)blockIdx.z * P + blockIdx.y;
    const cplx<T>* zr = z + r * M;
    T* orow = psf + r * Q;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < M; n += gridDim.x * blockDim.x) {
        const cplx<T> v = zr[n];
        orow[2 * n] = v.x;
        orow[2 * n + 1] = v.y;
    }
}

static inline dim3 long_row_grid(int M, int P, int nband) {
    int gx = (M + 255) / 256;
    if (gx > 64) gx = 64;
    return dim3((unsigned)gx, (unsigned)P, (unsigned)nband);
}

// psf (nband, P, Q) real -> out (nband, P, M+1): r2c of every row (shift: rows / samples read through ifftshift).
// twQ: exp(-2 pi i n / Q), n < Q.  Allocates (and frees) 2 x nband x P x M complex of scratch.
template <typename T>
static int long_rows_r2c(const T* psf, cplx<T>* out, const cplx<T>* twQ, int nband, int P, int Q, const FftFactors& f,
                         int shift, hipStream_t st) {
    const int M = Q / 2;
    const size_t n = (size_t)nband * P * M;
    void *z = nullptr, *w = nullptr;
    if (hipMalloc(&z, n * sizeof(cplx<T>)) != hipSuccess || hipMalloc(&w, n * sizeof(cplx<T>)) != hipSuccess) {
        if (z) (void)hipFree(z);
        set_error("long_rows_r2c: device allocation failed (%zu B)", 2 * n * sizeof(cplx<T>));
        return PFB_ERR_ALLOC;
    }
    hipLaunchKernelGGL((k_long_pack_rows<T>), long_row_grid(M, P, nband), dim3(256), 0, st, psf, (cplx<T>*)z, P, Q, shift);
    int rc = long_fft<T, false>((cplx<T>*)z, (cplx<T>*)w, n, f, twQ, 2, (size_t)nband * P, 1, (size_t)M, st);
    if (rc == PFB_OK) {
        hipLaunchKernelGGL((k_long_post_rows<T>), long_row_grid(M + 1, P, nband), dim3(256), 0, st, (const cplx<T>*)z, out,
                           twQ, P, Q);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            set_error("long_rows_r2c: kernel launch failed");
            rc = PFB_ERR_HIP;
        }
    }
    (void)hipFree(z);
    (void)hipFree(w);
    return rc;
}

// spec (nband, P, M+1) -> psf (nband, P, Q) real, unnormalised c2r of every row
template <typename T>
static int long_rows_c2r(const cplx<T>* spec, T* psf, const cplx<T>* twQ, int nband, int P, int Q, const FftFactors& f,
                         hipStream_t st) {
    const int M = Q / 2;
    const size_t n = (size_t)nband * P * M;
    void *z = nullptr, *w = nullptr;
    if (hipMalloc(&z, n * sizeof(cplx<T>)) != hipSuccess || hipMalloc(&w, n * sizeof(cplx<T>)) != hipSuccess) {
        if (z) (void)hipFree(z);
        set_error("long_rows_c2r: device allocation failed (%zu B)", 2 * n * sizeof(cplx<T>));
        return PFB_ERR_ALLOC;
    }
    hipLaunchKernelGGL((k_long_pre_rows<T>), long_row_grid(M, P, nband), dim3(256), 0, st, spec, (cplx<T>*)z, twQ, P, Q);
    int rc = long_fft<T, true>((cplx<T>*)z, (cplx<T>*)w, n, f, twQ, 2, (size_t)nband * P, 1, (size_t)M, st);
    if (rc == PFB_OK) {
        hipLaunchKernelGGL((k_long_unpack_rows<T>), long_row_grid(M, P, nband), dim3(256), 0, st, (const cplx<T>*)z, psf, P, Q);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            set_error("long_rows_c2r: kernel launch failed");
            rc = PFB_ERR_HIP;
        }
    }
    (void)hipFree(z);
    (void)hipFree(w);
    return rc;
}

// data (nband, P, M1) complex, row-major: FFT of length P down every column, in place.  twP: exp(-2 pi i n / P).
// Allocates (and frees) one band (P x M1 complex) of scratch.
template <typename T, bool INV>
static int long_cols(cplx<T>* data, const cplx<T>* twP, int nband, int P, int M1, const FftFactors& f, hipStream_t st) {
    const size_t n = (size_t)P * M1;
    void* w = nullptr;
    if (hipMalloc(&w, n * sizeof(cplx<T>)) != hipSuccess) {
        set_error("long_cols: device allocation failed (%zu B)", n * sizeof(cplx<T>));
        return PFB_ERR_ALLOC;
    }
    int rc = PFB_OK;
    for (int b = 0; b < nband && rc == PFB_OK; ++b)
        rc = long_fft<T, INV>(data + (size_t)b * n, (cplx<T>*)w, n, f, twP, 1, (size_t)M1, (size_t)M1, 1, st);
    if (rc == PFB_OK && hipStreamSynchronize(st) != hipSuccess) {
        set_error("long_cols: kernel launch failed");
        rc = PFB_ERR_HIP;
    }
    (void)hipFree(w);
    return rc;
}

}  // namespace pfb
