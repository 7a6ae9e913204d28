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

// ------------------------------------------------------------------------------------------------
// The convolution itself on long lines (coverage path of pfb_psfconv_apply when a line fits neither the LDS nor the
// fast path: e.g. nx > 8192, or fp64 rows of more than 8192 pixels).  Same three stages and the same T / psf_l layouts
// as the line-in-LDS kernels of fftconv.hip (VB = 1: T[band][v][i], psf_l[band][v][u]), every FFT as global passes.

// z[bl][i][n] = (x[2n], x[2n+1]) [* beam], zero beyond ny
template <typename T>
__global__ void __launch_bounds__(256)
k_long_pack_x(const T* __restrict__ x, const T* __restrict__ beam, cplx<T>* __restrict__ z, int nx, int ny, int M) {
    const size_t r = (size_t)blockIdx.z * nx + blockIdx.y;
    const T* xr = x + r * ny;
    const T* br = beam ? beam + r * ny : nullptr;
    cplx<T>* zr = z + r * M;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < M; n += gridDim.x * blockDim.x) {
        const int j0 = 2 * n, j1 = 2 * n + 1;
        T a = 0, b = 0;
        if (j0 < ny) a = br ? xr[j0] * br[j0] : xr[j0];
        if (j1 < ny) b = br ? xr[j1] * br[j1] : xr[j1];
        zr[n] = cplx<T>(a, b);
    }
}

// T[band][v][i] = X[v] of row i (Hermitian unpacking of the packed transform), v = 0..M
template <typename T>
__global__ void __launch_bounds__(256)
k_long_rows_to_T(const cplx<T>* __restrict__ z, cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ twQ,
                 int nx, int M, size_t T_band, int band0) {
    const int i = blockIdx.y, bl = blockIdx.z;
    const cplx<T>* zr = z + ((size_t)bl * nx + i) * M;
    cplx<T>* Tb = Tw + (size_t)(band0 + bl) * T_band;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v <= M; v += gridDim.x * blockDim.x) {
        const cplx<T> zv = zr[v == M ? 0 : v];
        const cplx<T> zm = conj(zr[v == 0 ? 0 : M - v]);
        Tb[(size_t)v * nx + i] = T(0.5) * ((zv + zm) + mul_mi(twQ[v] * (zv - zm)));
    }
}

// C[v][u] = u < nx ? T[band][v][u] : 0     (one zero-padded column per line, contiguous)
template <typename T>
__global__ void __launch_bounds__(256)
k_long_col_load(const cplx<T>* __restrict__ Tb, cplx<T>* __restrict__ Cw, int nx, int P) {
    const size_t v = blockIdx.y;
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < P; u += gridDim.x * blockDim.x)
        Cw[v * P + u] = u < nx ? Tb[v * nx + u] : cplx<T>(0, 0);
}
template <typename T>
__global__ void __launch_bounds__(256)
k_long_col_mul(cplx<T>* __restrict__ Cw, const cplx<T>* __restrict__ psf_b, size_t n) {
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
        Cw[k] = Cw[k] * psf_b[k];
}
template <typename T>
__global__ void __launch_bounds__(256)
k_long_col_store(const cplx<T>* __restrict__ Cw, cplx<T>* __restrict__ Tb, int nx, int P) {
    const size_t v = blockIdx.y;
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < nx; u += gridDim.x * blockDim.x)
        Tb[v * nx + u] = Cw[v * P + u];
}

// z[bl][i][v] = (Y[v] + conj Y[M-v]) + i conj(w_Q^v) (Y[v] - conj Y[M-v]),  Y[v] = T[band][v][i]
template <typename T>
__global__ void __launch_bounds__(256)
k_long_T_to_rows(const cplx<T>* __restrict__ Tw, cplx<T>* __restrict__ z, const cplx<T>* __restrict__ twQ,
                 int nx, int M, size_t T_band, int band0) {
    const int i = blockIdx.y, bl = blockIdx.z;
    const cplx<T>* Tb = Tw + (size_t)(band0 + bl) * T_band;
    cplx<T>* zr = z + ((size_t)bl * nx + i) * M;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < M; v += gridDim.x * blockDim.x) {
        cplx<T> yv = Tb[(size_t)v * nx + i], ym = Tb[(size_t)(M - v) * nx + i];
        if (v == 0) { yv.y = 0; ym.y = 0; }
        ym = conj(ym);
        zr[v] = (yv + ym) + mul_i(mulc(yv - ym, twQ[v]));
    }
}

// out = z * scale [* beam] + sigmainv x, one workgroup per row; fused <dot_with,out>, <dot_with2,out>, <out,out>
template <typename T>
__global__ void __launch_bounds__(256)
k_long_epilogue(const cplx<T>* __restrict__ z, const T* __restrict__ x, const T* __restrict__ beam,
                const T* __restrict__ dot_with, const T* __restrict__ dot_with2, T* __restrict__ out,
                double* __restrict__ partials, int nx, int ny, int M, T scale, T sigmainv) {
    __shared__ double red[3 * 4];
    const int i = blockIdx.x, bl = blockIdx.y;
    const size_t rowoff = ((size_t)bl * nx + i) * ny;
    const cplx<T>* zr = z + ((size_t)bl * nx + i) * M;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int j = threadIdx.x; j < ny; j += blockDim.x) {
        const cplx<T> zz = zr[j >> 1];
        T val = ((j & 1) ? zz.y : zz.x) * scale;
        if (beam) val *= beam[rowoff + j];
        val += sigmainv * x[rowoff + j];
        out[rowoff + j] = val;
        if (dot_with) {
            acc[0] += (double)dot_with[rowoff + j] * (double)val;
            if (dot_with2) acc[1] += (double)dot_with2[rowoff + j] * (double)val;
            acc[2] += (double)val * (double)val;
        }
    }
    if (dot_with) {
        block_sum<3>(acc, red);
        if (threadIdx.x == 0) {
            const size_t np = (size_t)gridDim.x * gridDim.y, k = (size_t)bl * nx + i;
            partials[k] = acc[0]; partials[np + k] = acc[1]; partials[2 * np + k] = acc[2];
        }
    }
}

static inline dim3 long_grid(int n, int y, int zdim) {
    int gx = (n + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)y, (unsigned)zdim);
}

// workspace bytes of apply_long for nb bands: row buffers 2 x nb x nx x M, column buffers 2 x (M+1) x P (one band at a time)
template <typename T>
static size_t long_apply_ws_bytes(int nb, int nx, int M, int P) {
    const size_t rows = 2 * (size_t)nb * nx * M, cols = 2 * (size_t)(M + 1) * P;
    return sizeof(cplx<T>) * (rows > cols ? rows : cols);
}

template <typename T>
static int apply_long(cplx<T>* ws, cplx<T>* Tw, const cplx<T>* psf_l, double* partials, const cplx<T>* twP,
                      const cplx<T>* twQ, const FftFactors& frow, const FftFactors& fcol, int nx, int ny, int P, int M,
                      size_t T_band, size_t psf_band, int band0, int nb, const T* x, const T* beam, double scale,
                      double sigmainv, T* out, const T* dot_with, const T* dot_with2, hipStream_t st) {
    const size_t nrow = (size_t)nb * nx * M, ncol = (size_t)(M + 1) * P;
    cplx<T>* z = ws;
    cplx<T>* zw = ws + nrow;
    // 1. rows forward
    hipLaunchKernelGGL((k_long_pack_x<T>), long_grid(M, nx, nb), dim3(256), 0, st, x, beam, z, nx, ny, M);
    int rc = long_fft<T, false>(z, zw, nrow, frow, twQ, 2, (size_t)nb * nx, 1, (size_t)M, st);
    if (rc != PFB_OK) return rc;
    hipLaunchKernelGGL((k_long_rows_to_T<T>), long_grid(M + 1, nx, nb), dim3(256), 0, st, (const cplx<T>*)z, Tw, twQ, nx, M,
                       T_band, band0);
    // 2. columns, band by band: zero-pad, forward, multiply by psfhat, inverse, keep the first nx samples
    cplx<T>* C1 = ws;
    cplx<T>* C2 = ws + ncol;
    for (int bl = 0; bl < nb && rc == PFB_OK; ++bl) {
        cplx<T>* Tb = Tw + (size_t)(band0 + bl) * T_band;
        hipLaunchKernelGGL((k_long_col_load<T>), long_grid(P, M + 1, 1), dim3(256), 0, st, (const cplx<T>*)Tb, C1, nx, P);
        rc = long_fft<T, false>(C1, C2, ncol, fcol, twP, 1, (size_t)(M + 1), 1, (size_t)P, st);
        if (rc != PFB_OK) break;
        hipLaunchKernelGGL((k_long_col_mul<T>), dim3(4096), dim3(256), 0, st, C1, psf_l + (size_t)(band0 + bl) * psf_band, ncol);
        rc = long_fft<T, true>(C1, C2, ncol, fcol, twP, 1, (size_t)(M + 1), 1, (size_t)P, st);
        if (rc != PFB_OK) break;
        hipLaunchKernelGGL((k_long_col_store<T>), long_grid(nx, M + 1, 1), dim3(256), 0, st, (const cplx<T>*)C1, Tb, nx, P);
    }
    if (rc != PFB_OK) return rc;
    // 3. rows inverse + epilogue
    hipLaunchKernelGGL((k_long_T_to_rows<T>), long_grid(M, nx, nb), dim3(256), 0, st, (const cplx<T>*)Tw, z, twQ, nx, M,
                       T_band, band0);
    rc = long_fft<T, true>(z, zw, nrow, frow, twQ, 2, (size_t)nb * nx, 1, (size_t)M, st);
    if (rc != PFB_OK) return rc;
    hipLaunchKernelGGL((k_long_epilogue<T>), dim3(nx, nb), dim3(256), 0, st, (const cplx<T>*)z, x, beam, dot_with, dot_with2,
                       out, partials, nx, ny, M, (T)scale, (T)sigmainv);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

}  // namespace pfb
