// fftconv.hip -- PSF convolution plan + the coverage ("generic") kernels + C-ABI.
//
// out = [beam*] crop(irfft2(rfft2(pad([beam*] x)) * psfhat)) [/wsum] + sigmainv * x
// (pfb/operators/psf.py:11-56, pfb/operators/hessian.py:129-158, 254-281)
//
// Three launches per apply, never materialising the zero-padded image:
//   1. row_fwd : one image row -> packed real FFT of length Q -> M+1 bins into T
//   2. col     : one frequency column of T -> FFT_P -> * psfhat -> IFFT_P -> first nx
//                samples back into T (in place)
//   3. row_inv : M+1 bins of one output row -> c2r of length Q -> crop, scale, beam,
//                Tikhonov term, fused <dot_with, out> partial sum
// The generic kernels here run one line per 256-thread workgroup with a runtime
// mixed-radix Stockham FFT in LDS; fftconv_pow2.hip supplies the fast versions of the
// same three stages on the same layouts.
#include "conv_plan.hpp"
#include "fft_long.hpp"
#include <vector>
#include <cstring>
#include <cstdlib>

namespace pfb {

thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// fast-path hooks (fftconv_pow2.hip)
bool pow2_supported(const pfb_conv_plan* p);
int pow2_apply(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam,
               double scale, double sigmainv, void* out, const void* dot_with, const void* dot_with2,
               hipStream_t st);
int pow2_prepare(pfb_conv_plan* p);
void pow2_release(pfb_conv_plan* p);
int pow2_rows_per_wg(const pfb_conv_plan* p);
int pow2_nblocks(const pfb_conv_plan* p);
int pow2_nvb(const pfb_conv_plan* p);
int pow2_set_psfhat(pfb_conv_plan* p, const void* psfhat, hipStream_t st);
int pow2_set_psf(pfb_conv_plan* p, const void* psf, void* psfhat_out, hipStream_t st);

struct ConvDims {
    int nx, ny, P, Q, M, VB, nvb;
    size_t T_band, psf_band;
};

__device__ __forceinline__ size_t t_index(const ConvDims& d, int band, int i, int v) {
    return (size_t)band * d.T_band + ((size_t)(v / d.VB) * d.nx + i) * d.VB + (v % d.VB);
}
__device__ __forceinline__ size_t psf_index(const ConvDims& d, int band, int u, int v) {
    return (size_t)band * d.psf_band + ((size_t)(v / d.VB) * d.P + u) * d.VB + (v % d.VB);
}

// ---------------------------------------------------------------- psfhat re-layout
template <typename T>
__global__ void k_relayout_psfhat(const cplx<T>* __restrict__ psfhat, cplx<T>* __restrict__ psf_l,
                                  ConvDims d) {
    // grid: (ceil(nvb*VB / 64), P, nband); thread -> v
    const int band = blockIdx.z;
    const int u = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nvb * d.VB) return;
    cplx<T> val(0, 0);
    if (v <= d.M) val = psfhat[((size_t)band * d.P + u) * (d.M + 1) + v];
    psf_l[psf_index(d, band, u, v)] = val;
}

// ------------------------------------------------------------------- row forward
template <typename T>
__global__ void __launch_bounds__(256)
k_row_fwd_generic(const T* __restrict__ x, const T* __restrict__ beam, cplx<T>* __restrict__ Tw,
                  const cplx<T>* __restrict__ twQ, ConvDims d, FftFactors f, int band0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T>* bufA = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* bufB = bufA + d.M;
    const int i = blockIdx.x;
    const int bl = blockIdx.y;                 // local band (x/beam index)
    const int band = band0 + bl;               // plan band (T/psf index)
    const T* xr = x + ((size_t)bl * d.nx + i) * d.ny;
    const T* br = beam ? beam + ((size_t)bl * d.nx + i) * d.ny : nullptr;
    // pack z[n] = x[2n] + i x[2n+1], zero beyond ny
    for (int n = threadIdx.x; n < d.M; n += blockDim.x) {
        const int j0 = 2 * n, j1 = 2 * n + 1;
        T a = 0, b = 0;
        if (j0 < d.ny) a = br ? xr[j0] * br[j0] : xr[j0];
        if (j1 < d.ny) b = br ? xr[j1] * br[j1] : xr[j1];
        bufA[n] = cplx<T>(a, b);
    }
    cplx<T>* Z = fft_lds_generic<T, false>(bufA, bufB, f, twQ, 2);
    // X[v] = 1/2 [ (Z[v] + conj Z[M-v]) - i w_Q^v (Z[v] - conj Z[M-v]) ],  v = 0..M
    for (int v = threadIdx.x; v <= d.M; v += blockDim.x) {
        const cplx<T> zv = Z[v == d.M ? 0 : v];
        const cplx<T> zm = conj(Z[v == 0 ? 0 : d.M - v]);
        const cplx<T> w = twQ[v];
        const cplx<T> s = zv + zm;
        const cplx<T> t = mul_mi(w * (zv - zm));
        Tw[t_index(d, band, i, v)] = T(0.5) * (s + t);
    }
}

// ------------------------------------------------------------------------ column
template <typename T>
__global__ void __launch_bounds__(256)
k_col_generic(cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ psf_l,
              const cplx<T>* __restrict__ twP, ConvDims d, FftFactors f, int band0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T>* bufA = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* bufB = bufA + d.P;
    const int v = blockIdx.x;
    const int band = band0 + blockIdx.y;
    for (int n = threadIdx.x; n < d.P; n += blockDim.x)
        bufA[n] = n < d.nx ? Tw[t_index(d, band, n, v)] : cplx<T>(0, 0);
    cplx<T>* X = fft_lds_generic<T, false>(bufA, bufB, f, twP, 1);
    for (int u = threadIdx.x; u < d.P; u += blockDim.x)
        X[u] = X[u] * psf_l[psf_index(d, band, u, v)];
    cplx<T>* other = (X == bufA) ? bufB : bufA;
    cplx<T>* Y = fft_lds_generic<T, true>(X, other, f, twP, 1);
    for (int n = threadIdx.x; n < d.nx; n += blockDim.x)
        Tw[t_index(d, band, n, v)] = Y[n];
}

// ------------------------------------------------------------------- row inverse
template <typename T>
__global__ void __launch_bounds__(256)
k_row_inv_generic(const cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ twQ,
                  const T* __restrict__ x, const T* __restrict__ beam,
                  const T* __restrict__ dot_with, const T* __restrict__ dot_with2,
                  T* __restrict__ out, double* __restrict__ partials, ConvDims d, FftFactors f,
                  int band0, T scale, T sigmainv) {
    // all LDS in the one dynamic array (a static __shared__ beside it would eat into
    // the 160 KB limit and shift the 16-byte alignment of the dynamic base)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);              // 3 sums x 4 waves = 96 B
    cplx<T>* bufA = reinterpret_cast<cplx<T>*>(smem + 128);
    cplx<T>* bufB = bufA + d.M;
    const int i = blockIdx.x;
    const int bl = blockIdx.y;
    const int band = band0 + bl;
    // Z[v] = (Y[v] + conj Y[M-v]) + i conj(w_Q^v) (Y[v] - conj Y[M-v]),  v < M
    // imaginary parts of the DC and Nyquist bins are ignored (ducc0/pocketfft c2r)
    for (int v = threadIdx.x; v < d.M; v += blockDim.x) {
        cplx<T> yv = Tw[t_index(d, band, i, v)];
        cplx<T> ym = Tw[t_index(d, band, i, d.M - v)];
        if (v == 0) { yv.y = 0; ym.y = 0; }
        ym = conj(ym);
        const cplx<T> w = twQ[v];
        bufA[v] = (yv + ym) + mul_i(mulc(yv - ym, w));
    }
    cplx<T>* z = fft_lds_generic<T, true>(bufA, bufB, f, twQ, 2);
    const size_t rowoff = ((size_t)bl * d.nx + i) * d.ny;
    // fused sums: <dot_with, out>, <dot_with2, out>, <out, out>
    double acc[3] = {0.0, 0.0, 0.0};
    for (int j = threadIdx.x; j < d.ny; j += blockDim.x) {
        const cplx<T> zz = z[j >> 1];
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
            const size_t np = (size_t)gridDim.x * gridDim.y, k = (size_t)bl * d.nx + i;
            partials[k] = acc[0]; partials[np + k] = acc[1]; partials[2 * np + k] = acc[2];
        }
    }
}

// ------------------------------------------------------------- PSFHAT producer
// psfhat = r2c(ifftshift(psf), axes=(0,1), forward, unnormalised)   (gridder.py:712-714, fft.py:7-9)
// ifftshift: y[i] = x[(i + n/2) % n] (n/2 rounded down).  One workgroup per line: rows as
// packed-real transforms of length M = Q/2 (+ Hermitian unpacking), then columns of length P
// in place -- once per gridding run, so the column pass simply takes the strided 8/16-byte
// accesses of the caller's (P, M+1) row-major layout.
template <typename T>
__global__ void __launch_bounds__(256)
k_psfhat_rows(const T* __restrict__ psf, cplx<T>* __restrict__ out, const cplx<T>* __restrict__ twQ,
              int P, int Q, FftFactors f, int shift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = Q / 2;
    cplx<T>* bufA = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* bufB = bufA + M;
    const int u = blockIdx.x, band = blockIdx.y;
    const int su = shift ? P / 2 : 0, sv = shift ? M : 0;                     // Q/2 = M
    const T* row = psf + ((size_t)band * P + (u + su) % P) * Q;
    for (int n = threadIdx.x; n < M; n += blockDim.x)
        bufA[n] = cplx<T>(row[(2 * n + sv) % Q], row[(2 * n + 1 + sv) % Q]);
    cplx<T>* Z = fft_lds_generic<T, false>(bufA, bufB, f, twQ, 2);
    cplx<T>* orow = out + ((size_t)band * P + u) * (M + 1);
    for (int v = threadIdx.x; v <= M; v += blockDim.x) {
        const cplx<T> zv = Z[v == M ? 0 : v];
        const cplx<T> zm = conj(Z[v == 0 ? 0 : M - v]);
        orow[v] = T(0.5) * ((zv + zm) + mul_mi(twQ[v] * (zv - zm)));
    }
}

template <typename T, bool INV>
__global__ void __launch_bounds__(256)
k_psfhat_cols(cplx<T>* __restrict__ out, const cplx<T>* __restrict__ twP, int P, int M1, FftFactors f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T>* bufA = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* bufB = bufA + P;
    cplx<T>* col = out + (size_t)blockIdx.y * P * M1 + blockIdx.x;
    for (int n = threadIdx.x; n < P; n += blockDim.x) bufA[n] = col[(size_t)n * M1];
    cplx<T>* X = fft_lds_generic<T, INV>(bufA, bufB, f, twP, 1);
    for (int n = threadIdx.x; n < P; n += blockDim.x) col[(size_t)n * M1] = X[n];
}

// inverse of k_psfhat_rows (no shift): half spectrum row (M+1 bins, after the inverse column
// pass) -> Q real samples, unnormalised; DC / Nyquist imaginary parts ignored like ducc's c2r
template <typename T>
__global__ void __launch_bounds__(256)
k_psf_rows_c2r(const cplx<T>* __restrict__ spec, T* __restrict__ psf, const cplx<T>* __restrict__ twQ,
               int P, int Q, FftFactors f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = Q / 2;
    cplx<T>* bufA = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* bufB = bufA + M;
    const int u = blockIdx.x, band = blockIdx.y;
    const cplx<T>* srow = spec + ((size_t)band * P + u) * (M + 1);
    for (int v = threadIdx.x; v < M; v += blockDim.x) {
        cplx<T> yv = srow[v], ym = srow[M - v];
        if (v == 0) { yv.y = 0; ym.y = 0; }
        ym = conj(ym);
        bufA[v] = (yv + ym) + mul_i(mulc(yv - ym, twQ[v]));
    }
    cplx<T>* z = fft_lds_generic<T, true>(bufA, bufB, f, twQ, 2);
    T* orow = psf + ((size_t)band * P + u) * Q;
    for (int n = threadIdx.x; n < M; n += blockDim.x) {
        orow[2 * n] = z[n].x;
        orow[2 * n + 1] = z[n].y;
    }
}

// psf2[u2][v2] = scale * psf[(du mod P)][(dv mod Q)] for the offsets |du| < nx, |dv| < ny a
// convolution of an (nx, ny) image can touch (origin at index 0, periodic in the OLD grid --
// which reproduces the wrap-around of grids with nx_psf < 2 nx exactly), zero elsewhere
template <typename T>
__global__ void __launch_bounds__(256)
k_psf_embed(const T* __restrict__ psf, T* __restrict__ psf2, int nx, int ny, int P, int Q, int P2, int Q2,
            T scale) {
    const int v2 = blockIdx.x * blockDim.x + threadIdx.x, u2 = blockIdx.y, band = blockIdx.z;
    if (v2 >= Q2) return;
    const int du = u2 < nx ? u2 : u2 - P2, dv = v2 < ny ? v2 : v2 - Q2;
    T val = 0;
    if (du > -nx && dv > -ny) {
        const int u = ((du % P) + P) % P, v = ((dv % Q) + Q) % Q;
        val = scale * psf[((size_t)band * P + u) * Q + v];
    }
    psf2[((size_t)band * P2 + u2) * Q2 + v2] = val;
}

// out[q] = sum of the n partials of quantity q (q < nq), fixed order => deterministic
__global__ void __launch_bounds__(256)
k_sum_partials(const double* __restrict__ partials, int n, int nq, double* __restrict__ out) {
    __shared__ double red[4];
    for (int q = 0; q < nq; ++q) {
        double acc[1] = {0.0};
        for (int k = threadIdx.x; k < n; k += blockDim.x) acc[0] += partials[(size_t)q * n + k];
        block_sum<1>(acc, red);
        if (threadIdx.x == 0) out[q] = acc[0];
    }
}

static ConvDims dims_of(const pfb_conv_plan* p) {
    ConvDims d;
    d.nx = p->nx; d.ny = p->ny; d.P = p->P; d.Q = p->Q; d.M = p->M;
    d.VB = p->VB; d.nvb = p->nvb;
    d.T_band = p->T_elems_per_band; d.psf_band = p->psf_elems_per_band;
    return d;
}

template <typename T>
static int upload_twiddles(int n, void** dev) {
    std::vector<cplx<T>> h(n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n; ++k) {
        // octant-exact: reduce the angle so that sin/cos see |a| <= pi/4 where possible
        long double a = two_pi * (long double)k / (long double)n;
        h[k].x = (T)cosl(a);
        h[k].y = (T)(-sinl(a));
    }
    PFB_HIP_CHECK(hipMalloc(dev, sizeof(cplx<T>) * (size_t)n));
    PFB_HIP_CHECK(hipMemcpy(*dev, h.data(), sizeof(cplx<T>) * (size_t)n, hipMemcpyHostToDevice));
    return PFB_OK;
}

template <typename T>
static int apply_generic(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam,
                         double scale, double sigmainv, void* out, const void* dot_with,
                         const void* dot_with2, hipStream_t st) {
    const ConvDims d = dims_of(p);
    const size_t lds_row = 128 + 2 * sizeof(cplx<T>) * (size_t)p->M;
    const size_t lds_col = 2 * sizeof(cplx<T>) * (size_t)p->P;
    prof_mark(p, st, 0);
    hipLaunchKernelGGL((k_row_fwd_generic<T>), dim3(p->nx, nb), dim3(256), lds_row, st,
                       (const T*)x, (const T*)beam, (cplx<T>*)p->T, (const cplx<T>*)p->twQ,
                       d, p->frow, band0);
    prof_mark(p, st, 1);
    hipLaunchKernelGGL((k_col_generic<T>), dim3(p->M + 1, nb), dim3(256), lds_col, st,
                       (cplx<T>*)p->T, (const cplx<T>*)p->psf_l, (const cplx<T>*)p->twP,
                       d, p->fcol, band0);
    prof_mark(p, st, 2);
    hipLaunchKernelGGL((k_row_inv_generic<T>), dim3(p->nx, nb), dim3(256), lds_row, st,
                       (const cplx<T>*)p->T, (const cplx<T>*)p->twQ, (const T*)x,
                       (const T*)beam, (const T*)dot_with, (const T*)dot_with2, (T*)out, p->partials, d,
                       p->frow, band0, (T)scale, (T)sigmainv);
    prof_mark(p, st, 3);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

template <typename T>
static int set_lds_limits(const pfb_conv_plan*) {
    // the attribute is per function, not per launch: always raise it to the whole
    // 160 KB of a gfx950 CU so plans of different sizes can coexist
    const int lds_max = 160 * 1024;
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_row_fwd_generic<T>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_row_inv_generic<T>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_col_generic<T>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_psfhat_rows<T>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_psfhat_cols<T, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_psfhat_cols<T, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_psf_rows_c2r<T>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    return PFB_OK;
}

// a line of n complex values fits the LDS ping-pong buffers of the one-workgroup-per-line kernels
template <typename T> static inline bool line_fits_lds(int n) { return 2 * sizeof(cplx<T>) * (size_t)n <= (size_t)160 * 1024; }

// psfhat_out (nband, P, M+1) = r2c(ifftshift(psf)) for ANY 13-smooth grid: rows, then columns in place; a line that
// fits the LDS takes the one-workgroup-per-line kernels, a longer one the global-memory passes (fft_long.hpp).
// twP / twQ: exp(-2 pi i n / P), exp(-2 pi i n / Q).  Synchronous.
template <typename T>
static int psfhat_from_psf_t(const void* psf, void* psfhat_out, int nband, int P, int Q, const FftFactors& frow,
                             const FftFactors& fcol, const void* twP, const void* twQ, hipStream_t st) {
    const int M = Q / 2;
    const size_t lds_r = 2 * sizeof(cplx<T>) * (size_t)M, lds_c = 2 * sizeof(cplx<T>) * (size_t)P;
    if (line_fits_lds<T>(M))
        hipLaunchKernelGGL((k_psfhat_rows<T>), dim3(P, nband), dim3(256), lds_r, st, (const T*)psf,
                           (cplx<T>*)psfhat_out, (const cplx<T>*)twQ, P, Q, frow, 1);
    else if (int rc = long_rows_r2c<T>((const T*)psf, (cplx<T>*)psfhat_out, (const cplx<T>*)twQ, nband, P, Q, frow, 1, st);
             rc != PFB_OK)
        return rc;
    if (line_fits_lds<T>(P))
        hipLaunchKernelGGL((k_psfhat_cols<T, false>), dim3(M + 1, nband), dim3(256), lds_c, st,
                           (cplx<T>*)psfhat_out, (const cplx<T>*)twP, P, M + 1, fcol);
    else if (int rc = long_cols<T, false>((cplx<T>*)psfhat_out, (const cplx<T>*)twP, nband, P, M + 1, fcol, st);
             rc != PFB_OK)
        return rc;
    PFB_HIP_CHECK(hipGetLastError());
    PFB_HIP_CHECK(hipStreamSynchronize(st));
    return PFB_OK;
}

// plan-less form (pfb_psfhat_from_psf): builds its own factor lists and twiddle tables
template <typename T>
static int psfhat_from_psf_grid(const void* psf, void* psfhat_out, int nband, int P, int Q, hipStream_t st) {
    FftFactors fr, fc;
    PFB_REQUIRE(plan_factors(Q / 2, &fr) && plan_factors(P, &fc), PFB_ERR_UNSUPPORTED,
                "psfhat_from_psf: (%d,%d) has a prime factor > 13", P, Q);
    void *twP = nullptr, *twQ = nullptr;
    int rc = upload_twiddles<T>(P, &twP);
    if (rc == PFB_OK) rc = upload_twiddles<T>(Q, &twQ);
    if (rc == PFB_OK) rc = set_lds_limits<T>(nullptr);
    if (rc == PFB_OK) rc = psfhat_from_psf_t<T>(psf, psfhat_out, nband, P, Q, fr, fc, twP, twQ, st);
    if (twP) (void)hipFree(twP);
    if (twQ) (void)hipFree(twQ);
    return rc;
}

// psfhat on the (P, Q) grid -> psfhat of the SAME image-space PSF on a (P2, Q2) grid, for images
// of (nx, ny) pixels: inverse transform, embed (k_psf_embed), forward transform.  Lets a plan for
// an arbitrary size run on the power-of-two fast path (P2 = 2 nx2 >= 2 nx) with identical results.
template <typename T>
static int psfhat_regrid_t(const void* psfhat, int nband, int nx, int ny, int P, int Q, int P2, int Q2,
                           void* psfhat2, hipStream_t st) {
    const int M = Q / 2, M2 = Q2 / 2;
    FftFactors fr, fc, fr2, fc2;
    PFB_REQUIRE(plan_factors(M, &fr) && plan_factors(P, &fc) && plan_factors(M2, &fr2) && plan_factors(P2, &fc2),
                PFB_ERR_UNSUPPORTED, "psfhat_regrid: a grid length has a prime factor > 13");
    void *twP = nullptr, *twQ = nullptr, *twP2 = nullptr, *twQ2 = nullptr, *spec = nullptr, *psf = nullptr, *psf2 = nullptr;
    int rc = upload_twiddles<T>(P, &twP);
    if (rc == PFB_OK) rc = upload_twiddles<T>(Q, &twQ);
    if (rc == PFB_OK) rc = upload_twiddles<T>(P2, &twP2);
    if (rc == PFB_OK) rc = upload_twiddles<T>(Q2, &twQ2);
    const size_t nspec = (size_t)nband * P * (M + 1), npsf = (size_t)nband * P * Q, npsf2 = (size_t)nband * P2 * Q2;
    if (rc == PFB_OK && (hipMalloc(&spec, nspec * sizeof(cplx<T>)) != hipSuccess ||
                         hipMalloc(&psf, npsf * sizeof(T)) != hipSuccess ||
                         hipMalloc(&psf2, npsf2 * sizeof(T)) != hipSuccess)) {
        set_error("psfhat_regrid: device allocation failed");
        rc = PFB_ERR_ALLOC;
    }
    if (rc == PFB_OK) {
        set_lds_limits<T>(nullptr);
        // every transform picks the one-workgroup-per-line LDS kernel when its line fits, else the global-memory passes
        (void)hipMemcpyAsync(spec, psfhat, nspec * sizeof(cplx<T>), hipMemcpyDeviceToDevice, st);
        if (line_fits_lds<T>(P))
            hipLaunchKernelGGL((k_psfhat_cols<T, true>), dim3(M + 1, nband), dim3(256), 2 * sizeof(cplx<T>) * (size_t)P, st,
                               (cplx<T>*)spec, (const cplx<T>*)twP, P, M + 1, fc);
        else
            rc = long_cols<T, true>((cplx<T>*)spec, (const cplx<T>*)twP, nband, P, M + 1, fc, st);
        if (rc == PFB_OK) {
            if (line_fits_lds<T>(M))
                hipLaunchKernelGGL((k_psf_rows_c2r<T>), dim3(P, nband), dim3(256), 2 * sizeof(cplx<T>) * (size_t)M, st,
                                   (const cplx<T>*)spec, (T*)psf, (const cplx<T>*)twQ, P, Q, fr);
            else
                rc = long_rows_c2r<T>((const cplx<T>*)spec, (T*)psf, (const cplx<T>*)twQ, nband, P, Q, fr, st);
        }
        if (rc == PFB_OK) {
            hipLaunchKernelGGL((k_psf_embed<T>), dim3((Q2 + 255) / 256, P2, nband), dim3(256), 0, st, (const T*)psf,
                               (T*)psf2, nx, ny, P, Q, P2, Q2, (T)(1.0 / ((double)P * (double)Q)));
            if (line_fits_lds<T>(M2))
                hipLaunchKernelGGL((k_psfhat_rows<T>), dim3(P2, nband), dim3(256), 2 * sizeof(cplx<T>) * (size_t)M2, st,
                                   (const T*)psf2, (cplx<T>*)psfhat2, (const cplx<T>*)twQ2, P2, Q2, fr2, 0);
            else
                rc = long_rows_r2c<T>((const T*)psf2, (cplx<T>*)psfhat2, (const cplx<T>*)twQ2, nband, P2, Q2, fr2, 0, st);
        }
        if (rc == PFB_OK) {
            if (line_fits_lds<T>(P2))
                hipLaunchKernelGGL((k_psfhat_cols<T, false>), dim3(M2 + 1, nband), dim3(256), 2 * sizeof(cplx<T>) * (size_t)P2,
                                   st, (cplx<T>*)psfhat2, (const cplx<T>*)twP2, P2, M2 + 1, fc2);
            else
                rc = long_cols<T, false>((cplx<T>*)psfhat2, (const cplx<T>*)twP2, nband, P2, M2 + 1, fc2, st);
        }
        if (rc == PFB_OK && (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
            set_error("psfhat_regrid: kernel launch failed");
            rc = PFB_ERR_HIP;
        }
    }
    for (void* q : {twP, twQ, twP2, twQ2, spec, psf, psf2}) if (q) (void)hipFree(q);
    return rc;
}

}  // namespace pfb

using namespace pfb;

extern "C" {

int pfb_abi_version(void) { return PFB_ABI_VERSION; }
const char* pfb_last_error(void) { return pfb::g_err; }

int pfb_psfconv_plan_create(int nx, int ny, int nx_psf, int ny_psf, int nband, int dtype,
                            pfb_conv_plan** plan) {
    PFB_REQUIRE(plan != nullptr, PFB_ERR_INVALID, "plan_create: null plan pointer");
    *plan = nullptr;
    PFB_REQUIRE(dtype == PFB_F32 || dtype == PFB_F64, PFB_ERR_INVALID, "plan_create: bad dtype %d", dtype);
    PFB_REQUIRE(nx > 0 && ny > 0 && nband > 0, PFB_ERR_INVALID, "plan_create: non-positive size");
    PFB_REQUIRE(nx <= nx_psf && ny <= ny_psf, PFB_ERR_INVALID,
                "plan_create: image (%d,%d) larger than psf grid (%d,%d)", nx, ny, nx_psf, ny_psf);
    PFB_REQUIRE(ny_psf % 2 == 0, PFB_ERR_UNSUPPORTED,
                "plan_create: ny_psf=%d must be even (pfb's grid worker makes it so)", ny_psf);
    PFB_REQUIRE(factorable(nx_psf) && factorable(ny_psf / 2), PFB_ERR_UNSUPPORTED,
                "plan_create: (%d,%d) has a prime factor > 13", nx_psf, ny_psf);
    pfb_conv_plan* p = (pfb_conv_plan*)calloc(1, sizeof(pfb_conv_plan));
    PFB_REQUIRE(p != nullptr, PFB_ERR_ALLOC, "plan_create: host alloc failed");
    p->nx = nx; p->ny = ny; p->P = nx_psf; p->Q = ny_psf; p->M = ny_psf / 2;
    p->nband = nband; p->dtype = dtype;
    const size_t csz = dtype == PFB_F32 ? 8 : 16;
    p->fast = pow2_supported(p) ? 1 : 0;
    if (const char* e = getenv("PFB_FORCE_GENERIC")) { if (atoi(e)) p->fast = 0; }
    int vb = 1;                 // the pow2 kernels are written for VB = 1 (pure transposed T)
    if (!p->fast) { if (const char* e = getenv("PFB_VB")) { int t = atoi(e); if (t >= 1 && t <= 16) vb = t; } }
    p->partials_per_band = p->fast ? nx / pow2_rows_per_wg(p) : nx;
    if (p->fast) {              // 16-byte column blocks per parity class (fftconv_pow2.hip)
        vb = pow2_nvb(p);
        p->nvb = pow2_nblocks(p);
    } else {
        p->nvb = (p->M + 1 + vb - 1) / vb;
    }
    p->VB = vb;
    p->T_elems_per_band = (size_t)p->nvb * nx * vb;
    p->psf_elems_per_band = (size_t)p->nvb * p->P * vb;
    if (!plan_factors(p->M, &p->frow) || !plan_factors(p->P, &p->fcol)) {
        free(p);
        set_error("plan_create: cannot factor (%d,%d)", nx_psf, ny_psf / 2);
        return PFB_ERR_UNSUPPORTED;
    }
    if (!p->fast) {
        // a line that does not fit the two LDS buffers of the line-per-workgroup kernels: every transform of this plan
        // runs as global-memory passes instead (fft_long.hpp) -- slow, but no grid is refused for its size
        const size_t lds_need = 128 + 2 * csz * (size_t)(p->P > p->M ? p->P : p->M);
        if (lds_need > 160 * 1024) p->long_lines = 1;
    }
    int rc = (dtype == PFB_F32) ? upload_twiddles<float>(p->P, &p->twP) : upload_twiddles<double>(p->P, &p->twP);
    if (rc == PFB_OK)
        rc = (dtype == PFB_F32) ? upload_twiddles<float>(p->Q, &p->twQ) : upload_twiddles<double>(p->Q, &p->twQ);
    const size_t tbytes = csz * p->T_elems_per_band * nband;
    const size_t pbytes = csz * p->psf_elems_per_band * nband;
    if (rc == PFB_OK && hipMalloc(&p->T, tbytes) != hipSuccess) rc = PFB_ERR_ALLOC;
    if (rc == PFB_OK && hipMalloc(&p->psf_l, pbytes) != hipSuccess) rc = PFB_ERR_ALLOC;
    if (rc == PFB_OK && hipMalloc((void**)&p->partials, sizeof(double) * 3 * (size_t)nx * nband) != hipSuccess)
        rc = PFB_ERR_ALLOC;
    if (rc == PFB_OK && (hipMalloc((void**)&p->tail_counter, 256) != hipSuccess ||
                         hipMemset(p->tail_counter, 0, 256) != hipSuccess))
        rc = PFB_ERR_ALLOC;
    if (rc == PFB_OK) {
        p->workspace_bytes = tbytes + pbytes;
        rc = (dtype == PFB_F32) ? set_lds_limits<float>(p) : set_lds_limits<double>(p);
        if (p->fast && rc == PFB_OK) rc = pow2_prepare(p);
    }
    if (rc != PFB_OK) {
        if (rc == PFB_ERR_ALLOC) set_error("plan_create: device allocation failed (%zu B)", tbytes + pbytes);
        pfb_psfconv_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return PFB_OK;
}

int pfb_psfconv_plan_destroy(pfb_conv_plan* p) {
    if (!p) return PFB_OK;
    if (p->twP) (void)hipFree(p->twP);
    if (p->twQ) (void)hipFree(p->twQ);
    if (p->psf_l) (void)hipFree(p->psf_l);
    if (p->T) (void)hipFree(p->T);
    if (p->partials) (void)hipFree(p->partials);
    if (p->tail_counter) (void)hipFree(p->tail_counter);
    if (p->long_ws) (void)hipFree(p->long_ws);
    pow2_release(p);
    if (p->prof_ev) {
        for (int k = 0; k < 4 * PROF_MAX; ++k) (void)hipEventDestroy(p->prof_ev[k]);
        free(p->prof_ev);
    }
    if (p->pcg_pin) {
        (void)hipHostFree(p->pcg_pin);
        (void)hipEventDestroy(p->pcg_ev[0]);
        (void)hipEventDestroy(p->pcg_ev[1]);
    }
    free(p);
    return PFB_OK;
}

int pfb_psfconv_plan_info(const pfb_conv_plan* p, int* fast_path, int* vb, size_t* workspace_bytes) {
    PFB_REQUIRE(p != nullptr, PFB_ERR_INVALID, "plan_info: null plan");
    if (fast_path) *fast_path = p->fast;
    if (vb) *vb = p->VB;
    if (workspace_bytes) *workspace_bytes = p->workspace_bytes;
    return PFB_OK;
}

int pfb_psfconv_set_profiling(pfb_conv_plan* p, int on) {
    PFB_REQUIRE(p != nullptr, PFB_ERR_INVALID, "set_profiling: null plan");
    if (on && !p->prof_ev) {
        p->prof_ev = (hipEvent_t*)calloc(4 * PROF_MAX, sizeof(hipEvent_t));
        PFB_REQUIRE(p->prof_ev != nullptr, PFB_ERR_ALLOC, "set_profiling: host alloc failed");
        for (int k = 0; k < 4 * PROF_MAX; ++k) PFB_HIP_CHECK(hipEventCreate(&p->prof_ev[k]));
    }
    p->prof_on = on > 0 ? on : 0;
    p->prof_n = 0;
    p->prof_tick = 0;
    return PFB_OK;
}

int pfb_psfconv_get_profile(pfb_conv_plan* p, double* stage_ms, int* napply) {
    PFB_REQUIRE(p && stage_ms && napply, PFB_ERR_INVALID, "get_profile: null argument");
    stage_ms[0] = stage_ms[1] = stage_ms[2] = 0.0;
    *napply = p->prof_n;
    for (int a = 0; a < p->prof_n; ++a) {
        PFB_HIP_CHECK(hipEventSynchronize(p->prof_ev[a * 4 + 3]));
        for (int k = 0; k < 3; ++k) {
            float ms = 0.f;
            PFB_HIP_CHECK(hipEventElapsedTime(&ms, p->prof_ev[a * 4 + k], p->prof_ev[a * 4 + k + 1]));
            stage_ms[k] += ms;
        }
    }
    p->prof_n = 0;
    return PFB_OK;
}

int pfb_psfconv_set_psfhat(pfb_conv_plan* p, const void* psfhat, void* stream) {
    PFB_REQUIRE(p && psfhat, PFB_ERR_INVALID, "set_psfhat: null argument");
    if (p->fast) {
        int rc = pow2_set_psfhat(p, psfhat, as_stream(stream));
        if (rc == PFB_OK) p->have_psf = 1;
        return rc;
    }
    const ConvDims d = dims_of(p);
    dim3 grid((p->nvb * p->VB + 63) / 64, p->P, p->nband);
    if (p->dtype == PFB_F32)
        hipLaunchKernelGGL((k_relayout_psfhat<float>), grid, dim3(64), 0, as_stream(stream),
                           (const cplx<float>*)psfhat, (cplx<float>*)p->psf_l, d);
    else
        hipLaunchKernelGGL((k_relayout_psfhat<double>), grid, dim3(64), 0, as_stream(stream),
                           (const cplx<double>*)psfhat, (cplx<double>*)p->psf_l, d);
    PFB_HIP_CHECK(hipGetLastError());
    p->have_psf = 1;
    return PFB_OK;
}

int pfb_psfconv_set_psf(pfb_conv_plan* p, const void* psf, void* psfhat_out, void* stream) {
    PFB_REQUIRE(p && psf, PFB_ERR_INVALID, "set_psf: null argument");
    if (p->fast) {             // power-of-two plan: the fast path's own row / column kernels (fftconv_pow2.hip)
        int rc = pow2_set_psf(p, psf, psfhat_out, as_stream(stream));
        if (rc == PFB_OK) p->have_psf = 1;
        return rc;
    }
    const size_t csz = p->dtype == PFB_F32 ? 8 : 16;
    hipStream_t st = as_stream(stream);
    void* tmp = nullptr;
    void* dst = psfhat_out;
    if (!dst) {
        PFB_HIP_CHECK(hipMalloc(&tmp, csz * (size_t)p->nband * p->P * (p->M + 1)));
        dst = tmp;
    }
    int rc = p->dtype == PFB_F32
        ? psfhat_from_psf_t<float>(psf, dst, p->nband, p->P, p->Q, p->frow, p->fcol, p->twP, p->twQ, st)
        : psfhat_from_psf_t<double>(psf, dst, p->nband, p->P, p->Q, p->frow, p->fcol, p->twP, p->twQ, st);
    if (rc == PFB_OK) rc = pfb_psfconv_set_psfhat(p, dst, stream);
    if (tmp) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(tmp);
    }
    return rc;
}

int pfb_psfhat_from_psf(int dtype, const void* psf, int nband, int nx_psf, int ny_psf, void* psfhat, void* stream) {
    PFB_REQUIRE(psf && psfhat && nband > 0 && nx_psf > 0 && ny_psf > 0, PFB_ERR_INVALID, "psfhat_from_psf: bad argument");
    PFB_REQUIRE(dtype == PFB_F32 || dtype == PFB_F64, PFB_ERR_INVALID, "psfhat_from_psf: bad dtype");
    PFB_REQUIRE(ny_psf % 2 == 0, PFB_ERR_UNSUPPORTED, "psfhat_from_psf: ny_psf=%d must be even", ny_psf);
    return dtype == PFB_F32 ? psfhat_from_psf_grid<float>(psf, psfhat, nband, nx_psf, ny_psf, as_stream(stream))
                            : psfhat_from_psf_grid<double>(psf, psfhat, nband, nx_psf, ny_psf, as_stream(stream));
}

int pfb_psfhat_regrid(int dtype, const void* psfhat, int nband, int nx, int ny, int nx_psf, int ny_psf,
                      int nx_psf2, int ny_psf2, void* psfhat2, void* stream) {
    PFB_REQUIRE(psfhat && psfhat2 && nband > 0 && nx > 0 && ny > 0, PFB_ERR_INVALID, "psfhat_regrid: bad argument");
    PFB_REQUIRE(dtype == PFB_F32 || dtype == PFB_F64, PFB_ERR_INVALID, "psfhat_regrid: bad dtype");
    PFB_REQUIRE(ny_psf % 2 == 0 && ny_psf2 % 2 == 0, PFB_ERR_UNSUPPORTED, "psfhat_regrid: odd last axis");
    PFB_REQUIRE(nx_psf2 >= 2 * nx - 1 && ny_psf2 >= 2 * ny - 1, PFB_ERR_INVALID,
                "psfhat_regrid: the new grid (%d,%d) must hold every offset of a (%d,%d) image", nx_psf2, ny_psf2, nx, ny);
    return dtype == PFB_F32
        ? psfhat_regrid_t<float>(psfhat, nband, nx, ny, nx_psf, ny_psf, nx_psf2, ny_psf2, psfhat2, as_stream(stream))
        : psfhat_regrid_t<double>(psfhat, nband, nx, ny, nx_psf, ny_psf, nx_psf2, ny_psf2, psfhat2, as_stream(stream));
}

static int apply_common(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam,
                        double wsum, double sigmainv, void* out, const void* dot_with,
                        const void* dot_with2, double* dots_out, int ndots, void* stream) {
    PFB_REQUIRE(p && x && out, PFB_ERR_INVALID, "apply: null argument");
    PFB_REQUIRE(p->have_psf, PFB_ERR_INVALID, "apply: pfb_psfconv_set_psfhat was never called");
    PFB_REQUIRE(band0 >= 0 && nb > 0 && band0 + nb <= p->nband, PFB_ERR_INVALID,
                "apply: band range [%d,%d) outside plan (nband=%d)", band0, band0 + nb, p->nband);
    PFB_REQUIRE(x != out, PFB_ERR_INVALID, "apply: out must not alias x");
    PFB_REQUIRE(!dot_with || dots_out || ndots == 0, PFB_ERR_INVALID, "apply: dot_with given without an output");
    PFB_REQUIRE(!dot_with2 || dot_with, PFB_ERR_INVALID, "apply: dot_with2 needs dot_with");
    hipStream_t st = as_stream(stream);
    double scale = 1.0 / ((double)p->P * (double)p->Q);
    if (wsum > 0) scale /= wsum;
    int rc;
    p->last_npartials = p->partials_per_band * nb;        // the persistent row-inverse kernel lowers it
    if (p->fast)
        rc = pow2_apply(p, band0, nb, x, beam, scale, sigmainv, out, dot_with, dot_with2, st);
    else if (p->long_lines) {
        const size_t need = p->dtype == PFB_F32 ? long_apply_ws_bytes<float>(nb, p->nx, p->M, p->P)
                                                : long_apply_ws_bytes<double>(nb, p->nx, p->M, p->P);
        if (need > p->long_ws_bytes) {
            PFB_HIP_CHECK(hipStreamSynchronize(st));
            if (p->long_ws) (void)hipFree(p->long_ws);
            p->long_ws = nullptr; p->long_ws_bytes = 0;
            if (hipMalloc(&p->long_ws, need) != hipSuccess) {
                set_error("apply: workspace allocation of %zu B for the long-line path failed", need);
                return PFB_ERR_ALLOC;
            }
            p->long_ws_bytes = need;
        }
        prof_mark(p, st, 0); prof_mark(p, st, 1); prof_mark(p, st, 2);
        if (p->dtype == PFB_F32)
            rc = apply_long<float>((cplx<float>*)p->long_ws, (cplx<float>*)p->T, (const cplx<float>*)p->psf_l, p->partials,
                                   (const cplx<float>*)p->twP, (const cplx<float>*)p->twQ, p->frow, p->fcol, p->nx, p->ny,
                                   p->P, p->M, p->T_elems_per_band, p->psf_elems_per_band, band0, nb, (const float*)x,
                                   (const float*)beam, scale, sigmainv, (float*)out, (const float*)dot_with,
                                   (const float*)dot_with2, st);
        else
            rc = apply_long<double>((cplx<double>*)p->long_ws, (cplx<double>*)p->T, (const cplx<double>*)p->psf_l,
                                    p->partials, (const cplx<double>*)p->twP, (const cplx<double>*)p->twQ, p->frow, p->fcol,
                                    p->nx, p->ny, p->P, p->M, p->T_elems_per_band, p->psf_elems_per_band, band0, nb,
                                    (const double*)x, (const double*)beam, scale, sigmainv, (double*)out,
                                    (const double*)dot_with, (const double*)dot_with2, st);
        prof_mark(p, st, 3);
    } else if (p->dtype == PFB_F32)
        rc = apply_generic<float>(p, band0, nb, x, beam, scale, sigmainv, out, dot_with, dot_with2, st);
    else
        rc = apply_generic<double>(p, band0, nb, x, beam, scale, sigmainv, out, dot_with, dot_with2, st);
    if (rc != PFB_OK) return rc;
    if (dot_with && ndots > 0) {           // ndots == 0: the caller sums p->partials itself (PCG driver)
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, p->partials,
                           p->last_npartials, ndots, dots_out);
        PFB_HIP_CHECK(hipGetLastError());
    }
    return PFB_OK;
}

int pfb_psfconv_apply(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam,
                      double wsum, double sigmainv, void* out, const void* dot_with,
                      double* dot_out, void* stream) {
    return apply_common(p, band0, nb, x, beam, wsum, sigmainv, out, dot_with, nullptr, dot_out, 1, stream);
}

int pfb_psfconv_apply_dots(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam,
                           double wsum, double sigmainv, void* out, const void* dot_with,
                           const void* dot_with2, double* dots_out, void* stream) {
    PFB_REQUIRE(dot_with && dots_out, PFB_ERR_INVALID, "apply_dots: dot_with and dots_out are required");
    return apply_common(p, band0, nb, x, beam, wsum, sigmainv, out, dot_with, dot_with2, dots_out, 3, stream);
}

}  // extern "C"

// internal (cgvec.hip): the convolution with its three fused dots left as per-workgroup partials in
// plan->partials ([3][plan->last_npartials]) -- the PCG driver sums them in the same launch that does
// its per-iteration scalar bookkeeping
int pfb::psfconv_apply_partials(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam,
                                double wsum, double sigmainv, void* out, const void* dot_with,
                                const void* dot_with2, void* stream) {
    PFB_REQUIRE(dot_with, PFB_ERR_INVALID, "apply_partials: dot_with is required");
    return apply_common(p, band0, nb, x, beam, wsum, sigmainv, out, dot_with, dot_with2, nullptr, 0, stream);
}
