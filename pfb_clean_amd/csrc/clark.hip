// clark.hip -- the sub-minor loop of the Clark CLEAN (pfb/deconv/clark.py:29-84), the only part of
// klean's minor cycle that is not an image-wide elementwise pass or the PSF convolution.
//
// The reference is a sequential greedy loop on the "active set" (pixels above the sub-minor
// threshold): pick the pixel with the largest |sum over bands|, add gamma * component / wsum to
// the model, subtract component * PSF(shift) from every active pixel, repeat (<= 1000 times).
// Every iteration depends on the previous arg-max, so it is latency, not bandwidth: ONE resident
// 1024-thread workgroup keeps the loop on the device -- one pass over the active set per
// iteration (subtract and next search fused), two barriers, no launches.
#include "common.hpp"

namespace pfb {

constexpr int CLK_T = 1024;
constexpr int CLK_MAXBAND = 64;

template <typename T>
__global__ void __launch_bounds__(CLK_T)
k_clark_subminor(T* __restrict__ A, size_t nact, int nband, const T* __restrict__ psf, int P, int Q,
                 const int* __restrict__ Ip, const int* __restrict__ Iq, T* __restrict__ model, int nx, int ny,
                 const T* __restrict__ wsums, T gamma, T th, int maxit, int* __restrict__ iters_out) {
    __shared__ T s_val[CLK_T / 64];
    __shared__ long long s_idx[CLK_T / 64];
    __shared__ T xh[CLK_MAXBAND];
    __shared__ int s_p, s_q, s_go;
    const int nxo2 = P / 2, nyo2 = Q / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int p = 0, q = 0, k = 0;
    bool have_comp = false;             // iteration 0 only searches
    for (;;) {
        // ---- (subtract the component chosen last round and) search: max of (sum_b A[b,i])^2, first index wins
        T best = T(-1);                  // NaN-safe: comparisons with NaN are false, index 0 stays valid
        long long besti = 0;
        for (size_t i = tid; i < nact; i += CLK_T) {
            T s = 0;
            if (have_comp) {
                const int pp = nxo2 - (p - Ip[i]), qq = nyo2 - (q - Iq[i]);
                const bool in = pp >= 0 && pp < P && qq >= 0 && qq < Q;      // always true for valid Ip, Iq
                for (int b = 0; b < nband; ++b) {
                    T a = A[(size_t)b * nact + i];
                    if (in) a -= xh[b] * psf[((size_t)b * P + pp) * Q + qq];
                    A[(size_t)b * nact + i] = a;
                    s += a;
                }
            } else {
                for (int b = 0; b < nband; ++b) s += A[(size_t)b * nact + i];
            }
            const T v = s * s;
            if (v > best) { best = v; besti = (long long)i; }       // ascending i per thread: first max kept
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const T ov = __shfl_down(best, off, 64);
            const long long oi = __shfl_down(besti, off, 64);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            T bv = s_val[0];
            long long bi = s_idx[0];
            for (int w = 1; w < CLK_T / 64; ++w)
                if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
            const T amax = sqrt(bv);         // NaN (all-NaN input or empty maxima) compares false: stop
            int go = (amax > th && k < maxit && bi >= 0 && (size_t)bi < nact) ? 1 : 0;
            int pn = 0, qn = 0;
            if (go) {
                pn = Ip[bi]; qn = Iq[bi];
                if (pn < 0 || pn >= nx || qn < 0 || qn >= ny) go = 0;        // corrupt index arrays: stop, never write
            }
            s_go = go;
            if (go) {
                s_p = pn; s_q = qn;
                for (int b = 0; b < nband; ++b) {
                    const T c = A[(size_t)b * nact + (size_t)bi];
                    xh[b] = c;
                    if (wsums[b] > T(0)) model[((size_t)b * nx + pn) * ny + qn] += gamma * c / wsums[b];
                }
            }
        }
        __syncthreads();
        if (!s_go) break;
        p = s_p; q = s_q;
        have_comp = true;
        ++k;
    }
    if (tid == 0 && iters_out) *iters_out = k;
}

// ----------------------------------------------------------------- Hogbom CLEAN
// pfb/deconv/hogbom.py:8-74: every iteration subtracts gamma * component * PSF(shifted window) from the
// WHOLE residual cube and searches the new peak of (sum_b IR)^2.  Two launches per iteration, loop state
// (peak position / value, k, done flag) on the device so that the host only looks once per batch:
//   k_hogbom_step   : [apply the component chosen last round] + per-workgroup arg-max partials
//   k_hogbom_select : final arg-max (first index wins), stop test, model update, next component
struct HogState { long long pq; int p, q, k, done; double irmax; };

template <typename T>
__global__ void __launch_bounds__(256)
k_hogbom_step(T* __restrict__ IR, const T* __restrict__ psf, int nband, int nx, int ny, int P, int Q,
              const T* __restrict__ comp /* gamma * xhat, nband */, const HogState* __restrict__ stt, int apply,
              T* __restrict__ pval, long long* __restrict__ pidx) {
    __shared__ T sv[4];
    __shared__ long long si[4];
    const HogState st = *stt;
    T best = T(-1);
    long long besti = 0;
    if (!st.done) {
        const size_t npix = (size_t)nx * ny;
        const int nx0 = P / 2, ny0 = Q / 2;
        for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < npix; e += (size_t)gridDim.x * blockDim.x) {
            const int i = (int)(e / ny), j = (int)(e - (size_t)i * ny);
            T s = 0;
            for (int b = 0; b < nband; ++b) {
                T v = IR[(size_t)b * npix + e];
                if (apply) {
                    v -= comp[b] * psf[((size_t)b * P + (nx0 - st.p + i)) * Q + (ny0 - st.q + j)];
                    IR[(size_t)b * npix + e] = v;
                }
                s += v;
            }
            const T val = s * s;
            if (val > best) { best = val; besti = (long long)e; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T ov = __shfl_down(best, off, 64);
        const long long oi = __shfl_down(besti, off, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < besti)) { best = sv[w]; besti = si[w]; }
        pval[blockIdx.x] = best;
        pidx[blockIdx.x] = besti;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_hogbom_select(const T* __restrict__ pval, const long long* __restrict__ pidx, int nparts, const T* __restrict__ IR,
                T* __restrict__ model, const T* __restrict__ wsums, T* __restrict__ comp, int nband, int nx, int ny,
                T gamma, double pf, double threshold, int maxit, int first, HogState* __restrict__ stt, double* tol_io) {
    __shared__ T sv[4];
    __shared__ long long si[4];
    if (stt->done) return;
    T best = T(-1);
    long long besti = 0;
    for (int g = threadIdx.x; g < nparts; g += blockDim.x)
        if (pval[g] > best || (pval[g] == best && pidx[g] < besti)) { best = pval[g]; besti = pidx[g]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T ov = __shfl_down(best, off, 64);
        const long long oi = __shfl_down(besti, off, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < besti)) { best = sv[w]; besti = si[w]; }
        HogState st = *stt;
        const double irmax = sqrt((double)best);
        if (first) *tol_io = fmax(pf * irmax, threshold);         // hogbom.py:30
        else st.k += 1;                                           // the component applied by the last step
        st.irmax = irmax;
        const size_t npix = (size_t)nx * ny;
        const bool ok = besti >= 0 && (size_t)besti < npix;
        if (!(irmax > *tol_io) || st.k >= maxit || !ok) {
            st.done = 1;
        } else {
            st.pq = besti;
            st.p = (int)(besti / ny);
            st.q = (int)(besti - (long long)st.p * ny);
            for (int b = 0; b < nband; ++b) {                     // xhat = IR[:, p, q] / wsums ; x += gamma xhat
                const T xh = IR[(size_t)b * npix + (size_t)besti] / wsums[b];
                model[(size_t)b * npix + (size_t)besti] += gamma * xh;
                comp[b] = gamma * xh;
            }
        }
        *stt = st;
    }
}

// ----------------------------------------------------------------- band coupling
// freqmul (pfb/utils/misc.py:1366-1375): out[k, i, j] = sum_l A[k, l] x[l, i, j] -- the nband x nband
// mixing of the fwdbwd parametrisations.  One thread per pixel: nband reads, nband writes, the
// small matrix (<= 64 x 64) from LDS; HBM bound, no MFMA at these sizes.
constexpr int FM_MAXBAND = 64;
template <typename T>
__global__ void __launch_bounds__(256)
k_freqmul(const T* __restrict__ A, const T* __restrict__ x, T* __restrict__ out, int nband, size_t npix,
          const T* __restrict__ pre, const T* __restrict__ post) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);
    for (int k = threadIdx.x; k < nband * nband; k += blockDim.x) As[k] = A[k];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        T xv[FM_MAXBAND];
#pragma unroll 1
        for (int l = 0; l < nband; ++l) xv[l] = pre ? x[(size_t)l * npix + i] * pre[(size_t)l * npix + i] : x[(size_t)l * npix + i];
#pragma unroll 1
        for (int k = 0; k < nband; ++k) {
            T s = 0;
            for (int l = 0; l < nband; ++l) s += As[k * nband + l] * xv[l];       // same order as the reference loop
            out[(size_t)k * npix + i] = post ? s * post[(size_t)k * npix + i] : s;
        }
    }
}

}  // namespace pfb

using namespace pfb;

extern "C" int pfb_hogbom(int dtype, void* IR, const void* psf, void* model, const void* wsums, int nband, int nx,
                          int ny, int nx_psf, int ny_psf, double gamma, double pf, double threshold, int maxit,
                          void* work, size_t work_bytes, int* k_out, double* irmax_out, void* stream) {
    PFB_REQUIRE(IR && psf && model && wsums && work, PFB_ERR_INVALID, "hogbom: null argument");
    PFB_REQUIRE(dtype == PFB_F32 || dtype == PFB_F64, PFB_ERR_INVALID, "hogbom: bad dtype");
    PFB_REQUIRE(nband >= 1 && nband <= 64 && nx > 0 && ny > 0, PFB_ERR_INVALID, "hogbom: bad shape");
    PFB_REQUIRE(nx_psf / 2 >= nx - 1 && ny_psf / 2 >= ny - 1 && nx_psf / 2 + nx - 1 < nx_psf && ny_psf / 2 + ny - 1 < ny_psf,
                PFB_ERR_UNSUPPORTED, "hogbom: the PSF (%d,%d) must cover every shift of the (%d,%d) image", nx_psf,
                ny_psf, nx, ny);
    const int G = 1024;
    const size_t esz = dtype == PFB_F32 ? 4 : 8;
    const size_t need = 256 + 64 * 8 + (size_t)G * (8 + 8);
    PFB_REQUIRE(work_bytes >= need, PFB_ERR_INVALID, "hogbom: work buffer too small (%zu < %zu)", work_bytes, need);
    hipStream_t st = as_stream(stream);
    char* w = (char*)work;
    HogState* stt = (HogState*)w;               // [0, 64)
    double* tol = (double*)(w + 64);            // [64, 72)
    void* comp = w + 256;                       // 64 values
    void* pval = w + 256 + 64 * 8;
    long long* pidx = (long long*)(w + 256 + 64 * 8 + (size_t)G * 8);
    PFB_HIP_CHECK(hipMemsetAsync(work, 0, 256 + 64 * 8, st));
    HogState h;
    (void)esz;
    int launched = 0;
    bool first = true;
    for (;;) {
        const int batch = 64;
        for (int it = 0; it < batch; ++it) {
            if (dtype == PFB_F32) {
                hipLaunchKernelGGL((k_hogbom_step<float>), dim3(G), dim3(256), 0, st, (float*)IR, (const float*)psf, nband,
                                   nx, ny, nx_psf, ny_psf, (const float*)comp, stt, first ? 0 : 1, (float*)pval, pidx);
                hipLaunchKernelGGL((k_hogbom_select<float>), dim3(1), dim3(256), 0, st, (const float*)pval, pidx, G,
                                   (const float*)IR, (float*)model, (const float*)wsums, (float*)comp, nband, nx, ny,
                                   (float)gamma, pf, threshold, maxit, first ? 1 : 0, stt, tol);
            } else {
                hipLaunchKernelGGL((k_hogbom_step<double>), dim3(G), dim3(256), 0, st, (double*)IR, (const double*)psf,
                                   nband, nx, ny, nx_psf, ny_psf, (const double*)comp, stt, first ? 0 : 1, (double*)pval, pidx);
                hipLaunchKernelGGL((k_hogbom_select<double>), dim3(1), dim3(256), 0, st, (const double*)pval, pidx, G,
                                   (const double*)IR, (double*)model, (const double*)wsums, (double*)comp, nband, nx, ny,
                                   gamma, pf, threshold, maxit, first ? 1 : 0, stt, tol);
            }
            first = false;
            ++launched;
        }
        PFB_HIP_CHECK(hipMemcpyAsync(&h, stt, sizeof(h), hipMemcpyDeviceToHost, st));
        PFB_HIP_CHECK(hipStreamSynchronize(st));
        if (h.done || launched > maxit + 2) break;
    }
    if (k_out) *k_out = h.k;
    if (irmax_out) *irmax_out = h.irmax;
    return PFB_OK;
}

extern "C" int pfb_freqmul(int dtype, const void* A, const void* x, void* out, int nband, size_t npix,
                           const void* pre, const void* post, void* stream) {
    PFB_REQUIRE(A && x && out && x != out, PFB_ERR_INVALID, "freqmul: bad argument (out must not alias x)");
    PFB_REQUIRE(dtype == PFB_F32 || dtype == PFB_F64, PFB_ERR_INVALID, "freqmul: bad dtype");
    PFB_REQUIRE(nband >= 1 && nband <= FM_MAXBAND, PFB_ERR_UNSUPPORTED, "freqmul: nband %d > %d", nband, FM_MAXBAND);
    hipStream_t st = as_stream(stream);
    size_t g = (npix + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    const size_t esz = dtype == PFB_F32 ? 4 : 8;
    if (dtype == PFB_F32)
        hipLaunchKernelGGL((k_freqmul<float>), dim3((unsigned)g), dim3(256), esz * nband * nband, st, (const float*)A,
                           (const float*)x, (float*)out, nband, npix, (const float*)pre, (const float*)post);
    else
        hipLaunchKernelGGL((k_freqmul<double>), dim3((unsigned)g), dim3(256), esz * nband * nband, st, (const double*)A,
                           (const double*)x, (double*)out, nband, npix, (const double*)pre, (const double*)post);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

extern "C" int pfb_clark_subminor(int dtype, void* A, size_t nact, int nband, const void* psf, int nx_psf, int ny_psf,
                                  const int* Ip, const int* Iq, void* model, int nx, int ny, const void* wsums,
                                  double gamma, double th, int maxit, int* iters_out, void* stream) {
    PFB_REQUIRE(A && psf && Ip && Iq && model && wsums, PFB_ERR_INVALID, "clark_subminor: null argument");
    PFB_REQUIRE(dtype == PFB_F32 || dtype == PFB_F64, PFB_ERR_INVALID, "clark_subminor: bad dtype");
    PFB_REQUIRE(nband >= 1 && nband <= CLK_MAXBAND, PFB_ERR_UNSUPPORTED, "clark_subminor: nband %d > %d", nband, CLK_MAXBAND);
    PFB_REQUIRE(nact >= 1, PFB_ERR_INVALID, "clark_subminor: empty active set");
    // every offset p - Ip[i] must index the PSF: the reference masks |p - Ip| <= nx_psf/2 (and then
    // mis-indexes its shrunken active set, clark.py:68-75); with nx_psf >= 2 nx - 1 the mask is all true
    PFB_REQUIRE(nx_psf / 2 >= nx - 1 && ny_psf / 2 >= ny - 1 && nx_psf / 2 + nx - 1 < nx_psf && ny_psf / 2 + ny - 1 < ny_psf,
                PFB_ERR_UNSUPPORTED, "clark_subminor: the PSF (%d,%d) must cover every offset of the (%d,%d) image", nx_psf,
                ny_psf, nx, ny);
    hipStream_t st = as_stream(stream);
    if (dtype == PFB_F32)
        hipLaunchKernelGGL((k_clark_subminor<float>), dim3(1), dim3(CLK_T), 0, st, (float*)A, nact, nband,
                           (const float*)psf, nx_psf, ny_psf, Ip, Iq, (float*)model, nx, ny, (const float*)wsums,
                           (float)gamma, (float)th, maxit, iters_out);
    else
        hipLaunchKernelGGL((k_clark_subminor<double>), dim3(1), dim3(CLK_T), 0, st, (double*)A, nact, nband,
                           (const double*)psf, nx_psf, ny_psf, Ip, Iq, (double*)model, nx, ny, (const double*)wsums,
                           gamma, th, maxit, iters_out);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}
