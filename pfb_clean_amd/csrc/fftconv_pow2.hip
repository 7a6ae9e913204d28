// fftconv_pow2.hip -- the fast PSF-convolution path: power-of-two images with the
// standard 2x PSF oversampling (nx_psf = 2 nx, ny_psf = 2 ny; pfb's default
// psf-oversize 2.0, parser/grid.yaml:65-67).  All BASELINE configs take this path.
//
// Pruned transforms.  A length-2H FFT whose input is zero beyond H splits exactly into
//     A_e = FFT_H(a)            (even output bins)
//     A_o = FFT_H(a .* w_2H^n)  (odd  output bins)
// and the inverse restricted to the first H samples is
//     b = IFFT_H(B_e) + conj(w_2H^n) .* IFFT_H(B_o)
// so neither the zero half of the padded image nor the discarded half of the result
// is ever computed, loaded or stored.  The same identity is used on the packed-real
// row transform (z[n] = x[2n] + i x[2n+1]).
//
// Kernels (VB = 1 layout: T[band][v][i], i contiguous; psf_l[band][v][parity][m]):
//   k_row_fwd_pow2  G image rows per workgroup -> X[v], written as G*8-byte pieces
//                   (LDS-transposed) into T[v][i0 .. i0+G)
//   k_col_pow2      one frequency column per thread group, contiguous 8-byte/lane
//                   streams: a -> FFT -> *psf_e -> IFFT, a*w -> FFT -> *psf_o -> IFFT,
//                   combine, store in place
//   k_row_inv_pow2  G output rows per workgroup, mirror image of row_fwd with the fused
//                   epilogue (1/(PQ wsum), beam, + sigmainv x, <dot_with, out> partials)
#include "conv_plan.hpp"
#include "fft_pow2.hpp"
#include <vector>

namespace pfb {

// elements per thread: the column kernel favours occupancy (small register arrays, it
// is the HBM-streaming kernel), the row kernels favour few threads per row so that 8
// rows (64-byte transposed pieces) fit one 1024-thread workgroup.
template <typename T> struct FastCfg;
template <> struct FastCfg<float>  { static constexpr int ECOL = 8; static constexpr int EROW = 16; static constexpr int WCOL = 4; };
template <> struct FastCfg<double> { static constexpr int ECOL = 8; static constexpr int EROW = 8;  static constexpr int WCOL = 2; };

constexpr int LDS_BUDGET = 152 * 1024;

// rows per workgroup for the row kernels
template <typename T, int L, int E>
constexpr int row_groups() {
    constexpr int TPB = L / E;
    int G = 256 / TPB > 8 ? 256 / TPB : 8;
    const int stride = (L + L / 16 + 4);
    while (G > 2 && (G * TPB > 1024 || (size_t)G * stride * sizeof(cplx<T>) + 64 > LDS_BUDGET)) G /= 2;
    return G;
}
template <int H, int E>
constexpr int col_groups() { return (H / E) >= 256 ? 1 : 256 / (H / E); }

// two adjacent complex values as one aligned access (16 B for fp32, 2 x 16 B for fp64)
template <typename T> struct alignas(2 * sizeof(cplx<T>) > 16 ? 16 : 2 * sizeof(cplx<T>)) Pair2 { cplx<T> a, b; };
template <typename T>
__device__ __forceinline__ void store2(cplx<T>* dst, cplx<T> a, cplx<T> b) {
    Pair2<T> p; p.a = a; p.b = b;
    *reinterpret_cast<Pair2<T>*>(dst) = p;
}
template <typename T>
__device__ __forceinline__ Pair2<T> load2(const cplx<T>* src) {
    return *reinterpret_cast<const Pair2<T>*>(src);
}

// Make a pointer opaque to the optimiser: loads through the result cannot be CSE'd with /
// hoisted above earlier loads of the same address (used where a value is deliberately
// RE-READ from L2 instead of being kept live in 32 VGPRs across an FFT).
template <typename P>
__device__ __forceinline__ P* opaque(P* p) {
    asm volatile("" : "+v"(p));
    return p;
}

struct FastDims {
    int nx, ny, M;              // M = ny (packed length), nv = M + 1 columns
    size_t T_band, psf_band;
};

// ---------------------------------------------------------------- psfhat re-layout
// psf_l[band][v][par][m] = psfhat[band][2m + par][v]      (P = 2H rows, v <= M)
template <typename T>
__global__ void k_relayout_psf_pow2(const cplx<T>* __restrict__ psfhat, cplx<T>* __restrict__ psf_l,
                                    int P, int nv, size_t psf_band) {
    // tile transpose through LDS: block handles 32 u x 32 v
    __shared__ cplx<T> tile[32][33];
    const int band = blockIdx.z;
    const int u0 = blockIdx.y * 32, v0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: ty < 8
    for (int r = ty; r < 32; r += 8) {
        const int u = u0 + r, v = v0 + tx;
        if (u < P && v < nv) tile[r][tx] = psfhat[((size_t)band * P + u) * nv + v];
    }
    __syncthreads();
    const int H = P / 2;
    for (int r = ty; r < 32; r += 8) {
        const int v = v0 + r, u = u0 + tx;
        if (u < P && v < nv)
            psf_l[(size_t)band * psf_band + (size_t)v * P + (size_t)(u & 1) * H + (u >> 1)] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------ column
template <typename T, int H, int E>
__global__ void __launch_bounds__((col_groups<H, E>() * (H / E)), FastCfg<T>::WCOL)
k_col_pow2(cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ psf_l,
           const cplx<T>* __restrict__ twP, const cplx<T>* __restrict__ ptw,
           int nv, size_t T_band, size_t psf_band, int band0) {
    using F = RegFft<T, H, E>;
    constexpr int TPB = F::TPB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* lds = reinterpret_cast<cplx<T>*>(smem) + (size_t)g * F::LDS_ELEMS;
    const int v = blockIdx.x * col_groups<H, E>() + g;
    const int band = band0 + blockIdx.y;
    const bool active = v < nv;
    cplx<T>* col = Tw + (size_t)band * T_band + (size_t)(active ? v : 0) * H;
    const cplx<T>* pe = psf_l + (size_t)band * psf_band + (size_t)(active ? v : 0) * (2 * H);
    const cplx<T>* po = pe + H;

    cplx<T> vv[E], ev[E];
#pragma unroll
    for (int j = 0; j < E; ++j) vv[j] = active ? col[t + TPB * j] : cplx<T>(0, 0);
    // ---- even bins
    F::template run<false>(vv, lds, t, ptw);
#pragma unroll
    for (int j = 0; j < E; ++j) vv[j] = vv[j] * pe[t + TPB * j];
    F::template run<true>(vv, lds, t, ptw);
#pragma unroll
    for (int j = 0; j < E; ++j) ev[j] = vv[j];
    // ---- odd bins: a .* w_P^n  (a and w re-read: L2 hits, saves 64 live VGPRs)
    {
        const cplx<T>* col2 = opaque(col + t);
        const cplx<T>* tw2 = opaque(twP + t);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const cplx<T> a = active ? col2[TPB * j] : cplx<T>(0, 0);
            vv[j] = a * tw2[TPB * j];
        }
    }
    F::template run<false>(vv, lds, t, ptw);
#pragma unroll
    for (int j = 0; j < E; ++j) vv[j] = vv[j] * po[t + TPB * j];
    F::template run<true>(vv, lds, t, ptw);
    if (active) {
        const cplx<T>* tw3 = opaque(twP + t);
#pragma unroll
        for (int j = 0; j < E; ++j) col[t + TPB * j] = ev[j] + mulc(vv[j], tw3[TPB * j]);
    }
}

// ------------------------------------------------------------------- row forward
// one parity (even / odd output bins) of the forward row transform for G rows
template <typename T, int L, int E, int PAR>
__device__ __forceinline__ void row_fwd_phase(const typename vec2<T>::type* xr_in,
                                              const typename vec2<T>::type* br_in,
                                              const cplx<T>* __restrict__ twQ,
                                              const cplx<T>* __restrict__ twM,
                                              const cplx<T>* __restrict__ ptw, cplx<T>* lds0,
                                              cplx<T>* lds, cplx<T>* __restrict__ Tb, int nx, int i0,
                                              int t) {
    using F = RegFft<T, L, E>;
    constexpr int TPB = F::TPB;
    constexpr int G = row_groups<T, L, E>();
    constexpr int NT = G * TPB;
    constexpr int STRIDE = F::LDS_ELEMS + 4;
    constexpr int HP = G / 2;
    constexpr int MSTEP = NT / HP;
    using V2 = typename vec2<T>::type;
    const int rp = threadIdx.x % HP, mi = threadIdx.x / HP;
    const cplx<T>* r0 = lds0 + (size_t)(2 * rp) * STRIDE;
    const cplx<T>* r1 = r0 + STRIDE;
    cplx<T> vv[E];
    {
        // z[n] = x[2n] + i x[2n+1]  (re-read per parity: an L2 hit that keeps the
        // kernel within 128 VGPRs at 1024 threads)
        const V2* xr = opaque(xr_in + t);
        const V2* br = br_in ? opaque(br_in + t) : nullptr;
        const cplx<T>* tm = twM + t;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            V2 a = xr[TPB * j];
            if (br) { V2 b = br[TPB * j]; a.x *= b.x; a.y *= b.y; }
            const cplx<T> zz(a.x, a.y);
            vv[j] = PAR ? zz * tm[TPB * j] : zz;
        }
    }
    F::template run<false>(vv, lds, t, ptw);
    __syncthreads();                             // everyone finished the last exchange read
    {
        cplx<T>* wp = lds + F::pad(t);
#pragma unroll
        for (int j = 0; j < E; ++j) wp[F::cpad(TPB * j)] = vv[j];
    }
    __syncthreads();
    // X[v] = 1/2 [ (Z[v] + conj Z[M-v]) - i w_Q^v (Z[v] - conj Z[M-v]) ]
    //   even v = 2m   : Z -> Ze[m mod L], Ze[(L-m) mod L]      m = 0..L
    //   odd  v = 2m+1 : Z -> Zo[m],       Zo[L-1-m]            m = 0..L-1
    constexpr int mend = PAR ? L : L + 1;
    for (int m = mi; m < mend; m += MSTEP) {
        const int ia = PAR ? m : (m == L ? 0 : m);
        const int ib = PAR ? (L - 1 - m) : (m == 0 ? 0 : L - m);
        const int v = 2 * m + PAR;
        const cplx<T> w = twQ[v];
        cplx<T> o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const cplx<T>* rr = h ? r1 : r0;
            const cplx<T> zv = rr[F::pad(ia)];
            const cplx<T> zm = conj(rr[F::pad(ib)]);
            o[h] = T(0.5) * ((zv + zm) + mul_mi(w * (zv - zm)));
        }
        // rows (i0 + 2rp, i0 + 2rp + 1) of column v: 16 contiguous bytes (fp32)
        store2<T>(Tb + (size_t)v * nx + i0 + 2 * rp, o[0], o[1]);
    }
    // the next FFT's first exchange starts with a barrier, protecting these LDS reads
}

template <typename T, int L, int E>
__global__ void __launch_bounds__((row_groups<T, L, E>() * (L / E)))
k_row_fwd_pow2(const T* __restrict__ x, const T* __restrict__ beam, cplx<T>* __restrict__ Tw,
               const cplx<T>* __restrict__ twQ, const cplx<T>* __restrict__ twM,
               const cplx<T>* __restrict__ ptw, FastDims d, int band0) {
    using F = RegFft<T, L, E>;
    constexpr int TPB = F::TPB;
    constexpr int G = row_groups<T, L, E>();
    constexpr int NT = G * TPB;
    constexpr int STRIDE = F::LDS_ELEMS + 4;
    constexpr int HP = G / 2;                    // row pairs
    using V2 = typename vec2<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T>* lds0 = reinterpret_cast<cplx<T>*>(smem);
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* lds = lds0 + (size_t)g * STRIDE;
    const int i0 = blockIdx.x * G;
    const int bl = blockIdx.y, band = band0 + bl;
    const size_t rowoff = ((size_t)bl * d.nx + (i0 + g)) * d.ny;
    const V2* xr = reinterpret_cast<const V2*>(x + rowoff);
    const V2* br = beam ? reinterpret_cast<const V2*>(beam + rowoff) : nullptr;

    cplx<T>* Tb = Tw + (size_t)band * d.T_band;
    row_fwd_phase<T, L, E, 0>(xr, br, twQ, twM, ptw, lds0, lds, Tb, d.nx, i0, t);
    row_fwd_phase<T, L, E, 1>(xr, br, twQ, twM, ptw, lds0, lds, Tb, d.nx, i0, t);
}

// ------------------------------------------------------------------- row inverse
// one parity of the inverse row transform: gathers Y[2m + PAR][i0 .. i0+G) (G*8-byte
// pieces), LDS-transposes them to per-row order, builds the packed spectrum and runs
// the inverse FFT; result in vv (register j <-> sample t + TPB j of IFFT_L)
template <typename T, int L, int E, int PAR>
__device__ __forceinline__ void row_inv_phase(const cplx<T>* __restrict__ Tb,
                                              const cplx<T>* __restrict__ twQ,
                                              const cplx<T>* __restrict__ ptw, cplx<T>* lds0,
                                              cplx<T>* lds, int nx, int i0, int t,
                                              cplx<T> (&vv)[E]) {
    using F = RegFft<T, L, E>;
    constexpr int TPB = F::TPB;
    constexpr int G = row_groups<T, L, E>();
    constexpr int NT = G * TPB;
    constexpr int STRIDE = F::LDS_ELEMS + 4;
    constexpr int HP = G / 2;
    constexpr int MSTEP = NT / HP;
    const int rp = threadIdx.x % HP, mi = threadIdx.x / HP;
    cplx<T>* w0 = lds0 + (size_t)(2 * rp) * STRIDE;
    cplx<T>* w1 = w0 + STRIDE;
    __syncthreads();                            // previous phase done with the LDS
    constexpr int mend = PAR ? L : L + 1;
    for (int m = mi; m < mend; m += MSTEP) {
        const Pair2<T> y2 = load2<T>(Tb + (size_t)(2 * m + PAR) * nx + i0 + 2 * rp);
        w0[F::pad(m)] = y2.a;
        w1[F::pad(m)] = y2.b;
    }
    __syncthreads();
    // Z[v] = (Y[v] + conj Y[M-v]) + i conj(w_Q^v) (Y[v] - conj Y[M-v]),  v = 2m + PAR
    //   even: Y[2m], Y[M-2m] = Yl[m], Yl[L-m]   (m = 0: DC and Nyquist, imag ignored)
    //   odd : Y[2m+1], Y[M-2m-1] = Yl[m], Yl[L-1-m]
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int m = t + TPB * j;
        cplx<T> yv = lds[F::pad(m)];
        cplx<T> ym = lds[F::pad(PAR ? (L - 1 - m) : (L - m))];
        if (!PAR && m == 0) { yv.y = 0; ym.y = 0; }
        ym = conj(ym);
        const cplx<T> w = twQ[2 * m + PAR];
        vv[j] = (yv + ym) + mul_i(mulc(yv - ym, w));
    }
    F::template run<true>(vv, lds, t, ptw);
}

template <typename T, int L, int E>
__global__ void __launch_bounds__((row_groups<T, L, E>() * (L / E)))
k_row_inv_pow2(const cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ twQ,
               const cplx<T>* __restrict__ twM, const cplx<T>* __restrict__ ptw,
               const T* __restrict__ x, const T* __restrict__ beam,
               const T* __restrict__ dot_with, T* __restrict__ out,
               double* __restrict__ partials, FastDims d, int band0, T scale, T sigmainv) {
    using F = RegFft<T, L, E>;
    constexpr int TPB = F::TPB;
    constexpr int G = row_groups<T, L, E>();
    constexpr int NT = G * TPB;
    constexpr int STRIDE = F::LDS_ELEMS + 4;
    constexpr int HP = G / 2;
    using V2 = typename vec2<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);                 // 16 waves * 8 B = 128 B
    cplx<T>* lds0 = reinterpret_cast<cplx<T>*>(smem + 128);
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* lds = lds0 + (size_t)g * STRIDE;
    const int i0 = blockIdx.x * G;
    const int bl = blockIdx.y, band = band0 + bl;
    const cplx<T>* Tb = Tw + (size_t)band * d.T_band;
    cplx<T> vv[E], ev[E];
    row_inv_phase<T, L, E, 0>(Tb, twQ, ptw, lds0, lds, d.nx, i0, t, ev);
    row_inv_phase<T, L, E, 1>(Tb, twQ, ptw, lds0, lds, d.nx, i0, t, vv);
    // z[n] = e[n] + conj(w_M^n) o[n] ;  y[2n] = Re z, y[2n+1] = Im z
    const size_t rowoff = ((size_t)bl * d.nx + (i0 + g)) * d.ny;
    const V2* xr = reinterpret_cast<const V2*>(x + rowoff);
    const V2* br = beam ? reinterpret_cast<const V2*>(beam + rowoff) : nullptr;
    const V2* dr = dot_with ? reinterpret_cast<const V2*>(dot_with + rowoff) : nullptr;
    V2* orow = reinterpret_cast<V2*>(out + rowoff);
    double acc[1] = {0.0};
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int n = t + TPB * j;
        const cplx<T> zz = ev[j] + mulc(vv[j], twM[n]);
        V2 val;
        val.x = zz.x * scale;
        val.y = zz.y * scale;
        if (br) { const V2 b = br[n]; val.x *= b.x; val.y *= b.y; }
        const V2 xx = xr[n];
        val.x += sigmainv * xx.x;
        val.y += sigmainv * xx.y;
        orow[n] = val;
        if (dr) {
            const V2 dw = dr[n];
            acc[0] += (double)dw.x * (double)val.x + (double)dw.y * (double)val.y;
        }
        // keep at most 4 elements' worth of x/beam/dot_with/twiddle loads in flight: the
        // scheduler otherwise hoists all 4*E loads to the top and spills
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (dot_with) {
        __syncthreads();
        block_sum<1>(acc, red);
        if (threadIdx.x == 0) partials[(size_t)bl * gridDim.x + blockIdx.x] = acc[0];
    }
}

// -------------------------------------------------------------------- host side
struct FastTables {            // device tables owned by the plan (stored behind p->fast_tables)
    void* ptw_col;
    void* ptw_row;
    void* twM;                 // exp(-2 pi i n / M), n < L
};

template <typename T, int N, int E>
static int upload_ptw(void** dev) {
    constexpr int n = ptw_total<N, E>();
    std::vector<cplx<T>> h(n > 0 ? n : 1);
    if (n > 0) fill_ptw<T, N, E>(h.data());
    PFB_HIP_CHECK(hipMalloc(dev, sizeof(cplx<T>) * h.size()));
    PFB_HIP_CHECK(hipMemcpy(*dev, h.data(), sizeof(cplx<T>) * h.size(), hipMemcpyHostToDevice));
    return PFB_OK;
}

template <typename T, int N, int E>
static int prep_ptw(void** dev) { return upload_ptw<T, N, E>(dev); }

// size switch helpers -------------------------------------------------------------
#define PFB_POW2_SIZES(X) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096) X(8192)

template <typename T>
static int prep_tables(pfb_conv_plan* p, FastTables* ft) {
    const int H = p->nx, L = p->ny / 2;
    int rc = PFB_ERR_UNSUPPORTED;
    constexpr int lds_max = 160 * 1024;
    switch (H) {
#define X(NN) case NN: rc = prep_ptw<T, NN, FastCfg<T>::ECOL>(&ft->ptw_col);                          \
        if (rc == PFB_OK) PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_col_pow2<T, NN, FastCfg<T>::ECOL>, \
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_max)); break;
        PFB_POW2_SIZES(X)
#undef X
        default: break;
    }
    if (rc != PFB_OK) return rc;
    rc = PFB_ERR_UNSUPPORTED;
    switch (L) {
#define X(NN) case NN: rc = prep_ptw<T, NN, FastCfg<T>::EROW>(&ft->ptw_row);                          \
        if (rc == PFB_OK) PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_row_fwd_pow2<T, NN, FastCfg<T>::EROW>, \
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));                                   \
        if (rc == PFB_OK) PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_row_inv_pow2<T, NN, FastCfg<T>::EROW>, \
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_max)); break;
        PFB_POW2_SIZES(X)
#undef X
        default: break;
    }
    if (rc != PFB_OK) return rc;
    // twM[n] = exp(-2 pi i n / M), n < L  (M = ny)
    std::vector<cplx<T>> h(L);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int n = 0; n < L; ++n) {
        long double a = two_pi * (long double)n / (long double)p->ny;
        h[n] = cplx<T>((T)cosl(a), (T)(-sinl(a)));
    }
    PFB_HIP_CHECK(hipMalloc(&ft->twM, sizeof(cplx<T>) * L));
    PFB_HIP_CHECK(hipMemcpy(ft->twM, h.data(), sizeof(cplx<T>) * L, hipMemcpyHostToDevice));
    return PFB_OK;
}

template <typename T, int L>
static int rows_per_wg() { return row_groups<T, L, FastCfg<T>::EROW>(); }

bool pow2_supported(const pfb_conv_plan* p) {
    if (!is_pow2(p->nx) || !is_pow2(p->ny)) return false;
    if (p->P != 2 * p->nx || p->Q != 2 * p->ny) return false;
    if (p->nx < 64 || p->nx > 8192) return false;
    if (p->ny < 128 || p->ny > 16384) return false;
    return true;
}

int pow2_rows_per_wg(const pfb_conv_plan* p) {
    const int L = p->ny / 2;
    const bool f32 = p->dtype == PFB_F32;
    switch (L) {
#define X(NN) case NN: return f32 ? rows_per_wg<float, NN>() : rows_per_wg<double, NN>();
        PFB_POW2_SIZES(X)
#undef X
        default: return 0;
    }
}

int pow2_prepare(pfb_conv_plan* p) {
    FastTables* ft = (FastTables*)calloc(1, sizeof(FastTables));
    PFB_REQUIRE(ft != nullptr, PFB_ERR_ALLOC, "pow2_prepare: host alloc failed");
    p->fast_tables = ft;
    return p->dtype == PFB_F32 ? prep_tables<float>(p, ft) : prep_tables<double>(p, ft);
}

void pow2_release(pfb_conv_plan* p) {
    FastTables* ft = (FastTables*)p->fast_tables;
    if (!ft) return;
    if (ft->ptw_col) (void)hipFree(ft->ptw_col);
    if (ft->ptw_row) (void)hipFree(ft->ptw_row);
    if (ft->twM) (void)hipFree(ft->twM);
    free(ft);
    p->fast_tables = nullptr;
}

template <typename T>
static int set_psfhat_t(pfb_conv_plan* p, const void* psfhat, hipStream_t st) {
    const int nv = p->M + 1;
    dim3 grid((nv + 31) / 32, (p->P + 31) / 32, p->nband);
    hipLaunchKernelGGL((k_relayout_psf_pow2<T>), grid, dim3(256), 0, st, (const cplx<T>*)psfhat,
                       (cplx<T>*)p->psf_l, p->P, nv, p->psf_elems_per_band);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

int pow2_set_psfhat(pfb_conv_plan* p, const void* psfhat, hipStream_t st) {
    return p->dtype == PFB_F32 ? set_psfhat_t<float>(p, psfhat, st) : set_psfhat_t<double>(p, psfhat, st);
}

template <typename T, int H>
static void launch_col(pfb_conv_plan* p, const FastTables* ft, int band0, int nb, hipStream_t st) {
    constexpr int E = FastCfg<T>::ECOL;
    using F = RegFft<T, H, E>;
    constexpr int GC = col_groups<H, E>();
    const int nv = p->M + 1;
    const size_t lds = sizeof(cplx<T>) * (size_t)GC * F::LDS_ELEMS;
    hipLaunchKernelGGL((k_col_pow2<T, H, E>), dim3((nv + GC - 1) / GC, nb), dim3(GC * F::TPB), lds, st,
                       (cplx<T>*)p->T, (const cplx<T>*)p->psf_l, (const cplx<T>*)p->twP,
                       (const cplx<T>*)ft->ptw_col, nv, p->T_elems_per_band, p->psf_elems_per_band, band0);
}

template <typename T, int L>
static void launch_row_fwd(pfb_conv_plan* p, const FastTables* ft, int band0, int nb, const void* x,
                           const void* beam, hipStream_t st) {
    constexpr int E = FastCfg<T>::EROW;
    using F = RegFft<T, L, E>;
    constexpr int G = row_groups<T, L, E>();
    FastDims d{p->nx, p->ny, p->M, p->T_elems_per_band, p->psf_elems_per_band};
    const size_t lds = sizeof(cplx<T>) * (size_t)G * (F::LDS_ELEMS + 4);
    hipLaunchKernelGGL((k_row_fwd_pow2<T, L, E>), dim3(p->nx / G, nb), dim3(G * F::TPB), lds, st,
                       (const T*)x, (const T*)beam, (cplx<T>*)p->T, (const cplx<T>*)p->twQ,
                       (const cplx<T>*)ft->twM, (const cplx<T>*)ft->ptw_row, d, band0);
}

template <typename T, int L>
static void launch_row_inv(pfb_conv_plan* p, const FastTables* ft, int band0, int nb, const void* x,
                           const void* beam, double scale, double sigmainv, void* out,
                           const void* dot_with, hipStream_t st) {
    constexpr int E = FastCfg<T>::EROW;
    using F = RegFft<T, L, E>;
    constexpr int G = row_groups<T, L, E>();
    FastDims d{p->nx, p->ny, p->M, p->T_elems_per_band, p->psf_elems_per_band};
    const size_t lds = 128 + sizeof(cplx<T>) * (size_t)G * (F::LDS_ELEMS + 4);
    hipLaunchKernelGGL((k_row_inv_pow2<T, L, E>), dim3(p->nx / G, nb), dim3(G * F::TPB), lds, st,
                       (const cplx<T>*)p->T, (const cplx<T>*)p->twQ, (const cplx<T>*)ft->twM,
                       (const cplx<T>*)ft->ptw_row, (const T*)x, (const T*)beam, (const T*)dot_with,
                       (T*)out, p->partials, d, band0, (T)scale, (T)sigmainv);
}

template <typename T>
static int apply_t(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam, double scale,
                   double sigmainv, void* out, const void* dot_with, hipStream_t st) {
    const FastTables* ft = (const FastTables*)p->fast_tables;
    const int H = p->nx, L = p->ny / 2;
    prof_mark(p, st, 0);
    switch (L) {
#define X(NN) case NN: launch_row_fwd<T, NN>(p, ft, band0, nb, x, beam, st); break;
        PFB_POW2_SIZES(X)
#undef X
        default: set_error("pow2_apply: unsupported ny"); return PFB_ERR_UNSUPPORTED;
    }
    prof_mark(p, st, 1);
    switch (H) {
#define X(NN) case NN: launch_col<T, NN>(p, ft, band0, nb, st); break;
        PFB_POW2_SIZES(X)
#undef X
        default: set_error("pow2_apply: unsupported nx"); return PFB_ERR_UNSUPPORTED;
    }
    prof_mark(p, st, 2);
    switch (L) {
#define X(NN) case NN: launch_row_inv<T, NN>(p, ft, band0, nb, x, beam, scale, sigmainv, out, dot_with, st); break;
        PFB_POW2_SIZES(X)
#undef X
        default: break;
    }
    prof_mark(p, st, 3);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

int pow2_apply(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam, double scale,
               double sigmainv, void* out, const void* dot_with, hipStream_t st) {
    return p->dtype == PFB_F32 ? apply_t<float>(p, band0, nb, x, beam, scale, sigmainv, out, dot_with, st)
                               : apply_t<double>(p, band0, nb, x, beam, scale, sigmainv, out, dot_with, st);
}

}  // namespace pfb
