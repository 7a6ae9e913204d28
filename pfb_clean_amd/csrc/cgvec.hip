// cgvec.hip -- fused CG vector kernels, deterministic fp64 reductions and the PCG driver.
//
// Replaces the ~12 numpy passes per iteration of pfb/opt/pcg.py:86-122 and the
// serial numba norm_diff (pfb/utils/misc.py:1316-1351) by three HBM-bound kernels:
//   k_pcg_init    r = A x0 - b ; y = M r ; p = -y           + <r,y>, any(y)
//   k_pcg_update  x' = x + a p ; r' = r + a Ap ; y' = M r'   + <r',y'>, |x'-x|^2, |x'|^2
//   k_pcg_dir     p = beta p - M r'                          + any(p)
// All inner products accumulate in fp64 (wave64 __shfl_down -> LDS -> one partial per
// workgroup -> single-workgroup final sum in a fixed order), scalars stay in device
// memory (alpha, beta are read by the kernels from there), the p.Ap product comes
// fused out of the convolution epilogue (fftconv*.hip).
#include "conv_plan.hpp"
#include "pcg_state.hpp"
#include <chrono>
#include <unistd.h>
#include <cstring>
#include <cstdlib>

namespace pfb {

constexpr int RED_BLOCK = 256;
constexpr int RED_MAX_GRID = 1024;

static inline int red_grid(size_t nvec) {
    static const int cap = [] {           // PFB_RED_GRID: A/B knob for the streaming kernels' grid cap
        const char* e = getenv("PFB_RED_GRID");
        const int v = e ? atoi(e) : 0;
        return v > 0 && v <= 2048 ? v : RED_MAX_GRID;   // 4 sums x grid <= PFB_REDUCE_WS_DOUBLES
    }();
    size_t g = (nvec + RED_BLOCK - 1) / RED_BLOCK;
    g = (g + 3) / 4;                      // >= 4 vectors per thread when large
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

// 16-byte vector view of T
template <typename T> struct V16;
template <> struct V16<float>  { using type = float4;  static constexpr int N = 4; };
template <> struct V16<double> { using type = double2; static constexpr int N = 2; };

template <typename T, int V> struct Pack { T e[V]; };

template <typename T, int V>
__device__ __forceinline__ Pack<T, V> ld(const T* p, size_t i) {
    Pack<T, V> r;
    if constexpr (V == 1) {
        r.e[0] = p[i];
    } else {
        using VT = typename V16<T>::type;
        VT v = reinterpret_cast<const VT*>(p)[i];
        memcpy(&r, &v, sizeof(VT));
    }
    return r;
}
template <typename T, int V>
__device__ __forceinline__ void st(T* p, size_t i, const Pack<T, V>& r) {
    if constexpr (V == 1) {
        p[i] = r.e[0];
    } else {
        using VT = typename V16<T>::type;
        VT v;
        memcpy(&v, &r, sizeof(VT));
        reinterpret_cast<VT*>(p)[i] = v;
    }
}

// non-temporal forms (global_load / global_store ... nt) for streams that are touched once per iteration and are not
// re-read before ~256 MiB of other traffic has passed: they should not displace what IS re-read soon (at one band per
// GPU the half spectrum T, p and A p live in the Infinity Cache between kernels)
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> ld_nt(const T* p, size_t i) {
    Pack<T, V> r;
    if constexpr (V == 1) {
        r.e[0] = __builtin_nontemporal_load(p + i);
    } else {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p) + i);
        memcpy(&r, &v, 16);
    }
    return r;
}
template <typename T, int V>
__device__ __forceinline__ void st_nt(T* p, size_t i, const Pack<T, V>& r) {
    if constexpr (V == 1) {
        __builtin_nontemporal_store(r.e[0], p + i);
    } else {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f v;
        memcpy(&v, &r, 16);
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p) + i);
    }
}

// write NQ block results to ws[q * gridDim.x + blockIdx.x]
template <int NQ>
__device__ __forceinline__ void emit_partials(double (&acc)[NQ], double* __restrict__ ws) {
    __shared__ double red[NQ * (RED_BLOCK / 64)];
    block_sum<NQ>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) ws[(size_t)q * gridDim.x + blockIdx.x] = acc[q];
    }
}

// final stage: out[q] = sum_g ws[q*G + g], q < nq, fixed order
__global__ void __launch_bounds__(RED_BLOCK)
k_final_sum(const double* __restrict__ ws, int G, int nq, double* __restrict__ out) {
    __shared__ double red[RED_BLOCK / 64];
    for (int q = 0; q < nq; ++q) {
        double acc[1] = {0.0};
        for (int g = threadIdx.x; g < G; g += blockDim.x) acc[0] += ws[(size_t)q * G + g];
        block_sum<1>(acc, red);
        if (threadIdx.x == 0) out[q] = acc[0];
    }
}

// the fused update's four sums in one launch: wave q sums quantity q (fixed order ->
// deterministic) and writes out[dst[q]] -- [r'.y', |x'-x|^2, |x'|^2] go to S_RHON.., count(p') to S_ANY
struct Dst4 { int d[4]; };
__global__ void __launch_bounds__(256)
k_final_sum_waves(const double* __restrict__ ws, int G, int nq, double* __restrict__ out, Dst4 dst) {
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (q >= nq) return;
    double acc = 0.0;
    for (int g = lane; g < G; g += 64) acc += ws[(size_t)q * G + g];
    acc = wave_sum(acc);
    if (lane == 0) out[dst.d[q]] = acc;
}

template <typename T, int V>
__global__ void __launch_bounds__(RED_BLOCK)
k_dot(const T* __restrict__ a, const T* __restrict__ b, size_t nvec, double* __restrict__ ws) {
    double acc[1] = {0.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        Pack<T, V> pa = ld<T, V>(a, i), pb = ld<T, V>(b, i);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[0] += (double)pa.e[e] * (double)pb.e[e];
    }
    emit_partials<1>(acc, ws);
}

template <typename T, int V>
__global__ void __launch_bounds__(RED_BLOCK)
k_norm_diff(const T* __restrict__ x, const T* __restrict__ xp, size_t nvec,
            double* __restrict__ ws) {
    double acc[2] = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        Pack<T, V> a = ld<T, V>(x, i), b = ld<T, V>(xp, i);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const double d = (double)a.e[e] - (double)b.e[e];
            acc[0] += d * d;
            acc[1] += (double)a.e[e] * (double)a.e[e];
        }
    }
    emit_partials<2>(acc, ws);
}

template <typename T, int V>
__global__ void __launch_bounds__(RED_BLOCK)
k_any(const T* __restrict__ a, size_t nvec, double* __restrict__ ws) {
    double acc[1] = {0.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        Pack<T, V> pa = ld<T, V>(a, i);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[0] += (pa.e[e] != T(0)) ? 1.0 : 0.0;   // NaN counts, like np.any
    }
    emit_partials<1>(acc, ws);
}

template <typename T, int V>
__global__ void __launch_bounds__(RED_BLOCK)
k_axpby(T a, const T* __restrict__ x, T b, T* __restrict__ y, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        Pack<T, V> px = ld<T, V>(x, i), py = ld<T, V>(y, i);
#pragma unroll
        for (int e = 0; e < V; ++e) py.e[e] = a * px.e[e] + b * py.e[e];
        st<T, V>(y, i, py);
    }
}

// M(r) = r / mdiv when mdiv > 0 (pcg.py:264-267: M = x / sigmainv), identity otherwise.
// r holds A(x0) on entry.  r = r - b ; y = M r ; p = -y ; sums: <r,y>, count(y != 0)
template <typename T, int V>
__global__ void __launch_bounds__(RED_BLOCK)
k_pcg_init(T* __restrict__ r, const T* __restrict__ b, T* __restrict__ p, T mdiv, size_t nvec,
           double* __restrict__ ws) {
    double acc[2] = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        Pack<T, V> pr = ld<T, V>(r, i), pb = ld<T, V>(b, i), pp;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T rr = pr.e[e] - pb.e[e];
            const T y = mdiv > T(0) ? rr / mdiv : rr;
            pr.e[e] = rr;
            pp.e[e] = -y;
            acc[0] += (double)rr * (double)y;
            acc[1] += (y != T(0)) ? 1.0 : 0.0;
        }
        st<T, V>(r, i, pr);
        st<T, V>(p, i, pp);
    }
    emit_partials<2>(acc, ws);
}

// x' = x + a p ; r' = r + a Ap ; y' = M r' ; sums: <r',y'>, |x'-x|^2, |x'|^2
// alpha is read from device memory (fp64), rounded to T like the reference's scalar.
template <typename T, int V>
__global__ void __launch_bounds__(RED_BLOCK)
k_pcg_update(const T* __restrict__ x, const T* __restrict__ r, const T* __restrict__ p,
             const T* __restrict__ Ap, T* __restrict__ xn, T* __restrict__ rn,
             const double* __restrict__ alpha_dev, T mdiv, size_t nvec,
             double* __restrict__ ws) {
    // alpha_dev points at S[S_ALPHA]; once the solve is "dead" (all-zero direction,
    // pcg.py:106-107 detected one iteration late) the step is forced to 0: x' = x, r' = r
    const bool dead = alpha_dev[S_DEAD - S_ALPHA] != 0.0;
    const T alpha = dead ? T(0) : (T)alpha_dev[0];
    double acc[3] = {0.0, 0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        Pack<T, V> px = ld<T, V>(x, i), pr = ld<T, V>(r, i), pp = ld<T, V>(p, i),
                   pa = ld<T, V>(Ap, i), ox, orr;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T xnew = px.e[e] + alpha * pp.e[e];
            const T rnew = pr.e[e] + alpha * pa.e[e];
            const T y = mdiv > T(0) ? rnew / mdiv : rnew;
            ox.e[e] = xnew;
            orr.e[e] = rnew;
            const double d = (double)xnew - (double)px.e[e];
            acc[0] += (double)rnew * (double)y;
            acc[1] += d * d;
            acc[2] += (double)xnew * (double)xnew;
        }
        st<T, V>(xn, i, ox);
        st<T, V>(rn, i, orr);
    }
    emit_partials<3>(acc, ws);
}

// p = beta p - M r ; sums: count(p != 0)
template <typename T, int V>
__global__ void __launch_bounds__(RED_BLOCK)
k_pcg_dir(T* __restrict__ p, const T* __restrict__ r, const double* __restrict__ beta_dev,
          T mdiv, size_t nvec, double* __restrict__ ws) {
    double acc[1] = {0.0};
    if (beta_dev[S_DEAD - S_BETA] != 0.0) {        // dead: leave p (all zero) alone
        emit_partials<1>(acc, ws);
        return;
    }
    const T beta = (T)beta_dev[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        Pack<T, V> pp = ld<T, V>(p, i), pr = ld<T, V>(r, i);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const T y = mdiv > T(0) ? pr.e[e] / mdiv : pr.e[e];
            const T v = beta * pp.e[e] - y;
            pp.e[e] = v;
            acc[0] += (v != T(0)) ? 1.0 : 0.0;
        }
        st<T, V>(p, i, pp);
    }
    emit_partials<1>(acc, ws);
}

// update + direction in ONE pass (7 instead of 6 + 3 vector streams): possible because the
// predictive line search already knows rnorm_next = rho(alpha) before the vectors are touched,
// so beta = rho(alpha)/rho is available up front (the reference forms beta from the recomputed
// <r',y'>; the two differ by rounding only -- the recomputed value is still what the NEXT
// iteration uses as rnorm).  sums: <r',y'>, |x'-x|^2, |x'|^2, count(p' != 0)
// XNT: x is read and x' written non-temporally (x is not touched again until the next update, ~20 N bytes later)
// REV: walk the vectors from the END.  The inverse row kernel before this one finishes on the last band (its Ap, p, r
// rows are the freshest lines of the Infinity Cache) and the forward row kernel after it starts on the first band,
// whose p' this kernel then has written last: one band's worth of each stream is served from that cache at both
// boundaries instead of none.
template <typename T, int V, int U = 1, bool XNT = false, bool REV = false>
__global__ void __launch_bounds__(RED_BLOCK)
k_pcg_update_dir(const T* __restrict__ x, const T* __restrict__ r, T* __restrict__ p,
                 const T* __restrict__ Ap, T* __restrict__ xn, T* __restrict__ rn,
                 const double* __restrict__ alpha_dev, T mdiv, size_t nvec, double* __restrict__ ws) {
    const bool dead = alpha_dev[S_DEAD - S_ALPHA] != 0.0 || alpha_dev[S_STOP - S_ALPHA] != 0.0;
    const T alpha = dead ? T(0) : (T)alpha_dev[0];
    const T beta = (T)alpha_dev[S_BETA - S_ALPHA];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < nvec; i0 += U * stride) {
        Pack<T, V> px[U], pr[U], pp[U], pa[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {              // all loads of the U strips in flight together
            const size_t ii = i0 + u * stride;
            const size_t i = REV ? nvec - 1 - ii : ii;
            if (ii < nvec) { px[u] = XNT ? ld_nt<T, V>(x, i) : ld<T, V>(x, i); pr[u] = ld<T, V>(r, i); pp[u] = ld<T, V>(p, i); pa[u] = ld<T, V>(Ap, i); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t ii = i0 + u * stride;
            const size_t i = REV ? nvec - 1 - ii : ii;
            if (ii >= nvec) break;
            Pack<T, V> ox, orr;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T xnew = px[u].e[e] + alpha * pp[u].e[e];
                const T rnew = pr[u].e[e] + alpha * pa[u].e[e];
                const T y = mdiv > T(0) ? rnew / mdiv : rnew;
                ox.e[e] = xnew;
                orr.e[e] = rnew;
                const double d = (double)xnew - (double)px[u].e[e];
                acc[0] += (double)rnew * (double)y;
                acc[1] += d * d;
                acc[2] += (double)xnew * (double)xnew;
                if (!dead) {
                    const T pn = beta * pp[u].e[e] - y;
                    pp[u].e[e] = pn;
                }
                acc[3] += (pp[u].e[e] != T(0)) ? 1.0 : 0.0;
            }
            if constexpr (XNT) st_nt<T, V>(xn, i, ox); else st<T, V>(xn, i, ox);
            st<T, V>(rn, i, orr);
            if (!dead) st<T, V>(p, i, pp[u]);
        }
    }
    emit_partials<4>(acc, ws);
}

// tiny scalar kernels on the device state
__global__ void k_set_alpha(double* S) { S[S_ALPHA] = S[S_RHO] / S[S_PAP]; S[S_NBT] = 0.0; }
// Predictive backtracking.  With M(r) = r/d linear, <r',M r'> along r' = r + a Ap is the
// quadratic  rho(a) = rho + (2 a <r,Ap> + a^2 <Ap,Ap>) / d,  so the reference's loop
// "while rnorm_next > rnorm: alpha *= 0.75" (pcg.py:96-101) is evaluated on three scalars
// instead of three more passes over the vectors; the accepted step is then applied ONCE and
// rnorm_next is recomputed from the actual r' exactly as the reference does.
__global__ void k_alpha_predict(double* S, double mdiv) {
    const double d = mdiv > 0.0 ? mdiv : 1.0;
    const double rho = S[S_RHO], s1 = S[S_RAP] / d, s2 = S[S_APAP] / d;
    double alpha = rho / S[S_PAP];
    int nbt = 0;
    while (rho + (2.0 * alpha * s1 + alpha * alpha * s2) > rho && nbt < 200) { alpha *= 0.75; ++nbt; }
    S[S_ALPHA] = alpha;
    S[S_NBT] = (double)nbt;
}
__global__ void k_scale_alpha(double* S) { S[S_ALPHA] *= 0.75; }
__global__ void k_set_beta(double* S) { S[S_BETA] = S[S_RHON] / S[S_RHO]; }
__global__ void k_accept_rho(double* S) { S[S_RHO] = S[S_RHON]; }

// the fused update's four sums AND the end-of-iteration bookkeeping in one launch (no all-reduce between)
__global__ void __launch_bounds__(256)
k_final_sum_waves_end(const double* __restrict__ ws, int G, double* __restrict__ S) {
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dst[4] = {S_RHON, S_NUM, S_DEN, S_ANY};
    double acc = 0.0;
    for (int g = lane; g < G; g += 64) acc += ws[(size_t)q * G + g];
    acc = wave_sum(acc);
    if (lane == 0) S[dst[q]] = acc;
    __syncthreads();
    if (threadIdx.x == 0) iter_end_dev(S, 1);
}
__global__ void k_iter_begin(double* S, double mdiv, int predict) { iter_begin_dev(S, mdiv, predict); }
__global__ void k_iter_end(double* S, int fused) { iter_end_dev(S, fused); }
// end of iteration k and begin of iteration k+1 in one launch (the merged all-reduce path)
__global__ void k_iter_end_begin(double* S, double mdiv, int predict) {
    iter_end_dev(S, 1);
    iter_begin_dev(S, mdiv, predict);
}
// One launch per iteration for every scalar the sync-free driver needs: the three fused dots of the
// convolution (per-workgroup partials `cp`), the four sums the
// previous iteration's fused update left in `ws` (when `have_upd`), then -- unless an all-reduce has
// to come first (`logic` == 0) -- the end of that iteration and the begin of this one.
__global__ void __launch_bounds__(256)
k_iter_sums(const double* __restrict__ cp, int ncp, const double* __restrict__ ws, int G, int have_upd,
            double* __restrict__ S, double mdiv, int predict, int logic) {
    __shared__ double vals[8];
    pcg_iter_sums<false>(cp, ncp, ws, G, have_upd, S, mdiv, predict, logic, vals);
}
__global__ void k_final_check(double* S) {
    if (S[S_DEAD] == 0.0 && S[S_ANY] == 0.0) { S[S_DEAD] = 1.0; S[S_K] -= 1.0; S[S_EPS] = S[S_EPSP]; }
}
__global__ void k_init_state(double* S, double tol, double minit, double maxit) {
    S[S_TOL] = tol; S[S_MINIT] = minit; S[S_MAXIT] = maxit;
    S[S_RHO] = S[S_RHON];
    S[S_ANY] = S[S_NUM];                   // count(y != 0) = count(p != 0) for p = -y
    S[S_EPS] = 1.0; S[S_EPSP] = 1.0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
static inline bool can_vec(size_t n, std::initializer_list<const void*> ptrs) {
    if (n % V16<T>::N) return false;
    for (const void* p : ptrs) if (p && !aligned16(p)) return false;
    return true;
}

// ------------------------------------------------------------------ launch helpers
#define PFB_LAUNCH_VEC(T, kern, n, ptrs, ...)                                              \
    do {                                                                                   \
        if (can_vec<T>(n, ptrs)) {                                                         \
            const size_t nvec = (n) / V16<T>::N;                                           \
            const int G = red_grid(nvec);                                                  \
            G_used = G;                                                                    \
            hipLaunchKernelGGL((kern<T, V16<T>::N>), dim3(G), dim3(RED_BLOCK), 0, st,      \
                               __VA_ARGS__, nvec, ws);                                     \
        } else {                                                                           \
            const int G = red_grid(n);                                                     \
            G_used = G;                                                                    \
            hipLaunchKernelGGL((kern<T, 1>), dim3(G), dim3(RED_BLOCK), 0, st, __VA_ARGS__, \
                               (size_t)(n), ws);                                           \
        }                                                                                  \
    } while (0)

template <typename T>
static int dot_impl(const void* a, const void* b, size_t n, double* out, double* ws, hipStream_t st) {
    int G_used = 0;
    using PL = std::initializer_list<const void*>;
    PFB_LAUNCH_VEC(T, k_dot, n, (PL{a, b}), (const T*)a, (const T*)b);
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(RED_BLOCK), 0, st, ws, G_used, 1, out);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}
template <typename T>
static int nd_impl(const void* x, const void* xp, size_t n, double* out, double* ws, hipStream_t st) {
    int G_used = 0;
    using PL = std::initializer_list<const void*>;
    PFB_LAUNCH_VEC(T, k_norm_diff, n, (PL{x, xp}), (const T*)x, (const T*)xp);
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(RED_BLOCK), 0, st, ws, G_used, 2, out);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}
template <typename T>
static int any_impl(const void* a, size_t n, double* out, double* ws, hipStream_t st) {
    int G_used = 0;
    using PL = std::initializer_list<const void*>;
    PFB_LAUNCH_VEC(T, k_any, n, (PL{a}), (const T*)a);
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(RED_BLOCK), 0, st, ws, G_used, 1, out);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}
template <typename T>
static int axpby_impl(double a, const void* x, double b, void* y, size_t n, hipStream_t st) {
    using PL = std::initializer_list<const void*>;
    if (can_vec<T>(n, PL{x, y})) {
        const size_t nvec = n / V16<T>::N;
        hipLaunchKernelGGL((k_axpby<T, V16<T>::N>), dim3(red_grid(nvec)), dim3(RED_BLOCK), 0, st,
                           (T)a, (const T*)x, (T)b, (T*)y, nvec);
    } else {
        hipLaunchKernelGGL((k_axpby<T, 1>), dim3(red_grid(n)), dim3(RED_BLOCK), 0, st,
                           (T)a, (const T*)x, (T)b, (T*)y, n);
    }
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

// The fused update streams 4 reads + 3 writes; measured on MI355X (tools/micro/hbm_stream.hip and
// the bench) that mix runs fastest with ONE 256-thread workgroup per CU -- few concurrent streams
// per HBM channel -- not with the chip oversubscribed: 256 workgroups 0.60 ms, 1024 0.78 ms at
// 8 x 4096^2 fp32.  Returns the grid (= number of partial sums per quantity).
static int stream_grid(size_t nvec) {
    static const int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        const char* e = getenv("PFB_UPD_GRID");        // A/B knob
        const int o = e ? atoi(e) : 0;
        return o > 0 && o <= 2048 ? o : v;
    }();
    size_t g = (nvec + RED_BLOCK - 1) / RED_BLOCK;
    g = (g + 3) / 4;
    if (g < 1) g = 1;
    if (g > (size_t)ncu) g = ncu;
    return (int)g;
}
template <typename T>
static int launch_update_dir(size_t n, const T* x, const T* r, T* p, const T* Ap, T* xn, T* rn,
                             const double* alpha_dev, T mdiv, double* ws, hipStream_t st) {
    // two strips per trip (all eight loads in flight before the first use): fp64 0.79 -> 0.71 ms at 4 x 4096^2,
    // fp32 0.64 -> 0.62 ms at 8 x 4096^2 (rocprofv3); PFB_UPD_UNROLL=1 for the single-strip loop
    static const int unroll = [] { const char* e = getenv("PFB_UPD_UNROLL"); return e ? atoi(e) : 2; }();
    using PL = std::initializer_list<const void*>;
    constexpr int V = V16<T>::N;
    if (can_vec<T>(n, PL{x, r, p, Ap, xn, rn})) {
        const size_t nvec = n / V;
        const int G = stream_grid(nvec);
        // x non-temporal only when the vectors are far beyond the caches anyway (>= 32 MB each): small problems live in
        // L2 / the Infinity Cache between iterations and nt would send x to HBM (1024^2: 0.060 -> 0.066 ms per iteration)
        static const bool xnt_on = [] { const char* e = getenv("PFB_UPD_XNT"); return !e || atoi(e); }();
        const bool xnt = xnt_on && n * sizeof(T) >= ((size_t)32 << 20);
        static const bool rev = [] { const char* e = getenv("PFB_UPD_REV"); return !e || atoi(e); }();
        if (unroll == 2 && xnt && rev)
            hipLaunchKernelGGL((k_pcg_update_dir<T, V, 2, true, true>), dim3(G), dim3(RED_BLOCK), 0, st, x, r, p, Ap, xn, rn, alpha_dev, mdiv, nvec, ws);
        else if (unroll == 2 && xnt)
            hipLaunchKernelGGL((k_pcg_update_dir<T, V, 2, true>), dim3(G), dim3(RED_BLOCK), 0, st, x, r, p, Ap, xn, rn, alpha_dev, mdiv, nvec, ws);
        else if (unroll == 2)
            hipLaunchKernelGGL((k_pcg_update_dir<T, V, 2>), dim3(G), dim3(RED_BLOCK), 0, st, x, r, p, Ap, xn, rn, alpha_dev, mdiv, nvec, ws);
        else
            hipLaunchKernelGGL((k_pcg_update_dir<T, V, 1>), dim3(G), dim3(RED_BLOCK), 0, st, x, r, p, Ap, xn, rn, alpha_dev, mdiv, nvec, ws);
        return G;
    }
    const int G = stream_grid(n);
    hipLaunchKernelGGL((k_pcg_update_dir<T, 1, 1>), dim3(G), dim3(RED_BLOCK), 0, st, x, r, p, Ap, xn, rn, alpha_dev, mdiv, n, ws);
    return G;
}

// ------------------------------------------------------------------------ PCG driver
struct PcgWork {
    char* r; char* p; char* Ap; char* xalt; char* ralt;
    double* S; double* ws;
};

static size_t vec_bytes(const pfb_conv_plan* plan, int nb) {
    const size_t esz = plan->dtype == PFB_F32 ? 4 : 8;
    size_t b = (size_t)nb * plan->nx * plan->ny * esz;
    return (b + 255) & ~(size_t)255;
}

template <typename T>
static int pcg_impl(pfb_conv_plan* plan, int band0, int nb, const void* b, void* x, void* r_out,
                    const void* beam, double wsum, double sigmainv, double mdiv_d, double tol,
                    int maxit, int minit, int backtrack, void* work, pfb_allreduce_fn allreduce,
                    void* actx, pfb_pcg_result* res, hipStream_t st) {
    const size_t n = (size_t)nb * plan->nx * plan->ny;
    const size_t vb = vec_bytes(plan, nb);
    char* w = (char*)work;
    T* r = (T*)w;
    T* p = (T*)(w + vb);
    T* Ap = (T*)(w + 2 * vb);
    T* xalt = (T*)(w + 3 * vb);
    T* ralt = (T*)(w + 4 * vb);
    double* S = (double*)(w + 5 * vb);
    double* ws = S + 64;
    const T mdiv = (T)mdiv_d;
    double h[S_NSCALAR];
    int G_used = 0;
    using PL = std::initializer_list<const void*>;
    T* xcur = (T*)x;    // current iterate lives alternately in x / xalt
    T* rcur = r;
    T* xnew = xalt;
    T* rnew = ralt;

    memset(res, 0, sizeof(*res));
    if (allreduce) set_error("%s", "");        // comm_fail() quotes the hook's message: no stale text from an earlier call
    PFB_HIP_CHECK(hipMemsetAsync(S, 0, sizeof(double) * 64, st));

    // ---- the exchange can lose a participant (include/pfb_hip.h, "Failure protocol"): with a hook the solver never
    // waits for the device unboundedly -- it polls an event, probes the exchange (count = 0) and gives up after
    // PFB_COMM_TIMEOUT_S seconds, aborting the exchange (count < 0) so that no rank stays blocked in a collective
    struct WaitEvent {
        hipEvent_t ev = nullptr;
        ~WaitEvent() { if (ev) (void)hipEventDestroy(ev); }
    } wev;
    double comm_timeout = 600.0;
    if (allreduce) {
        if (const char* e = getenv("PFB_COMM_TIMEOUT_S")) { const double v = atof(e); if (v > 0) comm_timeout = v; }
        PFB_HIP_CHECK(hipEventCreateWithFlags(&wev.ev, hipEventDisableTiming));
    }
    auto comm_fail = [&](const char* what) -> int {
        char msg[384];
        snprintf(msg, sizeof(msg), "%s", pfb_last_error());      // the hook's own message, if it left one
        (void)allreduce(actx, nullptr, -1, (void*)st);            // abort: peers blocked in the collective are released
        set_error("pcg: %s%s%s -- exchange aborted, this rank's result is void", what, msg[0] ? ": " : "", msg);
        return PFB_ERR_COMM;
    };
    auto reduce_hook = [&](int first, int count) -> int {
        if (!allreduce) return PFB_OK;
        if (allreduce(actx, S + first, count, (void*)st) != 0) return comm_fail("the all-reduce hook failed");
        return PFB_OK;
    };
    auto wait_stream = [&]() -> int {
        if (!allreduce) { PFB_HIP_CHECK(hipStreamSynchronize(st)); return PFB_OK; }
        PFB_HIP_CHECK(hipEventRecord(wev.ev, st));
        const auto t0 = std::chrono::steady_clock::now();
        for (long spins = 0;; ++spins) {
            const hipError_t q = hipEventQuery(wev.ev);
            if (q == hipSuccess) return PFB_OK;
            if (q != hipErrorNotReady) { set_error("pcg: hipEventQuery -> %s", hipGetErrorString(q)); return PFB_ERR_HIP; }
            if (spins < 4096) continue;                           // a look normally returns within tens of microseconds
            if (allreduce(actx, nullptr, 0, (void*)st) != 0) return comm_fail("the exchange reported a failure");
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > comm_timeout) {
                char what[128];
                snprintf(what, sizeof(what), "no progress on the solver's stream for %.0f s (PFB_COMM_TIMEOUT_S)", el);
                set_error("%s", "");
                return comm_fail(what);
            }
            usleep(el < 0.01 ? 20 : 500);
        }
    };
    auto fetch = [&]() -> int {
        PFB_HIP_CHECK(hipMemcpyAsync(h, S, sizeof(double) * S_NSCALAR, hipMemcpyDeviceToHost, st));
        return wait_stream();
    };
    int err;

    // r = A(x0) - b ; y = M r ; p = -y                       pcg.py:71-76
    err = pfb_psfconv_apply(plan, band0, nb, xcur, beam, wsum, sigmainv, rcur, nullptr, nullptr, (void*)st);
    if (err != PFB_OK) return err;
    res->matvecs = 1;
    PFB_LAUNCH_VEC(T, k_pcg_init, n, (PL{rcur, b, p}), rcur, (const T*)b, p, mdiv);
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(RED_BLOCK), 0, st, ws, G_used, 2, S + S_RHON);
    // S_RHON = <r,y>, S_NUM = count(y != 0): move into place after the hook
    if ((err = reduce_hook(S_RHON, 2)) != PFB_OK) return err;
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(1), 0, st, S, tol, (double)minit, (double)maxit);
    if ((err = fetch()) != PFB_OK) return err;
    if (h[S_NUM] == 0.0) {                               // "Initial residual is zero"
        res->status = PFB_PCG_ZERO_RESIDUAL;
        res->eps = 1.0;
        res->rnorm = h[S_RHO];
        if (r_out) PFB_HIP_CHECK(hipMemcpyAsync(r_out, rcur, n * sizeof(T), hipMemcpyDeviceToDevice, st));
        return wait_stream();
    }

    int k = 0;
    double eps = 1.0;
    double rho = h[S_RHO];
    int status = -1;
    if (backtrack != 1) {
        // ---- sync-free driver (backtrack off or predictive).  Everything an iteration
        // needs to decide lives in S on the device; the host only has to look when the
        // stopping rule `(eps > tol or k < minit) and k < maxit` can actually fire, i.e.
        // never while k < minit.  Two reduction points per iteration:
        //   [p.Ap, r.Ap, Ap.Ap, any(p)]  and  [r'.y', |x'-x|^2, |x'|^2]
        // -- with sharded bands one all-reduce each, merged into ONE per iteration while k < minit.
        int khost = 0;
        bool pending_end = false;
        bool go = (1.0 > tol || 0 < minit) && 0 < maxit;
        const char* nf = getenv("PFB_PCG_NO_FUSE_DIR");
        const bool fuse_dir = !(nf && atoi(nf));     // A/B switch: separate update / direction kernels
        // Past minit the stopping rule needs eps after every iteration.  Looking costs a copy + stream sync
        // (13-24 us, tools/exp_sync_cost.py): a third of an iteration at 1024^2, 1 % at 8 x 4096^2.  Small
        // problems therefore run ONE iteration ahead: iteration j is enqueued, then the pinned snapshot taken
        // after iteration j-1 is read; the device evaluates the rule itself (S_STOP) so that the speculative
        // iteration is a no-op once it fired.  Price: one wasted iteration per solve -- large problems keep
        // the synchronous look.  PFB_PCG_LOOKAHEAD=0/1 overrides the size rule.
        bool lookahead = fuse_dir && n <= ((size_t)4 << 20);
        if (const char* la = getenv("PFB_PCG_LOOKAHEAD")) lookahead = fuse_dir && atoi(la) != 0;
        if (lookahead && !plan->pcg_pin) {
            PFB_HIP_CHECK(hipHostMalloc((void**)&plan->pcg_pin, sizeof(double) * 2 * S_NSCALAR, hipHostMallocDefault));
            PFB_HIP_CHECK(hipEventCreateWithFlags(&plan->pcg_ev[0], hipEventDisableTiming));
            PFB_HIP_CHECK(hipEventCreateWithFlags(&plan->pcg_ev[1], hipEventDisableTiming));
        }
        int slot = 0;
        bool have_prev = false, stale_h = false;
        while (go) {
            if (fuse_dir) {
                // PFB_PCG_TAIL=1 (default OFF): without an exchange the scalar work of the iteration -- summing the
                // convolution's fused dots and the previous update's sums, ending that iteration, beginning this one --
                // rides in the TAIL of the convolution's inverse row kernel (its last-arriving workgroup,
                // pcg_state.hpp): 4 launches per iteration instead of 5, bit-identical iterates (tools/check_tail.py:
                // 240 solves at six sizes).  Measured (profiles/r03_d_pcg_tail_ab.md): a TIE -- the tail's three
                // dependent device-scope round trips (ticket, partials, state) cost the ~6 us the launch of k_iter_sums
                // cost (0.0646 / 0.0658 vs 0.0658 / 0.0645 ms per iteration at 1024^2, 0.3147 / 0.3139 vs 0.3139 / 0.3131
                // at 4096^2 x 1); with agent-scope fences instead of write-through partials it LOST 8 % (every
                // workgroup's release fence writes back its XCD's whole L2).  Kept as a switch, not as the default.
                static const bool tail_on = [] { const char* e = getenv("PFB_PCG_TAIL"); return e && atoi(e); }();
                plan->tail_done = 0;
                if (tail_on && !allreduce)
                    plan->tail = PcgTail{S, ws, plan->tail_counter, mdiv_d, G_used, pending_end ? 1 : 0, backtrack == 2 ? 2 : 3};
                err = psfconv_apply_partials(plan, band0, nb, p, beam, wsum, sigmainv, Ap, p, rcur, (void*)st);
                plan->tail.S = nullptr;
            } else if (backtrack == 2)
                err = pfb_psfconv_apply_dots(plan, band0, nb, p, beam, wsum, sigmainv, Ap, p, rcur, S + S_PAP, (void*)st);
            else
                err = pfb_psfconv_apply(plan, band0, nb, p, beam, wsum, sigmainv, Ap, p, S + S_PAP, (void*)st);
            if (err != PFB_OK) return err;
            if (fuse_dir) {
                // ONE scalar launch per iteration: the convolution's dots, the previous update's sums
                // (left pending while nobody can look at k / eps, i.e. while k < minit) and the
                // bookkeeping; with sharded bands the all-reduce of those 7 (or 4) scalars sits between
                // the sums and the bookkeeping -- one RCCL call per iteration instead of two.
                const int predict = backtrack == 2 ? 2 : 3;   // beta always comes from rho(alpha)
                if (!plan->tail_done)
                hipLaunchKernelGGL(k_iter_sums, dim3(1), dim3(256), 0, st, (const double*)plan->partials,
                                   plan->last_npartials, (const double*)ws, G_used, pending_end ? 1 : 0, S,
                                   mdiv_d, predict, allreduce ? 0 : 1);
                if (allreduce) {
                    if ((err = reduce_hook(S_PAP, pending_end ? 7 : 4)) != PFB_OK) return err;
                    if (pending_end)
                        hipLaunchKernelGGL(k_iter_end_begin, dim3(1), dim3(1), 0, st, S, mdiv_d, predict);
                    else
                        hipLaunchKernelGGL(k_iter_begin, dim3(1), dim3(1), 0, st, S, mdiv_d, predict);
                }
                pending_end = false;
                G_used = launch_update_dir<T>(n, xcur, rcur, p, Ap, xnew, rnew, S + S_ALPHA, mdiv, ws, st);
                if (khost + 1 < minit && khost + 1 < maxit) {
                    pending_end = true;        // summed by the next iteration's k_iter_sums
                } else if (!allreduce) {
                    hipLaunchKernelGGL(k_final_sum_waves_end, dim3(1), dim3(256), 0, st, ws, G_used, S);
                } else {
                    hipLaunchKernelGGL(k_final_sum_waves, dim3(1), dim3(256), 0, st, ws, G_used, 4, S,
                                       (Dst4{{S_RHON, S_NUM, S_DEN, S_ANY}}));
                    if ((err = reduce_hook(S_RHON, 3)) != PFB_OK) return err;
                    hipLaunchKernelGGL(k_iter_end, dim3(1), dim3(1), 0, st, S, 1);
                }
                { T* t = xcur; xcur = xnew; xnew = t; t = rcur; rcur = rnew; rnew = t; }
            } else {
                hipLaunchKernelGGL(k_iter_begin, dim3(1), dim3(1), 0, st, S, mdiv_d, backtrack == 2 ? 1 : 0);
                PFB_LAUNCH_VEC(T, k_pcg_update, n, (PL{xcur, rcur, p, Ap, xnew, rnew}), (const T*)xcur,
                               (const T*)rcur, (const T*)p, (const T*)Ap, xnew, rnew,
                               (const double*)(S + S_ALPHA), mdiv);
                hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(RED_BLOCK), 0, st, ws, G_used, 3, S + S_RHON);
                if ((err = reduce_hook(S_RHON, 3)) != PFB_OK) return err;
                hipLaunchKernelGGL(k_iter_end, dim3(1), dim3(1), 0, st, S, 0);
                { T* t = xcur; xcur = xnew; xnew = t; t = rcur; rcur = rnew; rnew = t; }
                PFB_LAUNCH_VEC(T, k_pcg_dir, n, (PL{p, rcur}), p, (const T*)rcur, (const double*)(S + S_BETA), mdiv);
                hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(RED_BLOCK), 0, st, ws, G_used, 1, S + S_ANY);
            }
            ++khost;
            if (khost < minit && khost < maxit) continue;          // cannot stop yet: no need to look
            if (lookahead && khost < maxit) {
                double* snap = plan->pcg_pin + (size_t)slot * S_NSCALAR;
                PFB_HIP_CHECK(hipMemcpyAsync(snap, S, sizeof(double) * S_NSCALAR, hipMemcpyDeviceToHost, st));
                PFB_HIP_CHECK(hipEventRecord(plan->pcg_ev[slot], st));
                stale_h = true;
                if (have_prev) {
                    if (allreduce) {            // bounded: the same poll / probe / timeout as every other look
                        const auto t0 = std::chrono::steady_clock::now();
                        for (long spins = 0;; ++spins) {
                            const hipError_t q = hipEventQuery(plan->pcg_ev[slot ^ 1]);
                            if (q == hipSuccess) break;
                            if (q != hipErrorNotReady) { set_error("pcg: hipEventQuery -> %s", hipGetErrorString(q)); return PFB_ERR_HIP; }
                            if (spins < 4096) continue;
                            if (allreduce(actx, nullptr, 0, (void*)st) != 0) return comm_fail("the exchange reported a failure");
                            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                            if (el > comm_timeout) { set_error("%s", ""); return comm_fail("no progress on the solver's stream (PFB_COMM_TIMEOUT_S)"); }
                            usleep(el < 0.01 ? 20 : 500);
                        }
                    } else
                    PFB_HIP_CHECK(hipEventSynchronize(plan->pcg_ev[slot ^ 1]));
                    const double* hp = plan->pcg_pin + (size_t)(slot ^ 1) * S_NSCALAR;
                    if (hp[S_DEAD] != 0.0) { status = PFB_PCG_BREAKDOWN; break; }
                    if (hp[S_STOP] != 0.0) go = false;             // the iteration just enqueued changes nothing
                }
                have_prev = true;
                slot ^= 1;
                continue;
            }
            if ((err = fetch()) != PFB_OK) return err;
            stale_h = false;
            k = (int)h[S_K];
            eps = h[S_EPS];
            if (h[S_DEAD] != 0.0) { status = PFB_PCG_BREAKDOWN; break; }
            go = (eps > tol || k < minit) && k < maxit;
        }
        if (status >= 0 && stale_h && (err = fetch()) != PFB_OK) return err;
        if (status < 0) {
            // the direction built by the last iteration has not been looked at yet
            if ((err = reduce_hook(S_ANY, 1)) != PFB_OK) return err;
            hipLaunchKernelGGL(k_final_check, dim3(1), dim3(1), 0, st, S);
            if ((err = fetch()) != PFB_OK) return err;
            if (h[S_DEAD] != 0.0) status = PFB_PCG_BREAKDOWN;
        }
        k = (int)h[S_K];
        eps = h[S_EPS];
        rho = h[S_RHO];
        res->backtracks = (int)h[S_NBTSUM];
        res->matvecs = 1 + k + (status == PFB_PCG_BREAKDOWN ? 1 : 0);
    }
    while (backtrack == 1 && (eps > tol || k < minit) && k < maxit) {
        // Ap = A(p); S_PAP = <p,Ap> (+ <r,Ap>, <Ap,Ap> for the predictive line search)  pcg.py:89-91
        if (backtrack == 2)
            err = pfb_psfconv_apply_dots(plan, band0, nb, p, beam, wsum, sigmainv, Ap, p, rcur, S + S_PAP, (void*)st);
        else
            err = pfb_psfconv_apply(plan, band0, nb, p, beam, wsum, sigmainv, Ap, p, S + S_PAP, (void*)st);
        if (err != PFB_OK) return err;
        res->matvecs++;
        if ((err = reduce_hook(S_PAP, backtrack == 2 ? 3 : 1)) != PFB_OK) return err;
        if (backtrack == 2)
            hipLaunchKernelGGL(k_alpha_predict, dim3(1), dim3(1), 0, st, S, mdiv_d);
        else
            hipLaunchKernelGGL(k_set_alpha, dim3(1), dim3(1), 0, st, S);
        for (;;) {
            PFB_LAUNCH_VEC(T, k_pcg_update, n, (PL{xcur, rcur, p, Ap, xnew, rnew}), (const T*)xcur,
                           (const T*)rcur, (const T*)p, (const T*)Ap, xnew, rnew,
                           (const double*)(S + S_ALPHA), mdiv);
            hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(RED_BLOCK), 0, st, ws, G_used, 3, S + S_RHON);
            if ((err = reduce_hook(S_RHON, 3)) != PFB_OK) return err;
            if ((err = fetch()) != PFB_OK) return err;
            if (backtrack == 2) { res->backtracks += (int)h[S_NBT]; break; }
            if (backtrack && h[S_RHON] > rho) {          // pcg.py:96-101
                hipLaunchKernelGGL(k_scale_alpha, dim3(1), dim3(1), 0, st, S);
                res->backtracks++;
                continue;
            }
            break;
        }
        // accept x', r'
        { T* t = xcur; xcur = xnew; xnew = t; t = rcur; rcur = rnew; rnew = t; }
        // beta = rnorm_next / rnorm ; p = beta p - y      pcg.py:103-107
        hipLaunchKernelGGL(k_set_beta, dim3(1), dim3(1), 0, st, S);
        PFB_LAUNCH_VEC(T, k_pcg_dir, n, (PL{p, rcur}), p, (const T*)rcur, (const double*)(S + S_BETA), mdiv);
        hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(RED_BLOCK), 0, st, ws, G_used, 1, S + S_ANY);
        if ((err = reduce_hook(S_ANY, 1)) != PFB_OK) return err;
        hipLaunchKernelGGL(k_accept_rho, dim3(1), dim3(1), 0, st, S);
        const double rho_next = h[S_RHON], num = h[S_NUM], den = h[S_DEN];
        if ((err = fetch()) != PFB_OK) return err;
        if (h[S_ANY] == 0.0) {                           // break BEFORE k += 1
            status = PFB_PCG_BREAKDOWN;
            rho = rho_next;
            break;
        }
        rho = rho_next;
        k += 1;
        eps = sqrt(num / (1e-12 + den));                 // norm_diff, misc.py:1326-1351
    }
    if (status < 0) status = (k >= maxit) ? PFB_PCG_MAXIT : PFB_PCG_CONVERGED;
    res->status = status;
    res->iters = k;
    res->eps = eps;
    res->rnorm = rho;
    if (xcur != (T*)x) PFB_HIP_CHECK(hipMemcpyAsync(x, xcur, n * sizeof(T), hipMemcpyDeviceToDevice, st));
    if (r_out) PFB_HIP_CHECK(hipMemcpyAsync(r_out, rcur, n * sizeof(T), hipMemcpyDeviceToDevice, st));
    if ((err = wait_stream()) != PFB_OK) return err;
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

}  // namespace pfb

using namespace pfb;

extern "C" {

int pfb_dot(int dtype, const void* a, const void* b, size_t n, double* out, double* ws, void* stream) {
    PFB_REQUIRE(a && b && out && ws, PFB_ERR_INVALID, "dot: null argument");
    return dtype == PFB_F32 ? dot_impl<float>(a, b, n, out, ws, as_stream(stream))
                            : dot_impl<double>(a, b, n, out, ws, as_stream(stream));
}

int pfb_norm_diff_sums(int dtype, const void* x, const void* xp, size_t n, double* out, double* ws,
                       void* stream) {
    PFB_REQUIRE(x && xp && out && ws, PFB_ERR_INVALID, "norm_diff_sums: null argument");
    return dtype == PFB_F32 ? nd_impl<float>(x, xp, n, out, ws, as_stream(stream))
                            : nd_impl<double>(x, xp, n, out, ws, as_stream(stream));
}

int pfb_any_nonzero(int dtype, const void* a, size_t n, double* out, double* ws, void* stream) {
    PFB_REQUIRE(a && out && ws, PFB_ERR_INVALID, "any_nonzero: null argument");
    return dtype == PFB_F32 ? any_impl<float>(a, n, out, ws, as_stream(stream))
                            : any_impl<double>(a, n, out, ws, as_stream(stream));
}

int pfb_axpby(int dtype, double a, const void* x, double b, void* y, size_t n, void* stream) {
    PFB_REQUIRE(x && y, PFB_ERR_INVALID, "axpby: null argument");
    return dtype == PFB_F32 ? axpby_impl<float>(a, x, b, y, n, as_stream(stream))
                            : axpby_impl<double>(a, x, b, y, n, as_stream(stream));
}

size_t pfb_pcg_work_bytes(const pfb_conv_plan* plan, int nb) {
    if (!plan || nb <= 0) return 0;
    return 5 * vec_bytes(plan, nb) + sizeof(double) * (64 + PFB_REDUCE_WS_DOUBLES);
}

int pfb_pcg_solve(pfb_conv_plan* plan, int band0, int nb, const void* b, void* x, void* r_out,
                  const void* beam, double wsum, double sigmainv, double mdiv, double tol,
                  int maxit, int minit, int backtrack, void* work, pfb_allreduce_fn allreduce,
                  void* allreduce_ctx, pfb_pcg_result* result, void* stream) {
    PFB_REQUIRE(plan && b && x && work && result, PFB_ERR_INVALID, "pcg_solve: null argument");
    PFB_REQUIRE(band0 >= 0 && nb > 0 && band0 + nb <= plan->nband, PFB_ERR_INVALID,
                "pcg_solve: band range [%d,%d) outside plan", band0, band0 + nb);
    PFB_REQUIRE((reinterpret_cast<uintptr_t>(work) & 255u) == 0, PFB_ERR_INVALID,
                "pcg_solve: work must be 256-byte aligned");
    if (plan->dtype == PFB_F32)
        return pcg_impl<float>(plan, band0, nb, b, x, r_out, beam, wsum, sigmainv, mdiv, tol, maxit,
                               minit, backtrack, work, allreduce, allreduce_ctx, result,
                               as_stream(stream));
    return pcg_impl<double>(plan, band0, nb, b, x, r_out, beam, wsum, sigmainv, mdiv, tol, maxit,
                            minit, backtrack, work, allreduce, allreduce_ctx, result,
                            as_stream(stream));
}

}  // extern "C"
