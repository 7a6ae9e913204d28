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
// Layout (16-byte blocking).  Frequency columns are grouped NVB at a time (2 for fp32,
// 1 for fp64: NVB complex values = 16 bytes) WITHIN a parity class of the row transform
// (even bins v = 2m, odd bins v = 2m+1), L = ny/2, NBE = ceil((L+1)/NVB), NBO = L/NVB:
//     block b <  NBE : even bins m = NVB b + c   (fp32: the last block holds m = L + a dummy)
//     block b >= NBE : odd  bins m = NVB (b - NBE) + c
//     T[band][b][i][c]                 c < NVB, i < nx        -> blocks of nx*16 B
//     psf_l[band][b][pu][mu][c]        psfhat[2 mu + pu][v(b,c)], mu < H = nx
// so that (measured, tools/micro/piece_bw.hip: 64-B pieces 3.1 TB/s, 128-B pieces
// 5.5 TB/s from HBM) every strided access is a full 128-byte line: 8 rows x 2 columns.
//
// Kernels:
//   k_row_fwd_pow2  G = 8 image rows per workgroup -> X[v]; per parity the bins are
//                   LDS-transposed and written as 128-byte pieces T[b][i0..i0+8)[0..1]
//   k_col_pow2      one block (two columns) per thread group, 16-byte/lane streams:
//                   a -> FFT -> *psf_e -> IFFT, a*w -> FFT -> *psf_o -> IFFT, combine,
//                   store in place; the two columns share barriers (RegFft NV = 2)
//   k_row_inv_pow2  mirror image of row_fwd with the fused epilogue
//                   (1/(PQ wsum), beam, + sigmainv x, <dot_with, out> partials)
#include "conv_plan.hpp"
#include "fft_pow2.hpp"
#include <vector>
#include <cstdlib>

// This file is compiled into TWO objects (csrc/Makefile): PFB_POW2_PART=1 -- everything but the column
// kernels -- and PFB_POW2_PART=2 -- the column kernels and their launcher, built with LLVM's max-ILP machine
// scheduler (-mllvm -amdgpu-sched-strategy=max-ilp).  Their occupancy is pinned by the launch bounds, so
// the default occupancy-driven scheduling only costs them instruction-level parallelism: col 0.977 ->
// 0.904 ms fp32 and 1.152 -> 1.078 ms fp64 at 4096^2; the row kernels LOSE 5-25 % under the same switch,
// hence the split.  (0 = one object with everything, default scheduler.)
#ifndef PFB_POW2_PART
#define PFB_POW2_PART 0
#endif
#define PFB_POW2_COL  (PFB_POW2_PART != 1)
#define PFB_POW2_REST (PFB_POW2_PART != 2)

namespace pfb {

// ---- diagnostic build (make stamp: -DPFB_STAMP=1 -> libpfb_hip_stamp.so; never in the product): thread 0 of
// every workgroup of the three persistent kernels records the 100 MHz wall clock at its phase boundaries for
// the tiles / items PFB_STAMP_FIRST .. +PFB_STAMP_ITS of its loop; tools/stamp_report.py turns the buffer into
// a per-phase time table.  Layout: buf[((kernel * 1024 + workgroup) * PFB_STAMP_ITS + it) * 16 + slot].
#ifndef PFB_STAMP
#define PFB_STAMP 0
#endif
#if PFB_STAMP
#define PFB_STAMP_ITS 4
#define PFB_STAMP_FIRST 2
static __device__ unsigned long long* g_stamp_buf = nullptr;
// experiment of the diagnostic build (tools/exp_fwd_extra_stream.py): k_row_fwd_pow2q additionally STREAMS the rows of
// g_xtra_src (same shape as x) through registers, two 8-byte loads per even-bin sweep step, to measure how much extra
// HBM traffic the kernel absorbs -- the question behind folding the CG update into this kernel
static __device__ const float* g_xtra_src = nullptr;     // (not void*: hipMemcpyToSymbol's C overload would take the VALUE)
static __device__ double* g_xtra_sink = nullptr;
#define STAMP(KID, it, slot)                                                                              \
    do {                                                                                                  \
        const unsigned _si = (unsigned)((it) - PFB_STAMP_FIRST);                                          \
        if (threadIdx.x == 0 && g_stamp_buf && _si < PFB_STAMP_ITS && blockIdx.x < 1024)                  \
            g_stamp_buf[(((size_t)(KID) * 1024 + blockIdx.x) * PFB_STAMP_ITS + _si) * 16 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define STAMP(KID, it, slot) do { (void)(it); } while (0)
#endif

// elements per thread: the column kernel favours occupancy (small register arrays, it
// is the HBM-streaming kernel), the row kernels favour few threads per row so that 8
// rows (64-byte transposed pieces) fit one 1024-thread workgroup.
template <typename T> struct FastCfg;
// NVB: frequency columns per block of the T / psf_l layout = columns per 16-byte access
// (2 x complex64, 1 x complex128), so that 8 rows of a block form one 128-byte line.
template <> struct FastCfg<float>  { static constexpr int ECOL = 8; static constexpr int EROWMAX = 32; static constexpr int WCOL = 4; static constexpr int NVB = 2; };
template <> struct FastCfg<double> { static constexpr int ECOL = 8; static constexpr int EROWMAX = 16; static constexpr int WCOL = 2; static constexpr int NVB = 1; };

// Row kernels: one WAVE per image row whenever the row transform fits 64 lanes x E registers
// (E <= 32 fp32 / 16 fp64): the FFT exchanges are then wave-local (no s_barrier, the 8 waves
// of a workgroup drift freely and overlap each other's memory phases) and nothing spills.
// Measured on MI355X at 4096^2 x 8 bands (ms per launch, fwd / inv):
//   E=16 G=8 (16 waves/CU, 128-B pieces): 0.570 / 1.420     E=16 G=4: 0.569 / 1.312
//   E=8  G=4 (32 waves/CU,  64-B pieces): 0.622 / 1.116     E=32 G=8 (wave per row): 0.603 / 1.673
// -> occupancy beats piece size for the latency-bound inverse kernel; the forward kernel
// keeps 8 rows per workgroup (its strided side is the WRITE, full 128-byte lines).
template <typename T, int L, int E, int GMAX> constexpr int row_groups();
#ifndef PFB_ROW_E64_INV          // experiment knobs: elements per thread of the PLAIN fp64 row kernels at L >= 4096 (the persistent
                                 // inverse kernel has its own: InvPE)
#define PFB_ROW_E64_INV 8
#endif
#ifndef PFB_ROW_E64_FWD
#define PFB_ROW_E64_FWD 8
#endif
template <typename T, int L, bool INVK> struct RowCfg {
#ifndef PFB_INV_E16          // experiment knob: 16 elements per thread (8-row tiles, 128-byte pieces) in the fp32 inverse rows at L = 2048:
                             // 128 VGPRs + 232 B of scratch, 0.792 against 0.609 ms per 8-band launch
#define PFB_INV_E16 0
#endif
#ifndef PFB_INV_E16_4096     // 16 elements per thread in the fp32 inverse rows at L = 4096: 4 rows / 64-byte pieces instead of 2 / 32,
                             // operands read in the epilogue (no registers left to prefetch them): 1.054 against 1.188 ms per 2 x 8192^2
#define PFB_INV_E16_4096 1
#endif
    static constexpr int EMAX = sizeof(T) == 4 ? (INVK ? (((PFB_INV_E16 && L == 2048) || (PFB_INV_E16_4096 && L == 4096)) ? 16 : 8) : 16)
                                               : (L >= 4096 ? (INVK ? PFB_ROW_E64_INV : PFB_ROW_E64_FWD) : 8);
    static constexpr int E = (L / 64 < 8) ? 8 : (L / 64 > EMAX ? EMAX : L / 64);
    static constexpr int TPB = L / E;
    static constexpr bool WAVE = TPB <= 64;
    static constexpr int GMAX = INVK ? 4 : 8;
};
// inverse row kernel: second exchange buffer (one barrier per exchange instead of two)
// whenever 2 x G rows fit the LDS
template <typename T, int L> struct InvDb {
    using C = RowCfg<T, L, true>;
    static constexpr int G = row_groups<T, L, C::E, C::GMAX>();
    static constexpr int STRIDE = RegFft<T, L, C::E>::LDS_ELEMS + 4;
    static constexpr bool ON = (size_t)(2 * G * STRIDE + L) * sizeof(cplx<T>) + 384 <= (size_t)152 * 1024;
    static constexpr int OFF = ON ? G * STRIDE : 0;
};

constexpr int LDS_BUDGET = 152 * 1024;

// elements per thread of the column transform: 8, but 16 for fp64 at H = 8192 -- with E = 8 that
// size needs a 1024-thread workgroup (128-VGPR cap) and spilled ~200 bytes per lane: 4.32 -> 3.29 ms
// per 2 bands.  (fp32 at 8192 is the other way round: E = 16 under its 4-waves/SIMD launch bound
// spills 788 bytes, 2.22 -> 3.58 ms, and still 2.66 ms with a 2-waves/SIMD bound and 84 bytes.)
// ... and 4 up to H = 1024 (the plain, non-persistent column kernel): at these sizes a launch is ONE wave of workgroups, each a
// serial chain of four transforms; twice the threads per column halve the chain (1024^2 x 8 fp32: col 93.8 -> 73.2 us,
// 1024^2 x 1: 23.6 -> 18.1, fp64 27.6 -> 24.1; profiles/r03_ab_col_e4_small.md)
#ifndef PFB_COL_E4_MAXH
#define PFB_COL_E4_MAXH 1024
#endif
template <typename T, int H> constexpr int ecol() { return (H >= 8192 && sizeof(T) == 8) ? 16 : (H <= PFB_COL_E4_MAXH && H >= 64 ? 4 : FastCfg<T>::ECOL); }

// rows per workgroup for the row kernels
template <typename T, int L, int E, int GMAX>
constexpr int row_groups() {
    constexpr int TPB = L / E;
    int G = 256 / TPB > GMAX ? 256 / TPB : GMAX;
    const int stride = RegFft<T, L, E>::LDS_ELEMS + 4;
    while (G > 1 && (G * TPB > 1024 || (size_t)G * stride * sizeof(cplx<T>) + 384 > LDS_BUDGET)) G /= 2;
    return G;
}
template <int H, int E>
constexpr int col_groups() { return (H / E) >= 256 ? 1 : 256 / (H / E); }
constexpr int ECOLX = 8;      // elements per thread of the two-level column kernel's sub-transforms (k_col_pow2x)

// NVB adjacent complex values as one aligned 16-byte access
template <typename T, int NVB> struct alignas(16) Blk { cplx<T> c[NVB]; };
template <typename T, int NVB>
__device__ __forceinline__ Blk<T, NVB> loadb(const cplx<T>* src) {
    return *reinterpret_cast<const Blk<T, NVB>*>(src);
}
// the same, marked non-temporal (global_load ... nt): for streams that are read exactly once per launch and are far
// larger than the 256 MiB Infinity Cache (the PSF spectrum: 268 MB per 4096^2 fp32 band) -- they should not push
// the half-spectrum T, which the NEXT kernel re-reads, out of it
template <typename T, int NVB>
__device__ __forceinline__ Blk<T, NVB> loadb_nt(const cplx<T>* src) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    static_assert(sizeof(Blk<T, NVB>) == 16, "16-byte blocks");
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src));
    Blk<T, NVB> b;
    __builtin_memcpy(&b, &v, 16);
    return b;
}

#ifndef PFB_FWD_ROT
#define PFB_FWD_ROT 0
#endif
#ifndef PFB_COL_ROT
#define PFB_COL_ROT 0
#endif
#ifndef PFB_FWD_ABL             // ablation builds of k_row_fwd_pow2q (timing only, results are wrong): 1 no next-tile rows, 4 no stores
#define PFB_FWD_ABL 0           // of the sweeps, 8 no transforms
#endif
template <typename T, int NVB>
__device__ __forceinline__ void storeb(cplx<T>* dst, const Blk<T, NVB>& b);
// the strided stores of the forward sweeps (an ablation build keeps the values alive without the memory operation)
template <typename T, int NVB>
__device__ __forceinline__ void storeb_sweep(cplx<T>* dst, const Blk<T, NVB>& b) {
    if constexpr ((PFB_FWD_ABL & 4) != 0) {
#pragma unroll
        for (int c = 0; c < NVB; ++c) asm volatile("" :: "v"(b.c[c].x), "v"(b.c[c].y));
    } else storeb<T, NVB>(dst, b);
}
template <typename T, int NVB>
__device__ __forceinline__ void storeb(cplx<T>* dst, const Blk<T, NVB>& b) {
    *reinterpret_cast<Blk<T, NVB>*>(dst) = b;
}

// Make a pointer opaque to the optimiser: loads through the result cannot be CSE'd with /
// hoisted above earlier loads of the same address (used where a value is deliberately
// RE-READ from L2 instead of being kept live in 32 VGPRs across an FFT).
template <typename P>
__device__ __forceinline__ P* opaque(P* p) {
    asm volatile("" : "+v"(p));
    return p;
}

// LAUNDER_EARLY: see the comment at the row kernels -- values derived from the thread index are loop invariant, and
// the optimiser keeps whatever it can compute from them live across a persistent kernel's whole loop
__device__ __forceinline__ int launder(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// exp(-2 pi i k / 32) for the kernels that rebuild a long twiddle table from a short one: w^(t + c j) = w^t x (a 32nd root
// of unity that is a compile-time constant once the loop over j is unrolled)
template <typename T>
__device__ __forceinline__ cplx<T> root32(int k) {              // exp(-2 pi i k / 32); folds to constants for constant k
    constexpr T C[9] = {T(1), T(0.98078528040323044912618223613423903697L), T(0.92387953251128675612818318939678828682L),
                        T(0.83146961230254523707878837761790575673L), T(0.70710678118654752440084436210484903928L),
                        T(0.55557023301960222474283081394853287438L), T(0.38268343236508977172845998403039886676L),
                        T(0.19509032201612826784828486847702224093L), T(0)};
    auto cs = [&](int q) -> T {                                 // cos(2 pi q / 32)
        q &= 31;
        return q <= 8 ? C[q] : q <= 16 ? -C[16 - q] : q <= 24 ? -C[q - 16] : C[32 - q];
    };
    return cplx<T>(cs(k), -cs(k - 8));
}

// XCD-aware row-group order of the plain row kernels.  A workgroup of G rows touches T in G x 16-byte pieces; 8 rows
// make a 128-byte line.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2), so the
// SH = 8 / G row groups of one line are given to blocks b, b + 8, .. b + 8 (SH - 1): their pieces meet in ONE L2 and
// leave it (or are fetched into it) as a full line (speed only: any bijection is correct).
template <int G>
__device__ __forceinline__ int xcd_row_group(int bx, int nbx) {
    constexpr int SH = 8 / G;
    if constexpr (SH > 1) {
        if (nbx % (8 * SH) == 0) {
            const int q = bx / (8 * SH), rem = bx % (8 * SH);
            return SH * (q * 8 + (rem & 7)) + (rem >> 3);
        }
    }
    return bx;
}

struct FastDims {
    int nx, ny, M;              // M = ny (packed length), nv = M + 1 columns
    size_t T_band, psf_band;
    size_t xpitch, xband;       // forward row kernels: elements between rows / bands of x and beam (ny, nx ny for
                                // an image cube; the PSFHAT producer reads quadrants of the (P, Q) PSF: pitch Q)
};

// ---------------------------------------------------------------- psfhat re-layout
// column v of the half spectrum -> (block, slot) of the pair layout
__host__ __device__ inline void block_of_bin(int v, int L, int nvb, int* blk, int* c) {
    const int m = v >> 1;
    const int nbe = (L + nvb) / nvb;
    *blk = ((v & 1) ? nbe : 0) + m / nvb;
    *c = m % nvb;
}
inline int fast_nblocks(int L, int nvb) { return (L + nvb) / nvb + L / nvb; }
// element offset of psfhat[u][v(blk, c)] inside a band of psf_l (H = nx = P / 2 rows per parity).
//   four = 0: parity-major  psf_l[blk][u & 1][u >> 1][c]     (k_col_pow2, k_col_pow2p)
//   four = 1: class-major   psf_l[blk][u & 3][u >> 2][c]     (k_col_pow2x: the two-level column transform)
__host__ __device__ inline size_t psf_off(int blk, int u, int c, int H, int nvb, int four) {
    return four ? (((size_t)blk * 4 + (u & 3)) * (size_t)(H / 2) + (u >> 2)) * nvb + c
                : (((size_t)blk * 2 + (u & 1)) * (size_t)H + (u >> 1)) * nvb + c;
}

#if PFB_POW2_REST
// psf_l[band][blk][pu][mu][c] = psfhat[band][2 mu + pu][v]
template <typename T>
__global__ void k_relayout_psf_pow2(const cplx<T>* __restrict__ psfhat, cplx<T>* __restrict__ psf_l,
                                    int P, int nv, int L, int nvb, size_t psf_band, int four) {
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
        if (u < P && v < nv) {
            int blk, c;
            block_of_bin(v, L, nvb, &blk, &c);
            psf_l[(size_t)band * psf_band + psf_off(blk, u, c, H, nvb, four)] = tile[tx][r];
        }
    }
}

// psfhat[band][u][v] = psf_l[band][blk][u & 1][u >> 1][c]: the reference's layout back out of the plan's
template <typename T>
__global__ void k_unrelayout_psf_pow2(const cplx<T>* __restrict__ psf_l, cplx<T>* __restrict__ psfhat,
                                      int P, int nv, int L, int nvb, size_t psf_band, int four) {
    __shared__ cplx<T> tile[32][33];
    const int band = blockIdx.z;
    const int u0 = blockIdx.y * 32, v0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int H = P / 2;
    for (int r = ty; r < 32; r += 8) {
        const int v = v0 + r, u = u0 + tx;
        if (u < P && v < nv) {
            int blk, c;
            block_of_bin(v, L, nvb, &blk, &c);
            tile[tx][r] = psf_l[(size_t)band * psf_band + psf_off(blk, u, c, H, nvb, four)];
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int u = u0 + r, v = v0 + tx;
        if (u < P && v < nv) psfhat[((size_t)band * P + u) * nv + v] = tile[r][tx];
    }
}

#endif  // PFB_POW2_REST

#if PFB_POW2_COL
// ------------------------------------------------------------------------ column
template <typename T, int H, int E>
__global__ void __launch_bounds__((col_groups<H, E>() * (H / E)), (E >= 16 ? 2 : FastCfg<T>::WCOL))
k_col_pow2(cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ psf_l,
           const cplx<T>* __restrict__ twP, const cplx<T>* __restrict__ ptw,
           int nblk, size_t T_band, size_t psf_band, int band0) {
    using F = RegFft<T, H, E>;
    constexpr int TPB = F::TPB;
    constexpr int NVB = FastCfg<T>::NVB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* lds = reinterpret_cast<cplx<T>*>(smem) + (size_t)g * (NVB * F::LDS_ELEMS);
    const int blk = blockIdx.x * col_groups<H, E>() + g;
    const int band = band0 + blockIdx.y;
    const bool active = blk < nblk;
    const size_t b = active ? blk : 0;
    cplx<T>* col = Tw + (size_t)band * T_band + (b * (size_t)H + t) * NVB;             // [i][c]
    const cplx<T>* pe = psf_l + (size_t)band * psf_band + (b * (2 * (size_t)H) + t) * NVB;
    const cplx<T>* po = pe + (size_t)H * NVB;

    // a is read from HBM exactly once (rocprofv3 FETCH_SIZE showed the former "re-read from
    // L2" going to HBM: +1 GB per 8-band launch).  aw = a .* w_P^n is formed at load time and
    // parked in registers; its live range (first FFT pair) does not overlap ev's (second pair),
    // so the peak register demand is unchanged.
#ifndef PFB_COL_PREQ2
#define PFB_COL_PREQ2 1
#endif
    // PREQ2 (8192-point fp32 column pairs): a .* w_P^n is NOT parked in registers -- the second parity re-reads a
    // (an L2 / Infinity-Cache hit: this workgroup fetched it microseconds ago) -- and its registers take the PSF slices,
    // requested a few per FFT pass: psf_e lands during the first transform, psf_o during the second.  2.19 -> 2.00 ms
    // per 2 x 8192^2 (128 VGPRs + 244 B of scratch; = 2 with the even-bin result round-tripping through T: 160 B,
    // 2.03 ms).  The same idea lost on the fp64 columns (fp64-issue and LDS bound); this one is latency bound.
    constexpr bool PREQ2 = PFB_COL_PREQ2 && sizeof(T) == 4 && H >= 8192;
    cplx<T> vv[NVB][E], aw[PREQ2 ? 1 : NVB][PREQ2 ? 1 : E];
    {
        const cplx<T>* tw2 = twP + t;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            Blk<T, NVB> a;
            if (active) a = loadb<T, NVB>(col + NVB * TPB * j);
            [[maybe_unused]] cplx<T> w;
            if constexpr (!PREQ2) w = tw2[TPB * j];
#pragma unroll
            for (int c = 0; c < NVB; ++c) {
                vv[c][j] = active ? a.c[c] : cplx<T>(0, 0);
                if constexpr (!PREQ2) aw[c][j] = vv[c][j] * w;
            }
        }
    }
    // PREQ (OFF): request psf_e BEFORE the first transform and psf_o before the second, so that two of the three HBM
    // latencies of a block hide behind a transform.  Meant for the 8192-point fp64 columns (one 512-thread workgroup per
    // CU, 256 VGPRs per thread; the persistent kernel does not fit there), but vv + aw + q are 192 registers before the
    // radix-16 butterfly's own 64: 84 -> 556 bytes of scratch per lane.  Kept for the record, not compiled in.
    constexpr bool PREQ = PREQ2 || (false && sizeof(T) == 8 && H >= 8192);
    constexpr bool QSPR = PREQ2;
    constexpr int NPQ = F::NPASS;
    Blk<T, NVB> q[PREQ ? E : 1];
    if constexpr (PREQ && !QSPR) {
#pragma unroll
        for (int j = 0; j < E; ++j) q[j] = loadb<T, NVB>(pe + NVB * TPB * j);
    }
    // ---- even bins of the column transform
    if constexpr (QSPR) {
        F::template runN<false, NVB>(vv, lds, t, ptw, [&](auto k) {
            constexpr int K = decltype(k)::value;
#pragma unroll
            for (int j = (K * E) / NPQ; j < ((K + 1) * E) / NPQ; ++j) q[j] = loadb<T, NVB>(pe + NVB * TPB * j);
        });
    } else
    F::template runN<false, NVB>(vv, lds, t, ptw);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        Blk<T, NVB> p;
        if constexpr (PREQ) p = q[j]; else p = loadb<T, NVB>(pe + NVB * TPB * j);
#pragma unroll
        for (int c = 0; c < NVB; ++c) vv[c][j] = vv[c][j] * p.c[c];
    }
    if constexpr (PREQ && !QSPR) {
#pragma unroll
        for (int j = 0; j < E; ++j) q[j] = loadb<T, NVB>(po + NVB * TPB * j);
    }
    if constexpr (QSPR) {
        F::template runN<true, NVB>(vv, lds, t, ptw, [&](auto k) {
            constexpr int K = decltype(k)::value;
#pragma unroll
            for (int j = (K * E) / NPQ; j < ((K + 1) * E) / NPQ; ++j) q[j] = loadb<T, NVB>(po + NVB * TPB * j);
        });
    } else
    F::template runN<true, NVB>(vv, lds, t, ptw);
    // PREQ2 == 2: the even-bin result does not wait in registers either: a is re-read first, then the even-bin result
    // takes its place in T (same addresses, same thread: the loads have landed before the stores are issued) and comes
    // back for the final combination -- both round trips stay in the L2 / Infinity Cache
    constexpr bool EVG = PREQ2 && PFB_COL_PREQ2 >= 2;
    cplx<T> ev[EVG ? 1 : NVB][EVG ? 1 : E];
    if constexpr (!EVG) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int c = 0; c < NVB; ++c) { ev[c][j] = vv[c][j]; if constexpr (!PREQ2) vv[c][j] = aw[c][j]; }
        }
    }
    if constexpr (PREQ2) {
        const cplx<T>* tw2 = twP + t;
        cplx<T>* col2 = opaque(col);
        Blk<T, NVB> a2[E];
#pragma unroll
        for (int j = 0; j < E; ++j) if (active) a2[j] = loadb<T, NVB>(col2 + NVB * TPB * j);
        if constexpr (EVG) {
            __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0): a is in registers before its place is overwritten
            if (active) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    Blk<T, NVB> o;
#pragma unroll
                    for (int c = 0; c < NVB; ++c) o.c[c] = vv[c][j];
                    storeb<T, NVB>(col2 + NVB * TPB * j, o);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const cplx<T> w = tw2[TPB * j];
#pragma unroll
            for (int c = 0; c < NVB; ++c) vv[c][j] = active ? a2[j].c[c] * w : cplx<T>(0, 0);
        }
    }
    // ---- odd bins
    F::template runN<false, NVB>(vv, lds, t, ptw);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        Blk<T, NVB> p;
        if constexpr (PREQ) p = q[j]; else p = loadb<T, NVB>(po + NVB * TPB * j);
#pragma unroll
        for (int c = 0; c < NVB; ++c) vv[c][j] = vv[c][j] * p.c[c];
    }
    F::template runN<true, NVB>(vv, lds, t, ptw);
    if (active) {
        const cplx<T>* tw3 = opaque(twP + t);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const cplx<T> w = tw3[TPB * j];
            Blk<T, NVB> o;
            if constexpr (EVG) {
                const Blk<T, NVB> e2 = loadb<T, NVB>(opaque(col) + NVB * TPB * j);
#pragma unroll
                for (int c = 0; c < NVB; ++c) o.c[c] = e2.c[c] + mulc(vv[c][j], w);
            } else {
#pragma unroll
                for (int c = 0; c < NVB; ++c) o.c[c] = ev[c][j] + mulc(vv[c][j], w);
            }
            storeb<T, NVB>(col + NVB * TPB * j, o);
        }
    }
}

// --------------------------------------------------- column, persistent + prefetching
// The plain column kernel above is latency bound (rocprofv3: 58 % of wave-cycles parked in
// s_waitcnt / s_barrier, VALU 18 %): every block pays the HBM latency of a, psf_e and psf_o
// one after the other.  This version keeps ONE workgroup per CU resident (256 VGPRs per
// thread), loops over the (band, block) items, and has at all times in flight
//   * psf_e and psf_o of the CURRENT item (issued before the first FFT), and
//   * a (the T block) of the NEXT item,
// so that in steady state no load latency is exposed; the column twiddles w_P^n are
// loaded once per kernel and stay in registers.
// DB: a second set of exchange buffers (when the LDS holds it) -- one barrier per exchange instead of two.
// SPR: the loads are not issued in two bursts (top of the trip / after the first FFT) but a few per FFT PASS
// (RegFft pass hook): a burst of 16 x 16-byte loads per thread blocks every wave of the workgroup at issue for ~3.8 us
// of a 16 us trip (profiles/r02_a_phase_stamps_*), with nothing computing meanwhile.  Per trip:
//     FFT (even)   <- first half of the NEXT item's a          multiply by psf_e (requested during the last IFFT)
//     IFFT (even)  <- psf_o of this item
//     FFT (odd)    <- second half of the next item's a         multiply by psf_o
//     IFFT (odd)   <- psf_e of the NEXT item
template <typename T, int H, int E, bool DB = false, bool SPR = false, bool NT = false>
__global__ void __launch_bounds__((col_groups<H, E>() * (H / E)), 2)
k_col_pow2p(cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ psf_l,
            const cplx<T>* __restrict__ twP, const cplx<T>* __restrict__ ptwc,
            int nblk, int nitems, size_t T_band, size_t psf_band, int band0, int bstep) {
    constexpr int NVB = FastCfg<T>::NVB;
    constexpr int XB = NVB * RegFft<T, H, E>::LDS_ELEMS;  // one set of exchange buffers (elements)
    using F = RegFft<T, H, E, false, DB ? XB : 0, true>;  // twiddles from LDS: no vmcnt wait in the passes
    constexpr int X1 = (DB && (F::template nxch<1>() & 1)) ? 1 : 0;   // start parity of every other transform
    constexpr int TPB = F::TPB;
    constexpr int GC = col_groups<H, E>();
    constexpr int NP = F::NPASS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* ltw = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* lds = ltw + ((F::PTWC + 1) & ~1) + (size_t)g * ((DB ? 2 : 1) * XB);
    const int stride = gridDim.x * GC;
    const int niter = (nitems + stride - 1) / stride;           // same for every workgroup

    cplx<T> tw[E];
#pragma unroll
    for (int j = 0; j < E; ++j) tw[j] = twP[t + TPB * j];
    for (int k = threadIdx.x; k < F::PTWC; k += GC * TPB) ltw[k] = ptwc[k];

    auto col_of = [&](int item) -> cplx<T>* {
        const int bl = item / nblk, blk = item - bl * nblk;
        return Tw + (size_t)(band0 + bstep * bl) * T_band + ((size_t)blk * H + t) * NVB;
    };
    auto psf_of = [&](int item) -> const cplx<T>* {
        const int bl = item / nblk, blk = item - bl * nblk;
        return psf_l + (size_t)(band0 + bstep * bl) * psf_band + ((size_t)blk * 2 * H + t) * NVB;
    };

    // PFB_COL_ROT: the workgroup -> item map rotates by that many workgroups per trip (see k_row_inv_pow2p)
    auto item_of = [&](int it) -> int {
        if constexpr (PFB_COL_ROT != 0)
            return (int)(((unsigned)blockIdx.x + (unsigned)it * (unsigned)PFB_COL_ROT) % gridDim.x) * GC + g + it * stride;
        else return blockIdx.x * GC + g + it * stride;
    };
    int item = item_of(0);
    Blk<T, NVB> an[E];
    Blk<T, NVB> q[E];
    {
        const bool act = item < nitems;
        const cplx<T>* c0 = col_of(act ? item : 0);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (act) an[j] = loadb<T, NVB>(c0 + NVB * TPB * j);
            else {
#pragma unroll
                for (int c = 0; c < NVB; ++c) an[j].c[c] = cplx<T>(0, 0);
            }
        }
        if constexpr (SPR) {
            const cplx<T>* pe0 = psf_of(act ? item : 0);
#pragma unroll
            for (int j = 0; j < E; ++j) q[j] = NT ? loadb_nt<T, NVB>(pe0 + NVB * TPB * j) : loadb<T, NVB>(pe0 + NVB * TPB * j);
        }
    }
    __syncthreads();                                            // twiddle table visible
#pragma unroll 1
    for (int it = 0; it < niter; ++it, item = item_of(it)) {
        STAMP(1, it, 0);
        const bool active = item < nitems;
        cplx<T>* col = col_of(active ? item : 0);
        const cplx<T>* pe = psf_of(active ? item : 0);
        const cplx<T>* po = pe + (size_t)H * NVB;
        // next item (clamped: a trip past the end re-reads a valid block and drops it)
        const int nxt = item_of(it + 1);
        const bool nact = nxt < nitems;
        const cplx<T>* cn = col_of(nact ? nxt : (active ? item : 0));
        const cplx<T>* pen = psf_of(nact ? nxt : (active ? item : 0));
        cplx<T> vv[NVB][E], aw[NVB][E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int c = 0; c < NVB; ++c) { vv[c][j] = an[j].c[c]; aw[c][j] = vv[c][j] * tw[j]; }
        }
        if constexpr (!SPR) {
            // in issue order (vmcnt retires in order): psf_e of this item, then a of the next
#pragma unroll
            for (int j = 0; j < E; ++j) q[j] = NT ? loadb_nt<T, NVB>(pe + NVB * TPB * j) : loadb<T, NVB>(pe + NVB * TPB * j);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if (nact) an[j] = loadb<T, NVB>(cn + NVB * TPB * j);
                else {
#pragma unroll
                    for (int c = 0; c < NVB; ++c) an[j].c[c] = cplx<T>(0, 0);
                }
            }
        }
        // loads [lo, hi) of a group of CNT, slice k of NP
        auto ld_an = [&](auto k, int base) {
            constexpr int K = decltype(k)::value;
            constexpr int CNT = E / 2;
#pragma unroll
            for (int j = (K * CNT) / NP; j < ((K + 1) * CNT) / NP; ++j)
                an[base + j] = loadb<T, NVB>(cn + NVB * TPB * (base + j));
        };
        STAMP(1, it, 1);
        // ---- even bins
        if constexpr (SPR) F::template runN<false, NVB, 0>(vv, lds, t, ltw, [&](auto k) { ld_an(k, 0); });
        else F::template runN<false, NVB, 0>(vv, lds, t, ltw);
        STAMP(1, it, 2);
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int c = 0; c < NVB; ++c) vv[c][j] = vv[c][j] * q[j].c[c];
        }
        STAMP(1, it, 3);
        if constexpr (SPR) {
            F::template runN<true, NVB, X1>(vv, lds, t, ltw, [&](auto k) {
                constexpr int K = decltype(k)::value;
#pragma unroll
                for (int j = (K * E) / NP; j < ((K + 1) * E) / NP; ++j) q[j] = NT ? loadb_nt<T, NVB>(po + NVB * TPB * j) : loadb<T, NVB>(po + NVB * TPB * j);
            });
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) q[j] = NT ? loadb_nt<T, NVB>(po + NVB * TPB * j) : loadb<T, NVB>(po + NVB * TPB * j);
            F::template runN<true, NVB, X1>(vv, lds, t, ltw);
        }
        STAMP(1, it, 4);
        cplx<T> ev[NVB][E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int c = 0; c < NVB; ++c) { ev[c][j] = vv[c][j]; vv[c][j] = aw[c][j]; }
        }
        // ---- odd bins
        if constexpr (SPR) F::template runN<false, NVB, 0>(vv, lds, t, ltw, [&](auto k) { ld_an(k, E / 2); });
        else F::template runN<false, NVB, 0>(vv, lds, t, ltw);
        STAMP(1, it, 5);
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int c = 0; c < NVB; ++c) vv[c][j] = vv[c][j] * q[j].c[c];
        }
        STAMP(1, it, 6);
        if constexpr (SPR) {
            F::template runN<true, NVB, X1>(vv, lds, t, ltw, [&](auto k) {
                constexpr int K = decltype(k)::value;
#pragma unroll
                for (int j = (K * E) / NP; j < ((K + 1) * E) / NP; ++j) q[j] = NT ? loadb_nt<T, NVB>(pen + NVB * TPB * j) : loadb<T, NVB>(pen + NVB * TPB * j);
            });
        } else {
            F::template runN<true, NVB, X1>(vv, lds, t, ltw);
        }
        STAMP(1, it, 7);
        if (active) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                Blk<T, NVB> o;
#pragma unroll
                for (int c = 0; c < NVB; ++c) o.c[c] = ev[c][j] + mulc(vv[c][j], tw[j]);
                storeb<T, NVB>(col + NVB * TPB * j, o);
            }
        }
        STAMP(1, it, 8);
    }
}

// ------------------------------------------- column, two-level (four residue classes), persistent
// Long columns (H = 8192: one 16-byte block of a column is 128 KB, the whole register file of a CU holds four of them)
// do not fit the kernel above: its exchange buffers fill the LDS, the tables do not fit next to them, 1024 threads cap
// the registers at 128 -- the plain kernel then runs one workgroup per CU barrier to barrier with every load latency
// exposed (1.1 TB/s of traffic at 8192^2 fp32 against 5.4 TB/s at 4096^2) and spills.  Here the padded length-4 HS
// transform (H = 2 HS samples, zero beyond) is split by the residue of the output bin modulo 4:
//     X[4m + r] = FFT_HS( (a[n] + (-i)^r a[n + HS]) w_P^(r n) )[m]                      r = 0..3,  n, m < HS
//     b[n + s HS] = sum_r  i^(s r) conj(w_P^(r n)) IFFT_HS( X[4 . + r] psf[4 . + r] )[n]        s = 0, 1
// i.e. FOUR rounds of (combine, FFT_HS, multiply, IFFT_HS, accumulate) on the persistent, spill-free HS-point tile:
// half the threads, half the exchange buffer, and the loads ride inside the transforms (RegFft pass hooks), in issue
// order per trip
//     round r:      IFFT passes  <- psf slice of the NEXT round (last round: class 0 of the next item)
//     rounds 1, 3:  FFT passes   <- upper / lower half of the next item's column (into the registers a just left)
// Rounds run in the order 0, 2, 1, 3: after round 0 both output halves are the same array, two accumulators exist from
// the end of the second round only -- and the upper one lives in the LDS (`park`: the exchange buffer is half the
// usual size), so that at most FIVE arrays of E blocks are live in registers at any time (the 4096-point persistent
// kernel: five).  The twiddles w_P^n, n = t + TPB j, are w_P^t (LDS table of TPB entries) times the compile-time
// constants exp(-2 pi i j / 4E).  The PSF spectrum is stored class-major for this kernel (psf_l[blk][r][m][c] =
// psfhat[4 m + r], see psf_off), so that every slice is a contiguous 16-byte-per-lane stream.
template <typename T, int H, int E>
struct ColX {
    static constexpr int HS = H / 2;
    using F = RegFft<T, HS, E, false, 0, true>;
    static constexpr int NVB = FastCfg<T>::NVB;
    static constexpr int GC = col_groups<HS, E>();
    static constexpr int NT = GC * F::TPB;
    static constexpr int XB = NVB * F::LDS_ELEMS;
    static constexpr int PTWP = (F::PTWC + 1) & ~1;
    static constexpr size_t LDS = sizeof(cplx<T>) * ((size_t)PTWP + F::TPB + (size_t)GC * XB + (size_t)NT * E * NVB);
    static_assert(E == 8, "root32: the twiddle constants are the 4E-th roots of unity");
};

#ifndef PFB_COLX_QLATE
#define PFB_COLX_QLATE 1
#endif
template <typename T, int H, int E, bool NT = false, bool QLATE = (PFB_COLX_QLATE != 0)>
__global__ void __launch_bounds__((ColX<T, H, E>::NT), 2)
k_col_pow2x(cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ psf_l,
            const cplx<T>* __restrict__ twP, const cplx<T>* __restrict__ ptwc,
            int nblk, int nitems, size_t T_band, size_t psf_band, int band0, int bstep) {
    using X = ColX<T, H, E>;
    using F = typename X::F;
    constexpr int NVB = X::NVB, HS = X::HS, TPB = F::TPB, GC = X::GC, NP = F::NPASS, NTH = X::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* ltw = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* ltx = ltw + X::PTWP;                               // w_P^n, n < TPB
    cplx<T>* lds = ltx + TPB + (size_t)g * X::XB;
    Blk<T, NVB>* park = reinterpret_cast<Blk<T, NVB>*>(ltx + TPB + (size_t)GC * X::XB) + threadIdx.x;   // [j][thread]
    const int stride = gridDim.x * GC;
    const int niter = (nitems + stride - 1) / stride;           // same for every workgroup

    for (int k = threadIdx.x; k < F::PTWC; k += NTH) ltw[k] = ptwc[k];
    for (int k = threadIdx.x; k < TPB; k += NTH) ltx[k] = twP[k];

    auto col_of = [&](int item) -> cplx<T>* {
        const int bl = item / nblk, blk = item - bl * nblk;
        return Tw + (size_t)(band0 + bstep * bl) * T_band + ((size_t)blk * H + t) * NVB;
    };
    auto psf_of = [&](int item) -> const cplx<T>* {             // class 0 of the item; class r: + r HS NVB
        const int bl = item / nblk, blk = item - bl * nblk;
        return psf_l + (size_t)(band0 + bstep * bl) * psf_band + ((size_t)blk * 2 * H + t) * NVB;
    };
    auto ldq = [&](const cplx<T>* p) __attribute__((always_inline)) { return NT ? loadb_nt<T, NVB>(p) : loadb<T, NVB>(p); };

    int item = blockIdx.x * GC + g;
    Blk<T, NVB> al[E], ah[E], q[E];
    {
        const bool act = item < nitems;
        const cplx<T>* c0 = col_of(act ? item : 0);
        const cplx<T>* p0 = psf_of(act ? item : 0);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            al[j] = loadb<T, NVB>(c0 + NVB * TPB * j);
            ah[j] = loadb<T, NVB>(c0 + NVB * (HS + TPB * j));
        }
        if constexpr (!QLATE) {
#pragma unroll
            for (int j = 0; j < E; ++j) q[j] = ldq(p0 + NVB * TPB * j);
        }
    }
    __syncthreads();                                            // tables visible
#pragma unroll 1
    for (int it = 0; it < niter; ++it, item += stride) {
        const bool active = item < nitems;
        cplx<T>* col = col_of(active ? item : 0);
        const cplx<T>* ps = psf_of(active ? item : 0);
        const int nxt = item + stride;
        const bool nact = nxt < nitems;                         // a trip past the end re-reads a valid block and drops it
        const cplx<T>* cn = col_of(nact ? nxt : (active ? item : 0));
        const cplx<T>* psn = psf_of(nact ? nxt : (active ? item : 0));
        cplx<T> vv[NVB][E], ol[NVB][E];
        // w_P^t, laundered at every use: t and the table are loop invariant, and the optimiser would otherwise keep
        // w, w^2, w^3 of all E samples live across the whole item loop
        auto wt = [&]() __attribute__((always_inline)) { return ltx[launder(t)]; };
        auto mulq = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
#pragma unroll
                for (int c = 0; c < NVB; ++c) vv[c][j] = vv[c][j] * q[j].c[c];
            }
        };
        // PSF slices: QLATE requests the slice of round r inside the FORWARD transform of round r (one transform of
        // lead) instead of inside the inverse transform of the round before (two): q is then dead across every
        // accumulate / combine step, which is where the registers run out (fp64: 140 B of scratch per lane = 2.35 GB of
        // spill traffic per 2-band launch, a quarter of the kernel's HBM bytes -- profiles/r03_m_c5_hbm_traffic.json)
        auto qslice = [&](auto k, const cplx<T>* pq) __attribute__((always_inline)) {
            constexpr int K = decltype(k)::value;
#pragma unroll
            for (int j = (K * E) / NP; j < ((K + 1) * E) / NP; ++j) q[j] = ldq(pq + NVB * TPB * j);
        };
        auto fft_q = [&](const cplx<T>* pq, auto&& extra) __attribute__((always_inline)) {     // forward transform of a round
            F::template runN<false, NVB>(vv, lds, t, ltw, [&](auto k) {
                if constexpr (QLATE) qslice(k, pq);
                extra(k);
            });
        };
        auto ifft_q = [&](const cplx<T>* pq, auto&& extra) __attribute__((always_inline)) {   // inverse transform (!QLATE: requests the slice at pq)
            F::template runN<true, NVB>(vv, lds, t, ltw, [&](auto k) {
                if constexpr (!QLATE) qslice(k, pq);
                else extra(k);
            });
        };
        auto none = [](auto) {};
        STAMP(1, it, 0);
        // ---- round r = 0:  a_lo + a_hi
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int c = 0; c < NVB; ++c) vv[c][j] = al[j].c[c] + ah[j].c[c];
        }
        fft_q(ps, none);
        STAMP(1, it, 1);
        mulq();
        ifft_q(ps + (size_t)2 * HS * NVB, none);
        STAMP(1, it, 2);
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int c = 0; c < NVB; ++c) ol[c][j] = vv[c][j];
        }
        // ---- round r = 2:  (a_lo - a_hi) w^2n
        {
            const cplx<T> w1 = wt(), wt2 = w1 * w1;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const cplx<T> w2 = wt2 * root32<T>(2 * j);
#pragma unroll
                for (int c = 0; c < NVB; ++c) vv[c][j] = (al[j].c[c] - ah[j].c[c]) * w2;
            }
        }
        STAMP(1, it, 3);
        fft_q(ps + (size_t)2 * HS * NVB, none);
        STAMP(1, it, 4);
        mulq();
        ifft_q(ps + (size_t)1 * HS * NVB, none);
        STAMP(1, it, 5);
        {
            const cplx<T> w1 = wt(), wt2 = w1 * w1;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const cplx<T> w2 = wt2 * root32<T>(2 * j);
                Blk<T, NVB> hi;
#pragma unroll
                for (int c = 0; c < NVB; ++c) {
                    const cplx<T> u = mulc(vv[c][j], w2);
                    hi.c[c] = ol[c][j] - u;
                    ol[c][j] = ol[c][j] + u;
                }
                park[j * NTH] = hi;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- round r = 1:  (a_lo - i a_hi) w^n.  The input of round 3, (a_lo + i a_hi) w^3n, is formed here as well and
        // parked in a_lo's registers: the column dies one round early and a_hi's registers take the upper half of the
        // next item's column during this round's forward transform
        {
            const cplx<T> w1t = wt(), w3t = w1t * (w1t * w1t);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const cplx<T> w1 = w1t * root32<T>(j), w3 = w3t * root32<T>(3 * j);
#pragma unroll
                for (int c = 0; c < NVB; ++c) {
                    const cplx<T> lo = al[j].c[c], hi = ah[j].c[c];
                    vv[c][j] = addrot<false>(lo, hi) * w1;
                    al[j].c[c] = addrot<true>(lo, hi) * w3;
                }
                __builtin_amdgcn_sched_barrier(0);          // one sample at a time: the temporaries of all E at once spill
            }
        }
        STAMP(1, it, 6);
        // the next item's column: upper half into a_hi's registers (free since the inputs of rounds 1 and 3 were formed),
        // lower half into a_lo's (free once round 3's input has moved to vv).  Without QLATE they ride in the forward
        // transforms of rounds 1 / 3; with it those carry the PSF slices and the column rides in the inverse transforms,
        // so that no transform has more than five arrays of E blocks live.
        auto ld_ah = [&](auto k) __attribute__((always_inline)) {
            constexpr int K = decltype(k)::value;
#pragma unroll
            for (int j = (K * E) / NP; j < ((K + 1) * E) / NP; ++j) ah[j] = loadb<T, NVB>(cn + NVB * (HS + TPB * j));
        };
        auto ld_al = [&](auto k) __attribute__((always_inline)) {
            constexpr int K = decltype(k)::value;
#pragma unroll
            for (int j = (K * E) / NP; j < ((K + 1) * E) / NP; ++j) al[j] = loadb<T, NVB>(cn + NVB * TPB * j);
        };
        if constexpr (QLATE) fft_q(ps + (size_t)1 * HS * NVB, none); else fft_q(ps + (size_t)1 * HS * NVB, ld_ah);
        STAMP(1, it, 7);
        mulq();
        ifft_q(ps + (size_t)3 * HS * NVB, ld_ah);
        STAMP(1, it, 8);
        {
            const cplx<T> w1t = wt();
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const cplx<T> w1 = w1t * root32<T>(j);
                Blk<T, NVB> hi = park[j * NTH];
#pragma unroll
                for (int c = 0; c < NVB; ++c) {
                    const cplx<T> u = mulc(vv[c][j], w1);
                    ol[c][j] = ol[c][j] + u;
                    hi.c[c] = addrot<true>(hi.c[c], u);              // + i u
                }
                park[j * NTH] = hi;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- round r = 3: its input leaves a_lo's registers, which take the lower half of the next item's column
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int c = 0; c < NVB; ++c) vv[c][j] = al[j].c[c];
        }
        STAMP(1, it, 9);
        if constexpr (QLATE) fft_q(ps + (size_t)3 * HS * NVB, none); else fft_q(ps + (size_t)3 * HS * NVB, ld_al);
        STAMP(1, it, 10);
        mulq();
        ifft_q(psn, ld_al);
        STAMP(1, it, 11);
        if (active) {
            const cplx<T> w1t = wt(), w3t = w1t * (w1t * w1t);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const cplx<T> w3 = w3t * root32<T>(3 * j);
                const Blk<T, NVB> hi = park[j * NTH];
                Blk<T, NVB> o0, o1;
#pragma unroll
                for (int c = 0; c < NVB; ++c) {
                    const cplx<T> u = mulc(vv[c][j], w3);
                    o0.c[c] = ol[c][j] + u;
                    o1.c[c] = addrot<false>(hi.c[c], u);         // - i u
                }
                storeb<T, NVB>(col + NVB * TPB * j, o0);
                storeb<T, NVB>(col + NVB * (HS + TPB * j), o1);
            }
        }
        STAMP(1, it, 12);
    }
}

// ------------------------------------------------- column, forward only: the PSFHAT producer
// psfhat = r2c(ifftshift(psf)) (gridder.py:712-714) on the fast path's own kernels.  The (P, Q) = (2 nx, 2 ny)
// shifted PSF s is cut into its four (nx, ny) quadrants s_ab (a: top / bottom, b: left / right); the PRUNED
// forward row kernel turns each into a half spectrum T_ab (same layout as the convolution's T), and with
// sigma = (-1)^v (+ for the even-bin blocks, - for the odd-bin blocks)
//     R_top = T_00 + sigma T_01,   R_bot = T_10 + sigma T_11            (un-pruned row transform of length Q)
//     psfhat[2k]   = FFT_nx(R_top + R_bot)[k]
//     psfhat[2k+1] = FFT_nx((R_top - R_bot) .* w_P^n)[k]                 (un-pruned column transform of length P)
// written straight into the plan's psf_l[blk][pu][mu][c] layout.
template <typename T, int H, int E>
__global__ void __launch_bounds__((col_groups<H, E>() * (H / E)), (E >= 16 ? 2 : FastCfg<T>::WCOL))
k_col_fwd_pow2(const cplx<T>* __restrict__ Tq, cplx<T>* __restrict__ psf_b,
               const cplx<T>* __restrict__ twP, const cplx<T>* __restrict__ ptw,
               int nblk, int nbe, size_t T_band, int four) {
    using F = RegFft<T, H, E>;
    constexpr int TPB = F::TPB;
    constexpr int NVB = FastCfg<T>::NVB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* lds = reinterpret_cast<cplx<T>*>(smem) + (size_t)g * (NVB * F::LDS_ELEMS);
    const int blk = blockIdx.x * col_groups<H, E>() + g;
    const bool active = blk < nblk;
    const size_t b = active ? blk : 0;
    const T sg = b >= (size_t)nbe ? T(-1) : T(1);
    const cplx<T>* c00 = Tq + (b * (size_t)H + t) * NVB;
    cplx<T> ve[NVB][E], vo[NVB][E];
    {
        const cplx<T>* tw2 = twP + t;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const Blk<T, NVB> a00 = loadb<T, NVB>(c00 + NVB * TPB * j);
            const Blk<T, NVB> a01 = loadb<T, NVB>(c00 + T_band + NVB * TPB * j);
            const Blk<T, NVB> a10 = loadb<T, NVB>(c00 + 2 * T_band + NVB * TPB * j);
            const Blk<T, NVB> a11 = loadb<T, NVB>(c00 + 3 * T_band + NVB * TPB * j);
            const cplx<T> w = tw2[TPB * j];
#pragma unroll
            for (int c = 0; c < NVB; ++c) {
                const cplx<T> top = a00.c[c] + sg * a01.c[c], bot = a10.c[c] + sg * a11.c[c];
                ve[c][j] = top + bot;
                vo[c][j] = (top - bot) * w;
            }
        }
    }
    // psfhat[2 k + pu] of this block: parity-major or class-major (psf_off), k = t + TPB j
    F::template runN<false, NVB>(ve, lds, t, ptw);
    if (active) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            Blk<T, NVB> o;
#pragma unroll
            for (int c = 0; c < NVB; ++c) o.c[c] = ve[c][j];
            storeb<T, NVB>(psf_b + psf_off((int)b, 2 * (t + TPB * j), 0, H, NVB, four), o);
        }
    }
    F::template runN<false, NVB>(vo, lds, t, ptw);
    if (active) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            Blk<T, NVB> o;
#pragma unroll
            for (int c = 0; c < NVB; ++c) o.c[c] = vo[c][j];
            storeb<T, NVB>(psf_b + psf_off((int)b, 2 * (t + TPB * j) + 1, 0, H, NVB, four), o);
        }
    }
}

#endif  // PFB_POW2_COL

#if PFB_POW2_REST
// ------------------------------------------------------------------- row forward
// Hermitian post-processing of one parity: the group's transform Z (registers, natural order)
// goes to LDS, then every lane produces the NVB bins of one block for one row and the G lanes
// of a block lane-group write one contiguous G*16-byte piece of T.
template <typename T, int L, int E, int PAR, typename F>
__device__ __forceinline__ void row_fwd_post(const cplx<T> (&z)[E], const cplx<T>* __restrict__ twQ,
                                             cplx<T>* lds0, cplx<T>* lds, cplx<T>* __restrict__ Tb,
                                             int nx, int i0, int t) {
    constexpr int TPB = F::TPB;
    constexpr int G = row_groups<T, L, E, RowCfg<T, L, false>::GMAX>();
    constexpr int NT = G * TPB;
    constexpr int STRIDE = F::LDS_ELEMS + 4;
    const int rr = threadIdx.x % G, bi = threadIdx.x / G;
    const cplx<T>* zr = lds0 + (size_t)rr * STRIDE;
    __syncthreads();                             // earlier readers of the row buffers are done
    {
        cplx<T>* wp = lds + F::pad(t);
#pragma unroll
        for (int j = 0; j < E; ++j) wp[F::cpad(TPB * j)] = z[j];
    }
    __syncthreads();
    // X[v] = 1/2 [ (Z[v] + conj Z[M-v]) - i w_Q^v (Z[v] - conj Z[M-v]) ]
    //   even v = 2m   : Z -> Ze[m mod L], Ze[(L-m) mod L]      m = 0..L
    //   odd  v = 2m+1 : Z -> Zo[m],       Zo[L-1-m]            m = 0..L-1
    constexpr int NVB = FastCfg<T>::NVB;
    constexpr int NBE = (L + NVB) / NVB;
    constexpr int NBP = PAR ? L / NVB : NBE;                 // blocks of this parity
    constexpr int BSTEP = NT / G;
    cplx<T>* Tp = Tb + ((size_t)(PAR ? NBE : 0) * nx + i0 + rr) * NVB;
#ifndef PFB_FWD_LIN
#define PFB_FWD_LIN 1
#endif
    constexpr int MS = NVB * BSTEP;               // m advances by MS per trip
    // (fp32: 0.0564 -> 0.0547 ms at 2048^2 x 4; the fp64 sweeps measured 1-2 % slower with it and keep the plain form)
    if constexpr (PFB_FWD_LIN && sizeof(T) == 4 && (L / NVB) % BSTEP == 0 && MS % 16 == 0) {
        // incremental addressing (see fwdp_post_lin): the trip advances every index by a multiple of 16, so the two
        // padded LDS streams, the twiddle stream and the store address move by constants instead of being rebuilt
        // from the block index (selects, two pad()s, a 64-bit multiply per trip); the Nyquist block of the even
        // bins -- the one trip only bi = 0 makes -- is handled after the loop.
        constexpr int NREG = (L / NVB) / BSTEP;
        const int m0 = NVB * bi;
        const cplx<T>* za = zr + F::pad(m0);
        const cplx<T>* zb[NVB];
#pragma unroll
        for (int h = 0; h < NVB; ++h) zb[h] = zr + F::pad(L - PAR - m0 - h);
        const cplx<T>* tq = twQ + 2 * m0 + PAR;
        cplx<T>* st = Tp + (size_t)bi * nx * NVB;
        const size_t ustep = (size_t)BSTEP * nx * NVB;
        const bool first0 = PAR == 0 && bi == 0;
        for (int k = 0; k < NREG; ++k) {
            Blk<T, NVB> o;
#pragma unroll
            for (int h = 0; h < NVB; ++h) {
                const cplx<T> w = tq[2 * h];
                const cplx<T> zv = za[h];                      // pad(m0 + h) = pad(m0) + h: m0 is a multiple of NVB <= 2
                cplx<T> zm = *zb[h];
                if (first0 && h == 0 && k == 0) zm = zv;      // m = 0: both are z[0]
                o.c[h] = T(0.5) * addrot<false>(addc(zv, zm), w * subc(zv, zm));
                zb[h] -= F::cpad(MS);
            }
            storeb<T, NVB>(st, o);
            za += F::cpad(MS);
            tq += 2 * MS;
            st += ustep;
        }
        if (PAR == 0 && bi == 0) {                 // block NBE - 1: bin m = L (partner z[0]), beyond it zeros
            const cplx<T> z0 = zr[F::pad(0)];
            Blk<T, NVB> o;
            o.c[0] = T(0.5) * addrot<false>(addc(z0, z0), twQ[2 * L] * subc(z0, z0));
#pragma unroll
            for (int h = 1; h < NVB; ++h) o.c[h] = cplx<T>(0, 0);
            storeb<T, NVB>(Tp + (size_t)(NBE - 1) * nx * NVB, o);
        }
        return;
    }
    for (int b = bi; b < NBP; b += BSTEP) {       // (unrolling this loop spills: 0.52 -> 0.84 ms)
        Blk<T, NVB> o;
#pragma unroll
        for (int h = 0; h < NVB; ++h) {
            const int m = NVB * b + h;
            const bool valid = PAR || m <= L;
            const int ia = PAR ? m : (m >= L ? 0 : m);
            const int ib = PAR ? (L - 1 - m) : ((m == 0 || m > L) ? 0 : L - m);
            const cplx<T> w = twQ[valid ? 2 * m + PAR : 0];
            const cplx<T> zv = zr[F::pad(ia)];
            const cplx<T> zm = zr[F::pad(ib)];
            o.c[h] = T(0.5) * addrot<false>(addc(zv, zm), w * subc(zv, zm));
            if (!valid) o.c[h] = cplx<T>(0, 0);
        }
        storeb<T, NVB>(Tp + (size_t)b * nx * NVB, o);
    }
}

template <typename T, int L, int E>
__global__ void __launch_bounds__((row_groups<T, L, E, RowCfg<T, L, false>::GMAX>() * (L / E)))
k_row_fwd_pow2(const T* __restrict__ x, const T* __restrict__ beam, cplx<T>* __restrict__ Tw,
               const cplx<T>* __restrict__ twQ, const cplx<T>* __restrict__ twM,
               const cplx<T>* __restrict__ ptw, FastDims d, int band0) {
    // the even-bin and odd-bin transforms advance together (NV = 2) through ONE row buffer
    // (SEQX): the image row is read from HBM once instead of once per parity
    using F = RegFft<T, L, E, RowCfg<T, L, false>::WAVE, 0, false, true>;
    constexpr int TPB = F::TPB;
    constexpr int G = row_groups<T, L, E, RowCfg<T, L, false>::GMAX>();
    constexpr int STRIDE = F::LDS_ELEMS + 4;
    using V2 = typename vec2<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T>* lds0 = reinterpret_cast<cplx<T>*>(smem);
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* lds = lds0 + (size_t)g * STRIDE;
    const int i0 = xcd_row_group<G>(blockIdx.x, gridDim.x) * G;      // 8192-point fp64 rows: G = 2, 32-byte pieces
    const int bl = blockIdx.y, band = band0 + bl;
    const size_t rowoff = (size_t)bl * d.xband + (size_t)(i0 + g) * d.xpitch;
    const V2* xr = reinterpret_cast<const V2*>(x + rowoff) + t;
    const V2* br = beam ? reinterpret_cast<const V2*>(beam + rowoff) + t : nullptr;
    cplx<T>* Tb = Tw + (size_t)band * d.T_band;

    cplx<T> vv[2][E];
    {
        const cplx<T>* tm = twM + t;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            V2 a = xr[TPB * j];
            if (br) { V2 b = br[TPB * j]; a.x *= b.x; a.y *= b.y; }
            vv[0][j] = cplx<T>(a.x, a.y);                    // z[n] = x[2n] + i x[2n+1]
            vv[1][j] = vv[0][j] * tm[TPB * j];               // z .* w_M^n  (odd bins)
        }
    }
    F::template runN<false, 2>(vv, lds, t, ptw);
    row_fwd_post<T, L, E, 0, F>(vv[0], twQ, lds0, lds, Tb, d.nx, i0, t);
    row_fwd_post<T, L, E, 1, F>(vv[1], twQ, lds0, lds, Tb, d.nx, i0, t);
}

// Values derived from the thread index are loop invariant, so the optimiser computes every
// LDS / global address of the tile loop once, keeps ~40 of them live across it and spills:
// scratch reloads are vector-memory operations, and one of them inside an FFT (in-order
// vmcnt) waits for the prefetch it was supposed to overlap with.  Laundering the index at
// the top of each iteration makes the (cheap) address arithmetic part of the loop body.
// (launder() itself is defined near the top of the file)

// ------------------------------------------------- row forward, persistent + pipelined
// Same recipe as k_row_inv_pow2p below (read its comment first): one resident 1024-thread
// workgroup per CU walks the 8-row tiles; the NEXT tile's image rows are requested into
// registers as soon as the even-bin spectrum sits in the LDS and land while the two
// Hermitian post-processing sweeps of the current tile run; every twiddle comes from the
// LDS (pass table, w_M^n, w_Q^(2m+1) = w_M^m w_Q) so that nothing else queues behind the
// in-order vmcnt; the post-processing sweep is fully unrolled with clamped block indices
// (static store count -> the compiler's vmcnt for the prefetched rows is exact).
template <typename T, int L>
struct FwdP {
    // 128 threads per row, 8 rows per workgroup -- and for the 4096-point rows of 8192-pixel lines (BIG) 256 threads per
    // row, 4 rows (64-byte pieces, XCD-paired to full lines), with the twiddle table w_M^n cut down to the 512 entries a
    // sweep step spans (SMT): the full table (32 KB) does not fit next to four 4096-point rows.  Both transforms use 16
    // elements per thread, and with M = 2 L = 32 TPB every other twiddle is a table entry times a 32nd root of unity:
    //     w_M^(t + TPB j)      = w_M^t       root32(j)        (odd-bin pre-multiply, j = register index)
    //     w_M^(m0 + h + MS K)  = w_M^(m0+h)  root32(2 K)      (sweep step K, MS = NVB BSTEP = M / 16)
    static constexpr bool BIG = L >= 4096;
    static constexpr int E = BIG ? 16 : L / 128;
    static constexpr int EOK = (E == 8 || E == 16) ? E : 8;
    using F = RegFft<T, L, EOK, false, 0, true, true>;
    static constexpr int TPR = L / EOK;               // threads per row
    // fp64 rows of 4096 points: two rows fill the LDS -> 512-thread workgroups (256 registers per thread)
#ifndef PFB_FWDP_HALF       // experiment: 2048-point fp32 rows as 4-row, 512-thread tiles with small tables (74 KB of LDS): TWO
#define PFB_FWDP_HALF 0     // independent workgroups per CU, 64-byte pieces XCD-paired to full lines
#endif
    static constexpr bool HALF = PFB_FWDP_HALF && sizeof(T) == 4 && L == 2048;
    static constexpr int NT = ((BIG && sizeof(T) == 8) || HALF) ? 512 : 1024;
    static constexpr int G = NT / TPR;
    static constexpr int STRIDE = F::LDS_ELEMS + 4;
    static constexpr int NVB = FastCfg<T>::NVB;
    static constexpr int NBE = (L + NVB) / NVB;
    static constexpr int NBO = L / NVB;
    static constexpr int BSTEP = NT / G;
    static constexpr bool SMT = BIG || HALF;
    static constexpr int NTM = SMT ? NVB * BSTEP : L; // entries of the w_M table kept in the LDS
    static constexpr int WG_PER_CU = HALF ? 2 : 1;
    static constexpr int PTWP = (F::PTWC + 1) & ~1;
    static constexpr size_t LDS = sizeof(cplx<T>) * ((size_t)PTWP + NTM + (size_t)G * STRIDE);
    // (an fp64 version with two-row 512-thread tiles -- what the LDS allows -- writes 32-byte pieces and
    // measured 0.90 vs 0.54 ms for the plain kernel: not used)
    // measured: 0.50 -> 0.42 ms at ny = 4096 (x 8 bands of 4096 rows); at ny = 2048 the plain
    // wave-per-row kernel (E = 16, barrier-free exchanges) is 6 % faster than this one with E = 8
#ifndef PFB_FWDP_E8          // experiment knob: the persistent forward kernels also at 8 elements per thread (fp32 ny = 2048):
                             // 0.0633 (sequential parities) / 0.0677 ms against 0.0561 ms for the plain kernel at 2048^2 x 4
#define PFB_FWDP_E8 0
#endif
    // sweep step K advances the bin index by MS = NVB BSTEP: w_M^(MS K) = root32(KR K)
    static constexpr int KR = (32 * NVB * BSTEP) / (2 * L);
    // fp64 rows of 4096 points as 2-row 512-thread persistent tiles: 1.87 against 1.21 ms per 2 x 8192^2 for the plain
    // kernel (profiles/r03_i_*; the same verdict as at 2048 points in round 1) -- PFB_FWDP_F64BIG=1 builds them anyway
#ifndef PFB_FWDP_F64BIG
#define PFB_FWDP_F64BIG 0
#endif
    static constexpr bool OK = (E == 16 || (PFB_FWDP_E8 && E == 8 && sizeof(T) == 4)) && LDS <= (size_t)160 * 1024 &&
                               (PFB_FWDP_F64BIG || !(BIG && sizeof(T) == 8)) &&
                               (!SMT || (EOK == 16 && 32 * TPR == 2 * L && KR * 2 * L == 32 * NVB * BSTEP && NTM >= TPR));
    // w_M^(t + TPR j) from the table in the LDS
    __device__ __forceinline__ static cplx<T> tw_row(const cplx<T>* ltm, int t, int j) {
        if constexpr (SMT) return j == 0 ? ltm[t] : ltm[t] * root32<T>(j);
        else return ltm[t + TPR * j];
    }
    // tile v of a band -> first row: G < 8 rows make G x 16-byte pieces of T; tiles b and b + 8 (dealt to the same XCD,
    // hence one L2) take the pieces of ONE 128-byte line (cf. k_row_inv_pow2p)
    __device__ __forceinline__ static int tile_row(int rg, int tiles_per_band) {
        constexpr int SH = 8 / G;
        if constexpr (SH > 1) {
            if (tiles_per_band % (8 * SH) == 0) {
                const int q = rg / (8 * SH), rem = rg % (8 * SH);
                rg = SH * (q * 8 + (rem & 7)) + (rem >> 3);
            }
        }
        return rg * G;
    }
};

template <typename T, int L, int PAR, int K, int NIT, typename Hook>
__device__ __forceinline__ void post_steps(const cplx<T>* zr, const cplx<T>* ltm, cplx<T> wq1, cplx<T>* Tp, int nx,
                                           int bi, const Hook& hook);
// hook(PassIdx<k>): called at the top of sweep step k (the kernel issues a slice of the next tile's loads there)
#ifndef PFB_FWD_LIN
#define PFB_FWD_LIN 1
#endif
template <typename T, int L, int PAR, typename Hook>
__device__ __forceinline__ void fwdp_post_lin(const cplx<T>* zr, const cplx<T>* ltm, cplx<T> wq1,
                                              cplx<T>* __restrict__ Tb, int nx, int i0, int rr, int bi, const Hook& hook);
template <typename T, int L, int PAR, typename Hook = NoPassHook>
__device__ __forceinline__ void fwdp_post(const cplx<T>* zr, const cplx<T>* ltm, cplx<T> wq1,
                                          cplx<T>* __restrict__ Tb, int nx, int i0, int rr, int bi,
                                          const Hook& hook = Hook()) {
    using P = FwdP<T, L>;
    constexpr int NVB = P::NVB;
    constexpr int NBP = PAR ? P::NBO : P::NBE;
    constexpr int NIT = (NBP + P::BSTEP - 1) / P::BSTEP;
#if PFB_FWD_LIN
    fwdp_post_lin<T, L, PAR, Hook>(zr, ltm, wq1, Tb, nx, i0, rr, bi, hook);
#else
    cplx<T>* Tp = Tb + ((size_t)(PAR ? P::NBE : 0) * nx + i0 + rr) * NVB;
    post_steps<T, L, PAR, 0, NIT>(zr, ltm, wq1, Tp, nx, bi, hook);
#endif
}

template <typename T, int L, int PAR, int K, int NIT, typename Hook>
__device__ __forceinline__ void post_steps(const cplx<T>* zr, const cplx<T>* ltm, cplx<T> wq1, cplx<T>* Tp, int nx,
                                           int bi, const Hook& hook) {
    using P = FwdP<T, L>;
    using F = typename P::F;
    constexpr int NVB = P::NVB;
    constexpr int NBP = PAR ? P::NBO : P::NBE;
    if constexpr (K < NIT) {
        constexpr int k = K;
        hook(PassIdx<K>{});
        int b = bi + k * P::BSTEP;
        if (b >= NBP) b = NBP - 1;                 // clamped: duplicates of the last block, same data
        Blk<T, NVB> o;
#pragma unroll
        for (int h = 0; h < NVB; ++h) {
            const int m = NVB * b + h;
            const bool valid = PAR || m <= L;
            const int ia = PAR ? m : (m >= L ? 0 : m);
            const int ib = PAR ? (L - 1 - m) : ((m == 0 || m > L) ? 0 : L - m);
            cplx<T> w = ltm[m < L ? m : 0];
            if (PAR) w = w * wq1;
            else if (m >= L) w = cplx<T>(T(-1), T(0));              // w_Q^(2L) = -1
            const cplx<T> zv = zr[F::pad(ia)];
            const cplx<T> zm = zr[F::pad(ib)];
            o.c[h] = T(0.5) * addrot<false>(addc(zv, zm), w * subc(zv, zm));
            if (!valid) o.c[h] = cplx<T>(0, 0);
        }
        storeb_sweep<T, NVB>(Tp + (size_t)b * nx * NVB, o);
        __builtin_amdgcn_sched_barrier(0);         // keep the sweeps' LDS reads from piling up (spills)
        post_steps<T, L, PAR, K + 1, NIT, Hook>(zr, ltm, wq1, Tp, nx, bi, hook);
    }
}

// ---- the same sweep with strength-reduced addressing (PFB_FWD_LIN, default on).
// The sweep above recomputes for every step the clamped block index, two padded LDS indices per value and a 64-bit
// `block * nx` product for the store: per tile and thread 18 v_mad_u64_u32 / v_mul_lo_u32 (quarter-rate on CDNA) and
// ~150 index instructions next to ~230 of arithmetic (ISA count).  A step advances m by NVB * BSTEP, a multiple of 16:
// pad(m + 16 c) = pad(m) + 17 c, so each LDS stream is ONE base register with compile-time offsets; the store address
// is a workgroup-uniform base (band + step, scalar unit) plus a 32-bit per-thread offset that never changes.
// The irregular items -- m = 0 in the first even-bin step, and the last even-bin step, which holds only block NBE - 1
// (the Nyquist column) -- are handled apart.
template <typename T, int L, int PAR, int K, int NREG, typename Hook>
__device__ __forceinline__ void post_steps_lin(const cplx<T>* const (&za)[FastCfg<T>::NVB], const cplx<T>* const (&zb)[FastCfg<T>::NVB],
                                               const cplx<T>* lw, cplx<T> wq1, cplx<T>* ub, size_t ustep, unsigned voff,
                                               bool first0, const Hook& hook) {
    using P = FwdP<T, L>;
    using F = typename P::F;
    constexpr int NVB = P::NVB, MS = NVB * P::BSTEP;
    if constexpr (K < NREG) {
        hook(PassIdx<K>{});
        Blk<T, NVB> o;
#pragma unroll
        for (int h = 0; h < NVB; ++h) {
            cplx<T> w;
            if constexpr (P::SMT) w = K == 0 ? lw[h] : lw[h] * root32<T>(P::KR * K);  // w_M^(m0 + h + MS K), MS = KR M / 32
            else w = lw[MS * K + h];
            if (PAR) w = w * wq1;
            const cplx<T> zv = za[h][F::cpad(MS * K)];
            cplx<T> zm = zb[h][F::cpad(MS * (NREG - 1 - K))];
            if constexpr (PAR == 0 && K == 0) { if (first0 && h == 0) zm = zv; }      // m = 0: both are z[0]
            o.c[h] = T(0.5) * addrot<false>(addc(zv, zm), w * subc(zv, zm));
        }
        storeb_sweep<T, NVB>(ub + (size_t)K * ustep + voff, o);
        __builtin_amdgcn_sched_barrier(0);
        post_steps_lin<T, L, PAR, K + 1, NREG, Hook>(za, zb, lw, wq1, ub, ustep, voff, first0, hook);
    }
}

template <typename T, int L, int PAR, typename Hook>
__device__ __forceinline__ void fwdp_post_lin(const cplx<T>* zr, const cplx<T>* ltm, cplx<T> wq1,
                                              cplx<T>* __restrict__ Tb, int nx, int i0, int rr, int bi,
                                              const Hook& hook) {
    using P = FwdP<T, L>;
    using F = typename P::F;
    constexpr int NVB = P::NVB, BS = P::BSTEP, MS = NVB * BS;
    constexpr int NBP = PAR ? P::NBO : P::NBE;
    constexpr int NIT = (NBP + BS - 1) / BS;
    constexpr int NREG = PAR ? NIT : NIT - 1;
    static_assert(MS % 16 == 0, "a step keeps the padding phase");
    static_assert(PAR ? (NBP == NIT * BS) : (NBP == (NIT - 1) * BS + 1), "block counts of the two sweeps");
    cplx<T>* ub = Tb + (size_t)(PAR ? P::NBE : 0) * nx * NVB;                         // workgroup-uniform
    const size_t ustep = (size_t)BS * nx * NVB;
    const unsigned voff = ((unsigned)bi * (unsigned)nx + (unsigned)(i0 + rr)) * NVB;  // elements; < 2^32
    const int m0 = NVB * bi;
    const cplx<T>* za[NVB];
    const cplx<T>* zb[NVB];
#pragma unroll
    for (int h = 0; h < NVB; ++h) {
        za[h] = zr + F::pad(m0 + h);
        // partner index of the LAST regular step (the lowest address of the stream): L - PAR - m - MS (NREG - 1)
        // (m = 0 makes step 0 read index L: inside the row's padded allotment, and the value is replaced there)
        zb[h] = zr + F::pad(L - PAR - (m0 + h) - MS * (NREG - 1));
    }
    post_steps_lin<T, L, PAR, 0, NREG, Hook>(za, zb, ltm + m0, wq1, ub, ustep, voff, bi == 0, hook);
    if constexpr (PAR == 0) {
        // the last even-bin step: block NBE - 1 = bins m = L (Nyquist: w_Q^(2L) = -1, partner z[0]) and, for NVB = 2,
        // m = L + 1 (beyond the spectrum: zero).  Every thread stores it, like the clamped step it replaces.
        hook(PassIdx<NREG>{});
        const cplx<T> z0 = zr[F::pad(0)];
        Blk<T, NVB> o;
        o.c[0] = T(0.5) * addrot<false>(addc(z0, z0), cplx<T>(T(-1), T(0)) * subc(z0, z0));
#pragma unroll
        for (int h = 1; h < NVB; ++h) o.c[h] = cplx<T>(0, 0);
        storeb_sweep<T, NVB>(ub + (size_t)(NBP - 1) * nx * NVB + (unsigned)(i0 + rr) * NVB, o);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// SPR: the next tile's rows are requested two at a time inside the even-bin sweep (its beam rows inside the odd-bin
// sweep) instead of in one burst before it: the burst parks all 16 waves at issue (profiles/r02_a_phase_stamps_*).
template <typename T, int L, bool BEAM, bool SPR = false>
__global__ void __launch_bounds__((FwdP<T, L>::NT))
k_row_fwd_pow2p(const T* __restrict__ x, const T* __restrict__ beam, cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ twM,
                const cplx<T>* __restrict__ ptwc, FastDims d, int band0, int tiles_per_band, int ntiles,
                cplx<T> wq1) {
    using P = FwdP<T, L>;
    using F = typename P::F;
    constexpr int E = P::EOK, TPB = F::TPB, G = P::G, NT = P::NT;
    using V2 = typename vec2<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T>* ltw = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* ltm = ltw + P::PTWP;
    cplx<T>* lds0 = ltm + P::NTM;
    int vb = blockIdx.x;
    if (vb >= ntiles) return;
    for (int k = threadIdx.x; k < F::PTWC; k += NT) ltw[k] = ptwc[k];
    for (int k = threadIdx.x; k < P::NTM; k += NT) ltm[k] = twM[k];
    V2 xa[E], ba[BEAM ? E : 1];
    {
        const int bl = vb / tiles_per_band, i0 = P::tile_row(vb - bl * tiles_per_band, tiles_per_band);
        const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
        const size_t off = (size_t)bl * d.xband + (size_t)(i0 + g) * d.xpitch;
        const V2* xr = reinterpret_cast<const V2*>(x + off) + t;
#pragma unroll
        for (int j = 0; j < E; ++j) xa[j] = xr[TPB * j];
        if constexpr (BEAM) {
            const V2* br = reinterpret_cast<const V2*>(beam + off) + t;
#pragma unroll
            for (int j = 0; j < E; ++j) ba[j] = br[TPB * j];
        }
    }
    __syncthreads();                                    // tables visible
    for (int sit = 0;; ++sit) {
        STAMP(0, sit, 0);
        const int vbn = vb + (int)gridDim.x < ntiles ? vb + (int)gridDim.x : vb;   // last tile: harmless repeat
        const int bl = vb / tiles_per_band, i0 = P::tile_row(vb - bl * tiles_per_band, tiles_per_band);
        cplx<T>* Tb = Tw + (size_t)(band0 + bl) * d.T_band;
        cplx<T> vv[2][E];
        {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            cplx<T>* lds = lds0 + (size_t)g * P::STRIDE;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                V2 a = xa[j];
                if constexpr (BEAM) { a.x *= ba[j].x; a.y *= ba[j].y; }
                vv[0][j] = cplx<T>(a.x, a.y);                    // z[n] = x[2n] + i x[2n+1]
                vv[1][j] = vv[0][j] * P::tw_row(ltm, t, j);      // z .* w_M^n  (odd bins)
                if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            STAMP(0, sit, 1);
            F::template runN<false, 2>(vv, lds, t, ltw);
            STAMP(0, sit, 2);
        }
        {   // fresh index: nothing thread-derived stays live (and gets spilled) across the FFT
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            __syncthreads();                                     // last exchange's readers are done
            cplx<T>* wp = lds0 + (size_t)g * P::STRIDE + F::pad(t);
#pragma unroll
            for (int j = 0; j < E; ++j) wp[F::cpad(TPB * j)] = vv[0][j];
            // next tile's rows: in flight during both post-processing sweeps
            if constexpr (!SPR) {
                const int bln = vbn / tiles_per_band, i0n = P::tile_row(vbn - bln * tiles_per_band, tiles_per_band);
                const V2* xr = reinterpret_cast<const V2*>(x + (size_t)bln * d.xband + (size_t)(i0n + g) * d.xpitch) + t;
#pragma unroll
                for (int j = 0; j < E; ++j) xa[j] = xr[TPB * j];
            }
            __syncthreads();
        }
        STAMP(0, sit, 3);
        constexpr int NITE_ = (P::NBE + P::BSTEP - 1) / P::BSTEP, NITO_ = (P::NBO + P::BSTEP - 1) / P::BSTEP;
        {
            const int tid = launder((int)threadIdx.x);
            if constexpr (SPR) {
                const int g = tid / TPB, t = tid % TPB;
                const int bln = vbn / tiles_per_band, i0n = P::tile_row(vbn - bln * tiles_per_band, tiles_per_band);
                const V2* xr = reinterpret_cast<const V2*>(x + (size_t)bln * d.xband + (size_t)(i0n + g) * d.xpitch) + t;
                fwdp_post<T, L, 0>(lds0 + (size_t)(tid % G) * P::STRIDE, ltm, wq1, Tb, d.nx, i0, tid % G, tid / G,
                                   [&](auto k) {
                                       constexpr int K = decltype(k)::value;
#pragma unroll
                                       for (int j = (K * E) / NITE_; j < ((K + 1) * E) / NITE_; ++j) xa[j] = xr[TPB * j];
                                   });
            } else {
                fwdp_post<T, L, 0>(lds0 + (size_t)(tid % G) * P::STRIDE, ltm, wq1, Tb, d.nx, i0, tid % G, tid / G);
            }
        }
        STAMP(0, sit, 4);
        if constexpr (BEAM && !SPR) {   // the next tile's beam rows: requested once the even-bin registers are free
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            const int bln = vbn / tiles_per_band, i0n = P::tile_row(vbn - bln * tiles_per_band, tiles_per_band);
            const V2* br = reinterpret_cast<const V2*>(beam + (size_t)bln * d.xband + (size_t)(i0n + g) * d.xpitch) + t;
#pragma unroll
            for (int j = 0; j < E; ++j) ba[j] = br[TPB * j];
        }
        {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            cplx<T>* wp = lds0 + (size_t)g * P::STRIDE + F::pad(t);
            __syncthreads();                                     // even-bin sweep has read the rows
#pragma unroll
            for (int j = 0; j < E; ++j) wp[F::cpad(TPB * j)] = vv[1][j];
            __syncthreads();
            STAMP(0, sit, 5);
            if constexpr (BEAM && SPR) {
                const int bln = vbn / tiles_per_band, i0n = P::tile_row(vbn - bln * tiles_per_band, tiles_per_band);
                const V2* br = reinterpret_cast<const V2*>(beam + (size_t)bln * d.xband + (size_t)(i0n + g) * d.xpitch) + t;
                fwdp_post<T, L, 1>(lds0 + (size_t)(tid % G) * P::STRIDE, ltm, wq1, Tb, d.nx, i0, tid % G, tid / G,
                                   [&](auto k) {
                                       constexpr int K = decltype(k)::value;
#pragma unroll
                                       for (int j = (K * E) / NITO_; j < ((K + 1) * E) / NITO_; ++j) ba[j] = br[TPB * j];
                                   });
            } else {
                fwdp_post<T, L, 1>(lds0 + (size_t)(tid % G) * P::STRIDE, ltm, wq1, Tb, d.nx, i0, tid % G, tid / G);
            }
            STAMP(0, sit, 6);
            __syncthreads();                                     // rows free for the next transform
        }
        STAMP(0, sit, 7);
        if (vbn == vb) break;
        vb = vbn;
    }
}

// ------------------------------------------- row forward, persistent, parities in sequence
template <int NP, typename H> __device__ __forceinline__ void fwd_abl_hooks(H&& h) {
    if constexpr (NP > 0) { fwd_abl_hooks<NP - 1>(h); h(std::integral_constant<int, NP - 1>{}); }
}
template <typename F, typename V, typename C, typename H>
__device__ __forceinline__ void fwd_run(V& vv, C* lds, int t, const C* ltw, H&& h) {
    if constexpr ((PFB_FWD_ABL & 8) != 0) fwd_abl_hooks<F::NPASS>(h);
    else F::template run<false>(vv, lds, t, ltw, h);
}
// k_row_fwd_pow2p above runs the even-bin and the odd-bin transform of a tile together and then the two
// post-processing sweeps; the next tile's rows are requested in one burst between them, which parks all 16 waves at
// issue (profiles/r02_a_phase_stamps_*: 8.0 of 28 us per trip, and the 9.3 us of the transforms run with NOTHING in
// flight).  Here the parities run one after the other:
//     z = x [* beam]              -> FFT  -> LDS -> even-bin sweep (stores)
//     z .* w_M^n  (x now dead)    -> FFT  -> LDS -> odd-bin sweep  (stores)
// and the next tile's rows are requested a few per PASS of the second transform INTO THE REGISTERS THE CURRENT ROWS
// JUST LEFT (no extra registers): the even-bin sweep's stores drain and the next rows arrive while the second
// transform computes, the odd-bin sweep's stores drain during the next tile's first transform.
template <typename T, int L, bool BEAM>
__global__ void __launch_bounds__((FwdP<T, L>::NT))
k_row_fwd_pow2q(const T* __restrict__ x, const T* __restrict__ beam, cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ twM,
                const cplx<T>* __restrict__ ptwc, FastDims d, int band0, int tiles_per_band, int ntiles,
                cplx<T> wq1) {
    using P = FwdP<T, L>;
    constexpr int E = P::EOK;
    using F = RegFft<T, L, E, false, 0, true, false>;        // one transform at a time, twiddles from LDS
    constexpr int TPB = F::TPB, G = P::G, NT = P::NT, NP = F::NPASS;
    using V2 = typename vec2<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T>* ltw = reinterpret_cast<cplx<T>*>(smem);
    cplx<T>* ltm = ltw + P::PTWP;
    cplx<T>* lds0 = ltm + P::NTM;
    int vb = blockIdx.x;
    if (vb >= ntiles) return;
    for (int k = threadIdx.x; k < F::PTWC; k += NT) ltw[k] = ptwc[k];
    for (int k = threadIdx.x; k < P::NTM; k += NT) ltm[k] = twM[k];
    V2 xa[E], ba[BEAM ? E : 1];
    {
        const int bl = vb / tiles_per_band, i0 = P::tile_row(vb - bl * tiles_per_band, tiles_per_band);
        const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
        const size_t off = (size_t)bl * d.xband + (size_t)(i0 + g) * d.xpitch;
        const V2* xr = reinterpret_cast<const V2*>(x + off) + t;
#pragma unroll
        for (int j = 0; j < E; ++j) xa[j] = xr[TPB * j];
        if constexpr (BEAM) {
            const V2* br = reinterpret_cast<const V2*>(beam + off) + t;
#pragma unroll
            for (int j = 0; j < E; ++j) ba[j] = br[TPB * j];
        }
    }
    __syncthreads();                                    // tables visible
    for (int sit = 0;; ++sit) {
        STAMP(0, sit, 0);
        int vbn;
        if constexpr (PFB_FWD_ROT != 0) {      // (see k_row_inv_pow2p: the workgroup -> tile map rotates per trip)
            const int s1 = sit + 1;
            const int cand = (int)(((unsigned)blockIdx.x + (unsigned)s1 * (unsigned)PFB_FWD_ROT) % gridDim.x) + s1 * (int)gridDim.x;
            vbn = cand < ntiles ? cand : vb;
        } else {
            vbn = vb + (int)gridDim.x < ntiles ? vb + (int)gridDim.x : vb;   // last tile: harmless repeat
        }
        const int bl = vb / tiles_per_band, i0 = P::tile_row(vb - bl * tiles_per_band, tiles_per_band);
        cplx<T>* Tb = Tw + (size_t)(band0 + bl) * d.T_band;
        cplx<T> vv[E];
        // ---- even bins: z[n] = x[2n] + i x[2n+1]
        {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            cplx<T>* lds = lds0 + (size_t)g * P::STRIDE;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if constexpr (BEAM) { xa[j].x *= ba[j].x; xa[j].y *= ba[j].y; }
                vv[j] = cplx<T>(xa[j].x, xa[j].y);
            }
            STAMP(0, sit, 1);
            if constexpr ((PFB_FWD_ABL & 8) == 0) F::template run<false>(vv, lds, t, ltw);
            STAMP(0, sit, 2);
        }
        {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            __syncthreads();                                     // last exchange's readers are done
            cplx<T>* wp = lds0 + (size_t)g * P::STRIDE + F::pad(t);
#pragma unroll
            for (int j = 0; j < E; ++j) wp[F::cpad(TPB * j)] = vv[j];
            __syncthreads();
        }
        STAMP(0, sit, 3);
#if PFB_STAMP
        if (g_xtra_src) {           // diagnostic build only: see g_xtra_src
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            constexpr int NITE_ = (P::NBE + P::BSTEP - 1) / P::BSTEP;
            const V2* er = reinterpret_cast<const V2*>((const T*)g_xtra_src + (size_t)bl * d.xband + (size_t)(i0 + g) * d.xpitch) + t;
            T esum = 0;
            fwdp_post<T, L, 0>(lds0 + (size_t)(tid % G) * P::STRIDE, ltm, wq1, Tb, d.nx, i0, tid % G, tid / G,
                               [&](auto k) {
                                   constexpr int K = decltype(k)::value;
#pragma unroll
                                   for (int j = (K * E) / NITE_; j < ((K + 1) * E) / NITE_; ++j) { const V2 e = er[TPB * j]; esum += e.x + e.y; }
                               });
            if (esum == T(12345.678)) g_xtra_sink[0] = (double)esum;
        } else
#endif
        {
            const int tid = launder((int)threadIdx.x);
            fwdp_post<T, L, 0>(lds0 + (size_t)(tid % G) * P::STRIDE, ltm, wq1, Tb, d.nx, i0, tid % G, tid / G);
        }
        STAMP(0, sit, 4);
        // ---- odd bins: z .* w_M^n; the rows die here and their registers take the NEXT tile's rows
        {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            cplx<T>* lds = lds0 + (size_t)g * P::STRIDE;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                vv[j] = cplx<T>(xa[j].x, xa[j].y) * P::tw_row(ltm, t, j);
                if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const int bln = vbn / tiles_per_band, i0n = P::tile_row(vbn - bln * tiles_per_band, tiles_per_band);
            const size_t offn = (size_t)bln * d.xband + (size_t)(i0n + g) * d.xpitch;
            const V2* xr = reinterpret_cast<const V2*>(x + offn) + t;
            const V2* br = BEAM ? reinterpret_cast<const V2*>(beam + offn) + t : nullptr;
            // the first exchange of this transform waits (barrier) for the even-bin sweep's LDS reads
            fwd_run<F>(vv, lds, t, ltw, [&](auto k) {
                constexpr int K = decltype(k)::value;
#pragma unroll
                for (int j = (K * E) / NP; j < ((K + 1) * E) / NP; ++j) {
                    if constexpr ((PFB_FWD_ABL & 1) != 0) { xa[j].x = (T)(t + j); xa[j].y = (T)1; }
                    else xa[j] = xr[TPB * j];
                    if constexpr (BEAM) ba[j] = br[TPB * j];
                }
            });
            STAMP(0, sit, 5);
        }
        {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            cplx<T>* wp = lds0 + (size_t)g * P::STRIDE + F::pad(t);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < E; ++j) wp[F::cpad(TPB * j)] = vv[j];
            __syncthreads();
            fwdp_post<T, L, 1>(lds0 + (size_t)(tid % G) * P::STRIDE, ltm, wq1, Tb, d.nx, i0, tid % G, tid / G);
            STAMP(0, sit, 6);
            __syncthreads();                                     // rows free for the next transform
        }
        STAMP(0, sit, 7);
        if (vbn == vb) break;
        vb = vbn;
    }
}

// ------------------------------------------------------------------- row inverse
// one parity of the inverse row transform: gathers Y[2m + PAR][i0 .. i0+G) (G*8-byte
// pieces), LDS-transposes them to per-row order, builds the packed spectrum and runs
// the inverse FFT; result in vv (register j <-> sample t + TPB j of IFFT_L)
template <typename T, int L, int E, int PAR, typename PF>
__device__ __forceinline__ void row_inv_phase(PF&& after_loads_issued, const cplx<T>* __restrict__ Tb,
                                              const cplx<T>* __restrict__ twQ,
                                              const cplx<T>* __restrict__ ptw, cplx<T>* lds0,
                                              cplx<T>* lds, int nx, int i0, int t,
                                              cplx<T> (&vv)[E]) {
    using F = RegFft<T, L, E, RowCfg<T, L, true>::WAVE, InvDb<T, L>::OFF, true>;
    constexpr int TPB = F::TPB;
    constexpr int G = row_groups<T, L, E, RowCfg<T, L, true>::GMAX>();
    constexpr int NT = G * TPB;
    constexpr int STRIDE = F::LDS_ELEMS + 4;
    const int rr = threadIdx.x % G, bi = threadIdx.x / G;
    cplx<T>* yr = lds0 + (size_t)rr * STRIDE;
    __syncthreads();                            // previous phase done with the LDS
    constexpr int NVB = FastCfg<T>::NVB;
    constexpr int NBE = (L + NVB) / NVB;
    constexpr int NBP = PAR ? L / NVB : NBE;
    constexpr int BSTEP = NT / G;
    const cplx<T>* Tp = Tb + ((size_t)(PAR ? NBE : 0) * nx + i0 + rr) * NVB;
    // issue ALL strided loads of this thread before the first LDS write: a rolled loop
    // (load, wait, store, next) pays the HBM latency once per trip
    constexpr int NIT = (NBP + BSTEP - 1) / BSTEP;
    Blk<T, NVB> y[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int b = bi + k * BSTEP;
        if (b < NBP) y[k] = loadb<T, NVB>(Tp + (size_t)b * nx * NVB);    // bins of block b, row rr
    }
    // vmcnt is in order: loads issued from here on are YOUNGER than the strided loads above,
    // so waiting for those does not wait for these (prefetch hook of the kernel).  (Also
    // hoisting the w_Q twiddle loads of the unpacking loop up here would keep the prefetch in
    // flight longer, but costs 16 more live registers: measured slower, 1.01 vs 0.93 ms.)
    after_loads_issued();
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int b = bi + k * BSTEP;
        if (b < NBP) {
#pragma unroll
            for (int h = 0; h < NVB; ++h)
                if (PAR || NVB * b + h <= L) yr[F::pad(NVB * b + h)] = y[k].c[h];
        }
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
        const cplx<T> w = twQ[2 * m + PAR];
        vv[j] = addrot<true>(addc(yv, ym), mulc(subc(yv, ym), w));
        // bound the number of LDS / twiddle loads in flight (else all 3 E are hoisted: spills)
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    F::template run<true>(vv, lds, t, ptw);
}

template <typename T, int L, int E>
__global__ void __launch_bounds__((row_groups<T, L, E, RowCfg<T, L, true>::GMAX>() * (L / E)))
k_row_inv_pow2(const cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ twQ,
               const cplx<T>* __restrict__ twM, const cplx<T>* __restrict__ ptw,
               const T* __restrict__ x, const T* __restrict__ beam,
               const T* __restrict__ dot_with, const T* __restrict__ dot_with2, T* __restrict__ out,
               double* __restrict__ partials, FastDims d, int band0, T scale, T sigmainv, PcgTail tail) {
    using F = RegFft<T, L, E, RowCfg<T, L, true>::WAVE, InvDb<T, L>::OFF, true>;
    constexpr int TPB = F::TPB;
    constexpr int G = row_groups<T, L, E, RowCfg<T, L, true>::GMAX>();
    constexpr int NT = G * TPB;
    constexpr int STRIDE = F::LDS_ELEMS + 4;
    constexpr int HP = G / 2;
    using V2 = typename vec2<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);                 // 3 sums * 16 waves * 8 B
    cplx<T>* ltw = reinterpret_cast<cplx<T>*>(smem + 384);          // compact twiddle table (LDS)
    cplx<T>* lds0 = ltw + ((F::PTWC + 1) & ~1);
    const int g = threadIdx.x / TPB, t = threadIdx.x % TPB;
    cplx<T>* lds = lds0 + (size_t)g * STRIDE;
    for (int k = threadIdx.x; k < F::PTWC; k += NT) ltw[k] = ptw[k];   // visible after the first barrier
    // XCD-aware row-group order: with G = 4 a 128-byte line of T is shared by two row groups;
    // workgroups are dealt round-robin over the 8 XCDs, so blocks b and b+8 share an L2 --
    // give THEM the two halves of a line (speed only, any mapping is correct).
    const int rg = xcd_row_group<G>(blockIdx.x, gridDim.x);
    const int i0 = rg * G;
    const int bl = blockIdx.y, band = band0 + bl;
    const cplx<T>* Tb = Tw + (size_t)band * d.T_band;
    cplx<T> vv[E], ev[E];
    row_inv_phase<T, L, E, 0>([] {}, Tb, twQ, ltw, lds0, lds, d.nx, i0, t, ev);
    // The kernel's phases do not overlap inside one workgroup (ablation: T loads 0.23 ms +
    // epilogue loads 0.31 ms + FFTs 0.20 ms + rest 0.22 ms = the 1.06 ms total at 4096^2 x 8),
    // and registers allow only one workgroup per CU: so the epilogue operands x and dot_with2
    // are requested right after the odd-bin strided loads and arrive while that transform runs.
    const size_t rowoff = ((size_t)bl * d.nx + (i0 + g)) * d.ny;
    const V2* xr = reinterpret_cast<const V2*>(x + rowoff) + t;
    const V2* br = beam ? reinterpret_cast<const V2*>(beam + rowoff) + t : nullptr;
    const V2* dr = dot_with ? reinterpret_cast<const V2*>(dot_with + rowoff) + t : nullptr;
    const V2* dr2 = dot_with2 ? reinterpret_cast<const V2*>(dot_with2 + rowoff) + t : nullptr;
    V2* orow = reinterpret_cast<V2*>(out + rowoff) + t;
    const bool dot_is_x = dot_with == x;            // PCG: <p, A p> with x = p
    // fp64: 2 x 32 more registers for the prefetched operands do not exist under the 128-VGPR cap of
    // a 1024-thread workgroup (it spilled 80-94 registers); they are read in the epilogue instead
#ifndef PFB_INV_PREF64
#define PFB_INV_PREF64 2
#endif
#ifndef PFB_INV_PREF32E16    // x / x and r prefetched there: 188 / 492 B of scratch, 1.29 / 1.63 ms
#define PFB_INV_PREF32E16 0
#endif
    constexpr bool BIG64 = sizeof(T) == 8 && E >= 16;           // 64-register operands: x only / none (knob)
    constexpr bool BIG32 = sizeof(T) == 4 && E >= 16 && L >= 4096;      // the same question for fp32 at 16 elements per thread
    constexpr bool PREF = BIG32 ? PFB_INV_PREF32E16 >= 1 : (sizeof(T) == 4 || (NT < 1024 && (!BIG64 || PFB_INV_PREF64 >= 1)));
    constexpr bool PREFR = PREF && (BIG32 ? PFB_INV_PREF32E16 >= 2 : (!BIG64 || PFB_INV_PREF64 >= 2));
    V2 xq[PREF ? E : 1], rq[PREFR ? E : 1];
    row_inv_phase<T, L, E, 1>([&] {
        if constexpr (PREF) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                xq[j] = xr[TPB * j];
                if constexpr (PREFR) { if (dr2) rq[j] = dr2[TPB * j]; }
            }
        }
    }, Tb, twQ, ltw, lds0, lds, d.nx, i0, t, vv);
    // z[n] = e[n] + conj(w_M^n) o[n] ;  y[2n] = Re z, y[2n+1] = Im z
    double acc[3] = {0.0, 0.0, 0.0};       // <dot_with,out>, <dot_with2,out>, <out,out>
    const cplx<T>* tm = twM + t;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const cplx<T> zz = ev[j] + mulc(vv[j], tm[TPB * j]);
        V2 val;
        val.x = zz.x * scale;
        val.y = zz.y * scale;
        if (br) { const V2 b = br[TPB * j]; val.x *= b.x; val.y *= b.y; }
        V2 xx;
        if constexpr (PREF) xx = xq[j]; else xx = xr[TPB * j];
        val.x += sigmainv * xx.x;
        val.y += sigmainv * xx.y;
        orow[TPB * j] = val;
        if (dr) {
            V2 dw = xx;
            if (!dot_is_x) dw = dr[TPB * j];
            acc[0] += (double)dw.x * (double)val.x + (double)dw.y * (double)val.y;
            if (dr2) {
                V2 d2;
                if constexpr (PREFR) d2 = rq[j]; else d2 = dr2[TPB * j];
                acc[1] += (double)d2.x * (double)val.x + (double)d2.y * (double)val.y;
            }
            acc[2] += (double)val.x * (double)val.x + (double)val.y * (double)val.y;
        }
    }
    if (dot_with) {
        __syncthreads();
        block_sum<3>(acc, red);
        if (threadIdx.x == 0) {
            const size_t np = (size_t)gridDim.x * gridDim.y, k = (size_t)bl * gridDim.x + rg;
            if (tail.S) { pcg_partial_store(partials + k, acc[0]); pcg_partial_store(partials + np + k, acc[1]); pcg_partial_store(partials + 2 * np + k, acc[2]); }
            else { partials[k] = acc[0]; partials[np + k] = acc[1]; partials[2 * np + k] = acc[2]; }
        }
        // PCG: the last workgroup to get here sums the partials and does the iteration's scalar bookkeeping
        pcg_tail(tail, partials, (int)(gridDim.x * gridDim.y), gridDim.x * gridDim.y, red);
    }
}

// ------------------------------------------------- row inverse, persistent + pipelined
// The plain kernel above keeps ONE 1024-thread workgroup per CU (registers), so its phases
// (strided T loads, FFT, epilogue loads, stores) run strictly one after the other.  This
// variant keeps the workgroup resident, loops over its row tiles and software-pipelines the
// strided loads: the 16-byte pieces of the NEXT phase (odd bins of this tile / even bins of
// the next tile) are requested into registers right after the current phase's pieces were
// scattered into the LDS and land while that phase's FFT runs.  For this to work no other
// global load may sit between the request and the FFT's end (s_waitcnt vmcnt retires in
// order), so w_M^n lives in the LDS as well (w_Q^(2m+1) = w_M^m * w_Q), and the even-bin
// result is parked in the LDS (not registers) while the odd-bin transform runs.
#ifndef PFB_INV_BARUP
#define PFB_INV_BARUP 1
#endif
// PFB_INV_ROT: the workgroup -> tile map of k_row_inv_pow2p rotates by that many workgroups per trip.  With the static map
// (tile = workgroup + trip x grid) a workgroup -- and its whole XCD, workgroup b runs on XCD b mod 8 -- reads the SAME 128 bytes
// of every 1-KB stretch of a column block for the whole launch, and one XCD ran 13-15 % behind the others from the first trip to
// the last (profiles/r03_z_phase_stamps_4096x8_f32.md, per-XCD table): the launch lasts as long as its slowest workgroup.
// Rotating gives every workgroup every row residue in turn: 8 x 4096^2 fp32 0.614 -> 0.541 ms, 4 x 4096^2 fp64 0.716 -> 0.581,
// 2 x 8192^2 fp64 1.644 -> 1.450, fp32 0.749 -> 0.679, with beam 0.720 -> 0.635; bit-identical results
// (profiles/r03_ab_tile_rotation.md).  The forward rows and the column kernel show no such spread and no effect.
#ifndef PFB_INV_ROT
#define PFB_INV_ROT 3
#endif
template <typename T, int L, int E>
struct InvP {
    using C = RowCfg<T, L, true>;
    using F = RegFft<T, L, E, C::WAVE, 0, true>;
    // fp32: always a full 1024-thread workgroup: 4 rows at L = 2048, 8 rows (128-byte pieces) at L = 1024.
    // fp64: two rows = 512 threads (the exchange buffers of four complex128 rows alone fill the LDS): with
    // one such workgroup per CU every thread may use 256 VGPRs, so the even-bin result stays in registers
    // (PARK = false) and the prefetched pieces / operands fit as well.
    static constexpr bool PARK = sizeof(T) == 4 && E < 16;     // (E = 16: eight rows fill the LDS, the even-bin result stays in registers)
#ifndef PFB_INV_HALF
#define PFB_INV_HALF 0          // experiment: fp32 16-element tiles on 512 threads, two workgroups per CU (de-phased barriers)
#endif
    static constexpr bool HALF = PFB_INV_HALF && sizeof(T) == 4 && E >= 16 && L == 2048;
    static constexpr int WG_PER_CU = HALF ? 2 : 1;
    static constexpr int G = (sizeof(T) == 4 && !HALF ? 1024 : 512) / F::TPB;
    static constexpr int NT = G * F::TPB;
    static constexpr int STRIDE = F::LDS_ELEMS + 4;
    static constexpr int NVB = FastCfg<T>::NVB;
    static constexpr int NBE = (L + NVB) / NVB;
    static constexpr int NBO = L / NVB;
    static constexpr int BSTEP = NT / G;
    static constexpr int NITE = (NBE + BSTEP - 1) / BSTEP;
    static constexpr int NITO = (NBO + BSTEP - 1) / BSTEP;
    static constexpr int PTWP = (F::PTWC + 1) & ~1;
    // 4096-point fp32 rows (8192-pixel lines), 16 elements per thread, 4 rows: the exchange buffers of four rows leave no
    // room for the full table w_M^n (32 KB) nor for the parked even-bin result.  SMT: only w_M^t, t < TPB, is kept and
    // w_M^(t + TPB j) = w_M^t root32(j) (M = 2 L = 32 TPB).  OPF = false: the epilogue operands (x, dot_with2, beam) are
    // NOT prefetched into registers (vv + ev + the strided pieces already take 100 of the 128) but read where they are
    // used; what the persistent kernel still buys over the plain one are the tables loaded once and the strided pieces of
    // the next phase / next tile in flight during every transform.
    static constexpr bool SMT = E >= 16 && (L >= 4096 || sizeof(T) == 8 || HALF);
    static constexpr bool OPF = E < 16;
    // NXT: the NEXT tile's even-bin pieces are requested during the odd-bin transform and stay in flight across the
    // epilogue (always with OPF; without it only where registers remain: the 512-thread fp64 tiles have 256)
#ifndef PFB_INV_NXT64
#define PFB_INV_NXT64 1          // 2 x 8192^2 fp64: 1.376 -> 1.195 ms (239 VGPRs, no scratch).  It had measured a tie (1.612 / 1.618)
#endif                           // while one XCD set the pace of every variant (PFB_INV_ROT)
    static constexpr bool NXT = OPF || (PFB_INV_NXT64 && sizeof(T) == 8);
    // NXE: neither -- but the next tile's even-bin pieces are requested at the START of the epilogue (the transforms are
    // over, their temporaries gone) instead of at the top of the next trip, where they were waited for right away
    // (profiles/r03_q_phase_stamps_8192x2_*: 8.4 us of a 45 us trip)
#ifndef PFB_INV_HOIST_B
#define PFB_INV_HOIST_B 8
#endif
#ifndef PFB_INV_NXE
#define PFB_INV_NXE 0           // measured: 1.955 vs 1.639 ms per 2 x 8192^2 fp64 (the pieces queue behind the epilogue's own loads): off
#endif
    // HOIST32 (default): the fp32 16-element tiles (4096-point rows) read their operand rows in two batches of 8 samples and
    // request the next tile's even-bin pieces behind the second batch: 2 x 8192^2 fp32 0.682 -> 0.665 ms (no change with beam)
#ifndef PFB_INV_HOIST32
#define PFB_INV_HOIST32 1
#endif
    static constexpr bool NXE = !NXT && ((PFB_INV_NXE && (sizeof(T) == 8 || PFB_INV_NXE > 1)) || (PFB_INV_HOIST32 && sizeof(T) == 4));
    // HOIST (fp32 NXE): all operand rows of the tile requested at once behind the combine loop (the even-bin result is dead
    // by then), the next tile's even-bin pieces half way through the loop that consumes them, in the registers it has freed
    static constexpr bool HOIST = NXE && sizeof(T) == 4;
    // BARUP: the barrier between the odd-bin transform's last exchange and the next tile's scatter sits right behind that
    // transform instead of at the top of the next trip (8 x 4096^2 fp32 0.6293 -> 0.6229 ms, 2 x 8192^2 fp64 1.673 -> 1.643); not
    // for the fp32 tiles that request a tile's pieces at the top of its own trip -- there the barrier was the only thing
    // those loads were in flight across (0.706 -> 0.732)
    static constexpr bool BARUP = PFB_INV_BARUP && (NXT || sizeof(T) == 8 || (HOIST && PFB_INV_HOIST32 > 1));
    static constexpr int NTM = SMT ? F::TPB : L;
    static constexpr size_t LDS = 384 + sizeof(cplx<T>) * ((size_t)PTWP + NTM + (size_t)G * STRIDE + (PARK ? (size_t)G * L : 0));
    static constexpr bool OK = LDS <= (size_t)160 * 1024 && !C::WAVE && (!SMT || 32 * F::TPB == 2 * L) &&
                               (sizeof(T) == 4 ? (NT == (HALF ? 512 : 1024) && G >= 4 && G <= 16) : (NT == 512 && (G == 2 || G == 4) && L >= 1024));
    __device__ __forceinline__ static cplx<T> tw_row(const cplx<T>* ltm, int t, int j) {     // w_M^(t + TPB j)
        if constexpr (SMT) return j == 0 ? ltm[t] : ltm[t] * root32<T>(j);
        else return ltm[t + F::TPB * j];
    }
};

// elements per thread of the PERSISTENT inverse row kernel: RowCfg's, except fp64 rows of 4096 points (C5) -- there the
// plain kernel is a 2-row, 1024-thread, 8-elements tile under the 128-VGPR cap, the persistent one a 2-row, 512-thread,
// 16-elements tile (174-199 VGPRs, no scratch; small tables, operands read in place): 2.30 -> 1.61 ms per 2 x 8192^2.
// (The plain kernel at 16 elements spills 648 B and takes 3.0 ms: the two kernels keep separate pass tables.)
#ifndef PFB_INVPE64_MINL       // smallest fp64 row length that takes the 16-element persistent tile
#define PFB_INVPE64_MINL 4096
#endif
template <typename T, int L> struct InvPE {
    static constexpr int E = (sizeof(T) == 8 && L >= PFB_INVPE64_MINL) ? 16 : RowCfg<T, L, true>::E;
};

#ifndef PFB_INV_LIN
#define PFB_INV_LIN 1
#endif
// PFB_INV_EARLY bit 0: the first pass's slice of loads (odd-bin pieces / dot_with2 rows) is requested right behind the scatter
// instead, so that something is in flight across the barrier and inv_build: 0.6075 -> 0.5999 ms per 8 x 4096^2 fp32, 0.7415 ->
// 0.7216 per 4 x 4096^2 fp64.  Bits 1-3 (all odd-bin pieces there / x rows behind the even transform / next tile's pieces in the
// last pass only) measured no better (profiles/r03_ab_inv_ablation.md).
#ifndef PFB_INV_EARLY
#define PFB_INV_EARLY 1
#endif
#ifndef PFB_INV_ABL             // ablation builds of k_row_inv_pow2p (timing only, results are wrong): 1 no strided pieces,
#define PFB_INV_ABL 0           // 2 no operand rows, 4 no result stores, 8 no transforms
#endif
template <int NP, typename H> __device__ __forceinline__ void abl_hooks(H&& h) {
    if constexpr (NP > 0) { abl_hooks<NP - 1>(h); h(std::integral_constant<int, NP - 1>{}); }
}
template <typename F, typename V, typename C, typename H>
__device__ __forceinline__ void inv_run(V& vv, C* lds, int t, const C* ltw, H&& h) {
    if constexpr ((PFB_INV_ABL & 8) != 0) abl_hooks<F::NPASS>(h);
    else F::template run<true>(vv, lds, t, ltw, h);
}
// strength-reduced addressing of the strided pieces (see fwdp_post_lin): the blocks of a thread are BSTEP apart, so the
// load address is a workgroup-uniform base (band, parity, step) + a 32-bit per-thread offset that never changes, and
// the padded LDS index of the scatter is one base + compile-time offsets.  Regular whenever BSTEP divides the odd-bin
// block count; the even bins then have exactly one more block (the Nyquist column), fetched by every thread in the last step.
template <typename T, int L, int E>
constexpr bool inv_lin_ok() {
    using P = InvP<T, L, E>;
    // (fp32 only: 1902 instead of 2147 instructions per thread and tile, 0.613 vs 0.616 ms -- the inverse kernel is not
    // issue bound the way the forward sweep was; the fp64 tiles measured 1 % slower with it)
    return PFB_INV_LIN && sizeof(T) == 4 && P::NBO % P::BSTEP == 0 && P::NBE == P::NBO + 1 && P::NITO * P::BSTEP == P::NBO &&
           P::NITE == P::NITO + 1 && (P::NVB * P::BSTEP) % 16 == 0 && P::F::TPB % 16 == 0;
}
template <typename T, int L, int E, int PAR, int K0, int K1>
__device__ __forceinline__ void inv_issue_slice(const cplx<T>* __restrict__ Tb, int nx, int i0, int rr, int bi,
                                                Blk<T, FastCfg<T>::NVB> (&y)[InvP<T, L, E>::NITE]);
template <typename T, int L, int E, int PAR>
__device__ __forceinline__ void inv_issue(const cplx<T>* __restrict__ Tb, int nx, int i0, int rr, int bi,
                                          Blk<T, FastCfg<T>::NVB> (&y)[InvP<T, L, E>::NITE]) {
    using P = InvP<T, L, E>;
    constexpr int NBP = PAR ? P::NBO : P::NBE, NIT = PAR ? P::NITO : P::NITE;
    if constexpr ((PFB_INV_ABL & 1) != 0) {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            for (int c = 0; c < P::NVB; ++c) y[k].c[c] = cplx<T>((T)(bi + k), (T)rr);
        return;
    }
    if constexpr (inv_lin_ok<T, L, E>()) {
        inv_issue_slice<T, L, E, PAR, 0, NIT>(Tb, nx, i0, rr, bi, y);
        return;
    }
    const cplx<T>* Tp = Tb + ((size_t)(PAR ? P::NBE : 0) * nx + i0 + rr) * P::NVB;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        int b = bi + k * P::BSTEP;
        if (b >= NBP) b = NBP - 1;                 // clamped (not skipped): the load count stays static
        y[k] = loadb<T, P::NVB>(Tp + (size_t)b * nx * P::NVB);
    }
}

// loads [K0, K1) of the same set (a compile-time slice: one FFT pass issues a few of them, see RegFft's pass hook)
template <typename T, int L, int E, int PAR, int K0, int K1>
__device__ __forceinline__ void inv_issue_slice(const cplx<T>* __restrict__ Tb, int nx, int i0, int rr, int bi,
                                                Blk<T, FastCfg<T>::NVB> (&y)[InvP<T, L, E>::NITE]) {
    using P = InvP<T, L, E>;
    constexpr int NBP = PAR ? P::NBO : P::NBE;
    if constexpr ((PFB_INV_ABL & 1) != 0) {
#pragma unroll
        for (int k = K0; k < K1; ++k)
            for (int c = 0; c < P::NVB; ++c) y[k].c[c] = cplx<T>((T)(bi + k), (T)rr);
        return;
    }
    if constexpr (inv_lin_ok<T, L, E>()) {
        const cplx<T>* ub = Tb + (size_t)(PAR ? P::NBE : 0) * nx * P::NVB;                    // workgroup-uniform
        const size_t ustep = (size_t)P::BSTEP * nx * P::NVB;
        const unsigned vrow = (unsigned)(i0 + rr) * P::NVB, voff = (unsigned)bi * (unsigned)nx * P::NVB + vrow;
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            if (PAR == 0 && k == P::NITE - 1) y[k] = loadb<T, P::NVB>(ub + (size_t)(NBP - 1) * nx * P::NVB + vrow);   // Nyquist block
            else y[k] = loadb<T, P::NVB>(ub + (size_t)k * ustep + voff);
        }
        return;
    }
    const cplx<T>* Tp = Tb + ((size_t)(PAR ? P::NBE : 0) * nx + i0 + rr) * P::NVB;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        int b = bi + k * P::BSTEP;
        if (b >= NBP) b = NBP - 1;
        y[k] = loadb<T, P::NVB>(Tp + (size_t)b * nx * P::NVB);
    }
}

template <typename T, int L, int E, int PAR>
__device__ __forceinline__ void inv_scatter(const Blk<T, FastCfg<T>::NVB> (&y)[InvP<T, L, E>::NITE],
                                            cplx<T>* yr, int bi) {
    using P = InvP<T, L, E>;
    constexpr int NBP = PAR ? P::NBO : P::NBE, NIT = PAR ? P::NITO : P::NITE;
    if constexpr (inv_lin_ok<T, L, E>()) {
        cplx<T>* y0 = yr + P::F::pad(P::NVB * bi);          // pad(NVB bi + h + 16 c) = pad(NVB bi) + h + 17 c
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (PAR == 0 && k == NIT - 1) yr[P::F::pad(L)] = y[k].c[0];      // bin m = L; every thread holds the same value
            else {
#pragma unroll
                for (int h = 0; h < P::NVB; ++h) y0[P::F::cpad(P::NVB * P::BSTEP * k) + h] = y[k].c[h];
            }
        }
        return;
    }
    // no `if (b < NBP)` here: a conditional use lets the compiler SINK the load into the branch
    // (load + vmcnt(0) back to back).  Out-of-range threads hold a copy of the last block
    // (inv_issue clamps the same way) and store the same values to the same place.
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        int b = bi + k * P::BSTEP;
        if (b >= NBP) b = NBP - 1;
#pragma unroll
        for (int h = 0; h < P::NVB; ++h)
            if (PAR || P::NVB * b + h <= L) yr[P::F::pad(P::NVB * b + h)] = y[k].c[h];
    }
}

template <typename T, int L, int E, int PAR>
__device__ __forceinline__ void inv_build(const cplx<T>* lds, const cplx<T>* ltm, cplx<T> wq1, int t,
                                          cplx<T> (&vv)[E]) {
    using F = typename InvP<T, L, E>::F;
    if constexpr (inv_lin_ok<T, L, E>()) {
        const cplx<T>* pa = lds + F::pad(t);                                  // Y[t + TPB j]      = pa[cpad(TPB j)]
        const cplx<T>* pb = lds + F::pad(L - PAR - t - F::TPB * (E - 1));     // partner of step j = pb[cpad(TPB (E-1-j))]
#pragma unroll
        for (int j = 0; j < E; ++j) {
            cplx<T> yv = pa[F::cpad(F::TPB * j)];
            cplx<T> ym = pb[F::cpad(F::TPB * (E - 1 - j))];
            if (!PAR && j == 0) { if (t == 0) { yv.y = 0; ym.y = 0; } }
            cplx<T> w = InvP<T, L, E>::tw_row(ltm, t, j);
            if (PAR) w = w * wq1;
            vv[j] = addrot<true>(addc(yv, ym), mulc(subc(yv, ym), w));
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int m = t + F::TPB * j;
        cplx<T> yv = lds[F::pad(m)];
        cplx<T> ym = lds[F::pad(PAR ? (L - 1 - m) : (L - m))];
        if (!PAR && m == 0) { yv.y = 0; ym.y = 0; }
        cplx<T> w = InvP<T, L, E>::tw_row(ltm, t, j);          // w_M^m, m = t + TPB j
        if (PAR) w = w * wq1;
        vv[j] = addrot<true>(addc(yv, ym), mulc(subc(yv, ym), w));
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

// MODE 0: no inner products; 1: <x, out>, <out, out>; 2: also <dot_with2, out> (the PCG call)
// SPR: the strided pieces and the epilogue operands are requested a few per FFT PASS (RegFft pass hook) instead of
// in two bursts behind the scatters: a burst of 21 loads per thread parks all 16 waves at issue for 7.6 us of a
// 22.6 us trip (profiles/r02_a_phase_stamps_*).  Per trip, in issue order (vmcnt retires in order):
//     IFFT (even)  <- y(odd bins of this tile), then x [, beam]   of this tile's rows
//     IFFT (odd)   <- dot_with2 rows, then y(even bins of the NEXT tile)
template <typename T, int L, int E, int MODE, bool BEAM, bool SPR = false>
__global__ void __launch_bounds__((InvP<T, L, E>::NT), (InvP<T, L, E>::NT / 256 * InvP<T, L, E>::WG_PER_CU))
k_row_inv_pow2p(const cplx<T>* __restrict__ Tw, const cplx<T>* __restrict__ twM,
                const cplx<T>* __restrict__ ptw, const T* __restrict__ x, const T* __restrict__ beam,
                const T* __restrict__ dot_with2, T* __restrict__ out,
                double* __restrict__ partials, FastDims d, int band0, int tiles_per_band, int ntiles,
                T scale, T sigmainv, cplx<T> wq1, int defer_stores, PcgTail tail) {
    using P = InvP<T, L, E>;
    using F = typename P::F;
    constexpr int TPB = F::TPB, G = P::G, NT = P::NT;
    // (the batched-operand variant of the fp32 16-element tiles keeps 12 B of scratch with beam + two inner products: that
    // instantiation stays on the plain order -- a tile's pieces requested at the top of its own trip)
    constexpr bool NXEK = P::NXE && !(P::HOIST && BEAM && MODE == 2), HOISTK = P::HOIST && NXEK;
    // (likewise the fp64 16-element tiles without the per-pass request schedule -- PFB_SPREAD=0, an A/B fallback -- keep 12 B
    // of scratch with the next tile's pieces prefetched: that instantiation requests them at the top of the trip)
    constexpr bool NXTK = P::NXT && (P::OPF || SPR);
    constexpr int NP = F::NPASS, NPA = NP / 2 > 0 ? NP / 2 : 1, NPB = NP - NPA > 0 ? NP - NPA : 1;
    using V2 = typename vec2<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);
    cplx<T>* ltw = reinterpret_cast<cplx<T>*>(smem + 384);
    cplx<T>* ltm = ltw + P::PTWP;
    cplx<T>* lds0 = ltm + P::NTM;
    cplx<T>* park = lds0 + (size_t)G * P::STRIDE;
    int vb = blockIdx.x;
    if (vb >= ntiles) return;
    for (int k = threadIdx.x; k < F::PTWC; k += NT) ltw[k] = ptw[k];
    for (int k = threadIdx.x; k < P::NTM; k += NT) ltm[k] = twM[k];
    constexpr int SH = 128 / (G * (int)sizeof(cplx<T>) * P::NVB) > 1 ? 128 / (G * (int)sizeof(cplx<T>) * P::NVB) : 1;
    const bool pairing = SH > 1 && (tiles_per_band % (8 * SH)) == 0;
    auto tile = [&](int v, int& bl, int& i0) {
        bl = v / tiles_per_band;
        int rg = v - bl * tiles_per_band;
        if (pairing) {                      // see k_row_inv_pow2: blocks b, b+8, .. share an XCD's L2: give
            const int q = rg / (8 * SH), rem = rg % (8 * SH);       // THEM the SH pieces of one 128-byte line
            rg = SH * (q * 8 + (rem & 7)) + (rem >> 3);
        }
        i0 = rg * G;
    };
    int bl, i0;
    tile(vb, bl, i0);
    Blk<T, P::NVB> y[P::NITE];
    // OPF kernels keep the NEXT tile's even-bin pieces in flight across the epilogue; without the registers for that
    // (!OPF: 16 elements per thread) a tile's even-bin pieces are requested at the top of its own trip instead
    if constexpr (NXTK || NXEK)
        inv_issue<T, L, E, 0>(Tw + (size_t)(band0 + bl) * d.T_band, d.nx, i0, threadIdx.x % G, threadIdx.x / G, y);
    double acc[3] = {0.0, 0.0, 0.0};
    // deferred stores (SPR, fp32): a tile's output rows stay in registers and are written two per pass of the NEXT
    // tile's even-bin transform instead of in one burst behind the epilogue
    const bool dst = SPR && P::PARK && defer_stores;
    V2 ov[(SPR && P::PARK) ? E : 1];
    V2* oprev = nullptr;
    for (int sit = 0;; ++sit) {
        STAMP(2, sit, 0);
        int vbn;
        if constexpr (PFB_INV_ROT != 0) {
            // the workgroup -> tile map rotates by PFB_INV_ROT workgroups per trip: over its trips a workgroup sees the row
            // groups of every XCD, not always the same 128 bytes of each 1-KB stretch of a column block
            const int s1 = sit + 1;
            const int cand = (int)(((unsigned)blockIdx.x + (unsigned)s1 * (unsigned)PFB_INV_ROT) % gridDim.x) + s1 * (int)gridDim.x;
            vbn = cand < ntiles ? cand : vb;
        } else {
            vbn = vb + (int)gridDim.x < ntiles ? vb + (int)gridDim.x : vb;   // last tile: harmless repeat
        }
        int bln, i0n;
        tile(vbn, bln, i0n);
        const cplx<T>* Tb = Tw + (size_t)(band0 + bl) * d.T_band;
        if constexpr (!NXTK && !NXEK) {
            const int tid = launder((int)threadIdx.x);
            inv_issue<T, L, E, 0>(Tb, d.nx, i0, tid % G, tid / G, y);
        }
        cplx<T> vv[E], ev[P::PARK ? 1 : E];
        constexpr bool OPF = P::OPF;             // epilogue operands prefetched into registers
        V2 xq[OPF ? E : 1], rq[(OPF && MODE == 2) ? E : 1], bq[(OPF && BEAM) ? E : 1];
        // ---- even bins
        {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB, rr = tid % G, bi = tid / G;
            cplx<T>* lds = lds0 + (size_t)g * P::STRIDE;
            if (!P::BARUP || sit == 0) __syncthreads();      // (BARUP: the barrier that protects lds0 sits behind the odd-bin transform)
            STAMP(2, sit, 1);
            inv_scatter<T, L, E, 0>(y, lds0 + (size_t)rr * P::STRIDE, bi);
            STAMP(2, sit, 2);
            if constexpr (!SPR) inv_issue<T, L, E, 1>(Tb, d.nx, i0, rr, bi, y);
            else if constexpr ((PFB_INV_EARLY & 2) != 0) inv_issue<T, L, E, 1>(Tb, d.nx, i0, rr, bi, y);
            else if constexpr ((PFB_INV_EARLY & 1) != 0) inv_issue_slice<T, L, E, 1, 0, P::NITO / NPA>(Tb, d.nx, i0, rr, bi, y);   // pass 0's slice: in flight across the barrier and inv_build
            __syncthreads();
            STAMP(2, sit, 3);
            inv_build<T, L, E, 0>(lds, ltm, wq1, t, vv);
            STAMP(2, sit, 4);
            if constexpr (SPR) {
                const size_t rowoff = ((size_t)bl * d.nx + (i0 + g)) * d.ny;
                const V2* xr = reinterpret_cast<const V2*>(x + rowoff) + t;
                const V2* br = BEAM ? reinterpret_cast<const V2*>(beam + rowoff) + t : nullptr;
                inv_run<F>(vv, lds, t, ltw, [&](auto k) {
                    constexpr int K = decltype(k)::value;
                    if constexpr (P::PARK) {
                        if (dst && oprev) {
#pragma unroll
                            for (int j = (K * E) / NP; j < ((K + 1) * E) / NP; ++j) { if (!(PFB_INV_ABL & 4) || scale == (T)123456) oprev[TPB * j] = ov[j]; }
                        }
                    }
                    if constexpr (K < NPA) {            // first passes: the odd-bin pieces (needed first)
                        if constexpr (!((PFB_INV_EARLY & 1) && K == 0) && !(PFB_INV_EARLY & 2))
                        inv_issue_slice<T, L, E, 1, (K * P::NITO) / NPA, ((K + 1) * P::NITO) / NPA>(Tb, d.nx, i0, rr, bi, y);
                    } else if constexpr (OPF && !(PFB_INV_EARLY & 4)) {         // then this tile's x (and beam) rows for the epilogue
                        constexpr int KK = K - NPA;
#pragma unroll
                        for (int j = (KK * E) / NPB; j < ((KK + 1) * E) / NPB; ++j) {
                            if constexpr ((PFB_INV_ABL & 2) == 0) xq[j] = xr[TPB * j]; else { xq[j].x = (T)(t + j); xq[j].y = (T)1; }
                            if constexpr (BEAM) bq[j] = br[TPB * j];
                        }
                    }
                });
            } else {
                F::template run<true>(vv, lds, t, ltw);
            }
            STAMP(2, sit, 5);
            if constexpr (SPR && OPF && (PFB_INV_EARLY & 4) != 0) {      // x rows requested here: in flight across park, scatter, build
                const size_t rowoff = ((size_t)bl * d.nx + (i0 + g)) * d.ny;
                const V2* xr = reinterpret_cast<const V2*>(x + rowoff) + t;
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    xq[j] = xr[TPB * j];
                    if constexpr (BEAM) bq[j] = reinterpret_cast<const V2*>(beam + rowoff)[t + TPB * j];
                }
            }
            if constexpr (P::BARUP && (PFB_INV_BARUP & 2) != 0) __syncthreads();   // (the even transform's readers are done: before the park, not behind it)
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if constexpr (P::PARK) park[j * NT + tid] = vv[j]; else ev[j] = vv[j];
            }
        }
        // ---- odd bins
        {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB, rr = tid % G, bi = tid / G;
            cplx<T>* lds = lds0 + (size_t)g * P::STRIDE;
            if constexpr (!(P::BARUP && (PFB_INV_BARUP & 2) != 0)) __syncthreads();
            STAMP(2, sit, 6);
            inv_scatter<T, L, E, 1>(y, lds0 + (size_t)rr * P::STRIDE, bi);
            STAMP(2, sit, 7);
            const size_t rowoff = ((size_t)bl * d.nx + (i0 + g)) * d.ny;
            const V2* dr2 = reinterpret_cast<const V2*>(dot_with2 + rowoff) + t;
            const cplx<T>* Tbn = Tw + (size_t)(band0 + bln) * d.T_band;
            if constexpr (!SPR) {
                if constexpr (OPF) {
                    const V2* xr = reinterpret_cast<const V2*>(x + rowoff) + t;
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        xq[j] = xr[TPB * j];
                        if constexpr (MODE == 2) rq[j] = dr2[TPB * j];
                    }
                    if constexpr (BEAM) {
                        const V2* br = reinterpret_cast<const V2*>(beam + rowoff) + t;
#pragma unroll
                        for (int j = 0; j < E; ++j) bq[j] = br[TPB * j];
                    }
                }
                if constexpr (NXTK) inv_issue<T, L, E, 0>(Tbn, d.nx, i0n, rr, bi, y);
            } else if constexpr ((PFB_INV_EARLY & 1) != 0 && MODE == 2 && OPF) {
#pragma unroll
                for (int j = 0; j < E / NPA; ++j) rq[j] = dr2[TPB * j];
            }
            __syncthreads();
            STAMP(2, sit, 8);
            inv_build<T, L, E, 1>(lds, ltm, wq1, t, vv);
            STAMP(2, sit, 9);
            if constexpr (SPR) {
                inv_run<F>(vv, lds, t, ltw, [&](auto k) {
                    constexpr int K = decltype(k)::value;
                    if constexpr (MODE == 2 && OPF) {
                        if constexpr (K < NPA) {        // dot_with2 rows first: the epilogue waits for them
                            if constexpr (!((PFB_INV_EARLY & 1) && K == 0))
#pragma unroll
                            for (int j = (K * E) / NPA; j < ((K + 1) * E) / NPA; ++j) { if constexpr ((PFB_INV_ABL & 2) == 0) rq[j] = dr2[TPB * j]; else { rq[j].x = (T)(t - j); rq[j].y = (T)2; } }
                        } else {                        // the next tile's even-bin pieces stay in flight past the epilogue
                            constexpr int KK = K - NPA;
                            if constexpr ((PFB_INV_EARLY & 8) != 0) { if constexpr (K == NP - 1) inv_issue<T, L, E, 0>(Tbn, d.nx, i0n, rr, bi, y); }
                            else
                            inv_issue_slice<T, L, E, 0, (KK * P::NITE) / NPB, ((KK + 1) * P::NITE) / NPB>(Tbn, d.nx, i0n, rr, bi, y);
                        }
                    } else if constexpr (NXTK) {
                        inv_issue_slice<T, L, E, 0, (K * P::NITE) / NP, ((K + 1) * P::NITE) / NP>(Tbn, d.nx, i0n, rr, bi, y);
                    }
                });
            } else {
                F::template run<true>(vv, lds, t, ltw);
            }
            STAMP(2, sit, 10);
            // the last exchange's readers are done BEFORE the epilogue instead of at the top of the next trip: the waves are
            // still in step here (cheap), and a wave that is through with its epilogue scatters the next tile's pieces while
            // the others finish theirs -- the skew is absorbed by the barrier behind the scatter
            if constexpr (P::BARUP) __syncthreads();
        }
        if constexpr (NXEK && !HOISTK) {
            const int tid = launder((int)threadIdx.x);
            inv_issue<T, L, E, 0>(Tw + (size_t)(band0 + bln) * d.T_band, d.nx, i0n, tid % G, tid / G, y);
        }
        // ---- z[n] = e[n] + conj(w_M^n) o[n] ;  y[2n] = Re z, y[2n+1] = Im z
        {
            const int tid = launder((int)threadIdx.x);
            int g = tid / TPB;
            const int t = tid % TPB;
            // a row is shared by whole waves (TPB a multiple of 64): its index is wave-uniform -- said explicitly, the
            // row bases live in SGPRs and the 2-KB steps of the unrolled operand / result streams cost scalar adds instead
            // of a 64-bit VGPR address per stream and sample
            if constexpr (TPB % 64 == 0 && !P::OPF) g = __builtin_amdgcn_readfirstlane(g);
            const size_t rowoff = ((size_t)bl * d.nx + (i0 + g)) * d.ny;
            V2* orow = reinterpret_cast<V2*>(out + rowoff) + t;
            [[maybe_unused]] const V2* xr_e = reinterpret_cast<const V2*>(x + rowoff) + t;
            [[maybe_unused]] const V2* br_e = BEAM ? reinterpret_cast<const V2*>(beam + rowoff) + t : nullptr;
            [[maybe_unused]] const V2* dr_e = MODE == 2 ? reinterpret_cast<const V2*>(dot_with2 + rowoff) + t : nullptr;
            if constexpr (!OPF) {
                // operands read in place: combine the two parities FIRST (no load involved), so that the even-bin result
                // is dead before the operand streams start -- 32 registers less across the loop below
#pragma unroll
                for (int j = 0; j < E; ++j) vv[j] = ev[j] + mulc(vv[j], P::tw_row(ltm, t, j));
                __builtin_amdgcn_sched_barrier(0);
            }
            [[maybe_unused]] V2 xh[HOISTK ? E : 1], rh[(HOISTK && MODE == 2) ? E : 1], bh[(HOISTK && BEAM) ? E : 1];

#pragma unroll
            for (int j = 0; j < E; ++j) {
                cplx<T> zz;
                if constexpr (!OPF) zz = vv[j];
                else {
                    cplx<T> e0;
                    if constexpr (P::PARK) e0 = park[j * NT + tid]; else e0 = ev[j];
                    zz = e0 + mulc(vv[j], P::tw_row(ltm, t, j));
                }
                if constexpr (HOISTK) {
                    // operand rows in batches of HB samples; behind the LAST batch the next tile's even-bin pieces (in order:
                    // nothing the epilogue still waits for is queued behind them)
                    constexpr int HB = PFB_INV_HOIST_B;
                    if (j % HB == 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = j; q < j + HB && q < E; ++q) {
                            xh[q] = xr_e[TPB * q];
                            if constexpr (MODE == 2) rh[q] = dr_e[TPB * q];
                            if constexpr (BEAM) bh[q] = br_e[TPB * q];
                        }
                        if (j + HB >= E)
                            inv_issue<T, L, E, 0>(Tw + (size_t)(band0 + bln) * d.T_band, d.nx, i0n, tid % G, tid / G, y);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                V2 val;
                V2 xx;
                [[maybe_unused]] V2 bb;
                if constexpr (OPF) { xx = xq[j]; if constexpr (BEAM) bb = bq[j]; }
                else if constexpr (HOISTK) { xx = xh[j]; if constexpr (BEAM) bb = bh[j]; }
                else if constexpr ((PFB_INV_ABL & 2) != 0) { xx.x = (T)(t + j); xx.y = (T)1; if constexpr (BEAM) bb = br_e[TPB * j]; }
                else { xx = xr_e[TPB * j]; if constexpr (BEAM) bb = br_e[TPB * j]; }
                if constexpr (BEAM) {
                    val.x = zz.x * scale * bb.x + sigmainv * xx.x;
                    val.y = zz.y * scale * bb.y + sigmainv * xx.y;
                } else {
                    val.x = zz.x * scale + sigmainv * xx.x;
                    val.y = zz.y * scale + sigmainv * xx.y;
                }
                if constexpr (SPR && P::PARK) { if (dst) ov[j] = val; else orow[TPB * j] = val; }
                else if constexpr ((PFB_INV_ABL & 4) != 0 && MODE >= 1) { }          // (the inner products keep val alive)
                else orow[TPB * j] = val;
                // operands read in place (!OPF): a few samples at a time, or every x / dot_with2 load of the tile is
                // hoisted to the top of the loop and spills
                if constexpr (!OPF && !HOISTK) { if (sizeof(T) == 4 && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
                if constexpr (MODE >= 1) {
                    acc[0] += (double)xx.x * (double)val.x + (double)xx.y * (double)val.y;
                    if constexpr (MODE == 2) {
                        V2 d2;
                        if constexpr (OPF) d2 = rq[j]; else if constexpr (HOISTK) d2 = rh[j];
                        else if constexpr ((PFB_INV_ABL & 2) != 0) { d2.x = (T)(t - j); d2.y = (T)2; }
                        else d2 = dr_e[TPB * j];
                        acc[1] += (double)d2.x * (double)val.x + (double)d2.y * (double)val.y;
                    }
                    acc[2] += (double)val.x * (double)val.x + (double)val.y * (double)val.y;
                }
            }
        }
        STAMP(2, sit, 11);
        if constexpr (SPR && P::PARK) {
            const int tid = launder((int)threadIdx.x);
            const int g = tid / TPB, t = tid % TPB;
            oprev = reinterpret_cast<V2*>(out + ((size_t)bl * d.nx + (i0 + g)) * d.ny) + t;
        }
        if (vbn == vb) break;
        vb = vbn; bl = bln; i0 = i0n;
    }
    if constexpr (SPR && P::PARK) {
        if (dst && oprev) {
#pragma unroll
            for (int j = 0; j < E; ++j) oprev[TPB * j] = ov[j];
        }
    }
    if constexpr (MODE >= 1) {
        __syncthreads();
        block_sum<3>(acc, red);
        if (threadIdx.x == 0) {           // one slot per WORKGROUP (launcher: last_npartials = grid)
            const size_t np = gridDim.x;
            bool coh = false;
            if constexpr (MODE == 2) coh = tail.S != nullptr;
            if (coh) { pcg_partial_store(partials + blockIdx.x, acc[0]); pcg_partial_store(partials + np + blockIdx.x, acc[1]); pcg_partial_store(partials + 2 * np + blockIdx.x, acc[2]); }
            else { partials[blockIdx.x] = acc[0]; partials[np + blockIdx.x] = acc[1]; partials[2 * np + blockIdx.x] = acc[2]; }
        }
        if constexpr (MODE == 2) pcg_tail(tail, partials, (int)gridDim.x, gridDim.x, red);
    }
}

#endif  // PFB_POW2_REST

#if PFB_STAMP
#if PFB_POW2_COL
int pow2_col_set_stamp(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? PFB_OK : PFB_ERR_HIP;
}
#endif
#if PFB_POW2_REST
int pow2_set_xtra(const void* src_, double* sink) {
    const float* src = (const float*)src_;
    PFB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_xtra_src), &src, sizeof(src)));
    PFB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_xtra_sink), &sink, sizeof(sink)));
    return PFB_OK;
}
int pow2_col_set_stamp(unsigned long long* buf);
int pow2_set_stamp(unsigned long long* buf) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)) != hipSuccess) return PFB_ERR_HIP;
    return pow2_col_set_stamp(buf);
}
#endif
#endif

// size switch helper
#define PFB_POW2_SIZES(X) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096) X(8192)

// -------------------------------------------------------------------- host side
// Grid of the persistent column kernel k_col_pow2p: not all `maxgrid` resident workgroups but as many as give every one of them the same number
// of trips.  `need` is rarely a multiple of the CU count (a band is 2^k + 1 column blocks), and a last trip with a handful
// of workgroups runs at their latency while the rest of the chip idles -- the same work on slightly fewer workgroups is
// bandwidth bound to the end: col 0.816 -> 0.800 ms at 8 x 4096^2 fp32 (253 instead of 256 workgroups), 45.5 -> 42.6 us
// at 1 x 2048^2 (342 instead of 512), 129.3 -> 126.5 us at 1 x 4096^2 (228), 0.889 -> 0.861 ms at 4 x 4096^2 fp64.  Only for
// that kernel (the row kernels' grids divide evenly at the usual sizes, the two-level column kernel loses).  PFB_GRID_BALANCE=0: off.
static inline int balanced_grid(int need, int maxgrid) {
    if (need <= maxgrid) return need;
    static const bool on = [] { const char* e = getenv("PFB_GRID_BALANCE"); return !e || atoi(e); }();
    if (!on) return maxgrid;
    const int trips = (need + maxgrid - 1) / maxgrid;
    return (need + trips - 1) / trips;
}

struct FastTables {            // device tables owned by the plan (stored behind p->fast_tables)
    void* ptw_col;
    void* ptwc_col;            // compact (w only) table of the column transform, copied to LDS
    void* ptw_row;             // forward row kernel (E = 16)
    void* ptw_row_inv;         // inverse row kernel (E = 8): COMPACT table, copied to LDS
    void* ptw_row_inv_p;       // the persistent inverse kernel's, where its elements per thread differ (InvPE); else == ptw_row_inv
    void* twM;                 // exp(-2 pi i n / M), n < L
    int col_persistent;        // PFB_COL_PERSIST (default: auto by size): persistent prefetching column kernel
    void* ptwc_row_fwd;        // compact pass table of the persistent forward row kernel (its own E)
    int fwd_persistent;        // PFB_FWD_PERSIST (default 1): persistent pipelined forward row kernel where it fits
    int inv_persistent;        // PFB_INV_PERSIST (default 1): persistent pipelined inverse row kernel where it fits
    int num_cu;
    void* ptwc_col_x;          // compact pass table of its HS = nx / 2 point sub-transform
    int col_x;                 // two-level column kernel k_col_pow2x + class-major psf_l (default: nx = 8192; PFB_COL_X=0/1 forces)
};
// the column object's entry points (defined under PFB_POW2_COL below)
int pow2_col_set_attr(int dtype, int H);
int pow2_col_launch(pfb_conv_plan* p, const FastTables* ft, int band0, int nb, hipStream_t st);
int pow2_col_fwd_launch(pfb_conv_plan* p, const FastTables* ft, const void* Tq, int band, hipStream_t st);

#if PFB_POW2_REST
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

template <typename T, int N, int E>
static int prep_ptw_compact(void** dev) {
    constexpr int n = ptw_total<N, E>() / 4;
    std::vector<cplx<T>> h(n > 0 ? n : 1);
    if (n > 0) fill_ptw_compact<T, N, E>(h.data());
    PFB_HIP_CHECK(hipMalloc(dev, sizeof(cplx<T>) * h.size()));
    PFB_HIP_CHECK(hipMemcpy(*dev, h.data(), sizeof(cplx<T>) * h.size(), hipMemcpyHostToDevice));
    return PFB_OK;
}

template <typename T, int L>
static int set_invp_attr() {
    constexpr int E = InvPE<T, L>::E;
    if constexpr (InvP<T, L, E>::OK) {
#define PFB_INVATTR(MODE, BM)                                                                           \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_row_inv_pow2p<T, L, E, MODE, BM, false>),      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_row_inv_pow2p<T, L, E, MODE, BM, true>),       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        PFB_INVATTR(0, false); PFB_INVATTR(1, false); PFB_INVATTR(2, false);
        PFB_INVATTR(0, true);  PFB_INVATTR(1, true);  PFB_INVATTR(2, true);
#undef PFB_INVATTR
    }
    return PFB_OK;
}

template <typename T, int L>
static int prep_fwdp(void** table) {
    if constexpr (FwdP<T, L>::OK) {
        int rc = prep_ptw_compact<T, L, FwdP<T, L>::EOK>(table);
        if (rc != PFB_OK) return rc;
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_row_fwd_pow2p<T, L, false, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_row_fwd_pow2p<T, L, true, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_row_fwd_pow2p<T, L, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_row_fwd_pow2p<T, L, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_row_fwd_pow2q<T, L, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_row_fwd_pow2q<T, L, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    return PFB_OK;
}


template <typename T>
static int prep_tables(pfb_conv_plan* p, FastTables* ft) {
    const int H = p->nx, L = p->ny / 2;
    int rc = PFB_ERR_UNSUPPORTED;
    constexpr int lds_max = 160 * 1024;
    switch (H) {
#define X(NN) case NN: rc = prep_ptw<T, NN, ecol<T, NN>()>(&ft->ptw_col);                          \
        if (rc == PFB_OK) rc = prep_ptw_compact<T, NN, ecol<T, NN>()>(&ft->ptwc_col);             \
        if (rc == PFB_OK && NN >= 2048) rc = prep_ptw_compact<T, (NN >= 2048 ? NN / 2 : 1024), ECOLX>(&ft->ptwc_col_x); \
        if (rc == PFB_OK) rc = pow2_col_set_attr(p->dtype, NN); break;
        PFB_POW2_SIZES(X)
#undef X
        default: break;
    }
    if (rc != PFB_OK) return rc;
    rc = PFB_ERR_UNSUPPORTED;
    switch (L) {
#define X(NN) case NN: rc = prep_ptw<T, NN, RowCfg<T, NN, false>::E>(&ft->ptw_row);                   \
        if (rc == PFB_OK) rc = prep_ptw_compact<T, NN, RowCfg<T, NN, true>::E>(&ft->ptw_row_inv);    \
        if (rc == PFB_OK) { if (InvPE<T, NN>::E != RowCfg<T, NN, true>::E) rc = prep_ptw_compact<T, NN, InvPE<T, NN>::E>(&ft->ptw_row_inv_p); \
                            else ft->ptw_row_inv_p = ft->ptw_row_inv; }                            \
        if (rc == PFB_OK) PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_row_fwd_pow2<T, NN, RowCfg<T, NN, false>::E>, \
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));                                   \
        if (rc == PFB_OK) PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_row_inv_pow2<T, NN, RowCfg<T, NN, true>::E>, \
            hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));                                   \
        if (rc == PFB_OK) rc = set_invp_attr<T, NN>();                                               \
        if (rc == PFB_OK) rc = prep_fwdp<T, NN>(&ft->ptwc_row_fwd); break;
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
static int rows_per_wg() { return row_groups<T, L, RowCfg<T, L, true>::E, RowCfg<T, L, true>::GMAX>(); }

bool pow2_supported(const pfb_conv_plan* p) {
    if (!is_pow2(p->nx) || !is_pow2(p->ny)) return false;
    if (p->P != 2 * p->nx || p->Q != 2 * p->ny) return false;
    if (p->nx < 64 || p->nx > 8192) return false;
    // rows: one length-ny/2 complex transform per image row in LDS: 8192 complex64 (ny = 16384) fit, complex128 do not
    if (p->ny < 128 || p->ny > (p->dtype == PFB_F32 ? 16384 : 8192)) return false;
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

int pow2_nblocks(const pfb_conv_plan* p) {
    return fast_nblocks(p->ny / 2, p->dtype == PFB_F32 ? FastCfg<float>::NVB : FastCfg<double>::NVB);
}
int pow2_nvb(const pfb_conv_plan* p) {
    return p->dtype == PFB_F32 ? FastCfg<float>::NVB : FastCfg<double>::NVB;
}

int pow2_prepare(pfb_conv_plan* p) {
    FastTables* ft = (FastTables*)calloc(1, sizeof(FastTables));
    PFB_REQUIRE(ft != nullptr, PFB_ERR_ALLOC, "pow2_prepare: host alloc failed");
    p->fast_tables = ft;
    // -1 = auto: the persistent prefetching kernel wins up to nx = 2048 (0.27 vs 0.33 ms at
    // 2048^2 x 8), the plain 2-workgroup/CU kernel at 4096 (1.17 vs 1.22 ms); 0 / 1 force one
    ft->col_persistent = -1;
    if (const char* e = getenv("PFB_COL_PERSIST")) ft->col_persistent = atoi(e) ? 1 : 0;
    ft->fwd_persistent = 1;
    if (const char* e = getenv("PFB_FWD_PERSIST")) ft->fwd_persistent = atoi(e) ? 1 : 0;
    ft->inv_persistent = 1;
    if (const char* e = getenv("PFB_INV_PERSIST")) ft->inv_persistent = atoi(e) ? 1 : 0;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        ft->num_cu = prop.multiProcessorCount;
    if (ft->num_cu <= 0) ft->num_cu = 256;
    ft->col_x = p->nx >= 8192 ? 1 : 0;
    if (const char* e = getenv("PFB_COL_X")) ft->col_x = (atoi(e) && p->nx >= 2048) ? 1 : 0;
    if (const char* e = getenv("PFB_CU_LIMIT")) { int v = atoi(e); if (v >= 16 && v < ft->num_cu) ft->num_cu = v; }   // experiments: leave CUs to a concurrent kernel
    return p->dtype == PFB_F32 ? prep_tables<float>(p, ft) : prep_tables<double>(p, ft);
}

void pow2_release(pfb_conv_plan* p) {
    FastTables* ft = (FastTables*)p->fast_tables;
    if (!ft) return;
    if (ft->ptw_col) (void)hipFree(ft->ptw_col);
    if (ft->ptwc_col) (void)hipFree(ft->ptwc_col);
    if (ft->ptwc_col_x) (void)hipFree(ft->ptwc_col_x);
    if (ft->ptw_row) (void)hipFree(ft->ptw_row);
    if (ft->ptw_row_inv_p && ft->ptw_row_inv_p != ft->ptw_row_inv) (void)hipFree(ft->ptw_row_inv_p);
    if (ft->ptw_row_inv) (void)hipFree(ft->ptw_row_inv);
    if (ft->twM) (void)hipFree(ft->twM);
    if (ft->ptwc_row_fwd) (void)hipFree(ft->ptwc_row_fwd);
    free(ft);
    p->fast_tables = nullptr;
}

template <typename T>
static int set_psfhat_t(pfb_conv_plan* p, const void* psfhat, hipStream_t st) {
    const int nv = p->M + 1;
    // the dummy partner of the last even bin (and nothing else) is never written: zero all
    PFB_HIP_CHECK(hipMemsetAsync(p->psf_l, 0, sizeof(cplx<T>) * p->psf_elems_per_band * p->nband, st));
    dim3 grid((nv + 31) / 32, (p->P + 31) / 32, p->nband);
    hipLaunchKernelGGL((k_relayout_psf_pow2<T>), grid, dim3(256), 0, st, (const cplx<T>*)psfhat,
                       (cplx<T>*)p->psf_l, p->P, nv, p->ny / 2, FastCfg<T>::NVB, p->psf_elems_per_band,
                       ((const FastTables*)p->fast_tables)->col_x);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

int pow2_set_psfhat(pfb_conv_plan* p, const void* psfhat, hipStream_t st) {
    return p->dtype == PFB_F32 ? set_psfhat_t<float>(p, psfhat, st) : set_psfhat_t<double>(p, psfhat, st);
}

template <typename T, int L>
static void launch_row_fwd(pfb_conv_plan* p, const FastTables* ft, void* Tbuf, int band0, int nb, const void* x,
                           const void* beam, size_t xpitch, size_t xband, hipStream_t st);

// psfhat = r2c(ifftshift(psf)) with the fast path's own kernels (see k_col_fwd_pow2), band by band:
// four pruned row passes over the PSF quadrants (ifftshift = which quadrant goes where), one forward column pass
// straight into psf_l; psfhat_out (optional) receives the reference's (P, M+1) layout.  Synchronous (plan time).
template <typename T>
static int set_psf_t(pfb_conv_plan* p, const void* psf, void* psfhat_out, hipStream_t st) {
    const FastTables* ft = (const FastTables*)p->fast_tables;
    const int nx = p->nx, ny = p->ny, P = p->P, Q = p->Q, L = ny / 2;
    void* Tq = nullptr;
    if (hipMalloc(&Tq, sizeof(cplx<T>) * 4 * p->T_elems_per_band) != hipSuccess) {
        set_error("pow2_set_psf: device allocation failed (%zu B)", sizeof(cplx<T>) * 4 * p->T_elems_per_band);
        return PFB_ERR_ALLOC;
    }
    int rc = PFB_OK;
    for (int band = 0; band < p->nband && rc == PFB_OK; ++band) {
        const T* pb = (const T*)psf + (size_t)band * P * Q;
        for (int q = 0; q < 4; ++q) {
            // quadrant (a, b) of the SHIFTED PSF = quadrant (1 - a, 1 - b) of the centred one
            const int a = q >> 1, b = q & 1;
            const T* src = pb + (size_t)((1 - a) * nx) * Q + (size_t)(1 - b) * ny;
            switch (L) {
#define X(NN) case NN: launch_row_fwd<T, NN>(p, ft, Tq, q, 1, src, nullptr, (size_t)Q, (size_t)0, st); break;
                PFB_POW2_SIZES(X)
#undef X
                default: set_error("pow2_set_psf: unsupported ny"); rc = PFB_ERR_UNSUPPORTED; break;
            }
        }
        if (rc == PFB_OK) rc = pow2_col_fwd_launch(p, ft, Tq, band, st);
    }
    if (rc == PFB_OK && psfhat_out) {
        const int nv = p->M + 1;
        dim3 grid((nv + 31) / 32, (P + 31) / 32, p->nband);
        hipLaunchKernelGGL((k_unrelayout_psf_pow2<T>), grid, dim3(256), 0, st, (const cplx<T>*)p->psf_l,
                           (cplx<T>*)psfhat_out, P, nv, L, FastCfg<T>::NVB, p->psf_elems_per_band, ft->col_x);
    }
    if (rc == PFB_OK && (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
        set_error("pow2_set_psf: kernel launch failed");
        rc = PFB_ERR_HIP;
    }
    (void)hipFree(Tq);
    return rc;
}

int pow2_set_psf(pfb_conv_plan* p, const void* psf, void* psfhat_out, hipStream_t st) {
    return p->dtype == PFB_F32 ? set_psf_t<float>(p, psf, psfhat_out, st) : set_psf_t<double>(p, psf, psfhat_out, st);
}

#endif  // PFB_POW2_REST

#if PFB_POW2_COL
template <typename T, int H>
static void launch_col(pfb_conv_plan* p, const FastTables* ft, int band0, int nb, hipStream_t st) {
    constexpr int E = ecol<T, H>();
    using F = RegFft<T, H, E>;
    constexpr int GC = col_groups<H, E>();
    const int nblk = fast_nblocks(p->ny / 2, FastCfg<T>::NVB);
    if constexpr (H >= 2048) {
        if (ft->col_x) {
            // two-level kernel: HS = H / 2 point sub-transforms, one resident workgroup set (2 waves per SIMD)
            using CX = ColX<T, H, ECOLX>;
            using FX = typename CX::F;
            constexpr int GX = CX::GC;
            const int nitems = nblk * nb;
            static const bool rev = [] { const char* e = getenv("PFB_COL_REV"); return !e || atoi(e); }();
            static const bool psf_nt_on = [] { const char* e = getenv("PFB_PSF_NT"); return !e || atoi(e); }();
            const bool psf_nt = psf_nt_on && sizeof(cplx<T>) * p->psf_elems_per_band * (size_t)nb >= ((size_t)200 << 20);
            const int wg_lds = (int)(((size_t)160 * 1024) / CX::LDS);
            int wg_per_cu = (8 * 64) / (GX * FX::TPB) > 0 ? (8 * 64) / (GX * FX::TPB) : 1;
            if (wg_per_cu > wg_lds) wg_per_cu = wg_lds > 0 ? wg_lds : 1;
            const int need = (nitems + GX - 1) / GX;
            // (not balanced_grid: this kernel is bound by the issue of its transforms, every workgroup counts --
            // 2.003 -> 2.047 ms fp64, 1.110 -> 1.128 fp32 with 253 / 249 instead of 256)
            const int grid = need < ft->num_cu * wg_per_cu ? need : ft->num_cu * wg_per_cu;
            const size_t ldsx = CX::LDS;
#define PFB_COLX(NTV)                                                                                          \
            hipLaunchKernelGGL((k_col_pow2x<T, H, ECOLX, NTV>), dim3(grid), dim3(GX * FX::TPB), ldsx, st,      \
                               (cplx<T>*)p->T, (const cplx<T>*)p->psf_l, (const cplx<T>*)p->twP,               \
                               (const cplx<T>*)ft->ptwc_col_x, nblk, nitems, p->T_elems_per_band,              \
                               p->psf_elems_per_band, rev ? band0 + nb - 1 : band0, rev ? -1 : 1)
            if (psf_nt) PFB_COLX(true); else PFB_COLX(false);
#undef PFB_COLX
            return;
        }
    }
    const size_t lds = sizeof(cplx<T>) * (size_t)GC * FastCfg<T>::NVB * F::LDS_ELEMS;
    // persistent kernel: needs its LDS (exchange buffers + twiddle table) to fit; measured faster
    // for H >= 2048 (fp32, packed arithmetic: 1.07 vs 1.32 ms at 4096^2 x 8, 0.276 vs 0.347 at 2048^2 x 8;
    // fp64: 1.19 vs 1.41 and 0.266 vs 0.333 at x 4; a tie or a small loss at H <= 1024)
    const size_t lds_p = lds + sizeof(cplx<T>) * (size_t)((F::PTWC + 1) & ~1);
    const bool fits_p = lds_p <= (size_t)160 * 1024;
    if (fits_p && (ft->col_persistent == 1 || (ft->col_persistent < 0 && H >= 2048))) {
        // one resident workgroup set: 8 waves per CU at 256 VGPRs
        const int nitems = nblk * nb;
        // bands in DESCENDING order: the row pass before and after run ascending, so each pass starts on
        // the band whose T the previous one touched last (part of it still in the 256 MiB Infinity Cache):
        // col 0.995 -> 0.987 ms, row_inv 0.666 -> 0.660 ms at 8 x 4096^2 fp32.  PFB_COL_REV=0 turns it off.
        static const bool rev = [] { const char* e = getenv("PFB_COL_REV"); return !e || atoi(e); }();
        const int wg_per_cu = (8 * 64) / (GC * F::TPB) > 0 ? (8 * 64) / (GC * F::TPB) : 1;
        const int need = (nitems + GC - 1) / GC;
        const int grid = balanced_grid(need, ft->num_cu * wg_per_cu);
        // second exchange buffer set when ONE workgroup per CU is resident anyway and it fits: col 0.98 ->
        // 0.96 ms at 8 x 4096^2 fp32, 1.16 -> 1.10 ms at 4 x 4096^2 fp64; with two workgroups per CU (H = 2048)
        // the doubled LDS costs residency (fp64 0.245 -> 0.304 ms).  PFB_COL_DB=0 turns it off.
        static const bool db_on = [] { const char* e = getenv("PFB_COL_DB"); return !e || atoi(e); }();
        // loads spread over the FFT passes (see the kernel); PFB_SPREAD=0 keeps the two-burst schedule (A/B)
        static const bool spread = [] { const char* e = getenv("PFB_SPREAD"); return !e || atoi(e); }();
        const size_t tab = sizeof(cplx<T>) * (size_t)((F::PTWC + 1) & ~1);
        const bool db = db_on && wg_per_cu == 1 && 2 * lds + tab <= (size_t)160 * 1024;
        // psf read with non-temporal loads (see loadb_nt); PFB_PSF_NT=0: A/B
        // ... only when this launch's share of the PSF spectrum is itself of the Infinity Cache's size or larger: a
        // smaller one is RE-READ from that cache by the next apply and must stay allocatable
        static const bool psf_nt_on = [] { const char* e = getenv("PFB_PSF_NT"); return !e || atoi(e); }();
        const bool psf_nt = psf_nt_on && sizeof(cplx<T>) * p->psf_elems_per_band * (size_t)nb >= ((size_t)200 << 20);
#define PFB_COLP(DBV, SPV)                                                                                     \
        if (SPV && psf_nt)                                                                                     \
        hipLaunchKernelGGL((k_col_pow2p<T, H, E, DBV, SPV, SPV>), dim3(grid), dim3(GC * F::TPB), (DBV ? 2 : 1) * lds + tab, st, \
                           (cplx<T>*)p->T, (const cplx<T>*)p->psf_l, (const cplx<T>*)p->twP,                   \
                           (const cplx<T>*)ft->ptwc_col, nblk, nitems, p->T_elems_per_band,                    \
                           p->psf_elems_per_band, rev ? band0 + nb - 1 : band0, rev ? -1 : 1);                 \
        else                                                                                                   \
        hipLaunchKernelGGL((k_col_pow2p<T, H, E, DBV, SPV, false>), dim3(grid), dim3(GC * F::TPB), (DBV ? 2 : 1) * lds + tab, st, \
                           (cplx<T>*)p->T, (const cplx<T>*)p->psf_l, (const cplx<T>*)p->twP,                   \
                           (const cplx<T>*)ft->ptwc_col, nblk, nitems, p->T_elems_per_band,                    \
                           p->psf_elems_per_band, rev ? band0 + nb - 1 : band0, rev ? -1 : 1)
        if (db) { if (spread) PFB_COLP(true, true); else PFB_COLP(true, false); }
        else    { if (spread) PFB_COLP(false, true); else PFB_COLP(false, false); }
#undef PFB_COLP
        return;
    }
    hipLaunchKernelGGL((k_col_pow2<T, H, E>), dim3((nblk + GC - 1) / GC, nb), dim3(GC * F::TPB), lds, st,
                       (cplx<T>*)p->T, (const cplx<T>*)p->psf_l, (const cplx<T>*)p->twP,
                       (const cplx<T>*)ft->ptw_col, nblk, p->T_elems_per_band, p->psf_elems_per_band, band0);
}

template <typename T>
static int col_set_attr_t(int H) {
    constexpr int lds_max = 160 * 1024;
    switch (H) {
#define X(NN) case NN:                                                                                        \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2<T, NN, ecol<T, NN>()>),                    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));              \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2p<T, NN, ecol<T, NN>(), false, false>),     \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));              \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2p<T, NN, ecol<T, NN>(), true, false>),      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));              \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2p<T, NN, ecol<T, NN>(), false, true>),      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));              \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2p<T, NN, ecol<T, NN>(), true, true>),       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));              \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2p<T, NN, ecol<T, NN>(), false, true, true>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));              \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2p<T, NN, ecol<T, NN>(), true, true, true>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));              \
        PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_fwd_pow2<T, NN, ecol<T, NN>()>),                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));              \
        if constexpr (NN >= 2048) {                                                                           \
            PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2x<T, NN, ECOLX, false>),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));          \
            PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_col_pow2x<T, NN, ECOLX, true>),                 \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));          \
        }                                                                                                     \
        break;
        PFB_POW2_SIZES(X)
#undef X
        default: set_error("pow2: unsupported nx"); return PFB_ERR_UNSUPPORTED;
    }
    return PFB_OK;
}
int pow2_col_set_attr(int dtype, int H) {
    return dtype == PFB_F32 ? col_set_attr_t<float>(H) : col_set_attr_t<double>(H);
}
template <typename T>
static int col_launch_t(pfb_conv_plan* p, const FastTables* ft, int band0, int nb, hipStream_t st) {
    switch (p->nx) {
#define X(NN) case NN: launch_col<T, NN>(p, ft, band0, nb, st); break;
        PFB_POW2_SIZES(X)
#undef X
        default: set_error("pow2_apply: unsupported nx"); return PFB_ERR_UNSUPPORTED;
    }
    return PFB_OK;
}
int pow2_col_launch(pfb_conv_plan* p, const FastTables* ft, int band0, int nb, hipStream_t st) {
    return p->dtype == PFB_F32 ? col_launch_t<float>(p, ft, band0, nb, st) : col_launch_t<double>(p, ft, band0, nb, st);
}
// PSFHAT producer: the four quadrant spectra in Tq (slots of T_elems_per_band) -> psf_l of `band`
template <typename T, int H>
static void launch_col_fwd(pfb_conv_plan* p, const FastTables* ft, const void* Tq, int band, hipStream_t st) {
    constexpr int E = ecol<T, H>();
    using F = RegFft<T, H, E>;
    constexpr int GC = col_groups<H, E>();
    constexpr int NVB = FastCfg<T>::NVB;
    const int L = p->ny / 2;
    const int nblk = fast_nblocks(L, NVB), nbe = (L + NVB) / NVB;
    const size_t lds = sizeof(cplx<T>) * (size_t)GC * NVB * F::LDS_ELEMS;
    hipLaunchKernelGGL((k_col_fwd_pow2<T, H, E>), dim3((nblk + GC - 1) / GC), dim3(GC * F::TPB), lds, st,
                       (const cplx<T>*)Tq, (cplx<T>*)p->psf_l + (size_t)band * p->psf_elems_per_band,
                       (const cplx<T>*)p->twP, (const cplx<T>*)ft->ptw_col, nblk, nbe, p->T_elems_per_band, ft->col_x);
}
template <typename T>
static int col_fwd_launch_t(pfb_conv_plan* p, const FastTables* ft, const void* Tq, int band, hipStream_t st) {
    switch (p->nx) {
#define X(NN) case NN: launch_col_fwd<T, NN>(p, ft, Tq, band, st); break;
        PFB_POW2_SIZES(X)
#undef X
        default: set_error("pow2_set_psf: unsupported nx"); return PFB_ERR_UNSUPPORTED;
    }
    return PFB_OK;
}
int pow2_col_fwd_launch(pfb_conv_plan* p, const FastTables* ft, const void* Tq, int band, hipStream_t st) {
    return p->dtype == PFB_F32 ? col_fwd_launch_t<float>(p, ft, Tq, band, st) : col_fwd_launch_t<double>(p, ft, Tq, band, st);
}
#endif  // PFB_POW2_COL

#if PFB_POW2_REST

// Tbuf: the half-spectrum buffer to fill (the plan's T, or the PSFHAT producer's quadrant scratch: `band0`
// is then the slot); xpitch / xband: elements between rows / bands of x (and beam)
template <typename T, int L>
static void launch_row_fwd(pfb_conv_plan* p, const FastTables* ft, void* Tbuf, int band0, int nb, const void* x,
                           const void* beam, size_t xpitch, size_t xband, hipStream_t st) {
    constexpr int E = RowCfg<T, L, false>::E;
    using F = RegFft<T, L, E, RowCfg<T, L, false>::WAVE>;
    constexpr int G = row_groups<T, L, E, RowCfg<T, L, false>::GMAX>();
    FastDims d{p->nx, p->ny, p->M, p->T_elems_per_band, p->psf_elems_per_band, xpitch, xband};
    if constexpr (FwdP<T, L>::OK) {
        if (ft->fwd_persistent) {
            using FP = FwdP<T, L>;
            const int tiles_per_band = p->nx / FP::G, ntiles = tiles_per_band * nb;
            const int grid = ntiles < ft->num_cu * FP::WG_PER_CU ? ntiles : ft->num_cu * FP::WG_PER_CU;
            const long double a = 6.283185307179586476925286766559005768L / (2.0L * (long double)p->ny);
            const cplx<T> wq1((T)cosl(a), (T)(-sinl(a)));
            // spreading the next tile's row requests over the even-bin sweep measured SLOWER here (0.389 -> 0.411 ms at
            // 8 x 4096^2 fp32: the sweep's stores and the loads then queue behind each other): off unless PFB_SPREAD_FWD=1
            // parities in sequence, next rows requested inside the second transform (k_row_fwd_pow2q); PFB_FWD_SEQ=0: A/B
            static const bool seq = [] { const char* e = getenv("PFB_FWD_SEQ"); return !e || atoi(e); }();
            if (seq) {
#define PFB_FWDQ(BM)                                                                                              \
                hipLaunchKernelGGL((k_row_fwd_pow2q<T, L, BM>), dim3(grid), dim3(FP::NT), FP::LDS, st, (const T*)x, \
                                   (const T*)beam, (cplx<T>*)Tbuf, (const cplx<T>*)ft->twM,                      \
                                   (const cplx<T>*)ft->ptwc_row_fwd, d, band0, tiles_per_band, ntiles, wq1)
                if (beam) PFB_FWDQ(true); else PFB_FWDQ(false);
#undef PFB_FWDQ
                return;
            }
            static const bool spread = [] { const char* e = getenv("PFB_SPREAD_FWD"); return e && atoi(e); }();
#define PFB_FWDP(BM, SP)                                                                                          \
            hipLaunchKernelGGL((k_row_fwd_pow2p<T, L, BM, SP>), dim3(grid), dim3(FP::NT), FP::LDS, st, (const T*)x, \
                               (const T*)beam, (cplx<T>*)Tbuf, (const cplx<T>*)ft->twM,                          \
                               (const cplx<T>*)ft->ptwc_row_fwd, d, band0, tiles_per_band, ntiles, wq1)
            if (beam) { if (spread) PFB_FWDP(true, true); else PFB_FWDP(true, false); }
            else      { if (spread) PFB_FWDP(false, true); else PFB_FWDP(false, false); }
#undef PFB_FWDP
            return;
        }
    }
    const size_t lds = sizeof(cplx<T>) * (size_t)G * (F::LDS_ELEMS + 4);
    hipLaunchKernelGGL((k_row_fwd_pow2<T, L, E>), dim3(p->nx / G, nb), dim3(G * F::TPB), lds, st,
                       (const T*)x, (const T*)beam, (cplx<T>*)Tbuf, (const cplx<T>*)p->twQ,
                       (const cplx<T>*)ft->twM, (const cplx<T>*)ft->ptw_row, d, band0);
}

template <typename T, int L>
static void launch_row_inv(pfb_conv_plan* p, const FastTables* ft, int band0, int nb, const void* x,
                           const void* beam, double scale, double sigmainv, void* out,
                           const void* dot_with, const void* dot_with2, hipStream_t st) {
    constexpr int E = RowCfg<T, L, true>::E;
    using F = RegFft<T, L, E, RowCfg<T, L, true>::WAVE>;
    constexpr int G = row_groups<T, L, E, RowCfg<T, L, true>::GMAX>();
    FastDims d{p->nx, p->ny, p->M, p->T_elems_per_band, p->psf_elems_per_band, (size_t)p->ny, (size_t)p->nx * p->ny};
    constexpr int EP = InvPE<T, L>::E;
    if constexpr (InvP<T, L, EP>::OK) {
        // pipelined persistent kernel: no beam, inner products only against x itself (+ dot_with2)
        const bool plain_dots = !dot_with || (dot_with == x);
        if (ft->inv_persistent && plain_dots && !(dot_with2 && !dot_with)) {
            using IP = InvP<T, L, EP>;
            const int tiles_per_band = p->nx / IP::G, ntiles = tiles_per_band * nb;
            const int grid = ntiles < IP::WG_PER_CU * ft->num_cu ? ntiles : IP::WG_PER_CU * ft->num_cu;
            const long double a = 6.283185307179586476925286766559005768L / (2.0L * (long double)p->ny);
            const cplx<T> wq1((T)cosl(a), (T)(-sinl(a)));
            static const bool spread = [] { const char* e = getenv("PFB_SPREAD"); return !e || atoi(e); }();
            static const int defer = [] { const char* e = getenv("PFB_INV_DEFER"); return e ? atoi(e) : 1; }();   // A/B
#define PFB_INVP3(MODE, BM, SP)                                                                         \
            hipLaunchKernelGGL((k_row_inv_pow2p<T, L, EP, MODE, BM, SP>), dim3(grid), dim3(IP::NT), IP::LDS, st, \
                               (const cplx<T>*)p->T, (const cplx<T>*)ft->twM,                           \
                               (const cplx<T>*)ft->ptw_row_inv_p, (const T*)x, (const T*)beam,          \
                               (const T*)dot_with2, (T*)out, p->partials, d, band0, tiles_per_band,     \
                               ntiles, (T)scale, (T)sigmainv, wq1, defer, tail)
#define PFB_INVP2(MODE, BM) do { if (spread) PFB_INVP3(MODE, BM, true); else PFB_INVP3(MODE, BM, false); } while (0)
#define PFB_INVP(MODE) do { if (beam) PFB_INVP2(MODE, true); else PFB_INVP2(MODE, false); } while (0)
            p->last_npartials = grid;
            // the PCG driver's bookkeeping rides in the tail of the MODE 2 kernel (pcg_state.hpp)
            PcgTail tail = p->tail;
            if (!(dot_with && dot_with2) || IP::NT < 256 || grid > 1024) tail.S = nullptr;
            if (tail.S) p->tail_done = 1;
            if (!dot_with) PFB_INVP(0);
            else if (!dot_with2) PFB_INVP(1);
            else PFB_INVP(2);
#undef PFB_INVP
#undef PFB_INVP2
#undef PFB_INVP3
            return;
        }
    }
    const size_t lds = 384 + sizeof(cplx<T>) * ((size_t)((F::PTWC + 1) & ~1) + (size_t)G * (F::LDS_ELEMS + 4) * (InvDb<T, L>::ON ? 2 : 1));
    PcgTail tail = p->tail;
    if (!(dot_with && dot_with2) || G * F::TPB < 256 || (size_t)(p->nx / G) * nb != (size_t)p->last_npartials ||
        (size_t)(p->nx / G) * nb > 1024) tail.S = nullptr;
    if (tail.S) p->tail_done = 1;
    hipLaunchKernelGGL((k_row_inv_pow2<T, L, E>), dim3(p->nx / G, nb), dim3(G * F::TPB), lds, st,
                       (const cplx<T>*)p->T, (const cplx<T>*)p->twQ, (const cplx<T>*)ft->twM,
                       (const cplx<T>*)ft->ptw_row_inv, (const T*)x, (const T*)beam, (const T*)dot_with,
                       (const T*)dot_with2, (T*)out, p->partials, d, band0, (T)scale, (T)sigmainv, tail);
}

template <typename T>
static int apply_t(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam, double scale,
                   double sigmainv, void* out, const void* dot_with, const void* dot_with2, hipStream_t st) {
    const FastTables* ft = (const FastTables*)p->fast_tables;
    const int H = p->nx, L = p->ny / 2;
    prof_mark(p, st, 0);
    switch (L) {
#define X(NN) case NN: launch_row_fwd<T, NN>(p, ft, p->T, band0, nb, x, beam, (size_t)p->ny, (size_t)p->nx * p->ny, st); break;
        PFB_POW2_SIZES(X)
#undef X
        default: set_error("pow2_apply: unsupported ny"); return PFB_ERR_UNSUPPORTED;
    }
    prof_mark(p, st, 1);
    if (int rc = pow2_col_launch(p, ft, band0, nb, st); rc != PFB_OK) return rc;
    prof_mark(p, st, 2);
    switch (L) {
#define X(NN) case NN: launch_row_inv<T, NN>(p, ft, band0, nb, x, beam, scale, sigmainv, out, dot_with, dot_with2, st); break;
        PFB_POW2_SIZES(X)
#undef X
        default: break;
    }
    prof_mark(p, st, 3);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

int pow2_apply(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam, double scale,
               double sigmainv, void* out, const void* dot_with, const void* dot_with2, hipStream_t st) {
    return p->dtype == PFB_F32 ? apply_t<float>(p, band0, nb, x, beam, scale, sigmainv, out, dot_with, dot_with2, st)
                               : apply_t<double>(p, band0, nb, x, beam, scale, sigmainv, out, dot_with, dot_with2, st);
}
#endif  // PFB_POW2_REST

}  // namespace pfb

#if PFB_STAMP && PFB_POW2_REST
// diagnostic build only: buf = 3 * 1024 * PFB_STAMP_ITS * 16 device uint64 (NULL switches the stamps off)
extern "C" int pfb_debug_set_stamps(void* buf) { return pfb::pow2_set_stamp((unsigned long long*)buf); }
// diagnostic build only: src = an array shaped like the image cube that k_row_fwd_pow2q streams in addition (NULL: off)
extern "C" int pfb_debug_set_extra_stream(const void* src, double* sink) { return pfb::pow2_set_xtra(src, sink); }
#endif
