// wavelet.hip -- psi / psi^H (multi-level separable Daubechies DWT, zero-extension mode),
// the l21 dual update / prox and the primal-dual image update.
//
// Replaces the numba loops of pfb/wavelets/wavelets.py:127-315 (dwt2d_level / idwt2d_level
// + copyT), pfb/operators/psi.py:187-256 (psi_band.dot / hdot), pfb/prox/prox_21m.py:31-103
// and the numexpr passes of pfb/opt/primal_dual.py:137-146.
//
// One fused kernel per decomposition level: a workgroup stages an input tile (with its
// F-tap halo) in LDS, runs BOTH 1-D passes on chip and writes each output once, already
// in the reference's transposed (y-major) packed coefficient layout -- the reference's
// separate row pass, copyT transpose and column pass (3 sweeps over HBM) become one.
// All memory-bound: coalesced row reads, LDS-staged taps, no MFMA (no contraction).
//
//   analysis 1-D :  out[o]   = sum_{j<F} filt[j] in[2o+1-j]          (zero outside)
//   synthesis 1-D:  out[2m+p] = sum_{j<F/2} lo[2j+p] a[m+h-1-j] + sum_j hi[2j+p] d[m+h-1-j]
//                   m = 0 .. N-h,  h = F/2                           (wavelets.py:99-123)
#include "common.hpp"
#include <vector>
#include <cstring>

namespace pfb {

constexpr int MAXF = 18;        // db9
constexpr int MAXLEV = 12;
// tile edges: analysis = coefficients per quadrant and edge, synthesis = image pixels per edge
template <typename T> struct Tile;
template <> struct Tile<float>  { static constexpr int TA = 32; static constexpr int TS = 64; };
template <> struct Tile<double> { static constexpr int TA = 16; static constexpr int TS = 32; };

template <typename T> struct Filt { T lo[MAXF]; T hi[MAXF]; int F; };

struct LevelInfo {
    int nxin, nyin;     // analysis input (= approx of the previous level) shape
    int Cx, Cy;         // coefficients per half at this level (sx, sy)
    int lowx, lowy;     // origin of this level's (2Cy, 2Cx) block in the packed plane
    int nxo, nyo;       // synthesis output shape (spx, spy)
};

struct BasisInfo {
    int K, F, Ntotx, Ntoty;
    LevelInfo lev[MAXLEV];
    double filt[4][MAXF];       // dec_lo, dec_hi, rec_lo, rec_hi
};

}  // namespace pfb

struct pfb_psi_plan {
    int nband, nx, ny, nbasis, nlevel, dtype;
    int Nxmax, Nymax;
    pfb::BasisInfo* bases;
    void* scratch[2];           // per band ping-pong approx / partial-image buffers
    size_t scratch_band;        // elements per band in each scratch buffer
    // fused finest synthesis level (k_idwt_finest_fused): per-basis parameter table on the device
    // and one level-1 partial image per basis and band (all alive when the fused kernel runs)
    void* fin_prm;
    void* fin_scratch;
    size_t fin_band, fin_basis;
    int fin_fmax;
    // level kernels batched over the wavelet bases: parameter tables [level][wavelet basis] and a
    // ping-pong scratch pair with one slice per wavelet basis
    void* ana_prm;
    void* syn_prm;
    void* bscr[2];
    size_t bscr_basis;
    int nwb;
    int gx_ana[pfb::MAXLEV], gy_ana[pfb::MAXLEV], gx_syn[pfb::MAXLEV], gy_syn[pfb::MAXLEV];
};

namespace pfb {

static inline int coeff_size(int n, int F) { return (n + F - 1) / 2; }
static inline int signal_size(int c, int F) { return 2 * c - F + 2; }

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs by linear id, so tiles that are neighbours
// along the contiguous axis (blockIdx.x) land in 8 different L2s: the rows of the packed coefficient plane are NOT
// line aligned (odd Cx, ld 2068), every 128-byte tile row straddles two lines, and each L2 then writes back (or
// fetches) its own partial copy of the shared line.  Within every run of 64 consecutive workgroups the 8 that share an
// XCD (ids j, j + 8, .. j + 56) are given 8 CONSECUTIVE tiles instead (speed only: any bijection is correct).
struct TileId { int x, y, z; };
__device__ __forceinline__ TileId xcd_tile(bool on) {
    TileId t{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    if (!on) return t;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if (L < (total & ~63u)) L = (L & ~63u) + ((L & 7u) << 3) + ((L >> 3) & 7u);
    t.x = (int)(L % gx);
    const unsigned q = L / gx;
    t.y = (int)(q % gy);
    t.z = (int)(q / gy);
    return t;
}

// ------------------------------------------------------------------ 'self' basis
// dst[c][r] = src[r][c]   (src R x C, ld ls; dst ld ld); ACC adds instead of storing
template <typename T, bool ACC>
__global__ void __launch_bounds__(256)
k_transpose(const T* __restrict__ src, size_t src_band, int ls, T* __restrict__ dst, size_t dst_band,
            int ld, int R, int C) {
    __shared__ T tile[32][33];
    const T* s = src + (size_t)blockIdx.z * src_band;
    T* d = dst + (size_t)blockIdx.z * dst_band;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < R && c0 + tx < C) tile[r][tx] = s[(size_t)(r0 + r) * ls + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < C && r0 + tx < R) {
            T* p = d + (size_t)(c0 + c) * ld + r0 + tx;
            if (ACC) *p += tile[tx][c]; else *p = tile[tx][c];
        }
}

// ------------------------------------------------------------------ analysis level
// in    : (nxin, nyin) row-major, ld = ldin
// coeffs: this level's block origin, ld = ldc; quadrants [0:Cy | Cy:2Cy] x [0:Cx | Cx:2Cx]
// approx: optional (Cx, Cy) row-major copy of the LL quadrant transposed (next level input)
// F (filter length) and TA (output tile edge per quadrant) are compile-time: the staging loop
// is fully unrolled so that ALL global loads of a thread are in flight before the first LDS
// store (a rolled load-wait-store loop pays the HBM latency once per trip: 104 us -> see
// DESIGN.md), and the tap loops are unrolled FMA chains.
// layout of the register-blocked tile (dwt_tile_fast below), shared with the host's LDS sizing
template <typename T> struct DwtFast {
    static constexpr int EV = 16 / (int)sizeof(T);          // elements per 16-byte access
    static constexpr int QB = 2 * EV;                       // y-pass outputs (per filter) of one work item
    // samples between the 16-byte aligned start of the staged window and the tile's first sample: the first sample is
    // global y = 2 oy0 + 2 - F with oy0 a multiple of the tile edge, i.e. 2 - F modulo the vector width
    static constexpr int off(int F, bool al = true) { return (al && EV == 4 && F % 4 == 0) ? 2 : 0; }
    // 16-byte reads per parity (the aligned layout's count also covers the unaligned one, whose window starts AT the tile)
    static constexpr int nrd(int F) { return (QB + F / 2 - 1 + off(F) / 2 + EV - 1) / EV; }
    // samples of one parity per tile row, padded so that (a) the last item's reads stay inside the row and (b) lanes
    // walking down the rows hit disjoint banks with 16-byte reads (row stride = 4 * odd words modulo 64)
    static constexpr int sa(int TA, int F) {
        int v = TA - QB + nrd(F) * EV;
        const int m = EV == 4 ? 8 : 4, r = EV == 4 ? 4 : 2;
        while (v % m != r) v += EV;
        return v;
    }
    static constexpr int ni(int TA, int F) { return 2 * TA + F - 2; }
    static constexpr int sb(int TA) { return 2 * TA + EV; }                                       // B row: lo | hi | pad
    static constexpr int bo(int TA, int F) {                                                      // odd-row half of B
        const int w = 64 / (EV == 4 ? 1 : 2), half = 32 / (EV == 4 ? 1 : 2);                      // 64 / 32 banks in elements
        return (ni(TA, F) / 2 * sb(TA) + w - 1) / w * w + half;
    }
    static constexpr size_t elems(int TA, int F) {
        return (size_t)2 * ni(TA, F) * sa(TA, F) + (size_t)bo(TA, F) + (size_t)(ni(TA, F) / 2) * sb(TA);
    }
};

template <typename T, int F, int TA>
__device__ __forceinline__ void dwt_tile(T* smem, const T* __restrict__ flo, const T* __restrict__ fhi,
                                         const T* __restrict__ src, int ldin, int nxin, int nyin,
                                         T* __restrict__ dst, int ldc, int Cx, int Cy, T* __restrict__ approx,
                                         int ox0, int oy0, bool allow_fast = true);

template <typename T, int F, int TA>
__global__ void __launch_bounds__(256)
k_dwt_level(const T* __restrict__ in, size_t in_band, int ldin, int nxin, int nyin,
            T* __restrict__ coeffs, size_t c_band, int ldc, int Cx, int Cy,
            T* __restrict__ approx, size_t a_band, Filt<T> f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    dwt_tile<T, F, TA>(reinterpret_cast<T*>(smem), f.lo, f.hi, in + (size_t)blockIdx.z * in_band, ldin, nxin, nyin,
                       coeffs + (size_t)blockIdx.z * c_band, ldc, Cx, Cy,
                       approx ? approx + (size_t)blockIdx.z * a_band : nullptr, blockIdx.x * TA, blockIdx.y * TA,
                       f.F > 0);                    // F < 0 (launch_dwt, PFB_DWT_FAST=0): plain tiles only
}

// Register-blocked tile for 16-byte aligned input rows (the finest level of an image whose width is a multiple of
// the vector width -- the level that carries 3/4 of the work).  The plain tile below spends one LDS read and a few
// address / guard instructions per tap and output; rocprofv3 and the ISA say the level kernel is ISSUE bound (its
// load, LDS and store phases add up).  Here
//   * staging writes each 16-byte global vector as two 8-byte LDS stores (even | odd samples), no per-sample guard;
//   * the y pass gives a work item one tile row and QB consecutive outputs of BOTH filters: 2 nrd 16-byte LDS reads
//     feed QB * F packed FMAs (the (lo, hi) pair of a tap is one v_pk_fma_f32 operand);
//   * the x pass gives a work item one output column and EV consecutive rows of B: F 16-byte reads per 2 EV outputs,
//     lanes along the column index so that the global stores stay coalesced and B (row stride 2TA + EV) is read
//     without bank conflicts.
// AL = false: rows of any alignment (the coarser levels read the (Cx, Cy) approximation of the level above, whose
// row length is odd): a wave takes every 4th tile row and a lane one (even, odd) sample pair, loaded unconditionally
// from clamped addresses and zeroed by a select -- the flat per-sample mapping of the plain tile costs ~20 instructions
// per sample.
// staging of a tile with the halo of an FS-tap filter (FS >= the F of every basis that will use it)
template <typename T, int FS, int TA, bool AL>
__device__ __forceinline__ void dwt_fast_stage(T* A, const T* __restrict__ src, int ldin, int nxin, int nyin,
                                               int ox0, int oy0) {
    using D = DwtFast<T>;
    constexpr int EV = D::EV;
    constexpr int NI = D::ni(TA, FS), OFF = D::off(FS, AL);
    static_assert(AL || NI / 2 <= 64, "a tile row's sample pairs fit a wavefront");
    constexpr int SA = D::sa(TA, FS), AO = NI * SA;
    struct alignas(16) Vec { T e[EV]; };
    struct alignas(8) Half { T e[EV / 2]; };
    const int tid = threadIdx.x;
    const int gx0 = 2 * ox0 + 2 - FS, ga = 2 * oy0 + 2 - FS - OFF;     // first input row; 16-byte aligned window start
    if constexpr (!AL) {
        // stage, any alignment: lane = sample pair (2 lane, 2 lane + 1) of the row, wave wv rows wv, wv + 4, ..
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int gye = ga + 2 * lane, gyo = gye + 1;
        const bool oke = lane < NI / 2 && gye >= 0 && gye < nyin, oko = lane < NI / 2 && gyo >= 0 && gyo < nyin;
        const int ce = gye < 0 ? 0 : (gye < nyin ? gye : nyin - 1), co = gyo < 0 ? 0 : (gyo < nyin ? gyo : nyin - 1);
        constexpr int NR = (NI + 3) / 4;
        T se[NR], so[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int lx = wv + 4 * k, gx = gx0 + lx;                    // wave-uniform
            const bool okx = lx < NI && gx >= 0 && gx < nxin;
            const T* row = src + (size_t)(gx < 0 ? 0 : (gx < nxin ? gx : nxin - 1)) * ldin;
            const T ve = row[ce], vo = row[co];
            se[k] = (okx && oke) ? ve : T(0);
            so[k] = (okx && oko) ? vo : T(0);
        }
        if (lane < NI / 2) {
#pragma unroll
            for (int k = 0; k < NR; ++k)
                if (wv + 4 * k < NI) { A[(wv + 4 * k) * SA + lane] = se[k]; A[AO + (wv + 4 * k) * SA + lane] = so[k]; }
        }
    } else {
        // vector w of tile row lx holds samples ga + EV w .. + EV - 1 (entirely inside or outside [0, nyin))
        constexpr int NWV = (NI + OFF + EV - 1) / EV;
        static_assert(NWV * (EV / 2) <= SA, "staged row fits");
        constexpr int NLV = (NI * NWV + 255) / 256;
        Vec stage[NLV];
#pragma unroll
        for (int k = 0; k < NLV; ++k) {
            const int e = tid + 256 * k;
            const int lx = e / NWV, w = e - lx * NWV;
            const int gx = gx0 + lx, gy = ga + EV * w;
            Vec v;
#pragma unroll
            for (int c = 0; c < EV; ++c) v.e[c] = 0;
            if (e < NI * NWV && gx >= 0 && gx < nxin && gy >= 0 && gy < nyin)
                v = *reinterpret_cast<const Vec*>(src + (size_t)gx * ldin + gy);
            stage[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NLV; ++k) {
            const int e = tid + 256 * k;
            const int lx = e / NWV, w = e - lx * NWV;
            if (e < NI * NWV) {
                Half ev, od;
#pragma unroll
                for (int c = 0; c < EV / 2; ++c) { ev.e[c] = stage[k].e[2 * c]; od.e[c] = stage[k].e[2 * c + 1]; }
                T* a = A + lx * SA + (EV / 2) * w;
                *reinterpret_cast<Half*>(a) = ev;
                *reinterpret_cast<Half*>(a + AO) = od;
            }
        }
    }
}

// the two passes of an F-tap basis on a tile staged with the halo of FS taps (OFFS = its window offset).
// LLB = false: the LL quadrant is collected in LDS on top of A (dead after the y pass: single-basis tiles);
// LLB = true : A stays valid for the next basis -- LL waits in registers and goes to LDS on top of B, one barrier later.
template <typename T, int F, int FS, int TA, int OFFS, bool LLB>
__device__ __forceinline__ void dwt_fast_passes(T* A, T* B, const T* __restrict__ flo, const T* __restrict__ fhi,
                                                T* __restrict__ dst, int ldc, int Cx, int Cy, T* __restrict__ approx,
                                                int ox0, int oy0) {
    using D = DwtFast<T>;
    typedef T T2 __attribute__((ext_vector_type(2)));
    static_assert(F <= FS && (FS - F) % 2 == 0, "staged halo covers the basis");
    constexpr int EV = D::EV, QB = D::QB, H = F / 2;
    constexpr int NI = D::ni(TA, F), NIS = D::ni(TA, FS), ROFF = FS - F, OFFH = (FS - F + OFFS) / 2, NRD = D::nrd(FS);
    constexpr int SA = D::sa(TA, FS), AO = NIS * SA, SB = D::sb(TA), BO = D::bo(TA, FS);
    static_assert(TA - QB + QB + H - 1 + OFFH <= SA, "reads stay inside a staged row");
    struct alignas(16) Vec { T e[EV]; };
    T* LL = LLB ? B : A;                            // [TA][TA + 1]
    const int tid = threadIdx.x;
    T2 f2[F];
#pragma unroll
    for (int j = 0; j < F; ++j) { f2[j].x = flo[j]; f2[j].y = fhi[j]; }
    if constexpr (LLB) __syncthreads();             // the previous basis has copied its LL out of B
    // y pass: B[lb][q] = sum_m f[F-1-2m] Ae[lx][q+m] + f[F-2-2m] Ao[lx][q+m]  (q < TA: lo, TA + q: hi), lx = lb + ROFF
    {
        constexpr int NQB = TA / QB;
        for (int it = tid; it < NI * NQB; it += 256) {
            const int qb = it / NI, lb = it - qb * NI;              // lanes walk down the rows
            const T* ae = A + (lb + ROFF) * SA + qb * QB;
            T se[NRD * EV], so[NRD * EV];
#pragma unroll
            for (int k = 0; k < NRD; ++k) {
                *reinterpret_cast<Vec*>(&se[EV * k]) = *reinterpret_cast<const Vec*>(ae + EV * k);
                *reinterpret_cast<Vec*>(&so[EV * k]) = *reinterpret_cast<const Vec*>(ae + AO + EV * k);
            }
            T2 acc[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                acc[q] = T2{0, 0};
#pragma unroll
                for (int m = 0; m < H; ++m) {
                    acc[q] += f2[F - 1 - 2 * m] * se[q + m + OFFH];
                    acc[q] += f2[F - 2 - 2 * m] * so[q + m + OFFH];
                }
            }
            T* brow = B + (lb & 1) * BO + (lb >> 1) * SB + qb * QB;
#pragma unroll
            for (int k = 0; k < QB / EV; ++k) {
                Vec l, h;
#pragma unroll
                for (int c = 0; c < EV; ++c) { l.e[c] = acc[EV * k + c].x; h.e[c] = acc[EV * k + c].y; }
                *reinterpret_cast<Vec*>(brow + EV * k) = l;
                *reinterpret_cast<Vec*>(brow + TA + EV * k) = h;
            }
        }
    }
    __syncthreads();
    // x pass + store: out[r][cc] = sum_m f[F-1-2m] Be[cc+m][r] + f[F-2-2m] Bo[cc+m][r]
    constexpr int NXI = TA * (2 * TA / EV), NXT = (NXI + 255) / 256;
    T llv[LLB ? NXT : 1][EV];
    {
        const bool full = ox0 + TA <= Cx && oy0 + TA <= Cy;
#pragma unroll
        for (int kt = 0; kt < NXT; ++kt) {
            const int it = tid + 256 * kt;
            if (it >= NXI) break;
            const int rb = it / TA, cc = it - rb * TA;              // lanes along the output column index
            const int r0 = rb * EV;
            const T* b = B + cc * SB + r0;
            T2 acc[EV];
#pragma unroll
            for (int r = 0; r < EV; ++r) acc[r] = T2{0, 0};
#pragma unroll
            for (int m = 0; m < H; ++m) {
                const Vec ve = *reinterpret_cast<const Vec*>(b + m * SB);
                const Vec vo = *reinterpret_cast<const Vec*>(b + BO + m * SB);
#pragma unroll
                for (int r = 0; r < EV; ++r) {
                    acc[r] += f2[F - 1 - 2 * m] * ve.e[r];
                    acc[r] += f2[F - 2 - 2 * m] * vo.e[r];
                }
            }
            const bool hiy = r0 >= TA;
            const int rr0 = hiy ? r0 - TA : r0;
            const int gx = ox0 + cc;
            T* row = dst + (size_t)((hiy ? Cy : 0) + oy0 + rr0) * ldc + gx;
            if (full) {
#pragma unroll
                for (int r = 0; r < EV; ++r) { row[(size_t)r * ldc] = acc[r].x; row[(size_t)r * ldc + Cx] = acc[r].y; }
            } else if (gx < Cx) {
#pragma unroll
                for (int r = 0; r < EV; ++r)
                    if (oy0 + rr0 + r < Cy) { row[(size_t)r * ldc] = acc[r].x; row[(size_t)r * ldc + Cx] = acc[r].y; }
            }
            if constexpr (LLB) {
#pragma unroll
                for (int r = 0; r < EV; ++r) llv[kt][r] = acc[r].x;
            } else if (!hiy) {
#pragma unroll
                for (int r = 0; r < EV; ++r) LL[cc * (TA + 1) + rr0 + r] = acc[r].x;
            }
        }
    }
    if (approx) {
        __syncthreads();
        if constexpr (LLB) {                        // B is dead now: park the LL quadrant there
#pragma unroll
            for (int kt = 0; kt < NXT; ++kt) {
                const int it = tid + 256 * kt;
                const int rb = it / TA, cc = it - rb * TA, r0 = rb * EV;
                if (it < NXI && r0 < TA) {
#pragma unroll
                    for (int r = 0; r < EV; ++r) LL[cc * (TA + 1) + r0 + r] = llv[kt][r];
                }
            }
            __syncthreads();
        }
        for (int e = tid; e < TA * TA; e += 256) {
            const int cc = e / TA, rr = e - cc * TA;           // rr (y) fastest: coalesced
            if (ox0 + cc < Cx && oy0 + rr < Cy) approx[(size_t)(ox0 + cc) * Cy + oy0 + rr] = LL[cc * (TA + 1) + rr];
        }
    }
}

template <typename T, int F, int TA, bool AL = true>
__device__ __forceinline__ void dwt_tile_fast(T* smem, const T* __restrict__ flo, const T* __restrict__ fhi,
                                              const T* __restrict__ src, int ldin, int nxin, int nyin,
                                              T* __restrict__ dst, int ldc, int Cx, int Cy, T* __restrict__ approx,
                                              int ox0, int oy0) {
    using D = DwtFast<T>;
    T* A = smem;                                    // [2 parities][NI rows][SA]
    T* B = A + 2 * D::ni(TA, F) * D::sa(TA, F);     // [2 row parities][NI / 2][SB]
    dwt_fast_stage<T, F, TA, AL>(A, src, ldin, nxin, nyin, ox0, oy0);
    __syncthreads();
    dwt_fast_passes<T, F, F, TA, D::off(F, AL), false>(A, B, flo, fhi, dst, ldc, Cx, Cy, approx, ox0, oy0);
}

// ------------------------------------------------ level kernels batched over the bases
// The coarse levels are tiny (level 3 of a 2048^2 image: 257^2 coefficients): launched basis by
// basis they are launch/latency bound -- a third of psi.dot's time at config #4.  The batched
// kernels take (band, wavelet basis) in blockIdx.z and a per-basis parameter table, so a level
// costs ONE launch for all bases (different filter lengths are a workgroup-uniform switch).
template <typename T> struct AnaPrm {           // analysis, one per (level, wavelet basis)
    long long coeff_off, in_off, approx_off;    // offsets inside alpha's band / the scratch bases
    int F, nxin, nyin, ldin, Cx, Cy, has_approx, pad_;
    T lo[MAXF], hi[MAXF];                       // dec_lo, dec_hi
};
template <typename T> struct SynPrm {           // synthesis, one per (level >= 1, wavelet basis)
    long long coeff_off, prev_off, out_off;
    int F, nax, nay, has_prev, ldp, nxw, nyw, ldo;
    T lo[MAXF], hi[MAXF];                       // rec_lo, rec_hi
};

// body of k_dwt_level for the tile (ox0, oy0); src / dst / approx already point at the band
template <typename T, int F, int TA>
__device__ __forceinline__ void dwt_tile(T* smem, const T* __restrict__ flo, const T* __restrict__ fhi,
                                         const T* __restrict__ src, int ldin, int nxin, int nyin,
                                         T* __restrict__ dst, int ldc, int Cx, int Cy, T* __restrict__ approx,
                                         int ox0, int oy0, bool allow_fast) {
    constexpr int VW0 = 16 / (int)sizeof(T);
    if (allow_fast && ldin % VW0 == 0 && nyin % VW0 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && TA % (2 * VW0) == 0) {
        dwt_tile_fast<T, F, TA, true>(smem, flo, fhi, src, ldin, nxin, nyin, dst, ldc, Cx, Cy, approx, ox0, oy0);
        return;
    }
    if constexpr (TA % (2 * VW0) == 0 && (2 * TA + F - 2) / 2 <= 64) {
        if (allow_fast) {
            dwt_tile_fast<T, F, TA, false>(smem, flo, fhi, src, ldin, nxin, nyin, dst, ldc, Cx, Cy, approx, ox0, oy0);
            return;
        }
    }
    constexpr int NI = 2 * TA + F - 2;        // input samples per tile edge
    // Both passes decimate by two (index 2q + d): with a plain row the 32 lanes of a half-wave touch only 16
    // banks (2-way conflicts, 46 % of the LDS-busy cycles in rocprofv3).  A and B are therefore stored
    // PARITY-SPLIT along the decimated axis -- even samples / rows first, odd ones AO / BO elements later --
    // so that a tap of fixed parity walks consecutive addresses.
    constexpr int SA = NI / 2;                // samples of one parity per tile row (NI is even)
    constexpr int AO = NI * SA;               // offset of the odd-sample half of A
    constexpr int SB = 2 * TA + 1;
    constexpr int BO = (NI / 2) * SB;         // offset of the odd-row half of B
    T* A = smem;                              // [NI][SA]   input tile  A[lx][ly]
    T* B = A + 2 * AO;                        // [NI][SB]   after the y pass  B[lx][q] (rows parity-split)
    T* LL = A;                                // [TA][TA+1] LL quadrant for the approx copy: aliases A, which is dead
                                              // after the y pass (42 -> 38 KB at F = 8: four workgroups per CU, not three)
    const int gx0 = 2 * ox0 + 1 - (F - 1), gy0 = 2 * oy0 + 1 - (F - 1);
    T lo[F], hi[F];
#pragma unroll
    for (int j = 0; j < F; ++j) { lo[j] = flo[j]; hi[j] = fhi[j]; }
    // 1. stage the input tile (zero extension outside the signal)
    constexpr int VW = 16 / (int)sizeof(T);                      // elements per 16-byte access
    if (ldin % VW == 0 && nyin % VW == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // rows are 16-byte aligned (the finest level of an image whose width is a multiple of 4):
        // aligned 16-byte loads of a slightly wider window -- 4x fewer load instructions, and every
        // vector lies entirely inside or entirely outside [0, nyin)
        struct alignas(16) Vec { T e[VW]; };
        const int ga = gy0 - (((gy0 % VW) + VW) % VW);           // window start, rounded down
        constexpr int NWV = (NI + 2 * (VW - 1) + VW - 1) / VW;   // vectors per tile row (upper bound)
        constexpr int NLV = (NI * NWV + 255) / 256;
        Vec stage[NLV];
#pragma unroll
        for (int k = 0; k < NLV; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int lx = e / NWV, w = e - lx * NWV;
            const int gx = gx0 + lx, gy = ga + VW * w;
            Vec v;
#pragma unroll
            for (int c = 0; c < VW; ++c) v.e[c] = 0;
            if (e < NI * NWV && gx >= 0 && gx < nxin && gy >= 0 && gy < nyin)
                v = *reinterpret_cast<const Vec*>(src + (size_t)gx * ldin + gy);
            stage[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NLV; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int lx = e / NWV, w = e - lx * NWV;
            const int ly0 = ga - gy0 + VW * w;
            if (e < NI * NWV) {
#pragma unroll
                for (int c = 0; c < VW; ++c)
                    if (ly0 + c >= 0 && ly0 + c < NI) A[((ly0 + c) & 1) * AO + lx * SA + ((ly0 + c) >> 1)] = stage[k].e[c];
            }
        }
    } else {
        constexpr int NLD = (NI * NI + 255) / 256;
        T stage[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int lx = e / NI, ly = e - lx * NI;
            const int gx = gx0 + lx, gy = gy0 + ly;
            T v = 0;
            if (e < NI * NI && gx >= 0 && gx < nxin && gy >= 0 && gy < nyin) v = src[(size_t)gx * ldin + gy];
            stage[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int lx = e / NI, ly = e - lx * NI;
            if (e < NI * NI) A[(ly & 1) * AO + lx * SA + (ly >> 1)] = stage[k];
        }
    }
    __syncthreads();
    // 2. y pass: B[lx][q] = sum_j filt[j] A[lx][2q' + F-1-j]   (q < TA: lo, q >= TA: hi)
    for (int e = threadIdx.x; e < NI * TA; e += 256) {
        const int lx = e / TA, qq = e - lx * TA;
        const T* a = A + lx * SA + qq;
        T sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < F; ++j) {                          // sample 2 qq + d, d = F-1-j
            const int d = F - 1 - j;
            const T v = a[(d & 1) * AO + (d >> 1)];
            sl += lo[j] * v; sh += hi[j] * v;
        }
        T* brow = B + (lx & 1) * BO + (lx >> 1) * SB;
        brow[qq] = sl;
        brow[TA + qq] = sh;
    }
    __syncthreads();
    // 3. x pass + store: out[r][c], r < 2TA (y coefficient, lo|hi), c (x coefficient) lo & hi
    for (int e = threadIdx.x; e < 2 * TA * TA; e += 256) {
        const int r = e / TA, cc = e - r * TA;
        const T* b = B + cc * SB + r;
        T sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < F; ++j) {                          // row 2 cc + d, d = F-1-j
            const int d = F - 1 - j;
            const T v = b[(d & 1) * BO + (d >> 1) * SB];
            sl += lo[j] * v; sh += hi[j] * v;
        }
        const bool hiy = r >= TA;
        const int rr = hiy ? r - TA : r;
        const int gy = oy0 + rr, gx = ox0 + cc;
        if (gy < Cy && gx < Cx) {
            T* row = dst + (size_t)((hiy ? Cy : 0) + gy) * ldc;
            row[gx] = sl;
            row[Cx + gx] = sh;
        }
        if (!hiy) LL[cc * (TA + 1) + rr] = sl;
    }
    if (approx) {
        __syncthreads();
        T* ap = approx;
        for (int e = threadIdx.x; e < TA * TA; e += 256) {
            const int cc = e / TA, rr = e - cc * TA;           // rr (y) fastest: coalesced
            if (ox0 + cc < Cx && oy0 + rr < Cy) ap[(size_t)(ox0 + cc) * Cy + oy0 + rr] = LL[cc * (TA + 1) + rr];
        }
    }
}

template <typename T, int TA, int FMAX>
__global__ void __launch_bounds__(256)
k_dwt_batched(const T* __restrict__ in_base, size_t in_band, T* __restrict__ alpha, size_t aband, int ldc,
              T* __restrict__ scr_out, size_t sband, const AnaPrm<T>* __restrict__ prm, int nwb) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const bool allow_fast = !(nwb & (1 << 30));     // PFB_DWT_FAST=0 (A/B): plain tiles only
    const TileId tl = xcd_tile(!(nwb & (1 << 29))); // PFB_PSI_XCD=0 (A/B): launch order
    nwb &= ~(3 << 29);
    const int j = tl.z % nwb, band = tl.z / nwb;
    const AnaPrm<T>& P = prm[j];
    const int ox0 = tl.x * TA, oy0 = tl.y * TA;
    if (ox0 >= P.Cx || oy0 >= P.Cy) return;
    const T* src = in_base + P.in_off + (size_t)band * in_band;
    T* dst = alpha + (size_t)band * aband + P.coeff_off;
    T* approx = P.has_approx ? scr_out + P.approx_off + (size_t)band * sband : nullptr;
    switch (P.F) {
#define X(FF) case FF: if constexpr (FF <= FMAX) dwt_tile<T, FF, TA>(reinterpret_cast<T*>(smem_raw), P.lo, P.hi, src, P.ldin, P.nxin, P.nyin, dst, ldc, P.Cx, P.Cy, approx, ox0, oy0, allow_fast); break;
        X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)
#undef X
        default: break;
    }
}


// Finest analysis level of ALL bases in one kernel (16-byte aligned image rows).  Launched basis by basis, level 0
// reads the image once per wavelet basis and the 'self' basis costs a transpose kernel of its own (read + write of the
// image): here a workgroup stages its image tile ONCE with the halo of the longest filter, writes the 'self' plane
// straight from that tile and runs the two passes of every wavelet basis on it.
template <typename T, int TA, int FS>
__global__ void __launch_bounds__(256)
k_dwt_l1_fused(const T* __restrict__ x, size_t xband, int ldin, int nxin, int nyin, T* __restrict__ alpha, size_t aband,
               int ldc, T* __restrict__ scr_out, size_t sband, const AnaPrm<T>* __restrict__ prm, int nwb,
               long long self_off) {
    using D = DwtFast<T>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* A = reinterpret_cast<T*>(smem_raw);
    constexpr int NIS = D::ni(TA, FS), SA = D::sa(TA, FS), AO = NIS * SA, OFFS = D::off(FS, true);
    T* B = A + 2 * AO;
    const TileId tl = xcd_tile(!(nwb & (1 << 29)));
    nwb &= ~(1 << 29);
    const int band = tl.z;
    const int ox0 = tl.x * TA, oy0 = tl.y * TA;
    dwt_fast_stage<T, FS, TA, true>(A, x + (size_t)band * xband, ldin, nxin, nyin, ox0, oy0);
    __syncthreads();
    if (self_off >= 0 && 2 * ox0 < nxin && 2 * oy0 < nyin) {
        // 'self': alpha_self[gy][gx] = x[gx][gy] for the tile's own 2TA x 2TA pixels (psi.py:199-202)
        T* sp = alpha + (size_t)band * aband + self_off;
        constexpr int W = 2 * TA, RPT = W / (256 / W);
        const int gxl = threadIdx.x % W, yq = threadIdx.x / W;
        const int gx = 2 * ox0 + gxl;
        const T* a = A + (gxl + FS - 2) * SA;
        if (gx < nxin) {
#pragma unroll
            for (int c = 0; c < RPT; ++c) {
                const int yy = yq * RPT + c, sidx = yy + FS - 2 + OFFS, gy = 2 * oy0 + yy;
                if (gy < nyin) sp[(size_t)gy * ldc + gx] = a[(sidx & 1) * AO + (sidx >> 1)];
            }
        }
    }
    for (int j = 0; j < nwb; ++j) {
        const AnaPrm<T>& P = prm[j];
        if (ox0 >= P.Cx || oy0 >= P.Cy) continue;                   // workgroup-uniform
        T* dst = alpha + (size_t)band * aband + P.coeff_off;
        T* approx = P.has_approx ? scr_out + P.approx_off + (size_t)band * sband : nullptr;
        switch (P.F) {
#define X(FF) case FF: if constexpr (FF <= FS) dwt_fast_passes<T, FF, FS, TA, OFFS, true>(A, B, P.lo, P.hi, dst, ldc, P.Cx, P.Cy, approx, ox0, oy0); break;
            X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)
#undef X
            default: break;
        }
    }
}

// ----------------------------------------------------------------- synthesis level
// ---- staging of a synthesis tile's coefficients: C[ry][cx], ry, cx < 2NC (lo | hi halves in both)
// When a C row is at least a wavefront wide (fp32 tiles: 2NC >= 64) a WAVE takes a row: the row index is
// wave-uniform (scalar address / bounds arithmetic), the lane is the column, and only the 2NC - 64 columns
// beyond the wavefront go through the flat per-element mapping.  The flat mapping cost ~22 VALU
// instructions per element (division by 2NC, quadrant selects, 64-bit address) -- more than the synthesis
// arithmetic itself (rocprofv3: 81.8 M VALU wave-instructions per finest-level launch against ~16 M of FMAs).
template <int NC> struct StageCfg {
    static constexpr bool ROWS = 2 * NC >= 64;
    static constexpr int NR = ROWS ? (2 * NC + 3) / 4 : 0;           // rows per wave (4 waves)
    static constexpr int NREM = ROWS ? 2 * NC - 64 : 0;              // columns beyond the 64 lanes
    static constexpr int NLR = (NREM * 2 * NC + 255) / 256;
    static constexpr int NLD = ROWS ? NR + NLR : (4 * NC * NC + 255) / 256;
};
template <typename T, int NC>
__device__ __forceinline__ void coef_load(T (&stage)[StageCfg<NC>::NLD], const T* __restrict__ src, int ldc,
                                          int nax, int nay, int mx0, int my0, bool skip_ll, int tid) {
    using S = StageCfg<NC>;
    if constexpr (S::ROWS) {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const bool hx = lane >= NC;
        const int gx = mx0 + (hx ? lane - NC : lane);
        const int coloff = (hx ? nax : 0) + gx;
        const bool okx = gx < nax;
#pragma unroll
        for (int k = 0; k < S::NR; ++k) {
            const int ry = wv + 4 * k;                               // wave-uniform
            const bool hy = ry >= NC;
            const int gy = my0 + (hy ? ry - NC : ry);
            T v = 0;
            if (ry < 2 * NC && gy < nay && okx && !(skip_ll && !hy && !hx))
                v = src[(size_t)((hy ? nay : 0) + gy) * ldc + coloff];
            stage[k] = v;
        }
        if constexpr (S::NREM > 0) {
#pragma unroll
            for (int k = 0; k < S::NLR; ++k) {
                const int e = tid + 256 * k;
                const int ry = e / S::NREM, cx = 64 + (e - ry * S::NREM);   // cx >= 64 > NC: the hi-x half
                const bool hy = ry >= NC;
                const int gy = my0 + (hy ? ry - NC : ry), gx2 = mx0 + cx - NC;
                T v = 0;
                if (e < S::NREM * 2 * NC && gy < nay && gx2 < nax)
                    v = src[(size_t)((hy ? nay : 0) + gy) * ldc + nax + gx2];
                stage[S::NR + k] = v;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < S::NLD; ++k) {
            const int e = tid + 256 * k;
            const int ry = e / (2 * NC), cx = e - ry * (2 * NC);
            const bool hy = ry >= NC, hx = cx >= NC;
            const int gy = my0 + (hy ? ry - NC : ry), gx = mx0 + (hx ? cx - NC : cx);
            T v = 0;
            if (e < 4 * NC * NC && gy < nay && gx < nax && !(skip_ll && !hy && !hx))
                v = src[(size_t)((hy ? nay : 0) + gy) * ldc + (hx ? nax : 0) + gx];
            stage[k] = v;
        }
    }
}
template <typename T, int NC, int SC>
__device__ __forceinline__ void coef_store(T* C, const T (&stage)[StageCfg<NC>::NLD], bool skip_ll, int tid) {
    using S = StageCfg<NC>;
    if constexpr (S::ROWS) {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll
        for (int k = 0; k < S::NR; ++k) {
            const int ry = wv + 4 * k;
            if (ry < 2 * NC && !(skip_ll && ry < NC && lane < NC)) C[ry * SC + lane] = stage[k];
        }
        if constexpr (S::NREM > 0) {
#pragma unroll
            for (int k = 0; k < S::NLR; ++k) {
                const int e = tid + 256 * k;
                const int ry = e / S::NREM, cx = 64 + (e - ry * S::NREM);
                if (e < S::NREM * 2 * NC) C[ry * SC + cx] = stage[S::NR + k];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < S::NLD; ++k) {
            const int e = tid + 256 * k;
            const int ry = e / (2 * NC), cx = e - ry * (2 * NC);
            if (e < 4 * NC * NC && !(skip_ll && ry < NC && cx < NC)) C[ry * SC + cx] = stage[k];
        }
    }
}

// coeffs : this level's (2 nay, 2 nax) block, ld = ldc (y-major)
// prev   : if non-null, the approx quadrant is prev[c][r] (previous level's image,
//          row-major ld = ldp) instead of coeffs[r][c]      (wavelets.py:303-309)
// out    : image (nxw, nyw) row-major ld = ldo; ACC adds (sum over bases, psi.py:252)
template <typename T, int F, int TS, bool ACC>
__global__ void __launch_bounds__(256)
k_idwt_level(const T* __restrict__ coeffs, size_t c_band, int ldc, int nax, int nay,
             const T* __restrict__ prev, size_t p_band, int ldp,
             T* __restrict__ out, size_t o_band, int ldo, int nxw, int nyw, Filt<T> f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int h = F / 2;
    constexpr int NC = TS / 2 + h - 1;        // coefficients needed per half and tile edge
    constexpr int SC = 2 * NC + 1;
    constexpr int ST = TS + 1;
    T* C = reinterpret_cast<T*>(smem);        // [2NC][SC]  C[ry][cx]  (lo|hi in both)
    T* Tm = C + 2 * NC * SC;                  // [2NC][ST]  after the x pass  Tm[ry][ox]
    const T* src = coeffs + (size_t)blockIdx.z * c_band;
    const T* pv = prev ? prev + (size_t)blockIdx.z * p_band : nullptr;
    T* dst = out + (size_t)blockIdx.z * o_band;
    const int ix0 = blockIdx.x * TS, iy0 = blockIdx.y * TS;
    const int mx0 = ix0 / 2, my0 = iy0 / 2;
    T lo[F], hi[F];
#pragma unroll
    for (int j = 0; j < F; ++j) { lo[j] = f.lo[j]; hi[j] = f.hi[j]; }
    // 1. stage coefficients (all loads first); rows/cols beyond (nay, nax) are only used by
    //    cropped outputs.  coeffs are read row-wise (cx fastest), prev column-wise (ry fastest).
    T stage[StageCfg<NC>::NLD];
    coef_load<T, NC>(stage, src, ldc, nax, nay, mx0, my0, pv != nullptr, (int)threadIdx.x);
    constexpr int NLP = (NC * NC + 255) / 256;
    T stagep[NLP];
    if (pv) {
#pragma unroll
        for (int k = 0; k < NLP; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int cx = e / NC, ry = e - cx * NC;              // ry fastest: prev is read along y
            const int gy = my0 + ry, gx = mx0 + cx;
            T v = 0;
            if (e < NC * NC && gy < nay && gx < nax) v = pv[(size_t)gx * ldp + gy];
            stagep[k] = v;
        }
    }
    coef_store<T, NC, SC>(C, stage, pv != nullptr, (int)threadIdx.x);
    if (pv) {
#pragma unroll
        for (int k = 0; k < NLP; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int cx = e / NC, ry = e - cx * NC;
            if (e < NC * NC) C[ry * SC + cx] = stagep[k];
        }
    }
    __syncthreads();
    // 2. x pass: Tm[ry][ox] = sum_j lo[2j+p] C[ry][m+h-1-j] + sum_j hi[2j+p] C[ry][NC+m+h-1-j]
    //    (one thread makes the even/odd output pair from the same h taps)
    for (int e = threadIdx.x; e < 2 * NC * (TS / 2); e += 256) {
        const int ry = e / (TS / 2), m = e - ry * (TS / 2);
        const T* c = C + ry * SC + m + h - 1;
        T sl0 = 0, sh0 = 0, sl1 = 0, sh1 = 0;
#pragma unroll
        for (int j = 0; j < h; ++j) {
            const T a = c[-j], d = c[NC - j];
            sl0 += lo[2 * j] * a;     sh0 += hi[2 * j] * d;
            sl1 += lo[2 * j + 1] * a; sh1 += hi[2 * j + 1] * d;
        }
        Tm[ry * ST + 2 * m] = sl0 + sh0;
        Tm[ry * ST + 2 * m + 1] = sl1 + sh1;
    }
    __syncthreads();
    // 3. y pass + store (iy fastest); a thread makes the pair (2m, 2m+1) of one image row
    for (int e = threadIdx.x; e < TS * (TS / 2); e += 256) {
        const int ox = e / (TS / 2), m = e - ox * (TS / 2);
        const T* t = Tm + (m + h - 1) * ST + ox;
        T sl0 = 0, sh0 = 0, sl1 = 0, sh1 = 0;
#pragma unroll
        for (int j = 0; j < h; ++j) {
            const T a = t[-j * ST], d = t[(NC - j) * ST];
            sl0 += lo[2 * j] * a;     sh0 += hi[2 * j] * d;
            sl1 += lo[2 * j + 1] * a; sh1 += hi[2 * j + 1] * d;
        }
        const int gx = ix0 + ox, gy = iy0 + 2 * m;
        if (gx < nxw) {
            T* q = dst + (size_t)gx * ldo + gy;
            if (gy < nyw)     { if (ACC) q[0] += sl0 + sh0; else q[0] = sl0 + sh0; }
            if (gy + 1 < nyw) { if (ACC) q[1] += sl1 + sh1; else q[1] = sl1 + sh1; }
        }
    }
}

// body of k_idwt_level (store, no accumulate) for the output tile (ix0, iy0)
template <typename T, int F, int TS>
__device__ __forceinline__ void idwt_tile_store(T* smem, const T* __restrict__ flo, const T* __restrict__ fhi,
                                                const T* __restrict__ src, int ldc, int nax, int nay,
                                                const T* __restrict__ pv, int ldp, T* __restrict__ dst, int ldo,
                                                int nxw, int nyw, int ix0, int iy0) {

    constexpr int h = F / 2;
    constexpr int NC = TS / 2 + h - 1;        // coefficients needed per half and tile edge
    constexpr int SC = 2 * NC + 1;
    constexpr int ST = TS + 1;
    T* C = smem;                              // [2NC][SC]  C[ry][cx]  (lo|hi in both)
    T* Tm = C + 2 * NC * SC;                  // [2NC][ST]  after the x pass  Tm[ry][ox]
    const int mx0 = ix0 / 2, my0 = iy0 / 2;
    T lo[F], hi[F];
#pragma unroll
    for (int j = 0; j < F; ++j) { lo[j] = flo[j]; hi[j] = fhi[j]; }
    // 1. stage coefficients (all loads first); rows/cols beyond (nay, nax) are only used by
    //    cropped outputs.  coeffs are read row-wise (cx fastest), prev column-wise (ry fastest).
    T stage[StageCfg<NC>::NLD];
    coef_load<T, NC>(stage, src, ldc, nax, nay, mx0, my0, pv != nullptr, (int)threadIdx.x);
    constexpr int NLP = (NC * NC + 255) / 256;
    T stagep[NLP];
    if (pv) {
#pragma unroll
        for (int k = 0; k < NLP; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int cx = e / NC, ry = e - cx * NC;              // ry fastest: prev is read along y
            const int gy = my0 + ry, gx = mx0 + cx;
            T v = 0;
            if (e < NC * NC && gy < nay && gx < nax) v = pv[(size_t)gx * ldp + gy];
            stagep[k] = v;
        }
    }
    coef_store<T, NC, SC>(C, stage, pv != nullptr, (int)threadIdx.x);
    if (pv) {
#pragma unroll
        for (int k = 0; k < NLP; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int cx = e / NC, ry = e - cx * NC;
            if (e < NC * NC) C[ry * SC + cx] = stagep[k];
        }
    }
    __syncthreads();
    // 2. x pass: Tm[ry][ox] = sum_j lo[2j+p] C[ry][m+h-1-j] + sum_j hi[2j+p] C[ry][NC+m+h-1-j]
    //    (one thread makes the even/odd output pair from the same h taps)
    for (int e = threadIdx.x; e < 2 * NC * (TS / 2); e += 256) {
        const int ry = e / (TS / 2), m = e - ry * (TS / 2);
        const T* c = C + ry * SC + m + h - 1;
        T sl0 = 0, sh0 = 0, sl1 = 0, sh1 = 0;
#pragma unroll
        for (int j = 0; j < h; ++j) {
            const T a = c[-j], d = c[NC - j];
            sl0 += lo[2 * j] * a;     sh0 += hi[2 * j] * d;
            sl1 += lo[2 * j + 1] * a; sh1 += hi[2 * j + 1] * d;
        }
        Tm[ry * ST + 2 * m] = sl0 + sh0;
        Tm[ry * ST + 2 * m + 1] = sl1 + sh1;
    }
    __syncthreads();
    // 3. y pass + store (iy fastest); a thread makes the pair (2m, 2m+1) of one image row
    for (int e = threadIdx.x; e < TS * (TS / 2); e += 256) {
        const int ox = e / (TS / 2), m = e - ox * (TS / 2);
        const T* t = Tm + (m + h - 1) * ST + ox;
        T sl0 = 0, sh0 = 0, sl1 = 0, sh1 = 0;
#pragma unroll
        for (int j = 0; j < h; ++j) {
            const T a = t[-j * ST], d = t[(NC - j) * ST];
            sl0 += lo[2 * j] * a;     sh0 += hi[2 * j] * d;
            sl1 += lo[2 * j + 1] * a; sh1 += hi[2 * j + 1] * d;
        }
        const int gx = ix0 + ox, gy = iy0 + 2 * m;
        if (gx < nxw) {
            T* q = dst + (size_t)gx * ldo + gy;
            if (gy < nyw)     { q[0] = sl0 + sh0; }
            if (gy + 1 < nyw) { q[1] = sl1 + sh1; }
        }
    }
}

template <typename T, int TS, int FMAX>
__global__ void __launch_bounds__(256)
k_idwt_batched(const T* __restrict__ alpha, size_t aband, int ldc, const T* __restrict__ scr_prev, size_t sband,
               T* __restrict__ out_base, size_t oband, const SynPrm<T>* __restrict__ prm, int nwb) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const TileId tl = xcd_tile(!(nwb & (1 << 29)));
    nwb &= ~(1 << 29);
    const int j = tl.z % nwb, band = tl.z / nwb;
    const SynPrm<T>& P = prm[j];
    const int ix0 = tl.x * TS, iy0 = tl.y * TS;
    if (ix0 >= P.nxw || iy0 >= P.nyw) return;
    const T* src = alpha + (size_t)band * aband + P.coeff_off;
    const T* pv = P.has_prev ? scr_prev + P.prev_off + (size_t)band * sband : nullptr;
    T* dst = out_base + P.out_off + (size_t)band * oband;
    switch (P.F) {
#define X(FF) case FF: if constexpr (FF <= FMAX) idwt_tile_store<T, FF, TS>(reinterpret_cast<T*>(smem_raw), P.lo, P.hi, src, ldc, P.nax, P.nay, pv, P.ldp, dst, P.ldo, P.nxw, P.nyw, ix0, iy0); break;
        X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)
#undef X
        default: break;
    }
}


template <typename T>
__global__ void __launch_bounds__(256) k_fill(T* p, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// --------------------------------------------------------------------- prox / PD
// dual_update_numba (prox_21m.py:76-103): vt = vp + sigma v; a = |sum_band vt / sigma|;
// v = vt * (a != 0 ? 1 - max(a - lam w / sigma, 0)/a : 1);  optionally vp_out = 2 v - vp
template <typename T>
__global__ void __launch_bounds__(256)
k_dual_update(const T* vp, T* __restrict__ v, const T* __restrict__ w, T lam, T sigma,
              int nband, size_t nper, T* vp_out) {        // vp_out may alias vp
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nper;
         i += (size_t)gridDim.x * blockDim.x) {
        T sum = 0;
        for (int b = 0; b < nband; ++b) sum += vp[(size_t)b * nper + i] + sigma * v[(size_t)b * nper + i];
        const T a = fabs(sum / sigma);
        T fac = 1;
        if (a != T(0)) {
            const T soft = fmax(a - lam * w[i] / sigma, T(0));
            fac = T(1) - soft / a;
        }
        for (int b = 0; b < nband; ++b) {
            const size_t k = (size_t)b * nper + i;
            const T vpk = vp[k];
            const T vt = vpk + sigma * v[k];
            const T vn = (a != T(0)) ? vt * fac : vt;
            v[k] = vn;
            if (vp_out) vp_out[k] = T(2) * vn - vpk;
        }
    }
}

// Same arithmetic, 16-byte accesses, every band's vp / v held in registers between the band sum and
// the threshold (one read of each instead of two).  NB = nband (compile time, <= 8); V = 16 / sizeof(T).
template <typename T, int NB, bool NTL = false>
__global__ void __launch_bounds__(256)
k_dual_update_vec(const T* vp, T* __restrict__ v, const T* __restrict__ w, T lam, T sigma,
                  size_t nper, T* vp_out) {              // vp_out may alias vp
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    const size_t nvec = nper / V;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (size_t)gridDim.x * blockDim.x) {
        VT a_vp[NB], a_v[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (NTL) {        // vp and psi^H(x) are read once per iteration: keep them out of the Infinity Cache, v' stays
                a_vp[b] = __builtin_nontemporal_load(reinterpret_cast<const VT*>(vp + (size_t)b * nper) + i);
                a_v[b] = __builtin_nontemporal_load(reinterpret_cast<const VT*>(v + (size_t)b * nper) + i);
            } else {
                a_vp[b] = reinterpret_cast<const VT*>(vp + (size_t)b * nper)[i];
                a_v[b] = reinterpret_cast<const VT*>(v + (size_t)b * nper)[i];
            }
        }
        const VT wv = reinterpret_cast<const VT*>(w)[i];
        T fac[V];
        bool nz[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            T sum = 0;
#pragma unroll
            for (int b = 0; b < NB; ++b) sum += a_vp[b][e] + sigma * a_v[b][e];
            const T a = fabs(sum / sigma);
            nz[e] = a != T(0);
            fac[e] = 1;
            if (nz[e]) {
                const T soft = fmax(a - lam * wv[e] / sigma, T(0));
                fac[e] = T(1) - soft / a;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            VT vn, vo;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T vt = a_vp[b][e] + sigma * a_v[b][e];
                vn[e] = nz[e] ? vt * fac[e] : vt;
                vo[e] = T(2) * vn[e] - a_vp[b][e];
            }
            reinterpret_cast<VT*>(v + (size_t)b * nper)[i] = vn;
            if (vp_out) reinterpret_cast<VT*>(vp_out + (size_t)b * nper)[i] = vo;
        }
    }
}

// Band-sharded dual update (bands split over GPUs): phase 1 forms the LOCAL band sum of
// vtilde = vp + sigma v per coefficient; after an all-reduce(sum) of that plane over the
// ranks, phase 2 applies the same soft threshold with the GLOBAL sum (prox_21m.py:95-103).
template <typename T>
__global__ void __launch_bounds__(256)
k_dual_bandsum(const T* __restrict__ vp, const T* __restrict__ v, T sigma, int nband, size_t nper,
               size_t bstride, T* __restrict__ sum_out) {
    // bstride: elements between the bands of vp / v (= nper for the whole plane; a CHUNK of the plane is the same
    // call on shifted pointers with the chunk's length as nper)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nper;
         i += (size_t)gridDim.x * blockDim.x) {
        T sum = 0;
        for (int b = 0; b < nband; ++b) sum += vp[(size_t)b * bstride + i] + sigma * v[(size_t)b * bstride + i];
        sum_out[i] = sum;
    }
}
template <typename T>
__global__ void __launch_bounds__(256)
k_dual_apply(const T* vp, T* __restrict__ v, const T* __restrict__ w, const T* __restrict__ sum_in,
             T lam, T sigma, int nband, size_t nper, size_t bstride, T* vp_out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nper;
         i += (size_t)gridDim.x * blockDim.x) {
        const T a = fabs(sum_in[i] / sigma);
        T fac = 1;
        if (a != T(0)) {
            const T soft = fmax(a - lam * w[i] / sigma, T(0));
            fac = T(1) - soft / a;
        }
        for (int b = 0; b < nband; ++b) {
            const size_t k = (size_t)b * bstride + i;
            const T vpk = vp[k];
            const T vt = vpk + sigma * v[k];
            const T vn = (a != T(0)) ? vt * fac : vt;
            v[k] = vn;
            if (vp_out) vp_out[k] = T(2) * vn - vpk;
        }
    }
}

// prox_21m_numba (prox_21m.py:31-61)
template <typename T>
__global__ void __launch_bounds__(256)
k_prox_21m(const T* __restrict__ v, T* __restrict__ res, const T* __restrict__ w, T lam, T sigma,
           int nband, size_t nper) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nper;
         i += (size_t)gridDim.x * blockDim.x) {
        T sum = 0;
        for (int b = 0; b < nband; ++b) sum += v[(size_t)b * nper + i];
        const T t = sum / sigma;
        if (t == T(0)) {
            for (int b = 0; b < nband; ++b) res[(size_t)b * nper + i] = 0;
            continue;
        }
        const T a = fabs(t);
        const T soft = fmax(a - lam * w[i] / sigma, T(0));
        for (int b = 0; b < nband; ++b) res[(size_t)b * nper + i] = v[(size_t)b * nper + i] * soft / a / sigma;
    }
}

// ---- the band-l2-NORM variants of pfb/prox/prox_21.py (the "m" kernels above threshold |sum over bands|, these the
// Euclidean norm over bands).  Off the hot path (the live workers use the "m" forms): plain grid-stride kernels, the
// norm accumulated in fp64.
// prox_21_numba (prox_21.py:23-48): a = ||v[:, i]|| / sigma; result = 0 where a == 0, else v max(a - lam w / sigma, 0) / a / sigma
template <typename T>
__global__ void __launch_bounds__(256)
k_prox_21_l2(const T* __restrict__ v, T* __restrict__ res, const T* __restrict__ w, T lam, T sigma, int nband, size_t nper) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nper; i += (size_t)gridDim.x * blockDim.x) {
        double ss = 0.0;
        for (int b = 0; b < nband; ++b) { const double t = (double)v[(size_t)b * nper + i]; ss += t * t; }
        const T a = (T)sqrt(ss) / sigma;
        if (a == T(0)) {
            for (int b = 0; b < nband; ++b) res[(size_t)b * nper + i] = T(0);
            continue;
        }
        const T soft = fmax(a - lam * w[i] / sigma, T(0));
        for (int b = 0; b < nband; ++b) res[(size_t)b * nper + i] = v[(size_t)b * nper + i] * soft / a / sigma;
    }
}
// dual_update_numba of prox_21.py:62-88, in place on v: vt = vp + sigma v; a = ||vt[:, i]|| / sigma; v = vt, and where
// a != 0: v *= 1 - max(a - lam w / sigma, 0) / a
template <typename T>
__global__ void __launch_bounds__(256)
k_dual_update_l2(const T* __restrict__ vp, T* __restrict__ v, const T* __restrict__ w, T lam, T sigma, int nband, size_t nper) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nper; i += (size_t)gridDim.x * blockDim.x) {
        double ss = 0.0;
        for (int b = 0; b < nband; ++b) {
            const size_t k = (size_t)b * nper + i;
            const double t = (double)(vp[k] + sigma * v[k]);
            ss += t * t;
        }
        const T a = (T)sqrt(ss) / sigma;
        T fac = T(1);
        // 1 - max(a - c, 0) / a written as min(c, a) / a: the same number without the cancellation that costs fp32 three
        // digits when the threshold c = lam w / sigma is small against a (the reference computes this in fp64 only)
        if (a != T(0)) fac = fmin(lam * w[i] / sigma, a) / a;
        for (int b = 0; b < nband; ++b) {
            const size_t k = (size_t)b * nper + i;
            const T vt = vp[k] + sigma * v[k];
            v[k] = (a != T(0)) ? vt * fac : vt;
        }
    }
}

// x = xp - tau (xout + g) ; positivity ; norm_diff partials + any(x)   (primal_dual.py:139-150)
// xprev (optional): xout is formed as 2 xout - xprev on the fly -- with a LINEAR synthesis psi^H(2 v - vp) =
// 2 psi^H(v) - psi^H(vp), and psi^H(vp) is the previous iteration's psi^H(v): the coefficient cube 2 v - vp
// (primal_dual.py:137) is then never written nor read.  gsub (optional): g - gsub is the gradient (the data term of
// grad(x) = conv(x) - dirty, workers/spotless.py:259-260, subtracted here instead of in a pass of its own).
template <typename T>
__global__ void __launch_bounds__(256)
k_pd_primal(const T* __restrict__ xp, const T* __restrict__ xout, const T* __restrict__ xprev,
            const T* __restrict__ g, const T* __restrict__ gsub, T tau,
            int positivity, int nband, size_t npix, T* __restrict__ x, double* __restrict__ ws) {
    __shared__ double red[3 * 4];
    double acc[3] = {0.0, 0.0, 0.0};
    auto value = [&](size_t k) -> T {
        T gk = g ? g[k] : T(0);
        if (gsub) gk -= gsub[k];
        T xo = xout[k];
        if (xprev) xo = T(2) * xo - xprev[k];
        return xp[k] - tau * (xo + gk);
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
         i += (size_t)gridDim.x * blockDim.x) {
        bool kill = false;
        if (positivity == 2) {
            for (int b = 0; b < nband; ++b)
                if (value((size_t)b * npix + i) <= T(0)) kill = true;
        }
        for (int b = 0; b < nband; ++b) {
            const size_t k = (size_t)b * npix + i;
            T val = value(k);
            if (positivity == 1 && val < T(0)) val = 0;
            if (kill) val = 0;
            x[k] = val;
            const double d = (double)val - (double)xp[k];
            acc[0] += d * d;
            acc[1] += (double)val * (double)val;
            acc[2] += (val != T(0)) ? 1.0 : 0.0;
        }
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        for (int q = 0; q < 3; ++q) ws[(size_t)q * gridDim.x + blockIdx.x] = acc[q];
    }
}

// the same with V pixels per thread and step (16-byte accesses; npix a multiple of V, 16-byte aligned arrays): the
// scalar kernel above moved 402 MB in 119 us at config #4, 3.4 TB/s
template <typename T, int V>
__global__ void __launch_bounds__(256)
k_pd_primal_vec(const T* __restrict__ xp, const T* __restrict__ xout, const T* __restrict__ xprev,
                const T* __restrict__ g, const T* __restrict__ gsub, T tau,
                int positivity, int nband, size_t nvec, T* __restrict__ x, double* __restrict__ ws) {
    struct alignas(V * sizeof(T)) Vec { T e[V]; };
    __shared__ double red[3 * 4];
    double acc[3] = {0.0, 0.0, 0.0};
    auto ldv = [](const T* p, size_t k) { return reinterpret_cast<const Vec*>(p)[k]; };
    auto value = [&](size_t k, Vec& xpv) -> Vec {
        xpv = ldv(xp, k);
        Vec xo = ldv(xout, k), gk, r;
#pragma unroll
        for (int c = 0; c < V; ++c) gk.e[c] = 0;
        if (g) gk = ldv(g, k);
        if (gsub) { const Vec s2 = ldv(gsub, k);
#pragma unroll
            for (int c = 0; c < V; ++c) gk.e[c] -= s2.e[c]; }
        if (xprev) { const Vec pr = ldv(xprev, k);
#pragma unroll
            for (int c = 0; c < V; ++c) xo.e[c] = T(2) * xo.e[c] - pr.e[c]; }
#pragma unroll
        for (int c = 0; c < V; ++c) r.e[c] = xpv.e[c] - tau * (xo.e[c] + gk.e[c]);
        return r;
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        bool kill[V];
#pragma unroll
        for (int c = 0; c < V; ++c) kill[c] = false;
        if (positivity == 2) {
            for (int b = 0; b < nband; ++b) {
                Vec dummy;
                const Vec v = value((size_t)b * nvec + i, dummy);
#pragma unroll
                for (int c = 0; c < V; ++c) if (v.e[c] <= T(0)) kill[c] = true;
            }
        }
        for (int b = 0; b < nband; ++b) {
            const size_t k = (size_t)b * nvec + i;
            Vec xpv;
            Vec val = value(k, xpv);
#pragma unroll
            for (int c = 0; c < V; ++c) {
                if (positivity == 1 && val.e[c] < T(0)) val.e[c] = 0;
                if (kill[c]) val.e[c] = 0;
                const double d = (double)val.e[c] - (double)xpv.e[c];
                acc[0] += d * d;
                acc[1] += (double)val.e[c] * (double)val.e[c];
                acc[2] += (val.e[c] != T(0)) ? 1.0 : 0.0;
            }
            reinterpret_cast<Vec*>(x)[k] = val;
        }
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        for (int q = 0; q < 3; ++q) ws[(size_t)q * gridDim.x + blockIdx.x] = acc[q];
    }
}

__global__ void __launch_bounds__(256)
k_final_sum3(const double* __restrict__ ws, int G, int nq, double* __restrict__ out) {
    __shared__ double red[4];
    for (int q = 0; q < nq; ++q) {
        double acc[1] = {0.0};
        for (int g = threadIdx.x; g < G; g += blockDim.x) acc[0] += ws[(size_t)q * G + g];
        block_sum<1>(acc, red);
        if (threadIdx.x == 0) out[q] = acc[0];
    }
}

// ------------------------------------------------------------------- host drivers
// ---------------------------------------------------- fused finest synthesis level
// psi.hdot sums the bases: x = sum_b idwt_b(alpha_b).  Run basis by basis, the finest level
// read-modify-writes the image once per basis (5 bases: 0.67 GB of the ~1 GB the whole hdot
// moves at 2048^2 x 4).  Here ONE kernel walks all bases for its image tile, accumulates the
// finest-level results in registers and stores the image once.
template <typename T> struct FinBasis {
    long long coeff_off;        // level-0 block origin inside a band's coefficient planes
    long long prev_off;         // this basis' level-1 partial image inside fin_scratch (per band 0)
    int K, F, nax, nay, has_prev, ldp;
    T lo[MAXF], hi[MAXF];       // rec_lo, rec_hi
};

// one basis' contribution to the thread's output pairs (same staging / passes as k_idwt_level)
template <typename T, int F, int TS>
__device__ __forceinline__ void idwt_tile_acc(T* smem, const FinBasis<T>& B, const T* __restrict__ src, int ldc,
                                              const T* __restrict__ pv, int ix0, int iy0, int tid,
                                              T (&acc)[TS * (TS / 2) / 256][2]) {
    constexpr int h = F / 2;
    constexpr int NC = TS / 2 + h - 1;
    constexpr int SC = 2 * NC + 1;
    constexpr int ST = TS + 1;
    T* C = smem;
    T* Tm = C + 2 * NC * SC;
    const int nax = B.nax, nay = B.nay, ldp = B.ldp;
    const int mx0 = ix0 / 2, my0 = iy0 / 2;
    T lo[F], hi[F];
#pragma unroll
    for (int j = 0; j < F; ++j) { lo[j] = B.lo[j]; hi[j] = B.hi[j]; }
    T stage[StageCfg<NC>::NLD];
    coef_load<T, NC>(stage, src, ldc, nax, nay, mx0, my0, pv != nullptr, tid);
    constexpr int NLP = (NC * NC + 255) / 256;
    T stagep[NLP];
    if (pv) {
#pragma unroll
        for (int k = 0; k < NLP; ++k) {
            const int e = tid + 256 * k;
            const int cx = e / NC, ry = e - cx * NC;
            const int gy = my0 + ry, gx = mx0 + cx;
            T v = 0;
            if (e < NC * NC && gy < nay && gx < nax) v = pv[(size_t)gx * ldp + gy];
            stagep[k] = v;
        }
    }
    __syncthreads();                                   // the previous basis is done with the LDS
    coef_store<T, NC, SC>(C, stage, pv != nullptr, tid);
    if (pv) {
#pragma unroll
        for (int k = 0; k < NLP; ++k) {
            const int e = tid + 256 * k;
            const int cx = e / NC, ry = e - cx * NC;
            if (e < NC * NC) C[ry * SC + cx] = stagep[k];
        }
    }
    __syncthreads();
    for (int e = tid; e < 2 * NC * (TS / 2); e += 256) {
        const int ry = e / (TS / 2), m = e - ry * (TS / 2);
        const T* c = C + ry * SC + m + h - 1;
        T sl0 = 0, sh0 = 0, sl1 = 0, sh1 = 0;
#pragma unroll
        for (int j = 0; j < h; ++j) {
            const T a = c[-j], d = c[NC - j];
            sl0 += lo[2 * j] * a;     sh0 += hi[2 * j] * d;
            sl1 += lo[2 * j + 1] * a; sh1 += hi[2 * j + 1] * d;
        }
        Tm[ry * ST + 2 * m] = sl0 + sh0;
        Tm[ry * ST + 2 * m + 1] = sl1 + sh1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TS * (TS / 2) / 256; ++k) {
        const int e = tid + 256 * k;
        const int ox = e / (TS / 2), m = e - ox * (TS / 2);
        const T* t = Tm + (m + h - 1) * ST + ox;
        T sl0 = 0, sh0 = 0, sl1 = 0, sh1 = 0;
#pragma unroll
        for (int j = 0; j < h; ++j) {
            const T a = t[-j * ST], d = t[(NC - j) * ST];
            sl0 += lo[2 * j] * a;     sh0 += hi[2 * j] * d;
            sl1 += lo[2 * j + 1] * a; sh1 += hi[2 * j + 1] * d;
        }
        acc[k][0] += sl0 + sh0;
        acc[k][1] += sl1 + sh1;
    }
}

template <typename T, int TS, int FMAX>
__global__ void __launch_bounds__(256)
k_idwt_finest_fused(const T* __restrict__ alpha, size_t aband, int ldc, const FinBasis<T>* __restrict__ prm,
                    int nbasis, const T* __restrict__ fin, size_t fin_band,
                    T* __restrict__ xo, size_t xband, int ldo, int nxw, int nyw) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    constexpr int NP = TS * (TS / 2) / 256;
    T acc[NP][2];
#pragma unroll
    for (int k = 0; k < NP; ++k) { acc[k][0] = 0; acc[k][1] = 0; }
    const TileId tl = xcd_tile(!(nbasis & (1 << 29)));
    nbasis &= ~(1 << 29);
    const int ix0 = tl.x * TS, iy0 = tl.y * TS;
    const T* ab = alpha + (size_t)tl.z * aband;
    for (int ib = 0; ib < nbasis; ++ib) {
        const FinBasis<T>& B = prm[ib];
        const T* src = ab + B.coeff_off;
        // the per-thread index arithmetic of every basis body is loop invariant: without this the
        // compiler hoists all of it out of the basis loop (256 VGPRs, occupancy 2; with it 8x less)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (B.K == 0) {
            // 'self': x[gx][gy] += alpha[gy][gx]  (psi.py:229-232) through an LDS transpose
            constexpr int SS = TS + 1;
            __syncthreads();
            for (int e = tid; e < TS * TS; e += 256) {
                const int ly = e / TS, lx = e - ly * TS;                  // lx fastest: rows of alpha
                T v = 0;
                if (iy0 + ly < nyw && ix0 + lx < nxw) v = src[(size_t)(iy0 + ly) * ldc + ix0 + lx];
                smem[ly * SS + lx] = v;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int e = tid + 256 * k;
                const int ox = e / (TS / 2), m = e - ox * (TS / 2);
                acc[k][0] += smem[(2 * m) * SS + ox];
                acc[k][1] += smem[(2 * m + 1) * SS + ox];
            }
            continue;
        }
        const T* pv = B.has_prev ? fin + B.prev_off + (size_t)tl.z * fin_band : nullptr;
        switch (B.F) {
#define X(FF) case FF: if constexpr (FF <= FMAX) idwt_tile_acc<T, FF, TS>(smem, B, src, ldc, pv, ix0, iy0, tid, acc); break;
            X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)
#undef X
            default: break;
        }
    }
    T* dst = xo + (size_t)tl.z * xband;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int ox = e / (TS / 2), m = e - ox * (TS / 2);
        const int gx = ix0 + ox, gy = iy0 + 2 * m;
        if (gx < nxw) {
            T* q = dst + (size_t)gx * ldo + gy;
            if (gy < nyw) q[0] = acc[k][0];
            if (gy + 1 < nyw) q[1] = acc[k][1];
        }
    }
}

// ------------------------------------------------ lean fused finest level (second generation)
// rocprofv3 + the ISA of the kernel above: ~4500 instructions per thread and tile, of which the arithmetic is a
// fifth -- the staging spends ~45 SCALAR instructions per wave-row on quadrant selects, 64-bit row addresses and
// exec-mask juggling (the CU has ONE scalar unit: 117 of the kernel's 185 us), the passes one LDS read and a few index
// instructions per tap.  Same mathematics here, organised around instruction count:
//   * staging: each of the four quadrants is a NC x NC rectangle, a wave takes every 4th row, the lane is the column;
//     loads are UNCONDITIONAL from clamped addresses (scalar row base + per-lane offset) and zeroed by a select,
//     LDS stores use compile-time offsets;
//   * x pass: an item is one row and 2 EV consecutive m: 2 NRX 16-byte LDS reads feed 2 EV h packed FMAs whose
//     operand pairs are (even, odd) output taps; lanes walk down the rows;
//   * y pass: thread (ox, block of MBY m): MBY + h - 1 rows of each half, scalar conflict-free reads, results stay
//     in registers across the bases;
//   * 'self' needs no LDS at all under this thread mapping (a lane is an image row = a column of the y-major plane);
//   * the tile is transposed through LDS once, at the end, and stored in full 16-byte row pieces.
template <typename T, int TS> struct SynFast {
    static constexpr int EV = 16 / (int)sizeof(T);
    static constexpr int MBX = 2 * EV;                      // m per x-pass item
    static constexpr int MBY = TS * TS / 512;               // m per thread in the y pass (256 threads, TS rows)
    static constexpr int nc(int F) { return TS / 2 + F / 2 - 1; }
    static constexpr int nrx(int F) { return (MBX + F / 2 - 1 + EV - 1) / EV; }
    static constexpr int ncp(int F) { return TS / 2 - MBX + nrx(F) * EV; }       // padded half row (>= nc)
    static constexpr int sc(int F) { return 2 * ncp(F); }
    static constexpr int ST = TS + EV;
    static constexpr size_t elems(int F) {
        const size_t a = (size_t)2 * nc(F) * sc(F) + (size_t)2 * nc(F) * ST, b = (size_t)TS * ST;
        return a > b ? a : b;
    }
    static constexpr bool ok = TS % (2 * MBX) == 0 && 256 % TS == 0 && MBY >= 1 && (TS / 2) % MBY == 0;
};

// rows [0, R) x lanes [0, W) of a row-major array: element (r, c) = base[(y0 + r) ld + x0 + c], zero outside
// [0, nyv) x [0, nxv); wave wv takes rows wv, wv + 4, ...
template <typename T, int R, int W>
__device__ __forceinline__ void rect_load(T (&st)[(R + 3) / 4], const T* __restrict__ base, int ld, int x0, int y0,
                                          int nxv, int nyv, int wv, int lane) {
    const int gx = x0 + lane;
    const bool okx = lane < W && gx < nxv;
    const int gxc = gx < nxv ? gx : nxv - 1;
#pragma unroll
    for (int k = 0; k < (R + 3) / 4; ++k) {
        const int gy = y0 + wv + 4 * k;                                // wave-uniform
        const bool oky = wv + 4 * k < R && gy < nyv;
        const T* row = base + (size_t)(gy < nyv ? gy : nyv - 1) * ld;
        const T v = row[gxc];
        st[k] = (okx && oky) ? v : T(0);
    }
}
// st -> LDS: (r, c) at dst[r * rs + c * cs]
template <typename T, int R, int W>
__device__ __forceinline__ void rect_store(const T (&st)[(R + 3) / 4], T* dst, int rs, int cs, int wv, int lane) {
    if (lane < W) {
#pragma unroll
        for (int k = 0; k < (R + 3) / 4; ++k)
            if (wv + 4 * k < R) dst[(wv + 4 * k) * rs + lane * cs] = st[k];
    }
}

// a thread's 2 MBY consecutive pixels of image row ox -> LDS -> full 16-byte row pieces (scalar when the rows of
// dst are not 16-byte aligned or at the image edge)
template <typename T, int TS>
__device__ __forceinline__ void tile_store_transposed(T* smem, const T (&acc)[2 * SynFast<T, TS>::MBY],
                                                      T* __restrict__ dst, int ldo, int nxw, int nyw, int ix0, int iy0) {
    using S = SynFast<T, TS>;
    constexpr int MBY = S::MBY, EV = S::EV, ST = S::ST;
    struct alignas(16) Vec { T e[EV]; };
    const int tid = threadIdx.x;
    const int ox = tid % TS, mb = tid / TS;
    __syncthreads();                               // the last basis is done with the LDS
    T* o = smem + ox * ST + 2 * mb * MBY;
#pragma unroll
    for (int g = 0; g < 2 * MBY / EV; ++g) {
        Vec w;
#pragma unroll
        for (int c = 0; c < EV; ++c) w.e[c] = acc[EV * g + c];
        *reinterpret_cast<Vec*>(o + EV * g) = w;
    }
    __syncthreads();
    const bool vec_ok = (ldo % EV) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int e = tid; e < TS * (TS / EV); e += 256) {
        const int r = e / (TS / EV), cv = e - r * (TS / EV);
        const int gx = ix0 + r, gy = iy0 + EV * cv;
        if (gx >= nxw) continue;
        const Vec w = *reinterpret_cast<const Vec*>(smem + r * ST + EV * cv);
        T* q = dst + (size_t)gx * ldo + gy;
        if (vec_ok && gy + EV <= nyw) *reinterpret_cast<Vec*>(q) = w;
        else {
#pragma unroll
            for (int c = 0; c < EV; ++c) if (gy + c < nyw) q[c] = w.e[c];
        }
    }
}

template <typename T, int F, int TS, typename PB>
__device__ __forceinline__ void idwt_tile_acc2(T* smem, const PB& B, const T* __restrict__ src, int ldc,
                                               const T* __restrict__ pv, int ix0, int iy0, int tid,
                                               T (&acc)[2 * SynFast<T, TS>::MBY]) {
    using S = SynFast<T, TS>;
    typedef T T2 __attribute__((ext_vector_type(2)));
    constexpr int EV = S::EV, h = F / 2, NC = S::nc(F), NCP = S::ncp(F), SC = S::sc(F), ST = S::ST;
    constexpr int MBX = S::MBX, NRX = S::nrx(F), MBY = S::MBY;
    static_assert(NC <= 64 && NCP >= NC, "a quadrant row fits a wavefront");
    struct alignas(16) Vec { T e[EV]; };
    T* C = smem;                                      // [2 NC][SC]: lo-x | hi-x halves at 0 / NCP
    T* Tm = C + 2 * NC * SC;                          // [2 NC][ST] after the x pass
    const int nax = B.nax, nay = B.nay;
    const int mx0 = ix0 / 2, my0 = iy0 / 2;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // 1. stage the four quadrants (all loads first)
    T q00[(NC + 3) / 4], q01[(NC + 3) / 4], q10[(NC + 3) / 4], q11[(NC + 3) / 4];
    if (pv) rect_load<T, NC, NC>(q00, pv, B.ldp, my0, mx0, nay, nax, wv, lane);       // LL = previous image, (x, y) order
    else    rect_load<T, NC, NC>(q00, src, ldc, mx0, my0, nax, nay, wv, lane);
    rect_load<T, NC, NC>(q01, src + nax, ldc, mx0, my0, nax, nay, wv, lane);
    rect_load<T, NC, NC>(q10, src + (size_t)nay * ldc, ldc, mx0, my0, nax, nay, wv, lane);
    rect_load<T, NC, NC>(q11, src + (size_t)nay * ldc + nax, ldc, mx0, my0, nax, nay, wv, lane);
    T2 L2[h], H2[h];
#pragma unroll
    for (int j = 0; j < h; ++j) { L2[j].x = B.lo[2 * j]; L2[j].y = B.lo[2 * j + 1]; H2[j].x = B.hi[2 * j]; H2[j].y = B.hi[2 * j + 1]; }
    __syncthreads();                                   // the previous basis is done with the LDS
    if (pv) rect_store<T, NC, NC>(q00, C, 1, SC, wv, lane);          // transposed: row index = lane (y), column = x
    else    rect_store<T, NC, NC>(q00, C, SC, 1, wv, lane);
    rect_store<T, NC, NC>(q01, C + NCP, SC, 1, wv, lane);
    rect_store<T, NC, NC>(q10, C + NC * SC, SC, 1, wv, lane);
    rect_store<T, NC, NC>(q11, C + NC * SC + NCP, SC, 1, wv, lane);
    __syncthreads();
    // 2. x pass: Tm[ry][2m + p] = sum_j lo[2j+p] C[ry][m+h-1-j] + hi[2j+p] C[ry][NCP+m+h-1-j]
    for (int it = tid; it < 2 * NC * (TS / 2 / MBX); it += 256) {
        const int mb = it / (2 * NC), ry = it - mb * (2 * NC);
        const T* c = C + ry * SC + mb * MBX;
        T a[NRX * EV], d[NRX * EV];
#pragma unroll
        for (int k = 0; k < NRX; ++k) {
            *reinterpret_cast<Vec*>(&a[EV * k]) = *reinterpret_cast<const Vec*>(c + EV * k);
            *reinterpret_cast<Vec*>(&d[EV * k]) = *reinterpret_cast<const Vec*>(c + NCP + EV * k);
        }
        T* t = Tm + ry * ST + 2 * mb * MBX;
#pragma unroll
        for (int g = 0; g < 2 * MBX / EV; ++g) {          // EV outputs = EV / 2 values of m per 16-byte store
            Vec o;
#pragma unroll
            for (int u = 0; u < EV / 2; ++u) {
                const int m = g * (EV / 2) + u;
                T2 s = T2{0, 0};
#pragma unroll
                for (int j = 0; j < h; ++j) { s += L2[j] * a[m + h - 1 - j]; s += H2[j] * d[m + h - 1 - j]; }
                o.e[2 * u] = s.x; o.e[2 * u + 1] = s.y;
            }
            *reinterpret_cast<Vec*>(t + EV * g) = o;
        }
    }
    __syncthreads();
    // 3. y pass: thread (ox, mb): image row ix0 + ox, pixels iy0 + 2 (mb MBY + m) + p
    {
        const int ox = tid % TS, mb = tid / TS;
        const T* t = Tm + (mb * MBY) * ST + ox;
        T a[MBY + h - 1], d[MBY + h - 1];
#pragma unroll
        for (int i = 0; i < MBY + h - 1; ++i) { a[i] = t[i * ST]; d[i] = t[(NC + i) * ST]; }
#pragma unroll
        for (int m = 0; m < MBY; ++m) {
            T2 s = T2{0, 0};
#pragma unroll
            for (int j = 0; j < h; ++j) { s += L2[j] * a[m + h - 1 - j]; s += H2[j] * d[m + h - 1 - j]; }
            acc[2 * m] += s.x; acc[2 * m + 1] += s.y;
        }
    }
}

template <typename T, int TS, int FMAX>
__global__ void __launch_bounds__(256)
k_idwt_finest_fused2(const T* __restrict__ alpha, size_t aband, int ldc, const FinBasis<T>* __restrict__ prm,
                     int nbasis, const T* __restrict__ fin, size_t fin_band,
                     T* __restrict__ xo, size_t xband, int ldo, int nxw, int nyw) {
    using S = SynFast<T, TS>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    constexpr int MBY = S::MBY, EV = S::EV, ST = S::ST;
    struct alignas(16) Vec { T e[EV]; };
    T acc[2 * MBY];
#pragma unroll
    for (int k = 0; k < 2 * MBY; ++k) acc[k] = 0;
    const TileId tl = xcd_tile(!(nbasis & (1 << 29)));
    nbasis &= ~(1 << 29);
    const int ix0 = tl.x * TS, iy0 = tl.y * TS;
    const T* ab = alpha + (size_t)tl.z * aband;
    for (int ib = 0; ib < nbasis; ++ib) {
        const FinBasis<T>& B = prm[ib];
        const T* src = ab + B.coeff_off;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                 // keep the per-basis index arithmetic inside the loop (registers)
        if (B.K == 0) {
            // 'self': x[gx][gy] += alpha[gy][gx]  (psi.py:229-232): lanes along gx read rows of the y-major plane
            const int ox = tid % TS, mb = tid / TS;
            const int gx = ix0 + ox, gy0 = iy0 + 2 * mb * MBY;
            const int gxc = gx < nxw ? gx : nxw - 1;
            T v[2 * MBY];
#pragma unroll
            for (int c = 0; c < 2 * MBY; ++c) {
                const int gy = gy0 + c;
                v[c] = src[(size_t)(gy < nyw ? gy : nyw - 1) * ldc + gxc];
            }
#pragma unroll
            for (int c = 0; c < 2 * MBY; ++c) acc[c] += v[c];         // out-of-range outputs are never stored
            continue;
        }
        const T* pv = B.has_prev ? fin + B.prev_off + (size_t)tl.z * fin_band : nullptr;
        switch (B.F) {
#define X(FF) case FF: if constexpr (FF <= FMAX) idwt_tile_acc2<T, FF, TS>(smem, B, src, ldc, pv, ix0, iy0, tid, acc); break;
            X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)
#undef X
            default: break;
        }
    }
    tile_store_transposed<T, TS>(smem, acc, xo + (size_t)tl.z * xband, ldo, nxw, nyw, ix0, iy0);
}

// one level of one basis with the lean tile (levels >= 1 of psi^H: the partial image of the next finer level)
template <typename T, int TS, int FMAX>
__global__ void __launch_bounds__(256)
k_idwt_batched2(const T* __restrict__ alpha, size_t aband, int ldc, const T* __restrict__ scr_prev, size_t sband,
                T* __restrict__ out_base, size_t oband, const SynPrm<T>* __restrict__ prm, int nwb) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const TileId tl = xcd_tile(!(nwb & (1 << 29)));
    nwb &= ~(1 << 29);
    const int j = tl.z % nwb, band = tl.z / nwb;
    const SynPrm<T>& P = prm[j];
    const int ix0 = tl.x * TS, iy0 = tl.y * TS;
    if (ix0 >= P.nxw || iy0 >= P.nyw) return;
    const T* src = alpha + (size_t)band * aband + P.coeff_off;
    const T* pv = P.has_prev ? scr_prev + P.prev_off + (size_t)band * sband : nullptr;
    T acc[2 * SynFast<T, TS>::MBY];
#pragma unroll
    for (int k = 0; k < 2 * SynFast<T, TS>::MBY; ++k) acc[k] = 0;
    const int tid = threadIdx.x;
    switch (P.F) {
#define X(FF) case FF: if constexpr (FF <= FMAX) idwt_tile_acc2<T, FF, TS>(reinterpret_cast<T*>(smem_raw), P, src, ldc, pv, ix0, iy0, tid, acc); break;
        X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)
#undef X
        default: break;
    }
    tile_store_transposed<T, TS>(reinterpret_cast<T*>(smem_raw), acc, out_base + P.out_off + (size_t)band * oband, P.ldo,
                                 P.nxw, P.nyw, ix0, iy0);
}

template <typename T>
static Filt<T> make_filt(const BasisInfo& b, int lo_idx, int hi_idx) {
    Filt<T> f;
    memset(&f, 0, sizeof(f));
    f.F = b.F;
    for (int k = 0; k < b.F; ++k) { f.lo[k] = (T)b.filt[lo_idx][k]; f.hi[k] = (T)b.filt[hi_idx][k]; }
    return f;
}

template <typename T>
static size_t dwt_lds(int F) {
    constexpr int TA = Tile<T>::TA;
    const int NI = 2 * TA + F - 2;
    const size_t plain = (size_t)NI * (NI + 1) + (size_t)NI * (2 * TA + 1);       // LL aliases A
    const size_t fast = DwtFast<T>::elems(TA, F);                                // dwt_tile_fast (aligned input rows)
    return sizeof(T) * (plain > fast ? plain : fast);
}
template <typename T>
static size_t idwt_lds(int F) {
    constexpr int TS = Tile<T>::TS;
    const int NC = TS / 2 + F / 2 - 1;
    return sizeof(T) * ((size_t)2 * NC * (2 * NC + 1) + (size_t)2 * NC * (TS + 1));
}

#define PFB_FOR_F(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)

template <typename T>
static void launch_dwt(int F, dim3 grid, size_t lds, hipStream_t st, const T* in, size_t in_band, int ldin,
                       int nxin, int nyin, T* blk, size_t c_band, int ldc, int Cx, int Cy, T* approx,
                       size_t a_band, const Filt<T>& f_in) {
    static const bool fast = [] { const char* e = getenv("PFB_DWT_FAST"); return !e || atoi(e); }();
    Filt<T> f = f_in;
    if (!fast) f.F = -f.F;
    switch (F) {
#define X(FF) case FF: hipLaunchKernelGGL((k_dwt_level<T, FF, Tile<T>::TA>), grid, dim3(256), lds, st, in, in_band, \
                                          ldin, nxin, nyin, blk, c_band, ldc, Cx, Cy, approx, a_band, f); break;
        PFB_FOR_F(X)
#undef X
        default: break;
    }
}
template <typename T, bool ACC>
static void launch_idwt(int F, dim3 grid, size_t lds, hipStream_t st, const T* blk, size_t c_band, int ldc,
                        int nax, int nay, const T* prev, size_t p_band, int ldp, T* out, size_t o_band, int ldo,
                        int nxw, int nyw, const Filt<T>& f) {
    switch (F) {
#define X(FF) case FF: hipLaunchKernelGGL((k_idwt_level<T, FF, Tile<T>::TS, ACC>), grid, dim3(256), lds, st, blk, \
                                          c_band, ldc, nax, nay, prev, p_band, ldp, out, o_band, ldo, nxw, nyw, f); break;
        PFB_FOR_F(X)
#undef X
        default: break;
    }
}
template <typename T>
static int set_wavelet_lds_limits() {
    const int lds_max = 160 * 1024;
#define X(FF) \
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_dwt_level<T, FF, Tile<T>::TA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max)); \
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_idwt_level<T, FF, Tile<T>::TS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max)); \
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)k_idwt_level<T, FF, Tile<T>::TS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    PFB_FOR_F(X)
#undef X
    return PFB_OK;
}

template <typename T> static int psi_dot_batched_t(pfb_psi_plan* p, const T* x, T* alpha, hipStream_t st);

template <typename T>
static int psi_dot_t(pfb_psi_plan* p, const T* x, T* alpha, hipStream_t st) {
    {
        const char* e = getenv("PFB_PSI_FUSED");
        if (!(e && !atoi(e)) && p->nbasis > 0) return psi_dot_batched_t<T>(p, x, alpha, st);
    }
    const size_t plane = (size_t)p->Nymax * p->Nxmax;
    const size_t aband = plane * p->nbasis;
    const size_t xband = (size_t)p->nx * p->ny;
    for (int ib = 0; ib < p->nbasis; ++ib) {
        const BasisInfo& b = p->bases[ib];
        T* ab = alpha + (size_t)ib * plane;
        if (b.K == 0) {        // 'self': alpha[b, ib, 0:ny, 0:nx] = x[b].T     (psi.py:196-199)
            dim3 grid((p->ny + 31) / 32, (p->nx + 31) / 32, p->nband);
            hipLaunchKernelGGL((k_transpose<T, false>), grid, dim3(256), 0, st, x, xband, p->ny, ab, aband,
                               p->Nxmax, p->nx, p->ny);
            continue;
        }
        const Filt<T> f = make_filt<T>(b, 0, 1);
        const T* in = x;
        size_t in_band = xband;
        int ldin = p->ny;
        for (int l = 0; l < p->nlevel; ++l) {
            const LevelInfo& L = b.lev[l];
            T* blk = ab + (size_t)L.lowy * p->Nxmax + L.lowx;
            const bool last = l == p->nlevel - 1;
            T* approx = last ? nullptr : (T*)p->scratch[l & 1];
            constexpr int TA = Tile<T>::TA;
            dim3 grid((L.Cx + TA - 1) / TA, (L.Cy + TA - 1) / TA, p->nband);
            launch_dwt<T>(b.F, grid, dwt_lds<T>(b.F), st, in, in_band, ldin, L.nxin, L.nyin, blk, aband,
                          p->Nxmax, L.Cx, L.Cy, approx, p->scratch_band, f);
            in = approx;
            in_band = p->scratch_band;
            ldin = L.Cy;
        }
    }
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

template <typename T>
static int psi_fin_prepare(pfb_psi_plan* p) {
    // parameter table + per-basis level-1 partial images of the fused finest level (once per plan)
    if (p->fin_prm) return PFB_OK;
    const size_t plane = (size_t)p->Nymax * p->Nxmax;
    std::vector<FinBasis<T>> h(p->nbasis);
    size_t fin_band = 1;
    int fmax = 2;
    for (int ib = 0; ib < p->nbasis; ++ib) {
        const BasisInfo& b = p->bases[ib];
        if (b.K != 0 && p->nlevel > 1) {
            const size_t e = (size_t)b.lev[1].nxo * b.lev[1].nyo;
            if (e > fin_band) fin_band = e;
        }
        if (b.K != 0 && b.F > fmax) fmax = b.F;
    }
    p->fin_band = fin_band;
    p->fin_basis = fin_band * p->nband;
    p->fin_fmax = fmax;
    for (int ib = 0; ib < p->nbasis; ++ib) {
        const BasisInfo& b = p->bases[ib];
        FinBasis<T>& q = h[ib];
        memset(&q, 0, sizeof(q));
        q.K = b.K;
        q.F = b.F;
        if (b.K == 0) { q.coeff_off = (long long)((size_t)ib * plane); continue; }
        const LevelInfo& L = b.lev[0];
        q.coeff_off = (long long)((size_t)ib * plane + (size_t)L.lowy * p->Nxmax + L.lowx);
        q.nax = L.Cx; q.nay = L.Cy;
        q.has_prev = p->nlevel > 1;
        q.ldp = p->nlevel > 1 ? b.lev[1].nyo : 0;
        q.prev_off = (long long)((size_t)ib * p->fin_basis);
        for (int k = 0; k < b.F; ++k) { q.lo[k] = (T)b.filt[2][k]; q.hi[k] = (T)b.filt[3][k]; }
    }
    PFB_HIP_CHECK(hipMalloc(&p->fin_prm, sizeof(FinBasis<T>) * h.size()));
    PFB_HIP_CHECK(hipMemcpy(p->fin_prm, h.data(), sizeof(FinBasis<T>) * h.size(), hipMemcpyHostToDevice));
    PFB_HIP_CHECK(hipMalloc(&p->fin_scratch, sizeof(T) * p->fin_basis * p->nbasis));
    constexpr int TS = Tile<T>::TS;
    constexpr int TA = Tile<T>::TA;
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_idwt_finest_fused<T, TS, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_idwt_finest_fused<T, TS, 18>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_idwt_batched2<T, TS, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_idwt_batched2<T, TS, 18>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_idwt_finest_fused2<T, TS, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_idwt_finest_fused2<T, TS, 18>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_dwt_l1_fused<T, TA, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_dwt_l1_fused<T, TA, 18>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_dwt_batched<T, TA, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_dwt_batched<T, TA, 18>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_idwt_batched<T, TS, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PFB_HIP_CHECK(hipFuncSetAttribute((const void*)(k_idwt_batched<T, TS, 18>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // ---- batched level tables: [level][wavelet basis]
    std::vector<int> wb;
    for (int ib = 0; ib < p->nbasis; ++ib) if (p->bases[ib].K != 0) wb.push_back(ib);
    p->nwb = (int)wb.size();
    if (p->nwb == 0) return PFB_OK;
    p->bscr_basis = p->scratch_band * p->nband;
    for (int k = 0; k < 2; ++k)
        PFB_HIP_CHECK(hipMalloc(&p->bscr[k], sizeof(T) * p->bscr_basis * p->nwb));
    std::vector<AnaPrm<T>> ha((size_t)p->nlevel * p->nwb);
    std::vector<SynPrm<T>> hs((size_t)p->nlevel * p->nwb);
    for (int l = 0; l < p->nlevel; ++l) {
        p->gx_ana[l] = p->gy_ana[l] = p->gx_syn[l] = p->gy_syn[l] = 1;
        for (int j = 0; j < p->nwb; ++j) {
            const int ib = wb[j];
            const BasisInfo& b = p->bases[ib];
            const LevelInfo& L = b.lev[l];
            AnaPrm<T>& A = ha[(size_t)l * p->nwb + j];
            memset(&A, 0, sizeof(A));
            A.coeff_off = (long long)((size_t)ib * plane + (size_t)L.lowy * p->Nxmax + L.lowx);
            A.in_off = l == 0 ? 0 : (long long)((size_t)j * p->bscr_basis);      // level 0 reads x
            A.approx_off = (long long)((size_t)j * p->bscr_basis);
            A.F = b.F; A.nxin = L.nxin; A.nyin = L.nyin;
            A.ldin = l == 0 ? p->ny : b.lev[l - 1].Cy;
            A.Cx = L.Cx; A.Cy = L.Cy;
            A.has_approx = l < p->nlevel - 1;
            SynPrm<T>& S = hs[(size_t)l * p->nwb + j];
            memset(&S, 0, sizeof(S));
            S.coeff_off = A.coeff_off;
            S.F = b.F; S.nax = L.Cx; S.nay = L.Cy;
            S.has_prev = l < p->nlevel - 1;
            S.ldp = l < p->nlevel - 1 ? b.lev[l + 1].nyo : 0;
            S.prev_off = (long long)((size_t)j * p->bscr_basis);
            S.out_off = l == 1 ? (long long)((size_t)ib * p->fin_basis) : (long long)((size_t)j * p->bscr_basis);
            S.nxw = L.nxo; S.nyw = L.nyo; S.ldo = L.nyo;
            for (int k = 0; k < b.F; ++k) {
                A.lo[k] = (T)b.filt[0][k]; A.hi[k] = (T)b.filt[1][k];
                S.lo[k] = (T)b.filt[2][k]; S.hi[k] = (T)b.filt[3][k];
            }
            const int ga = (L.Cx + TA - 1) / TA, gb = (L.Cy + TA - 1) / TA;
            if (ga > p->gx_ana[l]) p->gx_ana[l] = ga;
            if (gb > p->gy_ana[l]) p->gy_ana[l] = gb;
            const int gc = (L.nxo + TS - 1) / TS, gd = (L.nyo + TS - 1) / TS;
            if (gc > p->gx_syn[l]) p->gx_syn[l] = gc;
            if (gd > p->gy_syn[l]) p->gy_syn[l] = gd;
        }
    }
    PFB_HIP_CHECK(hipMalloc(&p->ana_prm, sizeof(AnaPrm<T>) * ha.size()));
    PFB_HIP_CHECK(hipMemcpy(p->ana_prm, ha.data(), sizeof(AnaPrm<T>) * ha.size(), hipMemcpyHostToDevice));
    PFB_HIP_CHECK(hipMalloc(&p->syn_prm, sizeof(SynPrm<T>) * hs.size()));
    PFB_HIP_CHECK(hipMemcpy(p->syn_prm, hs.data(), sizeof(SynPrm<T>) * hs.size(), hipMemcpyHostToDevice));
    return PFB_OK;
}

// psi.dot with ONE launch per level for all wavelet bases (+ the 'self' transposes)
template <typename T>
static int psi_dot_batched_t(pfb_psi_plan* p, const T* x, T* alpha, hipStream_t st) {
    int rc = psi_fin_prepare<T>(p);
    if (rc != PFB_OK) return rc;
    const size_t plane = (size_t)p->Nymax * p->Nxmax;
    const size_t aband = plane * p->nbasis;
    const size_t xband = (size_t)p->nx * p->ny;
    constexpr int TA = Tile<T>::TA;
    // finest level of all bases (and the first 'self' plane) in one kernel when the image rows are 16-byte aligned
    constexpr int VW = 16 / (int)sizeof(T);
    static const bool l1f_on = [] { const char* e = getenv("PFB_DWT_L1FUSED"); const char* f = getenv("PFB_DWT_FAST");
                                    return (!e || atoi(e)) && (!f || atoi(f)); }();     // built from the fast tile's passes
    const bool l1_fused = l1f_on && p->nwb > 0 && p->ny % VW == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                          TA % (2 * VW) == 0;
    long long self_off = -1;
    for (int ib = 0; ib < p->nbasis; ++ib) {
        if (p->bases[ib].K != 0) continue;
        if (l1_fused && self_off < 0) { self_off = (long long)((size_t)ib * plane); continue; }
        dim3 grid((p->ny + 31) / 32, (p->nx + 31) / 32, p->nband);
        hipLaunchKernelGGL((k_transpose<T, false>), grid, dim3(256), 0, st, x, xband, p->ny,
                           alpha + (size_t)ib * plane, aband, p->Nxmax, p->nx, p->ny);
    }
    if (p->nwb > 0) {
        const size_t lds = dwt_lds<T>(p->fin_fmax);
        static const int xoffa = [] { const char* e = getenv("PFB_PSI_XCD"); return (!e || atoi(e)) ? 0 : (1 << 29); }();
        if (l1_fused) {
            dim3 grid(p->gx_ana[0], p->gy_ana[0], p->nband);
            const AnaPrm<T>* prm = (const AnaPrm<T>*)p->ana_prm;
            if (p->fin_fmax <= 8)
                hipLaunchKernelGGL((k_dwt_l1_fused<T, TA, 8>), grid, dim3(256), sizeof(T) * DwtFast<T>::elems(TA, 8), st, x, xband,
                                   p->ny, p->nx, p->ny, alpha, aband, p->Nxmax, (T*)p->bscr[0], p->scratch_band, prm,
                                   p->nwb | xoffa, self_off);
            else
                hipLaunchKernelGGL((k_dwt_l1_fused<T, TA, 18>), grid, dim3(256), sizeof(T) * DwtFast<T>::elems(TA, 18), st, x, xband,
                                   p->ny, p->nx, p->ny, alpha, aband, p->Nxmax, (T*)p->bscr[0], p->scratch_band, prm,
                                   p->nwb | xoffa, self_off);
        }
        for (int l = l1_fused ? 1 : 0; l < p->nlevel; ++l) {
            dim3 grid(p->gx_ana[l], p->gy_ana[l], p->nband * p->nwb);
            const T* in = l == 0 ? x : (const T*)p->bscr[(l - 1) & 1];
            const size_t in_band = l == 0 ? xband : p->scratch_band;
            const AnaPrm<T>* prm = (const AnaPrm<T>*)p->ana_prm + (size_t)l * p->nwb;
            static const bool fast = [] { const char* e = getenv("PFB_DWT_FAST"); return !e || atoi(e); }();
            static const bool xcd = [] { const char* e = getenv("PFB_PSI_XCD"); return !e || atoi(e); }();
            const int nwb_arg = p->nwb | (fast ? 0 : (1 << 30)) | (xcd ? 0 : (1 << 29));
            if (p->fin_fmax <= 8)
                hipLaunchKernelGGL((k_dwt_batched<T, TA, 8>), grid, dim3(256), lds, st, in, in_band, alpha, aband,
                                   p->Nxmax, (T*)p->bscr[l & 1], p->scratch_band, prm, nwb_arg);
            else
                hipLaunchKernelGGL((k_dwt_batched<T, TA, 18>), grid, dim3(256), lds, st, in, in_band, alpha, aband,
                                   p->Nxmax, (T*)p->bscr[l & 1], p->scratch_band, prm, nwb_arg);
        }
    }
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

// all bases' finest level in one kernel (image written once); coarser levels basis by basis
template <typename T>
static int psi_hdot_fused_t(pfb_psi_plan* p, const T* alpha, T* xo, hipStream_t st) {
    int rc = psi_fin_prepare<T>(p);
    if (rc != PFB_OK) return rc;
    const size_t plane = (size_t)p->Nymax * p->Nxmax;
    const size_t aband = plane * p->nbasis;
    const size_t xband = (size_t)p->nx * p->ny;
    constexpr int TS = Tile<T>::TS;
    static const int xoff = [] { const char* e = getenv("PFB_PSI_XCD"); return (!e || atoi(e)) ? 0 : (1 << 29); }();
    if (p->nwb > 0) {
        const size_t ldsb = idwt_lds<T>(p->fin_fmax);
        for (int l = p->nlevel - 1; l >= 1; --l) {       // coarse levels: one launch per level, all bases
            dim3 g(p->gx_syn[l], p->gy_syn[l], p->nband * p->nwb);
            const SynPrm<T>* prm = (const SynPrm<T>*)p->syn_prm + (size_t)l * p->nwb;
            T* out = l == 1 ? (T*)p->fin_scratch : (T*)p->bscr[l & 1];
            const size_t oband = l == 1 ? p->fin_band : p->scratch_band;
            static const bool lean = [] { const char* e = getenv("PFB_PSI_FIN2"); return !e || atoi(e); }();
            if (lean && SynFast<T, TS>::ok) {
                const size_t lds2 = sizeof(T) * SynFast<T, TS>::elems(p->fin_fmax);
                if (p->fin_fmax <= 8)
                    hipLaunchKernelGGL((k_idwt_batched2<T, TS, 8>), g, dim3(256), lds2, st, alpha, aband, p->Nxmax,
                                       (const T*)p->bscr[(l + 1) & 1], p->scratch_band, out, oband, prm, p->nwb | xoff);
                else
                    hipLaunchKernelGGL((k_idwt_batched2<T, TS, 18>), g, dim3(256), lds2, st, alpha, aband, p->Nxmax,
                                       (const T*)p->bscr[(l + 1) & 1], p->scratch_band, out, oband, prm, p->nwb | xoff);
            } else if (p->fin_fmax <= 8)
                hipLaunchKernelGGL((k_idwt_batched<T, TS, 8>), g, dim3(256), ldsb, st, alpha, aband, p->Nxmax,
                                   (const T*)p->bscr[(l + 1) & 1], p->scratch_band, out, oband, prm, p->nwb | xoff);
            else
                hipLaunchKernelGGL((k_idwt_batched<T, TS, 18>), g, dim3(256), ldsb, st, alpha, aband, p->Nxmax,
                                   (const T*)p->bscr[(l + 1) & 1], p->scratch_band, out, oband, prm, p->nwb | xoff);
        }
    }
    dim3 grid((p->nx + TS - 1) / TS, (p->ny + TS - 1) / TS, p->nband);
    size_t lds = idwt_lds<T>(p->fin_fmax);
    const size_t lds_self = sizeof(T) * (size_t)TS * (TS + 1);
    if (lds < lds_self) lds = lds_self;
    static const bool fin2 = [] { const char* e = getenv("PFB_PSI_FIN2"); return !e || atoi(e); }();
    if (fin2 && SynFast<T, TS>::ok) {
        static_assert(SynFast<T, TS>::ok, "tile shape of the lean fused kernel");
        const size_t lds2 = sizeof(T) * SynFast<T, TS>::elems(p->fin_fmax);
        if (p->fin_fmax <= 8)
            hipLaunchKernelGGL((k_idwt_finest_fused2<T, TS, 8>), grid, dim3(256), lds2, st, alpha, aband, p->Nxmax,
                               (const FinBasis<T>*)p->fin_prm, p->nbasis | xoff, (const T*)p->fin_scratch, p->fin_band,
                               xo, xband, p->ny, p->nx, p->ny);
        else
            hipLaunchKernelGGL((k_idwt_finest_fused2<T, TS, 18>), grid, dim3(256), lds2, st, alpha, aband, p->Nxmax,
                               (const FinBasis<T>*)p->fin_prm, p->nbasis | xoff, (const T*)p->fin_scratch, p->fin_band,
                               xo, xband, p->ny, p->nx, p->ny);
    } else if (p->fin_fmax <= 8)
        hipLaunchKernelGGL((k_idwt_finest_fused<T, TS, 8>), grid, dim3(256), lds, st, alpha, aband, p->Nxmax,
                           (const FinBasis<T>*)p->fin_prm, p->nbasis | xoff, (const T*)p->fin_scratch, p->fin_band,
                           xo, xband, p->ny, p->nx, p->ny);
    else
        hipLaunchKernelGGL((k_idwt_finest_fused<T, TS, 18>), grid, dim3(256), lds, st, alpha, aband, p->Nxmax,
                           (const FinBasis<T>*)p->fin_prm, p->nbasis | xoff, (const T*)p->fin_scratch, p->fin_band,
                           xo, xband, p->ny, p->nx, p->ny);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

template <typename T>
static int psi_hdot_t(pfb_psi_plan* p, const T* alpha, T* xo, hipStream_t st) {
    {
        const char* e = getenv("PFB_PSI_FUSED");
        if (!(e && !atoi(e)) && p->nbasis > 0) return psi_hdot_fused_t<T>(p, alpha, xo, st);
    }
    const size_t plane = (size_t)p->Nymax * p->Nxmax;
    const size_t aband = plane * p->nbasis;
    const size_t xband = (size_t)p->nx * p->ny;
    bool first = true;          // first basis stores, the others accumulate (xo zeroed implicitly)
    for (int ib = 0; ib < p->nbasis; ++ib) {
        const BasisInfo& b = p->bases[ib];
        const T* ab = alpha + (size_t)ib * plane;
        if (b.K == 0) {        // xo[b] (+)= alpha[b, ib, 0:ny, 0:nx].T          (psi.py:229-232)
            dim3 grid((p->nx + 31) / 32, (p->ny + 31) / 32, p->nband);
            if (first)
                hipLaunchKernelGGL((k_transpose<T, false>), grid, dim3(256), 0, st, ab, aband, p->Nxmax, xo,
                                   xband, p->ny, p->ny, p->nx);
            else
                hipLaunchKernelGGL((k_transpose<T, true>), grid, dim3(256), 0, st, ab, aband, p->Nxmax, xo,
                                   xband, p->ny, p->ny, p->nx);
            first = false;
            continue;
        }
        const Filt<T> f = make_filt<T>(b, 2, 3);
        const T* prev = nullptr;
        int ldp = 0;
        for (int l = p->nlevel - 1; l >= 0; --l) {
            const LevelInfo& L = b.lev[l];
            const T* blk = ab + (size_t)L.lowy * p->Nxmax + L.lowx;
            const bool finest = l == 0;
            T* out = finest ? xo : (T*)p->scratch[l & 1];
            const size_t o_band = finest ? xband : p->scratch_band;
            const int nxw = finest ? (L.nxo < p->nx ? L.nxo : p->nx) : L.nxo;
            const int nyw = finest ? (L.nyo < p->ny ? L.nyo : p->ny) : L.nyo;
            const int ldo = finest ? p->ny : L.nyo;
            constexpr int TS = Tile<T>::TS;
            dim3 grid((nxw + TS - 1) / TS, (nyw + TS - 1) / TS, p->nband);
            if (finest && !first)
                launch_idwt<T, true>(b.F, grid, idwt_lds<T>(b.F), st, blk, aband, p->Nxmax, L.Cx, L.Cy, prev,
                                     p->scratch_band, ldp, out, o_band, ldo, nxw, nyw, f);
            else
                launch_idwt<T, false>(b.F, grid, idwt_lds<T>(b.F), st, blk, aband, p->Nxmax, L.Cx, L.Cy, prev,
                                      p->scratch_band, ldp, out, o_band, ldo, nxw, nyw, f);
            prev = out;
            ldp = ldo;
        }
        // signal_size(Cx) can fall short of nx only if nx were odd and ... it cannot: the
        // finest level always covers [0, nx) x [0, ny) (2 Cx - F + 2 >= nx)
        first = false;
    }
    if (first) {               // no bases at all: xo = 0
        hipLaunchKernelGGL((k_fill<T>), dim3(1024), dim3(256), 0, st, xo, xband * p->nband, (T)0);
    }
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

// grid for the streaming elementwise kernels that WRITE as much as they read: a few workgroups per
// CU (tools/micro/hbm_stream.hip: the write-heavy mixes lose 10-25 % when the chip is oversubscribed)
static inline int stream_grid(size_t nvec, int per_cu) {
    static const int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    size_t g = (nvec + 255) / 256;
    if (g > (size_t)ncu * per_cu) g = (size_t)ncu * per_cu;
    if (g < 1) g = 1;
    return (int)g;
}

static inline int ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

template <typename T>
static void dual_update_launch(const T* vp, T* v, const T* w, T lam, T sigma, int nband, size_t nper, T* vp_out,
                               hipStream_t st) {
    constexpr int V = 16 / sizeof(T);
    static const int per_cu = [] { const char* e = getenv("PFB_DUAL_PER_CU"); const int o = e ? atoi(e) : 0;
                                   return o > 0 && o <= 16 ? o : 1; }();
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec = nband <= 8 && nper % V == 0 && al(vp) && al(v) && al(w) && (!vp_out || al(vp_out));
    if (!vec) {
        hipLaunchKernelGGL((k_dual_update<T>), dim3(ew_grid(nper)), dim3(256), 0, st, vp, v, w, lam, sigma, nband,
                           nper, vp_out);
        return;
    }
    const dim3 grid(stream_grid(nper / V, per_cu));
    // non-temporal reads of vp / psi^H(x) once the cube is far beyond the caches (+0.8 % on the config #4 iteration)
    static const int ntl_env = [] { const char* e = getenv("PFB_DUAL_NT"); return e ? atoi(e) : -1; }();
    const bool ntl = ntl_env >= 0 ? ntl_env != 0 : (size_t)nband * nper * sizeof(T) >= ((size_t)64 << 20);
#define PFB_DU_CASE(NB) case NB: if (ntl) hipLaunchKernelGGL((k_dual_update_vec<T, NB, true>), grid, dim3(256), 0, st, vp, v, w, \
                                                    lam, sigma, nper, vp_out); \
                                 else hipLaunchKernelGGL((k_dual_update_vec<T, NB, false>), grid, dim3(256), 0, st, vp, v, w, \
                                                    lam, sigma, nper, vp_out); break;
    switch (nband) { PFB_DU_CASE(1) PFB_DU_CASE(2) PFB_DU_CASE(3) PFB_DU_CASE(4) PFB_DU_CASE(5) PFB_DU_CASE(6)
                     PFB_DU_CASE(7) PFB_DU_CASE(8) }
#undef PFB_DU_CASE
}

}  // namespace pfb

using namespace pfb;

extern "C" {

int pfb_psi_plan_create(int nband, int nx, int ny, int nbasis, const int* basis_k,
                        const double* filters, int nlevel, int dtype, pfb_psi_plan** plan) {
    PFB_REQUIRE(plan != nullptr, PFB_ERR_INVALID, "psi_plan_create: null plan pointer");
    *plan = nullptr;
    PFB_REQUIRE(dtype == PFB_F32 || dtype == PFB_F64, PFB_ERR_INVALID, "psi_plan_create: bad dtype");
    PFB_REQUIRE(nband > 0 && nx > 0 && ny > 0 && nbasis > 0 && basis_k && filters, PFB_ERR_INVALID,
                "psi_plan_create: bad argument");
    PFB_REQUIRE(nlevel >= 1 && nlevel <= MAXLEV, PFB_ERR_UNSUPPORTED, "psi_plan_create: nlevel %d", nlevel);
    pfb_psi_plan* p = (pfb_psi_plan*)calloc(1, sizeof(pfb_psi_plan));
    PFB_REQUIRE(p != nullptr, PFB_ERR_ALLOC, "psi_plan_create: host alloc failed");
    p->bases = (BasisInfo*)calloc(nbasis, sizeof(BasisInfo));
    if (!p->bases) { free(p); set_error("psi_plan_create: host alloc failed"); return PFB_ERR_ALLOC; }
    p->nband = nband; p->nx = nx; p->ny = ny; p->nbasis = nbasis; p->nlevel = nlevel; p->dtype = dtype;
    size_t scratch_elems = 1;
    for (int ib = 0; ib < nbasis; ++ib) {
        BasisInfo& b = p->bases[ib];
        b.K = basis_k[ib];
        if (b.K == 0) continue;
        if (b.K < 1 || b.K > 9) {
            free(p->bases); free(p);
            set_error("psi_plan_create: basis db%d unsupported (db1..db9)", basis_k[ib]);
            return PFB_ERR_UNSUPPORTED;
        }
        b.F = 2 * b.K;
        for (int q = 0; q < 4; ++q)
            for (int k = 0; k < MAXF; ++k) b.filt[q][k] = filters[((size_t)ib * 4 + q) * MAXF + k];
        // bookkeeping of pfb/operators/psi.py:49-94
        int Nx = nx, Ny = ny, totx = 0, toty = 0;
        int ainx = nx, ainy = ny;                 // actual analysis input shape per level
        for (int l = 0; l < nlevel; ++l) {
            LevelInfo& L = b.lev[l];
            L.Cx = coeff_size(Nx, b.F);
            L.Cy = coeff_size(Ny, b.F);
            L.nxin = ainx; L.nyin = ainy;
            totx += L.Cx; toty += L.Cy;
            Nx = L.Cx + L.Cx % 2;
            Ny = L.Cy + L.Cy % 2;
            L.nxo = signal_size(L.Cx, b.F);
            L.nyo = signal_size(L.Cy, b.F);
            ainx = L.Cx; ainy = L.Cy;             // the approx passed on is (Cx, Cy), wavelets.py:171
            // scratch holds the approx handed to the next analysis level (Cx*Cy, not for the
            // last level) and the partial images of the synthesis (nxo*nyo, not for level 0)
            if (l < nlevel - 1 && (size_t)L.Cx * L.Cy > scratch_elems) scratch_elems = (size_t)L.Cx * L.Cy;
            if (l > 0 && (size_t)L.nxo * L.nyo > scratch_elems) scratch_elems = (size_t)L.nxo * L.nyo;
        }
        b.Ntotx = totx + b.lev[nlevel - 1].Cx;
        b.Ntoty = toty + b.lev[nlevel - 1].Cy;
        int highx = 2 * b.lev[nlevel - 1].Cx, highy = 2 * b.lev[nlevel - 1].Cy;
        for (int l = nlevel - 1; l >= 0; --l) {
            LevelInfo& L = b.lev[l];
            if (l < nlevel - 1) { highx += L.Cx; highy += L.Cy; }
            L.lowx = highx - 2 * L.Cx;
            L.lowy = highy - 2 * L.Cy;
        }
        if (b.Ntotx > p->Nxmax) p->Nxmax = b.Ntotx;
        if (b.Ntoty > p->Nymax) p->Nymax = b.Ntoty;
    }
    // 'self' only dictionaries: the reference leaves Nxmax = Nymax = 0 (psi.py:32-33,77-78)
    // and cannot hold the image; we size the plane to the image so the operator stays usable
    bool any_db = false;
    for (int ib = 0; ib < nbasis; ++ib) any_db = any_db || p->bases[ib].K != 0;
    if (!any_db) { p->Nxmax = nx; p->Nymax = ny; }
    if (p->Nxmax < nx) p->Nxmax = nx;
    if (p->Nymax < ny) p->Nymax = ny;
    p->scratch_band = scratch_elems;
    const size_t esz = dtype == PFB_F32 ? 4 : 8;
    for (int k = 0; k < 2; ++k) {
        if (hipMalloc(&p->scratch[k], esz * scratch_elems * nband) != hipSuccess) {
            pfb_psi_plan_destroy(p);
            set_error("psi_plan_create: device allocation failed");
            return PFB_ERR_ALLOC;
        }
    }
    {
        int rc = dtype == PFB_F32 ? set_wavelet_lds_limits<float>() : set_wavelet_lds_limits<double>();
        if (rc != PFB_OK) { pfb_psi_plan_destroy(p); return rc; }
    }
    *plan = p;
    return PFB_OK;
}

int pfb_psi_plan_destroy(pfb_psi_plan* p) {
    if (!p) return PFB_OK;
    for (int k = 0; k < 2; ++k) if (p->scratch[k]) (void)hipFree(p->scratch[k]);
    if (p->fin_prm) (void)hipFree(p->fin_prm);
    if (p->fin_scratch) (void)hipFree(p->fin_scratch);
    if (p->ana_prm) (void)hipFree(p->ana_prm);
    if (p->syn_prm) (void)hipFree(p->syn_prm);
    for (int k = 0; k < 2; ++k) if (p->bscr[k]) (void)hipFree(p->bscr[k]);
    free(p->bases);
    free(p);
    return PFB_OK;
}

int pfb_psi_plan_dims(const pfb_psi_plan* p, int* nymax, int* nxmax) {
    PFB_REQUIRE(p != nullptr, PFB_ERR_INVALID, "psi_plan_dims: null plan");
    if (nymax) *nymax = p->Nymax;
    if (nxmax) *nxmax = p->Nxmax;
    return PFB_OK;
}

int pfb_psi_dot(pfb_psi_plan* p, const void* x, void* alpha, void* stream) {
    PFB_REQUIRE(p && x && alpha, PFB_ERR_INVALID, "psi_dot: null argument");
    return p->dtype == PFB_F32 ? psi_dot_t<float>(p, (const float*)x, (float*)alpha, as_stream(stream))
                               : psi_dot_t<double>(p, (const double*)x, (double*)alpha, as_stream(stream));
}

int pfb_psi_hdot(pfb_psi_plan* p, const void* alpha, void* xo, void* stream) {
    PFB_REQUIRE(p && alpha && xo, PFB_ERR_INVALID, "psi_hdot: null argument");
    return p->dtype == PFB_F32 ? psi_hdot_t<float>(p, (const float*)alpha, (float*)xo, as_stream(stream))
                               : psi_hdot_t<double>(p, (const double*)alpha, (double*)xo, as_stream(stream));
}

int pfb_dual_update(int dtype, const void* vp, void* v, const void* weight, double lam, double sigma,
                    int nband, size_t nper, void* vp_out, void* stream) {
    PFB_REQUIRE(vp && v && weight && nband > 0, PFB_ERR_INVALID, "dual_update: bad argument");
    hipStream_t st = as_stream(stream);
    if (dtype == PFB_F32) dual_update_launch<float>((const float*)vp, (float*)v, (const float*)weight, (float)lam,
                                                    (float)sigma, nband, nper, (float*)vp_out, st);
    else dual_update_launch<double>((const double*)vp, (double*)v, (const double*)weight, lam, sigma, nband, nper,
                                    (double*)vp_out, st);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

int pfb_dual_bandsum_chunk(int dtype, const void* vp, const void* v, double sigma, int nband, size_t count,
                           size_t band_stride, void* sum_out, void* stream) {
    PFB_REQUIRE(vp && v && sum_out && nband > 0 && band_stride >= count, PFB_ERR_INVALID, "dual_bandsum: bad argument");
    if (count == 0) return PFB_OK;
    hipStream_t st = as_stream(stream);
    if (dtype == PFB_F32)
        hipLaunchKernelGGL((k_dual_bandsum<float>), dim3(ew_grid(count)), dim3(256), 0, st, (const float*)vp,
                           (const float*)v, (float)sigma, nband, count, band_stride, (float*)sum_out);
    else
        hipLaunchKernelGGL((k_dual_bandsum<double>), dim3(ew_grid(count)), dim3(256), 0, st, (const double*)vp,
                           (const double*)v, sigma, nband, count, band_stride, (double*)sum_out);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}
int pfb_dual_bandsum(int dtype, const void* vp, const void* v, double sigma, int nband, size_t nper,
                     void* sum_out, void* stream) {
    return pfb_dual_bandsum_chunk(dtype, vp, v, sigma, nband, nper, nper, sum_out, stream);
}

int pfb_dual_apply_chunk(int dtype, const void* vp, void* v, const void* weight, const void* sum_in, double lam,
                         double sigma, int nband, size_t count, size_t band_stride, void* vp_out, void* stream) {
    PFB_REQUIRE(vp && v && weight && sum_in && nband > 0 && band_stride >= count, PFB_ERR_INVALID, "dual_apply: bad argument");
    if (count == 0) return PFB_OK;
    hipStream_t st = as_stream(stream);
    if (dtype == PFB_F32)
        hipLaunchKernelGGL((k_dual_apply<float>), dim3(ew_grid(count)), dim3(256), 0, st, (const float*)vp,
                           (float*)v, (const float*)weight, (const float*)sum_in, (float)lam, (float)sigma,
                           nband, count, band_stride, (float*)vp_out);
    else
        hipLaunchKernelGGL((k_dual_apply<double>), dim3(ew_grid(count)), dim3(256), 0, st, (const double*)vp,
                           (double*)v, (const double*)weight, (const double*)sum_in, lam, sigma, nband, count,
                           band_stride, (double*)vp_out);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}
int pfb_dual_apply(int dtype, const void* vp, void* v, const void* weight, const void* sum_in, double lam,
                   double sigma, int nband, size_t nper, void* vp_out, void* stream) {
    return pfb_dual_apply_chunk(dtype, vp, v, weight, sum_in, lam, sigma, nband, nper, nper, vp_out, stream);
}

int pfb_prox_21(int dtype, const void* v, void* result, const void* weight, double lam, double sigma,
                int nband, size_t nper, void* stream) {
    PFB_REQUIRE(v && result && weight && nband > 0 && sigma != 0.0, PFB_ERR_INVALID, "prox_21: bad argument");
    hipStream_t st = as_stream(stream);
    if (dtype == PFB_F32)
        hipLaunchKernelGGL((k_prox_21_l2<float>), dim3(ew_grid(nper)), dim3(256), 0, st, (const float*)v, (float*)result,
                           (const float*)weight, (float)lam, (float)sigma, nband, nper);
    else
        hipLaunchKernelGGL((k_prox_21_l2<double>), dim3(ew_grid(nper)), dim3(256), 0, st, (const double*)v, (double*)result,
                           (const double*)weight, lam, sigma, nband, nper);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

int pfb_dual_update_l2(int dtype, const void* vp, void* v, const void* weight, double lam, double sigma,
                       int nband, size_t nper, void* stream) {
    PFB_REQUIRE(vp && v && weight && nband > 0 && sigma != 0.0, PFB_ERR_INVALID, "dual_update_l2: bad argument");
    hipStream_t st = as_stream(stream);
    if (dtype == PFB_F32)
        hipLaunchKernelGGL((k_dual_update_l2<float>), dim3(ew_grid(nper)), dim3(256), 0, st, (const float*)vp, (float*)v,
                           (const float*)weight, (float)lam, (float)sigma, nband, nper);
    else
        hipLaunchKernelGGL((k_dual_update_l2<double>), dim3(ew_grid(nper)), dim3(256), 0, st, (const double*)vp, (double*)v,
                           (const double*)weight, lam, sigma, nband, nper);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

int pfb_prox_21m(int dtype, const void* v, void* result, const void* weight, double lam, double sigma,
                 int nband, size_t nper, void* stream) {
    PFB_REQUIRE(v && result && weight && nband > 0, PFB_ERR_INVALID, "prox_21m: bad argument");
    hipStream_t st = as_stream(stream);
    if (dtype == PFB_F32)
        hipLaunchKernelGGL((k_prox_21m<float>), dim3(ew_grid(nper)), dim3(256), 0, st, (const float*)v,
                           (float*)result, (const float*)weight, (float)lam, (float)sigma, nband, nper);
    else
        hipLaunchKernelGGL((k_prox_21m<double>), dim3(ew_grid(nper)), dim3(256), 0, st, (const double*)v,
                           (double*)result, (const double*)weight, lam, sigma, nband, nper);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

static int pd_primal_launch(int dtype, const void* xp, const void* xout, const void* xprev, const void* g,
                            const void* gsub, double tau, int positivity, int nband, size_t npix, void* x,
                            double* sums, double* ws, void* stream) {
    PFB_REQUIRE(xp && xout && x && sums && ws && nband > 0, PFB_ERR_INVALID, "pd_primal_update: bad argument");
    PFB_REQUIRE(!gsub || g, PFB_ERR_INVALID, "pd_primal_update: gsub given without g");
    hipStream_t st = as_stream(stream);
    int G = ew_grid(npix) > 1024 ? 1024 : ew_grid(npix);
    const uintptr_t al = (uintptr_t)xp | (uintptr_t)xout | (uintptr_t)xprev | (uintptr_t)g | (uintptr_t)gsub | (uintptr_t)x;
    const size_t V = dtype == PFB_F32 ? 4 : 2;
    static const bool vec_on = [] { const char* e = getenv("PFB_PD_VEC"); return !e || atoi(e); }();
    if (vec_on && (al & 15) == 0 && npix % V == 0) {          // band offsets (npix elements) stay 16-byte aligned
        const size_t nvec = npix / V;
        G = ew_grid(nvec) > 1024 ? 1024 : ew_grid(nvec);
        if (dtype == PFB_F32)
            hipLaunchKernelGGL((k_pd_primal_vec<float, 4>), dim3(G), dim3(256), 0, st, (const float*)xp, (const float*)xout,
                               (const float*)xprev, (const float*)g, (const float*)gsub, (float)tau, positivity, nband,
                               nvec, (float*)x, ws);
        else
            hipLaunchKernelGGL((k_pd_primal_vec<double, 2>), dim3(G), dim3(256), 0, st, (const double*)xp,
                               (const double*)xout, (const double*)xprev, (const double*)g, (const double*)gsub, tau,
                               positivity, nband, nvec, (double*)x, ws);
    } else if (dtype == PFB_F32)
        hipLaunchKernelGGL((k_pd_primal<float>), dim3(G), dim3(256), 0, st, (const float*)xp, (const float*)xout,
                           (const float*)xprev, (const float*)g, (const float*)gsub, (float)tau, positivity, nband,
                           npix, (float*)x, ws);
    else
        hipLaunchKernelGGL((k_pd_primal<double>), dim3(G), dim3(256), 0, st, (const double*)xp,
                           (const double*)xout, (const double*)xprev, (const double*)g, (const double*)gsub, tau,
                           positivity, nband, npix, (double*)x, ws);
    hipLaunchKernelGGL(k_final_sum3, dim3(1), dim3(256), 0, st, ws, G, 3, sums);
    PFB_HIP_CHECK(hipGetLastError());
    return PFB_OK;
}

int pfb_pd_primal_update(int dtype, const void* xp, const void* xout, const void* g, double tau,
                         int positivity, int nband, size_t npix, void* x, double* sums, double* ws,
                         void* stream) {
    return pd_primal_launch(dtype, xp, xout, nullptr, g, nullptr, tau, positivity, nband, npix, x, sums, ws, stream);
}

int pfb_pd_primal_update2(int dtype, const void* xp, const void* xout, const void* xout_prev, const void* g,
                          const void* gsub, double tau, int positivity, int nband, size_t npix, void* x,
                          double* sums, double* ws, void* stream) {
    return pd_primal_launch(dtype, xp, xout, xout_prev, g, gsub, tau, positivity, nband, npix, x, sums, ws, stream);
}

}  // extern "C"
