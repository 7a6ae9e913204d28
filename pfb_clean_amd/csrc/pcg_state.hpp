// pcg_state.hpp -- the device-resident scalar state of pfb_pcg_solve (cgvec.hip) and the per-iteration bookkeeping on
// it, shared with the inverse row kernels of the fast convolution path (fftconv_pow2.hip): their LAST-ARRIVING
// workgroup sums the fused inner products and runs the end-of-iteration / begin-of-iteration logic itself, which takes
// the one single-workgroup launch per iteration (k_iter_sums) out of the PCG loop.
#pragma once
#include "common.hpp"

namespace pfb {

// scalar slots in the device state array
enum { S_PAP = 0, S_RAP = 1, S_APAP = 2,          // <p,Ap>, <r,Ap>, <Ap,Ap>   (conv epilogue)
       S_ANY = 3,                                 // count(p != 0) of the direction in use
       S_RHON = 4, S_NUM = 5, S_DEN = 6,          // <r',y'>, |x'-x|^2, |x'|^2 (update kernel)
       S_RHO = 7, S_ALPHA = 8, S_BETA = 9, S_NBT = 10,
       S_DEAD = 11,                               // p became all-zero: later work is a no-op
       S_K = 12, S_EPS = 13, S_EPSP = 14, S_NBTSUM = 15,
       S_STOP = 16,                               // the stopping rule fired on the device: later work is a no-op
       S_TOL = 17, S_MINIT = 18, S_MAXIT = 19,    // the rule's parameters (set once per solve)
       S_NSCALAR = 20 };


// ---- device-side loop bookkeeping of the sync-free driver
__device__ __forceinline__ void iter_begin_dev(double* S, double mdiv, int predict) {
    if (S[S_DEAD] != 0.0 || S[S_STOP] != 0.0) return;
    if (S[S_ANY] == 0.0) {                 // the direction built last iteration is all zero:
        S[S_DEAD] = 1.0;                   // the reference broke BEFORE k += 1 (pcg.py:106-108)
        S[S_K] -= 1.0;
        S[S_EPS] = S[S_EPSP];
        return;
    }
    const double rho = S[S_RHO];
    double alpha = rho / S[S_PAP];
    int nbt = 0;
    if (predict == 1 || predict == 2) {
        const double d = mdiv > 0.0 ? mdiv : 1.0;
        const double s1 = S[S_RAP] / d, s2 = S[S_APAP] / d;
        while (rho + (2.0 * alpha * s1 + alpha * alpha * s2) > rho && nbt < 200) { alpha *= 0.75; ++nbt; }
    }
    S[S_ALPHA] = alpha;
    S[S_NBT] = (double)nbt;
    if (predict >= 2) {                    // fused update+direction: beta from rho(alpha)
        const double d = mdiv > 0.0 ? mdiv : 1.0;
        S[S_BETA] = (rho + (2.0 * alpha * S[S_RAP] + alpha * alpha * S[S_APAP]) / d) / rho;
    }
}
__device__ __forceinline__ void iter_end_dev(double* S, int fused) {
    if (S[S_DEAD] != 0.0 || S[S_STOP] != 0.0) return;
    if (!fused) S[S_BETA] = S[S_RHON] / S[S_RHO];
    S[S_RHO] = S[S_RHON];
    const double k = S[S_K] + 1.0;
    const double eps = sqrt(S[S_NUM] / (1e-12 + S[S_DEN]));
    S[S_K] = k;
    S[S_EPSP] = S[S_EPS];
    S[S_EPS] = eps;
    S[S_NBTSUM] += S[S_NBT];
    // the reference's loop condition (pcg.py:86), evaluated where the numbers are: an iteration the host
    // enqueued speculatively behind this one finds S_STOP set and changes nothing
    if (!((eps > S[S_TOL] || k < S[S_MINIT]) && k < S[S_MAXIT])) S[S_STOP] = 1.0;
}

// Seven independent sums, each by ONE wave in a fixed lane-strided order (deterministic, and the same whether it runs
// in k_iter_sums or in the tail of an inverse row kernel): wave w takes conv quantity w (w < 3: the per-workgroup
// partials `cp` of <p,Ap>, <r,Ap>, <Ap,Ap>) and update quantity w (w < 4: what the previous iteration's fused update
// left in `ws`, when `have_upd`), then -- unless an all-reduce has to come first (`logic` == 0) -- thread 0 ends that
// iteration and begins this one.  Call with >= 256 threads, all of them; `vals`: 8 doubles of LDS.
// COH: the partials were written by OTHER workgroups of the SAME launch (tail of a row kernel): read them with
// agent-scope atomic loads, which do not hit in this XCD's possibly stale L2 lines.
template <bool COH>
__device__ __forceinline__ double pcg_ld(const double* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void pcg_iter_sums(const double* __restrict__ cp, int ncp, const double* __restrict__ ws, int G,
                                              int have_upd, double* __restrict__ S, double mdiv, int predict, int logic,
                                              double* vals) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w < 3) {
        double acc = 0.0;
        if constexpr (COH) {
            // the coherent loads go to memory (~1-2 us each): all of a lane's loads are issued before the first add --
            // the adds keep k_iter_sums' order (k = lane, lane + 64, ...), so the sum is bit-identical
            constexpr int MAXL = 16;                          // ncp <= 1024 (checked on the host)
            double t[MAXL];
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                const int k = lane + 64 * j;
                t[j] = k < ncp ? pcg_ld<true>(cp + (size_t)w * ncp + k) : 0.0;
            }
#pragma unroll
            for (int j = 0; j < MAXL; ++j) if (lane + 64 * j < ncp) acc += t[j];
        } else {
            for (int k = lane; k < ncp; k += 64) acc += cp[(size_t)w * ncp + k];
        }
        acc = wave_sum(acc);
        if (lane == 0) vals[w] = acc;
    }
    if (have_upd && w < 4) {
        double acc = 0.0;
        for (int g = lane; g < G; g += 64) acc += ws[(size_t)w * G + g];
        acc = wave_sum(acc);
        if (lane == 0) vals[3 + w] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S[S_PAP] = vals[0]; S[S_RAP] = vals[1]; S[S_APAP] = vals[2];
        if (have_upd) { S[S_RHON] = vals[3]; S[S_NUM] = vals[4]; S[S_DEN] = vals[5]; S[S_ANY] = vals[6]; }
        if (logic) {
            if (have_upd) iter_end_dev(S, 1);
            iter_begin_dev(S, mdiv, predict);
        }
    }
}

// What the PCG driver hands to the convolution when it wants the bookkeeping done in the inverse row kernel's tail
// (S == nullptr: no tail).  `counter`: one zero-initialised device word per plan, reset by the workgroup that uses it.
struct PcgTail {
    double* S;
    const double* ws;
    unsigned* counter;
    double mdiv;
    int G, have_upd, predict;
};

// Tail of a kernel whose workgroups each wrote one slot per quantity of `cp` ([3][ncp]) THROUGH pcg_partial_store: the
// last workgroup to arrive (agent-scope ticket) does pcg_iter_sums.  Every thread of every workgroup must call it;
// `sh`: 9 doubles of LDS.
// Coherence without fences.  An agent-scope release fence here is `buffer_wbl2`: it writes back EVERY dirty line of the
// XCD's L2 -- megabytes of the kernel's own output rows -- once per workgroup (measured: +11 us on a 75 us kernel).
// Only the partials have to cross XCDs, so they alone are written with agent-scope atomic stores (sc1: written
// through), the thread waits for their acknowledgement (vmcnt) before it takes its ticket (a device-scope atomic), and
// the last workgroup reads them with agent-scope atomic loads (sc1: never served from its own, possibly stale L2).
__device__ __forceinline__ void pcg_partial_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pcg_tail(const PcgTail& t, const double* cp, int ncp, unsigned nblocks, double* sh) {
    if (!t.S) return;
    unsigned* ticket = reinterpret_cast<unsigned*>(sh + 8);
    if (threadIdx.x == 0) {                        // thread 0 wrote this workgroup's partials just before
        __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) (gfx9 encoding: expcnt / lgkmcnt left alone): the write-through stores are acknowledged
        *ticket = __hip_atomic_fetch_add(t.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*ticket != nblocks - 1) return;
    pcg_iter_sums<true>(cp, ncp, t.ws, t.G, t.have_upd, t.S, t.mdiv, t.predict, 1, sh);
    if (threadIdx.x == 0) __hip_atomic_store(t.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
}

}  // namespace pfb
