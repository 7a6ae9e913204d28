// conv_plan.hpp -- the PSF-convolution plan shared by the generic and the pow2 kernels.
//
// Data layout in HBM (per band b; P = nx_psf, Q = ny_psf, M = Q/2):
//   x, out     (nx, ny) real, row-major (caller-owned)
//   T          half-spectrum after the row (y) transform, BLOCKED-TRANSPOSED:
//                T[b][v / VB][i][v % VB]   v in [0, M], i in [0, nx)
//              so a group of VB adjacent frequency columns is ONE contiguous region
//              of nx*VB complex: the column (x) transform streams it with unit stride
//              and the row passes write VB-wide pieces.  The column pass works in
//              place (reads column v of T, writes column v of T).
//   psf_l      the caller's psfhat (P, M+1) re-laid out once into the same blocking:
//                psf_l[b][v / VB][u][v % VB]  u in [0, P)
//   twP, twQ   exp(-2 pi i n / P), exp(-2 pi i n / Q), computed in long double.
#pragma once
#include "common.hpp"
#include "fft_generic.hpp"
#include "pcg_state.hpp"

struct pfb_conv_plan {
    int nx, ny, P, Q, M;       // M = Q/2
    int nband, dtype;
    int VB, nvb;               // column blocking, number of column blocks
    int fast;                  // 1: pow2 register-resident kernels are used
    int long_lines;            // 1: a line does not fit the LDS and the fast path cannot hold the grid: every transform
                               //    runs as multi-launch global-memory passes (fft_long.hpp) -- coverage, not speed
    void* long_ws;             // its workspace (grown on demand, freed with the plan)
    size_t long_ws_bytes;
    pfb::FftFactors frow;      // length M  (row transform on packed reals)
    pfb::FftFactors fcol;      // length P
    void* twP;
    void* twQ;
    void* psf_l;
    void* T;
    double* partials;          // nx * nband doubles (fused dot)
    size_t T_elems_per_band;   // nvb * nx * VB
    size_t psf_elems_per_band; // nvb * P * VB
    size_t workspace_bytes;
    int have_psf;
    int partials_per_band;     // fused-dot partial sums emitted per band by row_inv
    int last_npartials;        // slots per quantity the LAST row-inverse launch wrote (k_sum_partials reads these)
    void* fast_tables;         // pow2 path: per-pass twiddle tables (fftconv_pow2.hip)
    // optional per-stage timing (bench.py roofline): 4 events per apply, up to PROF_MAX applies
    int prof_on, prof_n, prof_tick;   // prof_on = sampling period (every prof_on-th apply is timed)
    hipEvent_t* prof_ev;
    // PCG driver (cgvec.hip): pinned snapshots of the solver's state block + their events, for looking at
    // iteration j-1 while iteration j is already enqueued (created on first use)
    double* pcg_pin;
    hipEvent_t pcg_ev[2];
    // PCG driver -> fast path's inverse row kernel: do the per-iteration bookkeeping in the kernel's tail (pcg_state.hpp).
    // Set by the driver right before an apply and cleared after it; tail_done tells it whether the kernel took it.
    pfb::PcgTail tail;
    unsigned* tail_counter;    // the tail's ticket word (device, zero between launches)
    int tail_done;
};

namespace pfb {
constexpr int PROF_MAX = 512;
// record stage boundary `k` (0..3) of the current apply on `st` when profiling is on
inline void prof_mark(pfb_conv_plan* p, hipStream_t st, int k) {
    if (!p->prof_on) return;
    const bool sampled = (p->prof_tick % p->prof_on) == 0 && p->prof_n < PROF_MAX;
    if (sampled) (void)hipEventRecord(p->prof_ev[p->prof_n * 4 + k], st);
    if (k == 3) {
        if (sampled) p->prof_n++;
        p->prof_tick++;
    }
}
// fftconv.hip: convolution + fused dots left un-summed in plan->partials (see there)
int psfconv_apply_partials(pfb_conv_plan* p, int band0, int nb, const void* x, const void* beam,
                           double wsum, double sigmainv, void* out, const void* dot_with,
                           const void* dot_with2, void* stream);
}  // namespace pfb
