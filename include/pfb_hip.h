/*
 * pfb_hip.h -- C-ABI of libpfb_hip.so: the MI355X (gfx950) implementation of the
 * pfb-imaging PCG / PSF-convolution / wavelet hot path.
 *
 * The reference (ratt-ru/pfb-imaging 0.0.4) is pure Python and has no FFI of its own;
 * its boundary is the set of Python callables the workers import.  Each
 * entry point below names the reference function (file:line under /root/reference)
 * whose arithmetic it replaces; pfb_clean_amd/{operators,opt,prox,wavelets,utils}
 * re-export the reference's Python names on top of these and INTEGRATION.md shows
 * the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns an int status (PFB_OK = 0, negative = error) and never
 *     throws; pfb_last_error() gives a thread-local message for the last failure;
 *   - all array arguments are DEVICE pointers (hipMalloc / torch-ROCm storage),
 *     C-contiguous, in the dtype named by `dtype` (real arrays) or its complex pair
 *     (interleaved re,im);
 *   - `stream` is a hipStream_t passed as void*; nothing synchronises the device
 *     except functions documented to (the *_solve drivers read scalars back);
 *   - the library never allocates caller-visible memory: plans own their twiddles /
 *     re-laid-out PSF / workspaces, callers own every vector;
 *   - one plan may be used by one stream AND one host thread at a time (its spectrum
 *     workspace, fused-dot partials and profiling slots are per plan); distinct plans
 *     are independent and there is no global mutable state (re-entrant like the
 *     reference, which dask may call from several threads with one band each,
 *     pcg.py:346-356 -- one plan per band there).
 */
#ifndef PFB_HIP_H
#define PFB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFB_ABI_VERSION 1

/* dtype of the real arrays; complex arrays use the matching complex type */
#define PFB_F32 0
#define PFB_F64 1

/* status codes */
#define PFB_OK               0
#define PFB_ERR_INVALID     -1   /* bad argument (null pointer, shape mismatch ...)      */
#define PFB_ERR_UNSUPPORTED -2   /* size / radix / dtype outside the supported set       */
#define PFB_ERR_HIP         -3   /* a HIP runtime call failed                            */
#define PFB_ERR_NONFINITE   -4   /* NaN/Inf met where the reference drops into pdb       */
#define PFB_ERR_ALLOC       -5   /* device allocation for a plan failed                  */
#define PFB_ERR_COMM        -6   /* the band-shard exchange failed, was aborted or timed out */

/* pcg exit status (written to pfb_pcg_result.status) */
#define PFB_PCG_CONVERGED     0  /* "Success, converged after k iterations" pcg.py:131-132 */
#define PFB_PCG_MAXIT         1  /* "Max iters reached"                    pcg.py:124-126 */
#define PFB_PCG_ZERO_RESIDUAL 2  /* "Initial residual is zero": x0 returned pcg.py:73-75  */
#define PFB_PCG_BREAKDOWN     3  /* search direction became all-zero       pcg.py:106-107 */

int         pfb_abi_version(void);
const char* pfb_last_error(void);

/* ------------------------------------------------------------------ PSF convolution
 * Replaces pfb/operators/psf.py:11-29 (psf_convolve_slice), :32-56 (psf_convolve_cube)
 * and the Tikhonov-regularised wrappers pfb/operators/hessian.py:129-158
 * (_hessian_psf_slice) and :254-281 (hessian_psf_cube):
 *
 *   out = [beam *] crop( irfft2( rfft2( pad([beam *] x) ) * psfhat ) ) [/ wsum]
 *         + sigmainv * x
 *
 * rfft2 unnormalised (ducc0 r2c inorm=0), irfft2 scaled by 1/(nx_psf*ny_psf) (c2r
 * inorm=2, lastsize=ny_psf), imaginary parts of the DC / Nyquist bins of the last
 * axis ignored like ducc0/pocketfft do.
 */
typedef struct pfb_conv_plan pfb_conv_plan;

/* nx,ny: image; nx_psf,ny_psf(=lastsize): padded PSF grid, ny_psf even,
 * nx <= nx_psf, ny <= ny_psf; every 1-D length must factor into {2,3,5,7,11,13}.
 * nband: leading (imaging band) axis of the cubes this plan serves.
 * Three kernel families behind one plan, chosen here: power-of-two images with nx_psf = 2 nx, ny_psf = 2 ny
 * (64 <= nx <= 8192, 128 <= ny <= 16384 fp32 / 8192 fp64) take the register-FFT fast path; other grids the
 * line-in-LDS coverage kernels while a line fits (<= 10240 complex64 / 5120 complex128), and multi-launch
 * global-memory passes beyond that -- no grid is refused for its size (the host layer embeds arbitrary sizes in the
 * fast path through pfb_psfhat_regrid whenever that is possible: see pfb_clean_amd/operators/psf.py). */
int pfb_psfconv_plan_create(int nx, int ny, int nx_psf, int ny_psf, int nband,
                            int dtype, pfb_conv_plan** plan);
int pfb_psfconv_plan_destroy(pfb_conv_plan* plan);

/* psfhat: (nband, nx_psf, ny_psf/2+1) complex, as produced by
 * pfb/operators/gridder.py:712-714 (r2c(ifftshift(psf))) and divided by wsum in
 * pfb/utils/misc.py:721-723.  Copied into the plan's own layout (once). */
int pfb_psfconv_set_psfhat(pfb_conv_plan* plan, const void* psfhat, void* stream);

/* Produce psfhat ON THE DEVICE from the real PSF and install it:
 *   psfhat = r2c(ifftshift(psf), axes=(0,1), forward, unnormalised)
 * (pfb/operators/gridder.py:712-714 and pfb/utils/fft.py:7-9).  psf: (nband, nx_psf, ny_psf)
 * real, row-major.  psfhat_out: NULL, or (nband, nx_psf, ny_psf/2+1) complex that also
 * receives the transform in the reference's layout (e.g. for the DDS PSFHAT variable).
 * Power-of-two plans (nx_psf = 2 nx, ny_psf = 2 ny: every BASELINE size up to 16384 x 16384 fp64) run it on
 * the fast path's own register-FFT row / column kernels; other plans on the line-in-LDS kernels.  Synchronous. */
int pfb_psfconv_set_psf(pfb_conv_plan* plan, const void* psf, void* psfhat_out, void* stream);

/* The same transform without a plan, for ANY grid whose lengths are 13-smooth and whose ny_psf is even:
 * psfhat (nband, nx_psf, ny_psf/2+1) = r2c(ifftshift(psf)).  Lines that fit the LDS take the
 * one-workgroup-per-line kernels, longer ones (nx_psf > 10240 fp32 / 5120 fp64) multi-launch Stockham passes
 * in global memory -- no length limit.  Plan time (gridder.py:712-714 runs once per gridding run).  Synchronous. */
int pfb_psfhat_from_psf(int dtype, const void* psf, int nband, int nx_psf, int ny_psf, void* psfhat, void* stream);

/* Re-grid a PSF transform: psfhat on the (nx_psf, ny_psf) grid -> psfhat2 of the SAME image-space
 * PSF on an (nx_psf2, ny_psf2) grid (both (nband, n, m/2+1) complex, row-major), for images of
 * (nx, ny) pixels: only the offsets |du| < nx, |dv| < ny a convolution of such an image touches
 * are carried over (periodically in the old grid, so the wrap-around of grids with
 * nx_psf < 2 nx is reproduced), which needs nx_psf2 >= 2 nx - 1, ny_psf2 >= 2 ny - 1.
 * The convolution psf.py:11-56 on the old grid and on the new grid (image zero-padded to
 * nx_psf2/2 x ny_psf2/2, result cropped) then agree to rounding; the host layer uses this to run
 * arbitrary image sizes on the power-of-two fast path.  Plan-time, synchronous.  Lengths 13-smooth, last axes
 * even; lines of any length (beyond the LDS: multi-launch global-memory passes). */
int pfb_psfhat_regrid(int dtype, const void* psfhat, int nband, int nx, int ny, int nx_psf, int ny_psf,
                      int nx_psf2, int ny_psf2, void* psfhat2, void* stream);

/* Apply to bands [band0, band0+nb) of the plan.  x, out: (nb, nx, ny) real; out may
 * not alias x.  beam: (nb, nx, ny) or NULL.  wsum <= 0 means "no division"
 * (reference wsum=None).  sigmainv may be 0.
 * If dot_with != NULL (same shape as x) the fp64 sum over all nb bands of
 * dot_with*out is written to *dot_out (device double) -- the fused p.Ap of
 * pcg.py:91.  Asynchronous on `stream`. */
int pfb_psfconv_apply(pfb_conv_plan* plan, int band0, int nb,
                      const void* x, const void* beam, double wsum, double sigmainv,
                      void* out, const void* dot_with, double* dot_out, void* stream);

/* Same, with the three fused fp64 sums the predictive line search of pfb_pcg_solve needs:
 * dots_out[0] = <dot_with, out>, dots_out[1] = <dot_with2, out> (0 if dot_with2 is NULL),
 * dots_out[2] = <out, out>  -- all over the nb bands (pcg.py:91,95 and the backtracking
 * loop :96-101 evaluated without further passes over the vectors). */
int pfb_psfconv_apply_dots(pfb_conv_plan* plan, int band0, int nb,
                           const void* x, const void* beam, double wsum, double sigmainv,
                           void* out, const void* dot_with, const void* dot_with2,
                           double* dots_out, void* stream);

/* Introspection for benchmarks / tests */
int    pfb_psfconv_plan_info(const pfb_conv_plan* plan, int* fast_path, int* vb,
                             size_t* workspace_bytes);

/* Per-stage timing for benchmarks: while on > 0, every on-th apply (on = 1: every apply)
 * records HIP events on its stream around the three kernels (row-forward, column,
 * row-inverse) -- an event record costs ~6 us of stream time, hence the sampling period;
 * get_profile waits for them, returns the summed milliseconds per stage and the number of
 * applies covered (at most 512 between calls), and resets the counters. */
int pfb_psfconv_set_profiling(pfb_conv_plan* plan, int on);
int pfb_psfconv_get_profile(pfb_conv_plan* plan, double stage_ms[3], int* napply);

/* ------------------------------------------------------------- CG vector kernels
 * Fused replacements for the numpy passes of pfb/opt/pcg.py:77-111 and
 * pfb/utils/misc.py:1316-1351 (norm_diff).  `ws` is a caller-provided device
 * scratch of at least PFB_REDUCE_WS_DOUBLES doubles; results are device doubles,
 * accumulated in fp64 in a fixed (deterministic) order. */
#define PFB_REDUCE_WS_DOUBLES 8192

/* out[0] = sum a*b                                   (np.vdot on real arrays) */
int pfb_dot(int dtype, const void* a, const void* b, size_t n,
            double* out, double* ws, void* stream);
/* out[0] = sum (x-xp)^2, out[1] = sum x^2            (norm_diff partial sums) */
int pfb_norm_diff_sums(int dtype, const void* x, const void* xp, size_t n,
                       double* out, double* ws, void* stream);
/* out[0] = 1.0 if any element of a is non-zero else 0.0   (np.any) */
int pfb_any_nonzero(int dtype, const void* a, size_t n, double* out, double* ws,
                    void* stream);
/* y = a*x + b*y elementwise (a, b host scalars) */
int pfb_axpby(int dtype, double a, const void* x, double b, void* y, size_t n,
              void* stream);

/* -------------------------------------------------------------------- fused PCG
 * Replaces pfb/opt/pcg.py:53-136 (pcg) for the operator
 *   A(x) = hessian_psf(plan, beam, wsum, sigmainv)           (hessian.py:129-158/254-281)
 * and the preconditioner M(r) = r / mdiv (mdiv = sigmainv as pcg.py:264-267;
 * mdiv <= 0 means M = identity).  NaN/Inf propagate silently exactly as in the reference.
 * Works on bands [band0, band0+nb) as ONE system (np.vdot over the whole cube, the
 * fluxmop semantics fluxmop.py:193-199); call once per band for pcg_psf semantics.
 *
 * backtrack: 0 = off; 1 = the reference's loop verbatim (every rejected step re-runs the
 * vector update); 2 = predictive: the same decisions from the quadratic
 * rnorm(alpha) = rnorm + (2 alpha <r,Ap> + alpha^2 <Ap,Ap>)/mdiv (three scalars fused into the
 * convolution epilogue), then ONE vector update with the accepted alpha.  1 and 2 differ only
 * by rounding in the comparison rnorm_next > rnorm.
 *
 * allreduce: optional hook for band-sharded multi-GPU solves -- called on `stream`
 * order with a device buffer of `count` doubles that must be summed in place over
 * all ranks (RCCL); NULL for single-GPU.
 * Failure protocol (the reference's sums run over all bands in one process, pfb/opt/pcg.py:90-107, and cannot
 * lose a participant; a sharded solve can).  The same function doubles as the health interface of the exchange:
 *   count  > 0   the exchange itself; non-zero return = it could not be issued
 *   count == 0   PROBE (dev_buf NULL): non-zero = the exchange has failed asynchronously (a peer died, the
 *                communicator was aborted); called by the solver while it waits for the device
 *   count  < 0   ABORT (dev_buf NULL): tear the exchange down so that neither this rank nor its peers stay blocked
 *                in a collective; called once before the solver gives up
 * A solver with a hook never waits for the device unboundedly: it polls, probes, and after PFB_COMM_TIMEOUT_S
 * seconds (environment, default 600) without progress aborts the exchange.  In all three cases pfb_pcg_solve
 * returns PFB_ERR_COMM; the caller's buffers then hold no result and the process should exit non-zero.
 */
typedef int (*pfb_allreduce_fn)(void* ctx, double* dev_buf, int count, void* stream);

typedef struct {
    int    status;       /* PFB_PCG_* */
    int    iters;        /* k at exit */
    int    matvecs;      /* number of A applications the REFERENCE loop makes (k + 1 unless early exit); small
                          * problems run one iteration ahead of the host's look at eps past minit, so one
                          * further, discarded application may have been executed (PFB_PCG_LOOKAHEAD=0: never) */
    int    backtracks;   /* total backtracking steps taken */
    double eps;          /* last norm_diff(x, xp) */
    double rnorm;        /* last r.y */
} pfb_pcg_result;

/* b, x: (nb, nx, ny).  x holds x0 on entry and the solution on exit (for
 * PFB_PCG_ZERO_RESIDUAL it is left untouched = x0, as the reference returns x0).
 * r_out: optional (nb,nx,ny) residual A x - b on exit (return_resid=True).
 * work: device scratch of pfb_pcg_work_bytes() bytes.
 * Synchronises `stream` (reads scalars back once per iteration). */
size_t pfb_pcg_work_bytes(const pfb_conv_plan* plan, int nb);
int pfb_pcg_solve(pfb_conv_plan* plan, int band0, int nb,
                  const void* b, void* x, void* r_out,
                  const void* beam, double wsum, double sigmainv, double mdiv,
                  double tol, int maxit, int minit, int backtrack,
                  void* work, pfb_allreduce_fn allreduce, void* allreduce_ctx,
                  pfb_pcg_result* result, void* stream);

/* ------------------------------------------------ band-shard exchange (RCCL, called from C)
 * The reference sums the CG inner products over ALL bands inside one process
 * (pfb/opt/pcg.py:92-107 on (nband, nx, ny) arrays); with one process per GPU and the bands sharded
 * (pfb/opt/pcg.py:320-356 is the reference's per-band parallelism) those sums need one all-reduce
 * of 3..7 doubles per iteration.  pfb_comm_allreduce is a ready-made pfb_allreduce_fn (ctx = the
 * communicator): an RCCL all-reduce enqueued on the solver's own stream, nothing on the host per
 * iteration.  RCCL is bound at run time (the copy already loaded in the process -- PyTorch's --
 * else librccl.so.1; pfb_comm_bind names one explicitly); single-GPU callers never load it.
 *
 *   rank 0: pfb_comm_unique_id(id)  -> ship the 128 bytes to every rank (any channel)
 *   all   : pfb_comm_init(rank, nranks, id, &comm)   collective, binds to the CURRENT device
 *           pfb_pcg_solve(..., pfb_comm_allreduce, comm, ...)
 *           pfb_comm_destroy(comm)
 */
#define PFB_COMM_ID_BYTES 128
typedef struct pfb_comm pfb_comm;
int pfb_comm_bind(const char* librccl_path);            /* optional; NULL = default search */
int pfb_comm_unique_id(void* id128);
int pfb_comm_init(int rank, int nranks, const void* id128, pfb_comm** out);
int pfb_comm_destroy(pfb_comm* comm);
int pfb_comm_info(const pfb_comm* comm, int* rank, int* nranks, int* device, int* rccl_version);
int pfb_comm_allreduce(void* comm, double* dev_buf, int count, void* stream);   /* incl. the probe / abort protocol */
/* asynchronous state of the communicator (ncclCommGetAsyncError): PFB_OK, or PFB_ERR_COMM once a collective on it
 * has failed or it was aborted */
int pfb_comm_check(pfb_comm* comm);
/* ncclCommAbort: frees the communicator's resources and makes every collective still queued on it, here AND on the
 * peers, complete with an error instead of blocking; the handle stays valid for pfb_comm_destroy only */
int pfb_comm_abort(pfb_comm* comm);

/* ----------------------------------------------------------- wavelets / prox / PD
 * Replaces pfb/wavelets/wavelets.py:175-213 (dwt2d), :261-315 (idwt2d) and
 * pfb/operators/psi.py:187-256 (psi_band.dot / hdot) over all (band, basis) pairs.
 * Coefficient cube layout identical to the reference: (nband, nbasis, Nymax, Nxmax),
 * each basis block transposed (y-major) in [0:Ntoty, 0:Ntotx]; cells outside the
 * union of level blocks are neither written by dot nor read by hdot.
 */
typedef struct pfb_psi_plan pfb_psi_plan;

/* basis_k[i] = 0 for 'self', K for 'dbK' (1..9).  filters: nbasis*4*18 doubles,
 * [basis][dec_lo,dec_hi,rec_lo,rec_hi][tap] (unused taps 0), the table PyWavelets
 * would supply (psi.py:37-43). */
int pfb_psi_plan_create(int nband, int nx, int ny, int nbasis, const int* basis_k,
                        const double* filters, int nlevel, int dtype,
                        pfb_psi_plan** plan);
int pfb_psi_plan_destroy(pfb_psi_plan* plan);
int pfb_psi_plan_dims(const pfb_psi_plan* plan, int* nymax, int* nxmax);
/* analysis  psi.py:187-218 : x (nband,nx,ny) -> alpha (nband,nbasis,Nymax,Nxmax) */
int pfb_psi_dot(pfb_psi_plan* plan, const void* x, void* alpha, void* stream);
/* synthesis psi.py:221-256 : alpha -> xo (nband,nx,ny), summed over bases */
int pfb_psi_hdot(pfb_psi_plan* plan, const void* alpha, void* xo, void* stream);

/* pfb/prox/prox_21m.py:76-103 dual_update_numba, in place on v.
 * vp, v: (nband, nbasis, nymax, nxmax); weight: (nbasis, nymax, nxmax).
 * If vp_out != NULL it additionally receives 2*v_new - vp (primal_dual.py:137). */
int pfb_dual_update(int dtype, const void* vp, void* v, const void* weight,
                    double lam, double sigma, int nband, size_t nper,
                    void* vp_out, void* stream);
/* The same update with the bands sharded over GPUs: pfb_dual_bandsum writes the
 * LOCAL band sum of vtilde = vp + sigma v into sum_out (nper values); the caller all-reduces
 * that plane (RCCL, bandwidth bound) and pfb_dual_apply finishes with the GLOBAL sum. */
int pfb_dual_bandsum(int dtype, const void* vp, const void* v, double sigma, int nband,
                     size_t nper, void* sum_out, void* stream);
int pfb_dual_apply(int dtype, const void* vp, void* v, const void* weight, const void* sum_in,
                   double lam, double sigma, int nband, size_t nper, void* vp_out, void* stream);
/* The same two steps on a CHUNK of the coefficient plane, so that the exchange of one chunk (reduce-scatter +
 * all-gather over xGMI) overlaps the band sums / threshold of its neighbours: every pointer is pre-offset to the
 * chunk's first coefficient, `count` coefficients are processed, bands of vp / v / vp_out are `band_stride`
 * elements apart (the full plane's nper), weight / sum are the chunk's slices. */
int pfb_dual_bandsum_chunk(int dtype, const void* vp, const void* v, double sigma, int nband, size_t count,
                           size_t band_stride, void* sum_out, void* stream);
int pfb_dual_apply_chunk(int dtype, const void* vp, void* v, const void* weight, const void* sum_in,
                         double lam, double sigma, int nband, size_t count, size_t band_stride,
                         void* vp_out, void* stream);

/* The band-l2-NORM variants (pfb/prox/prox_21.py; the "m" functions threshold |sum over bands|, these the Euclidean norm
 * over bands).  pfb_prox_21: prox_21_numba (prox_21.py:23-48) -- and, with sigma = 1, the array form prox_21 (:5-20);
 * pfb_dual_update_l2: dual_update_numba of the same file (:62-88), in place on v.  Shapes as for the "m" forms. */
int pfb_prox_21(int dtype, const void* v, void* result, const void* weight,
                double lam, double sigma, int nband, size_t nper, void* stream);
int pfb_dual_update_l2(int dtype, const void* vp, void* v, const void* weight,
                       double lam, double sigma, int nband, size_t nper, void* stream);
/* pfb/prox/prox_21m.py:31-61 prox_21m_numba */
int pfb_prox_21m(int dtype, const void* v, void* result, const void* weight,
                 double lam, double sigma, int nband, size_t nper, void* stream);
/* primal_dual.py:140-146: x = xp - tau*(xout + g); positivity 0|1|2 over nband;
 * sums[0..1] receive the norm_diff partial sums of (x, xp), sums[2] = any(x). */
int pfb_pd_primal_update(int dtype, const void* xp, const void* xout, const void* g,
                         double tau, int positivity, int nband, size_t npix,
                         void* x, double* sums, double* ws, void* stream);
/* The same statement with two fusions (either pointer may be NULL):
 *   xout_prev: the synthesis term is 2*xout - xout_prev.  psi^H is linear, so psi^H(2 v - vp) of
 *              primal_dual.py:137-138 equals 2 psi^H(v) - psi^H(vp), and psi^H(vp) is the previous iteration's
 *              psi^H(v): the caller synthesises v itself and the cube 2 v - vp is never written nor read;
 *   gsub:      the gradient is g - gsub (grad(x) = conv(x) - dirty, workers/spotless.py:259-260: the data term is
 *              subtracted here instead of in a pass of its own). */
int pfb_pd_primal_update2(int dtype, const void* xp, const void* xout, const void* xout_prev, const void* g,
                          const void* gsub, double tau, int positivity, int nband, size_t npix,
                          void* x, double* sums, double* ws, void* stream);

/* ------------------------------------------------------------ Clark CLEAN sub-minor loop
 * pfb/deconv/clark.py:29-84 (subminor + subtract).  A: (nband, nact) active-set values, updated
 * in place; Ip, Iq: (nact) int32 pixel indices of the active set; psf: (nband, nx_psf, ny_psf);
 * model: (nband, nx, ny), receives gamma * component / wsums[band] at the chosen pixel for bands
 * with wsums > 0; loop `while |sum_b A[b, pq]| > th and k < maxit` with pq the FIRST arg-max of
 * (sum_b A)^2 like numpy.  The whole loop runs in one resident workgroup (no per-iteration
 * launch); *iters_out (device int, may be NULL) receives the number of components taken.
 * Needs nx_psf/2 >= nx - 1 and ny_psf/2 >= ny - 1 (the reference's overlap mask is then all
 * true; for smaller PSFs the reference mis-indexes its shrunken active set), nband <= 64. */
int pfb_clark_subminor(int dtype, void* A, size_t nact, int nband, const void* psf, int nx_psf,
                       int ny_psf, const int* Ip, const int* Iq, void* model, int nx, int ny,
                       const void* wsums, double gamma, double th, int maxit, int* iters_out,
                       void* stream);

/* Hogbom CLEAN, pfb/deconv/hogbom.py:8-74.  IR (nband, nx, ny): on entry the dirty cube, on exit the
 * residual; model (nband, nx, ny): zero on entry, receives the components; wsums (nband) = per-band PSF
 * peak (hogbom.py:27), all > 0.  Loop `while IRmax > max(pf * IRmax0, threshold) and k < maxit`, peak =
 * FIRST arg-max of (sum_b IR)^2.  work: device scratch of at least 17 KiB (loop state, arg-max partials).
 * Synchronous (the host looks at the loop state every 64 iterations); *k_out / *irmax_out are HOST
 * outputs.  The PSF must cover every shift: nx_psf/2 >= nx - 1, ny_psf/2 >= ny - 1. */
int pfb_hogbom(int dtype, void* IR, const void* psf, void* model, const void* wsums, int nband, int nx,
               int ny, int nx_psf, int ny_psf, double gamma, double pf, double threshold, int maxit,
               void* work, size_t work_bytes, int* k_out, double* irmax_out, void* stream);

/* Band coupling of the fwdbwd parametrisations, pfb/utils/misc.py:1366-1375 (freqmul):
 * out[k, :] = [post[k, :] *] sum_l A[k, l] * ([pre[l, :] *] x[l, :]) over npix pixels; A: (nband, nband)
 * row-major on the device; pre / post (nband, npix) or NULL fuse the elementwise factors of the 'exp'
 * parametrisation (misc.py:1412-1416).  out must not alias x; nband <= 64. */
int pfb_freqmul(int dtype, const void* A, const void* x, void* out, int nband, size_t npix,
                const void* pre, const void* post, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PFB_HIP_H */
