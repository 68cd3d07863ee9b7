/*
 * sunode_amd.h -- C ABI of the MI355X batched BDF + adjoint engine (libsunode_amd.so).
 *
 * This is the drop-in boundary for the ONE hot path of pymc-devs/sunode: where the
 * reference binds SUNDIALS CVODES through cffi (/root/reference/sunode/build_cvodes.py:60-75,
 * declarations /root/reference/include/cvodes/16_cvodes.h) and drives one integrator
 * per call, this library drives B integrators per call on one GPU.  Each entry point
 * names the reference call sequence it replaces.  Plain pointers and sizes only.
 *
 * Memory spaces: every array argument of one call lives either in host memory
 * (SA_MEM_HOST: the library stages it through its own device buffers and returns after
 * the results are back) or in device memory of the solver's GPU (SA_MEM_DEVICE: the
 * call only enqueues work on the solver's stream; use sa_synchronize()).  One exception,
 * stated here because callers that overlap work care: sa_solve_backward_batch[_all] waits
 * once, before it launches, for the preceding sa_solve_forward_batch to finish (a 4-byte
 * read-back that tells it whether every stored trajectory fitted the resident arena; if
 * not, it also fetches the per-instance point counts and re-integrates tile by tile).
 * sa_solve_batch, sa_solve_forward_batch and sa_eval_callbacks never synchronise.
 * The caller owns every array; the library owns the handle, the per-instance
 * trajectory arena (CVODES' adjoint "data points") and its staging buffers.
 *
 * Per-instance failures never abort a batch: status[b] carries the CVODES return code
 * (16_cvodes.h:45-106; 0 = CV_SUCCESS, -1 = CV_TOO_MUCH_WORK after the retry budget, ...)
 * and the instance's outputs are NaN (mirrors wrappers/as_pytensor.py:289-290,339-341).
 * Function return values < 0 are API / HIP errors (see SA_ERR_*; text via sa_last_error()).
 *
 * Threading: one handle per device; calls on one handle must be serialised by the caller
 * (same rule as one CVODES memory block per sunode solver object).
 */
#ifndef SUNODE_AMD_H
#define SUNODE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_ABI_VERSION 3          /* device ABI (sa_meta[3] of a code object) */

#define SA_MEM_HOST 0
#define SA_MEM_DEVICE 1

#define SA_OK 0
#define SA_ERR_HIP (-1001)        /* a HIP runtime call failed */
#define SA_ERR_ARG (-1002)        /* invalid argument / call sequence */
#define SA_ERR_MODULE (-1003)     /* code object missing, wrong arch or ABI mismatch */

/* per-instance status beyond the CVODES codes.  From sa_solve_forward_batch: the instance would store more than
   sa_options.traj_capacity points -- the integration stops there (bounded work, whatever the inputs) and y_out is
   NaN like for every other failure; the backward call answers CV_NO_FWD (-102) for it.  From
   sa_solve_backward_batch: the 64 instances of its group do not fit arena_bytes even alone (forward results are
   complete and valid; gradients NaN).  Distinct from CV_TOO_MUCH_WORK (-1), which keeps its CVODES meaning
   (mxstep x retries of one CVode call). */
#define SA_STATUS_ARENA_FULL (-9001)

/* statistics slots: stats[b*SA_N_STATS + slot], int64 (CVodeGetNumSteps & friends,
   16_cvodes.h:208-235, 15_cvodes_ls.h:94-106; summed over the backward restarts) */
#define SA_N_STATS 16
#define SA_ST_NST 0        /* accepted BDF steps */
#define SA_ST_NFE 1        /* right-hand-side evaluations */
#define SA_ST_NSETUPS 2    /* LU factorisations of I - gamma*J */
#define SA_ST_NJE 3        /* Jacobian evaluations */
#define SA_ST_NNI 4        /* Newton iterations */
#define SA_ST_NCFN 5       /* nonlinear convergence failures */
#define SA_ST_NETF 6       /* error-test failures */
#define SA_ST_QLAST 7      /* order of the last step */
#define SA_ST_NPTS 8       /* stored trajectory points */
#define SA_ST_NFQE 9       /* quadrature rhs evaluations (backward) */
#define SA_ST_NETFQ 10     /* quadrature error-test failures (backward) */
#define SA_ST_NINTERP 11   /* y(t) interpolations (backward) */
#define SA_ST_NREBUILD 12  /* divided-difference table rebuilds (backward) */
#define SA_ST_RETRIES 13   /* CV_TOO_MUCH_WORK returns absorbed by the retry loop */
#define SA_ST_ATTEMPTS 14  /* step attempts = trips through the kernel's main loop */

typedef struct sa_solver sa_solver;

/* Solver configuration; mirrors the keyword arguments / hard-coded settings of
   Solver.__init__ (solver.py:242-254) and AdjointSolver.__init__/_init_backward
   (solver.py:531-533, 599, 614-615). */
typedef struct sa_options {
    int32_t struct_size;       /* = sizeof(sa_options) */
    int32_t device;            /* HIP device ordinal */
    double rtol;               /* forward reltol (CVodeSStolerances / SVtolerances) */
    const double *atol;        /* forward abstol, host pointer to n_states values */
    double rtolB, atolB;       /* backward problem (CVodeSStolerancesB; sunode hard-codes 1e-10) */
    double rtolQB, atolQB;     /* backward quadratures (CVodeQuadSStolerancesB; 1e-10), errconQB = 1 */
    int32_t mxstep;            /* internal steps per CVode call (CVODES default 500) */
    int32_t max_retries_fwd;   /* sunode: 5  (solver.py:467) */
    int32_t max_retries_bwd;   /* sunode: 50 (solver.py:724) */
    int32_t traj_capacity;     /* most stored points per instance the caller allows (the role of CVodeAdjInit's
                                  steps x check points; sunode: 500 000).  NOT an allocation size: the arena holds
                                  what the batch needs, see arena_bytes */
    const double *constraints; /* CVodeSetConstraints (solver.py:230-233, 569-572): host pointer to n_states values
                                  in {0, +-1, +-2} or NULL; forward problem only; needs a code object built with
                                  constraint support (SA_CONSTRAINTS), otherwise the vector is ignored */
    int64_t arena_bytes;       /* budget of the trajectory arena; 0 = default (96 GiB, at most 60 % of the free HBM).
                                  A batch whose stored steps fit stays resident between sa_solve_forward_batch and
                                  sa_solve_backward_batch; a larger one is re-integrated tile by tile inside the
                                  backward call (check-point semantics; results identical, one extra forward pass) */
} sa_options;

int sa_abi_version(void);
const char *sa_last_error(void);

/* HIP devices visible to the library and their memory.  For callers that shard a batch over several handles from
   one process (sunode_amd.solver: AdjointSolver(problem, devices=[0..7]) -- the reference's call pattern is one
   solver object inside one PyMC process, wrappers/as_pytensor.py:279-344): one sa_solver per entry, each call
   issued from its own host thread (every entry point selects its handle's device for the calling thread; error
   strings are per thread), arena budgets of handles that share a device divided by the caller. */
int sa_device_count(int32_t *count);
int sa_device_memory(int32_t device, int64_t *free_bytes, int64_t *total_bytes);

/* Load the per-problem code object (generated callbacks + integrator kernels, built by
   sunode_amd._native from bdf_kernels.hip) on opt->device.
   Replaces CVodeCreate/CVodeInit/CVodeSetUserData/SUNLinSol_Dense/CVodeSetLinearSolver/
   CVodeSetJacFn (+ CVodeAdjInit/CVodeCreateB/CVodeInitB/...B), solver.py:214-240, 565-622. */
int sa_solver_create(const char *code_object_path, const sa_options *opt, sa_solver **out);
void sa_solver_destroy(sa_solver *s);
int sa_solver_set_options(sa_solver *s, const sa_options *opt);   /* tolerances / budgets */
int sa_solver_sizes(const sa_solver *s, int32_t *n_states, int32_t *n_sub, int32_t *n_rem);

/* Differential guard.  The reference accepts ANY sympy system (symode/problem.py:25-33): every user model is a new
   code object, compiled by a toolchain whose AMDGPU back end was caught miscompiling this source family once
   (SIOptimizeVGPRLiveRange, profiles/r04_sens_anomaly.txt) -- and a GPU box has no CPU oracle to notice.  So a handle
   can be given the CONSERVATIVE build of the same source (sunode_amd._native.build_code_object(..., safe=True): the
   pass off): on the first batch of each kind of call -- SA_GUARD_PLAIN sa_solve_batch, SA_GUARD_ADJOINT
   sa_solve_forward_batch + sa_solve_backward_batch[_all], SA_GUARD_SENS sa_solve_sens_batch -- the library runs a
   SAMPLE of min(B, n_sample) instances through BOTH code objects on two internal handles (created when the first
   check is due) and compares statuses, all counters and every fp64 output bit for bit.  The sample is chosen after the
   batch's own launch, from its statuses and counters: the first 16 instances, one instance per failure code, the
   instances with the most steps / error-test failures / convergence failures / set-ups / Jacobian evaluations /
   retries, and an even stride over the rest -- rarely taken paths are compared if any instance of the batch takes
   them.  Equal: the kind is verified (a check with fewer than 16 instances is repeated on the next call, at most
   three times).  Any difference: the handle switches to the conservative code object for good (sa_guard_state:
   using_safe) and, if the difference showed in the backward pass, repeats the forward pass of the batch with it
   before it integrates backward (the forward OUTPUTS the caller already received are not recomputed).
   While a kind is verified, SA_MEM_HOST calls -- whose statuses / counters reach the host anyway -- are scanned for a
   status code, a first error-test / convergence failure or a doubled step count the verified sample never showed;
   the first such batch within the next 32 calls of the kind is checked again, once (sa_guard_state reports
   SA_GUARD_RECHECK_OPEN in `pending` while that window is open).  Forward passes of the adjoint kind that no backward
   call follows are checked at most twice; the check resumes with the first forward call after a backward call.
   `verified_kinds`: kinds NOT to check -- those a previous process already verified for this pair of code objects
   (sunode_amd keeps that verdict in a file next to the code object) and those the caller will never run on this
   handle (an AdjointSolver never calls sa_solve_batch).
   Costs two extra launches of <= n_sample instances per kind, once; synchronises the handle's stream while it runs. */
#define SA_GUARD_PLAIN 1
#define SA_GUARD_ADJOINT 2
#define SA_GUARD_SENS 4
#define SA_GUARD_RECHECK_OPEN 0x80000000u   /* sa_guard_state, `pending`: a verified kind may still be re-checked */
int sa_solver_attach_guard(sa_solver *s, const char *safe_code_object_path, int32_t n_sample /* 0: 64 */,
                           uint32_t verified_kinds);
/* Any pointer may be NULL.  verified / differs: bit masks of SA_GUARD_*; n_sample[3]: instances of the largest check
   per kind (plain, adjoint, sens); detail: text of the first difference ("" if none), valid until the next call on
   the handle. */
int sa_guard_state(sa_solver *s, uint32_t *pending, uint32_t *verified, uint32_t *differs, int32_t *using_safe,
                   int32_t *n_sample, const char **detail);

/* Solver.solve without sensitivities (solver.py:467-527): CVodeReInit + CVode(CV_NORMAL)
   per tval with <= max_retries_fwd CV_TOO_MUCH_WORK retries.
   y0 [B][n], ps [B][p], pr [B][r] (rem_stride = r) or [r] (rem_stride = 0), tvals [n_t],
   y_out [B][n_t][n], status [B] int32, stats [B][SA_N_STATS] int64. */
int sa_solve_batch(sa_solver *s, int mem, int32_t B, const double *y0, const double *ps,
                   const double *pr, int32_t rem_stride, double t0, const double *tvals, int32_t n_t,
                   double *y_out, int32_t *status, int64_t *stats);

/* Solver(sens_mode=...).solve, batched (replaces solver.py:360-392 CVodeSensInit /
   CVodeSensEEtolerances / CVodeSetSensErrCon(1) and :467-527 CVodeSensReInit / CVode / CVodeGetSens).
   Needs a code object built with forward-sensitivity support (SA_SENS build); otherwise SA_ERR_ARG.
   ism: 0 = "simultaneous", 1 = "staggered".  scaling [n_sub] = CVodeSetSensParams pbar (NULL: ones).
   sens0 [B][n_sub][n_states]; sens_out [B][n_t][n_sub][n_states] (the reference's sens_out[i, j, :] per
   instance).  stats slots SA_ST_NFQE / NETFQ / NINTERP / NREBUILD carry nfSe / netfS / nniS / ncfnS here. */
int sa_solve_sens_batch(sa_solver *s, int mem, int ism, const double *scaling, int32_t B, const double *y0,
                        const double *params_sub, const double *params_rem, int32_t rem_stride,
                        const double *sens0, double t0, const double *tvals, int32_t n_t, double *y_out,
                        double *sens_out, int32_t *status, int64_t *stats);

/* AdjointSolver.solve_forward (solver.py:682-721): CVodeReInit + CVodeAdjReInit + CVodeF per
   tval; every internal step is stored in the solver's trajectory arena. */
int sa_solve_forward_batch(sa_solver *s, int mem, int32_t B, const double *y0, const double *ps,
                           const double *pr, int32_t rem_stride, double t0, const double *tvals,
                           int32_t n_t, double *y_out, int32_t *status, int64_t *stats);

/* AdjointSolver.solve_backward (solver.py:723-784) for the batch of the preceding
   sa_solve_forward_batch: per observation interval CVodeReInitB/CVodeQuadReInitB/CVodeB/
   CVodeGetB/CVodeGetQuadB, lamda -= grads[k] jumps.  t0 = final time, tend = initial time
   (sunode's argument naming).  grads [B][n_t][n] (grads_stride = n_t*n) or [n_t][n]
   (grads_stride = 0); grad_out [B][p] = dL/dp; lamda_out [B][n] = -dL/dy0. */
int sa_solve_backward_batch(sa_solver *s, int mem, int32_t B, const double *ps, const double *pr,
                            int32_t rem_stride, double t0, double tend, const double *tvals,
                            int32_t n_t, const double *grads, int64_t grads_stride,
                            double *grad_out, double *lamda_out, int32_t *status, int64_t *stats);

/* Same, with the optional per-output-time results of solve_backward (solver.py:778-781): the adjoint state and
   the accumulated quadrature right after every jump.  lamda_all_out [B][n_t][n], quad_all_out [B][n_t][p],
   either may be NULL; rows follow the reference's indexing lamda_all_out[-i] (i-th jump counted from the last
   output time: rows 0, n_t-1, n_t-2, ..., 1). */
int sa_solve_backward_batch_all(sa_solver *s, int mem, int32_t B, const double *params_sub,
                                const double *params_rem, int32_t rem_stride, double t0, double tend,
                                const double *tvals, int32_t n_t, const double *grads, int64_t grads_stride,
                                double *grad_out, double *lamda_out, double *lamda_all_out,
                                double *quad_all_out, int32_t *status, int64_t *stats);

/* Evaluate the generated callbacks on the device (what make_sundials_rhs / _jac_dense /
   _adjoint_rhs / _adjoint_quad_rhs / _adjoint_jac_dense compute, problem.py:156-383).
   t [npts], y/lam [npts][n], ps [npts][p], pr [npts][r]; jac/adjjac column-major n*n;
   codes [npts][5] callback return codes (1 = non-finite output). */
int sa_eval_callbacks(sa_solver *s, int mem, int32_t npts, const double *t, const double *y,
                      const double *lam, const double *ps, const double *pr, double *rhs, double *jac,
                      double *adj, double *quad, double *adjjac, int32_t *codes);

/* Device arithmetic probe used by the parity tests (deterministic pow, sqrt, divide). Host arrays. */
int sa_math_probe(sa_solver *s, int32_t n, const double *x, const double *y, double *pow_out,
                  double *sqrt_out, double *div_out);

/* Trajectory arena of the last sa_solve_forward_batch / sa_solve_backward_batch pair: bytes of the largest arena
   allocation used, number of re-integrated tiles so far on this handle, and whether the last forward batch is
   resident (0) or will be / was re-integrated tile by tile (1).  Any pointer may be NULL. */
int sa_arena_info(sa_solver *s, int64_t *arena_bytes, int64_t *tiles, int32_t *tiled);

/* HIP-event durations (ms) of the most recent forward / backward kernel launches (tiled batches: the whole
   sequence of re-integration + adjoint launches of the backward call). */
int sa_last_kernel_ms(sa_solver *s, float *forward_ms, float *backward_ms);
/* Stream ordering contract.  A handle launches on ONE stream: its own (created hipStreamNonBlocking, i.e. NOT
   ordered against the null stream or any caller stream) unless sa_set_stream() hands it the caller's.  With
   SA_MEM_DEVICE arguments the caller must therefore either (a) pass the stream its producers / consumers run on
   (e.g. torch.cuda.current_stream().cuda_stream, when that is not the null stream), or (b) synchronise its own
   stream before the call and call sa_synchronize() before touching the outputs.  SA_MEM_HOST calls return with
   the results on the host and need neither. */
int sa_set_stream(sa_solver *s, void *hip_stream);     /* hipStream_t; NULL = back to a library-owned stream */
int sa_synchronize(sa_solver *s);

#ifdef __cplusplus
}
#endif
#endif
