/*
 * Kernel-argument blocks shared by the host library (sunode_amd.cpp) and the per-problem
 * device code (bdf_kernels.hip).  Plain C layout; passed by value as the single kernarg.
 *
 * HBM layouts (all fp64 unless noted):
 *   y0      [B][n]            row per instance (what AdjointSolver.solve_forward receives, with
 *                             a leading batch axis; reference solver.py:682)
 *   ps      [B][p]            differentiated parameters, subset_paths order
 *   pr      [B][r] or [r]     remaining parameters (rem_stride = r or 0)
 *   tvals   [n_t]             shared output grid
 *   y_out   [B][n_t][n]
 *   grads   [B][n_t][n] or [n_t][n] (grads_stride = n_t*n or 0)
 *   stats   [B][16] int64     counters, see SA_ST_* in include/sunode_amd.h
 *   trajectory arena (the CVODES "data points" of CVodeAdjInit / CVodeF):
 *     traj [cap][stride][8+6n] records {order, dt, T[6], Y[6][n]} = the divided-difference table
 *     CVApolynomialGetY needs at that index, built once by the forward kernel; traj_np [B] (i32);
 *     instance index fastest: the lanes of a wave write one step as one contiguous block.
 *   ws      [W][ws_stride]    integrator workspace of the memory-resident kernels (bdf_mem.hip):
 *                             element e of instance i at ws[e*ws_stride + i]; W = sa_meta[5]
 */
#ifndef SA_DEVICE_ABI_H
#define SA_DEVICE_ABI_H

#include <stdint.h>

#define SA_MODE_PLAIN 0      /* Solver.solve: CVode(NORMAL) per tval, mxstep x max_retries budget */
#define SA_MODE_ADJ_FWD 1    /* AdjointSolver.solve_forward: CVodeF, every step stored */
#define SA_MODE_ADJ_COUNT 2  /* the same pass without arena writes: y_out / status / stats identical, traj_np = number of
                                points the instance WOULD store (sizes the arena tiles of the re-integration, see
                                sunode_amd.cpp "trajectory arena") */

/* status of an instance that would store more than traj_max (= sa_options.traj_capacity) points: the integration
   stops there (no unbounded pass, whatever the mode) and the outputs are NaN like for any other failure; the value is
   SA_STATUS_ARENA_FULL of include/sunode_amd.h.  An instance that merely outgrows the rows the arena was LAUNCHED
   with (traj_cap) is not a failure: it keeps integrating without arena writes, reports its full point count in
   traj_np and raises *overflow (atomic max of such counts); the host library sees that at the start of the backward
   call and re-integrates with exactly sized tiles (sunode_amd.cpp "trajectory arena"). */
#define SA_TRAJ_FULL (-9001)

#define SA_N_STATS 16

/* sa_meta[3] of a code object; sa_solver_create() rejects any other value (= SA_ABI_VERSION of sunode_amd.h).
   2: arena records instance-major (traj_istride), SA_MODE_ADJ_COUNT, traj_max / overflow in sa_fwd_args
   3: the arena layout follows the kernel family -- [instance][point] unless the code object exports
      sa_traj_point_major (the memory-resident kernel) -- instead of ws_doubles == 0 */
#define SA_DEVICE_ABI_VERSION 3

typedef struct {
    int32_t B, n_t, mode, mxstep, max_retries, traj_cap, rem_stride, traj_istride;
    int64_t traj_stride;      /* record (instance i, point s) at traj + (i * traj_istride + s * traj_stride) * record size */
    int32_t traj_max;         /* store modes: most points an instance may produce (SA_TRAJ_FULL beyond) */
    int32_t reserved0;
    int32_t *overflow;        /* store modes: atomic max of the point counts of instances with more than traj_cap points */
    double t0, rtol;
    const double *atol;
    const double *y0, *ps, *pr, *tvals;
    double *y_out;
    int32_t *status;
    int64_t *stats;
    double *traj;
    int32_t *traj_np;
    double *ws;               /* memory-resident kernels only: [sa_meta[5]][ws_stride] doubles */
    int64_t ws_stride;
    const double *constraints; /* [n] CVodeSetConstraints vector or NULL (read by SA_CONSTRAINTS builds only) */
} sa_fwd_args;

typedef struct {
    int32_t B, n_t, mxstep, max_retries, traj_cap, rem_stride, traj_istride, reserved1;
    int64_t traj_stride, grads_stride;
    double t0, tend, tinitial;
    double rtolB, atolB, rtolQB, atolQB;
    const double *ps, *pr, *tvals, *grads;
    double *grad_out, *lamda_out;
    int32_t *status;
    const int32_t *fwd_status;
    int64_t *stats;
    const double *traj;
    const int32_t *traj_np;
    double *ws;
    int64_t ws_stride;
    /* optional (NULL: not wanted): adjoint state / accumulated quadrature right after every jump, laid out
       [B][n_t][n] / [B][n_t][p] with the reference's row order (solver.py:778-781 writes row -i for the
       i-th jump counted from the last output time, i.e. row 0, n_t-1, n_t-2, ..., 1) */
    double *lamda_all, *quad_all;
} sa_bwd_args;

/* Solver(sens_mode=...).solve: forward solve + forward sensitivities (SA_SENS build of bdf_mem.hip).
   sens0 [B][p][n], sens_out [B][n_t][p][n] (row `is` = derivative w.r.t. differentiated parameter `is`,
   as the reference's sens_out[i, j, :], solver.py:527); ism 0 = simultaneous, 1 = staggered corrector;
   pbar [p] = |scaling_factors| (ones by default). */
typedef struct {
    int32_t B, n_t, ism, mxstep, max_retries, rem_stride, reserved0, reserved1;
    double t0, rtol;
    const double *atol, *pbar;
    const double *y0, *ps, *pr, *sens0, *tvals;
    double *y_out, *sens_out;
    int32_t *status;
    int64_t *stats;
    double *ws;
    int64_t ws_stride;
} sa_sens_args;

typedef struct {
    int32_t npts, reserved;
    const double *t, *y, *lam, *ps, *pr;      /* [npts], [npts][n], [npts][n], [npts][p], [npts][r] */
    double *rhs, *jac, *adj, *quad, *adjjac;  /* [npts][n], [npts][n*n] col-major, ... */
    int32_t *codes;                           /* [npts][5] */
} sa_eval_args;

typedef struct {
    int32_t n, reserved;
    const double *x, *y;
    double *pow_out, *sqrt_out, *div_out;
} sa_math_args;

#endif
