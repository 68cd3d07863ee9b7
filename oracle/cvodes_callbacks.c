/* CVODES-ABI callbacks around the generated problem header (rhs / jac / rhsB / quadB / jacB), for the optional
 * real-CVODES leg of bench.py's cpu_baseline (oracle/cvodes_driver.py).  TEST / BASELINE INFRASTRUCTURE, never linked
 * into the product.  Signatures: /root/reference/include/cvodes/16_cvodes.h:112-113,145-146,152-153 and
 * 15_cvodes_ls.h:49-51,117-120; what the reference generates with numba (problem.py:156-383) is generated here as C.
 * Only two SUNDIALS accessors are used, through pointers the driver fills in (no SUNDIALS headers needed to build). */
#include <math.h>
#include <stdint.h>
#define SA_FN static inline
#include SA_PROBLEM_HEADER

typedef void *N_Vector;
typedef void *SUNMatrix;
double *(*sa_nv_data)(N_Vector);          /* N_VGetArrayPointer */
double *(*sa_dm_data)(SUNMatrix);         /* SUNDenseMatrix_Data (column-major, matrix.py:251-253) */
typedef struct { const double *ps, *pr; } sa_user_data;   /* the reference's user_data record starts with params too */

int sa_cv_rhs(double t, N_Vector y, N_Vector ydot, void *ud)
{ const sa_user_data *u = ud; return sa_rhs(t, sa_nv_data(y), u->ps, u->pr, sa_nv_data(ydot)); }
int sa_cv_jac(double t, N_Vector y, N_Vector fy, SUNMatrix J, void *ud, N_Vector t1, N_Vector t2, N_Vector t3)
{ const sa_user_data *u = ud; (void)fy; (void)t1; (void)t2; (void)t3; return sa_jac(t, sa_nv_data(y), u->ps, u->pr, sa_dm_data(J)); }
int sa_cv_rhsB(double t, N_Vector y, N_Vector yB, N_Vector yBdot, void *ud)
{ const sa_user_data *u = ud; return sa_adj_rhs(t, sa_nv_data(y), sa_nv_data(yB), u->ps, u->pr, sa_nv_data(yBdot)); }
int sa_cv_quadB(double t, N_Vector y, N_Vector yB, N_Vector qBdot, void *ud)
{ const sa_user_data *u = ud; return sa_quad_rhs(t, sa_nv_data(y), sa_nv_data(yB), u->ps, u->pr, sa_nv_data(qBdot)); }
int sa_cv_jacB(double t, N_Vector y, N_Vector yB, N_Vector fyB, SUNMatrix JB, void *ud, N_Vector t1, N_Vector t2, N_Vector t3)
{ const sa_user_data *u = ud; (void)yB; (void)fyB; (void)t1; (void)t2; (void)t3; return sa_adj_jac(t, sa_nv_data(y), u->ps, u->pr, sa_dm_data(JB)); }
