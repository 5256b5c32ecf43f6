/*
 * ref_krylov_harness.c -- drives the reference's OWN Krylov kernels, compiled unmodified.
 *
 * TEST INFRASTRUCTURE ONLY (see the header of pysparse_oracle.c).  `make -C oracle ref`
 * compiles /root/reference/pysparse/itsolvers/src/{pcg,minres,cgs,bicgstab,qmrs,gmres}.c from
 * where they lie -- not one line of them is copied or edited -- together with this file into
 * oracle/_ref/libref_krylov.so.  Those six kernels need nothing of CPython at link time: they
 * reach the operator and the preconditioner only through the reference's inter-module table
 *     void **itsolvers_spmatrix          (pysparse/include/spmatrix_api.h:10-72,125-131;
 *                                         defined by itsolversmodule.c:25, which is not built)
 * slot 7 = SpMatrix_Matvec(PyObject*, int nx, double *x, int ny, double *y) -> int
 * slot 8 = SpMatrix_Precon(PyObject*, int n, double *x, double *y) -> int
 * (spmatrix_api.h:34-40), and forward their `PyObject *mat_obj / prec_obj` arguments to those
 * slots untouched.  This file defines that table and fills the two slots with trampolines to a
 * C callback + context (the protocol of pysparse_oracle.c: orc_csr_matvec_cb, orc_sss_matvec_cb,
 * orc_jacobi_apply, orc_ssor_apply).  BLAS-1 is the image's real OpenBLAS (scipy wheel), as for
 * libref_pcg.so.  No header, library or generated file of the reference is replaced.
 *
 * What it is for: oracle/make_golden.py generates tests/golden/ref_krylov.json from it, and
 * tests/test_oracle_golden.py checks every solver restatement of pysparse_oracle.c against it
 * live (when oracle/_ref exists) and against the committed goldens (always).
 */
#include <math.h>
#include <stddef.h>

typedef int (*refk_fn)(void *ctx, int n, const double *x, double *y);

typedef struct {
  refk_fn fn;
  void *ctx;
  int fail_after; /* >= 0: the (fail_after+1)-th application returns -1 (callback raised) */
  int calls;
} refk_obj;

/* the prototypes of the kernels as the reference declares them (pysparse/include/pcg.h, minres.h,
 * cgs.h, bicgstab.h, qmrs.h, gmres.h), with PyObject* spelled void*: the pointers are opaque here */
int Itsolvers_pcg_kernel(int n, double *x, double *b, double tol, int maxit, int clvl, int *iter,
                         double *relres, int *flag, double *work, void *mat_obj, void *prec_obj);
int Itsolvers_minres_kernel(int n, double errtol, int it_max, int *it, double *nrm_res, int clvl,
                            double *x, double *b, double *work, void *mat_obj, void *prec_obj);
int Itsolvers_cgs_kernel(int n, double *b, double *x, int maxit, double tol, double *work, int *iter,
                         double *res, void *mat_obj, void *prec_obj);
int Itsolvers_bicgstab_kernel(int n, double *x, double *b, double tol, int maxit, int clvl, int *iter,
                              double *relres, int *flag, double *work, void *mat_obj, void *prec_obj);
int Itsolvers_qmrs_kernel(int n, double *b, double *x, double *work, double tol, int maxitera,
                          int *itera, double *err, void *mat_obj, void *prec_obj);
int Itsolvers_gmres_kernel(int n, double errtol, int it_max, int *it, double *relres, int dim,
                           double *x, double *b, double *work, void *mat_obj, void *prec_obj);

static int apply(refk_obj *o, int n, const double *x, double *y) {
  if (o->fail_after >= 0 && o->calls++ >= o->fail_after)
    return -1;
  return o->fn(o->ctx, n, x, y);
}

static int tramp_matvec(void *obj, int nx, double *x, int ny, double *y) {
  (void)ny;
  return apply((refk_obj *)obj, nx, x, y);
}

static int tramp_precon(void *obj, int n, double *x, double *y) {
  return apply((refk_obj *)obj, n, x, y);
}

static void *refk_table[16] = {
    0, 0, 0, 0, 0, 0, 0, (void *)tramp_matvec, (void *)tramp_precon, 0, 0, 0, 0, 0, 0, 0};

/* the symbol the six kernels are compiled against (SPMATRIX_UNIQUE_SYMBOL, e.g. minres.c:34) */
__attribute__((visibility("default"))) void **itsolvers_spmatrix = refk_table;

/* solver: 0 pcg, 1 minres, 2 cgs, 3 bicgstab, 4 qmrs, 5 gmres.  work: 8n doubles.
 * *info is what the module's wrapper returns as `info` (itsolversmodule.c:93-117, :185-209,
 * :281-304, :376-399, :470-492, :563-585): pcg / bicgstab hand back the kernel's flag argument,
 * the others the kernel's return value.  The kernel's return value is returned as is.
 * *relres is left as the caller set it when the kernel does not write it (minres on -3 / -6). */
__attribute__((visibility("default"))) int refk_solve(int solver, int n, double *x, double *b, double tol,
                                                      int maxit, int dim, int *iter, double *relres,
                                                      int *info, double *work, refk_fn mv, void *mctx,
                                                      refk_fn pc, void *pctx, int mv_fail_after,
                                                      int pc_fail_after) {
  refk_obj A = {mv, mctx, mv_fail_after, 0};
  refk_obj K = {pc, pctx, pc_fail_after, 0};
  void *Kp = pc ? (void *)&K : NULL;
  int rc;
  switch (solver) {
  case 0:
    rc = Itsolvers_pcg_kernel(n, x, b, tol, maxit, 0, iter, relres, info, work, &A, Kp);
    return rc;
  case 1:
    rc = Itsolvers_minres_kernel(n, tol, maxit, iter, relres, 0, x, b, work, &A, Kp);
    break;
  case 2:
    rc = Itsolvers_cgs_kernel(n, b, x, maxit, tol, work, iter, relres, &A, Kp);
    break;
  case 3:
    rc = Itsolvers_bicgstab_kernel(n, x, b, tol, maxit, 0, iter, relres, info, work, &A, Kp);
    return rc;
  case 4:
    rc = Itsolvers_qmrs_kernel(n, b, x, work, tol, maxit, iter, relres, &A, Kp);
    break;
  case 5:
    rc = Itsolvers_gmres_kernel(n, tol, maxit, iter, relres, dim, x, b, work, &A, Kp);
    break;
  default:
    return -100;
  }
  *info = rc;
  return rc;
}
