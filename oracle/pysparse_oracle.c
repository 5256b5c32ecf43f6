/*
 * pysparse_oracle.c -- CPU restatement of PySparse's SpMV + Krylov hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP product in
 * pysparse_amd/: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it.  Nothing under pysparse_amd/
 * links, imports or executes anything from oracle/.
 *
 * Parity status: PINNED.  Every solver routine below is checked (tests/
 * test_oracle_*.py, oracle/make_golden.py) against
 *   (1) the reference's own six Krylov kernels, pysparse/itsolvers/src/{pcg,minres,cgs,
 *       bicgstab,qmrs,gmres}.c, compiled unmodified into oracle/_ref/libref_krylov.so
 *       (oracle/Makefile + oracle/ref_krylov_harness.c, which supplies the reference's
 *       itsolvers_spmatrix callback table), and the standalone examples/poisson_test/
 *       pcg.c built the same way into oracle/_ref/libref_pcg.so, and
 *   (2) the golden vectors committed under tests/golden/ that those builds produced
 *       (ref_krylov.json: 104 cases over the six solvers; ref_pcg.json: G1..G5 of
 *       BASELINE.md) plus the ten-digit known answer K1.
 * The matrix-vector products and the preconditioners (csr_mat.c, sss_mat.c,
 * preconmodule.c) do not compile in this image (numpy/noprefix.h, Python-2 C API):
 * they are restated here from the cited lines and are what the compiled kernels are
 * driven with.
 *
 * Each function cites the reference file:line (relative to /root/reference) whose
 * arithmetic -- operation order included -- it restates.  Plain C, single thread,
 * no FMA contraction (build with -ffp-contract=off): that is how the reference
 * itself is built (gcc -O2, x86-64 baseline has no FMA), and it is what makes the
 * HIP SpMV bit-comparable with this file.
 *
 * Third-party arithmetic: the reference calls Fortran BLAS-1 (dnrm2, ddot, daxpy,
 * dcopy -- pysparse/include/blas.h:97-118) from whatever system BLAS the build
 * found; no version is pinned.  The routines orc_ddot/orc_dnrm2/orc_daxpy/orc_dcopy
 * restate the published netlib reference BLAS (3.8 and earlier) algorithms:
 * ddot/daxpy are sequential left-to-right loops (netlib's 5-/4-way unrolling is
 * written as one left-associated expression, i.e. the same order), dnrm2 is the
 * classic scale/ssq one-pass form.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ BLAS-1 */

/* netlib ddot, incx = incy = 1: dtemp accumulated left to right. */
ORC_API double orc_ddot(int n, const double *x, const double *y) {
  double s = 0.0;
  int i;
  for (i = 0; i < n; i++)
    s = s + x[i] * y[i];
  return s;
}

/* netlib dnrm2 (reference BLAS <= 3.8): scale / sum-of-squares recurrence. */
ORC_API double orc_dnrm2(int n, const double *x) {
  double scale = 0.0, ssq = 1.0, absxi, t;
  int i;
  if (n < 1)
    return 0.0;
  if (n == 1)
    return fabs(x[0]);
  for (i = 0; i < n; i++) {
    if (x[i] != 0.0) {
      absxi = fabs(x[i]);
      if (scale < absxi) {
        t = scale / absxi;
        ssq = 1.0 + ssq * t * t;
        scale = absxi;
      } else {
        t = absxi / scale;
        ssq = ssq + t * t;
      }
    }
  }
  return scale * sqrt(ssq);
}

/* netlib daxpy: y := y + a*x (returns early for a == 0, as netlib does). */
ORC_API void orc_daxpy(int n, double a, const double *x, double *y) {
  int i;
  if (a == 0.0)
    return;
  for (i = 0; i < n; i++)
    y[i] = y[i] + a * x[i];
}

ORC_API void orc_dcopy(int n, const double *x, double *y) {
  memcpy(y, x, (size_t)n * sizeof(double));
}

/* --------------------------------------------------------------- CSR SpMV */

/* pysparse/sparse/src/csr_mat.c:49-54 (live branch, UNROLL_LOOPS == 0 at :11):
 * y[i] = sum_k va[k]*x[ja[k]], accumulated left to right from 0.0, y overwritten. */
ORC_API void orc_csr_matvec(int m, const double *x, double *y,
                            const double *va, const int *ja, const int *ia) {
  double s;
  int i, k;
  for (i = 0; i < m; i++) {
    s = 0.0;
    for (k = ia[i]; k < ia[i + 1]; k++)
      s += va[k] * x[ja[k]];
    y[i] = s;
  }
}

/* NOT the reference: the reference's product is single-threaded (no OpenMP, GIL held).  The same rows, each summed
 * exactly as above, handed to `nthreads` POSIX threads in contiguous ranges of about equal entry counts -- every y[i] has
 * the bits of orc_csr_matvec.  bench.py times it as the labelled "all host cores" line beside the one-core baseline. */
typedef struct {
  int r0, r1;
  const double *x, *va;
  double *y;
  const int *ja, *ia;
} orc_mv_range;

static void *orc_csr_matvec_range(void *arg) {
  const orc_mv_range *q = (const orc_mv_range *)arg;
  double s;
  int i, k;
  for (i = q->r0; i < q->r1; i++) {
    s = 0.0;
    for (k = q->ia[i]; k < q->ia[i + 1]; k++)
      s += q->va[k] * q->x[q->ja[k]];
    q->y[i] = s;
  }
  return NULL;
}

ORC_API int orc_csr_matvec_threads(int m, const double *x, double *y, const double *va, const int *ja,
                                   const int *ia, int nthreads) {
  enum { MAXT = 256 };
  pthread_t th[MAXT];
  orc_mv_range rg[MAXT];
  int started[MAXT];
  int t, r = 0, used;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > MAXT) nthreads = MAXT;
  if (nthreads > m) nthreads = m > 0 ? m : 1;
  for (t = 0; t < nthreads; t++) { /* range t ends at the first row whose offset reaches (t+1)/nthreads of the entries */
    const long want = (long)((double)ia[m] * (t + 1) / nthreads);
    int lo = r, hi = m;
    if (t == nthreads - 1)
      lo = m;
    else
      while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (ia[mid] < want) lo = mid + 1; else hi = mid;
      }
    rg[t].r0 = r; rg[t].r1 = lo; rg[t].x = x; rg[t].y = y; rg[t].va = va; rg[t].ja = ja; rg[t].ia = ia;
    r = lo;
  }
  used = 0;
  for (t = 1; t < nthreads; t++) {
    started[t] = pthread_create(&th[t], NULL, orc_csr_matvec_range, &rg[t]) == 0;
    if (!started[t]) orc_csr_matvec_range(&rg[t]); /* no thread to be had: this range inline */
    used += started[t];
  }
  orc_csr_matvec_range(&rg[0]);
  for (t = 1; t < nthreads; t++)
    if (started[t]) pthread_join(th[t], NULL);
  return used + 1;
}

/* csr_mat.c:58-72: element strides for non-contiguous NumPy views. */
ORC_API void orc_csr_matvec_stride(int m, const double *x, int incx, double *y, int incy,
                                   const double *va, const int *ja, const int *ia) {
  double s;
  int i, k;
  for (i = 0; i < m; i++) {
    s = 0.0;
    for (k = ia[i]; k < ia[i + 1]; k++)
      s += va[k] * x[(long)ja[k] * incx];
    y[(long)i * incy] = s;
  }
}

/* csr_mat.c:74-88: y = A^T x; zero y (length n = number of columns), then row-wise scatter. */
ORC_API void orc_csr_matvec_transp(int m, int n, const double *x, double *y,
                                   const double *va, const int *ja, const int *ia) {
  double xi;
  int i, k;
  for (i = 0; i < n; i++)
    y[i] = 0.0;
  for (i = 0; i < m; i++) {
    xi = x[i];
    for (k = ia[i]; k < ia[i + 1]; k++)
      y[ja[k]] += va[k] * xi;
  }
}

/* --------------------------------------------------------------- SSS SpMV */

/* pysparse/sparse/src/sss_mat.c:40-56: one ascending sweep; row i gathers its strict
 * lower entries, scatters the mirrored contribution into y[j] (j < i, already
 * assigned) and then ASSIGNS y[i] = s + diag[i]*x[i]. */
ORC_API void orc_sss_matvec(int n, const double *x, double *y, const double *va,
                            const double *da, const int *ja, const int *ia) {
  double s, v, xi;
  int i, j, k;
  for (i = 0; i < n; i++) {
    xi = x[i];
    s = 0.0;
    for (k = ia[i]; k < ia[i + 1]; k++) {
      j = ja[k];
      v = va[k];
      s += v * x[j];
      y[j] += v * xi;
    }
    y[i] = s + da[i] * xi;
  }
}

/* sss_mat.c:58-76 */
ORC_API void orc_sss_matvec_stride(int n, const double *x, int incx, double *y, int incy,
                                   const double *va, const double *da, const int *ja,
                                   const int *ia) {
  double s, v, xi;
  int i, j, k;
  for (i = 0; i < n; i++) {
    xi = x[(long)i * incx];
    s = 0.0;
    for (k = ia[i]; k < ia[i + 1]; k++) {
      j = ja[k];
      v = va[k];
      s += v * x[(long)j * incx];
      y[(long)j * incy] += v * xi;
    }
    y[(long)i * incy] = s + da[i] * xi;
  }
}

/* sss_mat.c:14-28: A[i,j] lookup the way getitem() intends it. */
ORC_API double orc_sss_getitem(int i, int j, const double *va, const double *da,
                               const int *ja, const int *ia) {
  int k, t;
  if (i == j)
    return da[i];
  if (i < j) {
    t = i;
    i = j;
    j = t;
  }
  for (k = ia[i]; k < ia[i + 1]; k++)
    if (ja[k] == j)
      return va[k];
  return 0.0;
}

/* ------------------------------------------------ operator callback protocol */

/* The reference reaches A and K through SpMatrix_Matvec / SpMatrix_Precon
 * (pysparse/sparse/src/spmatrixmodule.c:169-248).  Here: C callbacks + context. */
typedef int (*orc_matvec_fn)(void *ctx, int n, const double *x, double *y);
typedef int (*orc_precon_fn)(void *ctx, int n, const double *x, double *y);

typedef struct {
  int m, n;
  const double *va;
  const int *ja, *ia;
} orc_csr_t;

typedef struct {
  int n;
  const double *va, *da;
  const int *ja, *ia;
} orc_sss_t;

ORC_API int orc_csr_matvec_cb(void *ctx, int n, const double *x, double *y) {
  const orc_csr_t *A = (const orc_csr_t *)ctx;
  (void)n;
  orc_csr_matvec(A->m, x, y, A->va, A->ja, A->ia);
  return 0;
}

ORC_API int orc_sss_matvec_cb(void *ctx, int n, const double *x, double *y) {
  const orc_sss_t *A = (const orc_sss_t *)ctx;
  (void)n;
  orc_sss_matvec(A->n, x, y, A->va, A->da, A->ja, A->ia);
  return 0;
}

/* NOT the reference either: the same callback protocol with the rows handed to POSIX threads (orc_csr_matvec_threads:
 * every y[i] keeps the bits of orc_csr_matvec).  It exists so that the reference's COMPILED kernels (oracle/_ref) can be
 * driven through 20 iterations at BASELINE's 512^3 inside a test's time budget (tests/test_gpu_reference_sizes.py): the
 * kernel is the reference's own code, only its operator callback is row-parallel. */
typedef struct {
  orc_csr_t A;
  int nthreads;
} orc_csr_mt_t;

ORC_API int orc_csr_matvec_threads_cb(void *ctx, int n, const double *x, double *y) {
  const orc_csr_mt_t *M = (const orc_csr_mt_t *)ctx;
  (void)n;
  orc_csr_matvec_threads(M->A.m, x, y, M->A.va, M->A.ja, M->A.ia, M->nthreads);
  return 0;
}

/* y = x .* dinv (preconmodule.c:41-42, jacobi with steps == 1) over thread ranges: elementwise, the same bits */
typedef struct {
  int n;
  const double *dinv;
  int nthreads;
} orc_jacobi_mt_t;

typedef struct {
  int r0, r1;
  const double *x, *dinv;
  double *y;
} orc_jac_range;

static void *orc_jacobi_range(void *arg) {
  const orc_jac_range *q = (const orc_jac_range *)arg;
  int i;
  for (i = q->r0; i < q->r1; i++)
    q->y[i] = q->x[i] * q->dinv[i];
  return NULL;
}

ORC_API int orc_jacobi_threads_cb(void *ctx, int n, const double *x, double *y) {
  enum { MAXT = 256 };
  const orc_jacobi_mt_t *K = (const orc_jacobi_mt_t *)ctx;
  pthread_t th[MAXT];
  orc_jac_range rg[MAXT];
  int started[MAXT];
  int t, nt = K->nthreads;
  if (nt < 1) nt = 1;
  if (nt > MAXT) nt = MAXT;
  if (nt > n) nt = n > 0 ? n : 1;
  for (t = 0; t < nt; t++) {
    rg[t].r0 = (int)((long)n * t / nt);
    rg[t].r1 = (int)((long)n * (t + 1) / nt);
    rg[t].x = x; rg[t].dinv = K->dinv; rg[t].y = y;
  }
  for (t = 1; t < nt; t++) {
    started[t] = pthread_create(&th[t], NULL, orc_jacobi_range, &rg[t]) == 0;
    if (!started[t]) orc_jacobi_range(&rg[t]);
  }
  orc_jacobi_range(&rg[0]);
  for (t = 1; t < nt; t++)
    if (started[t]) pthread_join(th[t], NULL);
  return 0;
}

/* ------------------------------------------------------------------ Jacobi */

/* pysparse/precon/src/preconmodule.c:389-401: dinv[i] = omega / A[i,i]; a diagonal
 * entry d with 1.0 + d == 1.0 is "close to zero" -> error (returns the row + 1). */
ORC_API int orc_jacobi_setup(int n, const double *diag, double omega, double *dinv) {
  int i;
  double d;
  for (i = 0; i < n; i++) {
    d = diag[i];
    if (1.0 + d == 1.0)
      return i + 1;
    dinv[i] = omega / d;
  }
  return 0;
}

typedef struct {
  int n;
  const double *dinv;
  int steps;
  double *temp; /* n doubles, needed only when steps > 1 */
  orc_matvec_fn matvec;
  void *mctx;
} orc_jacobi_t;

/* preconmodule.c:35-54: y = x .* dinv, then (steps-1) sweeps
 * temp = y; y = A*temp; y = (x - y) .* dinv + temp. */
ORC_API int orc_jacobi_apply(void *ctx, int n, const double *x, double *y) {
  const orc_jacobi_t *K = (const orc_jacobi_t *)ctx;
  int i, step;
  for (i = 0; i < n; i++)
    y[i] = x[i] * K->dinv[i];
  for (step = 1; step < K->steps; step++) {
    orc_dcopy(n, y, K->temp);
    if (K->matvec(K->mctx, n, K->temp, y))
      return -1;
    for (i = 0; i < n; i++)
      y[i] = (x[i] - y[i]) * K->dinv[i] + K->temp[i];
  }
  return 0;
}

/* -------------------------------------------------------------------- SSOR */

/* pysparse/precon/src/preconmodule.c:149-193 (symgs_kernel, omega == 1): `steps` symmetric
 * Gauss-Seidel steps with zero initial guess on the SSS arrays; x is the output, y an n-vector
 * of work space.  Parity status of the two SSOR routines: UNPINNED at the bit level (no
 * compilable reference and no golden vector for them exists); tests pin them by identity
 * against dense triangular solves (tests/test_oracle_golden.py). */
ORC_API void orc_symgs(int n, const double *b, double *x, double *y, const double *va,
                       const double *da, const int *ja, const int *ia, int steps) {
  double s;
  int step, i, j, k;
  for (k = 0; k < n; k++) /* :164-165 */
    y[k] = 0.0;
  for (step = 0; step < steps; step++) {
    for (i = 0; i < n; i++) { /* :171-179 */
      s = 0.0;
      for (k = ia[i]; k < ia[i + 1]; k++) {
        j = ja[k];
        s += va[k] * x[j];
      }
      x[i] = (b[i] - y[i] - s) / da[i];
      y[i] = s;
    }
    for (k = 0; k < n; k++) { /* :182-185 */
      x[k] = y[k];
      y[k] = 0.0;
    }
    for (i = n - 1; i >= 0; i--) { /* :186-193 */
      x[i] = (b[i] - x[i] - y[i]) / da[i];
      s = x[i];
      for (k = ia[i]; k < ia[i + 1]; k++) {
        j = ja[k];
        y[j] += va[k] * s;
      }
    }
  }
}

/* preconmodule.c:95-143 (ssor_kernel, omega != 1); h and temp are n-vectors of work space */
ORC_API void orc_ssor(int n, const double *b, double *x, double *h, double *temp, const double *va,
                      const double *da, const int *ja, const int *ia, double omega, int steps) {
  double s;
  int step, i, j, k;
  for (step = 0; step < steps; step++) {
    if (step == 0) /* :110-115 */
      for (i = 0; i < n; i++)
        temp[i] = omega * b[i];
    else
      for (i = 0; i < n; i++)
        temp[i] = (1.0 - omega) * x[i] * da[i] + h[i] + omega * b[i];
    for (i = 0; i < n; i++) { /* :117-125 */
      s = 0.0;
      for (k = ia[i]; k < ia[i + 1]; k++) {
        j = ja[k];
        s -= va[k] * x[j];
      }
      h[i] = omega * s;
      x[i] = (temp[i] + h[i]) / da[i];
    }
    for (i = 0; i < n; i++) { /* :128-131 */
      temp[i] = (1.0 - omega) * x[i] * da[i] + h[i] + omega * b[i];
      h[i] = 0.0;
    }
    for (i = n - 1; i >= 0; i--) { /* :132-140 */
      h[i] = omega * h[i];
      x[i] = (temp[i] + h[i]) / da[i];
      s = x[i];
      for (k = ia[i]; k < ia[i + 1]; k++) {
        j = ja[k];
        h[j] -= va[k] * s;
      }
    }
  }
}

typedef struct {
  int n;
  const double *va, *da;
  const int *ja, *ia;
  double omega;
  int steps;
  double *temp, *temp2;
} orc_ssor_t;

/* SSOR_precon, preconmodule.c:199-223 */
ORC_API int orc_ssor_apply(void *ctx, int n, const double *x, double *y) {
  const orc_ssor_t *K = (const orc_ssor_t *)ctx;
  (void)n;
  if (K->omega == 1.0)
    orc_symgs(K->n, x, y, K->temp, K->va, K->da, K->ja, K->ia, K->steps);
  else
    orc_ssor(K->n, x, y, K->temp, K->temp2, K->va, K->da, K->ja, K->ia, K->omega, K->steps);
  return 0;
}

/* --------------------------------------------------------------------- PCG */

/* pysparse/itsolvers/src/pcg.c:22-171 (Itsolvers_pcg_kernel).  work = 4n doubles
 * laid out (r, z, p, q) as at :50-53.  hist (may be NULL, length >= maxit+1) records
 * the residual norm: hist[0] = initial, hist[it] = normr after iteration it.
 * Returns 0, or -1 if a callback failed (:8-11). */
ORC_API int orc_pcg(int n, double *x, const double *b, double tol, int maxit, int *iter,
                    double *relres, int *flag, double *work, orc_matvec_fn matvec,
                    void *mctx, orc_precon_fn precon, void *pctx, double *hist) {
  double n2b, tolb, normr, alpha, beta, rho, rho1, pq, dmax, ddum;
  int stag, it, i;
  double *r = work, *z = work + n, *p = work + 2 * (long)n, *q = work + 3 * (long)n;

  n2b = orc_dnrm2(n, b); /* :57 */
  if (n2b == 0.0) {      /* :58-67: zero rhs -> zero solution, flag 0 */
    for (i = 0; i < n; i++)
      x[i] = 0.0;
    *flag = 0;
    *relres = 0.0;
    *iter = 0;
    return 0;
  }

  *flag = -1; /* :70 */
  tolb = tol * n2b;
  if (matvec(mctx, n, x, r)) /* :72 */
    return -1;
  for (i = 0; i < n; i++) /* :73-74 */
    r[i] = b[i] - r[i];
  normr = orc_dnrm2(n, r); /* :75 */
  if (hist)
    hist[0] = normr;

  if (normr <= tolb) { /* :77-84 */
    *flag = 0;
    *relres = normr / n2b;
    *iter = 0;
    return 0;
  }

  rho = 1.0;
  stag = 0;

  for (it = 1; it <= maxit; it++) { /* :91 */
    if (precon) {                   /* :93-97 */
      if (precon(pctx, n, r, z))
        return -1;
    } else {
      orc_dcopy(n, r, z);
    }

    rho1 = rho;
    rho = orc_ddot(n, r, z); /* :100 */
    if (rho == 0.0) {
      *flag = -2;
      break;
    }
    if (it == 1) {
      orc_dcopy(n, z, p); /* :106 */
    } else {
      beta = rho / rho1;
      if (beta == 0.0) {
        *flag = -6;
        break;
      }
      for (i = 0; i < n; i++) /* :113-114 */
        p[i] = z[i] + beta * p[i];
    }
    if (matvec(mctx, n, p, q)) /* :116 */
      return -1;
    pq = orc_ddot(n, p, q); /* :117 */
    if (pq == 0.0) {
      *flag = -6;
      break;
    } else {
      alpha = rho / pq;
    }
    if (alpha == 0.0)
      stag = 1;

    if (stag == 0) { /* :127-139 */
      dmax = 0.0;
      for (i = 0; i < n; i++)
        if (x[i] != 0.0) {
          ddum = fabs(alpha * p[i] / x[i]);
          if (ddum > dmax)
            dmax = ddum;
        } else if (p[i] != 0.0)
          dmax = 1.0;
      stag = (1.0 + dmax == 1.0);
    }

    orc_daxpy(n, alpha, p, x);  /* :141 */
    orc_daxpy(n, -alpha, q, r); /* :142-143 */

    normr = orc_dnrm2(n, r); /* :152 (EXPENSIVE_CRIT undefined) */
    if (hist)
      hist[it] = normr;
    if (normr <= tolb) { /* :154-157 */
      *flag = 0;
      break;
    }
    if (stag == 1) { /* :159-162 */
      *flag = -5;
      break;
    }
  }

  *iter = it; /* :165 -- maxit+1 when the loop ran out */
  *relres = normr / n2b;
  return 0;
}

/* ------------------------------------------------------------------ MINRES */

/* pysparse/itsolvers/src/minres.c:43-200 (Itsolvers_minres_kernel).  work = 7n doubles
 * (v_hat_old, v_hat, y, w, w_old, v, av) as at :54-60.  Return value is the info code
 * (0, -1, -3, -6; or -100 for a failed callback); *nrm_res is left untouched on the
 * -3 / -6 exits exactly like the reference. */
ORC_API int orc_minres(int n, double errtol, int it_max, int *it, double *nrm_res, double *x,
                       const double *b, double *work, orc_matvec_fn matvec, void *mctx,
                       orc_precon_fn precon, void *pctx, double *hist) {
  double norm_r0, beta, beta_old, c, c_old, c_oold, s, s_old, s_oold, eta, norm_rmr, alpha,
      dconst1, dconst2, r1, r1_hat, r2, r3, tmp;
  int i;
  double *v_hat_old = work, *v_hat = work + n, *y = work + 2 * (long)n,
         *w = work + 3 * (long)n, *w_old = work + 4 * (long)n, *v = work + 5 * (long)n,
         *av = work + 6 * (long)n;

  *it = 0;
  for (i = 0; i < n; i++) /* :63-65 */
    v_hat_old[i] = 0.0;
  if (matvec(mctx, n, x, v_hat)) /* :67 */
    return -100;
  for (i = 0; i < n; i++)
    v_hat[i] = b[i] - v_hat[i];
  norm_r0 = orc_dnrm2(n, v_hat); /* :71 */
  if (precon) {                  /* :73-76 */
    if (precon(pctx, n, v_hat, y))
      return -100;
  } else {
    orc_dcopy(n, v_hat, y);
  }
  beta = orc_ddot(n, v_hat, y); /* :78 */
  if (beta < 0.0)
    return -3;
  beta = sqrt(beta);
  beta_old = 1.0;

  c = 1.0;
  c_old = 1.0;
  s = 0.0;
  s_old = 0.0;
  for (i = 0; i < n; i++)
    w[i] = 0.0;
  for (i = 0; i < n; i++)
    w_old[i] = 0.0;
  eta = beta;
  norm_rmr = norm_r0;
  if (hist)
    hist[0] = norm_rmr;

  while (1) {
    if (*it >= it_max || norm_rmr < errtol * norm_r0) /* :114 -- strict < */
      break;
    *it = *it + 1;

    for (i = 0; i < n; i++) /* :123-124: v = y / beta */
      v[i] = y[i] / beta;
    orc_dcopy(n, v_hat, y);     /* :125 */
    if (matvec(mctx, n, v, av)) /* :127 */
      return -100;
    alpha = orc_ddot(n, v, av); /* :129 */
    dconst1 = alpha / beta;
    dconst2 = beta / beta_old;
    for (i = 0; i < n; i++) /* :132-133 */
      v_hat[i] = av[i] - dconst1 * v_hat[i] - dconst2 * v_hat_old[i];
    orc_dcopy(n, y, v_hat_old); /* :135 */
    if (precon) {               /* :137-140 */
      if (precon(pctx, n, v_hat, y))
        return -100;
    } else {
      orc_dcopy(n, v_hat, y);
    }
    beta_old = beta;
    beta = orc_ddot(n, v_hat, y); /* :143 */
    if (beta < 0.0)
      return -3;
    beta = sqrt(beta);

    c_oold = c_old; /* :151 */
    c_old = c;
    s_oold = s_old;
    s_old = s;

    r1_hat = c_old * alpha - c_oold * s_old * beta_old;
    r1 = sqrt(r1_hat * r1_hat + beta * beta);
    r2 = s_old * alpha + c_oold * c_old * beta_old;
    r3 = s_oold * beta_old;

    if (r1 == 0.0) /* :161-162 */
      return -6;
    c = r1_hat / r1;
    s = beta / r1;

    for (i = 0; i < n; i++) { /* :172-176 */
      tmp = w[i];
      w[i] = (v[i] - r3 * w_old[i] - r2 * tmp) / r1;
      w_old[i] = tmp;
    }
    dconst1 = c * eta; /* :178-180 */
    for (i = 0; i < n; i++)
      x[i] += dconst1 * w[i];
    eta = -s * eta;

    norm_rmr *= fabs(s); /* :192 */
    if (hist)
      hist[*it] = norm_rmr;
  }

  *nrm_res = norm_rmr / norm_r0; /* :195 */
  if (norm_rmr < errtol * norm_r0)
    return 0;
  else
    return -1;
}

/* ---------------------------------------------------- ll_mat feeder semantics */

/* A minimal linked-list matrix with the insertion and conversion semantics of
 * pysparse/sparse/src/ll_mat.c: rows kept as singly linked lists sorted by ascending
 * column (:272-279), zero assignment deletes the entry unless storeZeros (:263,
 * :336-352), freed slots are recycled through a free chain (:282-287), growth by
 * factor 1.5 + 1 (:293-309), symmetric matrices reject writes with i < j (:256-260). */
typedef struct {
  int dim[2];
  int issym, store_zeros;
  int nnz, nalloc, free_;
  double *val;
  int *col, *link, *root;
} orc_ll_t;

ORC_API orc_ll_t *orc_ll_new(int m, int n, int size_hint, int issym, int store_zeros) {
  orc_ll_t *a = (orc_ll_t *)calloc(1, sizeof(orc_ll_t));
  int i;
  if (size_hint < 1)
    size_hint = 1;
  a->dim[0] = m;
  a->dim[1] = n;
  a->issym = issym;
  a->store_zeros = store_zeros;
  a->nalloc = size_hint;
  a->free_ = -1;
  a->val = (double *)malloc(sizeof(double) * size_hint);
  a->col = (int *)malloc(sizeof(int) * size_hint);
  a->link = (int *)malloc(sizeof(int) * size_hint);
  a->root = (int *)malloc(sizeof(int) * (m > 0 ? m : 1));
  for (i = 0; i < m; i++)
    a->root[i] = -1;
  return a;
}

ORC_API void orc_ll_free(orc_ll_t *a) {
  if (!a)
    return;
  free(a->val);
  free(a->col);
  free(a->link);
  free(a->root);
  free(a);
}

ORC_API int orc_ll_nnz(const orc_ll_t *a) { return a->nnz; }

/* ll_mat.c:250-356.  Returns 0, -1 (upper-triangle write on symmetric), -2 (range). */
ORC_API int orc_ll_set(orc_ll_t *a, int i, int j, double x) {
  int k, new_elem, last, col;
  if (a->issym && i < j)
    return -1;
  if (i < 0 || i >= a->dim[0] || j < 0 || j >= a->dim[1])
    return -2;
  col = last = -1;
  k = a->root[i];
  while (k != -1) {
    col = a->col[k];
    if (col >= j)
      break;
    last = k;
    k = a->link[k];
  }
  if (x != 0.0 || a->store_zeros == 1) {
    if (col == j) {
      a->val[k] = x;
    } else {
      if (a->free_ != -1) {
        new_elem = a->free_;
        a->free_ = a->link[new_elem];
      } else {
        /* the reference appends at index nnz (:290); that equals the high-water
         * mark only while nothing was ever freed, which holds here because the
         * free chain is consumed first and nnz counts live entries */
        new_elem = a->nnz;
        if (a->nnz == a->nalloc) {
          int nalloc_new = (int)(1.5 * a->nalloc) + 1;
          a->col = (int *)realloc(a->col, sizeof(int) * nalloc_new);
          a->link = (int *)realloc(a->link, sizeof(int) * nalloc_new);
          a->val = (double *)realloc(a->val, sizeof(double) * nalloc_new);
          a->nalloc = nalloc_new;
        }
      }
      a->val[new_elem] = x;
      a->col[new_elem] = j;
      a->link[new_elem] = k;
      if (last == -1)
        a->root[i] = new_elem;
      else
        a->link[last] = new_elem;
      a->nnz++;
    }
  } else if (col == j) {
    if (last == -1)
      a->root[i] = a->link[k];
    else
      a->link[last] = a->link[k];
    a->link[k] = a->free_;
    a->free_ = k;
    a->nnz--;
  }
  return 0;
}

/* ll_mat.c:210-244 */
ORC_API double orc_ll_get(const orc_ll_t *a, int i, int j) {
  int k, t;
  if (a->issym && i < j) {
    t = i;
    i = j;
    j = t;
  }
  for (k = a->root[i]; k != -1; k = a->link[k])
    if (a->col[k] == j)
      return a->val[k];
  return 0.0;
}

/* number of entries to_csr() will emit: ll_mat.c:1592-1593 (sym: 2*nzLo + nzDiag). */
ORC_API int orc_ll_csr_nnz(const orc_ll_t *a) {
  int i, k, lo = 0, dg = 0;
  if (!a->issym)
    return a->nnz;
  for (i = 0; i < a->dim[0]; i++)
    for (k = a->root[i]; k != -1; k = a->link[k]) {
      if (i > a->col[k])
        lo++;
      else if (i == a->col[k])
        dg++;
    }
  return 2 * lo + dg;
}

/* ll_mat.c:1577-1648.  General: copy each sorted row list.  Symmetric: per row i, the
 * stored (lower + diagonal) entries, then the mirrored entries (i, j > i) in ascending
 * j obtained from a column index built bottom-up (:135-184). */
ORC_API void orc_ll_to_csr(const orc_ll_t *a, double *val, int *col, int *ind) {
  int i, k, r = 0;
  ind[0] = 0;
  if (!a->issym) {
    for (i = 0; i < a->dim[0]; i++) {
      for (k = a->root[i]; k != -1; k = a->link[k]) {
        val[r] = a->val[k];
        col[r] = a->col[k];
        r++;
      }
      ind[i + 1] = r;
    }
    return;
  }
  {
    int n = a->dim[1];
    int *clink = (int *)malloc(sizeof(int) * (a->nalloc > 0 ? a->nalloc : 1));
    int *crow = (int *)malloc(sizeof(int) * (a->nalloc > 0 ? a->nalloc : 1));
    int *croot = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    int j;
    for (j = 0; j < n; j++)
      croot[j] = -1;
    for (i = a->dim[0] - 1; i >= 0; i--)
      for (k = a->root[i]; k != -1; k = a->link[k]) {
        j = a->col[k];
        if (i != j) {
          clink[k] = croot[j];
          croot[j] = k;
          crow[k] = i;
        }
      }
    for (i = 0; i < a->dim[0]; i++) {
      for (k = a->root[i]; k != -1; k = a->link[k]) {
        val[r] = a->val[k];
        col[r] = a->col[k];
        r++;
      }
      for (k = croot[i]; k != -1; k = clink[k]) {
        val[r] = a->val[k];
        col[r] = crow[k];
        r++;
      }
      ind[i + 1] = r;
    }
    free(clink);
    free(crow);
    free(croot);
  }
}

/* ll_mat.c:1671-1680: strict-lower count. */
ORC_API int orc_ll_sss_nnz(const orc_ll_t *a) {
  int i, k, nnz = 0;
  for (i = 0; i < a->dim[0]; i++)
    for (k = a->root[i]; k != -1; k = a->link[k])
      if (i > a->col[k])
        nnz++;
  return nnz;
}

/* ll_mat.c:1654-1708: strict lower -> (val, col, ind), diagonal -> diag (0.0 when
 * absent), upper entries dropped. */
ORC_API void orc_ll_to_sss(const orc_ll_t *a, double *val, double *diag, int *col, int *ind) {
  int i, j, k, r = 0, n = a->dim[0];
  for (i = 0; i < n; i++)
    diag[i] = 0.0;
  ind[0] = 0;
  for (i = 0; i < n; i++) {
    for (k = a->root[i]; k != -1; k = a->link[k]) {
      j = a->col[k];
      if (i > j) {
        val[r] = a->val[k];
        col[r] = j;
        r++;
      } else if (i == j)
        diag[i] = a->val[k];
    }
    ind[i + 1] = r;
  }
}

/* ll_mat.c:1262-1277 (general) and :1300-1320 (symmetric storage) matvec. */
ORC_API void orc_ll_matvec(const orc_ll_t *a, const double *x, double *y) {
  double s, v, xi;
  int i, j, k;
  if (!a->issym) {
    for (i = 0; i < a->dim[0]; i++) {
      s = 0.0;
      for (k = a->root[i]; k != -1; k = a->link[k])
        s += a->val[k] * x[a->col[k]];
      y[i] = s;
    }
  } else {
    for (i = 0; i < a->dim[0]; i++) {
      xi = x[i];
      s = 0.0;
      for (k = a->root[i]; k != -1; k = a->link[k]) {
        j = a->col[k];
        v = a->val[k];
        s += v * x[j];
        if (i != j)
          y[j] += v * xi;
      }
      y[i] = s;
    }
  }
}

/* -------------------------------------------------------- Poisson generators */

/* Direct CSR generators in the ordering of pysparse/tools/poisson.py:22-37
 * (k = i + nx*j, diag 4, off-diag -1, Dirichlet truncation) extended to 3-D
 * (k = i + nx*j + nx*ny*l, diag 6).  Columns ascending within a row, i.e. what
 * poisson2d(n).to_csr() yields through the sorted ll_mat lists.  nz == 0 selects 2-D.
 * Returns nnz; pass NULL arrays to only count. */
ORC_API long orc_poisson_csr(int nx, int ny, int nz, double *val, int *col, int *ind) {
  long r = 0, k, nxy = (long)nx * ny;
  int i, j, l, three_d = nz > 0;
  double dg = three_d ? 6.0 : 4.0;
  if (!three_d)
    nz = 1;
  if (ind)
    ind[0] = 0;
  for (l = 0; l < nz; l++)
    for (j = 0; j < ny; j++)
      for (i = 0; i < nx; i++) {
        k = i + (long)nx * j + nxy * l;
#define ORC_EMIT(c, v)    \
  do {                    \
    if (val) {            \
      val[r] = (v);       \
      col[r] = (int)(c);  \
    }                     \
    r++;                  \
  } while (0)
        if (three_d && l > 0)
          ORC_EMIT(k - nxy, -1.0);
        if (j > 0)
          ORC_EMIT(k - nx, -1.0);
        if (i > 0)
          ORC_EMIT(k - 1, -1.0);
        ORC_EMIT(k, dg);
        if (i < nx - 1)
          ORC_EMIT(k + 1, -1.0);
        if (j < ny - 1)
          ORC_EMIT(k + nx, -1.0);
        if (three_d && l < nz - 1)
          ORC_EMIT(k + nxy, -1.0);
        if (ind)
          ind[k + 1] = (int)r;
      }
  return r;
}

/* Symmetric-skyline form of the same operator: what poisson2d_sym(n).to_sss()
 * (tools/poisson.py:39-50 + ll_mat.c:1654-1708) yields.  Returns strict-lower nnz. */
ORC_API long orc_poisson_sss(int nx, int ny, int nz, double *val, double *diag, int *col,
                             int *ind) {
  long r = 0, k, nxy = (long)nx * ny;
  int i, j, l, three_d = nz > 0;
  double dg = three_d ? 6.0 : 4.0;
  if (!three_d)
    nz = 1;
  if (ind)
    ind[0] = 0;
  for (l = 0; l < nz; l++)
    for (j = 0; j < ny; j++)
      for (i = 0; i < nx; i++) {
        k = i + (long)nx * j + nxy * l;
        if (three_d && l > 0)
          ORC_EMIT(k - nxy, -1.0);
        if (j > 0)
          ORC_EMIT(k - nx, -1.0);
        if (i > 0)
          ORC_EMIT(k - 1, -1.0);
        if (diag)
          diag[k] = dg;
        if (ind)
          ind[k + 1] = (int)r;
      }
#undef ORC_EMIT
  return r;
}

/* ============================================================================
 * The four other Krylov kernels of pysparse/itsolvers.
 *
 * Parity status of THIS block: PINNED (round 3).  cgs.c, bicgstab.c, qmrs.c and gmres.c
 * compile unmodified against the image's Python.h / NumPy headers and need only BLAS-1
 * and the itsolvers_spmatrix table at link time: oracle/Makefile builds them into
 * oracle/_ref/libref_krylov.so (harness: oracle/ref_krylov_harness.c), tests/golden/
 * ref_krylov.json holds what they return on the cases of tests/krylov_cases.py, and
 * tests/test_oracle_krylov_golden.py checks every function below against both.
 * ==========================================================================*/

ORC_API void orc_dscal(int n, double a, double *x) {
  int i;
  for (i = 0; i < n; i++)
    x[i] = a * x[i];
}

/* pysparse/itsolvers/src/cgs.c:14-110.  work = 8n.  Returns info (0 / -1; -100 callback). */
ORC_API int orc_cgs(int n, const double *b, double *x, int maxit, double tol, double *work,
                    int *iter, double *res, orc_matvec_fn matvec, void *mctx,
                    orc_precon_fn precon, void *pctx) {
  double *r0 = work, *r = work + n, *p = work + 2 * (long)n, *q = work + 3 * (long)n,
         *u = work + 4 * (long)n, *v = work + 5 * (long)n, *tmp = work + 6 * (long)n,
         *tmp2 = work + 7 * (long)n;
  double alpha, beta, rho, rho_new, tol_sq = tol * tol, bnrm_sq, ddummy;
  *iter = 0;
  if (matvec(mctx, n, x, tmp)) return -100;
  orc_dcopy(n, b, r0);
  orc_daxpy(n, -1.0, tmp, r0);
  orc_dcopy(n, r0, r);
  orc_dcopy(n, r0, u);
  orc_dcopy(n, r0, p);
  rho = orc_ddot(n, r0, r0);
  bnrm_sq = orc_ddot(n, b, b);
  if (rho < bnrm_sq * tol_sq) {
    *res = sqrt(rho / bnrm_sq);
    return 0;
  }
  for (; *iter < maxit; (*iter)++) {
    if (precon) {
      if (precon(pctx, n, p, tmp)) return -100;
      if (matvec(mctx, n, tmp, v)) return -100;
    } else if (matvec(mctx, n, p, v))
      return -100;
    alpha = rho / orc_ddot(n, v, r0);
    ddummy = -alpha;
    orc_dcopy(n, u, q);
    orc_daxpy(n, ddummy, v, q);
    orc_dcopy(n, u, tmp);
    orc_daxpy(n, 1.0, q, tmp);
    if (precon) {
      if (precon(pctx, n, tmp, tmp2)) return -100;
    } else
      orc_dcopy(n, tmp, tmp2);
    orc_daxpy(n, alpha, tmp2, x);
    if (matvec(mctx, n, tmp2, tmp)) return -100;
    orc_daxpy(n, ddummy, tmp, r);
    *res = orc_ddot(n, r, r);
    if (*res < bnrm_sq * tol_sq) {
      *res = sqrt(*res / bnrm_sq);
      return 0;
    }
    rho_new = orc_ddot(n, r, r0);
    beta = rho_new / rho;
    rho = rho_new;
    orc_dcopy(n, r, u);
    orc_daxpy(n, beta, q, u);
    orc_dcopy(n, q, tmp);
    orc_daxpy(n, beta, p, tmp);
    orc_dcopy(n, u, p);
    orc_daxpy(n, beta, tmp, p);
  }
  *res = sqrt(*res / bnrm_sq);
  return -1;
}

/* pysparse/itsolvers/src/bicgstab.c:233-320 (Itsolvers_bicgstab_kernel).  work = 8n.
 * *info starts at -6 and keeps it on the early returns (rho == 0, omega == 0), where the
 * module ignores the kernel's return value (itsolversmodule.c:185-196). */
ORC_API int orc_bicgstab(int n, double *x, const double *b, double tol, int maxit, int *iter,
                         double *relres, int *info, double *work, orc_matvec_fn matvec,
                         void *mctx, orc_precon_fn precon, void *pctx) {
  double *r = work, *rhat = work + n, *p = work + 2 * (long)n, *phat = work + 3 * (long)n,
         *v = work + 4 * (long)n, *s = work + 5 * (long)n, *shat = work + 6 * (long)n,
         *t = work + 7 * (long)n;
  double alpha = 0.0, omega = 0.0, rho_im1, rho_im2 = 0.0, beta = 0.0, res, res0, n2b;
  int i;
  *info = -6;
  n2b = orc_dnrm2(n, b);
  if (n2b == 0.0) {
    for (i = 0; i < n; i++) x[i] = 0.0;
    *info = 0;
    *relres = 0.0;
    *iter = 0;
    return 0;
  }
  if (matvec(mctx, n, x, r)) return -100;
  for (i = 0; i < n; i++) r[i] = b[i] + -r[i];
  res0 = orc_dnrm2(n, r);
  orc_dcopy(n, r, rhat);
  *iter = 0;
  do {
    (*iter)++;
    rho_im1 = orc_ddot(n, rhat, r);
    if (rho_im1 == 0.0) return -1;
    if (*iter == 1) {
      orc_dcopy(n, r, p);
    } else {
      beta = (rho_im1 / rho_im2) * (alpha / omega);
      for (i = 0; i < n; i++) p[i] = r[i] + beta * (p[i] - omega * v[i]);
    }
    if (precon) {
      if (precon(pctx, n, p, phat)) return -100;
    } else
      orc_dcopy(n, p, phat);
    if (matvec(mctx, n, phat, v)) return -100;
    alpha = rho_im1 / orc_ddot(n, rhat, v);
    for (i = 0; i < n; i++) s[i] = r[i] + (-alpha) * v[i];
    if (precon) {
      if (precon(pctx, n, s, shat)) return -100;
    } else
      orc_dcopy(n, s, shat);
    if (matvec(mctx, n, shat, t)) return -100;
    omega = orc_ddot(n, t, s) / orc_ddot(n, t, t);
    for (i = 0; i < n; i++) x[i] = x[i] + alpha * phat[i] + omega * shat[i];
    for (i = 0; i < n; i++) r[i] = s[i] - omega * t[i];
    res = orc_dnrm2(n, r);
    if (omega == 0.0) return -1;
    rho_im2 = rho_im1;
  } while ((res / res0 > tol) && (*iter < maxit));
  *relres = res / res0;
  *info = (*relres >= tol) ? -1 : 0;
  return 0;
}

/* pysparse/itsolvers/src/qmrs.c:29-154.  work = 6n.  The initial guess is ignored (x := 0).
 * Returns info (0, -1, -2, -6; -100 callback). */
ORC_API int orc_qmrs(int n, const double *b, double *x, double *work, double tol, int maxit,
                     int *iter, double *err, orc_matvec_fn matvec, void *mctx,
                     orc_precon_fn precon, void *pctx) {
  double *wrk1 = work, *p = work + n, *d = work + 2 * (long)n, *v1 = work + 3 * (long)n,
         *t = work + 4 * (long)n, *g = work + 5 * (long)n;
  double beta, res_init, delta, theta, c0, c1, theta0, cc, xi1, rho1inv, tau, eta0, eps0, rho0,
      rho1, d1;
  int i;
  orc_dcopy(n, b, v1);
  rho0 = orc_dnrm2(n, v1);
  tau = rho0;
  for (i = 0; i < n; ++i) {
    v1[i] /= rho0;
    p[i] = 0.0;
    g[i] = 0.0;
    d[i] = 0.0;
    x[i] = 0.0;
  }
  c0 = 1.0;
  eps0 = 1.0;
  xi1 = 1.0;
  theta0 = 0.0;
  eta0 = -1.0;
  res_init = rho0;
  *err = 1.0;
  *iter = 0;
  while (*err > tol && *iter < maxit) {
    ++(*iter);
    if (eps0 == 0.0) return -6;
    if (precon) {
      if (precon(pctx, n, v1, wrk1)) return -100;
    } else
      orc_dcopy(n, v1, wrk1);
    delta = orc_ddot(n, wrk1, v1);
    if (delta == 0.0) return -2;
    cc = xi1 * (delta / eps0);
    for (i = 0; i < n; ++i) {
      p[i] = v1[i] - p[i] * cc;
      g[i] = wrk1[i] - g[i] * cc;
    }
    if (matvec(mctx, n, g, t)) return -100;
    eps0 = orc_ddot(n, g, t);
    beta = eps0 / delta;
    for (i = 0; i < n; ++i) v1[i] = t[i] - v1[i] * beta;
    rho1 = orc_dnrm2(n, v1);
    xi1 = rho1;
    if (c0 * fabs(beta) == 0.0) return -6;
    theta = rho1 / (c0 * fabs(beta));
    c1 = 1.0 / sqrt(theta * theta + 1.0);
    if (beta * (c0 * c0) == 0.0) return -6;
    eta0 = -eta0 * rho0 * (c1 * c1) / (beta * (c0 * c0));
    tau = tau * theta * c1;
    if (rho1 == 0.0) return -6;
    d1 = theta0 * c1;
    cc = d1 * d1;
    rho1inv = 1.0 / rho1;
    for (i = 0; i < n; ++i) {
      d[i] = p[i] * eta0 + d[i] * cc;
      x[i] += d[i];
      v1[i] *= rho1inv;
    }
    if (xi1 == 0.0) return -6;
    rho0 = rho1;
    *err = tau / res_init;
    c0 = c1;
    theta0 = theta;
  }
  if (precon) {
    if (precon(pctx, n, x, wrk1)) return -100;
    orc_dcopy(n, wrk1, x);
  }
  return (*err < tol) ? 0 : -1;
}

static void orc_gen_rot(double dx, double dy, double *cs, double *sn) { /* gmres.c:40-55 */
  if (dy == 0.0) {
    *cs = 1.0;
    *sn = 0.0;
  } else if (fabs(dy) > fabs(dx)) {
    double temp = dx / dy;
    *sn = 1.0 / sqrt(1.0 + temp * temp);
    *cs = temp * *sn;
  } else {
    double temp = dy / dx;
    *cs = 1.0 / sqrt(1.0 + temp * temp);
    *sn = temp * *cs;
  }
}
static void orc_app_rot(double *dx, double *dy, double cs, double sn) { /* gmres.c:56-61 */
  double temp = cs * *dx + sn * *dy;
  *dy = -sn * *dx + cs * *dy;
  *dx = temp;
}

/* pysparse/itsolvers/src/gmres.c:62-175: restarted GMRES(dim), right preconditioning.
 * Returns 0 (-100 callback); *relres is the TRUE residual reduction at exit. */
ORC_API int orc_gmres(int n, double errtol, int it_max, int *it, double *relres, int dim,
                      double *x, const double *b, orc_matvec_fn matvec, void *mctx,
                      orc_precon_fn precon, void *pctx) {
  int m1 = dim + 1, i, j, k, iter = 0, rc = 0;
  double beta, resid0 = 0.0, n2b, rel_resid = 0.0;
  double *H = (double *)malloc(sizeof(double) * dim * (dim + 1));
  double *s = (double *)malloc(sizeof(double) * (dim + 1));
  double *cs = (double *)malloc(sizeof(double) * dim), *sn = (double *)malloc(sizeof(double) * dim);
  double *V = (double *)malloc(sizeof(double) * (size_t)n * (dim + 1));
  double *W = (double *)malloc(sizeof(double) * (size_t)n * dim);
#define OV(i) (&V[(size_t)(i) * n])
#define OW(i) (&W[(size_t)(i) * n])
#define OH(i, j) (H[(j) * m1 + (i)])
  n2b = orc_dnrm2(n, b);
  if (n2b == 0.0) {
    for (i = 0; i < n; i++) x[i] = 0.0;
    *relres = 0.0;
    *it = 0;
    goto done;
  }
  do {
    if (matvec(mctx, n, x, OV(0))) { rc = -100; goto done; }
    orc_daxpy(n, -1.0, b, OV(0));
    beta = sqrt(orc_ddot(n, OV(0), OV(0)));
    orc_dscal(n, -1.0 / beta, OV(0));
    if (iter == 0) resid0 = beta;
    for (i = 1; i < dim + 1; i++) s[i] = 0.0;
    s[0] = beta;
    i = -1;
    do {
      i++;
      iter++;
      if (precon) {
        if (precon(pctx, n, OV(i), OW(i))) { rc = -100; goto done; }
      } else
        orc_dcopy(n, OV(i), OW(i));
      if (matvec(mctx, n, OW(i), OV(i + 1))) { rc = -100; goto done; }
      for (k = 0; k <= i; k++) {
        OH(k, i) = orc_ddot(n, OV(i + 1), OV(k));
        orc_daxpy(n, -OH(k, i), OV(k), OV(i + 1));
      }
      OH(i + 1, i) = sqrt(orc_ddot(n, OV(i + 1), OV(i + 1)));
      orc_dscal(n, 1.0 / OH(i + 1, i), OV(i + 1));
      for (k = 0; k < i; k++) orc_app_rot(&OH(k, i), &OH(k + 1, i), cs[k], sn[k]);
      orc_gen_rot(OH(i, i), OH(i + 1, i), &cs[i], &sn[i]);
      orc_app_rot(&OH(i, i), &OH(i + 1, i), cs[i], sn[i]);
      orc_app_rot(&s[i], &s[i + 1], cs[i], sn[i]);
      rel_resid = fabs(s[i + 1]) / resid0;
      if (rel_resid <= errtol) break;
    } while (i + 1 < dim && iter + 1 <= it_max);
    for (j = i; j >= 0; j--) {
      s[j] /= OH(j, j);
      for (k = j - 1; k >= 0; k--) s[k] -= OH(k, j) * s[j];
    }
    for (j = 0; j <= i; j++) orc_daxpy(n, s[j], OW(j), x);
  } while (rel_resid > errtol && iter + 1 <= it_max);
  if (matvec(mctx, n, x, OV(0))) { rc = -100; goto done; }
  orc_daxpy(n, -1.0, b, OV(0));
  beta = sqrt(orc_ddot(n, OV(0), OV(0)));
  *it = iter;
  *relres = beta / resid0;
done:
#undef OV
#undef OW
#undef OH
  free(H); free(s); free(cs); free(sn); free(V); free(W);
  return rc;
}

/* generic driver: solver = 0 cgs, 1 bicgstab, 2 qmrs, 3 gmres; operator CSR (da == NULL) or SSS */
ORC_API int orc_krylov_more(int solver, int n, const double *va, const double *da, const int *ja,
                            const int *ia, const double *dinv, double *x, const double *b,
                            double tol, int maxit, int dim, int *iter, double *relres) {
  orc_csr_t Ac = {n, n, va, ja, ia};
  orc_sss_t As = {n, va, da, ja, ia};
  orc_matvec_fn mv = da ? orc_sss_matvec_cb : orc_csr_matvec_cb;
  void *mctx = da ? (void *)&As : (void *)&Ac;
  orc_jacobi_t K = {n, dinv, 1, NULL, mv, mctx};
  orc_precon_fn pc = dinv ? orc_jacobi_apply : NULL;
  double *work = (double *)malloc(sizeof(double) * 8 * (size_t)n);
  int info = 0;
  if (solver == 0)
    info = orc_cgs(n, b, x, maxit, tol, work, iter, relres, mv, mctx, pc, &K);
  else if (solver == 1) {
    orc_bicgstab(n, x, b, tol, maxit, iter, relres, &info, work, mv, mctx, pc, &K);
  } else if (solver == 2)
    info = orc_qmrs(n, b, x, work, tol, maxit, iter, relres, mv, mctx, pc, &K);
  else
    info = orc_gmres(n, tol, maxit, iter, relres, dim, x, b, mv, mctx, pc, &K);
  free(work);
  return info;
}

/* PCG on an SSS operator with the SSOR preconditioner (examples/demo_pcg.py:78-98, third column) */
ORC_API int orc_pcg_sss_ssor(int n, const double *va, const double *da, const int *ja, const int *ia,
                             double omega, int steps, double *x, const double *b, double tol, int maxit,
                             int *iter, double *relres, int *flag, double *hist) {
  orc_sss_t As = {n, va, da, ja, ia};
  orc_ssor_t K = {n, va, da, ja, ia, omega, steps, NULL, NULL};
  double *work = (double *)malloc(sizeof(double) * 6 * (size_t)(n > 0 ? n : 1));
  int rc;
  if (!work) return -2;
  K.temp = work + 4 * (size_t)n;
  K.temp2 = work + 5 * (size_t)n;
  rc = orc_pcg(n, x, b, tol, maxit, iter, relres, flag, work, orc_sss_matvec_cb, &As, orc_ssor_apply, &K, hist);
  free(work);
  return rc;
}

/* --------------------------------------- bound shims for the compiled reference */

/* examples/poisson_test/pcg.c takes context-free callbacks
 * void (*matvec)(double *x, double *y), void (*precon)(double *x, double *y)
 * (examples/poisson_test/pcg.h:6-17).  These shims bind one operator / one
 * preconditioner in file-static state so that the reference kernel in oracle/_ref/
 * can be driven with exactly the arithmetic above. */
static orc_csr_t g_csr;
static orc_sss_t g_sss;
static const double *g_dinv;
static int g_n;

ORC_API void orc_bind_csr(int m, int n, const double *va, const int *ja, const int *ia) {
  g_csr.m = m;
  g_csr.n = n;
  g_csr.va = va;
  g_csr.ja = ja;
  g_csr.ia = ia;
}
ORC_API void orc_bind_sss(int n, const double *va, const double *da, const int *ja,
                          const int *ia) {
  g_sss.n = n;
  g_sss.va = va;
  g_sss.da = da;
  g_sss.ja = ja;
  g_sss.ia = ia;
}
ORC_API void orc_bind_dinv(int n, const double *dinv) {
  g_n = n;
  g_dinv = dinv;
}
ORC_API void orc_bound_csr_matvec(double *x, double *y) {
  orc_csr_matvec(g_csr.m, x, y, g_csr.va, g_csr.ja, g_csr.ia);
}
ORC_API void orc_bound_sss_matvec(double *x, double *y) {
  orc_sss_matvec(g_sss.n, x, y, g_sss.va, g_sss.da, g_sss.ja, g_sss.ia);
}
ORC_API void orc_bound_jacobi(double *x, double *y) {
  int i;
  for (i = 0; i < g_n; i++)
    y[i] = x[i] * g_dinv[i];
}

/* Convenience drivers used by the ctypes wrapper: whole solves on CSR / SSS operators
 * with K = None (dinv == NULL) or Jacobi(steps). */
ORC_API int orc_pcg_csr(int n, const double *va, const int *ja, const int *ia, const double *dinv,
                        int steps, double *x, const double *b, double tol, int maxit, int *iter,
                        double *relres, int *flag, double *hist) {
  orc_csr_t A = {n, n, va, ja, ia};
  orc_jacobi_t K = {n, dinv, steps, NULL, orc_csr_matvec_cb, &A};
  double *work = (double *)malloc(sizeof(double) * 4 * (size_t)n);
  int rc;
  if (dinv && steps > 1)
    K.temp = (double *)malloc(sizeof(double) * (size_t)n);
  rc = orc_pcg(n, x, b, tol, maxit, iter, relres, flag, work, orc_csr_matvec_cb, &A,
               dinv ? orc_jacobi_apply : NULL, &K, hist);
  free(K.temp);
  free(work);
  return rc;
}

ORC_API int orc_pcg_sss(int n, const double *va, const double *da, const int *ja, const int *ia,
                        const double *dinv, int steps, double *x, const double *b, double tol,
                        int maxit, int *iter, double *relres, int *flag, double *hist) {
  orc_sss_t A = {n, va, da, ja, ia};
  orc_jacobi_t K = {n, dinv, steps, NULL, orc_sss_matvec_cb, &A};
  double *work = (double *)malloc(sizeof(double) * 4 * (size_t)n);
  int rc;
  if (dinv && steps > 1)
    K.temp = (double *)malloc(sizeof(double) * (size_t)n);
  rc = orc_pcg(n, x, b, tol, maxit, iter, relres, flag, work, orc_sss_matvec_cb, &A,
               dinv ? orc_jacobi_apply : NULL, &K, hist);
  free(K.temp);
  free(work);
  return rc;
}

ORC_API int orc_minres_csr(int n, const double *va, const int *ja, const int *ia,
                           const double *dinv, int steps, double *x, const double *b, double tol,
                           int maxit, int *iter, double *relres, double *hist) {
  orc_csr_t A = {n, n, va, ja, ia};
  orc_jacobi_t K = {n, dinv, steps, NULL, orc_csr_matvec_cb, &A};
  double *work = (double *)malloc(sizeof(double) * 7 * (size_t)n);
  int info;
  if (dinv && steps > 1)
    K.temp = (double *)malloc(sizeof(double) * (size_t)n);
  info = orc_minres(n, tol, maxit, iter, relres, x, b, work, orc_csr_matvec_cb, &A,
                    dinv ? orc_jacobi_apply : NULL, &K, hist);
  free(K.temp);
  free(work);
  return info;
}

ORC_API int orc_minres_sss(int n, const double *va, const double *da, const int *ja,
                           const int *ia, const double *dinv, int steps, double *x,
                           const double *b, double tol, int maxit, int *iter, double *relres,
                           double *hist) {
  orc_sss_t A = {n, va, da, ja, ia};
  orc_jacobi_t K = {n, dinv, steps, NULL, orc_sss_matvec_cb, &A};
  double *work = (double *)malloc(sizeof(double) * 7 * (size_t)n);
  int info;
  if (dinv && steps > 1)
    K.temp = (double *)malloc(sizeof(double) * (size_t)n);
  info = orc_minres(n, tol, maxit, iter, relres, x, b, work, orc_sss_matvec_cb, &A,
                    dinv ? orc_jacobi_apply : NULL, &K, hist);
  free(K.temp);
  free(work);
  return info;
}

/* Generic driver with the argument list of refk_solve (oracle/ref_krylov_harness.c), so that the
 * restatements above and the compiled reference kernels can be run on the very same operator /
 * preconditioner contexts.  solver: 0 pcg, 1 minres, 2 cgs, 3 bicgstab, 4 qmrs, 5 gmres;
 * work: 8n doubles.  *info = what the module wrapper would return as `info`. */
ORC_API int orc_solve_cb(int solver, int n, double *x, const double *b, double tol, int maxit, int dim,
                         int *iter, double *relres, int *info, double *work, orc_matvec_fn mv,
                         void *mctx, orc_precon_fn pc, void *pctx) {
  int rc;
  switch (solver) {
  case 0:
    return orc_pcg(n, x, b, tol, maxit, iter, relres, info, work, mv, mctx, pc, pctx, NULL);
  case 1:
    rc = orc_minres(n, tol, maxit, iter, relres, x, b, work, mv, mctx, pc, pctx, NULL);
    break;
  case 2:
    rc = orc_cgs(n, b, x, maxit, tol, work, iter, relres, mv, mctx, pc, pctx);
    break;
  case 3:
    return orc_bicgstab(n, x, b, tol, maxit, iter, relres, info, work, mv, mctx, pc, pctx);
  case 4:
    rc = orc_qmrs(n, b, x, work, tol, maxit, iter, relres, mv, mctx, pc, pctx);
    break;
  case 5:
    rc = orc_gmres(n, tol, maxit, iter, relres, dim, x, b, mv, mctx, pc, pctx);
    break;
  default:
    return -101;
  }
  *info = rc;
  return rc;
}
