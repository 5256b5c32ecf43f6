// psp_cpu.hip -- the opt-in host mode of the library: PSP_DEVICE=cpu.
//
// BASELINE.json configs[0] is the reference's own CPU-runnable case ("2D Poisson 5-pt 100x100 CSR SpMV + Jacobi-PCG
// on CPU (examples/demo_pcg.py plumbing, no GPU)"), SURVEY.md section 8b asks that the library "be loadable with no
// GPU present (CPU fallback for config 1)".  This file is that mode and nothing more: plain single-threaded loops for
// the csr_mat / sss_mat products (csr_mat.c:49-54, :58-106; sss_mat.c:40-76), jacobi (preconmodule.c:35-54,
// :389-401), ssor (preconmodule.c:95-223; the sweeps live in psp_ssor.hip), pcg (pcg.c:22-171) and minres
// (minres.c:43-200) on host arrays, behind the same C ABI and the same
// handles, so that examples/demo_pcg.py runs on a machine without a GPU.
//
// It is NOT a fallback: nothing selects it but the environment variable PSP_DEVICE=cpu, read once when the library
// is first used.  Without the variable -- GPU or not -- every compute entry point goes to the HIP path and fails with
// PSP_ENODEV when there is no device (tests/test_capi_symbols.py keeps asserting that); with it, the entry points
// without a host loop (device-pointer variants, the phase kernels, cgs / bicgstab / qmrs / gmres, multi-device
// matrices) answer PSP_ENODEV with a message that names the mode.  psp_version() says which mode a process is in.
// None of this touches, links or reads anything under oracle/: the loops below are this library's own restatement
// of the cited reference lines (no FMA contraction: the library is built with -ffp-contract=off).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "psp_internal.h"

namespace psp {

bool cpu_mode() {
  static const bool on = [] {
    const char *e = getenv("PSP_DEVICE");
    return e && !strcmp(e, "cpu");
  }();
  return on;
}

namespace cpu {

namespace {

template <typename T>
T *dup(const T *src, size_t n) {
  T *p = (T *)malloc(sizeof(T) * (n ? n : 1));
  if (p && src && n) memcpy(p, src, sizeof(T) * n);
  return p;
}

// BLAS level 1 as the reference BLAS does it (blas.h:97-118): sequential sums, the scale / sum-of-squares dnrm2
double ddot(int n, const double *x, const double *y) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}
double dnrm2(int n, const double *x) {
  double scale = 0.0, ssq = 1.0;
  for (int i = 0; i < n; ++i)
    if (x[i] != 0.0) {
      const double a = std::fabs(x[i]);
      if (scale < a) {
        ssq = 1.0 + ssq * (scale / a) * (scale / a);
        scale = a;
      } else {
        ssq += (a / scale) * (a / scale);
      }
    }
  return scale * std::sqrt(ssq);
}
void daxpy(int n, double a, const double *x, double *y) {
  if (a == 0.0) return;  // the reference BLAS returns at once
  for (int i = 0; i < n; ++i) y[i] += a * x[i];
}

// csr_mat.c:49-54 (strided form :58-72)
void csr_mv(const psp_csr *A, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy) {
  for (int i = 0; i < A->nrows; ++i) {
    double s = 0.0;
    for (int k = A->ind[i]; k < A->ind[i + 1]; ++k) s += A->val[k] * x[(ptrdiff_t)A->col[k] * incx];
    y[(ptrdiff_t)i * incy] = s;
  }
}
// csr_mat.c:74-106
void csr_mv_transp(const psp_csr *A, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy) {
  for (int j = 0; j < A->ncols; ++j) y[(ptrdiff_t)j * incy] = 0.0;
  for (int i = 0; i < A->nrows; ++i) {
    const double xi = x[(ptrdiff_t)i * incx];
    for (int k = A->ind[i]; k < A->ind[i + 1]; ++k) y[(ptrdiff_t)A->col[k] * incy] += A->val[k] * xi;
  }
}
// sss_mat.c:40-56 (strided form :58-76): gather and scatter in one sweep, y[i] ASSIGNED at the end of row i
void sss_mv(const psp_sss *S, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy) {
  for (int i = 0; i < S->n; ++i) {
    const double xi = x[(ptrdiff_t)i * incx];
    double s = 0.0;
    for (int k = S->ind[i]; k < S->ind[i + 1]; ++k) {
      const int j = S->col[k];
      const double v = S->val[k];
      s += v * x[(ptrdiff_t)j * incx];
      y[(ptrdiff_t)j * incy] += v * xi;
    }
    y[(ptrdiff_t)i * incy] = s + S->diag[i] * xi;
  }
}

int apply(const psp_op *op, const double *x, double *y);

// preconmodule.c:35-54
int jacobi_apply(const psp_jacobi *K, const double *x, double *y) {
  const int n = K->n;
  for (int i = 0; i < n; ++i) y[i] = x[i] * K->dinv[i];
  for (int step = 1; step < K->steps; ++step) {
    memcpy(K->temp, y, sizeof(double) * (size_t)n);
    PSP_TRY(apply(&K->A, K->temp, y));
    for (int i = 0; i < n; ++i) y[i] = (x[i] - y[i]) * K->dinv[i] + K->temp[i];
  }
  return PSP_OK;
}

int apply(const psp_op *op, const double *x, double *y) {
  switch (op->kind) {
    case PSP_OP_CSR: csr_mv(op->csr, x, 1, y, 1); return PSP_OK;
    case PSP_OP_SSS: sss_mv(op->sss, x, 1, y, 1); return PSP_OK;
    case PSP_OP_JACOBI: return jacobi_apply(op->jac, x, y);
    case PSP_OP_SSOR: return ssor_apply_host(op->ssor, x, y);
    case PSP_OP_CALLBACK:
      if (op->fn(op->ctx, op->n, x, y)) return fail(PSP_ECALLBACK, "host operator callback failed");
      return PSP_OK;
    default: return fail(PSP_ENODEV, "PSP_DEVICE=cpu: this operator kind has no host loop");
  }
}

}  // namespace

// ------------------------------------------------------------------ containers

int csr_create(int nrows, int ncols, int nnz, const int *ind, const int *col, const double *val, psp_csr **out) {
  if (!out || !ind || (nnz > 0 && (!col || !val))) return fail(PSP_EINVAL, "psp_csr_create: NULL argument");
  if (nrows < 0 || ncols < 0 || nnz < 0 || ind[0] != 0 || ind[nrows] != nnz)
    return fail(PSP_EINVAL, "psp_csr_create: ind[0] must be 0 and ind[nrows] == nnz");
  for (int i = 0; i < nrows; ++i)
    if (ind[i + 1] < ind[i]) return fail(PSP_EINVAL, "psp_csr_create: ind not monotone at row %d", i);
  for (int k = 0; k < nnz; ++k)
    if (col[k] < 0 || col[k] >= ncols) return fail(PSP_EINVAL, "psp_csr_create: column index out of range");
  psp_csr *A = new psp_csr();
  A->nrows = nrows;
  A->ncols = ncols;
  A->nnz = nnz;
  A->nnz64 = nnz;
  A->host = true;
  A->ind = dup(ind, (size_t)nrows + 1);
  A->col = dup(col, (size_t)nnz);
  A->val = dup(val, (size_t)nnz);
  if (!A->ind || !A->col || !A->val) {
    csr_destroy(A);
    return fail(PSP_ENOMEM, "psp_csr_create: host allocation failed");
  }
  *out = A;
  return PSP_OK;
}

int csr_destroy(psp_csr *A) {
  free(A->ind);
  free(A->col);
  free(A->val);
  delete A;
  return PSP_OK;
}

// tools/poisson.py:22-37 ordering k = i + nx*j (+ nx*ny*l), diag 4 / 6, off-diagonals -1, columns ascending
int csr_poisson(int nx, int ny, int nz, psp_csr **out) {
  if (nx < 1 || ny < 1 || nz < 0) return fail(PSP_EINVAL, "psp_csr_poisson: bad grid");
  const int nzz = nz > 0 ? nz : 1;
  const long n = (long)nx * ny * nzz;
  if (7 * n > 0x7fffffffL) return fail(PSP_EINVAL, "psp_csr_poisson: grid too large for 32-bit offsets");
  std::vector<int> ind((size_t)n + 1), col;
  std::vector<double> val;
  col.reserve((size_t)(nz > 0 ? 7 : 5) * n);
  val.reserve(col.capacity());
  const double d = nz > 0 ? 6.0 : 4.0;
  for (int l = 0; l < nzz; ++l)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const long k = i + (long)nx * j + (long)nx * ny * l;
        ind[k] = (int)col.size();
        auto put = [&](long c, double v) {
          col.push_back((int)c);
          val.push_back(v);
        };
        if (l > 0) put(k - (long)nx * ny, -1.0);
        if (j > 0) put(k - nx, -1.0);
        if (i > 0) put(k - 1, -1.0);
        put(k, d);
        if (i < nx - 1) put(k + 1, -1.0);
        if (j < ny - 1) put(k + nx, -1.0);
        if (l < nzz - 1) put(k + (long)nx * ny, -1.0);
      }
  ind[n] = (int)col.size();
  return csr_create((int)n, (int)n, (int)col.size(), ind.data(), col.data(), val.data(), out);
}

int csr_download(const psp_csr *A, int *ind, int *col, double *val) {
  if (ind) memcpy(ind, A->ind, sizeof(int) * ((size_t)A->nrows + 1));
  if (col) memcpy(col, A->col, sizeof(int) * (size_t)A->nnz);
  if (val) memcpy(val, A->val, sizeof(double) * (size_t)A->nnz);
  return PSP_OK;
}

int csr_diagonal(const psp_csr *A, double *diag) {
  for (int i = 0; i < A->nrows; ++i) {
    double d = 0.0;
    for (int k = A->ind[i]; k < A->ind[i + 1]; ++k)
      if (A->col[k] == i) d = A->val[k];
    diag[i] = d;
  }
  return PSP_OK;
}

int csr_matvec(const psp_csr *A, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy, bool transp) {
  if (transp) csr_mv_transp(A, x, incx, y, incy);
  else csr_mv(A, x, incx, y, incy);
  return PSP_OK;
}

int sss_create(int n, int nnz, const int *ind, const int *col, const double *val, const double *diag, psp_sss **out) {
  if (!out || !ind || !diag || (nnz > 0 && (!col || !val))) return fail(PSP_EINVAL, "psp_sss_create: NULL argument");
  if (n < 0 || nnz < 0 || ind[0] != 0 || ind[n] != nnz)
    return fail(PSP_EINVAL, "psp_sss_create: ind[0] must be 0 and ind[n] == nnz");
  for (int i = 0; i < n; ++i) {
    if (ind[i + 1] < ind[i]) return fail(PSP_EINVAL, "psp_sss_create: ind not monotone at row %d", i);
    for (int k = ind[i]; k < ind[i + 1]; ++k)
      if (col[k] < 0 || col[k] >= i) return fail(PSP_EINVAL, "psp_sss_create: entry (%d,%d) is not strictly lower", i, col[k]);
  }
  psp_sss *S = new psp_sss();
  S->n = n;
  S->nnz_lower = nnz;
  S->host = true;
  S->ind = dup(ind, (size_t)n + 1);
  S->col = dup(col, (size_t)nnz);
  S->val = dup(val, (size_t)nnz);
  S->diag = dup(diag, (size_t)n);
  if (!S->ind || !S->col || !S->val || !S->diag) {
    sss_destroy(S);
    return fail(PSP_ENOMEM, "psp_sss_create: host allocation failed");
  }
  *out = S;
  return PSP_OK;
}

int sss_destroy(psp_sss *S) {
  free(S->ind);
  free(S->col);
  free(S->val);
  free(S->diag);
  delete S;
  return PSP_OK;
}

int sss_poisson(int nx, int ny, int nz, psp_sss **out) {
  psp_csr *A = nullptr;
  PSP_TRY(csr_poisson(nx, ny, nz, &A));
  const int n = A->nrows;
  std::vector<int> ind((size_t)n + 1), col;
  std::vector<double> val, diag((size_t)n, 0.0);
  for (int i = 0; i < n; ++i) {
    ind[i] = (int)col.size();
    for (int k = A->ind[i]; k < A->ind[i + 1]; ++k) {
      if (A->col[k] < i) {
        col.push_back(A->col[k]);
        val.push_back(A->val[k]);
      } else if (A->col[k] == i) {
        diag[i] = A->val[k];
      }
    }
  }
  ind[n] = (int)col.size();
  csr_destroy(A);
  return sss_create(n, (int)col.size(), ind.data(), col.data(), val.data(), diag.data(), out);
}

int sss_download(const psp_sss *S, int *ind, int *col, double *val, double *diag) {
  if (ind) memcpy(ind, S->ind, sizeof(int) * ((size_t)S->n + 1));
  if (col) memcpy(col, S->col, sizeof(int) * (size_t)S->nnz_lower);
  if (val) memcpy(val, S->val, sizeof(double) * (size_t)S->nnz_lower);
  if (diag) memcpy(diag, S->diag, sizeof(double) * (size_t)S->n);
  return PSP_OK;
}

// the intent of sss_mat.c:14-28: diagonal fast path, swap to the lower triangle, linear scan
int sss_getitem(const psp_sss *S, int i, int j, double *value) {
  if (i == j) {
    *value = S->diag[i];
    return PSP_OK;
  }
  if (i < j) std::swap(i, j);
  *value = 0.0;
  for (int k = S->ind[i]; k < S->ind[i + 1]; ++k)
    if (S->col[k] == j) {
      *value = S->val[k];
      break;
    }
  return PSP_OK;
}

int sss_matvec(const psp_sss *S, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy) {
  sss_mv(S, x, incx, y, incy);
  return PSP_OK;
}

// ------------------------------------------------------------------ jacobi (preconmodule.c:352-412)

int jacobi_create(int n, const double *diag, double omega, int steps, const psp_op *A, psp_jacobi **out) {
  if (steps < 1) return fail(PSP_EINVAL, "jacobi: steps must be >= 1");
  if (steps > 1 && !A) return fail(PSP_EINVAL, "jacobi: steps > 1 needs the matrix operator");
  psp_jacobi *K = new psp_jacobi();
  K->n = n;
  K->omega = omega;
  K->steps = steps;
  K->host = true;
  K->dinv = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
  if (steps > 1) K->temp = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
  if (A) K->A = *A;
  if (!K->dinv || (steps > 1 && !K->temp)) {
    jacobi_destroy(K);
    return fail(PSP_ENOMEM, "jacobi: host allocation failed");
  }
  for (int i = 0; i < n; ++i) {
    if (1.0 + diag[i] == 1.0) {  // preconmodule.c:395-397
      jacobi_destroy(K);
      return fail(PSP_ESINGULAR, "diagonal element close to zero");
    }
    K->dinv[i] = omega / diag[i];
  }
  *out = K;
  return PSP_OK;
}

int jacobi_destroy(psp_jacobi *K) {
  free(K->dinv);
  free(K->temp);
  delete K;
  return PSP_OK;
}

int jacobi_precon(const psp_jacobi *K, const double *x, double *y) { return jacobi_apply(K, x, y); }

// ------------------------------------------------------------------ solvers

// pcg.c:22-171
int pcg(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double tol, int maxit, int *info, int *iter,
        double *relres, double *hist) {
  std::vector<double> work((size_t)4 * n);
  double *r = work.data(), *z = r + n, *p = z + n, *q = p + n;
  const double n2b = dnrm2(n, b);
  if (n2b == 0.0) {  // :58-67
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    *info = 0;
    *relres = 0.0;
    *iter = 0;
    return PSP_OK;
  }
  *info = -1;
  const double tolb = tol * n2b;
  PSP_TRY(apply(A, x, r));
  for (int i = 0; i < n; ++i) r[i] = b[i] - r[i];
  double normr = dnrm2(n, r);
  if (hist) hist[0] = normr;
  if (normr <= tolb) {  // :77-84
    *info = 0;
    *relres = normr / n2b;
    *iter = 0;
    return PSP_OK;
  }
  double rho = 1.0, rho1, alpha, beta, pq;
  int stag = 0, it;
  for (it = 1; it <= maxit; ++it) {
    if (K) PSP_TRY(apply(K, r, z));
    else memcpy(z, r, sizeof(double) * (size_t)n);
    rho1 = rho;
    rho = ddot(n, r, z);
    if (rho == 0.0) {  // :101-104
      *info = -2;
      break;
    }
    if (it == 1) {
      memcpy(p, z, sizeof(double) * (size_t)n);
    } else {
      beta = rho / rho1;
      if (beta == 0.0) {  // :109-112
        *info = -6;
        break;
      }
      for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    PSP_TRY(apply(A, p, q));
    pq = ddot(n, p, q);
    if (pq == 0.0) {  // :118-120
      *info = -6;
      break;
    }
    alpha = rho / pq;
    if (alpha == 0.0) stag = 1;
    if (stag == 0) {  // :127-139
      double dmax = 0.0;
      for (int i = 0; i < n; ++i)
        if (x[i] != 0.0) {
          const double d = std::fabs(alpha * p[i] / x[i]);
          if (d > dmax) dmax = d;
        } else if (p[i] != 0.0) {
          dmax = 1.0;
        }
      stag = (1.0 + dmax == 1.0);
    }
    daxpy(n, alpha, p, x);
    daxpy(n, -alpha, q, r);
    normr = dnrm2(n, r);  // the recurred residual (EXPENSIVE_CRIT undefined, :146-153)
    if (hist) hist[it] = normr;
    if (normr <= tolb) {
      *info = 0;
      break;
    }
    if (stag == 1) {
      *info = -5;
      break;
    }
  }
  *iter = it;  // maxit + 1 when the loop ran out (:165)
  *relres = normr / n2b;
  return PSP_OK;
}

// minres.c:43-200; *relres is left untouched on the -3 / -6 exits, like the reference
int minres(const psp_op *A, const psp_op *K, int n, double *x, const double *b, double errtol, int it_max, int *info,
           int *iter, double *relres, double *hist) {
  std::vector<double> work((size_t)7 * n, 0.0);
  double *v_hat_old = work.data(), *v_hat = v_hat_old + n, *y = v_hat + n, *w = y + n, *w_old = w + n, *v = w_old + n,
         *av = v + n;
  *iter = 0;
  PSP_TRY(apply(A, x, v_hat));
  for (int i = 0; i < n; ++i) v_hat[i] = b[i] - v_hat[i];
  const double norm_r0 = dnrm2(n, v_hat);
  if (K) PSP_TRY(apply(K, v_hat, y));
  else memcpy(y, v_hat, sizeof(double) * (size_t)n);
  double beta = ddot(n, v_hat, y);
  if (beta < 0.0) {  // :79-80
    *info = -3;
    return PSP_OK;
  }
  beta = std::sqrt(beta);
  double beta_old = 1.0, c = 1.0, c_old = 1.0, s = 0.0, s_old = 0.0, eta = beta, norm_rmr = norm_r0;
  if (hist) hist[0] = norm_rmr;
  for (;;) {
    if (*iter >= it_max || norm_rmr < errtol * norm_r0) break;  // :114
    *iter += 1;
    for (int i = 0; i < n; ++i) v[i] = y[i] / beta;
    memcpy(y, v_hat, sizeof(double) * (size_t)n);
    PSP_TRY(apply(A, v, av));
    const double alpha = ddot(n, v, av);
    const double c1 = alpha / beta, c2 = beta / beta_old;
    for (int i = 0; i < n; ++i) v_hat[i] = av[i] - c1 * v_hat[i] - c2 * v_hat_old[i];
    memcpy(v_hat_old, y, sizeof(double) * (size_t)n);
    if (K) PSP_TRY(apply(K, v_hat, y));
    else memcpy(y, v_hat, sizeof(double) * (size_t)n);
    beta_old = beta;
    beta = ddot(n, v_hat, y);
    if (beta < 0.0) {  // :144-146
      *info = -3;
      return PSP_OK;
    }
    beta = std::sqrt(beta);
    const double c_oold = c_old, s_oold = s_old;
    c_old = c;
    s_old = s;
    const double r1_hat = c_old * alpha - c_oold * s_old * beta_old;
    const double r1 = std::sqrt(r1_hat * r1_hat + beta * beta);
    const double r2 = s_old * alpha + c_oold * c_old * beta_old;
    const double r3 = s_oold * beta_old;
    if (r1 == 0.0) {  // :160-162
      *info = -6;
      return PSP_OK;
    }
    c = r1_hat / r1;
    s = beta / r1;
    for (int i = 0; i < n; ++i) {
      const double tmp = w[i];
      w[i] = (v[i] - r3 * w_old[i] - r2 * tmp) / r1;
      w_old[i] = tmp;
    }
    const double ce = c * eta;
    for (int i = 0; i < n; ++i) x[i] += ce * w[i];
    eta = -s * eta;
    norm_rmr *= std::fabs(s);  // the estimate in the preconditioned norm (:192)
    if (hist) hist[*iter] = norm_rmr;
  }
  *relres = norm_rmr / norm_r0;
  *info = norm_rmr < errtol * norm_r0 ? 0 : -1;
  return PSP_OK;
}

}  // namespace cpu
}  // namespace psp
