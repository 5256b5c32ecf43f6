// psp_ssor.hip -- precon.ssor(A, omega, steps) on an sss_mat: symmetric Gauss-Seidel / SSOR sweeps.
//
// Reference: pysparse/precon/src/preconmodule.c -- ssor_kernel :95-143 (omega != 1),
// symgs_kernel :149-193 (omega == 1), SSOR_precon :199-223, newSSORObject :414-459.
// Both are sequential triangular sweeps: the forward half-step visits rows in ascending order
// and row i reads x[j] of its lower entries j < i; the backward half-step visits rows in
// descending order and row i SCATTERS va[k]*x[i] into y[j] (h[j]) of its lower entries.
//
// On the GPU the sweeps are level-scheduled, which keeps every floating-point operation and
// its order:
//   * forward: level(i) = 1 + max level(j) over the lower entries of row i; the rows of one
//     level are independent and each adds its lower products in storage order (ascending
//     column), exactly like the CPU loop;
//   * backward: the scatter is turned into a gather over the mirrored upper entries of the
//     handle's full CSR form.  y[j] receives its contributions in descending row order on the
//     CPU (the sweep runs i = n-1 .. 0); the gather walks row j's upper entries from the last
//     to the first, i.e. the same addends in the same order, and rlevel(j) = 1 + max rlevel(i)
//     over the upper entries guarantees the x[i] it reads are final.
// One small kernel per level (a 512^3 grid has 1534 levels in each direction); the rows of a
// level are kept sorted so neighbouring lanes touch neighbouring lines.
// Bandwidth is not the bound here (dependent launches are); see DESIGN.md.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "psp_internal.h"

using namespace psp;

struct psp_ssor {
  int n = 0;
  double omega = 1.0;
  int steps = 1;
  psp_sss *S = nullptr;  // borrowed; the Python object keeps the matrix alive
  int *rows_f = nullptr, *rows_b = nullptr;  // rows sorted by (level, row)
  std::vector<int> ptr_f, ptr_b;             // level l = rows[ptr[l] .. ptr[l+1])
  int *dptr_f = nullptr, *dptr_b = nullptr;  // the same on the device (runs of small levels)
  double *temp = nullptr;                    // y (symgs) / h (ssor)
};

namespace {

// one relaxation pass of the longest-path levels; dir 0: lower entries (col < i), 1: upper
__global__ void level_pass_kernel(int n, const int *__restrict__ ind, const int *__restrict__ col, int dir,
                                  int *level, int *changed) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int L = 0;
    for (int k = ind[i]; k < ind[i + 1]; ++k) {
      const int j = col[k];
      if (dir == 0 ? j < i : j > i) {
        const int lj = *(volatile int *)(level + j) + 1;
        L = lj > L ? lj : L;
      }
    }
    if (L > level[i]) {
      level[i] = L;
      *changed = 1;
    }
  }
}

__global__ void iota_kernel(int n, int *v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}

__global__ void level_ptr_kernel(int n, const int *__restrict__ keys, int *__restrict__ ptr) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (i == 0 || keys[i] != keys[i - 1]) ptr[keys[i]] = i;
}

// ---- one row of each sweep (KIND 0: symgs forward, 1: symgs backward, 2: ssor forward, 3: ssor backward)
// symgs_kernel, preconmodule.c:149-193 (omega == 1):
//   forward : s = sum_{lower} va*x[j]; x[i] = (b[i] - y[i] - s)/da[i]; y[i] = s          (:171-179)
//   backward: x[i] holds y of the forward step ("x[k] = y[k]", :182-185), y[i] is rebuilt from the
//             rows above in descending order, x[i] = (b[i] - x[i] - y[i]) / da[i]         (:186-193)
// ssor_kernel, preconmodule.c:95-143 (omega != 1): temp / h as at :110-140
template <int KIND>
__device__ __forceinline__ void ssor_row(int i, const int *__restrict__ ind, const int *__restrict__ col,
                                         const double *__restrict__ val, const double *__restrict__ da,
                                         const double *__restrict__ b, double *x, double *y, double omega,
                                         int first) {
  if constexpr (KIND == 0) {
    double s = 0.0;
    for (int k = ind[i]; k < ind[i + 1] && col[k] < i; ++k) s += val[k] * x[col[k]];
    x[i] = (b[i] - y[i] - s) / da[i];
    y[i] = s;
  } else if constexpr (KIND == 1) {
    const double yf = y[i];
    double acc = 0.0;
    for (int k = ind[i + 1] - 1; k >= ind[i] && col[k] > i; --k) acc += val[k] * x[col[k]];
    x[i] = (b[i] - yf - acc) / da[i];
    y[i] = acc;
  } else if constexpr (KIND == 2) {
    const double temp = first ? omega * b[i] : (1.0 - omega) * x[i] * da[i] + y[i] + omega * b[i];
    double s = 0.0;
    for (int k = ind[i]; k < ind[i + 1] && col[k] < i; ++k) s -= val[k] * x[col[k]];
    const double hi = omega * s;
    y[i] = hi;
    x[i] = (temp + hi) / da[i];
  } else {
    const double temp = (1.0 - omega) * x[i] * da[i] + y[i] + omega * b[i];
    double acc = 0.0;
    for (int k = ind[i + 1] - 1; k >= ind[i] && col[k] > i; --k) acc -= val[k] * x[col[k]];
    const double hi = omega * acc;
    y[i] = hi;
    x[i] = (temp + hi) / da[i];
  }
}

// the rows of ONE level (independent of each other)
template <int KIND>
__global__ void ssor_level_kernel(int cnt, const int *__restrict__ rows, const int *__restrict__ ind,
                                  const int *__restrict__ col, const double *__restrict__ val,
                                  const double *__restrict__ da, const double *__restrict__ b, double *x,
                                  double *y, double omega, int first) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < cnt) ssor_row<KIND>(rows[t], ind, col, val, da, b, x, y, omega, first);
}

// a RUN of small levels [l0, l1) in one launch: a single workgroup walks them with a barrier in
// between (stores of one level are visible to the whole workgroup after __syncthreads()).  Small
// problems and the thin ends of big ones are launch-bound otherwise: poisson2d(100) has 199 levels
// of at most 100 rows in each direction.
constexpr int kSmallLevel = 256;  // measured: 2048 made poisson2d(2048) 3x slower (one CU walks 18 us levels)
template <int KIND>
__global__ __launch_bounds__(256) void ssor_levels_kernel(int l0, int l1, const int *__restrict__ ptr,
                                                          const int *__restrict__ rows, const int *__restrict__ ind,
                                                          const int *__restrict__ col, const double *__restrict__ val,
                                                          const double *__restrict__ da, const double *__restrict__ b,
                                                          double *x, double *y, double omega, int first) {
  for (int l = l0; l < l1; ++l) {
    const int a = ptr[l], e = ptr[l + 1];
    for (int t = a + (int)threadIdx.x; t < e; t += (int)blockDim.x)
      ssor_row<KIND>(rows[t], ind, col, val, da, b, x, y, omega, first);
    __syncthreads();
  }
}

// longest-path levels of the lower (dir 0) / upper (dir 1) dependency graph, rows sorted by level
int build_schedule(const psp_csr *F, int dir, int **rows_out, std::vector<int> *ptr_out) {
  const int n = F->nrows;
  int *level = nullptr, *changed = nullptr, *keys = nullptr, *vals = nullptr, *rows = nullptr, *dptr = nullptr;
  void *tmp = nullptr;
  int rc = PSP_OK;
  const int grid = std::min((n + 255) / 256, 16384);
#define SS_HIP(call)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      rc = fail(e_ == hipErrorOutOfMemory ? PSP_ENOMEM : PSP_ENODEV, "%s: %s", #call,    \
                hipGetErrorString(e_));                                                 \
      goto done;                                                                        \
    }                                                                                   \
  } while (0)
  {
    SS_HIP(hipMalloc((void **)&level, sizeof(int) * (size_t)n));
    SS_HIP(hipMalloc((void **)&changed, sizeof(int)));
    SS_HIP(hipMemsetAsync(level, 0, sizeof(int) * (size_t)n, stream()));
    // in-place relaxation: monotone, converges to the longest-path level; several passes between
    // host checks (a pass usually propagates many levels because waves start in row order)
    int passes = 0;
    for (;;) {
      SS_HIP(hipMemsetAsync(changed, 0, sizeof(int), stream()));
      for (int p = 0; p < 8; ++p)
        hipLaunchKernelGGL(level_pass_kernel, dim3(grid), dim3(256), 0, stream(), n, F->ind, F->col, dir, level,
                           changed);
      SS_HIP(hipGetLastError());
      int h = 0;
      SS_HIP(hipMemcpyAsync(&h, changed, sizeof(int), hipMemcpyDeviceToHost, stream()));
      SS_HIP(hipStreamSynchronize(stream()));
      passes += 8;
      if (!h) break;
      if (passes > 8 * (n / 8 + 2)) {
        rc = fail(PSP_EINVAL, "ssor: level computation did not converge");
        goto done;
      }
    }
    SS_HIP(hipMalloc((void **)&keys, sizeof(int) * (size_t)n));
    SS_HIP(hipMalloc((void **)&vals, sizeof(int) * (size_t)n));
    SS_HIP(hipMalloc((void **)&rows, sizeof(int) * (size_t)n));
    hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(256), 0, stream(), n, vals);
    size_t bytes = 0;
    SS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, level, keys, vals, rows, n, 0, 32, stream()));
    SS_HIP(hipMalloc(&tmp, bytes ? bytes : 1));
    SS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, level, keys, vals, rows, n, 0, 32, stream()));  // stable
    int maxlev = 0;
    SS_HIP(hipMemcpyAsync(&maxlev, keys + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream()));
    SS_HIP(hipStreamSynchronize(stream()));
    const int nlev = maxlev + 1;
    SS_HIP(hipMalloc((void **)&dptr, sizeof(int) * ((size_t)nlev + 1)));
    hipLaunchKernelGGL(level_ptr_kernel, dim3(grid), dim3(256), 0, stream(), n, keys, dptr);
    SS_HIP(hipGetLastError());
    ptr_out->assign((size_t)nlev + 1, 0);
    SS_HIP(hipMemcpyAsync(ptr_out->data(), dptr, sizeof(int) * (size_t)nlev, hipMemcpyDeviceToHost, stream()));
    SS_HIP(hipStreamSynchronize(stream()));
    (*ptr_out)[(size_t)nlev] = n;
    *rows_out = rows;
    rows = nullptr;
  }
done:
#undef SS_HIP
  (void)hipFree(level);
  (void)hipFree(changed);
  (void)hipFree(keys);
  (void)hipFree(vals);
  (void)hipFree(rows);
  (void)hipFree(dptr);
  (void)hipFree(tmp);
  return rc;
}

}  // namespace

namespace psp {

template <int KIND>
static void sweep(const psp_ssor *K, const std::vector<int> &ptr, const int *dptr, const int *rows,
                  const double *b, double *x, double *y, int first) {
  const psp_csr *F = K->S->full;
  const double *da = K->S->diag;
  const int nl = (int)ptr.size() - 1;
  int l = 0;
  while (l < nl) {
    int e = l;  // maximal run of small levels starting at l
    while (e < nl && ptr[e + 1] - ptr[e] <= kSmallLevel) ++e;
    if (e - l >= 2) {
      hipLaunchKernelGGL(ssor_levels_kernel<KIND>, dim3(1), dim3(256), 0, stream(), l, e, dptr, rows, F->ind,
                         F->col, F->val, da, b, x, y, K->omega, first);
      l = e;
      continue;
    }
    const int a = ptr[l], cnt = ptr[l + 1] - a;
    if (cnt > 0)
      hipLaunchKernelGGL(ssor_level_kernel<KIND>, dim3((cnt + 255) / 256), dim3(256), 0, stream(), cnt, rows + a,
                         F->ind, F->col, F->val, da, b, x, y, K->omega, first);
    ++l;
  }
}

int ssor_apply_dev(psp_ssor *K, const double *b, double *x) {
  double *y = K->temp;
  const bool gs = K->omega == 1.0;
  if (gs) PSP_HIP(hipMemsetAsync(y, 0, sizeof(double) * (size_t)K->n, stream()));  // :164-165
  for (int step = 0; step < K->steps; ++step) {
    if (gs) {
      sweep<0>(K, K->ptr_f, K->dptr_f, K->rows_f, b, x, y, 0);
      sweep<1>(K, K->ptr_b, K->dptr_b, K->rows_b, b, x, y, 0);
    } else {
      sweep<2>(K, K->ptr_f, K->dptr_f, K->rows_f, b, x, y, step == 0 ? 1 : 0);
      sweep<3>(K, K->ptr_b, K->dptr_b, K->rows_b, b, x, y, 0);
    }
    PSP_LAUNCH_CHECK();
  }
  return PSP_OK;
}

}  // namespace psp

extern "C" {

int psp_ssor_create(psp_sss_t *S, double omega, int steps, psp_ssor_t **out) {
  PSP_API_GUARD;
  if (!S || !out) return fail(PSP_EINVAL, "psp_ssor_create: NULL argument");
  if (steps < 0) return fail(PSP_EINVAL, "ssor: steps must be >= 0");
  PSP_TRY(ensure_device());
  psp_ssor *K = new psp_ssor();
  K->n = S->n;
  K->omega = omega;
  K->steps = steps;
  K->S = S;
  int rc = PSP_OK;
  if (S->n > 0) {
    rc = build_schedule(S->full, 0, &K->rows_f, &K->ptr_f);
    if (rc == PSP_OK) rc = build_schedule(S->full, 1, &K->rows_b, &K->ptr_b);
    if (rc == PSP_OK && hipMalloc((void **)&K->temp, sizeof(double) * (size_t)S->n) != hipSuccess)
      rc = fail(PSP_ENOMEM, "ssor: work vector allocation failed");
    if (rc == PSP_OK) {
      const size_t bf = sizeof(int) * K->ptr_f.size(), bb = sizeof(int) * K->ptr_b.size();
      if (hipMalloc((void **)&K->dptr_f, bf) != hipSuccess || hipMalloc((void **)&K->dptr_b, bb) != hipSuccess ||
          hipMemcpy(K->dptr_f, K->ptr_f.data(), bf, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(K->dptr_b, K->ptr_b.data(), bb, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PSP_ENOMEM, "ssor: level table allocation failed");
    }
  } else {
    K->ptr_f.assign(1, 0);
    K->ptr_b.assign(1, 0);
  }
  if (rc != PSP_OK) {
    psp_ssor_destroy(K);
    return rc;
  }
  *out = K;
  return PSP_OK;
}

int psp_ssor_destroy(psp_ssor_t *K) {
  if (!K) return PSP_OK;
  (void)hipFree(K->rows_f);
  (void)hipFree(K->rows_b);
  (void)hipFree(K->dptr_f);
  (void)hipFree(K->dptr_b);
  (void)hipFree(K->temp);
  delete K;
  return PSP_OK;
}

int psp_ssor_info(const psp_ssor_t *K, int *n, int *levels_forward, int *levels_backward) {
  if (!K) return fail(PSP_EINVAL, "psp_ssor_info: NULL handle");
  if (n) *n = K->n;
  if (levels_forward) *levels_forward = (int)K->ptr_f.size() - 1;
  if (levels_backward) *levels_backward = (int)K->ptr_b.size() - 1;
  return PSP_OK;
}

int psp_ssor_precon_dev(psp_ssor_t *K, const double *x_dev, double *y_dev) {
  PSP_API_GUARD;
  if (!K || !x_dev || !y_dev) return fail(PSP_EINVAL, "psp_ssor_precon_dev: NULL argument");
  if (K->n == 0) return PSP_OK;
  return ssor_apply_dev(K, x_dev, y_dev);
}

int psp_ssor_precon(psp_ssor_t *K, const double *x_host, double *y_host) {
  PSP_API_GUARD;
  if (!K || !x_host || !y_host) return fail(PSP_EINVAL, "psp_ssor_precon: NULL argument");
  if (K->n == 0) return PSP_OK;
  PSP_TRY(ensure_device());
  double *x = nullptr, *y = nullptr;
  const size_t bytes = sizeof(double) * (size_t)K->n;
  PSP_HIP(hipMalloc((void **)&x, bytes));
  if (hipMalloc((void **)&y, bytes) != hipSuccess) {
    (void)hipFree(x);
    return fail(PSP_ENOMEM, "psp_ssor_precon: device allocation failed");
  }
  int rc = PSP_OK;
  hipError_t e = hipMemcpyAsync(x, x_host, bytes, hipMemcpyHostToDevice, stream());
  // the reference sweeps start from whatever y holds only for entries it has already written;
  // steps == 0 leaves y untouched, so hand the caller's y through
  if (e == hipSuccess) e = hipMemcpyAsync(y, y_host, bytes, hipMemcpyHostToDevice, stream());
  if (e == hipSuccess) rc = ssor_apply_dev(K, x, y);
  if (e == hipSuccess && rc == PSP_OK) e = hipMemcpyAsync(y_host, y, bytes, hipMemcpyDeviceToHost, stream());
  if (e == hipSuccess) e = hipStreamSynchronize(stream());
  (void)hipFree(x);
  (void)hipFree(y);
  if (rc != PSP_OK) return rc;
  if (e != hipSuccess) return fail(PSP_ENODEV, "psp_ssor_precon: %s", hipGetErrorString(e));
  return PSP_OK;
}

}  // extern "C"
