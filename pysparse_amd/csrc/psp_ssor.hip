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
// One small kernel per level (a 512^3 grid has 1534 levels in each direction).  Round 2: the triangle,
// the diagonal and the vectors are kept in LEVEL ORDER (struct psp_ssor), so a level sweep streams, and
// the launches of one application are replayed from a hipGraph; see DESIGN.md.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <initializer_list>
#include <vector>

#include "psp_internal.h"

using namespace psp;

struct psp_ssor {
  int n = 0;
  double omega = 1.0;
  int steps = 1;
  psp_sss *S = nullptr;  // borrowed; the Python object keeps the matrix alive
  // Everything the sweeps touch lives in FORWARD-LEVEL ORDER ("positions"): position t holds row pos2row[t],
  // the rows of forward level l are positions [ptr_f[l], ptr_f[l+1]).  A level's rows, their matrix entries
  // and their vector entries are then contiguous, so a level sweep streams instead of gathering 36-byte
  // row fragments from all over the triangle (round 1: 85 ms per application at 512^3, dependency- AND
  // gather-bound).  The triangle is copied twice in that order:
  //   forward : f_ptr / f_val / f_pos = the strict lower entries of each position's row, in storage order
  //             (ascending original column), columns as positions;
  //   backward: rows in (backward level, row) order; b_row[u] = position of the u-th row, b_ptr / b_val /
  //             b_pos = its mirrored upper entries from the LAST to the first (the order in which the CPU's
  //             descending sweep scatters into it, preconmodule.c:186-193).
  int *pos2row = nullptr, *row2pos = nullptr;
  std::vector<int> ptr_f, ptr_b;  // level l = positions / backward slots [ptr[l], ptr[l+1])
  int *dptr_f = nullptr, *dptr_b = nullptr;
  int *f_ptr = nullptr, *f_pos = nullptr;
  double *f_val = nullptr;
  int *b_row = nullptr, *b_ptr = nullptr, *b_pos = nullptr;
  double *b_val = nullptr;
  // Rows with at most 8 entries per sweep (every stencil) use a padded slot-major (ELL) form instead of the
  // ptr / val / pos triple: entry s of slot u at [s*n + u], counts in one byte per slot.  A level kernel is
  // bound by its chain of dependent loads, not by bytes (a level of the 512^3 operator is ~1 us of streaming):
  // with the entries at computable addresses the chain is two loads deep (entries -> x) instead of three
  // (ptr -> entries -> x; four in the backward sweep, whose slots are rows by indirection).
  int ell_f = 0, ell_b = 0;  // width (0: CSR form)
  unsigned char *fc8 = nullptr, *bc8 = nullptr;
  double *da = nullptr;                 // diagonal by position
  double *bp = nullptr, *xp = nullptr;  // right-hand side and iterate by position
  double *temp = nullptr;               // y (symgs) / h (ssor) by position
  // the level launches of one application, captured once (they only touch the buffers above)
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t cap_stream = nullptr;
  int graph_state = -1;  // -1 not tried, 0 unavailable (direct launches), 1 captured
  // PSP_DEVICE=cpu (psp_cpu.hip): the reference's two sequential sweeps on the host arrays of S; two n-vectors of work
  bool host = false;
  std::vector<double> h_temp, h_temp2;
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

// ---- the same levels by Kahn's algorithm, one launch per level (O(nnz) work in total instead of one sweep over
// the whole matrix per relaxation pass).  The pattern is structurally symmetric (F is the mirror of an sss_mat), so
// the rows that depend on row u are the entries on the OTHER side of the diagonal in row u itself.
// deps[i] = number of entries row i waits for; the rows with none form level 0
__global__ __launch_bounds__(256) void kahn_init_kernel(int n, const int *__restrict__ ind, const int *__restrict__ col,
                                                        int dir, int *__restrict__ deps, int *__restrict__ level,
                                                        int *__restrict__ front, int *__restrict__ cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool ready = false;
  if (i < n) {
    int d = 0;
    for (int k = ind[i]; k < ind[i + 1]; ++k) {
      const int j = col[k];
      d += (dir == 0 ? j < i : j > i) ? 1 : 0;
    }
    deps[i] = d;
    ready = d == 0;
    if (ready) level[i] = 0;
  }
  // one atomic per wave: the order inside a level does not matter (the rows are sorted by (level, row) afterwards)
  const unsigned long long m = __ballot(ready);
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(cnt, __popcll(m));
  base = __shfl(base, 0, 64);
  if (ready) front[base + __popcll(m & ((1ull << lane) - 1))] = i;
}

// level l -> l + 1: every row u of the frontier releases the rows waiting for it; a row whose last dependency
// this was joins the next frontier with level l + 1.  cnt[0] = size of this frontier, cnt[1] of the next.
__global__ __launch_bounds__(256) void kahn_level_kernel(const int *__restrict__ ind, const int *__restrict__ col,
                                                         int dir, int l, const int *__restrict__ front,
                                                         int *__restrict__ next, int *cnt, int *deps,
                                                         int *__restrict__ level) {
  const int nf = cnt[0];
  const int lane = threadIdx.x & 63;
  // 8 lanes per frontier row
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; (t >> 3) < (((long)nf + 7) & ~7L); t += (long)gridDim.x * 256) {
    const long f = t >> 3;
    const int sub = (int)(t & 7);
    int k0 = 0, k1 = 0, u = 0;
    if (f < nf) {
      u = front[f];
      k0 = ind[u];
      k1 = ind[u + 1];
    }
    const int span = k1 - k0;
    int most = span;  // longest row among the 8 rows of this wave: all lanes walk the same number of steps
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(most, off, 64);
      most = o > most ? o : most;
    }
    for (int s = sub; s < most + sub; s += 8) {
      bool ready = false;
      int v = 0;
      if (s < span) {
        v = col[k0 + s];
        if (dir == 0 ? v > u : v < u) ready = atomicSub(deps + v, 1) == 1;
      }
      if (ready) level[v] = l + 1;
      const unsigned long long m = __ballot(ready);
      int base = 0;
      if (m) {
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(cnt + 1, __popcll(m));
        base = __shfl(base, leader, 64);
        if (ready) next[base + __popcll(m & ((1ull << lane) - 1))] = v;
      }
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

// ---- one row of each sweep (KIND 0: symgs forward, 1: symgs backward, 2: ssor forward, 3: ssor backward),
// everything by position.  symgs_kernel, preconmodule.c:149-193 (omega == 1):
//   forward : s = sum_{lower} va*x[j]; x[i] = (b[i] - y[i] - s)/da[i]; y[i] = s          (:171-179)
//   backward: x[i] holds y of the forward step ("x[k] = y[k]", :182-185), y[i] is rebuilt from the
//             rows above in descending order, x[i] = (b[i] - x[i] - y[i]) / da[i]         (:186-193)
// ssor_kernel, preconmodule.c:95-143 (omega != 1): temp / h as at :110-140
// `t` = the row's position (vectors), [k0, k1) = its entries in the sweep's own copy of the triangle
template <int KIND>
__device__ __forceinline__ void ssor_row(int t, int k0, int k1, const int *__restrict__ pos,
                                         const double *__restrict__ val, const double *__restrict__ da,
                                         const double *__restrict__ b, double *x, double *y, double omega,
                                         int first) {
  if constexpr (KIND == 0) {
    double s = 0.0;
    for (int k = k0; k < k1; ++k) s += val[k] * x[pos[k]];
    x[t] = (b[t] - y[t] - s) / da[t];
    y[t] = s;
  } else if constexpr (KIND == 1) {
    const double yf = y[t];
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) acc += val[k] * x[pos[k]];
    x[t] = (b[t] - yf - acc) / da[t];
    y[t] = acc;
  } else if constexpr (KIND == 2) {
    const double temp = first ? omega * b[t] : (1.0 - omega) * x[t] * da[t] + y[t] + omega * b[t];
    double s = 0.0;
    for (int k = k0; k < k1; ++k) s -= val[k] * x[pos[k]];
    const double hi = omega * s;
    y[t] = hi;
    x[t] = (temp + hi) / da[t];
  } else {
    const double temp = (1.0 - omega) * x[t] * da[t] + y[t] + omega * b[t];
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) acc -= val[k] * x[pos[k]];
    const double hi = omega * acc;
    y[t] = hi;
    x[t] = (temp + hi) / da[t];
  }
}

// slots [a, e) of ONE level (independent rows).  Forward sweeps: slot == position (rowmap == nullptr);
// backward sweeps: rowmap[slot] = position
template <int KIND>
__global__ void ssor_level_kernel(int a, int e, const int *__restrict__ rowmap, const int *__restrict__ ptr,
                                  const int *__restrict__ pos, const double *__restrict__ val,
                                  const double *__restrict__ da, const double *__restrict__ b, double *x,
                                  double *y, double omega, int first) {
  const int u = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (u < e) ssor_row<KIND>(rowmap ? rowmap[u] : u, ptr[u], ptr[u + 1], pos, val, da, b, x, y, omega, first);
}

// a RUN of small levels [l0, l1) in one launch: a single workgroup walks them with a barrier in
// between (stores of one level are visible to the whole workgroup after __syncthreads()).  Small
// problems and the thin ends of big ones are launch-bound otherwise: poisson2d(100) has 199 levels
// of at most 100 rows in each direction.
constexpr int kSmallLevel = 256;  // measured: 2048 made poisson2d(2048) 3x slower (one CU walks 18 us levels)
template <int KIND>
__global__ __launch_bounds__(256) void ssor_levels_kernel(int l0, int l1, const int *__restrict__ lptr,
                                                          const int *__restrict__ rowmap, const int *__restrict__ ptr,
                                                          const int *__restrict__ pos, const double *__restrict__ val,
                                                          const double *__restrict__ da, const double *__restrict__ b,
                                                          double *x, double *y, double omega, int first) {
  for (int l = l0; l < l1; ++l) {
    const int a = lptr[l], e = lptr[l + 1];
    for (int u = a + (int)threadIdx.x; u < e; u += (int)blockDim.x)
      ssor_row<KIND>(rowmap ? rowmap[u] : u, ptr[u], ptr[u + 1], pos, val, da, b, x, y, omega, first);
    __syncthreads();
  }
}

// ---- the same rows with the entries in padded slot-major form (W <= 8 entries per slot): all entry loads are
// issued before anything depends on them; padding slots are loaded (valid addresses) but never added
template <int KIND, int W>
__device__ __forceinline__ void ssor_row_ell(int u, int t, int n, const unsigned char *__restrict__ cnt8,
                                             const int *__restrict__ pos, const double *__restrict__ val,
                                             const double *__restrict__ da, const double *__restrict__ b, double *x,
                                             double *y, double omega, int first) {
  const int cnt = cnt8[u];
  double v[W];
  int p[W];
#pragma unroll
  for (int s = 0; s < W; ++s) {
    v[s] = val[(size_t)s * n + u];
    p[s] = pos[(size_t)s * n + u];
  }
  double xs[W];
#pragma unroll
  for (int s = 0; s < W; ++s) xs[s] = x[p[s]];
  const double bt = b[t], dt = da[t], yt = y[t];
  if constexpr (KIND == 0) {
    double s_ = 0.0;
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const double tt = s_ + v[s] * xs[s];
      s_ = s < cnt ? tt : s_;
    }
    x[t] = (bt - yt - s_) / dt;
    y[t] = s_;
  } else if constexpr (KIND == 1) {
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const double tt = acc + v[s] * xs[s];
      acc = s < cnt ? tt : acc;
    }
    x[t] = (bt - yt - acc) / dt;
    y[t] = acc;
  } else {
    const double xt = (KIND == 2 && first) ? 0.0 : x[t];
    const double temp = (KIND == 2 && first) ? omega * bt : (1.0 - omega) * xt * dt + yt + omega * bt;
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const double tt = acc - v[s] * xs[s];
      acc = s < cnt ? tt : acc;
    }
    const double hi = omega * acc;
    y[t] = hi;
    x[t] = (temp + hi) / dt;
  }
}

template <int KIND, int W>
__global__ void ssor_level_ell_kernel(int a, int e, int n, const int *__restrict__ rowmap,
                                      const unsigned char *__restrict__ cnt8, const int *__restrict__ pos,
                                      const double *__restrict__ val, const double *__restrict__ da,
                                      const double *__restrict__ b, double *x, double *y, double omega, int first) {
  const int u = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (u < e) ssor_row_ell<KIND, W>(u, rowmap ? rowmap[u] : u, n, cnt8, pos, val, da, b, x, y, omega, first);
}

template <int KIND, int W>
__global__ __launch_bounds__(256) void ssor_levels_ell_kernel(int l0, int l1, int n, const int *__restrict__ lptr,
                                                              const int *__restrict__ rowmap,
                                                              const unsigned char *__restrict__ cnt8,
                                                              const int *__restrict__ pos,
                                                              const double *__restrict__ val,
                                                              const double *__restrict__ da,
                                                              const double *__restrict__ b, double *x, double *y,
                                                              double omega, int first) {
  for (int l = l0; l < l1; ++l) {
    const int a = lptr[l], e = lptr[l + 1];
    for (int u = a + (int)threadIdx.x; u < e; u += (int)blockDim.x)
      ssor_row_ell<KIND, W>(u, rowmap ? rowmap[u] : u, n, cnt8, pos, val, da, b, x, y, omega, first);
    __syncthreads();
  }
}

__global__ void row_len_kernel(int n, const int *__restrict__ ptr, int *__restrict__ len) {
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) len[u] = ptr[u + 1] - ptr[u];
}

__global__ void max_int_kernel(int n, const int *__restrict__ v, int *__restrict__ out) {
  int m = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, v[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// CSR-form copy (ptr / pos / val by slot) -> padded slot-major form of width W
__global__ void ssor_to_ell_kernel(int n, int W, const int *__restrict__ ptr, const int *__restrict__ pos,
                                   const double *__restrict__ val, unsigned char *__restrict__ cnt8,
                                   int *__restrict__ epos, double *__restrict__ eval) {
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
    const int a = ptr[u], c = ptr[u + 1] - a;
    cnt8[u] = (unsigned char)c;
    for (int s = 0; s < W; ++s) {
      epos[(size_t)s * n + u] = s < c ? pos[a + s] : 0;
      eval[(size_t)s * n + u] = s < c ? val[a + s] : 0.0;
    }
  }
}

// ---- builders of the level-ordered copies
__global__ void invert_perm_kernel(int n, const int *__restrict__ perm, int *__restrict__ inv) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) inv[perm[i]] = i;
}

// forward: entries of slot u = the strict lower entries of row rows[u] (the sss arrays hold exactly those,
// ascending column).  count pass (cnt[u]) or, with ptr, fill pass.
__global__ void ssor_fill_forward_kernel(int n, const int *__restrict__ rows, const int *__restrict__ sind,
                                         const int *__restrict__ scol, const double *__restrict__ sval,
                                         const int *__restrict__ row2pos, int *__restrict__ cnt,
                                         const int *__restrict__ ptr, int *__restrict__ pos,
                                         double *__restrict__ val) {
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
    const int r = rows[u];
    const int a = sind[r], e = sind[r + 1];
    if (cnt) {
      cnt[u] = e - a;
    } else {
      int o = ptr[u];
      for (int k = a; k < e; ++k, ++o) {
        pos[o] = row2pos[scol[k]];
        val[o] = sval[k];
      }
    }
  }
}

// backward: entries of slot u = the entries (r, j > r) of the full mirror's row r = rows[u], LAST to first
__global__ void ssor_fill_backward_kernel(int n, const int *__restrict__ rows, const int *__restrict__ find,
                                          const int *__restrict__ fcol, const double *__restrict__ fval,
                                          const int *__restrict__ row2pos, int *__restrict__ cnt,
                                          const int *__restrict__ ptr, int *__restrict__ pos,
                                          double *__restrict__ val, int *__restrict__ b_row) {
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
    const int r = rows[u];
    const int a = find[r], e = find[r + 1];
    if (cnt) {
      int c = 0;
      for (int k = e - 1; k >= a && fcol[k] > r; --k) ++c;
      cnt[u] = c;
      b_row[u] = row2pos[r];
    } else {
      int o = ptr[u];
      for (int k = e - 1; k >= a && fcol[k] > r; --k, ++o) {
        pos[o] = row2pos[fcol[k]];
        val[o] = fval[k];
      }
    }
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
    // Kahn's algorithm, one launch per level, kLevBatch levels between two looks at the counters
    // (PSP_SSOR_KAHN=0: the relaxation sweeps below, the round-1 method -- same levels)
    static const bool kahn = [] {
      const char *e = psp::tuning_env("PSP_SSOR_KAHN");
      return e ? atoi(e) != 0 : true;
    }();
    bool have_levels = false;
    if (kahn && n > 0) {
      constexpr int kLevBatch = 64;
      const long maxlev = std::min<long>(n, 1L << 22);
      int *deps = nullptr, *fr[2] = {nullptr, nullptr}, *cnt = nullptr;
      bool ok = hipMalloc((void **)&deps, sizeof(int) * (size_t)n) == hipSuccess &&
                hipMalloc((void **)&fr[0], sizeof(int) * (size_t)n) == hipSuccess &&
                hipMalloc((void **)&fr[1], sizeof(int) * (size_t)n) == hipSuccess &&
                hipMalloc((void **)&cnt, sizeof(int) * (size_t)(maxlev + kLevBatch + 2)) == hipSuccess &&
                hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)(maxlev + kLevBatch + 2), stream()) == hipSuccess;
      if (ok) {
        hipLaunchKernelGGL(kahn_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream(), n, F->ind, F->col, dir, deps,
                           level, fr[0], cnt);
        long l = 0, done_rows = 0;
        int sizes[kLevBatch];
        for (;;) {
          if (l + kLevBatch > maxlev) {
            ok = false;  // deeper than the counter table: the relaxation below takes over
            break;
          }
          for (int b = 0; b < kLevBatch; ++b)
            hipLaunchKernelGGL(kahn_level_kernel, dim3(1024), dim3(256), 0, stream(), F->ind, F->col, dir, (int)(l + b),
                               fr[(l + b) & 1], fr[(l + b + 1) & 1], cnt + l + b, deps, level);
          if (hipGetLastError() != hipSuccess ||
              hipMemcpyAsync(sizes, cnt + l, sizeof(sizes), hipMemcpyDeviceToHost, stream()) != hipSuccess ||
              hipStreamSynchronize(stream()) != hipSuccess) {
            ok = false;
            break;
          }
          int b = 0;
          while (b < kLevBatch && sizes[b] > 0) done_rows += sizes[b++];
          if (b < kLevBatch) break;  // frontier l + b is empty
          l += kLevBatch;
        }
        have_levels = ok && done_rows == n;  // a cycle (not a triangle) or a failure: fall through
      }
      (void)hipGetLastError();
      if (deps) (void)hipFree(deps);
      if (fr[0]) (void)hipFree(fr[0]);
      if (fr[1]) (void)hipFree(fr[1]);
      if (cnt) (void)hipFree(cnt);
      if (!have_levels) SS_HIP(hipMemsetAsync(level, 0, sizeof(int) * (size_t)n, stream()));
    }
    int passes = 0;
    for (; !have_levels;) {
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

int reorder_gather(int n, const int *perm_dev, const double *x, double *xp, const int *skip);  // psp_reorder.hip

// the launches of one sweep on stream st, everything by position
template <int KIND, int W>
static void sweep_w(const psp_ssor *K, hipStream_t st, bool forward, int first) {
  const std::vector<int> &lp = forward ? K->ptr_f : K->ptr_b;
  const int *dlp = forward ? K->dptr_f : K->dptr_b;
  const int *rowmap = forward ? nullptr : K->b_row;
  const int *ptr = forward ? K->f_ptr : K->b_ptr;
  const int *pos = forward ? K->f_pos : K->b_pos;
  const double *val = forward ? K->f_val : K->b_val;
  const unsigned char *c8 = forward ? K->fc8 : K->bc8;
  const int nl = (int)lp.size() - 1;
  int l = 0;
  while (l < nl) {
    int e = l;  // maximal run of small levels starting at l
    while (e < nl && lp[e + 1] - lp[e] <= kSmallLevel) ++e;
    if (e - l >= 2) {
      if constexpr (W > 0)
        hipLaunchKernelGGL((ssor_levels_ell_kernel<KIND, W>), dim3(1), dim3(256), 0, st, l, e, K->n, dlp, rowmap, c8,
                           pos, val, K->da, K->bp, K->xp, K->temp, K->omega, first);
      else
        hipLaunchKernelGGL(ssor_levels_kernel<KIND>, dim3(1), dim3(256), 0, st, l, e, dlp, rowmap, ptr, pos, val,
                           K->da, K->bp, K->xp, K->temp, K->omega, first);
      l = e;
      continue;
    }
    const int a = lp[l], cnt = lp[l + 1] - a;
    if (cnt > 0) {
      if constexpr (W > 0)
        hipLaunchKernelGGL((ssor_level_ell_kernel<KIND, W>), dim3((cnt + 255) / 256), dim3(256), 0, st, a, a + cnt,
                           K->n, rowmap, c8, pos, val, K->da, K->bp, K->xp, K->temp, K->omega, first);
      else
        hipLaunchKernelGGL(ssor_level_kernel<KIND>, dim3((cnt + 255) / 256), dim3(256), 0, st, a, a + cnt, rowmap,
                           ptr, pos, val, K->da, K->bp, K->xp, K->temp, K->omega, first);
    }
    ++l;
  }
}

template <int KIND>
static void sweep(const psp_ssor *K, hipStream_t st, bool forward, int first) {
  switch (forward ? K->ell_f : K->ell_b) {
    case 1: sweep_w<KIND, 1>(K, st, forward, first); break;
    case 2: sweep_w<KIND, 2>(K, st, forward, first); break;
    case 3: sweep_w<KIND, 3>(K, st, forward, first); break;
    case 4: sweep_w<KIND, 4>(K, st, forward, first); break;
    case 6: sweep_w<KIND, 6>(K, st, forward, first); break;
    case 8: sweep_w<KIND, 8>(K, st, forward, first); break;
    default: sweep_w<KIND, 0>(K, st, forward, first); break;
  }
}

// all sweeps of one application (they only touch K's own buffers)
static void enqueue_sweeps(const psp_ssor *K, hipStream_t st) {
  const bool gs = K->omega == 1.0;
  for (int step = 0; step < K->steps; ++step) {
    if (gs) {
      sweep<0>(K, st, true, 0);
      sweep<1>(K, st, false, 0);
    } else {
      sweep<2>(K, st, true, step == 0 ? 1 : 0);
      sweep<3>(K, st, false, 0);
    }
  }
}

// One application = thousands of tiny dependent launches (512^3: 2 x 1534 levels of ~1 us of streaming
// each): issued one by one the host is the bound (~3.5 us per launch), replayed from a hipGraph the device's
// own dependent-launch latency is (~1.5-2 us).  Captured once per handle; PSP_SSOR_GRAPH=0 keeps direct launches.
static void ensure_graph(psp_ssor *K) {
  if (K->graph_state >= 0) return;
  K->graph_state = 0;
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_SSOR_GRAPH");
    return e && atoi(e) == 0;
  }();
  const long launches = (long)K->steps * ((long)K->ptr_f.size() + (long)K->ptr_b.size());
  if (off || launches < 8) return;
  if (hipStreamCreateWithFlags(&K->cap_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    K->cap_stream = nullptr;
    return;
  }
  bool ok = hipStreamBeginCapture(K->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (ok) {
    enqueue_sweeps(K, K->cap_stream);
    ok = hipStreamEndCapture(K->cap_stream, &K->graph) == hipSuccess && K->graph != nullptr &&
         hipGraphInstantiate(&K->exec, K->graph, nullptr, nullptr, 0) == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    if (K->graph) (void)hipGraphDestroy(K->graph);
    K->graph = nullptr;
    K->exec = nullptr;
    return;
  }
  K->graph_state = 1;
}

int ssor_apply_dev(psp_ssor *K, const double *b, double *x) {
  if (K->steps <= 0) return PSP_OK;  // the reference leaves y untouched
  PSP_TRY(reorder_gather(K->n, K->pos2row, b, K->bp, nullptr));  // bp[t] = b[pos2row[t]]
  if (K->omega == 1.0) PSP_HIP(hipMemsetAsync(K->temp, 0, sizeof(double) * (size_t)K->n, stream()));  // :164-165
  ensure_graph(K);
  if (K->graph_state == 1 && hipGraphLaunch(K->exec, stream()) != hipSuccess) {
    (void)hipGetLastError();
    K->graph_state = 0;  // this runtime does not replay into the library's stream: direct launches from now on
  }
  if (K->graph_state != 1) {
    enqueue_sweeps(K, stream());
    PSP_LAUNCH_CHECK();
  }
  return reorder_gather(K->n, K->row2pos, K->xp, x, nullptr);  // x[i] = xp[row2pos[i]]
}

}  // namespace psp

namespace {

template <typename T>
int dev_alloc(T **p, size_t count) {
  PSP_HIP(hipMalloc((void **)p, sizeof(T) * (count ? count : 1)));
  return PSP_OK;
}

// exclusive prefix sum of cnt[0..n) into ptr[0..n]
int exclusive_scan(int n, const int *cnt, int *ptr, long *total) {
  size_t bytes = 0;
  void *tmp = nullptr;
  PSP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cnt, ptr, n, stream()));
  PSP_HIP(hipMalloc(&tmp, bytes ? bytes : 1));
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, bytes, cnt, ptr, n, stream());
  int last_ptr = 0, last_cnt = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&last_ptr, ptr + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream());
  if (e == hipSuccess) e = hipMemcpyAsync(&last_cnt, cnt + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream());
  if (e == hipSuccess) e = hipStreamSynchronize(stream());
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(PSP_ENODEV, "ssor: scan failed: %s", hipGetErrorString(e));
  *total = (long)last_ptr + last_cnt;
  const int tot = (int)*total;
  PSP_HIP(hipMemcpy(ptr + n, &tot, sizeof(int), hipMemcpyHostToDevice));
  return PSP_OK;
}

// the level-ordered copies of the triangle (see struct psp_ssor)
int build_level_ordered(psp_ssor *K, int *rows_f, int *rows_b) {
  const psp_sss *S = K->S;
  const psp_csr *F = S->full;
  const int n = K->n;
  const int grid = std::min((n + 255) / 256, 65536);
  K->pos2row = rows_f;  // takes ownership
  PSP_TRY(dev_alloc(&K->row2pos, n));
  hipLaunchKernelGGL(invert_perm_kernel, dim3(grid), dim3(256), 0, stream(), n, K->pos2row, K->row2pos);
  PSP_LAUNCH_CHECK();
  int *cnt = nullptr;
  PSP_TRY(dev_alloc(&cnt, n));
  int rc = PSP_OK;
  long tot = 0;
  // forward
  hipLaunchKernelGGL(ssor_fill_forward_kernel, dim3(grid), dim3(256), 0, stream(), n, K->pos2row, S->ind, S->col, S->val,
                     K->row2pos, cnt, (const int *)nullptr, (int *)nullptr, (double *)nullptr);
  if (rc == PSP_OK) rc = dev_alloc(&K->f_ptr, (size_t)n + 1);
  if (rc == PSP_OK) rc = exclusive_scan(n, cnt, K->f_ptr, &tot);
  if (rc == PSP_OK) rc = dev_alloc(&K->f_pos, (size_t)tot);
  if (rc == PSP_OK) rc = dev_alloc(&K->f_val, (size_t)tot);
  if (rc == PSP_OK)
    hipLaunchKernelGGL(ssor_fill_forward_kernel, dim3(grid), dim3(256), 0, stream(), n, K->pos2row, S->ind, S->col,
                       S->val, K->row2pos, (int *)nullptr, K->f_ptr, K->f_pos, K->f_val);
  // backward
  if (rc == PSP_OK) rc = dev_alloc(&K->b_row, n);
  if (rc == PSP_OK)
    hipLaunchKernelGGL(ssor_fill_backward_kernel, dim3(grid), dim3(256), 0, stream(), n, rows_b, F->ind, F->col, F->val,
                       K->row2pos, cnt, (const int *)nullptr, (int *)nullptr, (double *)nullptr, K->b_row);
  if (rc == PSP_OK) rc = dev_alloc(&K->b_ptr, (size_t)n + 1);
  if (rc == PSP_OK) rc = exclusive_scan(n, cnt, K->b_ptr, &tot);
  if (rc == PSP_OK) rc = dev_alloc(&K->b_pos, (size_t)tot);
  if (rc == PSP_OK) rc = dev_alloc(&K->b_val, (size_t)tot);
  if (rc == PSP_OK)
    hipLaunchKernelGGL(ssor_fill_backward_kernel, dim3(grid), dim3(256), 0, stream(), n, rows_b, F->ind, F->col, F->val,
                       K->row2pos, (int *)nullptr, K->b_ptr, K->b_pos, K->b_val, (int *)nullptr);
  // narrow rows: padded slot-major form, the ptr / pos / val triple is dropped
  if (rc == PSP_OK) {
    for (int dir = 0; dir < 2 && rc == PSP_OK; ++dir) {
      int **ptr = dir ? &K->b_ptr : &K->f_ptr, **pos = dir ? &K->b_pos : &K->f_pos;
      double **val = dir ? &K->b_val : &K->f_val;
      int *dmax = nullptr, maxc = 0;
      rc = dev_alloc(&dmax, 1);
      if (rc != PSP_OK) break;
      (void)hipMemsetAsync(dmax, 0, sizeof(int), stream());
      // widest row = largest difference of consecutive offsets
      hipLaunchKernelGGL(row_len_kernel, dim3(grid), dim3(256), 0, stream(), n, *ptr, cnt);
      hipLaunchKernelGGL(max_int_kernel, dim3(std::min(grid, 2048)), dim3(256), 0, stream(), n, cnt, dmax);
      if (hipMemcpy(&maxc, dmax, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(PSP_ENODEV, "ssor: max");
      (void)hipFree(dmax);
      if (rc != PSP_OK) break;
      static const bool ell_off = [] {
        const char *e = psp::tuning_env("PSP_SSOR_ELL");
        return e && atoi(e) == 0;
      }();
      int W = 0;
      for (int w : {1, 2, 3, 4, 6, 8})
        if (maxc <= w) {
          W = w;
          break;
        }
      if (maxc == 0 || ell_off) W = 0;
      if (W == 0) continue;
      int *epos = nullptr;
      double *eval = nullptr;
      unsigned char *c8 = nullptr;
      if (hipMalloc((void **)&epos, sizeof(int) * (size_t)W * n) != hipSuccess ||
          hipMalloc((void **)&eval, sizeof(double) * (size_t)W * n) != hipSuccess ||
          hipMalloc((void **)&c8, (size_t)n) != hipSuccess) {  // no room: keep the CSR form
        (void)hipGetLastError();
        (void)hipFree(epos);
        (void)hipFree(eval);
        (void)hipFree(c8);
        continue;
      }
      hipLaunchKernelGGL(ssor_to_ell_kernel, dim3(grid), dim3(256), 0, stream(), n, W, *ptr, *pos, *val, c8, epos, eval);
      if (hipStreamSynchronize(stream()) != hipSuccess) rc = fail(PSP_ENODEV, "ssor: ell build failed");
      (void)hipFree(*ptr);
      (void)hipFree(*pos);
      (void)hipFree(*val);
      *ptr = nullptr;
      *pos = epos;
      *val = eval;
      (dir ? K->bc8 : K->fc8) = c8;
      (dir ? K->ell_b : K->ell_f) = W;
    }
  }
  // vectors by position
  if (rc == PSP_OK) rc = dev_alloc(&K->da, n);
  if (rc == PSP_OK) rc = dev_alloc(&K->bp, n);
  if (rc == PSP_OK) rc = dev_alloc(&K->xp, n);
  if (rc == PSP_OK) rc = dev_alloc(&K->temp, n);
  if (rc == PSP_OK) rc = reorder_gather(n, K->pos2row, S->diag, K->da, nullptr);
  if (rc == PSP_OK && hipStreamSynchronize(stream()) != hipSuccess) rc = fail(PSP_ENODEV, "ssor: build failed");
  if (rc == PSP_OK && hipGetLastError() != hipSuccess) rc = fail(PSP_ENODEV, "ssor: build kernels failed");
  (void)hipFree(cnt);
  return rc;
}

}  // namespace

namespace psp {
// PSP_DEVICE=cpu: SSOR_precon (preconmodule.c:199-223) with its two kernels, symgs_kernel (:149-193, omega == 1) and
// ssor_kernel (:95-146), as sequential sweeps over the host arrays of the sss_mat; x is the output
int ssor_apply_host(psp_ssor *K, const double *b, double *x) {
  const psp_sss *S = K->S;
  const int n = K->n;
  const double *va = S->val, *da = S->diag;
  const int *ja = S->col, *ia = S->ind;
  if (K->omega == 1.0) {
    double *y = K->h_temp.data();
    for (int k = 0; k < n; ++k) y[k] = 0.0;
    for (int step = 0; step < K->steps; ++step) {
      for (int i = 0; i < n; ++i) {  // x = (L + D) \ (b - y), y = L x
        double s = 0.0;
        for (int k = ia[i]; k < ia[i + 1]; ++k) s += va[k] * x[ja[k]];
        x[i] = (b[i] - y[i] - s) / da[i];
        y[i] = s;
      }
      for (int k = 0; k < n; ++k) {
        x[k] = y[k];
        y[k] = 0.0;
      }
      for (int i = n - 1; i >= 0; --i) {  // x = (L^T + D) \ (b - y), y = L^T x
        x[i] = (b[i] - x[i] - y[i]) / da[i];
        const double s = x[i];
        for (int k = ia[i]; k < ia[i + 1]; ++k) y[ja[k]] += va[k] * s;
      }
    }
    return PSP_OK;
  }
  const double omega = K->omega;
  double *temp = K->h_temp.data(), *h = K->h_temp2.data();
  for (int step = 0; step < K->steps; ++step) {
    if (step == 0)
      for (int i = 0; i < n; ++i) temp[i] = omega * b[i];
    else
      for (int i = 0; i < n; ++i) temp[i] = (1.0 - omega) * x[i] * da[i] + h[i] + omega * b[i];
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int k = ia[i]; k < ia[i + 1]; ++k) s -= va[k] * x[ja[k]];
      h[i] = omega * s;
      x[i] = (temp[i] + h[i]) / da[i];
    }
    for (int i = 0; i < n; ++i) {
      temp[i] = (1.0 - omega) * x[i] * da[i] + h[i] + omega * b[i];
      h[i] = 0.0;
    }
    for (int i = n - 1; i >= 0; --i) {
      h[i] = omega * h[i];
      x[i] = (temp[i] + h[i]) / da[i];
      const double s = x[i];
      for (int k = ia[i]; k < ia[i + 1]; ++k) h[ja[k]] -= va[k] * s;
    }
  }
  return PSP_OK;
}
}  // namespace psp

extern "C" {

int psp_ssor_create(psp_sss_t *S, double omega, int steps, psp_ssor_t **out) {
  PSP_API_GUARD;
  if (!S || !out) return fail(PSP_EINVAL, "psp_ssor_create: NULL argument");
  if (steps < 0) return fail(PSP_EINVAL, "ssor: steps must be >= 0");
  if (S->host) {
    psp_ssor *K = new psp_ssor();
    K->n = S->n;
    K->omega = omega;
    K->steps = steps;
    K->S = S;
    K->host = true;
    K->h_temp.assign((size_t)S->n, 0.0);
    K->h_temp2.assign((size_t)S->n, 0.0);
    K->ptr_f.assign(1, 0);
    K->ptr_b.assign(1, 0);
    *out = K;
    return PSP_OK;
  }
  PSP_TRY(ensure_device());
  psp_ssor *K = new psp_ssor();
  K->n = S->n;
  K->omega = omega;
  K->steps = steps;
  K->S = S;
  int rc = PSP_OK;
  if (S->n > 0) {
    int *rows_f = nullptr, *rows_b = nullptr;
    rc = build_schedule(S->full, 0, &rows_f, &K->ptr_f);
    if (rc == PSP_OK) rc = build_schedule(S->full, 1, &rows_b, &K->ptr_b);
    if (rc == PSP_OK) {
      // backward slots are (level, row)-sorted rows; ptr_b indexes slots
      rc = build_level_ordered(K, rows_f, rows_b);
      rows_f = nullptr;  // owned by K now (pos2row), even on failure
    }
    (void)hipFree(rows_f);
    (void)hipFree(rows_b);
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
  if (K->host) {
    delete K;
    return PSP_OK;
  }
  if (K->exec) (void)hipGraphExecDestroy(K->exec);
  if (K->graph) (void)hipGraphDestroy(K->graph);
  if (K->cap_stream) (void)hipStreamDestroy(K->cap_stream);
  for (void *p : {(void *)K->pos2row, (void *)K->row2pos, (void *)K->dptr_f, (void *)K->dptr_b, (void *)K->f_ptr,
                  (void *)K->f_pos, (void *)K->f_val, (void *)K->b_row, (void *)K->b_ptr, (void *)K->b_pos,
                  (void *)K->b_val, (void *)K->da, (void *)K->bp, (void *)K->xp, (void *)K->temp, (void *)K->fc8,
                  (void *)K->bc8})
    (void)hipFree(p);
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
  if (K->host) return psp::ssor_apply_host(K, x_host, y_host);
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
