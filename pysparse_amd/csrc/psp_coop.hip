// psp_coop.hip -- small systems: the whole PCG / MINRES loop as ONE kernel.
//
// A system of 10^4 unknowns (BASELINE.json configs[0]: poisson2d(100)) fits the caches many times over; what its
// iteration costs on the GPU is the chain of ~9 dependent launches of the asynchronous loops (psp_solvers.hip):
// 20.7 us per PCG iteration at 100^2 in round 2, all of it launch latency.  Here the loop of pcg.c:91-163 (resp.
// minres.c:96-193) runs inside one kernel: a handful of co-resident workgroups, each owning a contiguous row range,
// separated by grid-wide barriers where the reference has a data dependency across rows --
//   PCG   : [p update] | [q = A p, p.q] | [stagnation scan, x, r, r.r, r.z]        3 barriers per iteration
//   MINRES: [v = y/beta] | [Av = A v, v.Av] | [Lanczos update, y = K v_hat, v_hat.y] | (w, x: rows of their own)
// -- and every workgroup evaluates the scalar recurrences itself from the same reduced values (the same operations
// on the same bits, so all of them take the same branch and leave the loop together).  A thread owns 1, 2 or 4 rows for
// the whole solve: their matrix entries (up to 8 per row) and their slices of x, r, ... stay in REGISTERS; the only
// vector that crosses workgroups is the one the product gathers (p, resp. v), and the partial sums.
//
// Arithmetic: per element the reference's operations, multiply and add rounded separately (-ffp-contract=off); a row's
// products are added left to right (csr_mat.c:49-54); a reduction is: a thread adds its rows in ascending order, the
// wave a fixed xor tree, the workgroup its waves in order, and the workgroups' partial sums are added in order by
// every workgroup -- fixed for a given matrix size, so runs are bitwise reproducible.  (The order differs from the
// asynchronous loops': iterates agree with theirs to rounding, not bit for bit; counts and goldens are tested.)
//
// Grid barrier (MI355X_MICROARCH.md "Correctness boundaries", cdna_hip_programming.md G16): the per-XCD L2s are not
// coherent with each other, so data handed from one workgroup to another goes through an agent-scope release by every
// storing thread before the workgroup barrier, one arrival counter (device-scope atomics), and an agent-scope acquire
// before the first load after the barrier.  The grid is at most 128 workgroups of 256 threads -- half a workgroup per CU --
// and every spin is bounded: a barrier that does not complete sets an error flag that ends all workgroups.
#include <algorithm>
#include <vector>

#include "psp_internal.h"

namespace psp {

namespace {

constexpr int kCoopMaxRows = 1 << 17;  // beyond that the asynchronous loops (whole-chip kernels) are faster
constexpr int kCoopMaxWg = 128;  // workgroups of 256 threads: far below one per CU slot, all co-resident

struct CoopCtl {
  unsigned count;
  unsigned gen;
  int error;
  int info, iter;
  double relres;
};

__device__ __forceinline__ bool coop_barrier(CoopCtl *c, int nwg, unsigned &gen) {
  if (nwg == 1) {
    __syncthreads();
    return true;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this thread's stores are visible device-wide ...
  __syncthreads();                                    // ... before the workgroup announces itself
  if (threadIdx.x == 0) {
    const unsigned arrived = __hip_atomic_fetch_add(&c->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == (unsigned)nwg - 1u) {
      __hip_atomic_store(&c->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&c->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long spins = 0;
      while (__hip_atomic_load(&c->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        if (++spins > (1L << 26) || __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&c->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  gen += 1;
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // nothing cached from before the barrier is read after it
  return __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// workgroup sum of NV values per thread: wave xor tree, then the waves in order; every thread gets the result
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *sh /* >= 16 * NV + NV */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double s = v[j];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) sh[j * 16 + wid] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += sh[threadIdx.x * 16 + w];
    sh[16 * NV + threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = sh[16 * NV + j];
  __syncthreads();
}

// the workgroups' partial sums (already in part[j * kCoopMaxWg + wg]) added in workgroup order by every workgroup
template <int NV>
__device__ __forceinline__ void grid_sum(double (&v)[NV], const double *part, int nwg, double *sh) {
  if (threadIdx.x < NV) {
    double t = 0.0;
    for (int w = 0; w < nwg; ++w) t += part[threadIdx.x * kCoopMaxWg + w];
    sh[threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = sh[j];
  __syncthreads();
}

constexpr int kRegNz = 8;  // entries of a row kept in registers (every 5- / 7-point row); longer rows are re-read

// The rows a thread owns for the whole solve: row j of thread t in workgroup wg is (wg * blockDim + t) + j * stride,
// stride = nwg * blockDim (consecutive threads on consecutive rows).  The matrix is constant across iterations, so a
// row's entries are loaded ONCE into registers; per iteration a product costs only the gathers of the shared vector.
template <int R>
struct OwnedRows {
  int row[R];
  int cnt[R];    // entries of the row (0 for a slot past the end)
  int first[R];  // ind[row]
  double val[R][kRegNz];
  int col[R][kRegNz];
  __device__ __forceinline__ void load(int n, int nwg, const int *__restrict__ ind, const int *__restrict__ cidx,
                                       const double *__restrict__ v) {
    const int stride = nwg * (int)blockDim.x;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int i = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x + j * stride;
      row[j] = i;
      cnt[j] = 0;
      first[j] = 0;
      if (i < n) {
        first[j] = ind[i];
        cnt[j] = ind[i + 1] - first[j];
      }
#pragma unroll
      for (int k = 0; k < kRegNz; ++k) {
        const bool have = k < cnt[j];
        val[j][k] = have ? v[first[j] + k] : 0.0;
        col[j][k] = have ? cidx[first[j] + k] : 0;
      }
    }
  }
  // csr_mat.c:49-54: the row's stored products added left to right from 0.0 (absent slots are not touched)
  __device__ __forceinline__ double product(int j, const int *__restrict__ cidx, const double *__restrict__ v,
                                            const double *x) const {
    double g[kRegNz];
#pragma unroll
    for (int k = 0; k < kRegNz; ++k) g[k] = k < cnt[j] ? x[col[j][k]] : 0.0;  // independent gathers
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kRegNz; ++k)
      if (k < cnt[j]) s += val[j][k] * g[k];
    for (int k = kRegNz; k < cnt[j]; ++k) s += v[first[j] + k] * x[cidx[first[j] + k]];  // a longer row: from memory
    return s;
  }
};

// pcg.c:91-166 from the head of iteration 1: r = b - A x, rho = r.z and ||r|| > tolb are the caller's.
// x, r and q live in registers for the whole solve; p is the one vector other workgroups read.
template <int R>
__global__ __launch_bounds__(256) void pcg_coop_kernel(int n, int nwg, const int *__restrict__ ind,
                                                       const int *__restrict__ col, const double *__restrict__ val,
                                                       const double *__restrict__ dinv, double *x, const double *r,
                                                       double *p, double n2b, double tolb, double normr0, double rho0,
                                                       int maxit, CoopCtl *ctl, double *part, double *hist) {
  __shared__ double sh[16 * 3 + 8];
  const int wg = blockIdx.x;
  OwnedRows<R> rows;
  rows.load(n, nwg, ind, col, val);
  double xr[R], rr[R], pr[R], qr[R], dr[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const bool in = rows.row[j] < n;
    xr[j] = in ? x[rows.row[j]] : 0.0;
    rr[j] = in ? r[rows.row[j]] : 0.0;
    dr[j] = (in && dinv) ? dinv[rows.row[j]] : 1.0;
    pr[j] = 0.0;
    qr[j] = 0.0;
  }
  unsigned gen = 0;
  double rho = rho0, rho1 = 1.0, normr = normr0, alpha, beta = 0.0;
  int flag = -1, it;
  for (it = 1; it <= maxit; ++it) {
    if (rho == 0.0) {  // pcg.c:101-104
      flag = -2;
      break;
    }
    if (it > 1) {
      beta = rho / rho1;
      if (beta == 0.0) {  // pcg.c:109-112
        flag = -6;
        break;
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {  // pcg.c:93-97, :106, :113-114
      const double z = dinv ? rr[j] * dr[j] : rr[j];
      pr[j] = it == 1 ? z : z + beta * pr[j];
      if (rows.row[j] < n) p[rows.row[j]] = pr[j];
    }
    if (!coop_barrier(ctl, nwg, gen)) return;
    double v1[1] = {0.0};
#pragma unroll
    for (int j = 0; j < R; ++j) {  // pcg.c:116-117
      qr[j] = rows.product(j, col, val, p);
      v1[0] += pr[j] * qr[j];
    }
    block_sum<1>(v1, sh);
    if (nwg > 1) {
      if (threadIdx.x == 0) part[wg] = v1[0];
      if (!coop_barrier(ctl, nwg, gen)) return;
      grid_sum<1>(v1, part, nwg, sh);
    }
    const double pq = v1[0];
    if (pq == 0.0) {  // pcg.c:118-120
      flag = -6;
      break;
    }
    alpha = rho / pq;
    const int stag0 = alpha == 0.0;  // pcg.c:124-125
    double v3[3] = {0.0, 0.0, 0.0};  // r.r, r.z, number of rows that did not stagnate
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (rows.row[j] >= n) continue;
      if (!stag0) {  // pcg.c:127-139 (the scan reads x before the update)
        if (xr[j] != 0.0) {
          if (1.0 + fabs(alpha * pr[j] / xr[j]) != 1.0) v3[2] += 1.0;
        } else if (pr[j] != 0.0) {
          v3[2] += 1.0;
        }
      }
      if (alpha != 0.0) {  // daxpy returns at once for a zero coefficient
        xr[j] = xr[j] + alpha * pr[j];
        rr[j] = rr[j] + (-alpha) * qr[j];
      }
      v3[0] += rr[j] * rr[j];
      v3[1] += rr[j] * (dinv ? rr[j] * dr[j] : rr[j]);
    }
    block_sum<3>(v3, sh);
    if (nwg > 1) {
      if (threadIdx.x < 3) part[(1 + threadIdx.x) * kCoopMaxWg + wg] = v3[threadIdx.x];
      if (!coop_barrier(ctl, nwg, gen)) return;
      grid_sum<3>(v3, part + kCoopMaxWg, nwg, sh);
    }
    normr = sqrt(v3[0]);  // the recurred residual (pcg.c:146-153)
    if (hist && wg == 0 && threadIdx.x == 0) hist[it] = normr;
    if (normr <= tolb) {  // pcg.c:154-157
      flag = 0;
      break;
    }
    if (stag0 || v3[2] == 0.0) {  // pcg.c:159-162
      flag = -5;
      break;
    }
    rho1 = rho;
    rho = v3[1];
  }
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (rows.row[j] < n) x[rows.row[j]] = xr[j];
  if (wg == 0 && threadIdx.x == 0) {
    ctl->info = flag;
    ctl->iter = it;  // maxit + 1 when the loop ran out (pcg.c:165)
    ctl->relres = normr / n2b;
  }
}

// minres.c:96-193; v_hat = b - A x, y = K v_hat (dinv), beta = sqrt(v_hat.y) are the caller's; w = w_old = v_hat_old = 0.
// Everything but v (the vector other workgroups read) lives in registers.
template <int R>
__global__ __launch_bounds__(256) void minres_coop_kernel(int n, int nwg, const int *__restrict__ ind,
                                                          const int *__restrict__ col, const double *__restrict__ val,
                                                          const double *__restrict__ dinv, double *x,
                                                          const double *v_hat, const double *y, double *v,
                                                          double norm_r0, double beta0, double errtol, int it_max,
                                                          CoopCtl *ctl, double *part, double *hist) {
  __shared__ double sh[16 + 8];
  const int wg = blockIdx.x;
  OwnedRows<R> rows;
  rows.load(n, nwg, ind, col, val);
  double xr[R], vh[R], vho[R], yr[R], wr[R], wo[R], vr[R], dr[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const bool in = rows.row[j] < n;
    xr[j] = in ? x[rows.row[j]] : 0.0;
    vh[j] = in ? v_hat[rows.row[j]] : 0.0;
    dr[j] = (in && dinv) ? dinv[rows.row[j]] : 1.0;
    yr[j] = dinv ? (in ? y[rows.row[j]] : 0.0) : vh[j];
    vho[j] = wr[j] = wo[j] = vr[j] = 0.0;
  }
  unsigned gen = 0;
  double beta = beta0, beta_old = 1.0, c = 1.0, c_old = 1.0, s = 0.0, s_old = 0.0, eta = beta0, norm_rmr = norm_r0;
  int it = 0, info = 1;  // 1: left by the loop test (0 / -1 decided below)
  for (;;) {
    if (it >= it_max || norm_rmr < errtol * norm_r0) break;  // minres.c:114
    it += 1;
#pragma unroll
    for (int j = 0; j < R; ++j) {  // :123-124
      vr[j] = yr[j] / beta;
      if (rows.row[j] < n) v[rows.row[j]] = vr[j];
    }
    if (!coop_barrier(ctl, nwg, gen)) return;
    double a1[1] = {0.0};
    double avr[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {  // :127-129
      avr[j] = rows.product(j, col, val, v);
      a1[0] += vr[j] * avr[j];
    }
    block_sum<1>(a1, sh);
    if (nwg > 1) {
      if (threadIdx.x == 0) part[wg] = a1[0];
      if (!coop_barrier(ctl, nwg, gen)) return;
      grid_sum<1>(a1, part, nwg, sh);
    }
    const double alpha = a1[0];
    const double c1 = alpha / beta, c2 = beta / beta_old;  // :131
    double b1[1] = {0.0};
#pragma unroll
    for (int j = 0; j < R; ++j) {  // :132-143
      const double t = avr[j] - c1 * vh[j] - c2 * vho[j];
      vho[j] = vh[j];
      vh[j] = t;
      yr[j] = dinv ? t * dr[j] : t;
      b1[0] += t * yr[j];
    }
    block_sum<1>(b1, sh);
    if (nwg > 1) {
      if (threadIdx.x == 0) part[kCoopMaxWg + wg] = b1[0];
      if (!coop_barrier(ctl, nwg, gen)) return;
      grid_sum<1>(b1, part + kCoopMaxWg, nwg, sh);
    }
    beta_old = beta;
    beta = b1[0];
    if (beta < 0.0) {  // :144-146
      info = -3;
      break;
    }
    beta = sqrt(beta);
    const double c_oold = c_old, s_oold = s_old;  // :151-164
    c_old = c;
    s_old = s;
    const double r1_hat = c_old * alpha - c_oold * s_old * beta_old;
    const double rr1 = sqrt(r1_hat * r1_hat + beta * beta);
    const double rr2 = s_old * alpha + c_oold * c_old * beta_old;
    const double rr3 = s_oold * beta_old;
    if (rr1 == 0.0) {
      info = -6;
      break;
    }
    c = r1_hat / rr1;
    s = beta / rr1;
    const double ce = c * eta;
#pragma unroll
    for (int j = 0; j < R; ++j) {  // :172-180
      const double tmp = wr[j];
      wr[j] = (vr[j] - rr3 * wo[j] - rr2 * tmp) / rr1;
      wo[j] = tmp;
      xr[j] += ce * wr[j];
    }
    eta = -s * eta;
    norm_rmr *= fabs(s);  // :192
    if (hist && wg == 0 && threadIdx.x == 0) hist[it] = norm_rmr;
  }
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (rows.row[j] < n) x[rows.row[j]] = xr[j];
  if (wg == 0 && threadIdx.x == 0) {
    ctl->iter = it;
    if (info == 1) {
      ctl->info = norm_rmr < errtol * norm_r0 ? 0 : -1;
      ctl->relres = norm_rmr / norm_r0;
    } else {
      ctl->info = info;  // -3 / -6: relres stays untouched, as in the reference
    }
  }
}

struct CoopMem {
  CoopCtl *ctl = nullptr;
  double *part = nullptr, *hist = nullptr;
  ~CoopMem() {
    if (ctl) (void)hipFree(ctl);
    if (part) (void)hipFree(part);
    if (hist) (void)hipFree(hist);
  }
  int init(int maxit, bool want_hist) {
    PSP_HIP(hipMalloc((void **)&ctl, sizeof(CoopCtl)));
    PSP_HIP(hipMalloc((void **)&part, sizeof(double) * 4 * kCoopMaxWg));
    PSP_HIP(hipMemsetAsync(ctl, 0, sizeof(CoopCtl), stream()));
    PSP_HIP(hipMemsetAsync(part, 0, sizeof(double) * 4 * kCoopMaxWg, stream()));
    if (want_hist) {
      PSP_HIP(hipMalloc((void **)&hist, sizeof(double) * ((size_t)maxit + 2)));
      PSP_HIP(hipMemsetAsync(hist, 0xff, sizeof(double) * ((size_t)maxit + 2), stream()));
    }
    return PSP_OK;
  }
  int fetch(CoopCtl *out) {
    PSP_HIP(hipMemcpyAsync(out, ctl, sizeof(CoopCtl), hipMemcpyDeviceToHost, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    if (out->error) return fail(PSP_ENODEV, "single-kernel solver: a grid barrier did not complete");
    return PSP_OK;
  }
};

// rows per thread (1, 2 or 4) so that at most kCoopMaxWg workgroups of 256 threads cover the rows
void coop_grid(int n, int *nwg, int *rpt) {
  int r = 1;
  while (r < 4 && (n + 256 * r - 1) / (256 * r) > kCoopMaxWg) r *= 2;
  *rpt = r;
  *nwg = std::max(1, (n + 256 * r - 1) / (256 * r));
}

bool coop_enabled() {
  static const bool on = [] {
    const char *e = tuning_env("PSP_COOP");
    return !(e && atoi(e) == 0);
  }();
  return on;
}

}  // namespace

// the operator as plain CSR arrays on this device, small enough for the single-kernel loops?
bool coop_applicable(const psp_csr *A, int n) {
  return coop_enabled() && A && !A->w4_only && !A->nparts && !A->multi && !A->host && A->ind && A->nrows == n &&
         A->ncols == n && n >= 1 && n <= kCoopMaxRows;
}

int pcg_coop_loop(const psp_csr *A, const double *dinv, int n, double *x, double *r, double *p, double *q, double n2b,
                  double tolb, double normr0, double rho0, int maxit, int *info, int *iter, double *relres,
                  double *hist) {
  CoopMem m;
  PSP_TRY(m.init(maxit, hist != nullptr));
  int nwg, rpt;
  coop_grid(n, &nwg, &rpt);
  (void)q;
#define PSP_COOP_PCG(R)                                                                                              \
  hipLaunchKernelGGL((pcg_coop_kernel<R>), dim3(nwg), dim3(256), 0, stream(), n, nwg, A->ind, A->col, A->val, dinv, x, \
                     r, p, n2b, tolb, normr0, rho0, maxit, m.ctl, m.part, m.hist)
  if (rpt == 1) PSP_COOP_PCG(1);
  else if (rpt == 2) PSP_COOP_PCG(2);
  else PSP_COOP_PCG(4);
#undef PSP_COOP_PCG
  PSP_LAUNCH_CHECK();
  CoopCtl c;
  PSP_TRY(m.fetch(&c));
  *info = c.info;
  *iter = c.iter;
  *relres = c.relres;
  if (hist) {
    const int cnt = std::min(c.iter, maxit);
    if (cnt >= 1) {
      std::vector<double> h((size_t)cnt);
      PSP_HIP(hipMemcpy(h.data(), m.hist + 1, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
      for (int i = 0; i < cnt; ++i)
        if (h[i] == h[i]) hist[1 + i] = h[i];  // the iteration that broke down wrote nothing
    }
  }
  return PSP_OK;
}

int minres_coop_loop(const psp_csr *A, const double *dinv, int n, double *x, double *v_hat, double *v_hat_old,
                     double *y, double *w, double *w_old, double *v, double *av, double norm_r0, double beta0,
                     double errtol, int it_max, int *info, int *iter, double *relres, double *hist) {
  CoopMem m;
  PSP_TRY(m.init(it_max, hist != nullptr));
  int nwg, rpt;
  coop_grid(n, &nwg, &rpt);
  (void)v_hat_old;
  (void)w;
  (void)w_old;
  (void)av;
#define PSP_COOP_MR(R)                                                                                               \
  hipLaunchKernelGGL((minres_coop_kernel<R>), dim3(nwg), dim3(256), 0, stream(), n, nwg, A->ind, A->col, A->val, dinv, \
                     x, v_hat, y, v, norm_r0, beta0, errtol, it_max, m.ctl, m.part, m.hist)
  if (rpt == 1) PSP_COOP_MR(1);
  else if (rpt == 2) PSP_COOP_MR(2);
  else PSP_COOP_MR(4);
#undef PSP_COOP_MR
  PSP_LAUNCH_CHECK();
  CoopCtl c;
  PSP_TRY(m.fetch(&c));
  *info = c.info;
  *iter = c.iter;
  if (c.info == 0 || c.info == -1) *relres = c.relres;
  if (hist) {
    const int cnt = std::min(c.iter, it_max);
    if (cnt >= 1) {
      std::vector<double> h((size_t)cnt);
      PSP_HIP(hipMemcpy(h.data(), m.hist + 1, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
      for (int i = 0; i < cnt; ++i)
        if (h[i] == h[i]) hist[1 + i] = h[i];
    }
  }
  return PSP_OK;
}

}  // namespace psp
