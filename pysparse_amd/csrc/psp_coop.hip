// psp_coop.hip -- small systems: the whole PCG / MINRES loop as ONE kernel.
//
// A system of 10^4 unknowns (BASELINE.json configs[0]: poisson2d(100)) fits the caches many times over; what its
// iteration costs on the GPU is the chain of ~9 dependent launches of the asynchronous loops (psp_solvers.hip):
// 20.7 us per PCG iteration at 100^2 in round 2, all of it launch latency.  Here the loop of pcg.c:91-163 (resp.
// minres.c:96-193) runs inside one kernel: a few co-resident workgroups of 1024 threads, one ROW PER THREAD for the whole
// solve -- the row's matrix entries (up to 8) and its slices of x, r, p, ... stay in registers -- and grid-wide barriers
// only where the reference has a dependency across rows:
//   PCG   : [p (own row and the row's columns), q = A p, p.q] | [stagnation scan, x, r, r.r, r.z; r published] | ...
//   MINRES: [v = y / beta at the row's columns, Av, v.Av] | [Lanczos update, y = K v_hat, v_hat.y; y published] | ...
// two barriers per iteration: the one vector that crosses workgroups (r, resp. y) is published at the barrier of a
// reduction, and a thread forms the entries of p (resp. v) it multiplies with itself from the gathered values -- the
// owner's own rounded operations on the same operands, hence the same bits.  Every workgroup evaluates the scalar
// recurrences itself from the same reduced values, so all of them take the same branch and leave the loop together.
//
// Arithmetic: per element the reference's operations, multiply and add rounded separately (-ffp-contract=off); a row's
// products are added left to right (csr_mat.c:49-54); a reduction is: a thread adds its rows in ascending order, the
// wave a fixed xor tree, the workgroup its waves in order, and the workgroups' partial sums are added in order by
// every workgroup -- fixed for a given matrix size, so runs are bitwise reproducible.  (The order differs from the
// asynchronous loops': iterates agree with theirs to rounding, not bit for bit; counts and goldens are tested.)
//
// Grid barrier (MI355X_MICROARCH.md "Correctness boundaries", cdna_hip_programming.md G16, second form): the per-XCD
// L2s are not coherent with each other, so every byte handed from one workgroup to another is stored AND loaded with
// agent-scope accesses, drained (s_waitcnt) before the workgroup announces itself on one arrival counter.  The grid is at most 256 workgroups of 1024 threads (one row per thread, one workgroup per CU; 128 until round 4) --
// and every spin is bounded: a barrier that does not complete sets an error flag that ends all workgroups.
// Round 4: the range went from 2^17 to 2^18 rows once the launch was cooperative and checked against the device's capacity:
// poisson2d(512), 256 workgroups: 21.8 / 19.0 us per PCG / MINRES iteration against 27 / 26 with one launch per phase
// (tools/coop_range_probe.py, profiles/r4_coop_range.txt).
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "psp_internal.h"

namespace psp {

namespace {

constexpr int kCoopMaxWg = 256;  // workgroups of 1024 threads: at most one per CU, all co-resident (round 4: 128 before)
constexpr int kCoopBlock = 1024;
constexpr int kCoopMaxRows = 256 * 1024;  // 2^18 rows; beyond that the asynchronous loops (whole-chip kernels) take over

struct CoopCtl {
  unsigned count;
  unsigned gen;
  int error;
  int info, iter;
  double relres;
};

// Vectors and partial sums that cross workgroups are written and read with agent-scope (device-coherent) accesses --
// relaxed atomics, i.e. plain global_store / global_load with the sc1 bit: they go past the non-coherent levels, so no
// cache-wide write-back (release) or invalidate (acquire) is needed around the barrier, which cost ~8 us per barrier
// when every wave issued them.
__device__ __forceinline__ void coh_store(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coh_load(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One monotone arrival counter: a workgroup adds 1 (no value returned: nothing to wait for) and polls until the count
// reaches nwg * (barriers so far) -- one memory round trip after the last arrival instead of two.
// Publishing: every wave drains its own vector-memory counter before the workgroup barrier that precedes the arrival.
// A workgroup-scope release fence does NOT do that on gfx9 (non-tgsplit mode): it emits no s_waitcnt vmcnt, so the sc1
// stores of the other waves could still be in flight when a remote workgroup passes the barrier and sc1-loads them.
// Waiting: bounded by the 100 MHz wall clock (kCoopSpinTicks), not by a poll count -- a barrier among
// co-resident workgroups completes in microseconds; one that cannot complete (a workgroup that is not resident) ends
// every workgroup through the error word and the host falls back to the launch-per-phase loop from the saved vectors.
constexpr long long kCoopSpinTicks = 50000000;  // s_memrealtime ticks (100 MHz): 0.5 s (20 ms until round 5: a GPU that is
                                                // time-sliced between processes can hold a resident workgroup back longer)
__device__ __forceinline__ bool coop_barrier(CoopCtl *c, int nwg, unsigned &gen) {
  if (nwg == 1) {
    __syncthreads();
    return true;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's coherent stores have left the CU ...
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();                                        // ... before the workgroup announces itself
  gen += 1;
  if (threadIdx.x == 0) {
    (void)__hip_atomic_fetch_add(&c->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = gen * (unsigned)nwg;
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(&c->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      if ((++spins & 15u) == 0 &&
          (wall_clock64() - t0 > kCoopSpinTicks || __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(&c->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // orders the coherent loads below after the barrier
  return __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// workgroup sum of NV values per thread: wave xor tree, then the waves in order; every thread gets the result
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *sh /* >= 16 * NV + NV */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    // lane 0 of the xor tree this loop used to be IS the shuffle-down tree's lane 0 (v[l] + v[l + 32], then 16, 8, ...):
    // psp_wave_sum's permlane / DPP form of it (psp_internal.h), the same bits
    const double s = psp_wave_sum(v[j]);
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

// the workgroups' partial sums (in part[j * kCoopMaxWg + wg]) added in workgroup order by every workgroup: all of them
// are fetched at once (one round trip), then thread j adds value j's in order
template <int NV>
__device__ __forceinline__ void grid_sum(double (&v)[NV], const double *part, int nwg, double *sh) {
  if (threadIdx.x < NV * kCoopMaxWg) {
    const int j = threadIdx.x / kCoopMaxWg, w = threadIdx.x % kCoopMaxWg;
    sh[threadIdx.x] = w < nwg ? coh_load(part + j * kCoopMaxWg + w) : 0.0;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double t = 0.0;
    for (int w = 0; w < nwg; ++w) t += sh[threadIdx.x * kCoopMaxWg + w];
    sh[NV * kCoopMaxWg + threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = sh[NV * kCoopMaxWg + j];
  __syncthreads();
}

constexpr int kRegNz = 8;  // entries of a row kept in registers (every 5- / 7-point row); longer rows are re-read

// The row a thread owns for the whole solve: row = wg * 1024 + t.  The matrix is constant across iterations, so the
// row's entries (and the Jacobi factors of its COLUMNS) are loaded once into registers; per iteration a product costs
// only the gathers of the one vector that crosses workgroups.
struct OwnedRow {
  int row, cnt, first;
  double val[kRegNz];
  int col[kRegNz];
  __device__ __forceinline__ void load(int n, const int *__restrict__ ind, const int *__restrict__ cidx,
                                       const double *__restrict__ v) {
    row = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x;
    cnt = 0;
    first = 0;
    if (row < n) {
      first = ind[row];
      cnt = ind[row + 1] - first;
    }
#pragma unroll
    for (int k = 0; k < kRegNz; ++k) {
      const bool have = k < cnt;
      val[k] = have ? v[first + k] : 0.0;
      col[k] = have ? cidx[first + k] : 0;
    }
  }
};

// pcg.c:91-166 from the head of iteration 1: r = b - A x, rho = r.z and ||r|| > tolb are the caller's.
// x, r, p, q of the thread's row live in registers for the whole solve.  What crosses workgroups per iteration is r
// alone, published at the SAME barrier as the partial sums of r.r / r.z: a thread forms the entries of p it multiplies
// with, p[c] = z[c] + beta p_old[c] with z[c] = r[c] dinv[c], itself -- the owner's own two rounded operations on the
// same operands, so the same bits -- from the gathered r[c] and the p[c] it kept from the previous iteration.  Two grid
// barriers per iteration (p.q; r.r / r.z) instead of three.
__global__ __launch_bounds__(kCoopBlock) void pcg_coop_kernel(int n, int nwg, const int *__restrict__ ind,
                                                              const int *__restrict__ col,
                                                              const double *__restrict__ val,
                                                              const double *__restrict__ dinv, const double *x,
                                                              double *xout, double *r, double n2b, double tolb,
                                                              double normr0, double rho0, int maxit, CoopCtl *ctl,
                                                              double *part, double *hist) {
  __shared__ double sh[3 * kCoopMaxWg + 8];
  const int wg = blockIdx.x;
  OwnedRow a;
  a.load(n, ind, col, val);
  const bool in = a.row < n;
  double xr = in ? x[a.row] : 0.0, rr = in ? r[a.row] : 0.0, pr = 0.0, qr = 0.0;
  const double dr = (in && dinv) ? dinv[a.row] : 1.0;
  double dc[kRegNz], rc[kRegNz], pc[kRegNz];  // Jacobi factor, r and p at the row's columns
#pragma unroll
  for (int k = 0; k < kRegNz; ++k) {
    const bool have = k < a.cnt;
    dc[k] = (have && dinv) ? dinv[a.col[k]] : 1.0;
    rc[k] = have ? r[a.col[k]] : 0.0;
    pc[k] = 0.0;
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
    {  // pcg.c:93-97, :106, :113-114 -- for the own row and for the row's columns
      const double z = dinv ? rr * dr : rr;
      pr = it == 1 ? z : z + beta * pr;
#pragma unroll
      for (int k = 0; k < kRegNz; ++k) {
        const double zc = dinv ? rc[k] * dc[k] : rc[k];
        pc[k] = it == 1 ? zc : zc + beta * pc[k];
      }
    }
    double v1[1];
    {  // pcg.c:116-117; csr_mat.c:49-54: the stored products left to right from 0.0
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < kRegNz; ++k)
        if (k < a.cnt) sum += a.val[k] * pc[k];
      qr = sum;
      v1[0] = pr * qr;
    }
    block_sum<1>(v1, sh);
    if (nwg > 1) {
      if (threadIdx.x == 0) coh_store(part + wg, v1[0]);
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
    double v3[3] = {0.0, 0.0, 0.0};  // r.r, r.z, number of threads whose row did not stagnate
    if (in) {
      double dmax = 0.0;
      if (!stag0) {  // pcg.c:127-139 (the scan reads x before the update); a NaN never replaces dmax, like `if (ddum > dmax)`;
                     // x == 0 and p != 0 ASSIGNS dmax = 1.0, which for the test 1 + dmax == 1 is max(dmax, 1.0)
        const double ddum = xr != 0.0 ? fabs(alpha * pr / xr) : (pr != 0.0 ? 1.0 : 0.0);
        dmax = (ddum > dmax) ? ddum : dmax;
      }
      if (alpha != 0.0) {  // daxpy returns at once for a zero coefficient
        xr = xr + alpha * pr;
        rr = rr + (-alpha) * qr;
      }
      v3[0] = rr * rr;
      v3[1] = rr * (dinv ? rr * dr : rr);
      v3[2] = (1.0 + dmax != 1.0) ? 1.0 : 0.0;
      if (nwg > 1) coh_store(r + a.row, rr);  // the one vector other workgroups read
      else r[a.row] = rr;
    }
    block_sum<3>(v3, sh);  // (one workgroup: its barriers also publish r)
    if (nwg > 1) {
      if (threadIdx.x < 3) coh_store(part + (1 + threadIdx.x) * kCoopMaxWg + wg, v3[threadIdx.x]);
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
#pragma unroll
    for (int k = 0; k < kRegNz; ++k)
      if (k < a.cnt) rc[k] = nwg > 1 ? coh_load(r + a.col[k]) : r[a.col[k]];
    if (nwg == 1) __syncthreads();  // nobody overwrites r before every thread has gathered
  }
  if (in) xout[a.row] = xr;  // a staging vector: the host copies it over x only when NO workgroup gave up at a barrier
  if (wg == 0 && threadIdx.x == 0) {
    ctl->info = flag;
    ctl->iter = it;  // maxit + 1 when the loop ran out (pcg.c:165)
    ctl->relres = normr / n2b;
  }
}

// minres.c:96-193; v_hat = b - A x, y = K v_hat (dinv), beta = sqrt(v_hat.y) are the caller's; w = w_old = v_hat_old = 0.
// Everything lives in registers; what crosses workgroups is the unnormalised Lanczos vector y (= K v_hat), published at
// the barrier of the v_hat.y reduction -- a thread divides the entries it gathers by beta itself (the owner's own
// correctly rounded division) -- so an iteration has two grid barriers (v.Av; v_hat.y).
__global__ __launch_bounds__(kCoopBlock) void minres_coop_kernel(int n, int nwg, const int *__restrict__ ind,
                                                                 const int *__restrict__ col,
                                                                 const double *__restrict__ val,
                                                                 const double *__restrict__ dinv, const double *x,
                                                                 double *xout, const double *v_hat, double *yv, double norm_r0,
                                                                 double beta0, double errtol, int it_max, CoopCtl *ctl,
                                                                 double *part, double *hist) {
  __shared__ double sh[kCoopMaxWg + 8];
  const int wg = blockIdx.x;
  OwnedRow a;
  a.load(n, ind, col, val);
  const bool in = a.row < n;
  double xr = in ? x[a.row] : 0.0, vh = in ? v_hat[a.row] : 0.0, vho = 0.0, wr = 0.0, wo = 0.0;
  const double dr = (in && dinv) ? dinv[a.row] : 1.0;
  double yr = in ? yv[a.row] : 0.0;  // yv holds K v_hat (the caller copies v_hat into it when there is no K)
  double yc[kRegNz];
#pragma unroll
  for (int k = 0; k < kRegNz; ++k) yc[k] = k < a.cnt ? yv[a.col[k]] : 0.0;
  unsigned gen = 0;
  double beta = beta0, beta_old = 1.0, c = 1.0, c_old = 1.0, s = 0.0, s_old = 0.0, eta = beta0, norm_rmr = norm_r0;
  int it = 0, info = 1;  // 1: left by the loop test (0 / -1 decided below)
  for (;;) {
    if (it >= it_max || norm_rmr < errtol * norm_r0) break;  // minres.c:114
    it += 1;
    const double vr = yr / beta;  // :123-124
    double avr = 0.0;             // :127-129, csr_mat.c:49-54
#pragma unroll
    for (int k = 0; k < kRegNz; ++k)
      if (k < a.cnt) avr += a.val[k] * (yc[k] / beta);
    double a1[1] = {vr * avr};
    block_sum<1>(a1, sh);
    if (nwg > 1) {
      if (threadIdx.x == 0) coh_store(part + wg, a1[0]);
      if (!coop_barrier(ctl, nwg, gen)) return;
      grid_sum<1>(a1, part, nwg, sh);
    }
    const double alpha = a1[0];
    const double c1 = alpha / beta, c2 = beta / beta_old;  // :131
    const double t = avr - c1 * vh - c2 * vho;              // :132-143
    vho = vh;
    vh = t;
    yr = dinv ? t * dr : t;
    double b1[1] = {t * yr};
    if (in) {
      if (nwg > 1) coh_store(yv + a.row, yr);
      else yv[a.row] = yr;
    }
    block_sum<1>(b1, sh);
    if (nwg > 1) {
      if (threadIdx.x == 0) coh_store(part + kCoopMaxWg + wg, b1[0]);
      if (!coop_barrier(ctl, nwg, gen)) return;
      grid_sum<1>(b1, part + kCoopMaxWg, nwg, sh);
    }
#pragma unroll
    for (int k = 0; k < kRegNz; ++k)
      if (k < a.cnt) yc[k] = nwg > 1 ? coh_load(yv + a.col[k]) : yv[a.col[k]];
    if (nwg == 1) __syncthreads();
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
    {  // :172-180
      const double tmp = wr;
      wr = (vr - rr3 * wo - rr2 * tmp) / rr1;
      wo = tmp;
      xr += ce * wr;
    }
    eta = -s * eta;
    norm_rmr *= fabs(s);  // :192
    if (hist && wg == 0 && threadIdx.x == 0) hist[it] = norm_rmr;
  }
  if (in) xout[a.row] = xr;  // staging, as in pcg_coop_kernel
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
  // control block and partial sums: the thread's slab (psp_internal.h: no allocation per solve); the history: the
  // solvers' vector pool
  CoopCtl *ctl = nullptr;
  double *part = nullptr, *hist = nullptr;
  size_t nhist = 0;
  ~CoopMem() { scratch_put(hist, nhist); }
  int init(int maxit, bool want_hist) {
    static_assert(sizeof(CoopCtl) <= kStateBytes && 4 * (size_t)kCoopMaxWg <= kCtlPartDoubles, "state slab");
    Workspace *ws;
    PSP_TRY(workspace(&ws));
    ctl = static_cast<CoopCtl *>(ws->state_dev);
    part = ws->ctl_part;
    PSP_HIP(hipMemsetAsync(ctl, 0, sizeof(CoopCtl), stream()));
    PSP_HIP(hipMemsetAsync(part, 0, sizeof(double) * 4 * kCoopMaxWg, stream()));
    if (want_hist) {
      nhist = (size_t)maxit + 2;
      PSP_TRY(scratch_get(nhist, &hist));
      PSP_HIP(hipMemsetAsync(hist, 0xff, sizeof(double) * nhist, stream()));
    }
    return PSP_OK;
  }
  // PSP_OK, kCoopFallback (a grid barrier gave up: the caller restores its vectors and runs the launch-per-phase loop)
  // or an error
  int fetch(CoopCtl *out) {
    PSP_HIP(hipMemcpyAsync(out, ctl, sizeof(CoopCtl), hipMemcpyDeviceToHost, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    return out->error ? kCoopFallback : PSP_OK;
  }
};

// one row per thread, workgroups of 1024 (as few arrivals per barrier as the hardware allows).  Measured (MI355X,
// profiles/r3_small_solvers.txt): poisson2d(100) 9.0 us per PCG iteration / 6.4 per MINRES iteration against 25 / 19 with
// one launch per phase; poisson2d(300), 88 workgroups: 12.1 / 9.7 against 22 / 16
int coop_grid(int n) { return std::max(1, (n + kCoopBlock - 1) / kCoopBlock); }

bool coop_enabled() {
  static const bool on = [] {
    const char *e = tuning_env("PSP_COOP");
    return !(e && atoi(e) == 0);
  }();
  return on;
}

// How many workgroups of the two kernels the current device can hold AT ONCE (occupancy x compute units; the smaller
// of the two kernels): a grid barrier among more workgroups than that cannot complete.  A partitioned (CPX) device
// reports its own CU count here.  Cached per device; 0 when the runtime cannot tell (the single-kernel loops are
// then not used).  PSP_COOP_CAPACITY (tuning switch) overrides the figure -- the tests use it to force the refusal.
int coop_capacity() {
  static std::mutex mu;
  static std::map<int, int> cap;
  if (const char *e = tuning_env("PSP_COOP_CAPACITY")) return atoi(e);
  std::lock_guard<std::mutex> lk(mu);
  const int dev = current_device();
  auto it = cap.find(dev);
  if (it != cap.end()) return it->second;
  int c = 0;
  Workspace *w = nullptr;
  if (workspace(&w) == PSP_OK && w->num_cu > 0) {
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, (const void *)pcg_coop_kernel, kCoopBlock, 0) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void *)minres_coop_kernel, kCoopBlock, 0) == hipSuccess)
      c = std::min(a, b) * w->num_cu;
    else
      (void)hipGetLastError();
  }
  cap[dev] = c;
  return c;
}

// Cooperative launch: the runtime refuses a grid it cannot make co-resident (hipErrorCooperativeLaunchTooLarge)
// instead of letting its barriers spin.  A refusal is not an error of the solve: kCoopFallback.
int coop_launch(const void *kernel, int nwg, void **args) {
  hipError_t e = hipLaunchCooperativeKernel(kernel, dim3(nwg), dim3(kCoopBlock), args, 0, stream());
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return kCoopFallback;
  }
  return PSP_OK;
}

bool coop_force_fail() {  // tests: behave as if a grid barrier had given up (PSP_TUNING=1 PSP_COOP_FAIL=1)
  const char *e = tuning_env("PSP_COOP_FAIL");
  return e && atoi(e) == 1;
}

}  // namespace

// the operator as plain CSR arrays on this device, small enough for the single-kernel loops, and the grid fits the
// device at once?
bool coop_applicable(const psp_csr *A, int n) {
  return coop_enabled() && A && !A->w4_only && !A->nparts && !A->multi && !A->host && A->ind && A->nrows == n &&
         A->ncols == n && n >= 1 && n <= kCoopMaxRows && A->max_row_nnz <= kRegNz &&  // every row fits the registers
         coop_grid(n) <= std::min(kCoopMaxWg, coop_capacity());
}

// On kCoopFallback x and r are what they were on entry and the caller continues with its other loops: the kernel leaves
// its x in a staging vector (p, which the single-kernel loop does not use otherwise) that is copied over x only after a
// launch in which no workgroup gave up -- a time-out that strikes in the last iteration lets some workgroups store and
// others not (round-4 advisor finding) -- and r is restored from the copy kept in q.
int pcg_coop_loop(const psp_csr *A, const double *dinv, int n, double *x, double *r, double *p, double *q, double n2b,
                  double tolb, double normr0, double rho0, int maxit, int *info, int *iter, double *relres,
                  double *hist) {
  CoopMem m;
  PSP_TRY(m.init(maxit, hist != nullptr));
  int nwg = coop_grid(n);
  PSP_HIP(hipMemcpyAsync(q, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  const int *ind = A->ind, *col = A->col;
  const double *val = A->val;
  const double *xin = x;
  void *args[] = {&n, &nwg, &ind, &col, &val, &dinv, &xin, &p, &r, &n2b, &tolb, &normr0, &rho0, &maxit, &m.ctl, &m.part, &m.hist};
  int rc = coop_force_fail() ? kCoopFallback : coop_launch((const void *)pcg_coop_kernel, nwg, args);
  CoopCtl c;
  if (rc == PSP_OK) rc = m.fetch(&c);
  if (rc == kCoopFallback)
    PSP_HIP(hipMemcpyAsync(r, q, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  if (rc != PSP_OK) return rc;
  PSP_HIP(hipMemcpyAsync(x, p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));  // x is final when the call returns
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

// On kCoopFallback x, v_hat and y are what they were on entry (x: staged in w, as in pcg_coop_loop; y is restored from the
// copy kept in av).
int minres_coop_loop(const psp_csr *A, const double *dinv, int n, double *x, double *v_hat, double *v_hat_old,
                     double *y, double *w, double *w_old, double *v, double *av, double norm_r0, double beta0,
                     double errtol, int it_max, int *info, int *iter, double *relres, double *hist) {
  CoopMem m;
  PSP_TRY(m.init(it_max, hist != nullptr));
  int nwg = coop_grid(n);
  (void)v_hat_old;
  (void)w_old;
  double *yv = y;  // the vector that crosses workgroups: K v_hat, or v_hat itself without a preconditioner
  if (!dinv) {
    yv = v;
    PSP_HIP(hipMemcpyAsync(yv, v_hat, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  } else {
    PSP_HIP(hipMemcpyAsync(av, y, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  }
  const int *ind = A->ind, *col = A->col;
  const double *val = A->val;
  const double *vh = v_hat, *xin = x;
  void *args[] = {&n, &nwg, &ind, &col, &val, &dinv, &xin, &w, &vh, &yv, &norm_r0, &beta0, &errtol, &it_max, &m.ctl, &m.part, &m.hist};
  int rc = coop_force_fail() ? kCoopFallback : coop_launch((const void *)minres_coop_kernel, nwg, args);
  CoopCtl c;
  if (rc == PSP_OK) rc = m.fetch(&c);
  if (rc == kCoopFallback) {
    if (dinv) PSP_HIP(hipMemcpyAsync(y, av, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    PSP_HIP(hipMemsetAsync(w, 0, sizeof(double) * (size_t)n, stream()));  // the staging vector is the caller's w = 0 again
  }
  if (rc != PSP_OK) return rc;
  PSP_HIP(hipMemcpyAsync(x, w, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));  // x is final when the call returns
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
