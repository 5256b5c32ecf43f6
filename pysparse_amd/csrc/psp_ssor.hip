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
// the launches of one application are replayed from a hipGraph; see DESIGN.md.  Round 3: runs of narrow
// levels (2-D operators, the thin ends of 3-D ones) are walked by ONE workgroup that hands x from level to
// level through an LDS ring (ssor_run_kernel), and 3-D grid operators with wide levels are swept in bricks
// of 32^3 points, a coarse wavefront of workgroups (ssor_brick_kernel); the per-level launches remain for
// everything else (rows with more than 8 entries per sweep, dependencies that reach too far back).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
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
  // Round 3: RUNS of narrow levels walked by one workgroup that hands x from level to level through LDS
  // (ssor_run_kernel below).  A run = consecutive levels of at most kRunWide rows whose dependencies all lie at most
  // kRunMaxDist slots back; its slots get compact copies of what the kernel streams (index c = coff + slot - s0).
  struct Run {
    int l0 = 0, l1 = 0;  // levels [l0, l1)
    int s0 = 0, s1 = 0;  // slots [s0, s1)
    int coff = 0;        // offset of slot s0 in the compact arrays
    int tick0 = 0, nticks = 0;
  };
  struct RunSet {
    std::vector<Run> runs;
    int m = 0;                 // slots in all runs = stride of dpk
    unsigned *dpk = nullptr;   // distances to the dependencies' slots, two 16-bit fields per word, word-major
    double2 *vp = nullptr;     // the row's values in pairs (entries 2k, 2k+1), pair-major: one 16-byte load per pair
    double *dar = nullptr;     // diagonal by compact index
    double2 *gd = nullptr;     // per application: (G, diagonal); G = everything of the row's formula that does not
                               // depend on this sweep
    int2 *ticks = nullptr;     // (first slot, count <= kRunTick), never across a level boundary; zero-padded per run
  } run_f, run_b;
  int *run_progress = nullptr;  // [0]: tick the walking workgroup has finished (throttles the L2 helpers), [1]: sink
  // Round 3: BRICKS for 3-D grid operators whose levels are too wide for the runs (ssor_brick_kernel below).  The
  // positions are then ordered (brick, level, row) instead of (level, row): a brick of 32^3 grid points is one run for
  // one workgroup, the bricks themselves are a coarse wavefront handed out in dependency order.
  struct BrickSet {
    int nbricks = 0;
    int2 *ticks = nullptr;     // every brick's ticks, each brick's followed by kRunPad empty ones
    int4 *info = nullptr;      // per brick in processing order: first tick, ticks, first halo entry, halo entries
    int4 *pred = nullptr;      // per brick: up to three bricks it waits for (processing-order indices, -1: none)
    int *halo_pos = nullptr;   // positions of the dependencies outside the brick, brick-major
    unsigned *dpk = nullptr;   // per entry 16 bits: 0 none, 1..0x7fff distance back in slots, 0x8000 | i: halo entry i
    double2 *vp = nullptr;
    double *dar = nullptr;
    double2 *gd = nullptr;
    int4 *tick_ext = nullptr;  // pipelined sweeps: per tick its range of the brick's halo (first, count) and, 10 bits per
                               // predecessor, how many ticks of that predecessor must be complete before it is gathered
    int *flags = nullptr;      // [0 .. nbricks): done; [nbricks]: next brick to hand out; [nbricks + 1]: error;
                               // [nbricks + 2]: scratch word of run_pre_kernel
  } brick_f, brick_b;
  bool brick_mode = false;
  bool brick_pipe = false;  // both directions have tick_ext: bricks start before their predecessors have finished
  // a brick sweep that gave up waiting for a predecessor (bounded spins: never seen, but then the result is wrong) leaves
  // a word behind; it is copied to pinned memory behind every application and looked at by the next call on this handle
  int *brick_err_host = nullptr;
  hipEvent_t brick_ev = nullptr;
  bool brick_ev_pending = false;
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

// ------------------------------------------------------------------ runs of narrow levels through LDS (round 3)
//
// A 2-D operator has as many levels as a 3-D one has diagonal planes and a thousandth of the rows per level
// (2048^2: 4095 levels of <= 2048 rows); so have the thin ends of a 3-D schedule.  One launch per level costs a
// dependent-launch boundary (~2.6 us); one workgroup walking the levels with __syncthreads() in between costs the same,
// because what a level waits for is a global-memory round trip either way: its x values were stored by the level
// before, and on gfx950 loads and stores share one counter, so making the stores visible also waits for whatever was
// prefetched.  Here the only thing that crosses a level boundary is LDS:
//   * a tick = up to 1024 slots of one level, one per thread; x of the last 8192 slots lives in an LDS ring
//     (ring[slot & 8191]); a row reads its dependencies there -- the builder admits a level only if all of them lie at most
//     kRunMaxDist = 7168 slots back, so a tick never overwrites what it still reads -- writes its own x there, and the
//     tick ends with s_waitcnt lgkmcnt(0) + s_barrier: no vector-memory wait;
//   * everything else a row needs is static for the sweep: its values, the distances to its dependencies (16 bits
//     each), its diagonal, and G = the part of its formula that only involves the previous sweep (b - y resp. the
//     `temp` of ssor_kernel, preconmodule.c:110-140; formed by run_pre_kernel with the reference's own expression).
//     A thread loads them D ticks ahead into registers (the tick table tells it which slot it will own), so the loads
//     of D ticks are in flight while a tick computes; x and y go to global memory as plain stores nobody waits for;
//   * one CU pulls 62 GB/s from HBM and 125 GB/s from its L2 (tools/cu_stream_probe.hip), and a sweep streams ~50
//     bytes per row: helper workgroups on the same XCD (blockIdx % 8 == 0) read the same static data a bounded number
//     of ticks ahead of the walker, so that its loads are L2 hits.
// Same operations in the same order per row as ssor_row_ell => the reference's bits.
constexpr int kRunWide = 4096;     // widest level that may join a run
constexpr int kRunTick = 1024;     // slots per tick = threads of the walking workgroup
constexpr int kRunRing = 8192;     // LDS ring (doubles): 64 KiB
constexpr int kRunMaxDist = kRunRing - kRunTick;
constexpr int kRunMinLevels = 8;   // shorter runs stay on the per-level launches
constexpr int kRunLead = 48;       // ticks the helpers may run ahead (~2 MB of static data: half an XCD's L2)
constexpr int kRunPad = 32;        // empty ticks behind a run's table (>= 3 * D)

template <int W>
struct RunPre {
  double2 v[(W + 1) / 2];
  unsigned dw[(W + 1) / 2];
  double2 gd;  // (G, diagonal)
  int u, t;    // u < 0: this thread has no row in the tick
};

template <int W>
constexpr int run_depth() {
  return W <= 2 ? 6 : W <= 4 ? 5 : W <= 6 ? 4 : 3;  // vector-memory operations of D ticks stay under vmcnt's 63
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the largest distance (in slots) from a row of a narrow level to one of its dependencies, per level
template <int W>
__global__ __launch_bounds__(256) void run_dist_kernel(int a, int e, int l0, int l1, const int *__restrict__ lptr, int n,
                                                       const unsigned char *__restrict__ cnt8,
                                                       const int *__restrict__ pos, const int *__restrict__ slot_of,
                                                       int *__restrict__ lev_maxd) {
  const int u = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= e) return;
  int lo = l0, hi = l1;  // lptr[lo] <= u < lptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (lptr[mid] <= u) lo = mid; else hi = mid;
  }
  const int cnt = cnt8[u];
  int d = 0;
  for (int s = 0; s < W; ++s)
    if (s < cnt) {
      const int p = pos[(size_t)s * n + u];
      d = max(d, u - (slot_of ? slot_of[p] : p));
    }
  if (d > 0) atomicMax(lev_maxd + lo, d);
}

template <int W>
__global__ __launch_bounds__(256) void run_pack_kernel(int a, int e, int coff, int m, int n,
                                                       const unsigned char *__restrict__ cnt8,
                                                       const int *__restrict__ pos, const int *__restrict__ slot_of,
                                                       const int *__restrict__ rowmap, const double *__restrict__ da,
                                                       const double *__restrict__ val, unsigned *__restrict__ dpk,
                                                       double2 *__restrict__ vp, double *__restrict__ dar) {
  const int u = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= e) return;
  const int c = coff + (u - a);
  const int cnt = cnt8[u];
  unsigned w[(W + 1) / 2] = {};
  double v[2 * ((W + 1) / 2)] = {};
  for (int s = 0; s < W; ++s)
    if (s < cnt) {
      const int p = pos[(size_t)s * n + u];
      const unsigned d = (unsigned)(u - (slot_of ? slot_of[p] : p));  // 1 .. kRunMaxDist
      w[s >> 1] |= d << ((s & 1) * 16);
      v[s] = val[(size_t)s * n + u];
    }
  for (int k = 0; k < (W + 1) / 2; ++k) {
    dpk[(size_t)k * m + c] = w[k];
    vp[(size_t)k * m + c] = make_double2(v[2 * k], v[2 * k + 1]);
  }
  dar[c] = da[rowmap ? rowmap[u] : u];
}

// G of a run's slots: forward / backward Gauss-Seidel: b - y (preconmodule.c:171-193); ssor_kernel: temp (:110-140)
template <int KIND>
__global__ __launch_bounds__(256) void run_pre_kernel(int a, int e, int coff, const int *__restrict__ rowmap,
                                                      const double *__restrict__ b, const double *__restrict__ x,
                                                      const double *__restrict__ y, const double *__restrict__ da,
                                                      double omega, int first, const double *__restrict__ dar,
                                                      double2 *__restrict__ gd, int *__restrict__ progress) {
  const int u = a + blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) progress[0] = -1;
  if (u >= e) return;
  const int t = rowmap ? rowmap[u] : u;
  const double bt = b[t];
  double g;
  if constexpr (KIND <= 1) {
    g = bt - y[t];
  } else {
    if (KIND == 2 && first) {
      g = omega * bt;
    } else {
      const double xt = x[t], dt = da[t], yt = y[t];
      g = (1.0 - omega) * xt * dt + yt + omega * bt;
    }
  }
  const int c = coff + (u - a);
  gd[c] = make_double2(g, dar[c]);
}

template <bool MINUS, bool BACK, int W, int D>
__global__ __launch_bounds__(1024) void ssor_run_kernel(int nticks, const int2 *__restrict__ ticks, int n, int m, int s0,
                                                        int coff, const int *__restrict__ rowmap,
                                                        const double2 *__restrict__ vp,
                                                        const unsigned *__restrict__ dpk,
                                                        const double2 *__restrict__ gd, double *x, double *y,
                                                        double omega, int *progress, int nhelp) {
  static_assert(3 * D <= kRunPad, "the tick table's padding must cover the prefetch distance");
  constexpr int DW = (W + 1) / 2;
  if (blockIdx.x & 7) return;  // only the workgroups of the walker's XCD stay (they share its L2)
  const int tid = threadIdx.x;
  const int role = blockIdx.x >> 3;
  if (role > 0) {
    // ---- helper: touch the static data of ticks ahead of the walker, four ticks per round, never more than
    // kRunLead ticks ahead of it; every spin is bounded (a helper that loses the walker just stops throttling)
    double sink = 0.0;
    for (int kb = (role - 1) * 4; kb < nticks; kb += nhelp * 4) {
      int spins = 0;
      while (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < kb - kRunLead && ++spins < 100000)
        __builtin_amdgcn_s_sleep(8);
      double2 v[4][DW + 1];
      unsigned dwv[4][DW];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int s = 0; s < DW + 1; ++s) v[q][s] = make_double2(0.0, 0.0);
#pragma unroll
        for (int s = 0; s < DW; ++s) dwv[q][s] = 0;
        const int2 tk = ticks[kb + q];  // padded behind the run
        if (tid < tk.y) {
          const int c = coff + (tk.x + tid - s0);
#pragma unroll
          for (int s = 0; s < DW; ++s) v[q][s] = vp[(size_t)s * m + c];
#pragma unroll
          for (int s = 0; s < DW; ++s) dwv[q][s] = dpk[(size_t)s * m + c];
          v[q][DW] = gd[c];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int s = 0; s < DW + 1; ++s) sink += v[q][s].x + v[q][s].y;
#pragma unroll
        for (int s = 0; s < DW; ++s) sink += (double)dwv[q][s];
      }
    }
    if (sink == 1.2345678e300) progress[1] = 1;  // keeps the loads alive
    return;
  }
  // ---- the walker
  __shared__ double ring[kRunRing];
  for (int q = s0 - kRunRing + tid; q < s0; q += kRunTick)
    if (q >= 0) ring[q & (kRunRing - 1)] = x[BACK ? rowmap[q] : q];  // dependencies from before the run
  __syncthreads();
  RunPre<W> pre[D];
  int2 ti[D];
  // every lane loads, whether it owns a row of the tick or not (lanes past the tick's end repeat its last slot, an empty
  // padding tick its first): with the loads under a branch the compiler has to assume the path on which none was
  // issued and waits for all but the last one or two (s_waitcnt vmcnt(2)) -- which is the prefetch gone
  auto issue = [&](RunPre<W> &p, const int2 tk) {
    const int u = tk.x + min(tid, max(tk.y - 1, 0)), c = coff + (u - s0);
    p.u = tid < tk.y ? u : -1;
    if constexpr (BACK) p.t = rowmap[u]; else p.t = u;
#pragma unroll
    for (int s = 0; s < DW; ++s) p.v[s] = vp[(size_t)s * m + c];
#pragma unroll
    for (int s = 0; s < DW; ++s) p.dw[s] = dpk[(size_t)s * m + c];
    p.gd = gd[c];
  };
#pragma unroll
  for (int j = 0; j < D; ++j) {
    issue(pre[j], ticks[j]);
    ti[j] = ticks[j + D];
  }
  // let the prologue's loads land (once): the scheduler regroups them by array, and the wait counts the compiler derives
  // at the loop header are the worse of the two ways in -- with the prologue pending that is vmcnt(0) in every tick
  __syncthreads();
  for (int k = 0; k < nticks; k += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const RunPre<W> p = pre[j];
      const int2 tn = ti[j];            // tick k + j + D
      ti[j] = ticks[k + j + 2 * D];   // (the table is padded with empty ticks)
      // the ring reads of every lane (a lane without a row reads some slot and drops it), then the next loads: their
      // address arithmetic runs while the LDS answers
      double xs[W];
#pragma unroll
      for (int s = 0; s < W; ++s) {
        const int d = (int)((p.dw[s >> 1] >> ((s & 1) * 16)) & 0xffffu);
        xs[s] = ring[(p.u - d) & (kRunRing - 1)];
      }
      issue(pre[j], tn);
      if (p.u >= 0) {
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < W; ++s) {
          const unsigned d = (p.dw[s >> 1] >> ((s & 1) * 16)) & 0xffffu;
          const double vs = (s & 1) ? p.v[s >> 1].y : p.v[s >> 1].x;
          const double tt = MINUS ? acc - vs * xs[s] : acc + vs * xs[s];
          acc = d ? tt : acc;
        }
        double xn, yn;
        if constexpr (!MINUS) {
          xn = (p.gd.x - acc) / p.gd.y;
          yn = acc;
        } else {
          const double hi = omega * acc;
          yn = hi;
          xn = (p.gd.x + hi) / p.gd.y;
        }
        ring[p.u & (kRunRing - 1)] = xn;
        x[p.t] = xn;
        y[p.t] = yn;
      }
      lds_barrier();
      if (tid == 0 && nhelp > 0) __hip_atomic_store(progress, k + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------ bricks: 3-D grid operators with wide levels (round 3)
//
// A 512^3 operator has 1534 levels of up to 196 608 rows: too wide for a run, and one launch per level costs 5.2 us
// where the level's data stream in 1.5 (16 ms per application against 3.5 ms of streaming).  For a grid operator -- lower
// offsets {-1, -nx, -nx*ny}, no coupling that wraps around a grid line or plane; detected and verified at set-up -- the
// rows are grouped into bricks of 32^3 grid points and ordered (brick, level, row).  Inside a brick every level has at
// most 768 rows and its dependencies lie one or two levels back: a brick is a run (ssor_run_kernel above) for one
// workgroup, with two differences: the dependencies outside the brick -- the three faces towards lower (backward sweep:
// higher) indices, finished before the brick starts -- are gathered once, at the start, into an LDS halo, and x is stored
// with agent-scope stores, because another XCD's workgroup gathers it.  The bricks form a coarse wavefront (a brick
// needs its three face neighbours): they are handed out in that order through a counter, a workgroup waits for its
// brick's three predecessors on flags in memory (bounded spins) -- no deadlock whatever is resident, because whoever took
// an earlier brick is running.  Same operations in the same order per row as ssor_row_ell => the reference's bits.
#ifndef PSP_BRICK_EDGE
#define PSP_BRICK_EDGE 32  // (24: 256^3 2.64 instead of 2.89 ms per application, 512^3 unchanged; 16: 2.71 / 10.5 against 10.3)
#endif
constexpr int kBrickEdge = PSP_BRICK_EDGE;
// LDS ring (doubles; at least three of the widest levels) and halo (the three faces): with the hand-out word inside
// 64 KiB of static LDS for edge 32
constexpr int kBrickRing = kBrickEdge >= 32 ? 4096 : kBrickEdge >= 20 ? 2048 : 1024;
constexpr int kBrickHalo = kBrickEdge >= 32 ? 4080 : 3 * kBrickEdge * kBrickEdge + 48;
constexpr int kBrickWgs = kBrickEdge >= 32 ? 256 : kBrickEdge >= 20 ? 512 : 1280;  // workgroups of a sweep (<= resident)
// flag words of a sweep, one 128-byte line each (hundreds of workgroups poll them): [0] next brick to hand out,
// [1] error, [2] scratch word of run_pre_kernel, [3 + b] brick b (done / ticks published)
constexpr int kFlagStride = 32;
__host__ __device__ constexpr size_t flag_words(int bricks) { return (size_t)(bricks + 3) * kFlagStride; }
#define PSP_FLAG_NEXT(f) (f)
#define PSP_FLAG_ERR(f) ((f) + kFlagStride)
#define PSP_FLAG_SCRATCH(f) ((f) + 2 * kFlagStride)
#define PSP_FLAG_BRICK(f, b) ((f) + (size_t)(3 + (b)) * kFlagStride)
constexpr int kBrickTick = (3 * kBrickEdge * kBrickEdge / 4 + 63) / 64 * 64;  // threads of a brick's workgroup = the widest
                                                                               // level of a brick (768 for 32^3)
constexpr int kBrickMaxDist = kBrickRing - kBrickTick;

// distinct values of row - col over the strict lower entries: smallest / largest, then the range of the others
__global__ __launch_bounds__(256) void grid_offsets_kernel(int n, const int *__restrict__ ind, const int *__restrict__ col,
                                                           int pass, int *mm) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int lo = mm[0], hi = mm[1];
  int a = 0x7fffffff, b = 0;  // the thread's own range first, one pair of atomics per wave (4e8 entries at 512^3)
  if (r < n)
    for (int k = ind[r]; k < ind[r + 1]; ++k) {
      const int o = r - col[k];
      if (pass == 0 || (o != lo && o != hi)) {
        a = min(a, o);
        b = max(b, o);
      }
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a = min(a, __shfl_xor(a, off, 64));
    b = max(b, __shfl_xor(b, off, 64));
  }
  if ((threadIdx.x & 63) == 0 && b > 0) {
    atomicMin(mm + (pass ? 2 : 0), a);
    atomicMax(mm + (pass ? 3 : 1), b);
  }
}

// couplings that wrap around a grid line / plane (the row is the first point of its line / plane) are not grid couplings
__global__ __launch_bounds__(256) void grid_verify_kernel(int n, int nx, int nxy, const int *__restrict__ ind,
                                                          const int *__restrict__ col, int *bad) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int k = ind[r]; k < ind[r + 1]; ++k) {
    const int o = r - col[k];
    const bool ok = (o == 1 && r % nx >= 1) || (o == nx && (r % nxy) / nx >= 1) || (o == nxy);
    if (!ok) atomicAdd(bad, 1);
  }
}

// sort key of a row: (processing rank of its brick, its level)
__global__ __launch_bounds__(256) void brick_key_kernel(int n, int nx, int ny, int nxy, int ba, int bb,
                                                        const int *__restrict__ rank, const int *__restrict__ level,
                                                        unsigned long long *__restrict__ key, int *__restrict__ iota) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int i = r % nx, j = (r % nxy) / nx, k = r / nxy;
  const int b = ((k / kBrickEdge) * bb + j / kBrickEdge) * ba + i / kBrickEdge;
  key[r] = ((unsigned long long)(unsigned)rank[b] << 32) | (unsigned)level[r];
  iota[r] = r;
}

// slot -> first slot of each brick (every brick of the grid is non-empty), starts of the (brick, level) groups
__global__ __launch_bounds__(256) void brick_bounds_kernel(int n, const unsigned long long *__restrict__ key,
                                                           int *__restrict__ brick_start, int *__restrict__ group_flag) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const unsigned long long k = key[u], kp = u ? key[u - 1] : ~0ull;
  group_flag[u] = k != kp;
  if ((k >> 32) != (kp >> 32)) brick_start[(int)(k >> 32)] = u;
}

__device__ __forceinline__ int brick_of_slot(int u, int nb, const int *__restrict__ brick_start) {
  int lo = 0, hi = nb;  // brick_start[lo] <= u < brick_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (brick_start[mid] <= u) lo = mid; else hi = mid;
  }
  return lo;
}

// feasibility of one direction before anything is committed: per lower entry (i, c) the forward sweep has row i depend
// on row c, the backward sweep row c on row i.  Inside a brick the dependency must lie 1 .. kBrickMaxDist slots back,
// outside it must belong to an earlier brick; counts the outside dependencies per brick
__global__ __launch_bounds__(256) void brick_check_kernel(int n, int dir, const int *__restrict__ ind,
                                                          const int *__restrict__ col, const int *__restrict__ slot_of,
                                                          int nb, const int *__restrict__ brick_start,
                                                          int *__restrict__ ext_cnt, int *bad) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int k = ind[r]; k < ind[r + 1]; ++k) {
    const int row = dir ? col[k] : r, dep = dir ? r : col[k];
    const int u = slot_of[row], q = slot_of[dep];
    const int bu = brick_of_slot(u, nb, brick_start), bq = brick_of_slot(q, nb, brick_start);
    if (bu == bq) {
      if (u - q < 1 || u - q > kBrickMaxDist) atomicAdd(bad, 1);
    } else {
      if (bq > bu) atomicAdd(bad, 1);
      atomicAdd(ext_cnt + bu, 1);
    }
  }
}

// number of a slot's dependencies outside its brick (ext[n] = 0: the scan's last element)
template <int W>
__global__ __launch_bounds__(256) void brick_extcnt_kernel(int n, const unsigned char *__restrict__ cnt8,
                                                           const int *__restrict__ pos, const int *__restrict__ slot_of,
                                                           int nb, const int *__restrict__ brick_start,
                                                           int *__restrict__ ext) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u > n) return;
  if (u == n) {
    ext[n] = 0;
    return;
  }
  const int b = brick_of_slot(u, nb, brick_start);
  const int bs = brick_start[b], be = brick_start[b + 1];
  const int cnt = cnt8[u];
  int c = 0;
  for (int s = 0; s < W; ++s)
    if (s < cnt) {
      const int p = pos[(size_t)s * n + u];
      const int q = slot_of ? slot_of[p] : p;
      c += !(q >= bs && q < be);
    }
  ext[u] = c;
}

// per slot: the 16-bit codes of its entries, its value pairs, its diagonal; outside dependencies get a halo entry each,
// numbered in slot order inside the brick (eoff = exclusive scan of brick_extcnt_kernel's counts): the entries a tick
// needs are then a contiguous range of the brick's halo
template <int W>
__global__ __launch_bounds__(256) void brick_pack_kernel(int n, const unsigned char *__restrict__ cnt8,
                                                         const int *__restrict__ pos, const double *__restrict__ val,
                                                         const int *__restrict__ slot_of, const int *__restrict__ rowmap,
                                                         const double *__restrict__ da, int nb,
                                                         const int *__restrict__ brick_start,
                                                         const int *__restrict__ eoff, int *__restrict__ halo_pos,
                                                         unsigned *__restrict__ dpk, double2 *__restrict__ vp,
                                                         double *__restrict__ dar) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const int b = brick_of_slot(u, nb, brick_start);
  const int bs = brick_start[b], be = brick_start[b + 1];
  const int cnt = cnt8[u];
  int h = eoff[u] - eoff[bs];
  int hg = eoff[u];
  unsigned w[(W + 1) / 2] = {};
  double v[2 * ((W + 1) / 2)] = {};
  for (int s = 0; s < W; ++s)
    if (s < cnt) {
      const int p = pos[(size_t)s * n + u];
      const int q = slot_of ? slot_of[p] : p;
      unsigned code;
      if (q >= bs && q < be) {
        code = (unsigned)(u - q);
      } else {
        halo_pos[hg++] = p;
        code = 0x8000u | (unsigned)h++;
      }
      w[s >> 1] |= code << ((s & 1) * 16);
      v[s] = val[(size_t)s * n + u];
    }
  for (int k = 0; k < (W + 1) / 2; ++k) {
    dpk[(size_t)k * n + u] = w[k];
    vp[(size_t)k * n + u] = make_double2(v[2 * k], v[2 * k + 1]);
  }
  dar[u] = da[rowmap ? rowmap[u] : u];
}

// ---- set-up of the pipelined brick sweeps: a slot's tick inside its brick, what a tick needs of the predecessors
__global__ __launch_bounds__(256) void brick_ticklocal_kernel(int nticks, const int2 *__restrict__ ticks,
                                                              const int *__restrict__ tick_brick,
                                                              const int4 *__restrict__ info, int *__restrict__ ticklocal) {
  const int g = blockIdx.x;
  if (g >= nticks) return;
  const int2 t = ticks[g];
  const int local = g - info[tick_brick[g]].x;
  for (int i = threadIdx.x; i < t.y; i += blockDim.x) ticklocal[t.x + i] = local;
}

template <int W>
__global__ __launch_bounds__(256) void brick_need_kernel(int n, const unsigned char *__restrict__ cnt8,
                                                         const int *__restrict__ pos, const int *__restrict__ slot_of,
                                                         int nb, const int *__restrict__ brick_start,
                                                         const int4 *__restrict__ info, const int4 *__restrict__ pred,
                                                         const int *__restrict__ ticklocal, int *__restrict__ need,
                                                         int *bad) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const int b = brick_of_slot(u, nb, brick_start);
  const int bs = brick_start[b], be = brick_start[b + 1];
  const int g = info[b].x + ticklocal[u];
  const int4 pr = pred[b];
  const int cnt = cnt8[u];
  for (int s = 0; s < W; ++s)
    if (s < cnt) {
      const int p = pos[(size_t)s * n + u];
      const int q = slot_of ? slot_of[p] : p;
      if (q >= bs && q < be) continue;
      const int bq = brick_of_slot(q, nb, brick_start);
      const int k = bq == pr.x ? 0 : bq == pr.y ? 1 : bq == pr.z ? 2 : -1;
      if (k < 0) {
        atomicAdd(bad, 1);  // a dependency in a brick that is not a face neighbour
        continue;
      }
      atomicMax(need + 3 * (size_t)g + k, ticklocal[q] + 1);
    }
}

__global__ __launch_bounds__(256) void brick_tickext_kernel(int nticks, const int2 *__restrict__ ticks,
                                                            const int *__restrict__ tick_brick,
                                                            const int *__restrict__ brick_start,
                                                            const int *__restrict__ eoff, const int *__restrict__ need,
                                                            int4 *__restrict__ tick_ext, int *bad) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nticks) return;
  const int2 t = ticks[g];
  int4 e = make_int4(0, 0, 0, 0);
  if (t.y > 0) {
    const int bs = brick_start[tick_brick[g]];
    e.x = eoff[t.x] - eoff[bs];
    e.y = eoff[t.x + t.y] - eoff[t.x];
    const int a = need[3 * (size_t)g], b = need[3 * (size_t)g + 1], c = need[3 * (size_t)g + 2];
    if (a > 1023 || b > 1023 || c > 1023 || e.y > kBrickTick) atomicAdd(bad, 1);
    e.z = a | (b << 10) | (c << 20);
  }
  tick_ext[g] = e;
}

// the first forward Gauss-Seidel sweep of an application starts from y = 0 (preconmodule.c:164-165), so its G is b itself
// (b - 0.0 is b, bit for bit): the pass that brings b into position order writes (G, diagonal) along with it, and neither
// the whole-chip pass that forms G nor the clearing of y is needed for that sweep
__global__ __launch_bounds__(256) void brick_first_gd_kernel(int n, const int *__restrict__ pos2row,
                                                             const double *__restrict__ b, const double *__restrict__ dar,
                                                             double *__restrict__ bp, double2 *__restrict__ gd) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double bt = b[pos2row[t]];
  bp[t] = bt;
  gd[t] = make_double2(bt - 0.0, dar[t]);
}

// clears a sweep's flags (done words, hand-out counter) -- NOT the error word: that one is sticky across applications
// until the host has reported it (brick_error_check), so that a solver, which looks once after its last application,
// cannot miss a sweep that gave up earlier in the solve.  A kernel, not a memset: inside the captured graph
// of an application a memset node was not ordered before the brick kernel behind it on every HIP runtime (the one PyTorch
// brings along replayed them concurrently: a second application then found its predecessors "done" from the first)
__global__ __launch_bounds__(256) void brick_begin_kernel(int count, int *__restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count && i != kFlagStride) flags[i] = 0;
}

template <bool MINUS, bool BACK, int W, int D>
__global__ __launch_bounds__(kBrickTick) void ssor_brick_kernel(int nbricks, const int4 *__restrict__ info,
                                                          const int4 *__restrict__ pred, const int2 *__restrict__ ticks,
                                                          const int *__restrict__ halo_pos, int m,
                                                          const int *__restrict__ rowmap, const double2 *__restrict__ vp,
                                                          const unsigned *__restrict__ dpk,
                                                          const double2 *__restrict__ gd, double *x, double *y,
                                                          double omega, int *flags) {
  static_assert(3 * D <= kRunPad, "the tick table's padding must cover the prefetch distance");
  constexpr int DW = (W + 1) / 2;
  __shared__ double ring[kBrickRing];
  __shared__ double halo[kBrickHalo];
  __shared__ int next_s;
  const int tid = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (;;) {
    // ---- the next brick in dependency order.  Wave 0 hands out and waits with wave-uniform control flow (every lane
    // polls the same word, the values are made uniform explicitly): written as `if (tid == 0) { ... }` the compiler split
    // this loop by lanes and let the other lanes run ahead through the barriers without thread 0
    if (wid == 0) {
      int nb_ = 0;
      if (tid == 0) nb_ = atomicAdd(PSP_FLAG_NEXT(flags), 1);
      nb_ = __builtin_amdgcn_readfirstlane(nb_);
      if (tid == 0) next_s = nb_;
    }
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(next_s);
    if (b >= nbricks) return;
    const int4 bi = info[b];
    // everything that does not depend on the predecessors is requested before waiting for them: the positions of the
    // halo entries and the static data of the first D ticks
    constexpr int NH = (kBrickHalo + kBrickTick - 1) / kBrickTick;
    int hp[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int h = tid + i * kBrickTick;
      hp[i] = h < bi.w ? halo_pos[bi.z + h] : -1;
    }
    const int2 *tk = ticks + bi.x;
    const int nticks = bi.y;
    RunPre<W> pre[D];
    int2 ti[D];
    auto issue = [&](RunPre<W> &p, const int2 t2) {  // every lane loads (see ssor_run_kernel)
      const int u = t2.x + min(tid, max(t2.y - 1, 0));
      p.u = tid < t2.y ? u : -1;
      if constexpr (BACK) p.t = rowmap[u]; else p.t = u;
#pragma unroll
      for (int s = 0; s < DW; ++s) p.v[s] = vp[(size_t)s * m + u];
#pragma unroll
      for (int s = 0; s < DW; ++s) p.dw[s] = dpk[(size_t)s * m + u];
      p.gd = gd[u];
    };
#pragma unroll
    for (int j = 0; j < D; ++j) {
      issue(pre[j], tk[j]);
      ti[j] = tk[j + D];
    }
    if (wid == 0) {  // the three face neighbours
      const int4 pr = pred[b];
      const int ps[3] = {pr.x, pr.y, pr.z};
      bool lost = false;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (ps[i] < 0 || lost) continue;
        int spins = 0;
        for (;;) {
          const int f = __builtin_amdgcn_readfirstlane(
              __hip_atomic_load(PSP_FLAG_BRICK(flags, ps[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (f != 0) break;
          __builtin_amdgcn_s_sleep(2);
          const int err = __builtin_amdgcn_readfirstlane(
              __hip_atomic_load(PSP_FLAG_ERR(flags), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (++spins > (1 << 22) || err) {  // gives up (seconds): every workgroup ends, the error word says so
            lost = true;
            break;
          }
        }
      }
      if (tid == 0) {
        if (lost) __hip_atomic_store(PSP_FLAG_ERR(flags), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        next_s = lost ? -1 : b;
      }
    }
    __syncthreads();  // (also lets the prologue's loads land, see ssor_run_kernel)
    if (__builtin_amdgcn_readfirstlane(next_s) < 0) return;
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if (hp[i] >= 0)
        halo[tid + i * kBrickTick] = __hip_atomic_load(x + hp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();  // the halo is in place
    for (int k = 0; k < nticks; k += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const RunPre<W> p = pre[j];
        const int2 tn = ti[j];
        ti[j] = tk[k + j + 2 * D];
        double xs[W];
#pragma unroll
        for (int s = 0; s < W; ++s) {
          const unsigned c = (p.dw[s >> 1] >> ((s & 1) * 16)) & 0xffffu;
          xs[s] = (c & 0x8000u) ? halo[c & 0x7fffu] : ring[(p.u - (int)c) & (kBrickRing - 1)];
        }
        issue(pre[j], tn);
        if (p.u >= 0) {
          double acc = 0.0;
#pragma unroll
          for (int s = 0; s < W; ++s) {
            const unsigned c = (p.dw[s >> 1] >> ((s & 1) * 16)) & 0xffffu;
            const double vs = (s & 1) ? p.v[s >> 1].y : p.v[s >> 1].x;
            const double tt = MINUS ? acc - vs * xs[s] : acc + vs * xs[s];
            acc = c ? tt : acc;
          }
          double xn, yn;
          if constexpr (!MINUS) {
            xn = (p.gd.x - acc) / p.gd.y;
            yn = acc;
          } else {
            const double hi = omega * acc;
            yn = hi;
            xn = (p.gd.x + hi) / p.gd.y;
          }
          ring[p.u & (kBrickRing - 1)] = xn;
          // another XCD gathers it.  (Agent-scope stores only for the rows another brick gathers -- one in ten -- and plain
          // stores for the rest was measured and is SLOWER: 256^3 3.03 against 2.89 ms, 512^3 10.97 against 10.3.)
          __hip_atomic_store(x + p.t, xn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          y[p.t] = yn;
        }
        lds_barrier();
      }
    }
    // ---- done: this brick's x has left the CU before the flag says so.  Every wave drains its own vector-memory
    // counter: a workgroup-scope release fence emits no s_waitcnt vmcnt on gfx9 (non-tgsplit mode), so without this the
    // sc1 stores of waves other than the publishing one could still be in flight when another XCD sc1-loads them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(PSP_FLAG_BRICK(flags, b), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// stages of the pipelined sweeps are kPipeDepth ticks apart (measured: 256^3 3.13 / 2.98 / 2.87 / 3.20 ms per application at
// 5 / 4 / 3 / 2; the skew between a brick and its predecessors is 32 ticks of geometry + 3 kPipeDepth + visibility)
constexpr int kPipeDepth = 3;
// The same sweep with the bricks PIPELINED (the default; PSP_SSOR_BRICK_PIPE=0 under PSP_TUNING keeps whole-brick waits): a
// brick does not wait for its three
// predecessors to finish, only for each of them to be far enough ahead.  flags[b] = number of b's ticks whose x is
// visible; a tick's halo entries are a contiguous range of the brick's halo (tick_ext), and they travel through a
// pipeline of their own, D ticks per stage: [positions + a poll of the predecessors' progress] -> [wave 0 checks the poll
// against what the tick needs and waits if a predecessor is behind; then the gathers] -> [LDS] -> the tick.  A brick
// publishes tick t - D once every wave has seen its static data of tick t arrive (loads return in order: what was issued
// before them, the stores of tick t - D, has then left the CU), so the wavefront of bricks is skewed by ~50 ticks
// instead of a whole brick's 94+.
template <bool MINUS, bool BACK, int W, int D>
__global__ __launch_bounds__(kBrickTick) void ssor_brick_pipe_kernel(
    int nbricks, const int4 *__restrict__ info, const int4 *__restrict__ pred, const int2 *__restrict__ ticks,
    const int4 *__restrict__ tick_ext, const int *__restrict__ halo_pos, int m, const int *__restrict__ rowmap,
    const double2 *__restrict__ vp, const unsigned *__restrict__ dpk, const double2 *__restrict__ gd, double *x, double *y,
    double omega, int *flags) {
  static_assert(3 * D <= kRunPad, "the tick tables' padding must cover the prefetch distance");
  constexpr int DW = (W + 1) / 2;
  constexpr int kOps = D * (2 * DW + 3 + (BACK ? 1 : 0));  // vector-memory operations every wave issues in D ticks
  static_assert(kOps < 60, "vmcnt");
  __shared__ double ring[kBrickRing];
  __shared__ double halo[kBrickHalo];
  __shared__ int next_s;
  const int tid = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (;;) {
    if (wid == 0) {
      int nb_ = 0;
      if (tid == 0) nb_ = atomicAdd(PSP_FLAG_NEXT(flags), 1);
      nb_ = __builtin_amdgcn_readfirstlane(nb_);
      if (tid == 0) next_s = nb_;
    }
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(next_s);
    if (b >= nbricks) return;
    const int4 bi = info[b];
    const int4 pr = pred[b];
    const int ps[3] = {pr.x, pr.y, pr.z};
    const int2 *tk = ticks + bi.x;
    const int4 *te = tick_ext + bi.x;
    const int nticks = bi.y, hb = bi.z;
    // wave 0: wait until predecessor k has published at least `nd` ticks (bounded; gives up with the error word set)
    auto wait_for = [&](int k, int nd, int have) {
      int spins = 0;
      while (have < nd) {
        __builtin_amdgcn_s_sleep(1);
        have = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(PSP_FLAG_BRICK(flags, ps[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (++spins > (1 << 22)) {
          if (tid == 0) __hip_atomic_store(PSP_FLAG_ERR(flags), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    };
    RunPre<W> pre[D];
    int2 ti[D];
    auto issue = [&](RunPre<W> &p, const int2 t2) {  // every lane loads (see ssor_run_kernel)
      const int u = t2.x + min(tid, max(t2.y - 1, 0));
      p.u = tid < t2.y ? u : -1;
      if constexpr (BACK) p.t = rowmap[u]; else p.t = u;
#pragma unroll
      for (int s = 0; s < DW; ++s) p.v[s] = vp[(size_t)s * m + u];
#pragma unroll
      for (int s = 0; s < DW; ++s) p.dw[s] = dpk[(size_t)s * m + u];
      p.gd = gd[u];
    };
#pragma unroll
    for (int j = 0; j < D; ++j) {
      issue(pre[j], tk[j]);
      ti[j] = tk[j + D];
    }
    // ---- the halo of the first 2 D ticks the plain way: wait for what they need, gather it
    {
      int n0 = 0, n1 = 0, n2 = 0, hcnt = 0;
#pragma unroll
      for (int t = 0; t < 2 * D; ++t) {
        const int4 e = te[t];  // (the tables are padded with empty ticks)
        n0 = max(n0, e.z & 1023);
        n1 = max(n1, (e.z >> 10) & 1023);
        n2 = max(n2, (e.z >> 20) & 1023);
        hcnt = max(hcnt, e.x + e.y);
      }
      if (wid == 0) {
        if (ps[0] >= 0 && n0) wait_for(0, n0, -1);
        if (ps[1] >= 0 && n1) wait_for(1, n1, -1);
        if (ps[2] >= 0 && n2) wait_for(2, n2, -1);
      }
      __syncthreads();
      for (int h = tid; h < hcnt; h += kBrickTick)
        halo[h] = __hip_atomic_load(x + halo_pos[hb + h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();  // (also lets the prologue's loads land, see ssor_run_kernel)
    }
    int4 teB[D], teC[D];
    int hpB[D], hpC[D], pollB[D][3];
    double xg[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      teB[j] = teC[j] = make_int4(0, 0, 0, 0);
      hpB[j] = hpC[j] = halo_pos[hb];  // some valid position
      pollB[j][0] = pollB[j][1] = pollB[j][2] = 0;
      xg[j] = 0.0;
    }
    for (int k = 0; k < nticks; k += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        constexpr int dummy = 0;
        (void)dummy;
        const int jm = (j + D - 1) % D, jn = (j + 1) % D;
        // (1) the tick itself
        const RunPre<W> p = pre[j];
        const int2 tn = ti[j];
        ti[j] = tk[k + j + 2 * D];
        const int4 e2 = te[k + j + 2 * D];  // (requested here, used in (5): a scalar load's round trip off the chain)
        double xs[W];
#pragma unroll
        for (int s = 0; s < W; ++s) {
          const unsigned c = (p.dw[s >> 1] >> ((s & 1) * 16)) & 0xffffu;
          xs[s] = (c & 0x8000u) ? halo[c & 0x7fffu] : ring[(p.u - (int)c) & (kBrickRing - 1)];
        }
        // (2) the static data of tick k + j + D
        issue(pre[j], tn);
        if (p.u >= 0) {
          double acc = 0.0;
#pragma unroll
          for (int s = 0; s < W; ++s) {
            const unsigned c = (p.dw[s >> 1] >> ((s & 1) * 16)) & 0xffffu;
            const double vs = (s & 1) ? p.v[s >> 1].y : p.v[s >> 1].x;
            const double tt = MINUS ? acc - vs * xs[s] : acc + vs * xs[s];
            acc = c ? tt : acc;
          }
          double xn, yn;
          if constexpr (!MINUS) {
            xn = (p.gd.x - acc) / p.gd.y;
            yn = acc;
          } else {
            const double hi = omega * acc;
            yn = hi;
            xn = (p.gd.x + hi) / p.gd.y;
          }
          ring[p.u & (kBrickRing - 1)] = xn;
          __hip_atomic_store(x + p.t, xn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          y[p.t] = yn;
        }
        // (3) the gathers of tick k + j + D - 1: its check was made before the previous barrier
        xg[jm] = __hip_atomic_load(x + hpC[jm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (4) tick k + j + D: are its predecessors far enough?  (polled D ticks ago; wave 0 waits if not, the others at
        // the barrier below)
        if (wid == 0) {
          const int z = __builtin_amdgcn_readfirstlane(teB[j].z);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const int nd = (z >> (10 * q)) & 1023;
            if (nd > 0 && ps[q] >= 0) wait_for(q, nd, __builtin_amdgcn_readfirstlane(pollB[j][q]));
          }
        }
        teC[j] = teB[j];
        hpC[j] = hpB[j];
        // (5) tick k + j + 2 D: where its halo entries come from, and a poll of the predecessors
        {
          const int4 e = e2;
          teB[j] = e;
          hpB[j] = halo_pos[hb + e.x + min(tid, max(e.y - 1, 0))];
          if (wid == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
              pollB[j][q] = ps[q] >= 0 ? __hip_atomic_load(PSP_FLAG_BRICK(flags, ps[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
          }
        }
        // (6) the halo entries of the next tick into LDS
        if (tid < teC[jn].y) halo[teC[jn].x + tid] = xg[jn];
        // (7) every wave: at most kOps vector-memory operations outstanding, i.e. the stores of tick k + j - D have left
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kOps));
        lds_barrier();
        // (8) ... so ticks 0 .. k + j - D are complete
        if (tid == 0 && k + j >= D)
          __hip_atomic_store(PSP_FLAG_BRICK(flags, b), min(k + j - D + 1, nticks), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its last ticks' stores have left (see above)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(PSP_FLAG_BRICK(flags, b), nticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
int build_schedule(const psp_csr *F, int dir, int **rows_out, std::vector<int> *ptr_out, int **level_out = nullptr) {
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
    if (level_out) {  // the caller keeps the level of every row (brick ordering)
      *level_out = level;
      level = nullptr;
    }
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

// helper workgroups beside the walker of a run (PSP_SSOR_LDS_HELPERS, tuning only)
static int run_helpers() {
  static const int h = [] {
    const char *e = psp::tuning_env("PSP_SSOR_LDS_HELPERS");
    return e ? std::max(0, std::min(atoi(e), 15)) : 3;
  }();
  return h;
}

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
  if constexpr (W >= 1 && W <= 3) {
    if (K->brick_mode) {  // a 3-D grid operator with wide levels: bricks of 32^3 points, a coarse wavefront of workgroups
      const psp_ssor::BrickSet &bs = forward ? K->brick_f : K->brick_b;
      const int n = K->n, nwg = std::min(bs.nbricks, kBrickWgs);
      hipLaunchKernelGGL(brick_begin_kernel, dim3((int)((flag_words(bs.nbricks) + 255) / 256)), dim3(256), 0, st,
                         (int)flag_words(bs.nbricks), bs.flags);
      if (!(KIND == 0 && first))  // (the application's first Gauss-Seidel sweep got its G from brick_first_gd_kernel)
        hipLaunchKernelGGL(run_pre_kernel<KIND>, dim3((n + 255) / 256), dim3(256), 0, st, 0, n, 0, rowmap, K->bp, K->xp,
                           K->temp, K->da, K->omega, first, bs.dar, bs.gd, PSP_FLAG_SCRATCH(bs.flags));
      if (K->brick_pipe) {
        if (forward)
          hipLaunchKernelGGL((ssor_brick_pipe_kernel<(KIND >= 2), false, W, kPipeDepth>), dim3(nwg), dim3(kBrickTick), 0, st,
                             bs.nbricks, bs.info, bs.pred, bs.ticks, bs.tick_ext, bs.halo_pos, n, rowmap, bs.vp, bs.dpk,
                             bs.gd, K->xp, K->temp, K->omega, bs.flags);
        else
          hipLaunchKernelGGL((ssor_brick_pipe_kernel<(KIND >= 2), true, W, kPipeDepth>), dim3(nwg), dim3(kBrickTick), 0, st,
                             bs.nbricks, bs.info, bs.pred, bs.ticks, bs.tick_ext, bs.halo_pos, n, rowmap, bs.vp, bs.dpk,
                             bs.gd, K->xp, K->temp, K->omega, bs.flags);
        return;
      }
      if (forward)
        hipLaunchKernelGGL((ssor_brick_kernel<(KIND >= 2), false, W, run_depth<W>()>), dim3(nwg), dim3(kBrickTick), 0, st,
                           bs.nbricks, bs.info, bs.pred, bs.ticks, bs.halo_pos, n, rowmap, bs.vp, bs.dpk, bs.gd, K->xp,
                           K->temp, K->omega, bs.flags);
      else
        hipLaunchKernelGGL((ssor_brick_kernel<(KIND >= 2), true, W, run_depth<W>()>), dim3(nwg), dim3(kBrickTick), 0, st,
                           bs.nbricks, bs.info, bs.pred, bs.ticks, bs.halo_pos, n, rowmap, bs.vp, bs.dpk, bs.gd, K->xp,
                           K->temp, K->omega, bs.flags);
      return;
    }
  }
  const int nl = (int)lp.size() - 1;
  const psp_ssor::RunSet &rs = forward ? K->run_f : K->run_b;
  size_t ri = 0;
  int l = 0;
  while (l < nl) {
    if constexpr (W > 0) {
      if (ri < rs.runs.size() && rs.runs[ri].l0 == l) {  // a run of narrow levels: one workgroup, x through LDS
        const psp_ssor::Run &r = rs.runs[ri++];
        const int cnt = r.s1 - r.s0, nh = run_helpers();
        hipLaunchKernelGGL(run_pre_kernel<KIND>, dim3((cnt + 255) / 256), dim3(256), 0, st, r.s0, r.s1, r.coff, rowmap,
                           K->bp, K->xp, K->temp, K->da, K->omega, first, rs.dar, rs.gd, K->run_progress);
        if (forward)
          hipLaunchKernelGGL((ssor_run_kernel<(KIND >= 2), false, W, run_depth<W>()>), dim3(8 * (1 + nh)), dim3(kRunTick),
                             0, st, r.nticks, rs.ticks + r.tick0, K->n, rs.m, r.s0, r.coff, rowmap, rs.vp, rs.dpk, rs.gd,
                             K->xp, K->temp, K->omega, K->run_progress, nh);
        else
          hipLaunchKernelGGL((ssor_run_kernel<(KIND >= 2), true, W, run_depth<W>()>), dim3(8 * (1 + nh)), dim3(kRunTick),
                             0, st, r.nticks, rs.ticks + r.tick0, K->n, rs.m, r.s0, r.coff, rowmap, rs.vp, rs.dpk, rs.gd,
                             K->xp, K->temp, K->omega, K->run_progress, nh);
        l = r.l1;
        continue;
      }
    }
    int e = l;  // maximal run of small levels starting at l
    while (e < nl && lp[e + 1] - lp[e] <= kSmallLevel && !(ri < rs.runs.size() && rs.runs[ri].l0 == e)) ++e;
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
      sweep<0>(K, st, true, (K->brick_mode && step == 0) ? 1 : 0);
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

static int brick_error_check(psp_ssor *K, bool wait) {
  if (!K->brick_mode || !K->brick_ev_pending) return PSP_OK;
  const hipError_t e = wait ? hipEventSynchronize(K->brick_ev) : hipEventQuery(K->brick_ev);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    return PSP_OK;
  }
  K->brick_ev_pending = false;
  if (e != hipSuccess) return fail(PSP_ENODEV, "ssor: %s", hipGetErrorString(e));
  if (K->brick_err_host[0] | K->brick_err_host[1]) {
    // reported once: the words are cleared so that the handle can be used again
    K->brick_err_host[0] = K->brick_err_host[1] = 0;
    (void)hipMemsetAsync(PSP_FLAG_ERR(K->brick_f.flags), 0, sizeof(int), stream());
    (void)hipMemsetAsync(PSP_FLAG_ERR(K->brick_b.flags), 0, sizeof(int), stream());
    return fail(PSP_ENODEV, "ssor: a brick sweep gave up waiting for its predecessors; the preconditioned vector is invalid");
  }
  return PSP_OK;
}

int ssor_apply_dev(psp_ssor *K, const double *b, double *x) {
  if (K->steps <= 0) return PSP_OK;  // the reference leaves y untouched
  PSP_TRY(brick_error_check(K, false));
  if (K->brick_mode && K->omega == 1.0) {
    // bp[t] = b[pos2row[t]] and the first sweep's (G, diagonal) in one pass; y need not be cleared (every row of the
    // first sweep writes its y before anything reads it)
    hipLaunchKernelGGL(brick_first_gd_kernel, dim3((K->n + 255) / 256), dim3(256), 0, stream(), K->n, K->pos2row, b,
                       K->brick_f.dar, K->bp, K->brick_f.gd);
  } else {
    PSP_TRY(reorder_gather(K->n, K->pos2row, b, K->bp, nullptr));  // bp[t] = b[pos2row[t]]
    if (K->omega == 1.0) PSP_HIP(hipMemsetAsync(K->temp, 0, sizeof(double) * (size_t)K->n, stream()));  // :164-165
  }
  ensure_graph(K);
  if (K->graph_state == 1 && hipGraphLaunch(K->exec, stream()) != hipSuccess) {
    (void)hipGetLastError();
    K->graph_state = 0;  // this runtime does not replay into the library's stream: direct launches from now on
  }
  if (K->graph_state != 1) {
    enqueue_sweeps(K, stream());
    PSP_LAUNCH_CHECK();
  }
  if (K->brick_mode && K->brick_err_host && K->brick_ev) {
    (void)hipMemcpyAsync(K->brick_err_host, PSP_FLAG_ERR(K->brick_f.flags), sizeof(int), hipMemcpyDeviceToHost,
                         stream());
    (void)hipMemcpyAsync(K->brick_err_host + 1, PSP_FLAG_ERR(K->brick_b.flags), sizeof(int),
                         hipMemcpyDeviceToHost, stream());
    K->brick_ev_pending = hipEventRecord(K->brick_ev, stream()) == hipSuccess;
  }
  return reorder_gather(K->n, K->row2pos, K->xp, x, nullptr);  // x[i] = xp[row2pos[i]]
}

int ssor_error_check(psp_ssor *K) { return brick_error_check(K, true); }

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

// runs of narrow levels for ssor_run_kernel (see there).  Never fails the handle: whatever goes wrong here leaves
// the direction on its per-level launches.
template <int W>
void build_runs_w(psp_ssor *K, int dir) {
  const std::vector<int> &lp = dir ? K->ptr_b : K->ptr_f;
  const int *dlp = dir ? K->dptr_b : K->dptr_f;
  const int *pos = dir ? K->b_pos : K->f_pos;
  const unsigned char *c8 = dir ? K->bc8 : K->fc8;
  const int *rowmap = dir ? K->b_row : nullptr;
  psp_ssor::RunSet &rs = dir ? K->run_b : K->run_f;
  const int nl = (int)lp.size() - 1, n = K->n;
  // candidates: maximal stretches of levels of at most kRunWide rows
  std::vector<std::pair<int, int>> cand;
  for (int l = 0; l < nl;) {
    if (lp[l + 1] - lp[l] > kRunWide) {
      ++l;
      continue;
    }
    int e = l;
    while (e < nl && lp[e + 1] - lp[e] <= kRunWide) ++e;
    if (e - l >= kRunMinLevels) cand.emplace_back(l, e);
    l = e;
  }
  if (cand.empty()) return;
  int *slot_of = nullptr, *lev_maxd = nullptr;
  std::vector<int> maxd((size_t)nl, 0);
  bool ok = true;
  if (dir) {  // a backward row's dependencies are positions; the ring is indexed by backward slots
    ok = hipMalloc((void **)&slot_of, sizeof(int) * (size_t)n) == hipSuccess;
    if (ok) hipLaunchKernelGGL(invert_perm_kernel, dim3(std::min((n + 255) / 256, 65536)), dim3(256), 0, stream(), n,
                               K->b_row, slot_of);
  }
  ok = ok && hipMalloc((void **)&lev_maxd, sizeof(int) * (size_t)nl) == hipSuccess &&
       hipMemsetAsync(lev_maxd, 0, sizeof(int) * (size_t)nl, stream()) == hipSuccess;
  if (ok) {
    for (const auto &c : cand) {
      const int a = lp[c.first], e = lp[c.second];
      hipLaunchKernelGGL(run_dist_kernel<W>, dim3((e - a + 255) / 256), dim3(256), 0, stream(), a, e, c.first, c.second,
                         dlp, n, c8, pos, slot_of, lev_maxd);
    }
    ok = hipGetLastError() == hipSuccess &&
         hipMemcpyAsync(maxd.data(), lev_maxd, sizeof(int) * (size_t)nl, hipMemcpyDeviceToHost, stream()) == hipSuccess &&
         hipStreamSynchronize(stream()) == hipSuccess;
  }
  if (ok) {
    // split the candidates at levels that reach further back than the ring allows
    std::vector<int2> ticks;
    for (const auto &c : cand) {
      for (int l = c.first; l < c.second;) {
        if (maxd[l] > kRunMaxDist) {
          ++l;
          continue;
        }
        int e = l;
        while (e < c.second && maxd[e] <= kRunMaxDist) ++e;
        if (e - l >= kRunMinLevels) {
          psp_ssor::Run r;
          r.l0 = l;
          r.l1 = e;
          r.s0 = lp[l];
          r.s1 = lp[e];
          r.coff = rs.m;
          r.tick0 = (int)ticks.size();
          for (int q = l; q < e; ++q)
          {  // a level of w rows = ceil(w / 1024) ticks of (nearly) equal size
            const int w = lp[q + 1] - lp[q], nt = (w + kRunTick - 1) / kRunTick;
            for (int i = 0, a = lp[q]; i < nt; ++i) {
              const int c = w / nt + (i < w % nt ? 1 : 0);
              ticks.push_back(make_int2(a, c));
              a += c;
            }
          }
          r.nticks = (int)ticks.size() - r.tick0;
          for (int q = 0; q < kRunPad; ++q) ticks.push_back(make_int2(r.s0, 0));  // empty, but a valid slot to load from
          rs.m += r.s1 - r.s0;
          rs.runs.push_back(r);
        }
        l = e;
      }
    }
    if (!rs.runs.empty()) {
      constexpr int DW = (W + 1) / 2;
      ok = hipMalloc((void **)&rs.dpk, sizeof(unsigned) * (size_t)DW * rs.m) == hipSuccess &&
           hipMalloc((void **)&rs.vp, sizeof(double2) * (size_t)DW * rs.m) == hipSuccess &&
           hipMalloc((void **)&rs.dar, sizeof(double) * (size_t)rs.m) == hipSuccess &&
           hipMalloc((void **)&rs.gd, sizeof(double2) * (size_t)rs.m) == hipSuccess &&
           hipMalloc((void **)&rs.ticks, sizeof(int2) * ticks.size()) == hipSuccess &&
           hipMemcpyAsync(rs.ticks, ticks.data(), sizeof(int2) * ticks.size(), hipMemcpyHostToDevice, stream()) ==
               hipSuccess;
      if (ok && !K->run_progress)
        ok = hipMalloc((void **)&K->run_progress, 2 * sizeof(int)) == hipSuccess &&
             hipMemsetAsync(K->run_progress, 0, 2 * sizeof(int), stream()) == hipSuccess;
      if (ok) {
        for (const psp_ssor::Run &r : rs.runs)
          hipLaunchKernelGGL(run_pack_kernel<W>, dim3((r.s1 - r.s0 + 255) / 256), dim3(256), 0, stream(), r.s0, r.s1,
                             r.coff, rs.m, n, c8, pos, slot_of, rowmap, K->da, dir ? K->b_val : K->f_val, rs.dpk, rs.vp,
                             rs.dar);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(stream()) == hipSuccess;
      }
    }
  }
  (void)hipFree(slot_of);
  (void)hipFree(lev_maxd);
  if (!ok) {
    (void)hipGetLastError();
    (void)hipFree(rs.dpk);
    (void)hipFree(rs.vp);
    (void)hipFree(rs.dar);
    (void)hipFree(rs.gd);
    (void)hipFree(rs.ticks);
    rs = psp_ssor::RunSet();
  }
}

// ---- bricks (see ssor_brick_kernel): the plan of one direction, made before anything is committed
struct BrickPlan {
  int nb = 0;
  int *rows = nullptr;            // device: rows in (brick, level, row) order = the direction's slot order
  std::vector<int> brick_start;   // slot ranges by processing index (nb + 1)
  std::vector<int2> ticks;
  std::vector<int> tick_brick;    // per (padded) tick: the brick it belongs to
  std::vector<int4> info, pred;
  std::vector<int> halo_base;     // nb + 1
  ~BrickPlan() { (void)hipFree(rows); }
};

struct GridShape {
  int nx = 0, ny = 0, nz = 0;
};

// is the strict lower triangle that of a 3-D grid operator in natural ordering (offsets 1, nx, nx * ny; nothing wraps)?
bool detect_grid(const psp_sss *S, GridShape *g) {
  const int n = S->n;
  if (n < 8 || S->nnz_lower < 1) return false;
  int *mm = nullptr;
  if (hipMalloc((void **)&mm, 5 * sizeof(int)) != hipSuccess) return false;
  const int init[5] = {0x7fffffff, 0, 0x7fffffff, 0, 0};
  int h[5] = {0, 0, 0, 0, 0};
  const dim3 grid((n + 255) / 256), block(256);
  bool ok = hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, stream()) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(grid_offsets_kernel, grid, block, 0, stream(), n, S->ind, S->col, 0, mm);
    hipLaunchKernelGGL(grid_offsets_kernel, grid, block, 0, stream(), n, S->ind, S->col, 1, mm);
    ok = hipMemcpyAsync(h, mm, sizeof(h), hipMemcpyDeviceToHost, stream()) == hipSuccess &&
         hipStreamSynchronize(stream()) == hipSuccess;
  }
  // smallest offset 1, exactly one value strictly between the smallest and the largest
  ok = ok && h[0] == 1 && h[2] == h[3] && h[2] > 1 && h[1] > h[2] && h[1] % h[2] == 0 && n % h[1] == 0;
  if (ok) {
    g->nx = h[2];
    g->ny = h[1] / h[2];
    g->nz = n / h[1];
    ok = g->ny >= 2 && g->nz >= 2;
  }
  if (ok) {
    hipLaunchKernelGGL(grid_verify_kernel, grid, block, 0, stream(), n, g->nx, g->nx * g->ny, S->ind, S->col, mm + 4);
    ok = hipMemcpyAsync(h, mm, sizeof(h), hipMemcpyDeviceToHost, stream()) == hipSuccess &&
         hipStreamSynchronize(stream()) == hipSuccess && h[4] == 0;
  }
  (void)hipGetLastError();
  (void)hipFree(mm);
  return ok;
}

// one direction's plan; `level` = the level of every row in this direction.  false: not feasible (nothing changed)
bool plan_bricks(const psp_sss *S, const GridShape &g, int dir, const int *level, BrickPlan *P) {
  const int n = S->n, nx = g.nx, ny = g.ny, nxy = g.nx * g.ny;
  const int ba = (g.nx + kBrickEdge - 1) / kBrickEdge, bb = (g.ny + kBrickEdge - 1) / kBrickEdge,
            bc = (g.nz + kBrickEdge - 1) / kBrickEdge;
  const int nb = ba * bb * bc;
  // processing order: coarse wavefront a + b + c (forward ascending, backward the reverse)
  std::vector<int> order((size_t)nb), rank((size_t)nb);
  for (int i = 0; i < nb; ++i) order[(size_t)i] = i;
  auto wave = [&](int id) { return id % ba + (id / ba) % bb + id / (ba * bb); };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wave(a) < wave(b); });
  if (dir) std::reverse(order.begin(), order.end());
  for (int i = 0; i < nb; ++i) rank[(size_t)order[(size_t)i]] = i;
  int *d_rank = nullptr, *d_iota = nullptr, *d_slot = nullptr, *d_bstart = nullptr, *d_flag = nullptr, *d_starts = nullptr,
      *d_cnt = nullptr, *d_ext = nullptr;
  unsigned long long *d_key = nullptr, *d_key2 = nullptr;
  void *tmp = nullptr;
  bool ok = true;
  const dim3 grid((n + 255) / 256), block(256);
  auto A = [&](void **p, size_t bytes) { ok = ok && hipMalloc(p, bytes ? bytes : 1) == hipSuccess; };
  A((void **)&d_rank, sizeof(int) * (size_t)nb);
  A((void **)&d_iota, sizeof(int) * (size_t)n);
  A((void **)&P->rows, sizeof(int) * (size_t)n);
  A((void **)&d_key, sizeof(unsigned long long) * (size_t)n);
  A((void **)&d_key2, sizeof(unsigned long long) * (size_t)n);
  A((void **)&d_slot, sizeof(int) * (size_t)n);
  A((void **)&d_bstart, sizeof(int) * ((size_t)nb + 1));
  A((void **)&d_flag, sizeof(int) * (size_t)n);
  A((void **)&d_starts, sizeof(int) * (size_t)n);
  A((void **)&d_cnt, sizeof(int) * 2);
  A((void **)&d_ext, sizeof(int) * (size_t)nb);
  ok = ok && hipMemcpyAsync(d_rank, rank.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice, stream()) == hipSuccess;
  int nbits = 1;
  while ((1 << nbits) < nb) ++nbits;
  if (ok) {
    hipLaunchKernelGGL(brick_key_kernel, grid, block, 0, stream(), n, nx, ny, nxy, ba, bb, d_rank, level, d_key, d_iota);
    size_t bytes = 0;
    ok = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, d_key, d_key2, d_iota, P->rows, n, 0, 32 + nbits, stream()) ==
         hipSuccess;
    A(&tmp, bytes);
    ok = ok && hipcub::DeviceRadixSort::SortPairs(tmp, bytes, d_key, d_key2, d_iota, P->rows, n, 0, 32 + nbits, stream()) ==
                   hipSuccess;  // stable: rows ascending inside a (brick, level) group
  }
  int ngroups = 0;
  std::vector<int> starts, ext;
  if (ok) {
    hipLaunchKernelGGL(invert_perm_kernel, dim3(std::min((n + 255) / 256, 65536)), dim3(256), 0, stream(), n, P->rows,
                       d_slot);
    hipLaunchKernelGGL(brick_bounds_kernel, grid, block, 0, stream(), n, d_key2, d_bstart, d_flag);
    ok = hipMemcpyAsync(d_bstart + nb, &n, sizeof(int), hipMemcpyHostToDevice, stream()) == hipSuccess &&
         hipMemsetAsync(d_ext, 0, sizeof(int) * (size_t)nb, stream()) == hipSuccess &&
         hipMemsetAsync(d_cnt, 0, 2 * sizeof(int), stream()) == hipSuccess;
  }
  if (ok) {
    hipLaunchKernelGGL(brick_check_kernel, grid, block, 0, stream(), n, dir, S->ind, S->col, d_slot, nb, d_bstart, d_ext,
                       d_cnt + 1);
    (void)hipFree(tmp);
    tmp = nullptr;
    size_t bytes = 0;
    ok = hipcub::DeviceSelect::Flagged(nullptr, bytes, d_iota, d_flag, d_starts, d_cnt, n, stream()) == hipSuccess;
    A(&tmp, bytes);
    // d_iota was the sort's input buffer and still holds 0 .. n-1
    ok = ok && hipcub::DeviceSelect::Flagged(tmp, bytes, d_iota, d_flag, d_starts, d_cnt, n, stream()) == hipSuccess;
    int cnt[2] = {0, 0};
    ok = ok && hipMemcpyAsync(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, stream()) == hipSuccess &&
         hipStreamSynchronize(stream()) == hipSuccess && cnt[1] == 0 && cnt[0] > 0;
    ngroups = cnt[0];
  }
  if (ok) {
    starts.resize((size_t)ngroups + 1);
    ext.resize((size_t)nb);
    P->brick_start.resize((size_t)nb + 1);
    ok = hipMemcpy(starts.data(), d_starts, sizeof(int) * (size_t)ngroups, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(ext.data(), d_ext, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(P->brick_start.data(), d_bstart, sizeof(int) * ((size_t)nb + 1), hipMemcpyDeviceToHost) == hipSuccess;
    starts[(size_t)ngroups] = n;
  }
  if (ok) {
    P->nb = nb;
    P->halo_base.assign((size_t)nb + 1, 0);
    for (int b = 0; b < nb && ok; ++b) {
      ok = ext[(size_t)b] <= kBrickHalo && ext[(size_t)b] < 0x8000;
      P->halo_base[(size_t)b + 1] = P->halo_base[(size_t)b] + ext[(size_t)b];
    }
  }
  if (ok) {
    P->info.resize((size_t)nb);
    P->pred.resize((size_t)nb);
    size_t gi = 0;
    for (int b = 0; b < nb; ++b) {
      const int bs = P->brick_start[(size_t)b], be = P->brick_start[(size_t)b + 1];
      const int t0 = (int)P->ticks.size();
      while (gi < (size_t)ngroups && starts[gi] < be) {  // the (brick, level) groups of this brick
        const int a = starts[gi], e = std::min(starts[gi + 1], be), w = e - a, nt = (w + kBrickTick - 1) / kBrickTick;
        for (int i = 0, u = a; i < nt; ++i) {
          const int c = w / nt + (i < w % nt ? 1 : 0);
          P->ticks.push_back(make_int2(u, c));
          u += c;
        }
        ++gi;
      }
      const int nt = (int)P->ticks.size() - t0;
      for (int q = 0; q < kRunPad; ++q) P->ticks.push_back(make_int2(bs, 0));
      P->tick_brick.resize(P->ticks.size(), b);
      P->info[(size_t)b] = make_int4(t0, nt, P->halo_base[(size_t)b], ext[(size_t)b]);
      // the three face neighbours this brick waits for
      const int id = order[(size_t)b];
      const int ia = id % ba, ib = (id / ba) % bb, ic = id / (ba * bb);
      const int st = dir ? 1 : -1;
      int pr[3] = {-1, -1, -1};
      if (ia + st >= 0 && ia + st < ba) pr[0] = rank[(size_t)(id + st)];
      if (ib + st >= 0 && ib + st < bb) pr[1] = rank[(size_t)(id + st * ba)];
      if (ic + st >= 0 && ic + st < bc) pr[2] = rank[(size_t)(id + st * ba * bb)];
      P->pred[(size_t)b] = make_int4(pr[0], pr[1], pr[2], 0);
    }
  }
  (void)hipGetLastError();
  for (void *q : {(void *)d_rank, (void *)d_iota, (void *)d_key, (void *)d_key2, (void *)d_slot, (void *)d_bstart,
                  (void *)d_flag, (void *)d_starts, (void *)d_cnt, (void *)d_ext, tmp})
    (void)hipFree(q);
  return ok;
}

// pipelined brick sweeps (a brick starts before its predecessors have finished); PSP_SSOR_BRICK_PIPE=0 under PSP_TUNING
// keeps the whole-brick waits of ssor_brick_kernel
bool brick_pipe_wanted() {
  static const int mode = [] {
    const char *e = psp::tuning_env("PSP_SSOR_BRICK_PIPE");
    return e ? atoi(e) : 1;
  }();
  return mode != 0;
}

// after build_level_ordered (whose slot orders came from the plans): the arrays the brick kernel streams
template <int W>
int finish_bricks_w(psp_ssor *K, int dir, const BrickPlan &P) {
  psp_ssor::BrickSet &bs = dir ? K->brick_b : K->brick_f;
  const int n = K->n, nb = P.nb;
  constexpr int DW = (W + 1) / 2;
  const int *pos = dir ? K->b_pos : K->f_pos;
  const double *val = dir ? K->b_val : K->f_val;
  const unsigned char *c8 = dir ? K->bc8 : K->fc8;
  const int *rowmap = dir ? K->b_row : nullptr;
  int *slot_of = nullptr, *d_bstart = nullptr, *d_ext = nullptr, *d_eoff = nullptr;
  void *scan_tmp = nullptr;
  int rc = PSP_OK;
  bs.nbricks = nb;
  const int nhalo = P.halo_base[(size_t)nb];
  bool ok = hipMalloc((void **)&bs.ticks, sizeof(int2) * P.ticks.size()) == hipSuccess &&
            hipMalloc((void **)&bs.info, sizeof(int4) * (size_t)nb) == hipSuccess &&
            hipMalloc((void **)&bs.pred, sizeof(int4) * (size_t)nb) == hipSuccess &&
            hipMalloc((void **)&bs.halo_pos, sizeof(int) * ((size_t)nhalo + 1)) == hipSuccess &&  // (+1: padding ticks read one)
            hipMalloc((void **)&bs.dpk, sizeof(unsigned) * (size_t)DW * n) == hipSuccess &&
            hipMalloc((void **)&bs.vp, sizeof(double2) * (size_t)DW * n) == hipSuccess &&
            hipMalloc((void **)&bs.dar, sizeof(double) * (size_t)n) == hipSuccess &&
            hipMalloc((void **)&bs.gd, sizeof(double2) * (size_t)n) == hipSuccess &&
            hipMalloc((void **)&bs.flags, sizeof(int) * flag_words(nb)) == hipSuccess &&
            hipMalloc((void **)&d_bstart, sizeof(int) * ((size_t)nb + 1)) == hipSuccess &&
            hipMalloc((void **)&d_ext, sizeof(int) * ((size_t)n + 1)) == hipSuccess &&
            hipMalloc((void **)&d_eoff, sizeof(int) * ((size_t)n + 1)) == hipSuccess &&
            (!dir || hipMalloc((void **)&slot_of, sizeof(int) * (size_t)n) == hipSuccess);
  ok = ok &&
       hipMemcpyAsync(bs.ticks, P.ticks.data(), sizeof(int2) * P.ticks.size(), hipMemcpyHostToDevice, stream()) ==
           hipSuccess &&
       hipMemcpyAsync(bs.info, P.info.data(), sizeof(int4) * (size_t)nb, hipMemcpyHostToDevice, stream()) == hipSuccess &&
       hipMemcpyAsync(bs.pred, P.pred.data(), sizeof(int4) * (size_t)nb, hipMemcpyHostToDevice, stream()) == hipSuccess &&
       hipMemcpyAsync(d_bstart, P.brick_start.data(), sizeof(int) * ((size_t)nb + 1), hipMemcpyHostToDevice, stream()) ==
           hipSuccess &&
       hipMemsetAsync(bs.flags, 0, sizeof(int) * flag_words(nb), stream()) == hipSuccess &&
       hipMemsetAsync(bs.halo_pos + nhalo, 0, sizeof(int), stream()) == hipSuccess;
  if (ok) {
    if (dir)
      hipLaunchKernelGGL(invert_perm_kernel, dim3(std::min((n + 255) / 256, 65536)), dim3(256), 0, stream(), n, K->b_row,
                         slot_of);
    hipLaunchKernelGGL(brick_extcnt_kernel<W>, dim3((n + 256) / 256), dim3(256), 0, stream(), n, c8, pos, slot_of, nb,
                       d_bstart, d_ext);
    size_t bytes = 0;
    ok = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_ext, d_eoff, n + 1, stream()) == hipSuccess &&
         hipMalloc(&scan_tmp, bytes ? bytes : 1) == hipSuccess &&
         hipcub::DeviceScan::ExclusiveSum(scan_tmp, bytes, d_ext, d_eoff, n + 1, stream()) == hipSuccess;
  }
  if (ok) {
    hipLaunchKernelGGL(brick_pack_kernel<W>, dim3((n + 255) / 256), dim3(256), 0, stream(), n, c8, pos, val, slot_of,
                       rowmap, K->da, nb, d_bstart, d_eoff, bs.halo_pos, bs.dpk, bs.vp, bs.dar);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(stream()) == hipSuccess;
  }
  if (ok && brick_pipe_wanted()) {  // never fails the handle: without tick_ext the sweeps wait for whole bricks
    const int nt = (int)P.ticks.size();
    int *d_tb = nullptr, *d_tl = nullptr, *d_need = nullptr, *d_bad = nullptr;
    int bad = 1;
    bool pk = hipMalloc((void **)&d_tb, sizeof(int) * (size_t)nt) == hipSuccess &&
              hipMalloc((void **)&d_tl, sizeof(int) * (size_t)n) == hipSuccess &&
              hipMalloc((void **)&d_need, sizeof(int) * 3 * (size_t)nt) == hipSuccess &&
              hipMalloc((void **)&d_bad, sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&bs.tick_ext, sizeof(int4) * (size_t)nt) == hipSuccess &&
              hipMemcpyAsync(d_tb, P.tick_brick.data(), sizeof(int) * (size_t)nt, hipMemcpyHostToDevice, stream()) ==
                  hipSuccess &&
              hipMemsetAsync(d_need, 0, sizeof(int) * 3 * (size_t)nt, stream()) == hipSuccess &&
              hipMemsetAsync(d_bad, 0, sizeof(int), stream()) == hipSuccess;
    if (pk) {
      hipLaunchKernelGGL(brick_ticklocal_kernel, dim3(nt), dim3(256), 0, stream(), nt, bs.ticks, d_tb, bs.info, d_tl);
      hipLaunchKernelGGL(brick_need_kernel<W>, dim3((n + 255) / 256), dim3(256), 0, stream(), n, c8, pos, slot_of, nb,
                         d_bstart, bs.info, bs.pred, d_tl, d_need, d_bad);
      hipLaunchKernelGGL(brick_tickext_kernel, dim3((nt + 255) / 256), dim3(256), 0, stream(), nt, bs.ticks, d_tb, d_bstart,
                         d_eoff, d_need, bs.tick_ext, d_bad);
      pk = hipGetLastError() == hipSuccess &&
           hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream()) == hipSuccess &&
           hipStreamSynchronize(stream()) == hipSuccess && bad == 0;
    }
    if (!pk) {
      (void)hipGetLastError();
      (void)hipFree(bs.tick_ext);
      bs.tick_ext = nullptr;
    }
    (void)hipFree(d_tb);
    (void)hipFree(d_tl);
    (void)hipFree(d_need);
    (void)hipFree(d_bad);
  }
  if (!ok) rc = fail(PSP_ENOMEM, "ssor: the brick schedule could not be built");
  (void)hipFree(slot_of);
  (void)hipFree(d_bstart);
  (void)hipFree(d_ext);
  (void)hipFree(d_eoff);
  (void)hipFree(scan_tmp);
  return rc;
}

int finish_bricks(psp_ssor *K, int dir, const BrickPlan &P) {
  switch (dir ? K->ell_b : K->ell_f) {
    case 1: return finish_bricks_w<1>(K, dir, P);
    case 2: return finish_bricks_w<2>(K, dir, P);
    case 3: return finish_bricks_w<3>(K, dir, P);
    default: return fail(PSP_EINVAL, "ssor: a grid operator with %d entries per sweep row", dir ? K->ell_b : K->ell_f);
  }
}

// bricks are for schedules the runs do not cover: some level wider than a run takes (PSP_SSOR_BRICK=0 / 1 under
// PSP_TUNING: never / whenever the operator is a 3-D grid operator)
bool bricks_wanted(const std::vector<int> &ptr_f) {
  static const int mode = [] {
    const char *e = psp::tuning_env("PSP_SSOR_BRICK");
    return e ? atoi(e) : -1;
  }();
  if (mode == 0) return false;
  if (mode > 0) return true;
  for (size_t l = 0; l + 1 < ptr_f.size(); ++l)
    if (ptr_f[l + 1] - ptr_f[l] > kRunWide) return true;
  return false;
}

void build_runs(psp_ssor *K) {
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_SSOR_LDS");
    return e && atoi(e) == 0;
  }();
  if (off) return;
  for (int dir = 0; dir < 2; ++dir) switch (dir ? K->ell_b : K->ell_f) {
      case 1: build_runs_w<1>(K, dir); break;
      case 2: build_runs_w<2>(K, dir); break;
      case 3: build_runs_w<3>(K, dir); break;
      case 4: build_runs_w<4>(K, dir); break;
      case 6: build_runs_w<6>(K, dir); break;
      case 8: build_runs_w<8>(K, dir); break;
      default: break;  // rows with more than 8 entries per sweep keep the ptr / pos / val form and its launches
    }
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

static int ssor_create_device(psp_sss_t *S, double omega, int steps, bool allow_bricks, psp_ssor_t **out);

extern "C" {

int psp_ssor_create(psp_sss_t *S, double omega, int steps, psp_ssor_t **out) {
  PSP_API_GUARD_H(S);
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
  return ssor_create_device(S, omega, steps, true, out);
}

// (allow_bricks = false: the retry after the brick arrays of a grid operator could not be built -- bricks are an
// optimisation, the handle must not fail because of them)
static int ssor_create_device(psp_sss_t *S, double omega, int steps, bool allow_bricks, psp_ssor_t **out) {
  psp_ssor *K = new psp_ssor();
  K->n = S->n;
  K->omega = omega;
  K->steps = steps;
  K->S = S;
  int rc = PSP_OK;
  if (S->n > 0) {
    int *rows_f = nullptr, *rows_b = nullptr, *lev_f = nullptr, *lev_b = nullptr;
    rc = build_schedule(S->full, 0, &rows_f, &K->ptr_f, &lev_f);
    if (rc == PSP_OK) rc = build_schedule(S->full, 1, &rows_b, &K->ptr_b, &lev_b);
    BrickPlan plan_f, plan_b;
    GridShape shape;
    if (rc == PSP_OK && allow_bricks && bricks_wanted(K->ptr_f) && detect_grid(S, &shape) &&
        plan_bricks(S, shape, 0, lev_f, &plan_f) &&
        plan_bricks(S, shape, 1, lev_b, &plan_b)) {
      // both directions are feasible: the slot orders become (brick, level, row)
      std::swap(rows_f, plan_f.rows);
      std::swap(rows_b, plan_b.rows);
      K->brick_mode = true;
    }
    (void)hipFree(lev_f);
    (void)hipFree(lev_b);
    if (rc == PSP_OK) {
      // backward slots are (level, row)-sorted rows; ptr_b indexes slots
      rc = build_level_ordered(K, rows_f, rows_b);
      rows_f = nullptr;  // owned by K now (pos2row), even on failure
    }
    (void)hipFree(rows_f);
    (void)hipFree(rows_b);
    if (rc == PSP_OK && K->brick_mode) {
      rc = finish_bricks(K, 0, plan_f);
      if (rc == PSP_OK) rc = finish_bricks(K, 1, plan_b);
      K->brick_pipe = rc == PSP_OK && K->brick_f.tick_ext && K->brick_b.tick_ext;
      if (rc == PSP_OK) {  // the brick sweeps stream their own copies: the slot-major triangle is not needed any more
        for (void **q : {(void **)&K->f_pos, (void **)&K->f_val, (void **)&K->b_pos, (void **)&K->b_val, (void **)&K->fc8,
                         (void **)&K->bc8}) {
          (void)hipFree(*q);
          *q = nullptr;
        }
      }
      if (rc != PSP_OK) {  // the slot orders are the bricks' by now: start over on the level schedule
        (void)hipGetLastError();
        psp_ssor_destroy(K);
        (void)psp_trim();
        return ssor_create_device(S, omega, steps, false, out);
      }
    }
    if (rc == PSP_OK && K->brick_mode) {
      if (hipHostMalloc((void **)&K->brick_err_host, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&K->brick_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();  // no error report then; the sweeps themselves do not need it
        if (K->brick_err_host) (void)hipHostFree(K->brick_err_host);
        K->brick_err_host = nullptr;
        K->brick_ev = nullptr;
      } else {
        K->brick_err_host[0] = K->brick_err_host[1] = 0;
      }
    }
    if (rc == PSP_OK) {
      const size_t bf = sizeof(int) * K->ptr_f.size(), bb = sizeof(int) * K->ptr_b.size();
      if (hipMalloc((void **)&K->dptr_f, bf) != hipSuccess || hipMalloc((void **)&K->dptr_b, bb) != hipSuccess ||
          hipMemcpy(K->dptr_f, K->ptr_f.data(), bf, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(K->dptr_b, K->ptr_b.data(), bb, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PSP_ENOMEM, "ssor: level table allocation failed");
    }
    if (rc == PSP_OK && !K->brick_mode) build_runs(K);
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
                  (void *)K->bc8, (void *)K->run_f.dpk, (void *)K->run_f.dar, (void *)K->run_f.gd, (void *)K->run_f.ticks,
                  (void *)K->run_f.vp, (void *)K->run_b.dpk, (void *)K->run_b.dar, (void *)K->run_b.gd,
                  (void *)K->run_b.ticks, (void *)K->run_b.vp,
                  (void *)K->run_progress})
    (void)hipFree(p);
  if (K->brick_ev) (void)hipEventDestroy(K->brick_ev);
  if (K->brick_err_host) (void)hipHostFree(K->brick_err_host);
  for (psp_ssor::BrickSet *b : {&K->brick_f, &K->brick_b})
    for (void *p : {(void *)b->ticks, (void *)b->info, (void *)b->pred, (void *)b->halo_pos, (void *)b->dpk, (void *)b->vp,
                    (void *)b->dar, (void *)b->gd, (void *)b->flags, (void *)b->tick_ext})
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

int psp_ssor_run_info(const psp_ssor_t *K, int *runs_forward, int *runs_backward, long *levels_in_runs,
                      long *slots_in_runs) {
  if (!K) return fail(PSP_EINVAL, "psp_ssor_run_info: NULL handle");
  long lv = 0;
  for (const psp_ssor::RunSet *rs : {&K->run_f, &K->run_b})
    for (const psp_ssor::Run &r : rs->runs) lv += r.l1 - r.l0;
  if (runs_forward) *runs_forward = (int)K->run_f.runs.size();
  if (runs_backward) *runs_backward = (int)K->run_b.runs.size();
  if (levels_in_runs) *levels_in_runs = lv;
  if (slots_in_runs) *slots_in_runs = (long)K->run_f.m + K->run_b.m;
  return PSP_OK;
}

int psp_ssor_brick_info(const psp_ssor_t *K, int *bricks, int *edge) {
  if (!K) return fail(PSP_EINVAL, "psp_ssor_brick_info: NULL handle");
  if (bricks) *bricks = K->brick_mode ? K->brick_f.nbricks : 0;
  if (edge) *edge = kBrickEdge;
  return PSP_OK;
}

int psp_ssor_precon_dev(psp_ssor_t *K, const double *x_dev, double *y_dev) {
  PSP_API_GUARD_H(K);
  if (!K || !x_dev || !y_dev) return fail(PSP_EINVAL, "psp_ssor_precon_dev: NULL argument");
  if (K->n == 0) return PSP_OK;
  return ssor_apply_dev(K, x_dev, y_dev);
}

int psp_ssor_precon(psp_ssor_t *K, const double *x_host, double *y_host) {
  PSP_API_GUARD_H(K);
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
  return psp::ssor_error_check(K);
}

}  // extern "C"
