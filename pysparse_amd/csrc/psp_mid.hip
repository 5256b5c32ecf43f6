// psp_mid.hip -- mid-size offset-structured systems: the whole PCG / MINRES loop as ONE cooperative kernel (round 5).
//
// Contents: pcg_mid_kernel / minres_mid_kernel (contiguous row blocks: 2-D grids and slim 3-D ones, up to 9 offsets, bit-exact
// with the launch-per-phase loops; CV = constant-coefficient forms), pcg_brick_kernel / minres_brick_kernel (the 7-offset
// operators of 3-D grids, the points dealt out in bricks; oracle parity), their plans and host loops.  What follows describes
// the row-block PCG kernel; the others say where they differ.
//
// Between 2^18 and 2^20 unknowns an iteration of the launch-per-phase loops (psp_solvers.hip) is five kernels that each
// sit on the ~5 us floor of a dependent launch: 47 us per PCG iteration at 1024^2 for 128 MB of traffic that the memory
// system moves in 21 (profiles/r4_extra_solvers_1024sq.txt; VERDICT r4 "What's weak" #2).  The small-system loops of
// psp_coop.hip stop at 2^18 rows because every gathered entry is a device-coherent load that bypasses the caches
// (84 MB of 64-byte sectors per iteration at 512^2: 21.8 us).  Here:
//
//   * one workgroup of 1024 or 512 threads per CU owns a CONTIGUOUS block of B = 2048 or 4096 rows for the whole solve; a
//     thread owns one to four pairs of adjacent rows: x, r, p, q of its rows, the rows' matrix entries (index-free layout
//     of csr_spmv_w4: up to 9 offsets) and their masks stay in registers -- NOTHING of the matrix or of x, r, q is read
//     again after the first iteration.  512 threads may keep 256 registers each: the 4096-row blocks and the 7-offset
//     operators run (nearly) without scratch memory that way (mid_block_threads);
//   * the direction vector p is exchanged through LDS: a window of H + B + H entries (H = largest |offset|); a row reads
//     the p entries it multiplies with from the window;
//   * what crosses workgroups is the residual r of the H rows at either end of a block, published with device-coherent
//     stores at the barrier of the r.r / r.z reduction; the neighbours form the halo entries of p themselves,
//     p[c] = z[c] + beta p_old[c] with z[c] = r[c] dinv[c], from the gathered r[c] and the p_old[c] their window still
//     holds -- the owner's own two rounded operations on the same operands.  Two grid barriers per iteration;
//   * per iteration the kernel touches ~50 KB of memory per workgroup (halo, partial sums) instead of 128 MB.
//
// Bit-exact with the launch-per-phase loops (unlike psp_coop.hip, whose reductions are ordered differently): a thread
// owns the same two elements the vector kernels' thread owns (2t, 2t + 1 of a 512-element span), forms the same
// per-thread sums, a wave the same shuffle tree, a span the same ((w0 + w1) + w2) + w3, and every workgroup adds the
// spans' partial sums in the canonical order R of psp_internal.h -- for p.q in the order of csr_spmv_w4's workgroups
// (its XCD-stripe remap of blockIdx), for r.r / r.z in span order.  Same operands, same operations, same order: the same
// bits at every truncation point (tests/test_gpu_mid.py).
//
// Reference loop: pysparse/itsolvers/src/pcg.c:91-166.
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "psp_internal.h"

namespace psp {

namespace {

constexpr int kMidMaxWg = 256;
constexpr int kMidMaxRows = 1 << 21;         // 256 workgroups x 8192 rows (constant-coefficient PCG; everything else: 4096)
constexpr int kMidSpan = 512;                // rows per partial sum (kVecSpan; csr_spmv_w4: 4 waves x 128 rows)
constexpr int kMidMaxSpans = kMidMaxRows / kMidSpan;
constexpr int kMidMaxLds = 150 * 1024;       // bytes of dynamic LDS the kernel may ask for (160 KB per CU)
static_assert(kMidSpan == kVecSpan, "partial sums must match the vector kernels' spans");

constexpr int kMidCounters = 16;  // arrival counters, one 128-byte line each
struct MidCtl {
  // Arrivals, monotone over the whole solve, spread over kMidCounters words in lines of their own: workgroup w adds to
  // word w % 16, and the polling wave reads all of them with one load per lane.  With ONE word the 256 arrivals of a barrier
  // queued up behind each other (and behind 256 polls of the same word) at one memory channel: 4 us per barrier
  // (in-kernel stamps, profiles/r5_mid_stamps.txt).
  unsigned count[kMidCounters * 32];
  int error;
  int info, iter;
  double relres;
  double nonstag;    // running count of (workgroup, wave) pairs whose rows did not stagnate: the scan's reduction needs no
                     // order (small integers), so it is an atomic counter and every iteration looks at its increase
};

struct MidArgs {
  int n, nwg, H;  // H: halo entries on either side of a block (even, >= the largest |offset|)
  int offs[12];
  double cval[12];  // CV kernels: the one value of every offset (constant-coefficient operators)
  const double *valT;
  const unsigned short *mask;
  const double *dinv;  // pre == 1
  double dc;           // pre == 2
  int pre;             // 0 no preconditioner, 1 jacobi (dinv array), 2 jacobi with a constant diagonal
  const double *x;
  double *xout;
  double *r;
  double n2b, tolb, normr0, rho0;
  int maxit;
  MidCtl *ctl;
  double *part;  // 4 x kMidMaxSpans: p.q | r.r | r.z | non-stagnated flags, by span
  double *hist;
  int np_w4, stripe;  // grid and XCD stripe of the launch-per-phase product: the order its p.q partials are added in
  int nspans;         // ceil(n / 512): grid of the vector kernels
  long long *stamps;  // PSP_MID_STAMPS (tuning): 8 wall-clock stamps (100 MHz) per iteration of workgroup 0, iterations 1..16
};

__device__ __forceinline__ void mcoh_store(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// two adjacent doubles (16-byte aligned) in ONE device-coherent store: the rows a workgroup publishes leave as half as many
// fabric writes (MI355X_MICROARCH.md: an 8-byte sc1 store costs 2.7 times a 16-byte one per byte)
__device__ __forceinline__ void mcoh_store2(double *p, double v0, double v1) {
  typedef double d2s __attribute__((ext_vector_type(2)));
  const d2s v = {v0, v1};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ double mcoh_load(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid barrier of psp_coop.hip (one monotone arrival counter, every wave's coherent stores drained first, spins bounded
// by the wall clock); false: some workgroup gave up -- everybody leaves
constexpr long long kMidSpinTicks = 50000000;  // 0.5 s of the 100 MHz clock
__device__ __forceinline__ bool mid_barrier(MidCtl *c, int nwg, unsigned &gen) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  gen += 1;
  if (nwg > 1 && threadIdx.x < 64) {  // wave 0: lane 0 announces the workgroup, lanes 0 .. 15 poll one counter each
    const int lane = threadIdx.x;
    if (lane == 0)
      (void)__hip_atomic_fetch_add(&c->count[((int)blockIdx.x % kMidCounters) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // counter k is complete at gen x (workgroups w with w % 16 == k)
    const unsigned mine = lane < kMidCounters ? (unsigned)((nwg - lane + kMidCounters - 1) / kMidCounters) : 0u;
    const unsigned target = gen * mine;
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
      if (lane < kMidCounters && mine)
        ok = (int)(__hip_atomic_load(&c->count[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 15u) == 0 &&
          (wall_clock64() - t0 > kMidSpinTicks || __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0) __hip_atomic_store(&c->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// csr_spmv_w4's XCD-stripe remap: workgroup b of its launch multiplies span remap(b)
__device__ __forceinline__ int w4_remap(int b, int stripe) {
  if (stripe <= 0) return b;
  const int k = b >> 3;
  return ((k / stripe) * 8 + (b & 7)) * stripe + k % stripe;
}

// reduce(parts, np) of psp_internal.h for NV value arrays (value j at src + j * kMidMaxSpans), by every workgroup for
// itself (all of them get the same bits):
//   np <= 256: R(parts); else R([R(group g of 256)]),   R(v) = wave_sum_l(v[l] + v[l + 64] + ...)
// The span-ordered partial sums are first copied into LDS by LDS-DMA (global_load_lds_dwordx4, device-coherent: sc1) --
// ONE round trip for all of them and no registers, while the whole solver state is live; with ordinary loads (sequential,
// for want of registers) the kernel spent most of its iteration waiting for them (profiles/r5_mid_ab_*.txt).  Entry i of
// the sequence that is added is stage[map(i)], or 0.0 where map(i) >= limit:
//   stripe < 0   map(i) = i                      (r.r, r.z: span order)
//   stripe >= 0  map(i) = w4_remap(i, stripe)    (p.q: the order of csr_spmv_w4's workgroups; those of its padded grid
//                                                 that multiply nothing add 0.0)
// stage: NV * pitch doubles (pitch = limit rounded up to 128); grp_lds: NV * 16 + NV doubles.
typedef __attribute__((address_space(3))) void *lds_void_ptr;
typedef const __attribute__((address_space(1))) void *global_void_ptr;

__device__ __forceinline__ double mid_wave_reduce_lds(const double *v, int first, int count, int stripe, int limit) {
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int i = lane; i < count; i += 64) {
    const int b = stripe >= 0 ? w4_remap(first + i, stripe) : first + i;
    s += b < limit ? v[b] : 0.0;
  }
  return psp_wave_sum(s);
}

template <int NV, int BLK>
__device__ __forceinline__ void mid_reduce(double (&out)[NV], const double *src, int np, int stripe, int limit,
                                           double *stage, double *grp_lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pitch = (limit + 127) & ~127;  // the partial-sum arrays are kMidMaxSpans long (a multiple of 128) and zero-filled
#pragma unroll
  for (int j = 0; j < NV; ++j)
    for (int base = wave * 128; base < pitch; base += (BLK / 64) * 128)  // one wave-instruction: 64 lanes x 16 bytes
      __builtin_amdgcn_global_load_lds((global_void_ptr)(src + (size_t)j * kMidMaxSpans + base + 2 * lane),
                                       (lds_void_ptr)(stage + j * pitch + base), 16, 0, 16 /* sc1: device scope */);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int ngroups = (np + kTailGroup - 1) / kTailGroup;  // <= 16 (np <= 4096)
  if (ngroups <= 1) {
    if (wave < NV) {
      const double s = mid_wave_reduce_lds(stage + wave * pitch, 0, np, stripe, limit);
      if (lane == 0) grp_lds[NV * 16 + wave] = s;
    }
  } else {
    for (int task = wave; task < NV * ngroups; task += BLK / 64) {
      const int j = task / ngroups, g = task % ngroups;
      const int cnt = min(kTailGroup, np - g * kTailGroup);
      const double s = mid_wave_reduce_lds(stage + j * pitch, g * kTailGroup, cnt, stripe, limit);
      if (lane == 0) grp_lds[j * 16 + g] = s;
    }
    __syncthreads();
    if (wave < NV) {
      const double s = mid_wave_reduce_lds(grp_lds + wave * 16, 0, ngroups, -1, ngroups);
      if (lane == 0) grp_lds[NV * 16 + wave] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) out[j] = grp_lds[NV * 16 + j];
  __syncthreads();
}

// The residual of the rows on either side of the block -- [base - H, base) and [base + B, base + B + H), what the
// neighbours published at the last barrier -- copied into LDS (hst[0, H) and hst[H, 2H)) by LDS-DMA, a pair of rows per
// lane (rows outside the matrix skipped).  Asynchronous: the caller waits (s_waitcnt vmcnt(0)) before it reads.
// Issued together with the partial sums' copies, the halo costs no round trip of its own; read with ordinary loads in a
// loop it cost three (no registers to keep them in flight).
template <int BLK>
__device__ __forceinline__ void mid_halo_dma(const double *r, long base, int B, int H, int n, double *hst, int zone1 = -1) {
  // zone1 < 0: the second zone follows the first (hst[H ..)); else it starts at hst[zone1] (the window's upper halo)
  const int z1 = zone1 < 0 ? H : zone1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int zone = 0; zone < 2; ++zone) {
    const long row0 = zone == 0 ? base - H : base + B;  // even (base is a multiple of 2048, H is even): 16-byte aligned pairs
    for (int e0 = wave * 128; e0 < H; e0 += (BLK / 64) * 128) {  // one wave-instruction: 64 lanes x one pair of rows
      const int e = e0 + 2 * lane;
      const long g = row0 + e;
      if (e < H && g >= 0 && g + 1 < n)
        __builtin_amdgcn_global_load_lds((global_void_ptr)(r + g), (lds_void_ptr)(hst + zone * z1 + e0), 16, 0, 16);
      else if (e < H && g >= 0 && g < n)  // the matrix' last row when n is odd: one lane in the whole grid
        hst[zone * z1 + e] = mcoh_load(r + g);
    }
  }
}

// pcg.c:128-139's stagnation test for one row WITHOUT its division.  The reference forms d = |alpha p / x| (1.0 if x == 0 != p),
// takes the maximum over the rows and asks whether 1 + dmax == 1; a row contributes to "not stagnated" iff its own
// 1 + d != 1, i.e. (round to nearest even) iff d > 2^-53.  With a = |fl(alpha p)|, b = |x| and t = 2^-53 b (exact for
// b >= 2^-960): fl(a / b) > 2^-53  <=>  a / b > 2^-53 (1 + 2^-53)  <=>  a >= the double after t  <=>  a > t.  NaNs compare false on
// both sides as the reference's "ddum > dmax" does, an infinite quotient is "moves" on both; tiny |x| takes the division.
__device__ __forceinline__ bool mid_row_moves(double ap, double pv, double xv) {
  if (xv == 0.0) return pv != 0.0;
  const double a = fabs(ap), b = fabs(xv);
  if (b < 0x1p-960) {
    const double q = a / b;
    return q == q && 1.0 + q != 1.0;
  }
  return a > 0x1p-53 * b;
}

// grid barrier, then the reduction by every workgroup for itself; false: some workgroup gave up -- everybody leaves
template <int NV, int BLK>
__device__ __forceinline__ bool mid_barrier_reduce(MidCtl *c, int nwg, unsigned &gen, double (&out)[NV], const double *src,
                                                   int np, int stripe, int limit, double *stage, double *grp_lds) {
  if (!mid_barrier(c, nwg, gen)) return false;
  mid_reduce<NV, BLK>(out, src, np, stripe, limit, stage, grp_lds);
  return true;
}

// pcg.c:91-166 from the head of iteration 1 (r = b - A x, rho = r.z, ||r|| > tolb are the caller's)
// CV: constant coefficients (one value per offset) -- NO scalars instead of NO registers per row
template <int NO, int LAYERS, int BLK, bool CV = false>
__global__ __launch_bounds__(BLK) void pcg_mid_kernel(MidArgs a) {
  extern __shared__ double lds[];
  constexpr int kMidLayer = 2 * BLK;  // rows one layer of a workgroup covers: a pair of rows per thread
  constexpr int kMidBlock = BLK;
  constexpr int B = LAYERS * kMidLayer;
  constexpr int NW = kMidBlock / 64;  // 16 waves (8 with 512 threads)
  const int H = a.H;
  // Two layers: x and q of the own rows live in LDS as well (each is touched once per iteration; 16 registers less --
  // with them the kernel spilled 24 registers per lane for the 5-point operator).  One layer: registers.
  constexpr bool XQ_LDS = B == 4096;
  double *win = lds;                  // p at rows [base - H, base + B + H)
  double *xl = lds + (2 * H + B);     // XQ_LDS: x of the own rows, then q
  double *ql = xl + (XQ_LDS ? B : 0);
  double *red = ql + (XQ_LDS ? B : 0);  // 3 x (LAYERS * NW) wave sums
  double *grp = red + 3 * LAYERS * NW;  // mid_reduce's scratch: 3 * 16 + 3
  // the partial sums, staged for the reduction: two arrays of nspans (rounded up to 128) -- behind the scratch, or, for
  // blocks of 8192 rows (up to 4096 spans: 64 KB), in the window's own part, which is dead at both barriers when the own p
  // lives in registers (the halo zones, which carry p_old of the neighbours' rows across iterations, are not touched)
  constexpr bool STAGE_IN_WIN = B > 4096;
  static_assert(!STAGE_IN_WIN || !XQ_LDS, "staging in the window needs the own p in registers");
  double *stage = STAGE_IN_WIN ? win + H : grp + 3 * 16 + 4;
  // the neighbours' residual, staged for the p update of the halo: q's place (dead between the r update and the next
  // product) with two layers, a region of its own with one
  double *hst = XQ_LDS ? ql : (STAGE_IN_WIN ? grp + 3 * 16 + 4 : stage + 2 * ((a.nspans + 127) & ~127));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wg = blockIdx.x, nwg = a.nwg, n = a.n;
  const long base = (long)wg * B;
  const int pre = a.pre;
  const double dc = a.dc;
  // ---- the thread's rows: everything that does not change stays in registers for the whole solve
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 v[CV ? 1 : LAYERS][NO];
  double cv[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) cv[o] = a.cval[o];
  unsigned m0[LAYERS], m1[LAYERS];
  double xr[LAYERS][2], rr[LAYERS][2], pr[LAYERS][2], qr[LAYERS][2];
  bool in0[LAYERS], in1[LAYERS];
#pragma unroll
  for (int L = 0; L < LAYERS; ++L) {
    const long row = base + (long)L * kMidLayer + 2 * t;
    in0[L] = row < n;
    in1[L] = row + 1 < n;
    m0[L] = m1[L] = 0;
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if constexpr (!CV) v[L][o] = d2{0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 2; ++u) xr[L][u] = rr[L][u] = pr[L][u] = qr[L][u] = 0.0;
    if (in0[L]) {
      const unsigned mm = *reinterpret_cast<const unsigned *>(a.mask + row);  // (padded to a whole block)
      m0[L] = mm & 0xffffu;
      m1[L] = mm >> 16;
      const double *vp = a.valT + (size_t)(row / 128) * NO * 128 + (size_t)(row % 128);
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if constexpr (!CV) v[L][o] = *reinterpret_cast<const d2 *>(vp + o * 128);
      xr[L][0] = a.x[row];
      rr[L][0] = a.r[row];
      if (in1[L]) {
        xr[L][1] = a.x[row + 1];
        rr[L][1] = a.r[row + 1];
      }
    }
    if constexpr (XQ_LDS) {
      xl[L * kMidLayer + 2 * t] = xr[L][0];
      xl[L * kMidLayer + 2 * t + 1] = xr[L][1];
    }
  }
  for (int i = t; i < 2 * H + B; i += kMidBlock) win[i] = 0.0;
  for (int i = t; i < 2 * H; i += kMidBlock) hst[i] = 0.0;  // (rows outside the matrix are never copied)
  __syncthreads();
  mid_halo_dma<BLK>(a.r, base, B, H, n, hst);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned gen = 0;
  double nonstag_seen = 0.0;
  double rho = a.rho0, rho1 = 1.0, normr = a.normr0, alpha = 0.0, beta = 0.0;
  int flag = -1, it;
#define PSP_MID_STAMP(K)                                                                     \
  if (a.stamps && wg == (int)(a.stamps[0]) && t == 0 && it <= 16) a.stamps[8 + (it - 1) * 8 + (K)] = wall_clock64()
  for (it = 1; it <= a.maxit; ++it) {
    PSP_MID_STAMP(0);
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
    // ---- p = z (+ beta p): own rows from registers into the window (pcg.c:93-97, :106, :113-114; px_update_kernel's
    // expressions), halo entries from the neighbours' published r and the window's own previous p
#pragma unroll
    for (int L = 0; L < LAYERS; ++L) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        double z = rr[L][u];
        if (pre == 1 && (u == 0 ? in0[L] : in1[L])) z = z * a.dinv[base + L * kMidLayer + 2 * t + u];  // (cached: constant data)
        if (pre == 2) z = z * dc;
        // (two layers: the own p lives in the window only -- the same thread wrote it there last iteration)
        if (it > 1) z = z + beta * (XQ_LDS ? win[H + L * kMidLayer + 2 * t + u] : pr[L][u]);
        pr[L][u] = z;
      }
      if (in0[L]) {
        win[H + L * kMidLayer + 2 * t] = pr[L][0];
        win[H + L * kMidLayer + 2 * t + 1] = in1[L] ? pr[L][1] : 0.0;
      }
    }
    for (int h = t; h < 2 * H; h += kMidBlock) {
      const int wi = h < H ? h : B + h;
      const long g = base - H + wi;
      if (g >= 0 && g < n) {
        double z = hst[h];
        if (pre == 1) z = z * a.dinv[g];
        if (pre == 2) z = z * dc;
        if (it > 1) z = z + beta * win[wi];
        win[wi] = z;
      }
    }
    __syncthreads();
    PSP_MID_STAMP(1);
    // ---- q = A p for the own rows (csr_spmv_w4: products added in offset order where the row stores an entry), p.q
#pragma unroll
    for (int L = 0; L < LAYERS; ++L) {
      double a0 = 0.0, a1 = 0.0;
      const int c0 = H + L * kMidLayer + 2 * t;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const double p0 = win[c0 + a.offs[o]], p1 = win[c0 + a.offs[o] + 1];
        const double t0 = a0 + (CV ? cv[o] : v[CV ? 0 : L][o].x) * p0;
        const double t1 = a1 + (CV ? cv[o] : v[CV ? 0 : L][o].y) * p1;
        a0 = ((m0[L] >> o) & 1u) ? t0 : a0;
        a1 = ((m1[L] >> o) & 1u) ? t1 : a1;
      }
      if constexpr (XQ_LDS) {
        ql[L * kMidLayer + 2 * t] = a0;
        ql[L * kMidLayer + 2 * t + 1] = a1;
      } else {
        qr[L][0] = a0;
        qr[L][1] = a1;
      }
      double dsum = 0.0;
      if (in0[L]) {
        dsum += (XQ_LDS ? win[c0] : pr[L][0]) * a0;
        if (in1[L]) dsum += (XQ_LDS ? win[c0 + 1] : pr[L][1]) * a1;
      }
      dsum = psp_wave_sum(dsum);
      if (lane == 0) red[L * NW + wave] = dsum;
    }
    __syncthreads();
    if (t < B / kMidSpan) {  // span t of this workgroup: its four waves in order
      const int gs = wg * (B / kMidSpan) + t;
      if (gs < a.nspans) mcoh_store(a.part + gs, red[4 * t] + red[4 * t + 1] + red[4 * t + 2] + red[4 * t + 3]);
    }
    double s1[1];
    PSP_MID_STAMP(2);
    if (!mid_barrier(a.ctl, nwg, gen)) return;
    PSP_MID_STAMP(3);
    mid_reduce<1, BLK>(s1, a.part, a.np_w4, a.stripe > 0 ? a.stripe : 0, a.nspans, stage, grp);
    PSP_MID_STAMP(4);
    const double pq = s1[0];
    if (pq == 0.0) {  // pcg.c:118-120
      flag = -6;
      break;
    }
    alpha = rho / pq;
    const int stag0 = alpha == 0.0;  // pcg.c:124-125
    // ---- r -= alpha q; r.r, r.z -- then the stagnation scan and x += alpha p (x_update_kernel / r_update_kernel's
    // expressions).  The residual comes first: what the neighbours and the reduction need -- the block-boundary rows of r and
    // the spans' partial sums -- is on its way to memory while the scan and the x update, which nobody else reads, still run.
    const bool upd = alpha != 0.0;
    const double malpha = -alpha;
#pragma unroll
    for (int L = 0; L < LAYERS; ++L) {
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 0 ? in0[L] : in1[L]) {
          const int li = L * kMidLayer + 2 * t + u;
          const double qv = XQ_LDS ? ql[li] : qr[L][u];
          const double tt = upd ? rr[L][u] + malpha * qv : rr[L][u];
          rr[L][u] = tt;
          acc0 += tt * tt;
          if (pre != 0) {
            const double z = tt * (pre == 1 ? a.dinv[base + li] : dc);
            acc1 += tt * z;
          }
        }
      }
      // the rows the neighbours' halos cover: published for the next iteration's p
      const int lr = L * kMidLayer + 2 * t;  // row inside the block
      if (in0[L] && (lr < H || lr + 2 > B - H)) {
        const long row = base + lr;
        if (in1[L]) mcoh_store2(a.r + row, rr[L][0], rr[L][1]);
        else mcoh_store(a.r + row, rr[L][0]);
      }
      acc0 = psp_wave_sum(acc0);
      acc1 = pre == 0 ? acc0 : psp_wave_sum(acc1);  // no preconditioner: z is r, the same sum
      if (lane == 0) {
        red[L * NW + wave] = acc0;
        red[LAYERS * NW + L * NW + wave] = acc1;
      }
    }
    __syncthreads();
    if (t < 2 * (B / kMidSpan)) {
      const int j = t / (B / kMidSpan), s = t % (B / kMidSpan);
      const int gs = wg * (B / kMidSpan) + s;
      const double *rj = red + j * LAYERS * NW;
      if (gs < a.nspans)
        mcoh_store(a.part + (size_t)(1 + j) * kMidMaxSpans + gs, rj[4 * s] + rj[4 * s + 1] + rj[4 * s + 2] + rj[4 * s + 3]);
    }
#pragma unroll
    for (int L = 0; L < LAYERS; ++L) {
      bool moves = false;  // some row of this lane has 1 + |alpha p / x| != 1 (pcg.c:128-139)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 0 ? in0[L] : in1[L]) {
          const int li = L * kMidLayer + 2 * t + u;
          double xv = XQ_LDS ? xl[li] : xr[L][u];
          const double pv = XQ_LDS ? win[H + li] : pr[L][u];
          const double ap = alpha * pv;
          moves = moves || mid_row_moves(ap, pv, xv);
          if (upd) xv = xv + ap;
          if constexpr (XQ_LDS) xl[li] = xv;
          else xr[L][u] = xv;
        }
      }
      const bool wave_moves = __ballot(moves) != 0ull;  // (the launch-per-phase loops: 1 + max over the wave != 1 -- the same)
      if (lane == 0) red[2 * LAYERS * NW + L * NW + wave] = wave_moves ? 1.0 : 0.0;
    }
    __syncthreads();
    if (t == 64) {  // the workgroup's non-stagnated waves (exact small integers: any order)
      double f = 0.0;
      for (int w = 0; w < LAYERS * NW; ++w) f += red[2 * LAYERS * NW + w];
      if (f != 0.0) (void)__hip_atomic_fetch_add(&a.ctl->nonstag, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double s3[3];
    {
      double s2[2];
      PSP_MID_STAMP(5);
      if (!mid_barrier(a.ctl, nwg, gen)) return;
      PSP_MID_STAMP(6);
      mid_halo_dma<BLK>(a.r, base, B, H, n, hst);  // the next iteration's halo rides with the partial sums: one round trip
      mid_reduce<2, BLK>(s2, a.part + kMidMaxSpans, a.nspans, -1, a.nspans, stage, grp);
      PSP_MID_STAMP(7);
      s3[0] = s2[0];
      s3[1] = s2[1];
      const double seen = mcoh_load(&a.ctl->nonstag);  // complete: every workgroup added its share before it arrived
      s3[2] = seen - nonstag_seen;
      nonstag_seen = seen;
    }
    normr = sqrt(s3[0]);  // the recurred residual (pcg.c:146-153)
    if (a.hist && wg == 0 && t == 0) a.hist[it] = normr;
    if (normr <= a.tolb) {  // pcg.c:154-157
      flag = 0;
      break;
    }
    if (stag0 || s3[2] == 0.0) {  // pcg.c:159-162
      flag = -5;
      break;
    }
    rho1 = rho;
    rho = s3[1];
  }
#pragma unroll
  for (int L = 0; L < LAYERS; ++L) {
    const long row = base + (long)L * kMidLayer + 2 * t;
    if (in0[L]) a.xout[row] = XQ_LDS ? xl[L * kMidLayer + 2 * t] : xr[L][0];
    if (in1[L]) a.xout[row + 1] = XQ_LDS ? xl[L * kMidLayer + 2 * t + 1] : xr[L][1];
  }
  if (wg == 0 && t == 0) {
    a.ctl->info = flag;
    a.ctl->iter = it;  // maxit + 1 when the loop ran out (pcg.c:165)
    a.ctl->relres = normr / a.n2b;
  }
}

// minres.c:96-193 from the head of iteration 1 (v_hat = b - A x, y = K v_hat, beta = sqrt(v_hat . y), the first loop test and
// w = w_old = v_hat_old = 0 are the caller's) -- the same structure as pcg_mid_kernel.  What crosses workgroups is the
// unnormalised Lanczos vector y (= K v_hat), published at the barrier of the v_hat . y reduction; every reader divides the
// entries it gathers by beta itself (the owner's own correctly rounded division), so v = y / beta is never exchanged.  The
// window holds nothing between iterations here, so the partial sums are staged in its own part and the neighbours' y is
// copied straight into its halo zones.  Expressions: csr_spmv_w4's scaled product, lanczos_kernel, minres_wx_kernel
// (psp_vec.hip), minres_scalar_alpha / minres_scalar_beta (psp_solvers.hip) -- the launch-per-phase loop's bits.
struct MidMinresArgs {
  int n, nwg, H;
  int offs[12];
  double cval[12];
  const double *valT;
  const unsigned short *mask;
  const double *dinv;
  double dc;
  int pre;
  const double *x;
  double *xout;
  const double *v_hat;
  double *yv;  // in: K v_hat (v_hat itself without a preconditioner); the kernel publishes its block-boundary rows here
  double norm_r0, beta0, errtol;
  int it_max;
  MidCtl *ctl;
  double *part;  // 2 x kMidMaxSpans: v . Av | v_hat . y, by span
  double *hist;
  int np_w4, stripe, nspans;
};

template <int NO, int LAYERS, int BLK, bool CV = false>
__global__ __launch_bounds__(BLK) void minres_mid_kernel(MidMinresArgs a) {
  extern __shared__ double lds[];
  constexpr int kMidLayer = 2 * BLK;
  constexpr int kMidBlock = BLK;
  constexpr int B = LAYERS * kMidLayer;
  constexpr int NW = kMidBlock / 64;
  constexpr bool X_LDS = B == 4096;  // blocks of 4096 rows: x, w, w_old of the own rows live in LDS
  const int H = a.H;
  double *win = lds;                   // v = y / beta at rows [base - H, base + B + H); between iterations: staging
  double *xl = lds + (2 * H + B);
  double *wl = xl + (X_LDS ? B : 0);
  double *wol = wl + (X_LDS ? B : 0);
  double *red = wol + (X_LDS ? B : 0);  // LAYERS * NW wave sums
  double *grp = red + LAYERS * NW;      // mid_reduce's scratch: 16 + 1
  double *stage = win + H;              // the partial sums, staged in the window's own part (dead at both barriers)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wg = blockIdx.x, nwg = a.nwg, n = a.n;
  const long base = (long)wg * B;
  const int pre = a.pre;
  const double dc = a.dc;
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 v[CV ? 1 : LAYERS][NO];
  double cv[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) cv[o] = a.cval[o];
  unsigned m0[LAYERS], m1[LAYERS];
  double xr[LAYERS][2], wr[LAYERS][2], wo[LAYERS][2], vh[LAYERS][2], vho[LAYERS][2], yr[LAYERS][2], avr[LAYERS][2];
  bool in0[LAYERS], in1[LAYERS];
#pragma unroll
  for (int L = 0; L < LAYERS; ++L) {
    const long row = base + (long)L * kMidLayer + 2 * t;
    in0[L] = row < n;
    in1[L] = row + 1 < n;
    m0[L] = m1[L] = 0;
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if constexpr (!CV) v[L][o] = d2{0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 2; ++u) xr[L][u] = wr[L][u] = wo[L][u] = vh[L][u] = vho[L][u] = yr[L][u] = avr[L][u] = 0.0;
    if (in0[L]) {
      const unsigned mm = *reinterpret_cast<const unsigned *>(a.mask + row);
      m0[L] = mm & 0xffffu;
      m1[L] = mm >> 16;
      const double *vp = a.valT + (size_t)(row / 128) * NO * 128 + (size_t)(row % 128);
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if constexpr (!CV) v[L][o] = *reinterpret_cast<const d2 *>(vp + o * 128);
      xr[L][0] = a.x[row];
      vh[L][0] = a.v_hat[row];
      yr[L][0] = a.yv[row];
      if (in1[L]) {
        xr[L][1] = a.x[row + 1];
        vh[L][1] = a.v_hat[row + 1];
        yr[L][1] = a.yv[row + 1];
      }
    }
    if constexpr (X_LDS) {
      const int li = L * kMidLayer + 2 * t;
      xl[li] = xr[L][0];
      xl[li + 1] = xr[L][1];
      wl[li] = wl[li + 1] = 0.0;
      wol[li] = wol[li + 1] = 0.0;
    }
  }
  for (int i = t; i < 2 * H + B; i += kMidBlock) win[i] = 0.0;
  __syncthreads();
  // the neighbours' y into the window's halo zones (rows outside the matrix stay 0)
  mid_halo_dma<BLK>(a.yv, base, B, H, n, win, H + B);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned gen = 0;
  double beta = a.beta0, beta_old = 1.0, c = 1.0, c_old = 1.0, sn = 0.0, s_old = 0.0, eta = a.beta0, norm_rmr = a.norm_r0;
  int it = 1, info = 1;  // 1: left by the loop test (0 / -1 decided there)
  double relres = 0.0;
  for (;;) {
    // ---- v = y / beta (minres.c:123-124): own rows and halo, into the window
#pragma unroll
    for (int L = 0; L < LAYERS; ++L)
      if (in0[L]) {
        win[H + L * kMidLayer + 2 * t] = yr[L][0] / beta;
        win[H + L * kMidLayer + 2 * t + 1] = in1[L] ? yr[L][1] / beta : 0.0;
      }
    for (int h = t; h < 2 * H; h += kMidBlock) {
      const int wi = h < H ? h : B + h;
      const long g = base - H + wi;
      if (g >= 0 && g < n) win[wi] = win[wi] / beta;
    }
    __syncthreads();
    // ---- Av = A v (:127-129; csr_spmv_w4's order), alpha = v . Av
#pragma unroll
    for (int L = 0; L < LAYERS; ++L) {
      double a0 = 0.0, a1 = 0.0;
      const int c0 = H + L * kMidLayer + 2 * t;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const double p0 = win[c0 + a.offs[o]], p1 = win[c0 + a.offs[o] + 1];
        const double t0 = a0 + (CV ? cv[o] : v[CV ? 0 : L][o].x) * p0;
        const double t1 = a1 + (CV ? cv[o] : v[CV ? 0 : L][o].y) * p1;
        a0 = ((m0[L] >> o) & 1u) ? t0 : a0;
        a1 = ((m1[L] >> o) & 1u) ? t1 : a1;
      }
      avr[L][0] = a0;
      avr[L][1] = a1;
      double dsum = 0.0;
      if (in0[L]) {
        dsum += win[c0] * a0;
        if (in1[L]) dsum += win[c0 + 1] * a1;
      }
      dsum = psp_wave_sum(dsum);
      if (lane == 0) red[L * NW + wave] = dsum;
    }
    __syncthreads();
    if (t < B / kMidSpan) {
      const int gs = wg * (B / kMidSpan) + t;
      if (gs < a.nspans) mcoh_store(a.part + gs, red[4 * t] + red[4 * t + 1] + red[4 * t + 2] + red[4 * t + 3]);
    }
    double s1[1];
    if (!mid_barrier_reduce<1, BLK>(a.ctl, nwg, gen, s1, a.part, a.np_w4, a.stripe > 0 ? a.stripe : 0, a.nspans, stage, grp)) return;
    const double alpha = s1[0];
    const double c1 = alpha / beta, c2 = beta / beta_old;  // :131
    // ---- Lanczos update (:131-143; lanczos_kernel's expressions), beta^2 = v_hat . y; y published for the neighbours
    double yold[LAYERS][2];
#pragma unroll
    for (int L = 0; L < LAYERS; ++L) {
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        yold[L][u] = yr[L][u];
        if (u == 0 ? in0[L] : in1[L]) {
          const double nv = avr[L][u] - c1 * vh[L][u] - c2 * vho[L][u];
          vho[L][u] = vh[L][u];
          vh[L][u] = nv;
          if (pre != 0) {
            const double yy = nv * (pre == 1 ? a.dinv[base + L * kMidLayer + 2 * t + u] : dc);
            yr[L][u] = yy;
            acc += nv * yy;
          } else {
            yr[L][u] = nv;
            acc += nv * nv;
          }
        }
      }
      acc = psp_wave_sum(acc);
      if (lane == 0) red[L * NW + wave] = acc;
      const int lr = L * kMidLayer + 2 * t;
      if (in0[L] && (lr < H || lr + 2 > B - H)) {
        const long row = base + lr;
        if (in1[L]) mcoh_store2(a.yv + row, yr[L][0], yr[L][1]);
        else mcoh_store(a.yv + row, yr[L][0]);
      }
    }
    __syncthreads();
    if (t < B / kMidSpan) {
      const int gs = wg * (B / kMidSpan) + t;
      if (gs < a.nspans)
        mcoh_store(a.part + kMidMaxSpans + gs, red[4 * t] + red[4 * t + 1] + red[4 * t + 2] + red[4 * t + 3]);
    }
    if (!mid_barrier(a.ctl, nwg, gen)) return;
    mid_halo_dma<BLK>(a.yv, base, B, H, n, win, H + B);  // the next iteration's halo rides with the partial sums
    mid_reduce<1, BLK>(s1, a.part + kMidMaxSpans, a.nspans, -1, a.nspans, stage, grp);
    // ---- minres_scalar_beta: :143-164, :180, :192
    const double beta_start = beta;  // beta at the start of this iteration
    beta_old = beta;
    double b2 = s1[0];
    if (b2 < 0.0) {  // :144-146
      info = -3;
      break;
    }
    beta = sqrt(b2);
    const double c_oold = c_old, s_oold = s_old;
    c_old = c;
    s_old = sn;
    const double r1_hat = c_old * alpha - c_oold * s_old * beta_old;
    const double r1 = sqrt(r1_hat * r1_hat + beta * beta);
    const double r2 = s_old * alpha + c_oold * c_old * beta_old;
    const double r3 = s_oold * beta_old;
    if (r1 == 0.0) {  // :160-162
      info = -6;
      break;
    }
    c = r1_hat / r1;
    sn = beta / r1;
    const double c_eta = c * eta;
    eta = -sn * eta;
    norm_rmr = norm_rmr * fabs(sn);
    if (a.hist && wg == 0 && t == 0) a.hist[it] = norm_rmr;
    // ---- w / x update (:172-180; minres_wx_kernel's expressions, v = y_old / beta formed again)
#pragma unroll
    for (int L = 0; L < LAYERS; ++L) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 0 ? in0[L] : in1[L]) {
          const int li = L * kMidLayer + 2 * t + u;
          const double vv = yold[L][u] / beta_start;
          const double ww = X_LDS ? wl[li] : wr[L][u];
          const double wov = X_LDS ? wol[li] : wo[L][u];
          double xv = X_LDS ? xl[li] : xr[L][u];
          const double nw = (vv - r3 * wov - r2 * ww) / r1;
          xv += c_eta * nw;
          if constexpr (X_LDS) {
            wol[li] = ww;
            wl[li] = nw;
            xl[li] = xv;
          } else {
            wo[L][u] = ww;
            wr[L][u] = nw;
            xr[L][u] = xv;
          }
        }
      }
    }
    // ---- the loop test at the head of the next iteration (:114, strict <)
    const bool conv = norm_rmr < a.errtol * a.norm_r0;
    if (it >= a.it_max || conv) {
      info = conv ? 0 : -1;
      relres = norm_rmr / a.norm_r0;
      break;
    }
    it += 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a halo copy may still be in flight)
#pragma unroll
  for (int L = 0; L < LAYERS; ++L) {
    const long row = base + (long)L * kMidLayer + 2 * t;
    if (in0[L]) a.xout[row] = X_LDS ? xl[L * kMidLayer + 2 * t] : xr[L][0];
    if (in1[L]) a.xout[row + 1] = X_LDS ? xl[L * kMidLayer + 2 * t + 1] : xr[L][1];
  }
  if (wg == 0 && t == 0) {
    a.ctl->info = info;
    a.ctl->iter = it;
    a.ctl->relres = relres;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3-D grid operators (7 offsets: -nx ny, -nx, -1, 0, 1, nx, nx ny) -- the same loop with the rows dealt out in BRICKS.
// A contiguous block of rows of a 3-D grid is a slab whose halo is a whole grid plane on either side (80^3: 6400 rows for a
// block of 2048): the window does not fit the LDS and forming p there costs six times the block's own work -- the kernels
// above refuse such operators (80^3 ran 33.5 us per iteration against 29.9 launch-per-phase).  Here a workgroup owns a
// brick of bx x by x bz <= 4096 grid points (eight per thread, 512 threads): its halo is the brick's surface (16^3: 1536
// cells), p of the brick and its halo lives in one LDS array addressed by (a + 1) + (bx + 2)((b + 1) + (by + 2)(c + 1)),
// x and q of the own points in LDS, r and the points' seven matrix entries in registers.  What crosses workgroups is r
// of a brick's surface points (published with device-coherent stores at the barrier of the r.r / r.z reduction; the
// neighbours form their halo entries of p from it, as above) and one partial sum per workgroup and reduction.
// Arithmetic: per element the reference's operations (pcg.c:91-166), a row's products added in ascending column order
// (csr_mat.c:49-54); a reduction adds a thread's points in a fixed order, the wave tree, the workgroup's waves in order,
// the workgroups' sums by R -- fixed for a given grid, so runs are bitwise reproducible; the order differs from the
// launch-per-phase loops' (rows of a span lie in several bricks), so iterates agree with theirs to rounding, not bit for
// bit -- as psp_coop.hip's do (tests/test_gpu_brick.py: the oracle's counts and iterates within 1e-12).
constexpr int kBrickBlock = 512;
constexpr int kBrickPPT = 8;   // points per thread: bricks of <= 4096 points
constexpr int kBrickHPT = 5;   // halo cells per thread: <= 2560 (a brick of 4096 points has at most 2 (bx by + by bz + bx bz) of them)

struct BrickArgs {
  int n, nwg;
  int nx, ny, nz;  // the grid
  int bx, by, bz;  // points of a brick along each axis
  int cx, cy;      // bricks along x and y (workgroup w owns brick (w % cx, (w / cx) % cy, w / (cx cy)))
  double cval[7];  // CV kernels: the one value of every offset
  const double *valT;
  const unsigned short *mask;
  const double *dinv;
  double dc;
  int pre;
  const double *x;
  double *xout;
  double *r;
  double n2b, tolb, normr0, rho0;
  int maxit;
  MidCtl *ctl;
  double *part;  // 3 x kMidMaxWg: p.q | r.r | r.z by workgroup
  double *hist;
};

// R over the workgroups' partial sums of NV values (value j at part + j * kMidMaxWg), by every workgroup for itself
template <int NV>
__device__ __forceinline__ void brick_reduce(double (&out)[NV], const double *part, int nwg, double *sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < NV) {
    double t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = lane + 64 * k < nwg ? mcoh_load(part + wave * kMidMaxWg + lane + 64 * k) : 0.0;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += t[k];
    s = psp_wave_sum(s);
    if (lane == 0) sh[wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) out[j] = sh[j];
  __syncthreads();
}

// CV: constant coefficients -- seven scalars instead of seven registers per grid point (56 of 256): no scratch memory
template <bool CV>
__global__ __launch_bounds__(kBrickBlock) void pcg_brick_kernel(BrickArgs a) {
  extern __shared__ double lds[];
  constexpr int NW = kBrickBlock / 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wg = blockIdx.x, nwg = a.nwg;
  const int bx = a.bx, by = a.by, bz = a.bz, nx = a.nx, ny = a.ny, nz = a.nz;
  const int x0 = (wg % a.cx) * bx, y0 = ((wg / a.cx) % a.cy) * by, z0 = (wg / (a.cx * a.cy)) * bz;
  const int ex = min(bx, nx - x0), ey = min(by, ny - y0), ez = min(bz, nz - z0);  // (the last bricks may be cut)
  const int sx = bx + 2, sxy = sx * (by + 2);
  const int vol = bx * by * bz, npad = sxy * (bz + 2);
  double *P = lds;                         // p of the brick and its halo
  double *xl = P + ((npad + 1) & ~1);      // x of the own points, by point number t + 512 m
  double *ql = xl + kBrickBlock * kBrickPPT;  // q
  double *rl = ql + kBrickBlock * kBrickPPT;  // r (in LDS like x and q: with it in registers the kernel spilled)
  double *red = rl + kBrickBlock * kBrickPPT;  // 3 x NW wave sums, then 4 for brick_reduce
  const int pre = a.pre;
  const double dc = a.dc;
  const int loff[7] = {-sxy, -sx, -1, 0, 1, sx, sxy};
  // ---- the thread's points
  double v[CV ? 1 : kBrickPPT][7];
  double cv[7];
#pragma unroll
  for (int o = 0; o < 7; ++o) cv[o] = a.cval[o];
  unsigned long long mk = 0;  // 7 mask bits per point
  double rg[CV ? kBrickPPT : 1], qg[CV ? kBrickPPT : 1];  // constant coefficients: r and q in the registers the matrix left
  int li[kBrickPPT], row[kBrickPPT];
  unsigned inmask = 0, surf = 0;
#pragma unroll
  for (int m = 0; m < kBrickPPT; ++m) {
    const int l = t + kBrickBlock * m;
    const int pa = l % bx, pb = (l / bx) % by, pc = l / (bx * by);
    const bool in = l < vol && pa < ex && pb < ey && pc < ez;
    li[m] = (pa + 1) + sx * (pb + 1) + sxy * (pc + 1);
    row[m] = in ? (x0 + pa) + nx * ((y0 + pb) + ny * (z0 + pc)) : 0;
    double xv = 0.0, rv = 0.0;
    if constexpr (!CV) {
#pragma unroll
      for (int o = 0; o < 7; ++o) v[m][o] = 0.0;
    }
    if (in) {
      inmask |= 1u << m;
      if (pa == 0 || pa == ex - 1 || pb == 0 || pb == ey - 1 || pc == 0 || pc == ez - 1) surf |= 1u << m;
      const int rw = row[m];
      mk |= (unsigned long long)(a.mask[rw] & 0x7fu) << (7 * m);
      if constexpr (!CV) {
        const double *vp = a.valT + (size_t)(rw / 128) * 7 * 128 + (size_t)(rw % 128);
#pragma unroll
        for (int o = 0; o < 7; ++o) v[m][o] = vp[o * 128];
      }
      xv = a.x[rw];
      rv = a.r[rw];
    }
    xl[l] = xv;
    if constexpr (CV) rg[m] = rv;
    else rl[l] = rv;
  }
  // ---- the thread's halo cells: LDS index and grid row (-1: outside the grid, or beyond a cut brick's faces)
  int hidx[kBrickHPT], hrow[kBrickHPT];
  {
    const int f0 = bx * by, f1 = bx * bz, f2 = by * bz;
#pragma unroll
    for (int j = 0; j < kBrickHPT; ++j) {
      int h = t + kBrickBlock * j;
      int pa = 0, pb = 0, pc = 0;
      bool ok = true;
      if (h < 2 * f0) {
        pc = h < f0 ? -1 : ez;
        h = h < f0 ? h : h - f0;
        pa = h % bx;
        pb = h / bx;
        ok = pa < ex && pb < ey;
      } else if (h < 2 * f0 + 2 * f1) {
        h -= 2 * f0;
        pb = h < f1 ? -1 : ey;
        h = h < f1 ? h : h - f1;
        pa = h % bx;
        pc = h / bx;
        ok = pa < ex && pc < ez;
      } else if (h < 2 * f0 + 2 * f1 + 2 * f2) {
        h -= 2 * f0 + 2 * f1;
        pa = h < f2 ? -1 : ex;
        h = h < f2 ? h : h - f2;
        pb = h % by;
        pc = h / by;
        ok = pb < ey && pc < ez;
      } else {
        ok = false;
      }
      const int gx = x0 + pa, gy = y0 + pb, gz = z0 + pc;
      ok = ok && gx >= 0 && gx < nx && gy >= 0 && gy < ny && gz >= 0 && gz < nz;
      hidx[j] = ok ? (pa + 1) + sx * (pb + 1) + sxy * (pc + 1) : -1;
      hrow[j] = ok ? gx + nx * (gy + ny * gz) : 0;
    }
  }
  for (int i = t; i < npad; i += kBrickBlock) P[i] = 0.0;
  double rh[kBrickHPT];  // r of the halo cells, fetched after the barrier that published it
#pragma unroll
  for (int j = 0; j < kBrickHPT; ++j) rh[j] = hidx[j] >= 0 ? a.r[hrow[j]] : 0.0;
  __syncthreads();
  unsigned gen = 0;
  double nonstag_seen = 0.0;
  double rho = a.rho0, rho1 = 1.0, normr = a.normr0, alpha = 0.0, beta = 0.0;
  int flag = -1, it;
  for (it = 1; it <= a.maxit; ++it) {
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
    // ---- p = z (+ beta p): the own points and the halo cells (pcg.c:93-97, :106, :113-114)
#pragma unroll
    for (int m = 0; m < kBrickPPT; ++m)
      if ((inmask >> m) & 1u) {
        double z = CV ? rg[CV ? m : 0] : rl[t + kBrickBlock * m];
        if (pre == 1) z = z * a.dinv[row[m]];
        if (pre == 2) z = z * dc;
        if (it > 1) z = z + beta * P[li[m]];
        P[li[m]] = z;
      }
#pragma unroll
    for (int j = 0; j < kBrickHPT; ++j)
      if (hidx[j] >= 0) {
        double z = rh[j];
        if (pre == 1) z = z * a.dinv[hrow[j]];
        if (pre == 2) z = z * dc;
        if (it > 1) z = z + beta * P[hidx[j]];
        P[hidx[j]] = z;
      }
    __syncthreads();
    // ---- q = A p for the own points (ascending column order), p.q
    {
      double dsum = 0.0;
#pragma unroll
      for (int m = 0; m < kBrickPPT; ++m) {
        double acc = 0.0;
        if ((inmask >> m) & 1u) {
#pragma unroll
          for (int o = 0; o < 7; ++o) {
            const double tt = acc + (CV ? cv[o] : v[CV ? 0 : m][o]) * P[li[m] + loff[o]];
            acc = ((mk >> (7 * m + o)) & 1ull) ? tt : acc;
          }
          dsum += P[li[m]] * acc;
        }
        if constexpr (CV) qg[m] = acc;
        else ql[t + kBrickBlock * m] = acc;
      }
      dsum = psp_wave_sum(dsum);
      if (lane == 0) red[wave] = dsum;
    }
    __syncthreads();
    if (t == 0) {
      double s = red[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += red[w];
      mcoh_store(a.part + wg, s);
    }
    if (!mid_barrier(a.ctl, nwg, gen)) return;
    double s1[1];
    brick_reduce<1>(s1, a.part, nwg, red + 3 * NW);
    const double pq = s1[0];
    if (pq == 0.0) {  // pcg.c:118-120
      flag = -6;
      break;
    }
    alpha = rho / pq;
    const int stag0 = alpha == 0.0;  // pcg.c:124-125
    const bool upd = alpha != 0.0;
    const double malpha = -alpha;
    // ---- r -= alpha q; r.r, r.z; the surface points' r published; then the stagnation scan and x += alpha p
    {
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int m = 0; m < kBrickPPT; ++m)
        if ((inmask >> m) & 1u) {
          const double qv = CV ? qg[CV ? m : 0] : ql[t + kBrickBlock * m];
          const double ro = CV ? rg[CV ? m : 0] : rl[t + kBrickBlock * m];
          const double tt = upd ? ro + malpha * qv : ro;
          if constexpr (CV) rg[m] = tt;
          else rl[t + kBrickBlock * m] = tt;
          acc0 += tt * tt;
          if (pre != 0) {
            const double z = tt * (pre == 1 ? a.dinv[row[m]] : dc);
            acc1 += tt * z;
          }
          if ((surf >> m) & 1u) mcoh_store(a.r + row[m], tt);
        }
      acc0 = psp_wave_sum(acc0);
      acc1 = pre == 0 ? acc0 : psp_wave_sum(acc1);
      if (lane == 0) {
        red[wave] = acc0;
        red[NW + wave] = acc1;
      }
    }
    __syncthreads();
    if (t < 2) {
      const double *rj = red + t * NW;
      double s = rj[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += rj[w];
      mcoh_store(a.part + (size_t)(1 + t) * kMidMaxWg + wg, s);
    }
    {
      bool moves = false;
#pragma unroll
      for (int m = 0; m < kBrickPPT; ++m)
        if ((inmask >> m) & 1u) {
          double xv = xl[t + kBrickBlock * m];
          const double pv = P[li[m]];
          const double ap = alpha * pv;
          moves = moves || mid_row_moves(ap, pv, xv);
          if (upd) xv = xv + ap;
          xl[t + kBrickBlock * m] = xv;
        }
      const bool wave_moves = __ballot(moves) != 0ull;
      if (lane == 0) red[2 * NW + wave] = wave_moves ? 1.0 : 0.0;
    }
    __syncthreads();
    if (t == 64) {
      double f = 0.0;
      for (int w = 0; w < NW; ++w) f += red[2 * NW + w];
      if (f != 0.0) (void)__hip_atomic_fetch_add(&a.ctl->nonstag, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!mid_barrier(a.ctl, nwg, gen)) return;
#pragma unroll
    for (int j = 0; j < kBrickHPT; ++j) rh[j] = hidx[j] >= 0 ? mcoh_load(a.r + hrow[j]) : 0.0;  // the next iteration's halo
    double s2[2];
    brick_reduce<2>(s2, a.part + kMidMaxWg, nwg, red + 3 * NW);
    const double seen = mcoh_load(&a.ctl->nonstag);
    const double nonstag = seen - nonstag_seen;
    nonstag_seen = seen;
    normr = sqrt(s2[0]);  // the recurred residual (pcg.c:146-153)
    if (a.hist && wg == 0 && t == 0) a.hist[it] = normr;
    if (normr <= a.tolb) {  // pcg.c:154-157
      flag = 0;
      break;
    }
    if (stag0 || nonstag == 0.0) {  // pcg.c:159-162
      flag = -5;
      break;
    }
    rho1 = rho;
    rho = s2[1];
  }
#pragma unroll
  for (int m = 0; m < kBrickPPT; ++m)
    if ((inmask >> m) & 1u) a.xout[row[m]] = xl[t + kBrickBlock * m];
  if (wg == 0 && t == 0) {
    a.ctl->info = flag;
    a.ctl->iter = it;
    a.ctl->relres = normr / a.n2b;
  }
}

// minres.c:96-193 in bricks (as minres_mid_kernel is to pcg_mid_kernel): what crosses workgroups is the unnormalised
// Lanczos vector y = K v_hat of the bricks' surface points; every reader divides what it gathers by beta itself.
// P holds v = y / beta of the brick and its halo; x, w, w_old of the own points live in LDS, v_hat, v_hat_old and y in
// registers.
struct BrickMinresArgs {
  int n, nwg;
  int nx, ny, nz, bx, by, bz, cx, cy;
  double cval[7];  // CV kernels: the one value of every offset
  const double *valT;
  const unsigned short *mask;
  const double *dinv;
  double dc;
  int pre;
  const double *x;
  double *xout;
  const double *v_hat;
  double *yv;  // in: K v_hat (v_hat itself without a preconditioner); the kernel publishes the surface points' rows here
  double norm_r0, beta0, errtol;
  int it_max;
  MidCtl *ctl;
  double *part;  // 2 x kMidMaxWg: v . Av | v_hat . y by workgroup
  double *hist;
};

template <bool CV>
__global__ __launch_bounds__(kBrickBlock) void minres_brick_kernel(BrickMinresArgs a) {
  extern __shared__ double lds[];
  constexpr int NW = kBrickBlock / 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wg = blockIdx.x, nwg = a.nwg;
  const int bx = a.bx, by = a.by, bz = a.bz, nx = a.nx, ny = a.ny, nz = a.nz;
  const int x0 = (wg % a.cx) * bx, y0 = ((wg / a.cx) % a.cy) * by, z0 = (wg / (a.cx * a.cy)) * bz;
  const int ex = min(bx, nx - x0), ey = min(by, ny - y0), ez = min(bz, nz - z0);
  const int sx = bx + 2, sxy = sx * (by + 2);
  const int vol = bx * by * bz, npad = sxy * (bz + 2);
  double *P = lds;
  double *vol_ = P + ((npad + 1) & ~1);           // v_hat_old of the own points, by point number t + 512 m
  double *wl = vol_ + kBrickBlock * kBrickPPT;    // w
  double *wol = wl + kBrickBlock * kBrickPPT;     // w_old
  double *red = wol + kBrickBlock * kBrickPPT;    // NW wave sums, then 4 for brick_reduce
  // x of the own points stays in memory (xout: touched once per iteration, by the thread that owns the point); y = K v_hat
  // is formed again from v_hat where it is needed (the same rounded product) -- with v_hat_old, y and x in registers too
  // the kernel spilled 330 bytes per lane
  const int pre = a.pre;
  const double dc = a.dc;
  const int loff[7] = {-sxy, -sx, -1, 0, 1, sx, sxy};
  double v[CV ? 1 : kBrickPPT][7], vh[kBrickPPT];
  double xr[CV ? kBrickPPT : 1];  // constant coefficients: the registers the matrix left hold x (else it stays in memory)
  double cv[7];
#pragma unroll
  for (int o = 0; o < 7; ++o) cv[o] = a.cval[o];
  unsigned long long mk = 0;
  int li[kBrickPPT], row[kBrickPPT];
  unsigned inmask = 0, surf = 0;
#pragma unroll
  for (int m = 0; m < kBrickPPT; ++m) {
    const int l = t + kBrickBlock * m;
    const int pa = l % bx, pb = (l / bx) % by, pc = l / (bx * by);
    const bool in = l < vol && pa < ex && pb < ey && pc < ez;
    li[m] = (pa + 1) + sx * (pb + 1) + sxy * (pc + 1);
    row[m] = in ? (x0 + pa) + nx * ((y0 + pb) + ny * (z0 + pc)) : 0;
    vh[m] = 0.0;
    if constexpr (!CV) {
#pragma unroll
      for (int o = 0; o < 7; ++o) v[m][o] = 0.0;
    }
    if (in) {
      inmask |= 1u << m;
      if (pa == 0 || pa == ex - 1 || pb == 0 || pb == ey - 1 || pc == 0 || pc == ez - 1) surf |= 1u << m;
      const int rw = row[m];
      mk |= (unsigned long long)(a.mask[rw] & 0x7fu) << (7 * m);
      if constexpr (!CV) {
        const double *vp = a.valT + (size_t)(rw / 128) * 7 * 128 + (size_t)(rw % 128);
#pragma unroll
        for (int o = 0; o < 7; ++o) v[m][o] = vp[o * 128];
      }
      if constexpr (CV) xr[m] = a.x[rw];
      else a.xout[rw] = a.x[rw];
      vh[m] = a.v_hat[rw];
    }
    vol_[l] = 0.0;
    wl[l] = 0.0;
    wol[l] = 0.0;
  }
  int hidx[kBrickHPT], hrow[kBrickHPT];
  {
    const int f0 = bx * by, f1 = bx * bz, f2 = by * bz;
#pragma unroll
    for (int j = 0; j < kBrickHPT; ++j) {
      int h = t + kBrickBlock * j;
      int pa = 0, pb = 0, pc = 0;
      bool ok = true;
      if (h < 2 * f0) {
        pc = h < f0 ? -1 : ez;
        h = h < f0 ? h : h - f0;
        pa = h % bx;
        pb = h / bx;
        ok = pa < ex && pb < ey;
      } else if (h < 2 * f0 + 2 * f1) {
        h -= 2 * f0;
        pb = h < f1 ? -1 : ey;
        h = h < f1 ? h : h - f1;
        pa = h % bx;
        pc = h / bx;
        ok = pa < ex && pc < ez;
      } else if (h < 2 * f0 + 2 * f1 + 2 * f2) {
        h -= 2 * f0 + 2 * f1;
        pa = h < f2 ? -1 : ex;
        h = h < f2 ? h : h - f2;
        pb = h % by;
        pc = h / by;
        ok = pb < ey && pc < ez;
      } else {
        ok = false;
      }
      const int gx = x0 + pa, gy = y0 + pb, gz = z0 + pc;
      ok = ok && gx >= 0 && gx < nx && gy >= 0 && gy < ny && gz >= 0 && gz < nz;
      hidx[j] = ok ? (pa + 1) + sx * (pb + 1) + sxy * (pc + 1) : -1;
      hrow[j] = ok ? gx + nx * (gy + ny * gz) : 0;
    }
  }
  for (int i = t; i < npad; i += kBrickBlock) P[i] = 0.0;
  double yh[kBrickHPT];  // y of the halo cells
#pragma unroll
  for (int j = 0; j < kBrickHPT; ++j) yh[j] = hidx[j] >= 0 ? a.yv[hrow[j]] : 0.0;
  __syncthreads();
  unsigned gen = 0;
  double beta = a.beta0, beta_old = 1.0, c = 1.0, c_old = 1.0, sn = 0.0, s_old = 0.0, eta = a.beta0, norm_rmr = a.norm_r0;
  int it = 1, info = 1;
  double relres = 0.0;
  for (;;) {
    // ---- v = y / beta (minres.c:123-124): the own points and the halo cells
#pragma unroll
    for (int m = 0; m < kBrickPPT; ++m)
      if ((inmask >> m) & 1u) {
        double yy = vh[m];  // y = K v_hat (lanczos_kernel's expression)
        if (pre == 1) yy = yy * a.dinv[row[m]];
        if (pre == 2) yy = yy * dc;
        P[li[m]] = yy / beta;
      }
#pragma unroll
    for (int j = 0; j < kBrickHPT; ++j)
      if (hidx[j] >= 0) P[hidx[j]] = yh[j] / beta;
    __syncthreads();
    // ---- Av = A v (:127-129), alpha = v . Av
    double avr[kBrickPPT];
    {
      double dsum = 0.0;
#pragma unroll
      for (int m = 0; m < kBrickPPT; ++m) {
        double acc = 0.0;
        if ((inmask >> m) & 1u) {
#pragma unroll
          for (int o = 0; o < 7; ++o) {
            const double tt = acc + (CV ? cv[o] : v[CV ? 0 : m][o]) * P[li[m] + loff[o]];
            acc = ((mk >> (7 * m + o)) & 1ull) ? tt : acc;
          }
          dsum += P[li[m]] * acc;
        }
        avr[m] = acc;
      }
      dsum = psp_wave_sum(dsum);
      if (lane == 0) red[wave] = dsum;
    }
    __syncthreads();
    if (t == 0) {
      double s = red[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += red[w];
      mcoh_store(a.part + wg, s);
    }
    if (!mid_barrier(a.ctl, nwg, gen)) return;
    double s1[1];
    brick_reduce<1>(s1, a.part, nwg, red + NW);
    const double alpha = s1[0];
    const double c1 = alpha / beta, c2 = beta / beta_old;  // :131
    // ---- Lanczos update (:131-143), beta^2 = v_hat . y; the surface points' y published
    {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < kBrickPPT; ++m)
        if ((inmask >> m) & 1u) {
          const double nv = avr[m] - c1 * vh[m] - c2 * vol_[t + kBrickBlock * m];
          vol_[t + kBrickBlock * m] = vh[m];
          vh[m] = nv;
          double yy = nv;
          if (pre != 0) {
            yy = nv * (pre == 1 ? a.dinv[row[m]] : dc);
            acc += nv * yy;
          } else {
            acc += nv * nv;
          }
          if ((surf >> m) & 1u) mcoh_store(a.yv + row[m], yy);
        }
      acc = psp_wave_sum(acc);
      if (lane == 0) red[wave] = acc;
    }
    __syncthreads();
    if (t == 0) {
      double s = red[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += red[w];
      mcoh_store(a.part + kMidMaxWg + wg, s);
    }
    if (!mid_barrier(a.ctl, nwg, gen)) return;
#pragma unroll
    for (int j = 0; j < kBrickHPT; ++j) yh[j] = hidx[j] >= 0 ? mcoh_load(a.yv + hrow[j]) : 0.0;  // the next iteration's halo
    brick_reduce<1>(s1, a.part + kMidMaxWg, nwg, red + NW);
    // ---- minres_scalar_beta: :143-164, :180, :192
    const double beta_start = beta;
    beta_old = beta;
    const double b2 = s1[0];
    if (b2 < 0.0) {  // :144-146
      info = -3;
      break;
    }
    beta = sqrt(b2);
    const double c_oold = c_old, s_oold = s_old;
    c_old = c;
    s_old = sn;
    const double r1_hat = c_old * alpha - c_oold * s_old * beta_old;
    const double r1 = sqrt(r1_hat * r1_hat + beta * beta);
    const double r2 = s_old * alpha + c_oold * c_old * beta_old;
    const double r3 = s_oold * beta_old;
    if (r1 == 0.0) {  // :160-162
      info = -6;
      break;
    }
    c = r1_hat / r1;
    sn = beta / r1;
    const double c_eta = c * eta;
    eta = -sn * eta;
    norm_rmr = norm_rmr * fabs(sn);
    if (a.hist && wg == 0 && t == 0) a.hist[it] = norm_rmr;
    (void)beta_start;
    // ---- w / x update (:172-180); v of this iteration is still in P
#pragma unroll
    for (int m = 0; m < kBrickPPT; ++m)
      if ((inmask >> m) & 1u) {
        const int l = t + kBrickBlock * m;
        const double vv = P[li[m]];
        const double ww = wl[l], wov = wol[l];
        const double nw = (vv - r3 * wov - r2 * ww) / r1;
        wol[l] = ww;
        wl[l] = nw;
        if constexpr (CV) xr[m] = xr[m] + c_eta * nw;
        else a.xout[row[m]] = a.xout[row[m]] + c_eta * nw;
      }
    const bool conv = norm_rmr < a.errtol * a.norm_r0;
    if (it >= a.it_max || conv) {
      info = conv ? 0 : -1;
      relres = norm_rmr / a.norm_r0;
      break;
    }
    it += 1;
  }
  if constexpr (CV) {
#pragma unroll
    for (int m = 0; m < kBrickPPT; ++m)
      if ((inmask >> m) & 1u) a.xout[row[m]] = xr[m];
  }
  if (wg == 0 && t == 0) {
    a.ctl->info = info;
    a.ctl->iter = it;
    a.ctl->relres = relres;
  }
}

std::atomic<long long> g_mid_solves{0}, g_mid_fallbacks{0}, g_brick_solves{0}, g_brick_fallbacks{0};

bool mid_enabled() {
  static const bool on = [] {
    const char *e = tuning_env("PSP_MID");
    return !(e && atoi(e) == 0);
  }();
  return on;
}

// PSP_MID_MIN (tuning switch): from how many rows on.  Default 2^16 (MINRES), 2^15 (PCG): against psp_coop.hip's one-row-per-thread loops (which
// remain for general matrices of <= 8 entries per row and for everything smaller) the kernels here take 9.7 / 11.7 us per
// PCG iteration at 256^2, 9.9 / 13.1 at 300^2, 12.5 / 21.4 at 512^2; MINRES 9.2 / 10.4 at 300^2, 10.9 / 18.8 at 512^2;
// below ~200^2 the order flips (100^2: 10.1 / 9.2) (profiles/r5_mid_vs_coop.txt)
// (PCG from 2^15: 200^2 9.8 / 10.6; MINRES's one-row-per-thread loop holds out longer: 200^2 9.2 / 8.6)
int mid_min_rows(bool minres) {
  const char *e = tuning_env("PSP_MID_MIN");
  return e ? atoi(e) : (minres ? 1 << 16 : 1 << 15);
}

// kernel of (offsets, rows per workgroup, threads per workgroup); nullptr: not built
#define PSP_MID_TABLE(FN, KERNEL)                                                      \
  const void *FN(int no, int rows, int blk) {                                          \
    const int key = rows == 2048 ? (blk == 1024 ? 0 : 1) : (blk == 1024 ? 2 : 3);      \
    switch (no * 4 + key) {                                                            \
      PSP_MID_ROW(KERNEL, 1) PSP_MID_ROW(KERNEL, 2) PSP_MID_ROW(KERNEL, 3) PSP_MID_ROW(KERNEL, 4) \
      PSP_MID_ROW(KERNEL, 5) PSP_MID_ROW(KERNEL, 6) PSP_MID_ROW(KERNEL, 7)             \
      PSP_MID_ROW(KERNEL, 8) PSP_MID_ROW(KERNEL, 9)                                    \
      default:                                                                         \
        return nullptr;                                                                \
    }                                                                                  \
  }
#define PSP_MID_ROW(KERNEL, NO)                             \
  case NO * 4 + 0:                                          \
    return (const void *)KERNEL<NO, 1, 1024>;               \
  case NO * 4 + 1:                                          \
    return (const void *)KERNEL<NO, 2, 512>;                \
  case NO * 4 + 2:                                          \
    return (const void *)KERNEL<NO, 2, 1024>;               \
  case NO * 4 + 3:                                          \
    return (const void *)KERNEL<NO, 4, 512>;
PSP_MID_TABLE(mid_kernel, pcg_mid_kernel)
PSP_MID_TABLE(mid_minres_kernel, minres_mid_kernel)
#undef PSP_MID_ROW
#undef PSP_MID_TABLE
// the constant-coefficient forms: 3 / 5 / 7 / 9 offsets (the Poisson-like operators)
#define PSP_MID_CV_ROW(KERNEL, NO)                          \
  case NO * 4 + 0:                                          \
    return (const void *)KERNEL<NO, 1, 1024, true>;         \
  case NO * 4 + 1:                                          \
    return (const void *)KERNEL<NO, 2, 512, true>;          \
  case NO * 4 + 2:                                          \
    return (const void *)KERNEL<NO, 2, 1024, true>;         \
  case NO * 4 + 3:                                          \
    return (const void *)KERNEL<NO, 4, 512, true>;
#define PSP_MID_CV_TABLE(FN, KERNEL)                                                   \
  const void *FN(int no, int rows, int blk) {                                          \
    const int key = rows == 2048 ? (blk == 1024 ? 0 : 1) : (blk == 1024 ? 2 : 3);      \
    switch (no * 4 + key) {                                                            \
      PSP_MID_CV_ROW(KERNEL, 3) PSP_MID_CV_ROW(KERNEL, 5) PSP_MID_CV_ROW(KERNEL, 7) PSP_MID_CV_ROW(KERNEL, 9) \
      default:                                                                         \
        return nullptr;                                                                \
    }                                                                                  \
  }
PSP_MID_CV_TABLE(mid_kernel_cv, pcg_mid_kernel)
// blocks of 8192 rows (2^20 < n <= 2^21): PCG, constant coefficients, 512 threads with eight row pairs each
const void *mid_kernel_cv8(int no) {
  switch (no) {
    case 3: return (const void *)pcg_mid_kernel<3, 8, 512, true>;
    case 5: return (const void *)pcg_mid_kernel<5, 8, 512, true>;
    case 7: return (const void *)pcg_mid_kernel<7, 8, 512, true>;
    case 9: return (const void *)pcg_mid_kernel<9, 8, 512, true>;
    default: return nullptr;
  }
}
PSP_MID_CV_TABLE(mid_minres_kernel_cv, minres_mid_kernel)
#undef PSP_MID_CV_ROW
#undef PSP_MID_CV_TABLE

struct MidPlan {
  W4View w4;
  int rows, blk, nwg, H;  // rows per workgroup (2048 / 4096), threads per workgroup (1024 / 512)
  size_t lds;
  const void *kernel;
};

// threads per workgroup: 512 threads may keep 256 registers each -- the blocks of 4096 rows, whose 1024-thread kernels
// spill (132 .. 250 bytes per lane with 5 offsets), run without scratch memory.  PSP_MID_BLK (tuning switch, read per solve)
int mid_block_threads(int rows, int no, bool cv) {
  if (const char *e = tuning_env("PSP_MID_BLK")) {
    const int v = atoi(e);
    if (v == 512 || v == 1024) return v;
  }
  // measured (profiles/r5_mid_blk_ab.txt): blocks of 4096 rows 24.0 -> 21.7 us (PCG), 29.3 -> 19.0 us (MINRES) per iteration at
  // 1024^2 with 512 threads; blocks of 2048 rows the same either way with 5 offsets, 512 ahead with 7 (40 x 40 x 300:
  // MINRES 19.5 -> 15.9 us), whose 1024-thread kernels spill
  // constant-coefficient forms (no matrix registers, no scratch either way): 1024 threads -- 1024^2 PCG 21.7 -> 19.6 us,
  // MINRES 18.7 -> 18.1, 40 x 40 x 300 PCG 18.6 -> 17.6 (profiles/r5_mid_cv_ab.txt)
  if (cv) return 1024;
  return (rows == 4096 || no >= 6) ? 512 : 1024;
}

// the grid must be resident at once: one workgroup per CU with this much LDS
bool mid_capacity_ok(const MidPlan *P) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, int> cap;  // (device, kernel) -> workgroups the device holds at once
  std::lock_guard<std::mutex> lk(mu);
  if (const char *e = tuning_env("PSP_COOP_CAPACITY")) return P->nwg <= atoi(e);
  const auto key = std::make_pair(current_device(), P->kernel);
  auto it = cap.find(key);
  if (it == cap.end()) {
    int c = 0, per = 0;
    Workspace *w = nullptr;
    if (hipFuncSetAttribute(P->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMidMaxLds) == hipSuccess &&
        workspace(&w) == PSP_OK && w->num_cu > 0 &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, P->kernel, P->blk, kMidMaxLds) == hipSuccess)
      c = per * w->num_cu;
    else
      (void)hipGetLastError();
    it = cap.emplace(key, c).first;
  }
  return P->nwg <= it->second;
}

// the plan for this operator, or false: no index-free layout of <= 9 offsets, too many rows, a halo that does not fit
// the LDS, or a grid the device cannot hold at once
bool mid_plan(const psp_csr *A, int n, MidPlan *P, bool minres = false) {
  if (!mid_enabled() || !A || A->nrows != n || A->ncols != n || n < 1024 || n > kMidMaxRows) return false;
  int av = 0;
  if (csr_w4_view(A, &P->w4, &av) != PSP_OK || !av || P->w4.no > 9) return false;
  // below mid_min_rows psp_coop.hip's one-row-per-thread loops are faster -- where they apply: rows of 9 entries are
  // beyond them at every size (launch-per-phase loops: 22-25 us per iteration at 400^2 against 11-13 here)
  if (n < mid_min_rows(minres) && !(P->w4.no > 8 && !tuning_env("PSP_MID_MIN"))) return false;
  int omax = 1;
  for (int i = 0; i < P->w4.no; ++i) omax = std::max(omax, std::abs(P->w4.offs[i]));
  P->H = (omax + 2) & ~1;  // even, and one pair beyond the farthest entry (a row pair reads offset + 1)
  if (P->H > 2048) return false;  // the halo is staged in LDS and updated in <= 4 passes: 2-D grids up to 2046 wide, slim 3-D ones
  P->rows = n > kMidMaxWg * 4096 ? 8192 : (n > kMidMaxWg * 2048 ? 4096 : 2048);
  if (P->rows == 8192) {
    // 2^20 < n <= 2^21: constant-coefficient PCG only (no matrix registers: eight row pairs per thread fit 512 threads; the
    // partial sums are staged in the window) -- measured against the launch-per-phase loops in profiles/r5_mid_8192_ab.txt
    if (minres || !P->w4.constv || tuning_env("PSP_MID_NOCV") || !mid_kernel_cv8(P->w4.no)) return false;
    P->blk = 512;
    P->nwg = (n + 8191) / 8192;
    P->lds = sizeof(double) * (size_t)(4 * P->H + 8192 + 3 * (8192 / 128) + 3 * 16 + 8);
    if (P->nwg > kMidMaxWg || P->lds > (size_t)kMidMaxLds) return false;
    if (P->w4.grid > 4096 || (n + kMidSpan - 1) / kMidSpan > kMidMaxSpans) return false;
    P->kernel = mid_kernel_cv8(P->w4.no);
    return mid_capacity_ok(P);
  }
  const bool cv = P->w4.constv && !tuning_env("PSP_MID_NOCV") &&
                  (minres ? mid_minres_kernel_cv(P->w4.no, P->rows, 1024) : mid_kernel_cv(P->w4.no, P->rows, 1024)) != nullptr;
  P->blk = mid_block_threads(P->rows, P->w4.no, cv);
  if (P->rows == 4096 && P->blk == 1024 && P->w4.no > 5 && !cv) return false;  // register budget of two row pairs per thread
  const int B = P->rows;
  P->nwg = (n + B - 1) / B;
  const int nspans = (n + kMidSpan - 1) / kMidSpan;
  if (minres)  // window (also the staging area) + x, w, w_old for blocks of 4096 rows + the wave sums and mid_reduce's scratch
    P->lds = sizeof(double) * (size_t)(2 * P->H + B + (B == 4096 ? 3 * B : 0) + B / 128 + 18);
  else
    P->lds = sizeof(double) * (size_t)(2 * P->H + B + (B == 4096 ? 2 * B : 0) + 3 * (B / 128) + 3 * 16 + 4 +
                                       2 * ((nspans + 127) & ~127) + (B == 4096 ? 0 : 2 * P->H));
  if (P->nwg > kMidMaxWg || P->lds > (size_t)kMidMaxLds) return false;
  if (P->w4.grid > 4096 || (n + kMidSpan - 1) / kMidSpan > kMidMaxSpans) return false;
  P->kernel = nullptr;
  if (P->w4.constv && !tuning_env("PSP_MID_NOCV"))  // (PSP_MID_NOCV: tuning switch, A/B of the constant-coefficient forms)
    P->kernel = minres ? mid_minres_kernel_cv(P->w4.no, P->rows, P->blk) : mid_kernel_cv(P->w4.no, P->rows, P->blk);
  if (!P->kernel) P->kernel = minres ? mid_minres_kernel(P->w4.no, P->rows, P->blk) : mid_kernel(P->w4.no, P->rows, P->blk);
  if (!P->kernel) return false;
  return mid_capacity_ok(P);
}

}  // namespace

bool mid_applicable(const psp_csr *A, int n, const double *dinv) {
  (void)dinv;
  MidPlan P;
  return mid_plan(A, n, &P);
}

bool mid_minres_applicable(const psp_csr *A, int n) {
  MidPlan P;
  return mid_plan(A, n, &P, true);
}

// On kCoopFallback x, v_hat and y are what they were on entry (x: staged in w, which is zeroed again; y -- whose
// block-boundary rows the kernel overwrites -- restored from the copy kept in av); as minres_coop_loop (psp_coop.hip).
int minres_mid_loop(const psp_csr *A, const double *dinv, int n, double *x, double *v_hat, double *v_hat_old, double *y,
                    double *w, double *w_old, double *v, double *av, double norm_r0, double beta0, double errtol, int it_max,
                    int *info, int *iter, double *relres, double *hist) {
  MidPlan P;
  if (!mid_plan(A, n, &P, true)) return kCoopFallback;
  (void)v_hat_old;
  (void)w_old;
  // control block and partial sums: the thread's slab (psp_internal.h); the history: the solvers' vector pool
  static_assert(sizeof(MidCtl) <= kStateBytes && 4 * (size_t)kMidMaxSpans <= kCtlPartDoubles, "state slab");
  Workspace *ws;
  PSP_TRY(workspace(&ws));
  struct Mem {
    MidCtl *ctl = nullptr;
    double *part = nullptr, *hist = nullptr;
    size_t nhist = 0;
    ~Mem() { scratch_put(hist, nhist); }
  } m;
  m.ctl = static_cast<MidCtl *>(ws->state_dev);
  m.part = ws->ctl_part;
  PSP_HIP(hipMemsetAsync(m.ctl, 0, sizeof(MidCtl), stream()));
  PSP_HIP(hipMemsetAsync(m.part, 0, sizeof(double) * 2 * kMidMaxSpans, stream()));
  if (hist) {
    m.nhist = (size_t)it_max + 2;
    PSP_TRY(scratch_get(m.nhist, &m.hist));
    PSP_HIP(hipMemsetAsync(m.hist, 0xff, sizeof(double) * m.nhist, stream()));
  }
  double *yv = y;  // the vector that crosses workgroups: K v_hat, or v_hat itself without a preconditioner
  if (!dinv) {
    yv = v;
    PSP_HIP(hipMemcpyAsync(yv, v_hat, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  } else {
    PSP_HIP(hipMemcpyAsync(av, y, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  }
  MidMinresArgs a;
  a.n = n;
  a.nwg = P.nwg;
  a.H = P.H;
  for (int i = 0; i < 12; ++i) a.offs[i] = P.w4.offs[i];
  for (int i = 0; i < 12; ++i) a.cval[i] = P.w4.constv ? P.w4.cval[i] : 0.0;
  a.valT = P.w4.valT;
  a.mask = P.w4.mask;
  a.dinv = dinv;
  a.dc = 0.0;
  a.pre = !dinv ? 0 : (dinv_constant(dinv, n, &a.dc) ? 2 : 1);
  a.x = x;
  a.xout = w;
  a.v_hat = v_hat;
  a.yv = yv;
  a.norm_r0 = norm_r0;
  a.beta0 = beta0;
  a.errtol = errtol;
  a.it_max = it_max;
  a.ctl = m.ctl;
  a.part = m.part;
  a.hist = m.hist;
  a.np_w4 = P.w4.grid;
  a.stripe = P.w4.stripe;
  a.nspans = (n + kMidSpan - 1) / kMidSpan;
  void *args[] = {&a};
  int rc = PSP_OK;
  const char *ff = tuning_env("PSP_COOP_FAIL");
  if (ff && atoi(ff) == 1) {
    rc = kCoopFallback;
  } else if (hipLaunchCooperativeKernel(P.kernel, dim3(P.nwg), dim3(P.blk), args, (unsigned)P.lds, stream()) != hipSuccess) {
    (void)hipGetLastError();
    rc = kCoopFallback;
  }
  MidCtl c;
  if (rc == PSP_OK) {
    PSP_HIP(hipMemcpyAsync(&c, m.ctl, sizeof(MidCtl), hipMemcpyDeviceToHost, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    if (c.error) rc = kCoopFallback;
  }
  if (rc == kCoopFallback) {
    g_mid_fallbacks.fetch_add(1);
    if (dinv) PSP_HIP(hipMemcpyAsync(y, av, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    PSP_HIP(hipMemsetAsync(w, 0, sizeof(double) * (size_t)n, stream()));
  }
  if (rc != PSP_OK) return rc;
  g_mid_solves.fetch_add(1);
  PSP_HIP(hipMemcpyAsync(x, w, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
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

// On kCoopFallback x and r are what they were on entry: the kernel leaves its x in a staging vector (p) that is copied
// over x only after a launch in which no workgroup gave up; r (whose block-boundary rows the kernel overwrites) is
// restored from the copy kept in q.
int pcg_mid_loop(const psp_csr *A, const double *dinv, int n, double *x, double *r, double *p, double *q, double n2b,
                 double tolb, double normr0, double rho0, int maxit, int *info, int *iter, double *relres, double *hist) {
  MidPlan P;
  if (!mid_plan(A, n, &P)) return kCoopFallback;
  // control block and partial sums: the thread's slab (psp_internal.h); the history: the solvers' vector pool
  static_assert(sizeof(MidCtl) <= kStateBytes && 4 * (size_t)kMidMaxSpans <= kCtlPartDoubles, "state slab");
  Workspace *ws;
  PSP_TRY(workspace(&ws));
  struct Mem {
    MidCtl *ctl = nullptr;
    double *part = nullptr, *hist = nullptr;
    size_t nhist = 0;
    ~Mem() { scratch_put(hist, nhist); }
  } m;
  m.ctl = static_cast<MidCtl *>(ws->state_dev);
  m.part = ws->ctl_part;
  PSP_HIP(hipMemsetAsync(m.ctl, 0, sizeof(MidCtl), stream()));
  PSP_HIP(hipMemsetAsync(m.part, 0, sizeof(double) * 4 * kMidMaxSpans, stream()));
  if (hist) {
    m.nhist = (size_t)maxit + 2;
    PSP_TRY(scratch_get(m.nhist, &m.hist));
    PSP_HIP(hipMemsetAsync(m.hist, 0xff, sizeof(double) * m.nhist, stream()));
  }
  PSP_HIP(hipMemcpyAsync(q, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  MidArgs a;
  a.n = n;
  a.nwg = P.nwg;
  a.H = P.H;
  for (int i = 0; i < 12; ++i) a.offs[i] = P.w4.offs[i];
  for (int i = 0; i < 12; ++i) a.cval[i] = P.w4.constv ? P.w4.cval[i] : 0.0;
  a.valT = P.w4.valT;
  a.mask = P.w4.mask;
  a.dinv = dinv;
  a.dc = 0.0;
  a.pre = !dinv ? 0 : (dinv_constant(dinv, n, &a.dc) ? 2 : 1);
  a.x = x;
  a.xout = p;
  a.r = r;
  a.n2b = n2b;
  a.tolb = tolb;
  a.normr0 = normr0;
  a.rho0 = rho0;
  a.maxit = maxit;
  a.ctl = m.ctl;
  a.part = m.part;
  a.hist = m.hist;
  a.np_w4 = P.w4.grid;
  a.stripe = P.w4.stripe;
  a.nspans = (n + kMidSpan - 1) / kMidSpan;
  a.stamps = nullptr;
  long long *stamps_dev = nullptr;
  const char *se = tuning_env("PSP_MID_STAMPS");  // workgroup to stamp
  if (se) {
    PSP_HIP(hipMalloc((void **)&stamps_dev, sizeof(long long) * (8 + 16 * 8)));
    PSP_HIP(hipMemsetAsync(stamps_dev, 0, sizeof(long long) * (8 + 16 * 8), stream()));
    const long long which = atoll(se);
    PSP_HIP(hipMemcpyAsync(stamps_dev, &which, sizeof(long long), hipMemcpyHostToDevice, stream()));
    a.stamps = stamps_dev;
  }
  void *args[] = {&a};
  int rc = PSP_OK;
  const char *ff = tuning_env("PSP_COOP_FAIL");
  if (ff && atoi(ff) == 1) {
    rc = kCoopFallback;
  } else if (hipLaunchCooperativeKernel(P.kernel, dim3(P.nwg), dim3(P.blk), args, (unsigned)P.lds, stream()) != hipSuccess) {
    (void)hipGetLastError();
    rc = kCoopFallback;
  }
  MidCtl c;
  if (rc == PSP_OK) {
    PSP_HIP(hipMemcpyAsync(&c, m.ctl, sizeof(MidCtl), hipMemcpyDeviceToHost, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    if (c.error) rc = kCoopFallback;
  }
  if (stamps_dev) {
    long long st[8 + 16 * 8];
    (void)hipMemcpy(st, stamps_dev, sizeof(st), hipMemcpyDeviceToHost);
    (void)hipFree(stamps_dev);
    fprintf(stderr, "[psp_mid] n %d nwg %d rows %d threads %d H %d: 10 ns ticks per phase (p update | product | barrier 1 | reduce 1 | updates | barrier 2 | reduce 2 + halo | loop end)\n",
            n, P.nwg, P.rows, P.blk, P.H);
    for (int it = 2; it < 12 && it < maxit; ++it) {
      const long long *q0 = st + 8 + it * 8, *q1 = st + 8 + (it + 1) * 8;
      fprintf(stderr, "[psp_mid]   it %2d:", it + 1);
      for (int k = 0; k < 7; ++k) fprintf(stderr, " %5lld", q0[k + 1] - q0[k]);
      fprintf(stderr, " %5lld | total %lld\n", q1[0] - q0[7], q1[0] - q0[0]);
    }
  }
  if (rc == kCoopFallback) {
    g_mid_fallbacks.fetch_add(1);
    PSP_HIP(hipMemcpyAsync(r, q, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  }
  if (rc != PSP_OK) return rc;
  g_mid_solves.fetch_add(1);
  PSP_HIP(hipMemcpyAsync(x, p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));  // x is final when the call returns, as after the other loops
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

// ---- 3-D grid operators in bricks (pcg_brick_kernel)
namespace {

struct BrickPlan {
  W4View w4;
  int nx, ny, nz, bx, by, bz, cx, cy, cz, nwg;
  size_t lds;
  const void *kernel;
};

bool brick_enabled() {
  static const bool on = [] {
    const char *e = tuning_env("PSP_BRICK");
    return !(e && atoi(e) == 0);
  }();
  return on;
}

// PSP_BRICK_MIN (tuning switch, read per solve): from how many rows on.  Measured against psp_coop.hip's one-row-per-thread
// loop (up to 2^18 rows) and the launch-per-phase loops (tools/brick_ab.py, profiles/r5_brick_ab.txt), microseconds per PCG
// iteration: 48^3 15.1 / 12.2, 56^3 15.0 / 16.1, 64^3 15.2 / 20.5, 80^3 18.2 / 28.0, 96^3 22.7 / 36.6, 100^3 24.4 / 38.0
int brick_min_rows() {
  const char *e = tuning_env("PSP_BRICK_MIN");
  return e ? atoi(e) : 150000;
}

size_t brick_lds(int bx, int by, int bz) {
  const int npad = (bx + 2) * (by + 2) * (bz + 2);
  return sizeof(double) * (size_t)(((npad + 1) & ~1) + 3 * kBrickBlock * kBrickPPT + 3 * (kBrickBlock / 64) + 8);
}

// the bricks for this operator, or false: not a 3-D grid operator, or no decomposition into <= capacity bricks of <= 4096
// points whose surfaces fit
bool brick_plan(const psp_csr *A, int n, BrickPlan *P, bool minres = false) {
  const void *kernel = nullptr;  // (set below, once the view says whether the coefficients are constant)
  if (!brick_enabled() || !mid_enabled() || !A || A->nrows != n || A->ncols != n || n < brick_min_rows() ||
      n > kMidMaxWg * kBrickBlock * kBrickPPT)
    return false;
  int av = 0;
  if (csr_w4_view(A, &P->w4, &av) != PSP_OK || !av || P->w4.no != 7 || P->w4.grid3[0] == 0) return false;
  const int nx = P->w4.grid3[0], ny = P->w4.grid3[1], nz = P->w4.grid3[2];
  if ((long)nx * ny * nz != n) return false;
  P->kernel = kernel = minres ? (P->w4.constv ? (const void *)minres_brick_kernel<true> : (const void *)minres_brick_kernel<false>)
                              : (P->w4.constv ? (const void *)pcg_brick_kernel<true> : (const void *)pcg_brick_kernel<false>);
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, int> cap;  // (device, kernel) -> workgroups the device holds at once
  int capacity;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (const char *e = tuning_env("PSP_COOP_CAPACITY")) {
      capacity = atoi(e);
    } else {
      const auto key = std::make_pair(current_device(), kernel);
      auto it = cap.find(key);
      if (it == cap.end()) {
        int c = 0, per = 0;
        Workspace *w = nullptr;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMidMaxLds) == hipSuccess &&
            workspace(&w) == PSP_OK && w->num_cu > 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, kBrickBlock, kMidMaxLds) == hipSuccess)
          c = per * w->num_cu;
        else
          (void)hipGetLastError();
        it = cap.emplace(key, c).first;
      }
      capacity = it->second;
    }
  }
  capacity = std::min(capacity, kMidMaxWg);
  // bricks: as many as the device holds, as close to cubes as the grid allows -- least work (points + halo cells) per
  // workgroup among the decompositions that fit
  long best = -1;
  for (int cz = 1; cz <= std::min(nz, capacity); ++cz)
    for (int cy = 1; cy <= std::min(ny, capacity / cz); ++cy)
      for (int cx = 1; cx <= std::min(nx, capacity / (cz * cy)); ++cx) {
        const int bx = (nx + cx - 1) / cx, by = (ny + cy - 1) / cy, bz = (nz + cz - 1) / cz;
        if ((long)(cx - 1) * bx >= nx || (long)(cy - 1) * by >= ny || (long)(cz - 1) * bz >= nz) continue;  // an empty last brick
        const long vol = (long)bx * by * bz, halo = 2L * (bx * by + by * bz + bx * bz);
        if (vol > kBrickBlock * kBrickPPT || halo > kBrickBlock * kBrickHPT || brick_lds(bx, by, bz) > (size_t)kMidMaxLds) continue;
        const long work = vol + halo;
        if (best < 0 || work < best) {
          best = work;
          P->bx = bx; P->by = by; P->bz = bz;
          P->cx = cx; P->cy = cy; P->cz = cz;
        }
      }
  if (best < 0) return false;
  P->nx = nx; P->ny = ny; P->nz = nz;
  P->nwg = P->cx * P->cy * P->cz;
  P->lds = brick_lds(P->bx, P->by, P->bz);
  return true;
}

}  // namespace

bool brick_applicable(const psp_csr *A, int n) {
  BrickPlan P;
  return brick_plan(A, n, &P);
}

bool brick_minres_applicable(const psp_csr *A, int n) {
  BrickPlan P;
  return brick_plan(A, n, &P, true);
}

// as minres_mid_loop: on kCoopFallback x, v_hat and y are what they were on entry
int minres_brick_loop(const psp_csr *A, const double *dinv, int n, double *x, double *v_hat, double *v_hat_old, double *y,
                      double *w, double *w_old, double *v, double *av, double norm_r0, double beta0, double errtol, int it_max,
                      int *info, int *iter, double *relres, double *hist) {
  BrickPlan P;
  if (!brick_plan(A, n, &P, true)) return kCoopFallback;
  (void)v_hat_old;
  (void)w_old;
  Workspace *ws;
  PSP_TRY(workspace(&ws));
  struct Mem {
    double *hist = nullptr;
    size_t nhist = 0;
    ~Mem() { scratch_put(hist, nhist); }
  } m;
  MidCtl *ctl = static_cast<MidCtl *>(ws->state_dev);
  PSP_HIP(hipMemsetAsync(ctl, 0, sizeof(MidCtl), stream()));
  PSP_HIP(hipMemsetAsync(ws->ctl_part, 0, sizeof(double) * 2 * kMidMaxWg, stream()));
  if (hist) {
    m.nhist = (size_t)it_max + 2;
    PSP_TRY(scratch_get(m.nhist, &m.hist));
    PSP_HIP(hipMemsetAsync(m.hist, 0xff, sizeof(double) * m.nhist, stream()));
  }
  double *yv = y;
  if (!dinv) {
    yv = v;
    PSP_HIP(hipMemcpyAsync(yv, v_hat, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  } else {
    PSP_HIP(hipMemcpyAsync(av, y, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  }
  BrickMinresArgs a;
  a.n = n;
  a.nwg = P.nwg;
  a.nx = P.nx; a.ny = P.ny; a.nz = P.nz;
  a.bx = P.bx; a.by = P.by; a.bz = P.bz;
  a.cx = P.cx; a.cy = P.cy;
  for (int o = 0; o < 7; ++o) a.cval[o] = P.w4.cval[o];
  a.valT = P.w4.valT;
  a.mask = P.w4.mask;
  a.dinv = dinv;
  a.dc = 0.0;
  a.pre = !dinv ? 0 : (dinv_constant(dinv, n, &a.dc) ? 2 : 1);
  a.x = x;
  a.xout = w;
  a.v_hat = v_hat;
  a.yv = yv;
  a.norm_r0 = norm_r0;
  a.beta0 = beta0;
  a.errtol = errtol;
  a.it_max = it_max;
  a.ctl = ctl;
  a.part = ws->ctl_part;
  a.hist = m.hist;
  void *args[] = {&a};
  int rc = PSP_OK;
  const char *ff = tuning_env("PSP_COOP_FAIL");
  if (ff && atoi(ff) == 1) {
    rc = kCoopFallback;
  } else if (hipLaunchCooperativeKernel(P.kernel, dim3(P.nwg), dim3(kBrickBlock), args,
                                        (unsigned)P.lds, stream()) != hipSuccess) {
    (void)hipGetLastError();
    rc = kCoopFallback;
  }
  MidCtl c;
  if (rc == PSP_OK) {
    PSP_HIP(hipMemcpyAsync(&c, ctl, sizeof(MidCtl), hipMemcpyDeviceToHost, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    if (c.error) rc = kCoopFallback;
  }
  if (rc == kCoopFallback) {
    g_brick_fallbacks.fetch_add(1);
    if (dinv) PSP_HIP(hipMemcpyAsync(y, av, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    PSP_HIP(hipMemsetAsync(w, 0, sizeof(double) * (size_t)n, stream()));
  }
  if (rc != PSP_OK) return rc;
  g_brick_solves.fetch_add(1);
  PSP_HIP(hipMemcpyAsync(x, w, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
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

int pcg_brick_loop(const psp_csr *A, const double *dinv, int n, double *x, double *r, double *p, double *q, double n2b,
                   double tolb, double normr0, double rho0, int maxit, int *info, int *iter, double *relres, double *hist) {
  BrickPlan P;
  if (!brick_plan(A, n, &P)) return kCoopFallback;
  static_assert(3 * (size_t)kMidMaxWg <= kCtlPartDoubles, "state slab");
  Workspace *ws;
  PSP_TRY(workspace(&ws));
  struct Mem {
    double *hist = nullptr;
    size_t nhist = 0;
    ~Mem() { scratch_put(hist, nhist); }
  } m;
  MidCtl *ctl = static_cast<MidCtl *>(ws->state_dev);
  PSP_HIP(hipMemsetAsync(ctl, 0, sizeof(MidCtl), stream()));
  PSP_HIP(hipMemsetAsync(ws->ctl_part, 0, sizeof(double) * 3 * kMidMaxWg, stream()));
  if (hist) {
    m.nhist = (size_t)maxit + 2;
    PSP_TRY(scratch_get(m.nhist, &m.hist));
    PSP_HIP(hipMemsetAsync(m.hist, 0xff, sizeof(double) * m.nhist, stream()));
  }
  PSP_HIP(hipMemcpyAsync(q, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));  // r as it is, for a fall-back
  BrickArgs a;
  a.n = n;
  a.nwg = P.nwg;
  a.nx = P.nx; a.ny = P.ny; a.nz = P.nz;
  a.bx = P.bx; a.by = P.by; a.bz = P.bz;
  a.cx = P.cx; a.cy = P.cy;
  for (int o = 0; o < 7; ++o) a.cval[o] = P.w4.cval[o];
  a.valT = P.w4.valT;
  a.mask = P.w4.mask;
  a.dinv = dinv;
  a.dc = 0.0;
  a.pre = !dinv ? 0 : (dinv_constant(dinv, n, &a.dc) ? 2 : 1);
  a.x = x;
  a.xout = p;
  a.r = r;
  a.n2b = n2b;
  a.tolb = tolb;
  a.normr0 = normr0;
  a.rho0 = rho0;
  a.maxit = maxit;
  a.ctl = ctl;
  a.part = ws->ctl_part;
  a.hist = m.hist;
  void *args[] = {&a};
  int rc = PSP_OK;
  const char *ff = tuning_env("PSP_COOP_FAIL");
  if (ff && atoi(ff) == 1) {
    rc = kCoopFallback;
  } else if (hipLaunchCooperativeKernel(P.kernel, dim3(P.nwg), dim3(kBrickBlock), args, (unsigned)P.lds,
                                        stream()) != hipSuccess) {
    (void)hipGetLastError();
    rc = kCoopFallback;
  }
  MidCtl c;
  if (rc == PSP_OK) {
    PSP_HIP(hipMemcpyAsync(&c, ctl, sizeof(MidCtl), hipMemcpyDeviceToHost, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    if (c.error) rc = kCoopFallback;
  }
  if (rc == kCoopFallback) {
    g_brick_fallbacks.fetch_add(1);
    PSP_HIP(hipMemcpyAsync(r, q, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  }
  if (rc != PSP_OK) return rc;
  g_brick_solves.fetch_add(1);
  PSP_HIP(hipMemcpyAsync(x, p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  *info = c.info;
  *iter = c.iter;
  *relres = c.relres;
  if (hist) {
    const int cnt = std::min(c.iter, maxit);
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

extern "C" int psp_debug_mid_count(long long *solves, long long *fallbacks) {
  if (solves) *solves = psp::g_mid_solves.load();
  if (fallbacks) *fallbacks = psp::g_mid_fallbacks.load();
  return PSP_OK;
}

extern "C" int psp_debug_brick_count(long long *solves, long long *fallbacks) {
  if (solves) *solves = psp::g_brick_solves.load();
  if (fallbacks) *fallbacks = psp::g_brick_fallbacks.load();
  return PSP_OK;
}
