// psp_csr.hip -- csr_mat / sss_mat device containers and the SpMV kernels.
//
// Reference loop: pysparse/sparse/src/csr_mat.c:49-54
//     for i: s = 0; for k in [ia[i], ia[i+1]): s += va[k]*x[ja[k]]; y[i] = s
// Every kernel in this file adds each row's separately rounded products left to right (the
// library is built with -ffp-contract=off), so y is bit-identical to that loop whichever kernel
// runs.  csr_spmv_launch picks, per matrix:
//     csr_spmv_w4 / w4x   offset-structured operators (<= 32 distinct col - row): values in
//                         offset-major blocks + row masks, no column indices, no LDS
//     sss_spmv_w4         the same for sss_mat, strict lower triangle only (read twice, shifted)
//     csr_spmv_w3         banded CSR: x blocks staged in LDS, 16-bit chunk-local columns
//     csr_spmv_w2 / w1    any CSR with short rows: one wavefront per chunk of ~1024 nonzeros,
//                         products parked in LDS, one lane per row adds them
//     csr_spmv_stream     rows longer than half a tile (workgroup-wide tiles, carried sums)
// Contents, in order:
//     chunk table builder; csr_spmv_stream; csr_spmv_wave (persistent, ablation); csr_spmv_w1;
//     csr_spmv_w2 (+ row-offset table); csr_spmv_w3 (+ build_w3_kernel);
//     w4 family: dia_offsets / dia_build kernels, poisson_w4_kernel, csr_spmv_w4, sss_spmv_w4,
//       csr_spmv_w4x, csr_spmv_w4_pf (PCG p-update folded in);
//     fold / transpose / diagonal kernels; Poisson generators;
//     host side: variant decoding, per-handle side tables (ChunkTable, CsrExtra) and their ensure_*
//       builders, the plane-sweeping schedule, csr_spmv_launch and friends, the halo-overlap split;
//     C ABI: psp_csr_*, psp_sss_*.
// HBM traffic in the CSR model: 12*nnz + 4*(n+1) + 8*n (y) + 8*n (x, once) = 12 nnz + 20 n + 4.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <new>
#include <thread>
#include <cmath>
#include <vector>

#include <chrono>

#include "psp_internal.h"

using namespace psp;

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------ chunk table

// chunk c covers rows [tab[c].x, tab[c+1].x) and nonzeros [tab[c].y, tab[c+1].y):
// tab[c].x = first row r with ind[r] >= c*target  (binary search, one thread per chunk)
__global__ void build_chunk_table(int nrows, const int *__restrict__ ind, int target, int nchunks,
                                  int2 *__restrict__ tab) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > nchunks) return;
  int r;
  if (c == nchunks) {
    r = nrows;
  } else {
    long want = (long)c * target;
    int lo = 0, hi = nrows;  // first r in [0, nrows] with ind[r] >= want
    while (lo < hi) {
      int mid = lo + ((hi - lo) >> 1);
      if ((long)ind[mid] >= want)
        hi = mid;
      else
        lo = mid + 1;
    }
    r = lo;
  }
  tab[c] = make_int2(r, ind[r]);
}

// ------------------------------------------------------------------ SpMV kernel

// native clang vectors: legal operands of the non-temporal load builtin
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));

template <bool NT, typename T>
__device__ __forceinline__ T ldg(const T *p) {
  if constexpr (NT)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}

__device__ __forceinline__ double wave_sum(double v) { return psp::psp_wave_sum(v); }

// block-wide sum of v; result valid in thread 0.  sh: 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

template <int TILE, int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void csr_spmv_stream(
    int nchunks, int map_mode, int colmask, const int2 *__restrict__ tab,
    const int *__restrict__ ind, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ dotv,
    double *__restrict__ partials) {
  static_assert(TILE % (kBlock * VEC) == 0, "tile must be a whole number of steps");
  constexpr int STEPS = TILE / (kBlock * VEC);
  __shared__ double prod[TILE];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int G = gridDim.x;
  double dsum = 0.0;

  for (int it = 0;; ++it) {
    int chunk;
    if (map_mode == 0) {
      chunk = it * G + (int)blockIdx.x;  // neighbouring chunks run at the same time chip-wide
    } else {
      // XCD-aware: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares
      // an L2), so give each XCD a contiguous stripe of G/8 chunks per sweep: the x
      // segments of neighbouring grid lines then hit in that XCD's own L2.
      const int W = G >> 3;
      chunk = (it * 8 + ((int)blockIdx.x & 7)) * W + ((int)blockIdx.x >> 3);
    }
    if (chunk >= nchunks) break;

    const int2 c0 = tab[chunk];
    const int2 c1 = tab[chunk + 1];
    const int r0 = c0.x, r1 = c1.x;
    const int s = c0.y, e = c1.y;
    double carry = 0.0;

    for (int ts = s & ~3; ts < e || ts == (s & ~3); ts += TILE) {
      const int te = ts + TILE;
      // ---- stream phase: products of this tile into LDS
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int off = (st * kBlock + tid) * VEC;
        int k = ts + off;
        k = (k < e) ? k : ts;  // past the chunk: re-read the (cached) tile head, result unused
        if constexpr (VEC == 4) {
          const i4v c = ldg<NT>(reinterpret_cast<const i4v *>(col + k));
          const d2v v0 = ldg<NT>(reinterpret_cast<const d2v *>(val + k));
          const d2v v1 = ldg<NT>(reinterpret_cast<const d2v *>(val + k + 2));
          d2v p0, p1;
          p0.x = v0.x * x[c.x & colmask];
          p0.y = v0.y * x[c.y & colmask];
          p1.x = v1.x * x[c.z & colmask];
          p1.y = v1.y * x[c.w & colmask];
          *reinterpret_cast<d2v *>(&prod[off]) = p0;
          *reinterpret_cast<d2v *>(&prod[off + 2]) = p1;
        } else if constexpr (VEC == 2) {
          const i2v c = ldg<NT>(reinterpret_cast<const i2v *>(col + k));
          const d2v v0 = ldg<NT>(reinterpret_cast<const d2v *>(val + k));
          d2v p0;
          p0.x = v0.x * x[c.x & colmask];
          p0.y = v0.y * x[c.y & colmask];
          *reinterpret_cast<d2v *>(&prod[off]) = p0;
        } else {
          const int c = ldg<NT>(col + k);
          const double v0 = ldg<NT>(val + k);
          prod[off] = v0 * x[c & colmask];
        }
      }
      __syncthreads();

      // ---- reduce phase: one lane per row, products added left to right
      for (int r = r0 + tid; r < r1; r += kBlock) {
        const int lo = ind[r], hi = ind[r + 1];
        // A row is finished in the tile that holds its last product (hi <= te); an empty
        // row sitting exactly on a tile boundary counts for the earlier tile.
        const bool done_earlier = hi <= ts && ts != (s & ~3);
        const bool starts_later = lo >= te && hi > te;
        if (done_earlier || starts_later) continue;
        double acc = (lo < ts) ? carry : 0.0;
        const int a = lo > ts ? lo : ts;
        const int b = hi < te ? hi : te;
        for (int k = a; k < b; k += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int idx = k + u - ts;
            idx = idx < TILE ? idx : TILE - 1;
            v[u] = prod[idx];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += (k + u < b) ? v[u] : 0.0;
        }
        if (hi <= te) {
          y[r] = acc;
          if (dotv) dsum += dotv[r] * acc;
        } else {
          carry = acc;  // the one row that crosses into the next tile stays with this lane
        }
      }
      __syncthreads();
      if (te >= e) break;
    }
  }

  if (partials) {
    const double t = block_sum(dsum, red);
    if (tid == 0) partials[blockIdx.x] = t;
  }
}

// ------------------------------------------------------------------ wave-level pipeline
//
// Same algorithm with ONE WAVEFRONT per chunk (tile of WT nonzeros) and no workgroup
// barrier: the four waves of a workgroup run decoupled, each with a private LDS slice.
// The loop is software-pipelined one chunk deep: while chunk i's x gathers return and its
// rows are reduced, the val/col/row-bound loads of chunk i+1 are already in flight, so
// every wave keeps HBM requests outstanding all the time (vmcnt waits only for the older
// gathers, never for the younger prefetch).  Requires every chunk to fit one tile, i.e.
// max row length <= WT/2 (the launcher falls back to csr_spmv_stream otherwise).
template <int WT>
struct WaveStage {
  static constexpr int STEPS = WT / 256;
  i4v c[STEPS];
  d2v v0[STEPS], v1[STEPS];
  int r0, r1, s, e;
  int lo0, hi0, lo1, hi1;
};

template <int WT, bool NT>
__device__ __forceinline__ void wave_issue(WaveStage<WT> &S, int chunk, int lane,
                                           const int2 *__restrict__ tab,
                                           const int *__restrict__ ind,
                                           const int *__restrict__ col,
                                           const double *__restrict__ val) {
  const int2 c0 = tab[chunk];
  const int2 c1 = tab[chunk + 1];
  S.r0 = c0.x;
  S.r1 = c1.x;
  S.s = c0.y;
  S.e = c1.y;
  const int ts = S.s & ~3;
#pragma unroll
  for (int st = 0; st < WaveStage<WT>::STEPS; ++st) {
    int k = ts + (st * 64 + lane) * 4;
    k = (k < S.e) ? k : ts;
    S.c[st] = ldg<NT>(reinterpret_cast<const i4v *>(col + k));
    S.v0[st] = ldg<NT>(reinterpret_cast<const d2v *>(val + k));
    S.v1[st] = ldg<NT>(reinterpret_cast<const d2v *>(val + k + 2));
  }
  const int ra = S.r0 + lane, rb = ra + 64;
  S.lo0 = S.hi0 = S.lo1 = S.hi1 = 0;
  if (ra < S.r1) {
    S.lo0 = ind[ra];
    S.hi0 = ind[ra + 1];
  }
  if (rb < S.r1) {
    S.lo1 = ind[rb];
    S.hi1 = ind[rb + 1];
  }
}

template <int WT, bool NT>
__global__ __launch_bounds__(kBlock) void csr_spmv_wave(
    int nchunks, int map_mode, int colmask, const int2 *__restrict__ tab,
    const int *__restrict__ ind, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ dotv,
    double *__restrict__ partials) {
  constexpr int STEPS = WT / 256;
  __shared__ double prod_all[4 * WT];
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *prod = prod_all + wid * WT;
  const int nwaves = gridDim.x * 4;
  // chunk visited by this wave in sweep `it`
  int base, stride;
  if (map_mode == 0) {
    base = (int)blockIdx.x * 4 + wid;  // neighbouring chunks run at the same time chip-wide
    stride = nwaves;
  } else {
    const int W = nwaves >> 3;  // waves per XCD: each XCD sweeps a contiguous stripe
    base = ((int)blockIdx.x & 7) * W + ((int)blockIdx.x >> 3) * 4 + wid;
    stride = nwaves;
  }
  double dsum = 0.0;
  int chunk = base;
  if (chunk < nchunks) {
    WaveStage<WT> cur;
    wave_issue<WT, NT>(cur, chunk, lane, tab, ind, col, val);
    while (true) {
      // x gathers of the current chunk (addresses = the col values just loaded)
      double xv[STEPS][4];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        xv[st][0] = x[cur.c[st].x & colmask];
        xv[st][1] = x[cur.c[st].y & colmask];
        xv[st][2] = x[cur.c[st].z & colmask];
        xv[st][3] = x[cur.c[st].w & colmask];
      }
      // prefetch the next chunk behind them
      const int next = chunk + stride;
      const bool has_next = next < nchunks;
      WaveStage<WT> nxt;
      if (has_next) wave_issue<WT, NT>(nxt, next, lane, tab, ind, col, val);

      const int ts = cur.s & ~3;
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int off = (st * 64 + lane) * 4;
        d2v p0, p1;
        p0.x = cur.v0[st].x * xv[st][0];
        p0.y = cur.v0[st].y * xv[st][1];
        p1.x = cur.v1[st].x * xv[st][2];
        p1.y = cur.v1[st].y * xv[st][3];
        *reinterpret_cast<d2v *>(&prod[off]) = p0;
        *reinterpret_cast<d2v *>(&prod[off + 2]) = p1;
      }
      // LDS operations of one wave execute in order; the fence only stops the compiler
      // from moving the reads above the writes
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();

      int m = 0;
      for (int r = cur.r0 + lane; r < cur.r1; r += 64, ++m) {
        int lo, hi;
        if (m == 0) {
          lo = cur.lo0;
          hi = cur.hi0;
        } else if (m == 1) {
          lo = cur.lo1;
          hi = cur.hi1;
        } else {
          lo = ind[r];
          hi = ind[r + 1];
        }
        double acc = 0.0;
        for (int k = lo; k < hi; k += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int idx = k + u - ts;
            idx = idx < WT ? idx : WT - 1;
            v[u] = prod[idx];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += (k + u < hi) ? v[u] : 0.0;
        }
        y[r] = acc;
        if (dotv) dsum += dotv[r] * acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (!has_next) break;
      cur = nxt;
      chunk = next;
    }
  }
  if (partials) {
    const double t = block_sum(dsum, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// ------------------------------------------------------------------ one chunk per wave, no loop
//
// The fastest form measured on MI355X (profiles/): the grid holds ONE WAVE PER CHUNK and the
// hardware dispatcher, not a persistent loop, walks the matrix -- waves start in address
// order, which keeps the HBM request stream nearly linear, and a CU always has fresh waves
// to cover the tab -> val/col -> x-gather dependency chain.  WPB waves share a workgroup
// only to share its LDS allocation and the dot-product epilogue.
//   LAYOUT 0: each lane loads 4 consecutive nonzeros per step (16-byte col, 2x16-byte val)
//   LAYOUT 1: each lane loads 1 nonzero per step (4-byte col, 8-byte val): the x gather of
//             one instruction then covers 64 consecutive nonzeros (~9 stencil rows) and
//             touches about half as many cache lines
template <int WT, int WPB, int LAYOUT, bool NT>
__global__ __launch_bounds__(64 * WPB) void csr_spmv_w1(
    int nchunks, int colmask, int stripe, int target, int kmax, const int2 *__restrict__ tab,
    const int *__restrict__ ind, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ dotv,
    double *__restrict__ partials) {
  __shared__ double prod_all[WPB * WT];
  __shared__ double red[WPB];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *prod = prod_all + wid * WT;
  // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8
  // share an L2; observed, used for speed only).  With stripe > 0 the k-th workgroup of
  // XCD j takes position ((k / stripe) * 8 + j) * stripe + k % stripe, i.e. every XCD walks
  // contiguous stripes of `stripe` workgroups while the eight of them stay on adjacent
  // stripes -- the x entries of neighbouring grid lines are then re-used in that XCD's L2
  // instead of being fetched once per XCD.
  int vb = (int)blockIdx.x;
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int chunk = vb * WPB + wid;
  double dsum = 0.0;
  if (chunk < nchunks) {
    // The chunk's nonzeros lie in the window [kb, kb + WT) with kb = chunk*target known
    // WITHOUT the chunk table, so the val/col stream is issued first and the table / row
    // bounds (needed only by the reduce phase) load behind it: two dependent memory
    // levels (val+col -> x gather) instead of three.
    const int kb = chunk * target;
    const int2 c0 = tab[chunk];
    const int2 c1 = tab[chunk + 1];
    const int ra_base = c0.x, r1 = c1.x;
    int lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
    if constexpr (LAYOUT == 0) {
      constexpr int STEPS = WT / 256;
      i4v c[STEPS];
      d2v v0[STEPS], v1[STEPS];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        int k = kb + (st * 64 + lane) * 4;
        k = (k < kmax) ? k : kmax;  // the last window may run past the (padded) arrays
        c[st] = ldg<NT>(reinterpret_cast<const i4v *>(col + k));
        v0[st] = ldg<NT>(reinterpret_cast<const d2v *>(val + k));
        v1[st] = ldg<NT>(reinterpret_cast<const d2v *>(val + k + 2));
      }
      const int ra = ra_base + lane, rb = ra + 64;
      if (ra < r1) { lo0 = ind[ra]; hi0 = ind[ra + 1]; }
      if (rb < r1) { lo1 = ind[rb]; hi1 = ind[rb + 1]; }
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int off = (st * 64 + lane) * 4;
        d2v p0, p1;
        p0.x = v0[st].x * x[c[st].x & colmask];
        p0.y = v0[st].y * x[c[st].y & colmask];
        p1.x = v1[st].x * x[c[st].z & colmask];
        p1.y = v1[st].y * x[c[st].w & colmask];
        *reinterpret_cast<d2v *>(&prod[off]) = p0;
        *reinterpret_cast<d2v *>(&prod[off + 2]) = p1;
      }
    } else {
      constexpr int STEPS = WT / 64;
      int c[STEPS];
      double v[STEPS];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        int k = kb + st * 64 + lane;
        k = (k < kmax + 3) ? k : kmax + 3;
        c[st] = ldg<NT>(col + k);
        v[st] = ldg<NT>(val + k);
      }
      const int ra = ra_base + lane, rb = ra + 64;
      if (ra < r1) { lo0 = ind[ra]; hi0 = ind[ra + 1]; }
      if (rb < r1) { lo1 = ind[rb]; hi1 = ind[rb + 1]; }
      double xv[STEPS];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) xv[st] = x[c[st] & colmask];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) prod[st * 64 + lane] = v[st] * xv[st];
    }
    // LDS operations of one wave execute in order; the fence only pins the compiler
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    int m = 0;
    for (int r = ra_base + lane; r < r1; r += 64, ++m) {
      int lo, hi;
      if (m == 0) { lo = lo0; hi = hi0; }
      else if (m == 1) { lo = lo1; hi = hi1; }
      else { lo = ind[r]; hi = ind[r + 1]; }
      double acc = 0.0;
      for (int k = lo; k < hi; k += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          int idx = k + u - kb;
          idx = idx < WT ? idx : WT - 1;
          t[u] = prod[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (k + u < hi) ? t[u] : 0.0;
      }
      y[r] = acc;
      if (dotv) dsum += dotv[r] * acc;
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < WPB; ++i) t += red[i];
      partials[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------ w2: no dependent loads
//
// csr_spmv_w1 still reads `ind` for its row bounds, and that address depends on the chunk
// table: tab -> ind is a chain of two memory latencies that the reduce phase has to wait
// for (in-kernel stamps: ~12.6 k cycles from wave start to "everything landed" against
// ~4.6 k for the val/col stream alone, profiles/).  w2 replaces `ind` by a per-chunk table
// of 16-bit row offsets relative to the chunk's window, stored at a FIXED stride
// (rowoff[chunk*E + i] = ind[r0+i] - chunk*target, padded with the chunk's end offset), so
// every load of a wave -- val, col, row offsets, table entry -- is issued at wave start
// and the only dependent level left is col -> x.  HBM traffic: 2*E bytes per chunk (384 B
// for the 7-point operator, ~2.6 B/row) instead of 4 B/row of `ind`.
// interleave col/val into 768-byte tiles of 64 nonzeros (see csr_spmv_w2<..., PACKED>)
__global__ __launch_bounds__(256) void pack_kernel(long count, const int *__restrict__ col,
                                                   const double *__restrict__ val,
                                                   char *__restrict__ packed) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (long)gridDim.x * blockDim.x) {
    char *tile = packed + (size_t)(k >> 6) * 768;
    const int o = (int)(k & 63);
    *reinterpret_cast<int *>(tile + o * 4) = col[k];
    *reinterpret_cast<double *>(tile + 256 + o * 8) = val[k];
  }
}

template <int NP>
__global__ __launch_bounds__(256) void build_rowoff_kernel(int nchunks, int target,
                                                           const int2 *__restrict__ tab,
                                                           const int *__restrict__ ind,
                                                           unsigned short *__restrict__ rowoff) {
  constexpr int E = 64 * NP;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunk = (int)(gid / E), i = (int)(gid % E);
  if (chunk >= nchunks) return;
  const int r0 = tab[chunk].x, r1 = tab[chunk + 1].x;
  const int r = r0 + i < r1 ? r0 + i : r1;
  rowoff[gid] = (unsigned short)(ind[r] - chunk * target);
}

__global__ void max_chunk_rows_kernel(int nchunks, const int2 *__restrict__ tab, int *__restrict__ out) {
  int m = 0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += gridDim.x * blockDim.x)
    m = max(m, tab[c + 1].x - tab[c].x);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// PACKED: val/col come from ONE interleaved stream (tiles of 64 nonzeros: 256 B of column
// indices followed by 512 B of values) instead of two arrays -- fewer concurrent HBM streams
template <int WT, int NP, int WPB, bool NTL = false, bool NTS = false, bool PACKED = false>
__global__ __launch_bounds__(64 * WPB) void csr_spmv_w2(
    int chunk0, int nchunks, int colmask, int stripe, int target, int kmax, const int2 *__restrict__ tab,
    const unsigned short *__restrict__ rowoff, const int *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ dotv, double *__restrict__ partials, const int *__restrict__ skip) {
  constexpr int STEPS = WT / 256;
  constexpr int E = 64 * NP;
  if (skip && *skip) return;  // asynchronous solver loop already finished: no-op launch
  // exactly 32 KiB for 4 waves x 1024 products: five workgroups fit the CU's 160 KiB (a separate
  // array for the dot partials would cost the fifth)
  __shared__ double prod_all[WPB * WT];
  double *red = prod_all;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *prod = prod_all + wid * WT;
  int vb = (int)blockIdx.x;  // XCD-aware placement, see csr_spmv_w1
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int chunk = chunk0 + vb * WPB + wid;  // chunks [chunk0, nchunks) belong to this launch
  double dsum = 0.0;
  if (chunk < nchunks) {
    const int kb = chunk * target;
    // --- every load of this wave, none depends on another
    i4v c[STEPS];
    d2v v0[STEPS], v1[STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      int k = kb + (st * 64 + lane) * 4;
      k = (k < kmax) ? k : kmax;
      if constexpr (PACKED) {
        const char *tile = reinterpret_cast<const char *>(val) + (size_t)(k >> 6) * 768;
        const int o = k & 63;
        c[st] = *reinterpret_cast<const i4v *>(tile + o * 4);
        v0[st] = *reinterpret_cast<const d2v *>(tile + 256 + o * 8);
        v1[st] = *reinterpret_cast<const d2v *>(tile + 256 + o * 8 + 16);
      } else {
        c[st] = ldg<NTL>(reinterpret_cast<const i4v *>(col + k));
        v0[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k));
        v1[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k + 2));
      }
    }
    const unsigned short *ro = rowoff + (size_t)chunk * E;
    int lo[NP], hi[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int i = 64 * m + lane;
      lo[m] = ro[i];
      hi[m] = ro[i + 1 < E ? i + 1 : E - 1];
    }
    const int r0 = tab[chunk].x;
    const int nr = tab[chunk + 1].x - r0;
    // --- x gathers (the one dependent level), products into the wave's LDS slice
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int off = (st * 64 + lane) * 4;
      d2v p0, p1;
      p0.x = v0[st].x * x[c[st].x & colmask];
      p0.y = v0[st].y * x[c[st].y & colmask];
      p1.x = v1[st].x * x[c[st].z & colmask];
      p1.y = v1[st].y * x[c[st].w & colmask];
      *reinterpret_cast<d2v *>(&prod[off]) = p0;
      *reinterpret_cast<d2v *>(&prod[off + 2]) = p1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // --- one lane per row, products added left to right (reference order, csr_mat.c:49-54)
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int i = 64 * m + lane;
      if (i < nr) {
        double acc = 0.0;
        for (int k = lo[m]; k < hi[m]; k += 8) {
          double t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int idx = k + u;
            idx = idx < WT ? idx : WT - 1;
            t[u] = prod[idx];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += (k + u < hi[m]) ? t[u] : 0.0;
        }
        if constexpr (NTS)
          __builtin_nontemporal_store(acc, &y[r0 + i]);
        else
          y[r0 + i] = acc;
        if (dotv) dsum += dotv[r0 + i] * acc;
      }
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    __syncthreads();  // red[] lives in wave 0's slice: every wave must be done with its rows
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < WPB; ++i) t += red[i];
      partials[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------ w3: x staged in LDS
//
// What holds csr_spmv_w2 at ~70 % is not HBM but the per-CU L1: every x gather instruction
// touches ~20 cache lines (counters: TCP busy 92 %, a third of the cycles in pending-miss
// stalls).  w3 takes the gathers off the L1.  Per chunk the builder below lists the
// distinct 128-byte blocks of x (16 doubles) that the chunk's nonzeros reference -- for a
// banded operator a handful of short windows -- and rewrites the chunk's column indices as
// 16-bit offsets into that list.  The wave then
//   1. issues the val stream, the 16-bit columns, the row offsets and the block list
//      (all independent, fixed-stride addresses),
//   2. loads the listed x blocks with fully coalesced 16-byte-per-lane loads (8 lanes per
//      block, 8 blocks per instruction) and parks them in its LDS slice,
//   3. gathers x from LDS, overwrites the slice with the rounded products, and
//   4. reduces rows left to right exactly like w2 (same products, same order => the
//      same bits as csr_mat.c:49-54).
// HBM traffic per nonzero drops from 12 to 10 bytes (+ 4*NB bytes of block list per
// chunk), L1 requests per chunk from ~400 lines to ~150.  Matrices whose chunks reference
// more than NB blocks stay on w2.
typedef unsigned short us4v __attribute__((ext_vector_type(4)));

// SHIFT = 4: ids are 16-entry x blocks (csr_spmv_w3); SHIFT = 0: ids are the columns themselves and the
// 16-bit value is the column's rank in the chunk's sorted list of distinct columns (csr_spmv_w5)
template <int NB, int SHIFT = 4>
__global__ __launch_bounds__(64) void build_w3_kernel(int nchunks, int target, int write,
                                                      const int2 *__restrict__ tab,
                                                      const int *__restrict__ col,
                                                      int *__restrict__ blist,
                                                      unsigned short *__restrict__ col16,
                                                      int *__restrict__ maxblocks) {
  constexpr int WT = 1024;
  constexpr int kNone = 0x7fffffff;
  __shared__ int keys[WT];
  __shared__ int ulist[WT];
  const int chunk = blockIdx.x;
  const int lane = threadIdx.x;
  if (chunk >= nchunks) return;
  const int s = tab[chunk].y, e = tab[chunk + 1].y;
  const long kb = (long)chunk * target;
  for (int i = lane; i < WT; i += 64) {
    const long k = kb + i;
    keys[i] = (k >= s && k < e) ? (col[k] >> SHIFT) : kNone;
  }
  __syncthreads();
  // bitonic sort of the 1024 block ids
  for (int size = 2; size <= WT; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < WT / 2; t += 64) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const bool up = (i & size) == 0;
        const int a = keys[i], b = keys[j];
        if ((a > b) == up) {
          keys[i] = b;
          keys[j] = a;
        }
      }
      __syncthreads();
    }
  // distinct ids, in ascending order
  int count = 0;
  for (int base = 0; base < WT; base += 64) {
    const int i = base + lane;
    const int k = keys[i];
    const bool flag = k != kNone && (i == 0 || k != keys[i - 1]);
    const unsigned long long bal = __ballot(flag);
    const int pos = count + __popcll(bal & ((1ull << lane) - 1ull));
    if (flag) ulist[pos] = k;
    count += __popcll(bal);
  }
  __syncthreads();
  int nrun = 0;  // runs of consecutive ids in the sorted list (csr_spmv_w6 keeps up to kW6Runs of them in registers)
  if constexpr (SHIFT == 4) {
    for (int base = 0; base < count; base += 64) {
      const int i = base + lane;
      const bool flag = i < count && (i == 0 || ulist[i] != ulist[i - 1] + 1);
      nrun += __popcll(__ballot(flag));
    }
  }
  if (lane == 0) {
    atomicMax(maxblocks, count);
    if constexpr (SHIFT == 4) {  // (the w5 builder passes a single counter)
      if (count > 64) atomicAdd(maxblocks + 1, 1);  // chunks that do not fit csr_spmv_w3's 64-block list
      if (count > 32) atomicAdd(maxblocks + 2, 1);  // ... its 32-block list
      if (count > 64 || nrun > 8) atomicAdd(maxblocks + 3, 1);  // chunks csr_spmv_w6 serves through memory
    }
  }
  if (!write) return;
  if (SHIFT == 4 && count > NB) {
    // an OUTLIER chunk of a matrix that otherwise qualifies (ensure_w3): no list -- the kernel sees the -1 and
    // gathers this chunk's x entries from memory through the int32 columns
    for (int i = lane; i < NB; i += 64) blist[(size_t)chunk * NB + i] = -1;
    if (col16)
      for (int i = lane; i < WT; i += 64) col16[(size_t)chunk * WT + i] = 0;
    return;
  }
  if (count > NB) return;
  // unused list slots hold -2: the kernel issues no load for them (SHIFT 0, csr_spmv_w5: padded with the last column)
  for (int i = lane; i < NB; i += 64)
    blist[(size_t)chunk * NB + i] = i < count ? ulist[i] : (SHIFT == 4 ? -2 : (count ? ulist[count - 1] : 0));
  if (!col16) return;  // csr_spmv_w6: the list alone (the columns stay the csr_mat's own)
  for (int i = lane; i < WT; i += 64) {
    const long k = kb + i;
    unsigned short v = 0;
    if (k >= s && k < e) {
      const int c = col[k];
      const int b = c >> SHIFT;
      int lo = 0, hi = count - 1;  // b is in ulist[0, count)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ulist[mid] < b)
          lo = mid + 1;
        else
          hi = mid;
      }
      v = (unsigned short)(SHIFT ? lo * 16 + (c & 15) : lo);
    }
    col16[(size_t)chunk * WT + i] = v;
  }
}

typedef unsigned short us2v __attribute__((ext_vector_type(2)));

// PAIRS: each lane takes 2 consecutive nonzeros per step (8 steps) instead of 4 (4 steps): every
// value load instruction then covers 1 KB contiguous -- 8 cache lines instead of the 16 that the two
// half-loads of the 4-wide form touch twice
// OUTL: a few chunks of the matrix reference more than NB blocks (their block list holds -1): those gather x
// from memory through the int32 columns `colfull`, like csr_spmv_w2 -- same products, same order
typedef int i2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
template <int NP, int NB, int WPB, bool NTS, bool NTL = false, bool PAIRS = false, bool OUTL = false>
__global__ __launch_bounds__(64 * WPB) void csr_spmv_w3(
    int chunk0, int nchunks, int stripe, int target, int kmax, int ncols,
    const int2 *__restrict__ tab, const unsigned short *__restrict__ rowoff,
    const unsigned short *__restrict__ col16, const int *__restrict__ blist,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ dotv, double *__restrict__ partials, const int *__restrict__ skip,
    const int *__restrict__ perm, const int *__restrict__ rowperm, const int *__restrict__ colfull = nullptr,
    int colmod = 0) {
  // colmod > 0 (PSP_W3_COLMOD under PSP_TUNING=1; WRONG RESULTS, timing only): chunk c reads the 16-bit columns of chunk
  // c % colmod -- the column stream then comes out of L2 instead of HBM while every other access, the LDS gathers and the
  // arithmetic stay what they are: the time this buys bounds what ANY compression of the columns can buy (round 4)
  // rowperm (renumbered operators, psp_reorder.hip): row r of this matrix is row rowperm[r] of the
  // caller's: its sum is stored to y[rowperm[r]] and meets dotv[rowperm[r]]
  constexpr int WT = 1024;
  constexpr int STEPS = WT / 256;
  constexpr int E = 64 * NP;
  constexpr int XW = NB * 16;             // doubles in the x window
  constexpr int LW = XW > WT ? XW : WT;   // the products overwrite the window
  constexpr int XL = NB / 8;              // 16-byte x loads per lane
  static_assert(NB == 32 || NB == 64 || NB == 128, "block list is read one or two entries per lane");
  if (skip && *skip) return;  // asynchronous solver loop already finished: no-op launch
  __shared__ double lds_all[WPB * LW];  // 32 KiB at NB <= 64: five workgroups per CU
  double *red = lds_all;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *buf = lds_all + wid * LW;
  int vb = (int)blockIdx.x;  // XCD-aware placement, see csr_spmv_w1
  if (perm) {
    vb = perm[vb];  // plane-sweeping schedule (build_schedule); < 0: padding slot
  } else if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int chunk = chunk0 + vb * WPB + wid;  // chunks [chunk0, nchunks) belong to this launch
  double dsum = 0.0;
  if (vb >= 0 && chunk < nchunks) {
    const int kb = chunk * target;
    // --- independent loads: values, 16-bit columns, row offsets, block list, table entry
    d2v v0[STEPS], v1[STEPS];
    us4v c[STEPS];
    const unsigned short *cp = col16 + (size_t)(colmod > 0 ? chunk % colmod : chunk) * WT;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if constexpr (PAIRS) {  // v0 = nonzeros (2 st) * 128 + 2 lane .. +1, v1 = the same in the next 128
        int k0 = kb + (2 * st) * 128 + 2 * lane, k1 = k0 + 128;
        k0 = (k0 < kmax + 2) ? k0 : kmax + 2;
        k1 = (k1 < kmax + 2) ? k1 : kmax + 2;
        v0[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k0));
        v1[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k1));
        const us2v c0 = ldg<NTL>(reinterpret_cast<const us2v *>(cp + (2 * st) * 128 + 2 * lane));
        const us2v c1 = ldg<NTL>(reinterpret_cast<const us2v *>(cp + (2 * st + 1) * 128 + 2 * lane));
        c[st].x = c0.x; c[st].y = c0.y; c[st].z = c1.x; c[st].w = c1.y;
      } else {
        int k = kb + (st * 64 + lane) * 4;
        k = (k < kmax) ? k : kmax;
        v0[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k));
        v1[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k + 2));
        c[st] = ldg<NTL>(reinterpret_cast<const us4v *>(cp + (st * 64 + lane) * 4));
      }
    }
    const int *bl = blist + (size_t)chunk * NB;
    const int blk0 = bl[NB == 32 ? (lane & 31) : lane];
    int blk1 = 0;
    if constexpr (NB == 128) blk1 = bl[64 + lane];
    const unsigned short *ro = rowoff + (size_t)chunk * E;
    int lo[NP], hi[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int i = 64 * m + lane;
      lo[m] = ro[i];
      hi[m] = ro[i + 1 < E ? i + 1 : E - 1];
    }
    const int r0 = tab[chunk].x;
    const int nr = tab[chunk + 1].x - r0;
    d2v p0[STEPS], p1[STEPS];
    bool outlier = false;
    if constexpr (OUTL) outlier = __builtin_amdgcn_readfirstlane(blk0) == -1;  // wave-uniform: the whole list is -1
    if (OUTL && outlier) {
      // --- outlier chunk: x straight from memory through the int32 columns (padding holds valid columns)
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        if constexpr (PAIRS) {
          int k0 = kb + (2 * st) * 128 + 2 * lane, k1 = k0 + 128;
          k0 = (k0 < kmax + 2) ? k0 : kmax + 2;
          k1 = (k1 < kmax + 2) ? k1 : kmax + 2;
          const i2v c0 = *reinterpret_cast<const i2v *>(colfull + k0);
          const i2v c1 = *reinterpret_cast<const i2v *>(colfull + k1);
          p0[st].x = v0[st].x * x[c0.x];
          p0[st].y = v0[st].y * x[c0.y];
          p1[st].x = v1[st].x * x[c1.x];
          p1[st].y = v1[st].y * x[c1.y];
        } else {
          int k = kb + (st * 64 + lane) * 4;
          k = (k < kmax) ? k : kmax;
          const i4v cc = *reinterpret_cast<const i4v *>(colfull + k);
          p0[st].x = v0[st].x * x[cc.x];
          p0[st].y = v0[st].y * x[cc.y];
          p1[st].x = v1[st].x * x[cc.z];
          p1[st].y = v1[st].y * x[cc.w];
        }
      }
    } else {
      // --- the chunk's x blocks: 8 lanes per 128-byte block, 8 blocks per load instruction
      d2v xw[XL];
#pragma unroll
      for (int j = 0; j < XL; ++j) {
        const int src = (j & 7) * 8 + (lane >> 3);
        const int b = __shfl((NB == 128 && j >= 8) ? blk1 : blk0, src, 64);
        const long e0 = (long)b * 16 + (lane & 7) * 2;
        if (b < 0) {  // unused slot of a chunk with fewer than NB blocks: nothing to fetch
          xw[j].x = 0.0;
          xw[j].y = 0.0;
        } else if (e0 + 1 < ncols) {
          xw[j] = *reinterpret_cast<const d2v *>(x + e0);
        } else {  // the block that holds the end of x
          xw[j].x = e0 < ncols ? x[e0] : 0.0;
          xw[j].y = 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < XL; ++j) *reinterpret_cast<d2v *>(&buf[(j * 64 + lane) * 2]) = xw[j];
      // LDS operations of one wave execute in order; the fences only pin the compiler
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // --- gathers from LDS, then the products take the window's place
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        p0[st].x = v0[st].x * buf[c[st].x];
        p0[st].y = v0[st].y * buf[c[st].y];
        p1[st].x = v1[st].x * buf[c[st].z];
        p1[st].y = v1[st].y * buf[c[st].w];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if constexpr (PAIRS) {
        *reinterpret_cast<d2v *>(&buf[(2 * st) * 128 + 2 * lane]) = p0[st];
        *reinterpret_cast<d2v *>(&buf[(2 * st + 1) * 128 + 2 * lane]) = p1[st];
      } else {
        const int off = (st * 64 + lane) * 4;
        *reinterpret_cast<d2v *>(&buf[off]) = p0[st];
        *reinterpret_cast<d2v *>(&buf[off + 2]) = p1[st];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // --- one lane per row, products added left to right (reference order, csr_mat.c:49-54)
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int i = 64 * m + lane;
      if (i < nr) {
        double acc = 0.0;
        for (int k = lo[m]; k < hi[m]; k += 8) {
          double t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int idx = k + u;
            idx = idx < WT ? idx : WT - 1;
            t[u] = buf[idx];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += (k + u < hi[m]) ? t[u] : 0.0;
        }
        const int ro_ = rowperm ? rowperm[r0 + i] : r0 + i;
        if constexpr (NTS)
          __builtin_nontemporal_store(acc, &y[ro_]);
        else
          y[ro_] = acc;
        if (dotv) dsum += dotv[ro_] * acc;
      }
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    __syncthreads();  // red[] lives in wave 0's slice: every wave must be done with its rows
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < WPB; ++i) t += red[i];
      partials[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------ w6: the CSR arrays as stored + x staged in LDS
//
// Round 5 (VERDICT r4 "Next" #6).  csr_spmv_w2 is the one kernel that streams a csr_mat the way north_star words it --
// int32 col_ind + fp64 val, 12 bytes per nonzero, nothing re-encoded -- and it is bound by the L1, not by HBM: every x
// gather instruction touches ~20 cache lines (1.14e9 L1 accesses per launch at 512^3 against csr_spmv_w4's 2.7e8, 1.15x
// fabric traffic; profiles/r4_spmv_w2_pmc_summary.txt).  w3 takes the gathers off the L1 but pays for it with a second
// copy of the columns (16-bit, chunk-local).  w6 keeps w3's staging and w2's streams: the chunk's x blocks come from the
// same block list (64 ids per chunk, 0.25 bytes per nonzero -- the only side table besides w2's row offsets), and a
// nonzero's LDS slot is computed from its int32 column on the fly.  That works because the list of a banded matrix is a
// handful of RUNS of consecutive blocks (seven for the 7-point operator: one per offset): the wave finds the runs with one
// ballot over the sorted list, keeps (first block, first slot) of up to kW6Runs of them in scalar registers, and a
// column's slot is ((c >> 4) + base_r) * 16 + (c & 15) with base_r picked by at most kW6Runs compares.  A chunk with
// more runs, or more than 64 blocks, gathers through memory like w2 (wave-uniform branch).  Same products, same order
// (csr_mat.c:49-54): the same bits as every other kernel.
constexpr int kW6Runs = 8;

// PAIRS (as in csr_spmv_w3): a lane takes 2 consecutive nonzeros per load (8 col + 16 val bytes) instead of 4: every load
// instruction then covers one contiguous run of cache lines; with NTL the value / column streams are non-temporal
template <int NP, int WPB, bool NTS, bool NTL = false, bool PAIRS = false>
__global__ __launch_bounds__(64 * WPB) void csr_spmv_w6(
    int chunk0, int nchunks, int stripe, int target, int kmax, int ncols, const int2 *__restrict__ tab,
    const unsigned short *__restrict__ rowoff, const int *__restrict__ col, const int *__restrict__ blist,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ dotv, double *__restrict__ partials, const int *__restrict__ skip) {
  constexpr int WT = 1024;
  constexpr int STEPS = WT / 256;
  constexpr int E = 64 * NP;
  constexpr int NB = 64;
  constexpr int XL = NB / 8;  // 16-byte x loads per lane
  if (skip && *skip) return;
  __shared__ double lds_all[WPB * WT];  // the x window (64 blocks x 16 doubles), then the products in its place
  double *red = lds_all;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *buf = lds_all + wid * WT;
  int vb = (int)blockIdx.x;  // XCD-aware placement, see csr_spmv_w1
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int chunk = chunk0 + vb * WPB + wid;
  double dsum = 0.0;
  if (chunk < nchunks) {
    const int kb = chunk * target;
    // --- independent loads: values and columns exactly as the csr_mat stores them, row offsets, block list, table entry
    d2v v0[STEPS], v1[STEPS];
    i4v c[STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if constexpr (PAIRS) {  // v0 = nonzeros (2 st) * 128 + 2 lane .. + 1, v1 = the same in the next 128
        int k0 = kb + (2 * st) * 128 + 2 * lane, k1 = k0 + 128;
        k0 = (k0 < kmax + 2) ? k0 : kmax + 2;
        k1 = (k1 < kmax + 2) ? k1 : kmax + 2;
        v0[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k0));
        v1[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k1));
        const i2v c0 = ldg<NTL>(reinterpret_cast<const i2v *>(col + k0));
        const i2v c1 = ldg<NTL>(reinterpret_cast<const i2v *>(col + k1));
        c[st].x = c0.x; c[st].y = c0.y; c[st].z = c1.x; c[st].w = c1.y;
      } else {
        int k = kb + (st * 64 + lane) * 4;
        k = (k < kmax) ? k : kmax;
        c[st] = ldg<NTL>(reinterpret_cast<const i4v *>(col + k));
        v0[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k));
        v1[st] = ldg<NTL>(reinterpret_cast<const d2v *>(val + k + 2));
      }
    }
    const int blk0 = blist[(size_t)chunk * NB + lane];
    const unsigned short *ro = rowoff + (size_t)chunk * E;
    int lo[NP], hi[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int i = 64 * m + lane;
      lo[m] = ro[i];
      hi[m] = ro[i + 1 < E ? i + 1 : E - 1];
    }
    const int r0 = tab[chunk].x;
    const int nr = tab[chunk + 1].x - r0;
    // --- the runs of the sorted list (unused slots hold -2, an over-full chunk's list -1 everywhere)
    const int prev = __shfl_up(blk0, 1, 64);
    const bool starts = blk0 >= 0 && (lane == 0 || blk0 != prev + 1);
    unsigned long long runs = __ballot(starts);
    const int nruns = __popcll(runs);
    const bool direct = __builtin_amdgcn_readfirstlane(blk0) == -1 || nruns > kW6Runs;  // wave-uniform
    d2v p0[STEPS], p1[STEPS];
    if (direct) {
      // --- x straight from memory through the columns (padding holds valid columns), like csr_spmv_w2
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        p0[st].x = v0[st].x * x[c[st].x];
        p0[st].y = v0[st].y * x[c[st].y];
        p1[st].x = v1[st].x * x[c[st].z];
        p1[st].y = v1[st].y * x[c[st].w];
      }
    } else {
      // (first block, slot - first block) of every run, wave-uniform scalars (named one by one: an array that a lambda
      // captures by reference ends up in scratch memory -- 48 bytes per lane, 2.8 GB of extra writes per launch at 512^3,
      // measured); unused runs can never be chosen
#define PSP_W6_RUN(R)                                                          \
  const int i##R = runs ? __builtin_ctzll(runs) : 0;                           \
  const int f##R = __builtin_amdgcn_readlane(blk0, i##R);                      \
  const int rs##R = (R < nruns) ? f##R : 0x7fffffff;                           \
  const int rb##R = i##R - f##R;                                               \
  runs &= runs - 1;
      PSP_W6_RUN(0) PSP_W6_RUN(1) PSP_W6_RUN(2) PSP_W6_RUN(3) PSP_W6_RUN(4) PSP_W6_RUN(5) PSP_W6_RUN(6) PSP_W6_RUN(7)
#undef PSP_W6_RUN
      (void)rs0;
      // --- the chunk's x blocks: 8 lanes per 128-byte block, 8 blocks per load instruction
      d2v xw[XL];
#pragma unroll
      for (int j = 0; j < XL; ++j) {
        const int b = __shfl(blk0, j * 8 + (lane >> 3), 64);
        const long e0 = (long)b * 16 + (lane & 7) * 2;
        if (b < 0) {  // unused slot: nothing to fetch
          xw[j].x = 0.0;
          xw[j].y = 0.0;
        } else if (e0 + 1 < ncols) {
          xw[j] = *reinterpret_cast<const d2v *>(x + e0);
        } else {  // the block that holds the end of x
          xw[j].x = e0 < ncols ? x[e0] : 0.0;
          xw[j].y = 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < XL; ++j) *reinterpret_cast<d2v *>(&buf[(j * 64 + lane) * 2]) = xw[j];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // --- slot of a column: its block's rank in the list (from the runs) * 16 + its place in the block.  Entries of the
      // window that belong to the next chunk may name blocks outside this list: their slot is meaningless, masked into the
      // slice, and their product is never added
#define PSP_W6_SLOT(OUT, CC)                                                   \
  {                                                                            \
    const int b_ = (CC) >> 4;                                                  \
    int base_ = rb0;                                                           \
    base_ = (b_ >= rs1) ? rb1 : base_;                                         \
    base_ = (b_ >= rs2) ? rb2 : base_;                                         \
    base_ = (b_ >= rs3) ? rb3 : base_;                                         \
    base_ = (b_ >= rs4) ? rb4 : base_;                                         \
    base_ = (b_ >= rs5) ? rb5 : base_;                                         \
    base_ = (b_ >= rs6) ? rb6 : base_;                                         \
    base_ = (b_ >= rs7) ? rb7 : base_;                                         \
    OUT = (((b_ + base_) << 4) + ((CC) & 15)) & (WT - 1);                      \
  }
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        int s0, s1, s2, s3;
        PSP_W6_SLOT(s0, c[st].x)
        PSP_W6_SLOT(s1, c[st].y)
        PSP_W6_SLOT(s2, c[st].z)
        PSP_W6_SLOT(s3, c[st].w)
        p0[st].x = v0[st].x * buf[s0];
        p0[st].y = v0[st].y * buf[s1];
        p1[st].x = v1[st].x * buf[s2];
        p1[st].y = v1[st].y * buf[s3];
      }
#undef PSP_W6_SLOT
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if constexpr (PAIRS) {
        *reinterpret_cast<d2v *>(&buf[(2 * st) * 128 + 2 * lane]) = p0[st];
        *reinterpret_cast<d2v *>(&buf[(2 * st + 1) * 128 + 2 * lane]) = p1[st];
      } else {
        const int off = (st * 64 + lane) * 4;
        *reinterpret_cast<d2v *>(&buf[off]) = p0[st];
        *reinterpret_cast<d2v *>(&buf[off + 2]) = p1[st];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // --- one lane per row, products added left to right (reference order, csr_mat.c:49-54)
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int i = 64 * m + lane;
      if (i < nr) {
        double acc = 0.0;
        for (int k = lo[m]; k < hi[m]; k += 8) {
          double t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int idx = k + u;
            idx = idx < WT ? idx : WT - 1;
            t[u] = buf[idx];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += (k + u < hi[m]) ? t[u] : 0.0;
        }
        if constexpr (NTS)
          __builtin_nontemporal_store(acc, &y[r0 + i]);
        else
          y[r0 + i] = acc;
        if (dotv) dsum += dotv[r0 + i] * acc;
      }
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    __syncthreads();  // red[] lives in wave 0's slice: every wave must be done with its rows
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < WPB; ++i) t += red[i];
      partials[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------ w5: distinct columns staged in LDS
//
// For numberings that scatter a chunk's columns over more x blocks than csr_spmv_w3's list holds
// (unstructured meshes, shuffled node ids) the block form stages mostly unused entries.  w5 stages exactly
// what the chunk needs: the builder lists the chunk's DISTINCT columns (sorted; 330-410 of them for 1000
// nonzeros of a 3-D FEM operator with 3 unknowns per node) and rewrites the chunk's columns as 16-bit ranks
// in that list.  The wave loads the list (coalesced, fixed stride), gathers x once per distinct column --
// a third of the gathers csr_spmv_w2 issues, and consecutive lanes take neighbouring columns, so every cache
// line of x is looked up once per chunk instead of ~10 times -- parks the entries in its LDS slice and goes on
// exactly like w3: products from LDS, rows added left to right (csr_mat.c:49-54) => the same bits.
// No renumbering, no extra passes over x or y.  Bytes: 10 per nonzero + 4 per list slot.
template <int NP, int NU64, int WPB, bool NTS>
__global__ __launch_bounds__(64 * WPB) void csr_spmv_w5(
    int chunk0, int nchunks, int stripe, int target, int kmax, const int2 *__restrict__ tab,
    const unsigned short *__restrict__ rowoff, const unsigned short *__restrict__ col16,
    const int *__restrict__ ulist, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, const double *__restrict__ dotv, double *__restrict__ partials,
    const int *__restrict__ skip) {
  constexpr int WT = 1024;
  constexpr int STEPS = WT / 256;
  constexpr int E = 64 * NP;
  constexpr int NU = 64 * NU64;  // list slots per chunk (<= WT: the products overwrite the staged entries)
  static_assert(NU <= WT, "the staged entries must fit the product slice");
  if (skip && *skip) return;
  __shared__ double lds_all[WPB * WT];
  double *red = lds_all;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *buf = lds_all + wid * WT;
  int vb = (int)blockIdx.x;  // XCD-aware placement, see csr_spmv_w1
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int chunk = chunk0 + vb * WPB + wid;
  double dsum = 0.0;
  if (chunk < nchunks) {
    const int kb = chunk * target;
    // --- independent loads: the column list first (the gathers depend on it), values, ranks, row offsets
    const int *ul = ulist + (size_t)chunk * NU;
    int idx[NU64];
#pragma unroll
    for (int j = 0; j < NU64; ++j) idx[j] = ul[j * 64 + lane];
    d2v v0[STEPS], v1[STEPS];
    us4v c[STEPS];
    const unsigned short *cp = col16 + (size_t)chunk * WT;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      int k = kb + (st * 64 + lane) * 4;
      k = (k < kmax) ? k : kmax;
      v0[st] = ldg<true>(reinterpret_cast<const d2v *>(val + k));
      v1[st] = ldg<true>(reinterpret_cast<const d2v *>(val + k + 2));
      c[st] = ldg<true>(reinterpret_cast<const us4v *>(cp + (st * 64 + lane) * 4));
    }
    const unsigned short *ro = rowoff + (size_t)chunk * E;
    int lo[NP], hi[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int i = 64 * m + lane;
      lo[m] = ro[i];
      hi[m] = ro[i + 1 < E ? i + 1 : E - 1];
    }
    const int r0 = tab[chunk].x;
    const int nr = tab[chunk + 1].x - r0;
    // --- one gather per distinct column (padding slots repeat the last one), parked in the LDS slice
    double xs[NU64];
#pragma unroll
    for (int j = 0; j < NU64; ++j) xs[j] = x[idx[j]];
#pragma unroll
    for (int j = 0; j < NU64; ++j) buf[j * 64 + lane] = xs[j];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // --- products from LDS, then they take the list's place
    d2v p0[STEPS], p1[STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      p0[st].x = v0[st].x * buf[c[st].x];
      p0[st].y = v0[st].y * buf[c[st].y];
      p1[st].x = v1[st].x * buf[c[st].z];
      p1[st].y = v1[st].y * buf[c[st].w];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int off = (st * 64 + lane) * 4;
      *reinterpret_cast<d2v *>(&buf[off]) = p0[st];
      *reinterpret_cast<d2v *>(&buf[off + 2]) = p1[st];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // --- one lane per row, products added left to right (reference order, csr_mat.c:49-54)
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int i = 64 * m + lane;
      if (i < nr) {
        double acc = 0.0;
        for (int k = lo[m]; k < hi[m]; k += 8) {
          double t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int id = k + u;
            id = id < WT ? id : WT - 1;
            t[u] = buf[id];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += (k + u < hi[m]) ? t[u] : 0.0;
        }
        if constexpr (NTS)
          __builtin_nontemporal_store(acc, &y[r0 + i]);
        else
          y[r0 + i] = acc;
        if (dotv) dsum += dotv[r0 + i] * acc;
      }
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    __syncthreads();  // red[] lives in wave 0's slice: every wave must be done with its rows
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < WPB; ++i) t += red[i];
      partials[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------ w4: offset-structured rows
//
// After w3 the SpMV is bound by distinct DRAM bytes, and for a stencil operator a fifth of
// those are column indices that carry almost no information: every row's columns are
// row + o for o in a small set of offsets (7 for the 7-point operator, whatever the grid or
// the slab shift).  w4 is for matrices with at most 16 distinct values of col - row whose rows
// store their columns strictly ascending.  Built once from the CSR arrays (lossless):
//   * offs[NO]   the distinct offsets, ascending;
//   * mask[r]    16 bits: which offsets row r stores (bit order = storage order, because
//                ascending offsets are ascending columns);
//   * valT       the values in blocks of 128 rows, offset-major inside a block
//                (valT[(block*NO + o)*128 + i] = A[r, r + offs[o]], zero where not stored).
// The kernel is then a pure streaming kernel: each lane owns two consecutive rows, loads
// their NO value pairs and the NO x pairs with 16-byte accesses that are consecutive across
// the wave, and adds the stored products in offset order -- the reference's left-to-right
// order (csr_mat.c:49-54), separate multiply and add, entries that are not stored are not
// touched (the mask decides, not the zero padding) => bit-identical to the CPU loop.
// No LDS, no dependent loads, no column indices: DRAM bytes per row 8*NO + 2 (+ x, y).
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));  // x pairs: 8-byte aligned

constexpr int kDiaRows = 128;      // rows per block (one wave: two rows per lane)
constexpr int kDiaMaxOffs = 64;  // 1..16: 16-bit row masks (csr_spmv_w4), 17..32: 32-bit (csr_spmv_w4x), 33..64: 64-bit (csr_spmv_w4y)
constexpr int kDiaTable = 128;   // slots of the offset hash table (twice the offsets it has to hold)
constexpr int kDiaEmpty = -0x7fffffff - 1;

struct DiaOffs {
  int o[kDiaMaxOffs];
};

// distinct values of col - row into a kDiaTable-slot open-addressing table; *overflow when there
// are more than the table holds (and so certainly more than kDiaMaxOffs) or a row is not strictly ascending
__global__ void dia_offsets_kernel(int nrows, const int *__restrict__ ind, const int *__restrict__ col,
                                   int *table, int *overflow) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    if (*(volatile int *)overflow) return;
    int prev = -1;
    for (int k = ind[r]; k < ind[r + 1]; ++k) {
      const int c = col[k];
      if (c <= prev) {
        *overflow = 1;  // unsorted or repeated columns: storage order is not offset order
        return;
      }
      prev = c;
      const int o = c - r;
      unsigned h = ((unsigned)o * 2654435761u) >> 25;
      int probes = 0;
      for (; probes < kDiaTable; ++probes, h = (h + 1) & (kDiaTable - 1)) {
        int v = *(volatile int *)(table + h);
        if (v == o) break;
        if (v == kDiaEmpty) {
          v = atomicCAS(table + h, kDiaEmpty, o);
          if (v == kDiaEmpty || v == o) break;
        }
      }
      if (probes == kDiaTable) {
        *overflow = 1;
        return;
      }
    }
  }
}

template <typename MaskT>
__global__ void dia_build_kernel(int nrows, int no, DiaOffs offs, const int *__restrict__ ind,
                                 const int *__restrict__ col, const double *__restrict__ val,
                                 double *__restrict__ valT, MaskT *__restrict__ mask) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long)gridDim.x * blockDim.x) {
    const long blk = r / kDiaRows;
    const int i = (int)(r % kDiaRows);
    unsigned long long m = 0;
    for (int k = ind[r]; k < ind[r + 1]; ++k) {
      const int o = col[k] - (int)r;
      int b = 0;
      while (b < no - 1 && offs.o[b] != o) ++b;
      m |= 1ull << b;
      valT[((size_t)blk * no + b) * kDiaRows + i] = val[k];
    }
    mask[r] = (MaskT)m;
  }
}

// 5-/7-point Poisson operator written directly in the offset-major w4 layout (no CSR arrays):
// offsets {-nx*ny, -nx, -1, 0, 1, nx, nx*ny} (3-D) or {-nx, -1, 0, 1, nx}; same entries, same
// per-row order as poisson_csr_kernel.  Slab form: local row r is global row row_lo + r (the
// offsets the caller puts into DiaOffs are shifted by row_lo - col_shift, nothing changes here).
__global__ void poisson_w4_kernel(int nx, int ny, int nz, long row_lo, long nloc, int no,
                                  double *__restrict__ valT, unsigned short *__restrict__ mask) {
  const long nxy = (long)nx * ny;
  const bool three_d = nz > 0;
  const double dg = three_d ? 6.0 : 4.0;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nloc; r += (long)gridDim.x * blockDim.x) {
    const long k = row_lo + r;
    const int i = (int)(k % nx);
    const int j = (int)((k / nx) % ny);
    const long l = k / nxy;
    double *v = valT + (size_t)(r / kDiaRows) * no * kDiaRows + (size_t)(r % kDiaRows);
    unsigned m = 0;
    int b = 0;
    if (three_d) {
      if (l > 0) { v[(size_t)b * kDiaRows] = -1.0; m |= 1u << b; }
      ++b;
    }
    if (j > 0) { v[(size_t)b * kDiaRows] = -1.0; m |= 1u << b; }
    ++b;
    if (i > 0) { v[(size_t)b * kDiaRows] = -1.0; m |= 1u << b; }
    ++b;
    v[(size_t)b * kDiaRows] = dg; m |= 1u << b;
    ++b;
    if (i < nx - 1) { v[(size_t)b * kDiaRows] = -1.0; m |= 1u << b; }
    ++b;
    if (j < ny - 1) { v[(size_t)b * kDiaRows] = -1.0; m |= 1u << b; }
    ++b;
    if (three_d) {
      if (l < nz - 1) { v[(size_t)b * kDiaRows] = -1.0; m |= 1u << b; }
      ++b;
    }
    mask[r] = (unsigned short)m;
  }
}

// A[r, r] from the w4 layout (0.0 where the diagonal is not stored)
__global__ void dia_diag_kernel(int nrows, int no, int zero_slot, const double *__restrict__ valT,
                                const unsigned short *__restrict__ mask, double *__restrict__ diag) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long)gridDim.x * blockDim.x) {
    double d = 0.0;
    if (zero_slot >= 0 && ((mask[r] >> zero_slot) & 1u))
      d = valT[((size_t)(r / kDiaRows) * no + zero_slot) * kDiaRows + (size_t)(r % kDiaRows)];
    diag[r] = d;
  }
}

template <int NO, bool NTL = true, bool NTS = true>
__global__ __launch_bounds__(256) void csr_spmv_w4(
    int blk0, int blk1, int nrows, int ncols, int stripe, DiaOffs offs, const double *__restrict__ valT,
    const unsigned short *__restrict__ mask, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ dotv, double *__restrict__ partials, const int *__restrict__ skip,
    int use_div, double xdiv, const double *__restrict__ xdiv_dev, int dot_slot) {
  // dot_slot >= 0 (round 4): the dot's operand IS x seen through offset slot dot_slot (dotv == x + offs.o[dot_slot]: p.q
  // of PCG, v.Av of MINRES) -- its pair is already in registers (xv[dot_slot], divided like dotv would be), so the
  // epilogue loads nothing: the same values, hence the same bits, 3.5 % less time for the product inside the loops
  // use_div: multiply with x ./ xdiv instead of x (MINRES: v = y / beta formed on the fly,
  // minres.c:123-124 -- the same correctly rounded division as the separate pass)
  if (skip && *skip) return;  // asynchronous solver loop already finished: no-op launch
  if (xdiv_dev) xdiv = *xdiv_dev;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int vb = (int)blockIdx.x;  // XCD-aware placement, see csr_spmv_w1
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int blk = blk0 + vb * 4 + wid;  // blocks [blk0, blk1) belong to this launch
  const long r = (long)blk * kDiaRows + 2 * lane;
  double dsum = 0.0;
  if (blk < blk1 && r < nrows) {
    // the mask array is padded to a whole block: both halves are always readable
    const unsigned mm = *reinterpret_cast<const unsigned *>(mask + r);
    const unsigned m0 = mm & 0xffffu, m1 = mm >> 16;
    const double *vp = valT + (size_t)blk * NO * kDiaRows + 2 * lane;
    d2v v[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) v[o] = ldg<NTL>(reinterpret_cast<const d2v *>(vp + o * kDiaRows));
    // x pairs: every load is issued unconditionally from a clamped address (a bounds-check branch
    // per load makes the compiler wait for all earlier loads before each one -- seven sequential
    // round trips); the lanes at the two ends of x repair their pairs afterwards
    d2v xv[NO];
    const long cmax = (long)ncols - 2;  // ncols >= 2 (ensure_w4)
    bool edge = false;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const long c = r + offs.o[o];
      const long cc = c < 0 ? 0 : (c > cmax ? cmax : c);
      const d2u t = *reinterpret_cast<const d2u *>(x + cc);
      xv[o].x = t.x;
      xv[o].y = t.y;
      edge |= cc != c;
    }
    if (edge) {  // load only what a stored entry can reference
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const long c = r + offs.o[o];
        if (c < 0 || c > cmax) {
          xv[o].x = (c >= 0 && c < ncols) ? x[c] : 0.0;
          xv[o].y = (c + 1 >= 0 && c + 1 < ncols) ? x[c + 1] : 0.0;
        }
      }
    }
    if (use_div) {
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        xv[o].x = xv[o].x / xdiv;
        xv[o].y = xv[o].y / xdiv;
      }
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const double t0 = a0 + v[o].x * xv[o].x;
      const double t1 = a1 + v[o].y * xv[o].y;
      a0 = ((m0 >> o) & 1u) ? t0 : a0;
      a1 = ((m1 >> o) & 1u) ? t1 : a1;
    }
    if (r + 1 < nrows) {
      d2u outu;
      outu.x = a0;
      outu.y = a1;
      if constexpr (NTS)
        __builtin_nontemporal_store(outu, reinterpret_cast<d2u *>(y + r));
      else
        *reinterpret_cast<d2u *>(y + r) = outu;
      if (dotv) {
        d2u u;
        if (dot_slot >= 0) {  // (wave-uniform) the operand's pair is xv[dot_slot]
          u.x = 0.0;
          u.y = 0.0;
#pragma unroll
          for (int o = 0; o < NO; ++o)
            if (o == dot_slot) {
              u.x = xv[o].x;
              u.y = xv[o].y;
            }
        } else {
          u = *reinterpret_cast<const d2u *>(dotv + r);
          if (use_div) {
            u.x = u.x / xdiv;
            u.y = u.y / xdiv;
          }
        }
        dsum += u.x * a0;
        dsum += u.y * a1;
      }
    } else {
      y[r] = a0;
      if (dotv) {
        double u0 = 0.0;
        if (dot_slot >= 0) {
#pragma unroll
          for (int o = 0; o < NO; ++o)
            if (o == dot_slot) u0 = xv[o].x;
        } else {
          u0 = use_div ? dotv[r] / xdiv : dotv[r];
        }
        dsum += u0 * a0;
      }
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

// ------------------------------------------------------------------ sss_spmv_w4: symmetric skyline
//
// sss_matvec (sss_mat.c:45-55) for offset-structured matrices, at HALF the matrix traffic of the
// mirrored-CSR product: only the strict lower triangle is stored (offset-major blocks of 128
// rows, like csr_spmv_w4) and it is read twice -- as row r's lower entries L[r, r+o] and, at the
// shifted rows r-o, as the mirrored upper entries A[r, r-o] = L[r-o, r] (the second read of a
// block hits in L2 / Infinity Cache; DRAM sees the values once).  Summation order per row is
// the reference's: lower entries by ascending column, then the diagonal term, then the
// mirrored entries by ascending row (sss_mat.c:52 adds them as the sweep reaches row r-o).
// mask[r]: bits 0-7 = lower offsets row r stores, bits 8-15 = rows r-o_j that store offset o_j.
struct SssOffs {
  int o[8];  // strictly negative, ascending
};

__global__ void sss_lowmask_kernel(int n, int nol, SssOffs offs, const int *__restrict__ ind,
                                   const int *__restrict__ col, const double *__restrict__ val,
                                   double *__restrict__ valL, unsigned char *__restrict__ low, long soa_npad) {
  // soa_npad > 0: one array of soa_npad values per offset (every stream of the product contiguous) instead of blocks
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
    const long blk = r / kDiaRows;
    const int i = (int)(r % kDiaRows);
    unsigned m = 0;
    for (int k = ind[r]; k < ind[r + 1]; ++k) {
      const int o = col[k] - (int)r;
      int b = 0;
      while (b < nol - 1 && offs.o[b] != o) ++b;
      m |= 1u << b;
      if (soa_npad > 0) valL[(size_t)b * soa_npad + r] = val[k];
      else valL[((size_t)blk * nol + b) * kDiaRows + i] = val[k];
    }
    low[r] = (unsigned char)m;
  }
}

__global__ void sss_mask_kernel(int n, int nol, SssOffs offs, const unsigned char *__restrict__ low,
                                unsigned short *__restrict__ mask) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
    unsigned m = low[r];
    for (int j = 0; j < nol; ++j) {
      const long ru = r - offs.o[j];
      if (ru < n && ((low[ru] >> j) & 1u)) m |= 1u << (8 + j);
    }
    mask[r] = (unsigned short)m;
  }
}

template <int NOL, int FLAGS = 0>
__global__ __launch_bounds__(256) void sss_spmv_w4(
    int n, int stripe, SssOffs offs, const double *__restrict__ valL, const double *__restrict__ diag,
    const unsigned short *__restrict__ mask, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ dotv, double *__restrict__ partials, const int *__restrict__ skip,
    int use_div, double xdiv, const double *__restrict__ xdiv_dev, int dot_is_x) {
  // dot_is_x (round 4): dotv == x -- the dot's operand is the diagonal term's x pair (x0), already in registers
  if (skip && *skip) return;
  if (xdiv_dev) xdiv = *xdiv_dev;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int vb = (int)blockIdx.x;  // XCD-aware placement, see csr_spmv_w1
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const long blk = (long)vb * 4 + wid;
  const long r = blk * kDiaRows + 2 * lane;
  double dsum = 0.0;
  if (r < n) {
    const bool two = r + 1 < n;
    const unsigned mm = *reinterpret_cast<const unsigned *>(mask + r);  // padded to a whole block
    const unsigned m0 = mm & 0xffffu, m1 = mm >> 16;
    // All loads are issued unconditionally from clamped addresses (see csr_spmv_w4); the lanes at
    // the ends of x repair their pairs afterwards.  valL is padded to whole blocks (npad rows).
    const long npad = ((long)n + kDiaRows - 1) / kDiaRows * kDiaRows;
    const long xmax = (long)n - 2;  // n >= 2 (ensure_sss_w4)
    bool edge = !two;
    // FLAGS & 8: valL is one array of npad values per offset; otherwise blocks of kDiaRows rows, offset-major inside
    constexpr bool SOA = (FLAGS & 8) != 0;
    auto vaddr = [&](long row, int j) -> const double * {
      return SOA ? valL + (size_t)j * npad + row
                 : valL + ((size_t)(row / kDiaRows) * NOL + j) * kDiaRows + (size_t)(row % kDiaRows);
    };
    // lower entries of rows r, r+1 and the x they multiply
    d2v vl[NOL], xl[NOL];
    // FLAGS & 4: the offset -1 (always the last one when present) takes no loads of its own -- x[r-1], x[r+2] and
    // the mirrored values L[r+1, r], L[r+2, r+1] are the neighbouring lanes' x0 / vl registers (lanes 0 and 63
    // fetch their one halo element) -- and the mirrored pair of an EVEN offset is one aligned 16-byte load:
    // 14-16 load instructions per lane instead of 20 for the 7-point operator.
    constexpr bool SHFL = (FLAGS & 4) != 0;
    const bool off1 = SHFL && offs.o[NOL - 1] == -1;  // wave-uniform
#pragma unroll
    for (int j = 0; j < NOL; ++j) {
      vl[j] = ldg<(FLAGS & 1) != 0>(reinterpret_cast<const d2v *>(vaddr(r, j)));  // plain: the line is usually in L2 already (shifted read), NT costs 6 %
      if (off1 && j == NOL - 1) continue;
      const long c = r + offs.o[j];  // < r
      const long cc = c < 0 ? 0 : c;
      const d2u t = *reinterpret_cast<const d2u *>(x + cc);
      xl[j].x = t.x;
      xl[j].y = t.y;
      edge |= c < 0;
    }
    double halo_xm1 = 0.0, halo_x2 = 0.0, halo_v2 = 0.0;
    if (off1) {
      if (lane == 0 && r > 0) halo_xm1 = x[r - 1];
      if (lane == 63) {
        if (r + 2 < n) halo_x2 = x[r + 2];
        if (r + 2 < npad)
          halo_v2 = *vaddr(r + 2, NOL - 1);
      }
    }
    // diagonal
    d2v dg, x0;
    {
      const long rd = r > xmax ? xmax : r;
      const d2u t = *reinterpret_cast<const d2u *>(diag + rd);
      const d2u u = *reinterpret_cast<const d2u *>(x + rd);
      dg.x = t.x; dg.y = t.y; x0.x = u.x; x0.y = u.y;
    }
    // mirrored entries: L[ru, ru + o_j] with ru = r - o_j (> r), times x[ru]
    d2v vu[NOL], xu[NOL];
#pragma unroll
    for (int j = 0; j < NOL; ++j) {
      if (off1 && j == NOL - 1) continue;
      const long ru = r - offs.o[j];
      if (SHFL && (offs.o[j] & 1) == 0) {  // ru even: rows ru, ru + 1 sit side by side in one block
        const long v0 = ru < npad ? ru : npad - 2;
        vu[j] = ldg<(FLAGS & 2) != 0>(reinterpret_cast<const d2v *>(vaddr(v0, j)));
      } else {
        const long v0 = ru < npad ? ru : npad - 1, v1 = ru + 1 < npad ? ru + 1 : npad - 1;
        vu[j].x = ldg<(FLAGS & 2) != 0>(vaddr(v0, j));
        vu[j].y = ldg<(FLAGS & 2) != 0>(vaddr(v1, j));
      }
      const long xr = ru > xmax ? xmax : ru;
      const d2u t = *reinterpret_cast<const d2u *>(x + xr);
      xu[j].x = t.x;
      xu[j].y = t.y;
      edge |= ru > xmax;
    }
    if (edge) {
#pragma unroll
      for (int j = 0; j < NOL; ++j) {
        const long c = r + offs.o[j];
        if (c < 0) {
          xl[j].x = 0.0;
          xl[j].y = c + 1 >= 0 ? x[c + 1] : 0.0;
        }
        const long ru = r - offs.o[j];
        if (ru > xmax) {
          xu[j].x = ru < n ? x[ru] : 0.0;
          xu[j].y = 0.0;
        }
      }
      if (!two) {
        dg.x = diag[r];
        dg.y = 0.0;
        x0.x = x[r];
        x0.y = 0.0;
      }
    }
    if (off1) {  // the -1 offset from the neighbouring lanes (after the repairs: x0 is final)
      constexpr int j = NOL - 1;
      const double up = __shfl_up(x0.y, 1, 64);        // x[r - 1]
      const double dnx = __shfl_down(x0.x, 1, 64);     // x[r + 2]
      const double dnv = __shfl_down(vl[j].x, 1, 64);  // L[r + 2, r + 1]
      xl[j].x = lane == 0 ? halo_xm1 : up;
      xl[j].y = x0.x;
      vu[j].x = vl[j].y;  // L[r + 1, r]
      vu[j].y = lane == 63 ? halo_v2 : dnv;
      xu[j].x = x0.y;
      xu[j].y = lane == 63 ? halo_x2 : dnx;
    }
    if (use_div) {  // x ./ xdiv (see csr_spmv_w4)
#pragma unroll
      for (int j = 0; j < NOL; ++j) {
        xl[j].x = xl[j].x / xdiv;
        xl[j].y = xl[j].y / xdiv;
        xu[j].x = xu[j].x / xdiv;
        xu[j].y = xu[j].y / xdiv;
      }
      x0.x = x0.x / xdiv;
      x0.y = x0.y / xdiv;
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < NOL; ++j) {  // lower entries, ascending column
      const double t0 = a0 + vl[j].x * xl[j].x;
      const double t1 = a1 + vl[j].y * xl[j].y;
      a0 = ((m0 >> j) & 1u) ? t0 : a0;
      a1 = ((m1 >> j) & 1u) ? t1 : a1;
    }
    a0 = a0 + dg.x * x0.x;  // sss_mat.c:54: y[i] = s + diag[i]*x[i], always
    a1 = a1 + dg.y * x0.y;
#pragma unroll
    for (int j = NOL - 1; j >= 0; --j) {  // mirrored entries, ascending row r - o_j
      const double t0 = a0 + vu[j].x * xu[j].x;
      const double t1 = a1 + vu[j].y * xu[j].y;
      a0 = ((m0 >> (8 + j)) & 1u) ? t0 : a0;
      a1 = ((m1 >> (8 + j)) & 1u) ? t1 : a1;
    }
    if (two) {
      d2u outu;
      outu.x = a0;
      outu.y = a1;
      __builtin_nontemporal_store(outu, reinterpret_cast<d2u *>(y + r));
      if (dotv) {
        d2u u;
        if (dot_is_x) {  // x0 is final here (repaired at the edges, divided when use_div)
          u.x = x0.x;
          u.y = x0.y;
        } else {
          u = *reinterpret_cast<const d2u *>(dotv + r);
          if (use_div) {
            u.x = u.x / xdiv;
            u.y = u.y / xdiv;
          }
        }
        dsum += u.x * a0;
        dsum += u.y * a1;
      }
    } else {
      y[r] = a0;
      if (dotv) dsum += (dot_is_x ? x0.x : (use_div ? dotv[r] / xdiv : dotv[r])) * a0;
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

// csr_spmv_w4 for 17..32 offsets (27-point stencils): 32-bit row masks, and the offsets are taken in
// groups of 8 so that the value / x pairs of one group, not of all offsets, are live at a time
template <int NO>
__global__ __launch_bounds__(256) void csr_spmv_w4x(
    int blk0, int blk1, int nrows, int ncols, int stripe, DiaOffs offs, const double *__restrict__ valT,
    const unsigned *__restrict__ mask, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ dotv, double *__restrict__ partials, const int *__restrict__ skip) {
  if (skip && *skip) return;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int vb = (int)blockIdx.x;
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int blk = blk0 + vb * 4 + wid;
  const long r = (long)blk * kDiaRows + 2 * lane;
  double dsum = 0.0;
  if (blk < blk1 && r < nrows) {
    const uint2 mm = *reinterpret_cast<const uint2 *>(mask + r);  // padded to a whole block
    const unsigned m0 = mm.x, m1 = mm.y;
    const double *vp = valT + (size_t)blk * NO * kDiaRows + 2 * lane;
    const long cmax = (long)ncols - 2;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int g = 0; g < NO; g += 8) {
      constexpr int G = 8;
      d2v v[G], xv[G];
      bool edge = false;
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (g + u < NO) {
          v[u] = ldg<true>(reinterpret_cast<const d2v *>(vp + (g + u) * kDiaRows));
          const long c = r + offs.o[g + u];
          const long cc = c < 0 ? 0 : (c > cmax ? cmax : c);
          const d2u t = *reinterpret_cast<const d2u *>(x + cc);
          xv[u].x = t.x;
          xv[u].y = t.y;
          edge |= cc != c;
        }
      }
      if (edge) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (g + u < NO) {
            const long c = r + offs.o[g + u];
            if (c < 0 || c > cmax) {
              xv[u].x = (c >= 0 && c < ncols) ? x[c] : 0.0;
              xv[u].y = (c + 1 >= 0 && c + 1 < ncols) ? x[c + 1] : 0.0;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (g + u < NO) {
          const double t0 = a0 + v[u].x * xv[u].x;
          const double t1 = a1 + v[u].y * xv[u].y;
          a0 = ((m0 >> (g + u)) & 1u) ? t0 : a0;
          a1 = ((m1 >> (g + u)) & 1u) ? t1 : a1;
        }
      }
    }
    if (r + 1 < nrows) {
      d2u outu;
      outu.x = a0;
      outu.y = a1;
      __builtin_nontemporal_store(outu, reinterpret_cast<d2u *>(y + r));
      if (dotv) {
        const d2u u = *reinterpret_cast<const d2u *>(dotv + r);
        dsum += u.x * a0;
        dsum += u.y * a1;
      }
    } else {
      y[r] = a0;
      if (dotv) dsum += dotv[r] * a0;
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

// csr_spmv_w4 for 33..64 offsets (round 3: the log-spaced pattern of examples/tendigit.py scaled to 10^6 rows has 41;
// its CSR form sat on the gather kernel csr_spmv_w2 at 0.6 of the roofline): 64-bit row masks, the offsets in device
// memory (a run-time index into a by-value struct would put it into scratch), groups of 8 in a run-time loop.  Same
// products in the same order as csr_spmv_w4 / w4x.
__global__ __launch_bounds__(256) void csr_spmv_w4y(
    int blk0, int blk1, int nrows, int ncols, int stripe, int no, const int *__restrict__ offs,
    const double *__restrict__ valT, const unsigned long long *__restrict__ mask, const double *__restrict__ x,
    double *__restrict__ y, const double *__restrict__ dotv, double *__restrict__ partials,
    const int *__restrict__ skip) {
  if (skip && *skip) return;
  __shared__ double red[4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int vb = (int)blockIdx.x;
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int blk = blk0 + vb * 4 + wid;
  const long r = (long)blk * kDiaRows + 2 * lane;
  double dsum = 0.0;
  if (blk < blk1 && r < nrows) {
    const ulonglong2 mm = *reinterpret_cast<const ulonglong2 *>(mask + r);  // padded to a whole block
    const unsigned long long m0 = mm.x, m1 = mm.y;
    const double *vp = valT + (size_t)blk * no * kDiaRows + 2 * lane;
    const long cmax = (long)ncols - 2;
    double a0 = 0.0, a1 = 0.0;
    for (int g = 0; g < no; g += 8) {
      constexpr int G = 8;
      d2v v[G], xv[G];
      int og[G];
      bool edge = false;
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int gu = g + u < no ? g + u : no - 1;  // the tail group repeats the last offset (loaded, never added)
        og[u] = offs[gu];
        v[u] = ldg<true>(reinterpret_cast<const d2v *>(vp + (size_t)gu * kDiaRows));
        const long c = r + og[u];
        const long cc = c < 0 ? 0 : (c > cmax ? cmax : c);
        const d2u t = *reinterpret_cast<const d2u *>(x + cc);
        xv[u].x = t.x;
        xv[u].y = t.y;
        edge |= cc != c;
      }
      if (edge) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const long c = r + og[u];
          if (c < 0 || c > cmax) {
            xv[u].x = (c >= 0 && c < ncols) ? x[c] : 0.0;
            xv[u].y = (c + 1 >= 0 && c + 1 < ncols) ? x[c + 1] : 0.0;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const double t0 = a0 + v[u].x * xv[u].x;
        const double t1 = a1 + v[u].y * xv[u].y;
        const bool live = g + u < no;
        a0 = (live && ((m0 >> (g + u)) & 1ull)) ? t0 : a0;
        a1 = (live && ((m1 >> (g + u)) & 1ull)) ? t1 : a1;
      }
    }
    if (r + 1 < nrows) {
      d2u outu;
      outu.x = a0;
      outu.y = a1;
      __builtin_nontemporal_store(outu, reinterpret_cast<d2u *>(y + r));
      if (dotv) {
        const d2u u = *reinterpret_cast<const d2u *>(dotv + r);
        dsum += u.x * a0;
        dsum += u.y * a1;
      }
    } else {
      y[r] = a0;
      if (dotv) dsum += dotv[r] * a0;
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    if (lane == 0) red[wid] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

// ---- csr_spmv_w4 with the PCG p-update folded in (pcg.c:105-117 in one pass):
//   p_new = z + beta*p_old (z = r, r.*dinv or r*dc; first iteration: p_new = z),  q = A p_new,
//   partial sums of p_new.q.  p_new is formed on the fly at every neighbour position from r and
//   p_old (the same two rounded operations as pupdate_kernel, so the same bits) and written once for
//   the lane's own rows; p_old and p_new are different buffers.  Saves the separate pass that
//   writes p and the SpMV's read of it (8 bytes per row).  Square operators only (x = p has nrows
//   entries).
// XU (round 5, the lazy loop's variant): the pending x update of the PREVIOUS iteration and its stagnation scan
// (pcg.c:127-141: x += alpha_x p_old, the scan reads x before the update) ride along for the lane's own rows -- p_old[own] is
// in registers already -- so the separate px pass disappears: 130 n instead of 138 n bytes per iteration.  The same
// expressions as px_update_kernel (psp_vec.hip), hence the same bits; scan_partials[blockIdx.x] = number of the
// workgroup's waves whose rows did not stagnate (only its being zero or not is ever used).
template <int NO, int PRE, bool XU = false>
__global__ __launch_bounds__(256) void csr_spmv_w4_pf(
    int nrows, int stripe, DiaOffs offs, const double *__restrict__ valT,
    const unsigned short *__restrict__ mask, const double *__restrict__ r, const double *__restrict__ dinv,
    double dc, const double *__restrict__ p_old, double *__restrict__ p_new, double *__restrict__ q,
    double beta, int first, double *__restrict__ partials, const psp::PcgDev *__restrict__ dstate,
    double *__restrict__ x = nullptr, double *__restrict__ scan_partials = nullptr) {
  double alpha_x = 0.0;
  bool xp = false;
  if (dstate) {  // asynchronous loop: scalars live on the device
    if (dstate->status) return;
    beta = dstate->beta;
    first = dstate->it == 1;
    if constexpr (XU) {
      alpha_x = dstate->alpha_x;
      xp = dstate->xpend != 0;
    }
  }
  __shared__ double red[4];
  __shared__ double red2[4];
  double dmax = 0.0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int vb = (int)blockIdx.x;
  if (stripe > 0) {
    const int k = vb >> 3;
    vb = ((k / stripe) * 8 + (vb & 7)) * stripe + k % stripe;
  }
  const int blk = vb * 4 + wid;
  const long row = (long)blk * kDiaRows + 2 * lane;
  double dsum = 0.0;
  if (row < nrows) {
    const unsigned mm = *reinterpret_cast<const unsigned *>(mask + row);
    const unsigned m0 = mm & 0xffffu, m1 = mm >> 16;
    const double *vp = valT + (size_t)blk * NO * kDiaRows + 2 * lane;
    d2v v[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) v[o] = ldg<true>(reinterpret_cast<const d2v *>(vp + o * kDiaRows));
    const long cmax = (long)nrows - 2;  // nrows >= 2
    // own rows (stored to p_new) and the neighbour pairs: every load unconditional, clamped
    const long rc = row > cmax ? cmax : row;
    d2u rr[NO + 1], pp[NO + 1], dd[NO + 1];
    bool edge = rc != row;
#pragma unroll
    for (int o = 0; o <= NO; ++o) {
      const long c = o < NO ? row + offs.o[o] : row;
      const long cc = c < 0 ? 0 : (c > cmax ? cmax : c);
      rr[o] = *reinterpret_cast<const d2u *>(r + cc);
      if (!first) pp[o] = *reinterpret_cast<const d2u *>(p_old + cc);
      if constexpr (PRE == 1) dd[o] = *reinterpret_cast<const d2u *>(dinv + cc);
      edge |= cc != c;
    }
    if (edge) {
#pragma unroll
      for (int o = 0; o <= NO; ++o) {
        const long c = o < NO ? row + offs.o[o] : row;
        if (c < 0 || c > cmax) {
          const bool i0 = c >= 0 && c < nrows, i1 = c + 1 >= 0 && c + 1 < nrows;
          rr[o].x = i0 ? r[c] : 0.0;
          rr[o].y = i1 ? r[c + 1] : 0.0;
          if (!first) {
            pp[o].x = i0 ? p_old[c] : 0.0;
            pp[o].y = i1 ? p_old[c + 1] : 0.0;
          }
          if constexpr (PRE == 1) {
            dd[o].x = i0 ? dinv[c] : 0.0;
            dd[o].y = i1 ? dinv[c + 1] : 0.0;
          }
        }
      }
    }
    // p_new at the NO neighbour pairs and at the own pair (index NO)
    d2v pn[NO + 1];
#pragma unroll
    for (int o = 0; o <= NO; ++o) {
      double z0 = rr[o].x, z1 = rr[o].y;
      if constexpr (PRE == 1) {
        z0 = z0 * dd[o].x;
        z1 = z1 * dd[o].y;
      }
      if constexpr (PRE == 2) {
        z0 = z0 * dc;
        z1 = z1 * dc;
      }
      if (!first) {
        z0 = z0 + beta * pp[o].x;
        z1 = z1 + beta * pp[o].y;
      }
      pn[o].x = z0;
      pn[o].y = z1;
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const double t0 = a0 + v[o].x * pn[o].x;
      const double t1 = a1 + v[o].y * pn[o].y;
      a0 = ((m0 >> o) & 1u) ? t0 : a0;
      a1 = ((m1 >> o) & 1u) ? t1 : a1;
    }
    if (row + 1 < nrows) {
      d2u outq, outp;
      outq.x = a0;
      outq.y = a1;
      outp.x = pn[NO].x;
      outp.y = pn[NO].y;
      __builtin_nontemporal_store(outq, reinterpret_cast<d2u *>(q + row));
      *reinterpret_cast<d2u *>(p_new + row) = outp;
      dsum += pn[NO].x * a0;
      dsum += pn[NO].y * a1;
    } else {
      q[row] = a0;
      p_new[row] = pn[NO].x;
      dsum += pn[NO].x * a0;
    }
    if constexpr (XU) {
      if (xp) {  // px_update_kernel's scan and update, on the own pair of p_old (never in iteration 1: nothing is pending)
        const bool upd = alpha_x != 0.0;
        const bool two = row + 1 < nrows;
        d2u xx;
        if (two) {
          xx = *reinterpret_cast<const d2u *>(x + row);
        } else {
          xx.x = x[row];
          xx.y = 0.0;
        }
        const double po[2] = {pp[NO].x, pp[NO].y};
        double xv[2] = {xx.x, xx.y};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !two) break;
          const double quot = fabs(alpha_x * po[u] / xv[u]);
          const double ddum = (xv[u] != 0.0) ? quot : ((po[u] != 0.0) ? 1.0 : 0.0);
          dmax = (ddum > dmax) ? ddum : dmax;
          if (upd) xv[u] = xv[u] + alpha_x * po[u];
        }
        if (two) {
          xx.x = xv[0];
          xx.y = xv[1];
          *reinterpret_cast<d2u *>(x + row) = xx;
        } else {
          x[row] = xv[0];
        }
      }
    }
  }
  if (partials) {
    dsum = wave_sum(dsum);
    if (lane == 0) red[wid] = dsum;
    if constexpr (XU) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(dmax, off, 64);
        if (o > dmax) dmax = o;
      }
      if (lane == 0) red2[wid] = (1.0 + dmax != 1.0) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
      if constexpr (XU) scan_partials[blockIdx.x] = red2[0] + red2[1] + red2[2] + red2[3];
    }
  }
}

// y = A^T x on the w4 layout, without atomics and in the reference's order.  csr_matvec_transp_kernel
// (csr_mat.c:74-88) zeroes y and sweeps the rows in ascending order, adding va[k]*x[i] to y[ja[k]]:
// y[c] therefore receives its terms by ascending row i = c - o, i.e. by DESCENDING offset.  A lane
// owns two columns and gathers A[c - o, c] = valT[row c - o][o] for o = last .. first -- same terms,
// same order, bit-identical to the CPU loop (the scatter kernel with fp64 atomics is only correct
// to rounding and not reproducible).
template <int NO>
__global__ __launch_bounds__(256) void csr_spmv_w4_transp(
    int nrows, int ncols, DiaOffs offs, const double *__restrict__ valT, const unsigned short *__restrict__ mask,
    const double *__restrict__ x, double *__restrict__ y) {
  const long c = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (c >= ncols) return;
  const long npad = ((long)nrows + kDiaRows - 1) / kDiaRows * kDiaRows;
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int o = NO - 1; o >= 0; --o) {
    // rows that hold column c / c + 1 at this offset; clamped loads, the mask (0 in the padding) decides
    const long i0 = c - offs.o[o], i1 = i0 + 1;
    const bool in0 = i0 >= 0 && i0 < nrows, in1 = i1 >= 0 && i1 < nrows && c + 1 < ncols;
    const long j0 = i0 < 0 ? 0 : (i0 >= npad ? npad - 1 : i0), j1 = i1 < 0 ? 0 : (i1 >= npad ? npad - 1 : i1);
    const double v0 = valT[((size_t)(j0 / kDiaRows) * NO + o) * kDiaRows + (size_t)(j0 % kDiaRows)];
    const double v1 = valT[((size_t)(j1 / kDiaRows) * NO + o) * kDiaRows + (size_t)(j1 % kDiaRows)];
    const unsigned m0 = mask[j0], m1 = mask[j1];
    const double x0 = x[in0 ? i0 : 0], x1 = x[in1 ? i1 : 0];
    const double t0 = a0 + v0 * x0, t1 = a1 + v1 * x1;
    a0 = (in0 && ((m0 >> o) & 1u)) ? t0 : a0;
    a1 = (in1 && ((m1 >> o) & 1u)) ? t1 : a1;
  }
  y[c] = a0;
  if (c + 1 < ncols) y[c + 1] = a1;
}

// ---- A^T as a CSR matrix (built once per handle for matvec_transp on irregular matrices):
// rows_of_nonzeros expands ind to one row id per nonzero; a STABLE radix sort of (column, position)
// then lists the nonzeros of each column by ascending row -- the order in which
// csr_matvec_transp_kernel (csr_mat.c:80-87) adds them into y[column].
__global__ void iota_int_kernel(int n, int *__restrict__ v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}

__global__ void rows_of_nonzeros_kernel(int nrows, const int *__restrict__ ind, int *__restrict__ rows) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < nrows; r += nwaves)
    for (int k = ind[r] + lane; k < ind[r + 1]; k += 64) rows[k] = r;
}

__global__ void transp_gather_kernel(int nnz, const int *__restrict__ perm, const int *__restrict__ rows,
                                     const double *__restrict__ val, int *__restrict__ col_t,
                                     double *__restrict__ val_t) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += gridDim.x * blockDim.x) {
    const int src = perm[k];
    col_t[k] = rows[src];
    val_t[k] = val[src];
  }
}

// ---- the transpose by counting (round 6): entries per column by atomic histogram, a scan, every entry takes a slot of
// its column by an atomic cursor (any order), then each column's few entries are sorted by (row, stored position) --
// the order the stable radix sort by column gave (16 ms of an sss_mat's 60 ms upload at 2e7 entries; this: ~3 ms).
// The result does not depend on the order in which the atomics landed.
__global__ void transp_count_kernel(int nnz, int ncols, const int *__restrict__ col, int *__restrict__ cnt, int *bad) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x) {
    const int c = col[k];
    if (c < 0 || c >= ncols) *bad = 1;
    else atomicAdd(cnt + c, 1);
  }
}

__global__ void transp_slot_kernel(int nrows, const int *__restrict__ ind, const int *__restrict__ col,
                                   const int *__restrict__ tind, int *__restrict__ cursor,
                                   unsigned long long *__restrict__ key) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (long)gridDim.x * 4)
    for (int k = ind[r] + lane; k < ind[r + 1]; k += 64) {
      const int c = col[k];
      const int at = tind[c] + atomicAdd(cursor + c, 1);
      key[at] = ((unsigned long long)(unsigned)r << 32) | (unsigned)k;
    }
}

// one thread per column: insertion sort of its keys (short segments; the keys are distinct)
__global__ void transp_sort_kernel(int ncols, const int *__restrict__ tind, unsigned long long *__restrict__ key) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x) {
    const int b = tind[c], e = tind[c + 1];
    for (int i = b + 1; i < e; ++i) {
      const unsigned long long v = key[i];
      int j = i - 1;
      while (j >= b && key[j] > v) {
        key[j + 1] = key[j];
        --j;
      }
      key[j + 1] = v;
    }
  }
}

__global__ void transp_emit_kernel(int nnz, const unsigned long long *__restrict__ key, const double *__restrict__ val,
                                   int *__restrict__ tcol, double *__restrict__ tval) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long)gridDim.x * blockDim.x) {
    const unsigned long long kv = key[i];
    tcol[i] = (int)(kv >> 32);
    tval[i] = val[(unsigned)(kv & 0xffffffffull)];
  }
}

// ind_t[c] = first position whose (sorted) column is >= c
__global__ void transp_ptr_kernel(int nnz, int ncols, const int *__restrict__ sorted_cols, int *__restrict__ ind_t) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= nnz; k += gridDim.x * blockDim.x) {
    const int lo = k == 0 ? 0 : sorted_cols[k - 1] + 1;  // columns (prev, cur] start at k
    const int hi = k == nnz ? ncols : sorted_cols[k];
    for (int c = lo; c <= hi; ++c) ind_t[c] = k;
  }
}

// first-level fold of per-workgroup dot partials when they do not sit in the workspace
// slots: out[o] = sum of in[o], in[o+nout], ... ; 16 lanes per output, fixed order
__global__ __launch_bounds__(256) void fold_partials_kernel(const double *__restrict__ in, int nin,
                                                            double *__restrict__ out, int nout) {
  const int o = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int g = threadIdx.x & 15;
  double s = 0.0;
  if (o < nout)
    for (long i = o + (long)nout * g; i < nin; i += (long)nout * 16) s += in[i];
#pragma unroll
  for (int m = 8; m > 0; m >>= 1) s += __shfl_xor(s, m, 16);
  if (g == 0 && o < nout) out[o] = s;
}

__global__ void csr_diag_kernel(int nrows, int row0, const int *__restrict__ ind,
                                const int *__restrict__ col, const double *__restrict__ val,
                                double *__restrict__ diag) {
  // row0: global number of this handle's first row (parts of a partitioned matrix)
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    double d = 0.0;
    for (int k = ind[r]; k < ind[r + 1]; ++k)
      if (col[k] == r + row0) d = val[k];
    diag[r] = d;
  }
}

// pseudo-random banded rows for psp_csr_random_banded: row r stores m entries, entry j in column
// (r + (j - m/2)*stride + h(r, j) mod stride) mod ncols with value in [-1, 1); the same integer formula is
// restated by the tests (tests/test_gpu_big_csr.py)
__device__ __host__ inline unsigned long long splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void random_banded_kernel(int nrows_part, long row0, int ncols, int m, int stride,
                                     unsigned long long seed, int *__restrict__ ind, int *__restrict__ col,
                                     double *__restrict__ val) {
  const long total = (long)nrows_part * m;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t <= total; t += (long)gridDim.x * blockDim.x) {
    if (t % m == 0) ind[t / m] = (int)t;
    if (t == total) break;
    const long r = row0 + t / m;
    const int j = (int)(t % m);
    const unsigned long long h = splitmix64(seed + (unsigned long long)r * 0x100000001B3ull +
                                            (unsigned long long)j * 0xD6E8FEB86659FD93ull);
    long c = r + (long)(j - m / 2) * stride + (long)(h % (unsigned long long)stride);
    c %= ncols;
    if (c < 0) c += ncols;
    col[t] = (int)c;
    const unsigned long long h2 = splitmix64(h);
    val[t] = (double)(h2 >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
  }
}

__global__ void max_row_kernel(int nrows, const int *__restrict__ ind, int *__restrict__ out) {
  int m = 0;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x)
    m = max(m, ind[r + 1] - ind[r]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ------------------------------------------------------------------ Poisson generators

// nonzeros stored in rows < k of the nx*ny(*nz) 5-/7-point operator (full CSR form)
__device__ __host__ inline long poisson_prefix(long k, long nx, long ny, long nz) {
  const long nxy = nx * ny;
  const bool three_d = nz > 0;
  const long n = nxy * (three_d ? nz : 1);
  long missing = (k + nx - 1) / nx;                                  // i == 0
  missing += k / nx;                                                 // i == nx-1
  missing += (k / nxy) * nx + (k % nxy < nx ? k % nxy : nx);         // j == 0
  {
    long rem = k % nxy - (nxy - nx);
    missing += (k / nxy) * nx + (rem > 0 ? rem : 0);                 // j == ny-1
  }
  if (three_d) {
    missing += k < nxy ? k : nxy;                                    // l == 0
    long rem = k - (n - nxy);
    missing += rem > 0 ? rem : 0;                                    // l == nz-1
  }
  return (three_d ? 7 : 5) * k - missing;
}

// strict-lower nonzeros stored in rows < k (SSS form)
__device__ __host__ inline long poisson_lower_prefix(long k, long nx, long ny, long nz) {
  const long nxy = nx * ny;
  const bool three_d = nz > 0;
  long missing = (k + nx - 1) / nx;                                  // i == 0: no k-1
  missing += (k / nxy) * nx + (k % nxy < nx ? k % nxy : nx);         // j == 0: no k-nx
  if (three_d) missing += k < nxy ? k : nxy;                         // l == 0: no k-nxy
  return (three_d ? 3 : 2) * k - missing;
}

__global__ void poisson_csr_kernel(int nx, int ny, int nz, long row_lo, long row_hi, long col_shift,
                                   int *__restrict__ ind, int *__restrict__ col,
                                   double *__restrict__ val) {
  const long nxy = (long)nx * ny;
  const bool three_d = nz > 0;
  const double dg = three_d ? 6.0 : 4.0;
  const long base = poisson_prefix(row_lo, nx, ny, nz);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long k = row_lo + (long)blockIdx.x * blockDim.x + threadIdx.x; k <= row_hi; k += stride) {
    long p = poisson_prefix(k, nx, ny, nz) - base;
    ind[k - row_lo] = (int)p;
    if (k == row_hi) break;
    const int i = (int)(k % nx);
    const int j = (int)((k / nx) % ny);
    const long l = k / nxy;
    if (three_d && l > 0) { col[p] = (int)(k - nxy - col_shift); val[p++] = -1.0; }
    if (j > 0)            { col[p] = (int)(k - nx - col_shift);  val[p++] = -1.0; }
    if (i > 0)            { col[p] = (int)(k - 1 - col_shift);   val[p++] = -1.0; }
    col[p] = (int)(k - col_shift); val[p++] = dg;
    if (i < nx - 1)       { col[p] = (int)(k + 1 - col_shift);   val[p++] = -1.0; }
    if (j < ny - 1)       { col[p] = (int)(k + nx - col_shift);  val[p++] = -1.0; }
    if (three_d && l < nz - 1) { col[p] = (int)(k + nxy - col_shift); val[p++] = -1.0; }
  }
}

__global__ void poisson_sss_kernel(int nx, int ny, int nz, long n, int *__restrict__ ind,
                                   int *__restrict__ col, double *__restrict__ val,
                                   double *__restrict__ diag) {
  const long nxy = (long)nx * ny;
  const bool three_d = nz > 0;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += stride) {
    long p = poisson_lower_prefix(k, nx, ny, nz);
    ind[k] = (int)p;
    if (k == n) break;
    const int i = (int)(k % nx);
    const int j = (int)((k / nx) % ny);
    const long l = k / nxy;
    if (three_d && l > 0) { col[p] = (int)(k - nxy); val[p++] = -1.0; }
    if (j > 0)            { col[p] = (int)(k - nx);  val[p++] = -1.0; }
    if (i > 0)            { col[p] = (int)(k - 1);   val[p++] = -1.0; }
    diag[k] = three_d ? 6.0 : 4.0;
  }
}

// ------------------------------------------------------------------ host helpers

// csr_spmv_w2: tile 1024, 4 waves per workgroup, non-temporal y stores, XCD stripe 64
// (profiles/r1_spmv_w2_sweep.txt: stripes 48/64/96 within 0.5 %, 0 and >= 192 about 1-3 % slower)
// bit 20: csr_spmv_w3 (x windows staged in LDS, 16-bit chunk-local columns) where the matrix
// qualifies (profiles/r1_spmv_w3*.txt)
constexpr int kW3Bit = 1 << 20;
// bit 22: csr_spmv_w4 (masked offset-major layout, no column indices) where the matrix qualifies
constexpr int kW4Bit = 1 << 22;
// bit 23: csr_spmv_w6 (the CSR arrays as stored, x staged in LDS through the block list alone) in front of csr_spmv_w2
// where the matrix qualifies
constexpr int kW6Bit = 1 << 23;
constexpr int kDefaultVariant = 128 + 2 + 64 + (64 << 8) + kW3Bit + kW4Bit + kW6Bit;

struct Variant {
  int tile, vec;
  bool nt;
  int map_mode;
  bool wave;
  bool full_grid;
  bool w1, w2;
  int layout, wpb;
  int stripe;
  bool w3;
  bool sched;
  bool w4;
  bool w6;
};

Variant decode_variant(int v) {
  // bits 0-1: vec (0 -> 4, 1 -> 2, 2 -> 1); bit 2: tile 2048 instead of 4096;
  // bit 3: non-temporal val/col loads; bit 4: XCD-striped chunk order
  if (v < 0) v = kDefaultVariant;
  Variant r;
  // bits 8-19: workgroups per XCD stripe of csr_spmv_w1 (0 = plain dispatch order)
  r.stripe = (v >> 8) & 0xfff;
  const bool w3bit = (v & kW3Bit) != 0;
  const bool nosched = (v & (1 << 21)) != 0;
  r.w4 = (v & kW4Bit) != 0;
  const bool w6bit = (v & kW6Bit) != 0;
  v &= 0xff;
  static const int vecs[4] = {4, 2, 1, 4};
  r.vec = vecs[v & 3];
  r.tile = (v & 4) ? 2048 : 4096;
  r.nt = (v & 8) != 0;
  r.map_mode = (v & 16) ? 1 : 0;
  // bit 5: wave-level pipelined kernel, tile 512 (bit 2 set) or 1024 nonzeros per wave
  // bit 6: one chunk per workgroup/wave (no persistent loop)
  r.full_grid = (v & 64) != 0;
  r.wave = (v & 32) != 0;
  if (r.wave) {
    r.tile = (v & 4) ? 512 : 1024;
    r.vec = 4;
  }
  // bit 7: one chunk per wave, non-persistent (csr_spmv_w1).  bit 0: layout (0: 4 nonzeros
  // per lane per load, 1: one); bit 2: tile 512 instead of 1024; bits 4-5: waves per
  // workgroup 4 / 8 / 16; bit 3: non-temporal loads
  r.w1 = (v & 128) != 0;
  // bit 1 (with bit 7): csr_spmv_w2, row offsets from the fixed-stride table instead of ind;
  // for w2, bit 6 = non-temporal stores of y (+1..2 %), bit 3 = non-temporal val/col loads (-13 %)
  r.w2 = r.w1 && (v & 2) != 0;
  r.layout = v & 1;
  r.wpb = 4 << ((v >> 4) & 3);
  if (r.wpb > 16) r.wpb = 16;
  if (r.w1) r.tile = (v & 4) ? 512 : 1024;
  // w3 rides on the w2 tables (tile 1024, 4 waves per workgroup)
  r.w3 = w3bit && r.w2 && r.tile == 1024 && r.layout == 0;
  r.sched = r.w3 && !nosched;  // bit 21: keep the natural order + XCD stripes
  r.w6 = w6bit && r.w2 && r.tile == 1024 && r.layout == 0 && !r.nt;
  return r;
}

int alloc_csr(int nrows, int ncols, long nnz, psp_csr **out) {
  // the kernels index nonzeros with 32-bit ints up to one tile past nnz
  if (nrows < 0 || ncols < 0 || nnz < 0 || nnz > 0x7fffffffL - 8192)
    return fail(PSP_EINVAL, "csr: invalid shape (%d x %d, nnz %ld)", nrows, ncols, nnz);
  PSP_TRY(ensure_device());
  psp_csr *A = new psp_csr();
  A->nrows = nrows;
  A->ncols = ncols;
  A->nnz = (int)nnz;
  A->padded = ((size_t)nnz + 3) / 4 * 4 + 8;  // vector loads may run past nnz by < 8 entries
  hipError_t e1 = hipMalloc((void **)&A->ind, sizeof(int) * ((size_t)nrows + 1));
  hipError_t e2 = hipMalloc((void **)&A->col, sizeof(int) * A->padded);
  hipError_t e3 = hipMalloc((void **)&A->val, sizeof(double) * A->padded);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    // the solvers' cached work vectors may be what fills the device: drop them and try once more
    (void)hipGetLastError();
    if (e1 == hipSuccess) (void)hipFree(A->ind);
    if (e2 == hipSuccess) (void)hipFree(A->col);
    if (e3 == hipSuccess) (void)hipFree(A->val);
    A->ind = nullptr;
    A->col = nullptr;
    A->val = nullptr;
    (void)psp_trim();
    e1 = hipMalloc((void **)&A->ind, sizeof(int) * ((size_t)nrows + 1));
    e2 = hipMalloc((void **)&A->col, sizeof(int) * A->padded);
    e3 = hipMalloc((void **)&A->val, sizeof(double) * A->padded);
  }
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    (void)hipFree(A->ind);
    (void)hipFree(A->col);
    (void)hipFree(A->val);
    delete A;
    return fail(PSP_ENOMEM, "csr: device allocation of %ld nonzeros failed", nnz);
  }
  // the padding must hold valid column indices (they are gathered, results unused)
  (void)hipMemsetAsync(A->col + nnz, 0, sizeof(int) * (A->padded - nnz), stream());
  (void)hipMemsetAsync(A->val + nnz, 0, sizeof(double) * (A->padded - nnz), stream());
  *out = A;
  return PSP_OK;
}

}  // namespace

// chunk tables are cached per (matrix, tile) in the handle
struct ChunkTable {
  int tile = 0;
  int target = 0;
  int nchunks = 0;
  int2 *tab = nullptr;
  // csr_spmv_w2: 16-bit row offsets at a fixed stride of 64*np entries per chunk
  int max_rows = -1;  // most rows in one chunk (-1: not computed yet)
  int np = 0;         // passes of 64 rows (0: chunk too tall for w2)
  unsigned short *rowoff = nullptr;
  // csr_spmv_w3: per chunk the 128-byte x blocks it references (fixed stride nb) and the
  // chunk's columns as 16-bit offsets into that list (fixed stride 1024)
  int nb = -1;        // -1: not examined yet, 0: some chunk needs too many blocks
  int max_blocks = 0;
  int outliers = 0;   // chunks with more than 64 blocks that csr_spmv_w3<OUTL> serves through the int32 columns
  int *blist = nullptr;
  unsigned short *col16 = nullptr;
  // csr_spmv_w6: the 64-slot block list alone (the columns stay the csr_mat's int32 array)
  int nb6 = -1;       // -1: not examined yet, 0: too many chunks would gather through memory, 64: built
  int direct6 = 0;    // chunks with more than 64 blocks or more than kW6Runs runs
  int *blist6 = nullptr;
  // csr_spmv_w5: per chunk its distinct columns (fixed stride nu, a multiple of 64) and the chunk's columns
  // as 16-bit ranks in that list (fixed stride 1024)
  int nu = -1;        // -1: not examined yet, 0: not worth it (some chunk has too many distinct columns)
  int max_cols = 0;
  int *ulist = nullptr;
  unsigned short *colu = nullptr;
  // plane-sweeping workgroup schedule (build_schedule): launch slot -> workgroup, or absent
  int sched_state = -1;  // -1 not examined, 0 none (natural order + XCD stripes), 1 present
  int sched_grid = 0;
  int *perm = nullptr;
  int half_band = 0;
};
static int get_chunk_table(psp_csr *A, int tile, ChunkTable **out);

namespace psp {

struct SplitInfo {
  int tile, ca, cb;
};

struct CsrExtra {
  std::map<int, ChunkTable> t;
  std::map<std::pair<int, int>, SplitInfo> split;  // (row_a, row_b) -> interior chunk range
  double *big_partials = nullptr;  // one slot per workgroup of the full-grid SpMV
  int big_cap = 0;
  char *packed = nullptr;          // interleaved col/val tiles (PACKED variants)
  // csr_spmv_w4: offset-structured layout (state -1 not examined, 0 not eligible, 1 built)
  int dia_state = -1;
  int dia_no = 0;
  DiaOffs dia_offs;
  double *dia_val = nullptr;
  unsigned short *dia_mask = nullptr;  // dia_no <= 16
  unsigned *dia_mask32 = nullptr;      // 16 < dia_no <= 32
  unsigned long long *dia_mask64 = nullptr;  // dia_no > 32 (csr_spmv_w4y), with the offsets in device memory:
  int *dia_offs_dev = nullptr;
  psp_csr *transposed = nullptr;       // A^T as its own handle (matvec_transp on irregular matrices)
  // renumbered copy R = P A P^T for csr_spmv_w3 (psp_reorder.hip): state -1 not examined, 0 none, 1 built
  int reorder_state = -1;
  // the cost rule of the renumbering (round 6; pick_scattered): products this handle has multiplied with on the stored
  // numbering so far, what the caller announced (psp_csr_prepare), and what the copy cost when it was built
  long products = 0;
  long expected_products = 0;
  double reorder_ms = 0.0;
  psp_csr *reordered = nullptr;
  int *perm = nullptr;     // new -> old (device)
  int *inv = nullptr;      // old -> new
  double *xp = nullptr;    // x, then y, in the new numbering (scratch, 2 * nrows doubles)
  int orig_max_blocks = 0;
  bool reorder_on_device = false;  // the numbering was computed by reorder_rcm_device
  // csr_w4_view: is the operator a 7-offset one of a 3-D grid WITHOUT couplings across the ends of a grid line (entries
  // at offset +-1 / +-nx only between cells that are neighbours on the grid)?  -1 not examined, 0 no, 1 yes
  int grid_state = -1;
  int grid_nx = 0, grid_ny = 0;
  // csr_w4_view: does every offset carry ONE value (constant-coefficient stencils)?  -1 not examined, 0 no, 1 yes
  int constv_state = -1;
  double constv[16] = {0};
};

}  // namespace psp

// one side table per handle, keyed by pointer (keeps psp_csr POD-like for the solvers)
#include <mutex>
#include <unordered_map>
static std::unordered_map<const psp_csr *, psp::CsrExtra> g_extra;
static std::mutex g_extra_mu;

static int finalize_csr(psp_csr *A) {
  // max row length decides the chunk target (TILE - max_row - 3 keeps a chunk in one tile)
  int *d_max;
  PSP_HIP(hipMalloc((void **)&d_max, sizeof(int)));
  PSP_HIP(hipMemsetAsync(d_max, 0, sizeof(int), stream()));
  if (A->nrows > 0) {
    int grid = std::min((A->nrows + 255) / 256, 2048);
    hipLaunchKernelGGL(max_row_kernel, dim3(grid), dim3(256), 0, stream(), A->nrows, A->ind, d_max);
    PSP_LAUNCH_CHECK();
  }
  PSP_HIP(hipMemcpyAsync(&A->max_row_nnz, d_max, sizeof(int), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  PSP_HIP(hipFree(d_max));
  return PSP_OK;
}

static int get_chunk_table(psp_csr *A, int tile, ChunkTable **out) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  psp::CsrExtra &ex = g_extra[A];
  ChunkTable &t = ex.t[tile];
  if (t.tab == nullptr) {
    // chunk c holds the rows that START in nonzeros [c*target, (c+1)*target); with
    // target <= tile - max_row it never needs more than `tile` nonzeros from c*target on
    int target = (tile - A->max_row_nnz) & ~3;
    if (target < tile / 2) target = tile / 2;  // very long rows: chunks spill into more tiles
    long nch = ((long)A->nnz + target - 1) / target;
    if (nch < 1) nch = 1;
    t.tile = tile;
    t.target = target;
    t.nchunks = (int)nch;
    PSP_HIP(hipMalloc((void **)&t.tab, sizeof(int2) * (nch + 1)));
    int grid = (int)((nch + 1 + 255) / 256);
    hipLaunchKernelGGL(build_chunk_table, dim3(grid), dim3(256), 0, stream(), A->nrows, A->ind,
                       target, (int)nch, t.tab);
    PSP_LAUNCH_CHECK();
  }
  *out = &t;
  return PSP_OK;
}

// row-offset table of csr_spmv_w2 (built on first use)
static int ensure_rowoff(const psp_csr *A, ChunkTable *t) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  if (t->max_rows >= 0) return PSP_OK;
  int *d_max;
  PSP_HIP(hipMalloc((void **)&d_max, sizeof(int)));
  PSP_HIP(hipMemsetAsync(d_max, 0, sizeof(int), stream()));
  hipLaunchKernelGGL(max_chunk_rows_kernel, dim3(std::min((t->nchunks + 255) / 256, 2048)), dim3(256),
                     0, stream(), t->nchunks, t->tab, d_max);
  PSP_LAUNCH_CHECK();
  int mr = 0;
  PSP_HIP(hipMemcpyAsync(&mr, d_max, sizeof(int), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  PSP_HIP(hipFree(d_max));
  t->max_rows = mr;
  const int np = (mr + 1 + 63) / 64;  // + 1: the end offset of the last row
  if (np > 4) {
    t->np = 0;  // many short/empty rows: stay with the ind-based kernel
    return PSP_OK;
  }
  const int npp = np < 2 ? 2 : np;
  const size_t entries = (size_t)t->nchunks * 64 * npp;
  PSP_HIP(hipMalloc((void **)&t->rowoff, sizeof(unsigned short) * entries));
  const int grid = (int)((entries + 255) / 256);
  if (npp == 2)
    hipLaunchKernelGGL(build_rowoff_kernel<2>, dim3(grid), dim3(256), 0, stream(), t->nchunks, t->target, t->tab, A->ind, t->rowoff);
  else if (npp == 3)
    hipLaunchKernelGGL(build_rowoff_kernel<3>, dim3(grid), dim3(256), 0, stream(), t->nchunks, t->target, t->tab, A->ind, t->rowoff);
  else
    hipLaunchKernelGGL(build_rowoff_kernel<4>, dim3(grid), dim3(256), 0, stream(), t->nchunks, t->target, t->tab, A->ind, t->rowoff);
  PSP_LAUNCH_CHECK();
  t->np = npp;
  return PSP_OK;
}

// scratch device allocation released on every exit path
struct ScratchDev {
  void *p = nullptr;
  ~ScratchDev() {
    if (p) (void)hipFree(p);
  }
};

// block lists + 16-bit columns of csr_spmv_w3 (built on first use; needs the w2 tables)
static int w3_nb_cap() {
  static const int cap = [] {
    const char *e = psp::tuning_env("PSP_SPMV_W3_NB");  // largest block list tried (32 / 64 / 128), 0 = never
    return e ? atoi(e) : 64;
  }();
  return cap;
}

static int ensure_w3(const psp_csr *A, ChunkTable *t) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  if (t->nb >= 0) return PSP_OK;
  t->nb = 0;
  if (t->tile != 1024 || t->np == 0 || A->nnz == 0) return PSP_OK;
  int *d_max;
  PSP_HIP(hipMalloc((void **)&d_max, 4 * sizeof(int)));
  PSP_HIP(hipMemsetAsync(d_max, 0, 4 * sizeof(int), stream()));
  // pass 1: most distinct x blocks referenced by one chunk, and how many chunks need more than 64
  hipLaunchKernelGGL(build_w3_kernel<64>, dim3(t->nchunks), dim3(64), 0, stream(), t->nchunks, t->target,
                     0, t->tab, A->col, (int *)nullptr, (unsigned short *)nullptr, d_max);
  PSP_LAUNCH_CHECK();
  int st[3] = {0, 0, 0};
  PSP_HIP(hipMemcpyAsync(st, d_max, sizeof(st), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  const int mb = st[0];
  t->max_blocks = mb;
  const int cap = w3_nb_cap();
  // a matrix that is banded except for a few rows (constraint / boundary rows, a handful of long-range
  // couplings) keeps the LDS-staged kernel: up to 2 % of the chunks may be outliers (PSP_SPMV_W3_OUTLIERS=0: none)
  static const bool outl_on = [] {
    const char *e = psp::tuning_env("PSP_SPMV_W3_OUTLIERS");
    return e ? atoi(e) != 0 : true;
  }();
  // (the shorter list is worth having: the 64-slot kernel is ~4 % slower on a matrix that fits 32)
  int nb = 0;
  if (mb <= 32 && cap >= 32) nb = 32;
  else if (outl_on && cap >= 32 && st[2] > 0 && (long)st[2] * 50 <= (long)t->nchunks) {
    nb = 32;
    t->outliers = st[2];
  } else if (mb <= 64 && cap >= 64) nb = 64;
  else if (mb <= 128 && cap >= 128) nb = 128;
  else if (outl_on && cap >= 64) {
    // Some chunks need more than 64 blocks.  Estimated cost per chunk against the 32-slot kernel on a matrix that
    // fits it: an outlier chunk (x gathered through the int32 columns, like csr_spmv_w2) ~1.4, a 64-slot chunk ~1.045
    // (0.0723 vs 0.0693 ms on the natural-order stand-in; 1.13 before unused list slots stopped costing a load).
    // Worth it up to 1.2 -- what the renumbered copy costs with its two permutation passes -- and up to 1.25 for that
    // copy itself (csr_spmv_w5 on the stored numbering, ~1.3-1.45, is the alternative then).  Measured on the FEM stand-in with
    // 1000 / 4000 wild rows in natural order: 0.103 / 0.105 ms through the copy, 0.075 / 0.08 directly.
    const double f32 = (double)st[2] / t->nchunks, f64 = (double)st[1] / t->nchunks;
    const double c32 = cap >= 32 ? 1.0 + 0.4 * f32 : 1e9, c64 = 1.045 + 0.355 * f64;
    const double limit = A->no_reorder ? 1.25 : 1.20;
    if (c32 <= c64 && c32 <= limit) {
      nb = 32;
      t->outliers = st[2];
    } else if (c64 <= limit) {
      nb = 64;
      t->outliers = st[1];
    }
  }
  if (nb == 0) {
    PSP_HIP(hipFree(d_max));
    return PSP_OK;
  }
  hipError_t e1 = hipMalloc((void **)&t->blist, sizeof(int) * (size_t)t->nchunks * nb);
  hipError_t e2 = hipMalloc((void **)&t->col16, sizeof(unsigned short) * (size_t)t->nchunks * 1024);
  if (e1 != hipSuccess || e2 != hipSuccess) {  // no room for the extra tables: stay on w2
    (void)hipGetLastError();
    if (e1 == hipSuccess) (void)hipFree(t->blist);
    if (e2 == hipSuccess) (void)hipFree(t->col16);
    t->blist = nullptr;
    t->col16 = nullptr;
    (void)hipFree(d_max);
    return PSP_OK;
  }
  // pass 2: write the tables
#define PSP_BUILD_W3(NB)                                                                          \
  hipLaunchKernelGGL(build_w3_kernel<NB>, dim3(t->nchunks), dim3(64), 0, stream(), t->nchunks,     \
                     t->target, 1, t->tab, A->col, t->blist, t->col16, d_max)
  if (nb == 32) PSP_BUILD_W3(32);
  else if (nb == 64) PSP_BUILD_W3(64);
  else PSP_BUILD_W3(128);
#undef PSP_BUILD_W3
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(stream()));
  PSP_HIP(hipFree(d_max));
  t->nb = nb;
  return PSP_OK;
}


// block lists of csr_spmv_w6 (built on first use; needs the w2 tables).  The kernel is chosen when at most 2 % of the
// chunks would gather through memory (more than 64 blocks, or more than kW6Runs runs of consecutive blocks).
static int ensure_w6(const psp_csr *A, ChunkTable *t) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  if (t->nb6 >= 0) return PSP_OK;
  t->nb6 = 0;
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_SPMV_W6");
    return e && atoi(e) == 0;
  }();
  if (off || t->tile != 1024 || t->np == 0 || A->nnz == 0) return PSP_OK;
  ScratchDev cnt;
  PSP_HIP(hipMalloc(&cnt.p, 4 * sizeof(int)));
  int *d_max = (int *)cnt.p;
  PSP_HIP(hipMemsetAsync(d_max, 0, 4 * sizeof(int), stream()));
  hipLaunchKernelGGL(build_w3_kernel<64>, dim3(t->nchunks), dim3(64), 0, stream(), t->nchunks, t->target, 0, t->tab,
                     A->col, (int *)nullptr, (unsigned short *)nullptr, d_max);
  PSP_LAUNCH_CHECK();
  int st[4] = {0, 0, 0, 0};
  PSP_HIP(hipMemcpyAsync(st, d_max, sizeof(st), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  t->direct6 = st[3];
  if (t->max_blocks == 0) t->max_blocks = st[0];
  if ((long)st[3] * 50 > (long)t->nchunks) return PSP_OK;
  if (hipMalloc((void **)&t->blist6, sizeof(int) * (size_t)t->nchunks * 64) != hipSuccess) {  // no room: stay on w2
    (void)hipGetLastError();
    t->blist6 = nullptr;
    return PSP_OK;
  }
  hipLaunchKernelGGL(build_w3_kernel<64>, dim3(t->nchunks), dim3(64), 0, stream(), t->nchunks, t->target, 1, t->tab,
                     A->col, t->blist6, (unsigned short *)nullptr, d_max);
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(stream()));
  t->nb6 = 64;
  return PSP_OK;
}

// column lists + 16-bit ranks of csr_spmv_w5 (built on first use; needs the w2 tables)
static int ensure_w5(const psp_csr *A, ChunkTable *t) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  if (t->nu >= 0) return PSP_OK;
  t->nu = 0;
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_SPMV_W5");
    return e && atoi(e) == 0;
  }();
  if (off || t->tile != 1024 || t->np == 0 || A->nnz == 0) return PSP_OK;
  ScratchDev max_mem;
  PSP_HIP(hipMalloc(&max_mem.p, sizeof(int)));
  int *d_max = (int *)max_mem.p;
  PSP_HIP(hipMemsetAsync(d_max, 0, sizeof(int), stream()));
  // pass 1: most distinct columns referenced by one chunk
  hipLaunchKernelGGL((build_w3_kernel<1024, 0>), dim3(t->nchunks), dim3(64), 0, stream(), t->nchunks, t->target, 0,
                     t->tab, A->col, (int *)nullptr, (unsigned short *)nullptr, d_max);
  PSP_LAUNCH_CHECK();
  int mc = 0;
  PSP_HIP(hipMemcpyAsync(&mc, d_max, sizeof(int), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  t->max_cols = mc;
  // staging pays while a column is used more than once on average: lists of up to half a tile
  int nu = 0;
  if (mc <= 256) nu = 256;
  else if (mc <= 384) nu = 384;
  else if (mc <= 512) nu = 512;
  if (nu == 0) return PSP_OK;
  hipError_t e1 = hipMalloc((void **)&t->ulist, sizeof(int) * (size_t)t->nchunks * nu);
  hipError_t e2 = hipMalloc((void **)&t->colu, sizeof(unsigned short) * (size_t)t->nchunks * 1024);
  if (e1 != hipSuccess || e2 != hipSuccess) {  // no room for the extra tables: stay on w2
    (void)hipGetLastError();
    if (e1 == hipSuccess) (void)hipFree(t->ulist);
    if (e2 == hipSuccess) (void)hipFree(t->colu);
    t->ulist = nullptr;
    t->colu = nullptr;
    return PSP_OK;
  }
#define PSP_BUILD_W5(NU)                                                                              \
  hipLaunchKernelGGL((build_w3_kernel<NU, 0>), dim3(t->nchunks), dim3(64), 0, stream(), t->nchunks,    \
                     t->target, 1, t->tab, A->col, t->ulist, t->colu, d_max)
  if (nu == 256) PSP_BUILD_W5(256);
  else if (nu == 384) PSP_BUILD_W5(384);
  else PSP_BUILD_W5(512);
#undef PSP_BUILD_W5
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(stream()));
  t->nu = nu;
  return PSP_OK;
}

// ---- plane-sweeping schedule -------------------------------------------------------------
// A banded operator whose half band width D is large (the 7-point stencil: D = nx*ny rows)
// touches every x line from three places D rows apart; in row order those are ~2*D*88 bytes
// of streaming apart, far more than an XCD's 4 MiB L2, so the line is fetched over the fabric
// three times (counters: 13.2 GB read per launch at 512^3 against 11.2 GB of distinct bytes,
// and w3 runs AT the fabric's streaming rate, so those bytes are time).  The schedule makes
// each XCD own "strips" -- the rows whose index modulo D falls in one interval of ~8 K rows --
// and walk a strip period by period (plane by plane): the three uses of a line then fall
// within two strip-planes (~1.5 MiB of streaming) of the same L2.  It is a permutation of
// workgroups only (launch slot -> workgroup, dealt so that slot % 8, the XCD, owns whole
// strips); any value of D gives correct results, a poor one only a poor order.
__global__ void band_kernel(int nrows, const int *__restrict__ ind, const int *__restrict__ col,
                            int *__restrict__ out) {
  int lo = 0x7fffffff, hi = -0x7fffffff;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    const int a = ind[r], b = ind[r + 1];
    if (b > a) {  // columns ascend within a row
      lo = min(lo, col[a] - r);
      hi = max(hi, col[b - 1] - r);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_down(lo, off, 64));
    hi = max(hi, __shfl_down(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(out, lo);
    atomicMax(out + 1, hi);
  }
}

static int sched_strip_rows() {
  static const int v = [] {
    // rows per strip-plane; 0 = natural order.  Default OFF: measured on MI355X at 512^3 the
    // schedule cuts fabric reads from 13.2 to 11.3 GB per launch (L2 hits 29 M -> 44 M) and is
    // 0.5-10 % SLOWER -- the re-fetches it removes were Infinity-Cache hits, and DRAM bytes, not
    // fabric bytes, bound the kernel (profiles/r1_spmv_w3_schedule.txt)
    const char *e = psp::tuning_env("PSP_SPMV_STRIP_ROWS");
    return e ? atoi(e) : 0;
  }();
  return v;
}

static int ensure_schedule(const psp_csr *A, ChunkTable *t) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  if (t->sched_state >= 0) return PSP_OK;
  t->sched_state = 0;
  const bool forced = A->sched_strip_rows >= 0;  // psp_csr_set_schedule: no size heuristics
  const int strip_rows = forced ? A->sched_strip_rows : sched_strip_rows();
  const int nwg = (t->nchunks + 3) / 4;
  if (strip_rows <= 0 || A->nrows < 1 || (!forced && nwg < 4096)) return PSP_OK;
  int *d_band;
  PSP_HIP(hipMalloc((void **)&d_band, 2 * sizeof(int)));
  const int init[2] = {0x7fffffff, -0x7fffffff};
  PSP_HIP(hipMemcpyAsync(d_band, init, sizeof(init), hipMemcpyHostToDevice, stream()));
  hipLaunchKernelGGL(band_kernel, dim3(std::min((A->nrows + 255) / 256, 4096)), dim3(256), 0, stream(),
                     A->nrows, A->ind, A->col, d_band);
  PSP_LAUNCH_CHECK();
  int band[2];
  PSP_HIP(hipMemcpyAsync(band, d_band, sizeof(band), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  PSP_HIP(hipFree(d_band));
  if (band[1] < band[0]) return PSP_OK;
  // half the span of (col - row): independent of a constant column shift (ghost-extended slabs)
  const long D = ((long)band[1] - band[0]) / 2;
  t->half_band = (int)D;
  // worth it only when the natural order cannot keep a period in one L2 (D rows * ~88 B >> 1 MiB)
  // and the matrix spans several periods
  if (D < 1 || (!forced && (D < 4L * strip_rows || D > A->nrows / 3))) return PSP_OK;
  long nstrips = (D + strip_rows / 2) / strip_rows;
  nstrips = (nstrips + 7) / 8 * 8;  // whole strips per XCD
  const double w = (double)D / (double)nstrips;
  // first row of every workgroup (4 consecutive chunks)
  std::vector<int2> tab((size_t)t->nchunks + 1);
  PSP_HIP(hipMemcpy(tab.data(), t->tab, sizeof(int2) * tab.size(), hipMemcpyDeviceToHost));
  struct Key {
    int strip, period, wg;
  };
  std::vector<Key> keys((size_t)nwg);
  for (int g = 0; g < nwg; ++g) {
    const long r0 = tab[(size_t)g * 4].x;
    long strip = (long)((double)(r0 % D) / w);
    if (strip >= nstrips) strip = nstrips - 1;
    keys[g] = {(int)strip, (int)(r0 / D), g};
  }
  std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
    if (a.strip != b.strip) return a.strip < b.strip;
    if (a.period != b.period) return a.period < b.period;
    return a.wg < b.wg;
  });
  // XCD j (= slot % 8) walks the j-th eighth of the sorted list
  const int per = (nwg + 7) / 8;
  const int grid = per * 8;
  std::vector<int> perm((size_t)grid, -1);
  for (int j = 0; j < 8; ++j)
    for (int p = 0; p < per; ++p) {
      const long src = (long)j * per + p;
      if (src < nwg) perm[(size_t)p * 8 + j] = keys[(size_t)src].wg;
    }
  PSP_HIP(hipMalloc((void **)&t->perm, sizeof(int) * (size_t)grid));
  PSP_HIP(hipMemcpy(t->perm, perm.data(), sizeof(int) * (size_t)grid, hipMemcpyHostToDevice));
  t->sched_grid = grid;
  t->sched_state = 1;
  return PSP_OK;
}

// offset-structured layout of csr_spmv_w4 (built on first use)
static int ensure_w4(const psp_csr *A, psp::CsrExtra **out) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  psp::CsrExtra &ex = g_extra[A];
  *out = &ex;
  if (ex.dia_state >= 0) return PSP_OK;
  ex.dia_state = 0;
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_SPMV_W4");
    return e && atoi(e) == 0;
  }();
  if (off || A->nrows < 1 || A->ncols < 2 || A->nnz < 1 || A->max_row_nnz > kDiaMaxOffs) return PSP_OK;
  ScratchDev tab_mem;
  PSP_HIP(hipMalloc(&tab_mem.p, (kDiaTable + 1) * sizeof(int)));
  int *d_tab = (int *)tab_mem.p;
  int init[kDiaTable + 1];
  for (int i = 0; i < kDiaTable; ++i) init[i] = kDiaEmpty;
  init[kDiaTable] = 0;
  PSP_HIP(hipMemcpyAsync(d_tab, init, sizeof(init), hipMemcpyHostToDevice, stream()));
  hipLaunchKernelGGL(dia_offsets_kernel, dim3(std::min((A->nrows + 255) / 256, 8192)), dim3(256), 0, stream(),
                     A->nrows, A->ind, A->col, d_tab, d_tab + kDiaTable);
  PSP_LAUNCH_CHECK();
  int tab[kDiaTable + 1];
  PSP_HIP(hipMemcpyAsync(tab, d_tab, sizeof(tab), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  if (tab[kDiaTable]) return PSP_OK;
  std::vector<int> offs;
  for (int i = 0; i < kDiaTable; ++i)
    if (tab[i] != kDiaEmpty) offs.push_back(tab[i]);
  if (offs.empty() || (int)offs.size() > kDiaMaxOffs) return PSP_OK;
  std::sort(offs.begin(), offs.end());
  const int no = (int)offs.size();
  // padding: rows without an entry at some offset still occupy a slot; refuse layouts that would
  // move more value bytes than the CSR form moves values + indices (12 per stored entry)
  const size_t nblk = ((size_t)A->nrows + kDiaRows - 1) / kDiaRows;
  const double slots = (double)nblk * kDiaRows * no;
  if (slots * 8.0 > 11.0 * (double)A->nnz) return PSP_OK;
  for (int i = 0; i < kDiaMaxOffs; ++i) ex.dia_offs.o[i] = i < no ? offs[i] : 0;
  const size_t nval = nblk * kDiaRows * no;
  const bool m32 = no > 16 && no <= 32, m64 = no > 32;
  const size_t nmask = nblk * kDiaRows + 2;
  hipError_t e1 = hipMalloc((void **)&ex.dia_val, sizeof(double) * nval);
  if (e1 != hipSuccess) {  // cached solver work vectors may be in the way: drop them and try once more
    (void)hipGetLastError();
    (void)psp_trim();
    e1 = hipMalloc((void **)&ex.dia_val, sizeof(double) * nval);
  }
  hipError_t e2 = m64   ? hipMalloc((void **)&ex.dia_mask64, sizeof(unsigned long long) * nmask)
                  : m32 ? hipMalloc((void **)&ex.dia_mask32, sizeof(unsigned) * nmask)
                        : hipMalloc((void **)&ex.dia_mask, sizeof(unsigned short) * nmask);
  hipError_t e3 = m64 ? hipMalloc((void **)&ex.dia_offs_dev, sizeof(int) * kDiaMaxOffs) : hipSuccess;
  if (e3 == hipSuccess && m64)
    e3 = hipMemcpyAsync(ex.dia_offs_dev, ex.dia_offs.o, sizeof(int) * kDiaMaxOffs, hipMemcpyHostToDevice, stream());
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {  // no room: stay with the CSR kernels
    (void)hipGetLastError();
    if (e1 == hipSuccess) (void)hipFree(ex.dia_val);
    if (e2 == hipSuccess) (void)hipFree(m64 ? (void *)ex.dia_mask64 : m32 ? (void *)ex.dia_mask32 : (void *)ex.dia_mask);
    if (ex.dia_offs_dev) (void)hipFree(ex.dia_offs_dev);
    ex.dia_val = nullptr;
    ex.dia_mask = nullptr;
    ex.dia_mask32 = nullptr;
    ex.dia_mask64 = nullptr;
    ex.dia_offs_dev = nullptr;
    return PSP_OK;
  }
  PSP_HIP(hipMemsetAsync(ex.dia_val, 0, sizeof(double) * nval, stream()));
  const int bgrid = std::min((A->nrows + 255) / 256, 65536);
  if (m64) {
    PSP_HIP(hipMemsetAsync(ex.dia_mask64, 0, sizeof(unsigned long long) * nmask, stream()));
    hipLaunchKernelGGL(dia_build_kernel<unsigned long long>, dim3(bgrid), dim3(256), 0, stream(), A->nrows, no,
                       ex.dia_offs, A->ind, A->col, A->val, ex.dia_val, ex.dia_mask64);
  } else if (m32) {
    PSP_HIP(hipMemsetAsync(ex.dia_mask32, 0, sizeof(unsigned) * nmask, stream()));
    hipLaunchKernelGGL(dia_build_kernel<unsigned>, dim3(bgrid), dim3(256), 0, stream(), A->nrows, no, ex.dia_offs,
                       A->ind, A->col, A->val, ex.dia_val, ex.dia_mask32);
  } else {
    PSP_HIP(hipMemsetAsync(ex.dia_mask, 0, sizeof(unsigned short) * nmask, stream()));
    hipLaunchKernelGGL(dia_build_kernel<unsigned short>, dim3(bgrid), dim3(256), 0, stream(), A->nrows, no,
                       ex.dia_offs, A->ind, A->col, A->val, ex.dia_val, ex.dia_mask);
  }
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(stream()));
  ex.dia_no = no;
  ex.dia_state = 1;
  return PSP_OK;
}

// PSP_W4_DOT_RELOAD=1 (tuning switch, read per launch: tools/dot_reuse_ab.py alternates it in one process): the dot
// epilogue of the index-free kernels loads its operand again even when it is x (rounds 1-3)
static bool w4_dot_reload() {
  const char *e = psp::tuning_env("PSP_W4_DOT_RELOAD");
  return e && atoi(e) != 0;
}

// csr_spmv_w4 over row blocks [b0, b1)
static int launch_w4(const psp_csr *A, const psp::CsrExtra *ex, int stripe, int b0, int b1, const double *x,
                     double *y, const double *dotv, double *pbuf, const int *skip, int grid, int use_div = 0,
                     double xdiv = 1.0, const double *xdiv_dev = nullptr) {
  if (use_div && ex->dia_no > 16) return fail(PSP_EINVAL, "csr_spmv_w4x has no scaled form");
  const int flags = (A->variant >= 0 ? A->variant : 0) >> 23 & 3;  // A/B knobs: bit 23 plain val loads, 24 plain y stores
  // the dot's operand as a view of x: dotv == x + offs[k] for some stored offset k (p.q, v.Av: k = the diagonal's slot)
  int dot_slot = -1;
  if (dotv && ex->dia_no <= 16 && !w4_dot_reload()) {
    const intptr_t diff = (intptr_t)dotv - (intptr_t)x;
    if (diff % (intptr_t)sizeof(double) == 0)
      for (int k = 0; k < ex->dia_no; ++k)
        if ((intptr_t)ex->dia_offs.o[k] == diff / (intptr_t)sizeof(double)) dot_slot = k;
  }
#define PSP_W4_F(NO, NTL, NTS)                                                                       \
  hipLaunchKernelGGL((csr_spmv_w4<NO, NTL, NTS>), dim3(grid), dim3(256), 0, stream(), b0, b1, A->nrows, \
                     A->ncols, stripe, ex->dia_offs, ex->dia_val, ex->dia_mask, x, y, dotv, pbuf, skip,  \
                     use_div, xdiv, xdiv_dev, dot_slot)
#define PSP_W4(NO)                                                                                   \
  case NO:                                                                                           \
    if (flags == 0) PSP_W4_F(NO, true, true);                                                        \
    else if (flags == 1) PSP_W4_F(NO, false, true);                                                  \
    else if (flags == 2) PSP_W4_F(NO, true, false);                                                  \
    else PSP_W4_F(NO, false, false);                                                                 \
    break
  switch (ex->dia_no) {
    PSP_W4(1); PSP_W4(2); PSP_W4(3); PSP_W4(4); PSP_W4(5); PSP_W4(6); PSP_W4(7); PSP_W4(8);
    PSP_W4(9); PSP_W4(10); PSP_W4(11); PSP_W4(12); PSP_W4(13); PSP_W4(14); PSP_W4(15); PSP_W4(16);
#define PSP_W4X(NO)                                                                                  \
  case NO:                                                                                           \
    hipLaunchKernelGGL((csr_spmv_w4x<NO>), dim3(grid), dim3(256), 0, stream(), b0, b1, A->nrows,      \
                       A->ncols, stripe, ex->dia_offs, ex->dia_val, ex->dia_mask32, x, y, dotv, pbuf, \
                       skip);                                                                        \
    break
    PSP_W4X(17); PSP_W4X(18); PSP_W4X(19); PSP_W4X(20); PSP_W4X(21); PSP_W4X(22); PSP_W4X(23); PSP_W4X(24);
    PSP_W4X(25); PSP_W4X(26); PSP_W4X(27); PSP_W4X(28); PSP_W4X(29); PSP_W4X(30); PSP_W4X(31); PSP_W4X(32);
#undef PSP_W4X
    default:
      if (ex->dia_no > 32 && ex->dia_no <= kDiaMaxOffs && ex->dia_mask64) {
        hipLaunchKernelGGL(csr_spmv_w4y, dim3(grid), dim3(256), 0, stream(), b0, b1, A->nrows, A->ncols, stripe,
                           ex->dia_no, ex->dia_offs_dev, ex->dia_val, ex->dia_mask64, x, y, dotv, pbuf, skip);
        break;
      }
      return fail(PSP_EINVAL, "csr_spmv_w4: %d offsets", ex->dia_no);
  }
#undef PSP_W4
#undef PSP_W4_F
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

static int w4_grid(int nblocks, int stripe) {
  int grid = (nblocks + 3) / 4;
  if (stripe > 0) grid = (grid + 8 * stripe - 1) / (8 * stripe) * (8 * stripe);
  return grid;
}

static int ensure_big_partials(psp::CsrExtra *ex, int cap) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  if (ex->big_cap < cap) {
    if (ex->big_partials) (void)hipFree(ex->big_partials);
    ex->big_partials = nullptr;
    ex->big_cap = 0;
    PSP_HIP(hipMalloc((void **)&ex->big_partials, sizeof(double) * (size_t)cap));
    ex->big_cap = cap;
  }
  return PSP_OK;
}

// sss_spmv_w4 tables of a symmetric-skyline handle (built on first use)
static std::mutex g_sss_mu;
static int ensure_sss_w4(psp_sss *S) {
  std::lock_guard<std::mutex> lk(g_sss_mu);
  if (S->w4_state >= 0) return PSP_OK;
  S->w4_state = 0;
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_SSS_W4");
    return e && atoi(e) == 0;
  }();
  if (off || S->n < 2 || S->nnz_lower < 1) return PSP_OK;
  ScratchDev tab_mem;
  PSP_HIP(hipMalloc(&tab_mem.p, (kDiaTable + 1) * sizeof(int)));
  int *d_tab = (int *)tab_mem.p;
  int init[kDiaTable + 1];
  for (int i = 0; i < kDiaTable; ++i) init[i] = kDiaEmpty;
  init[kDiaTable] = 0;
  PSP_HIP(hipMemcpyAsync(d_tab, init, sizeof(init), hipMemcpyHostToDevice, stream()));
  hipLaunchKernelGGL(dia_offsets_kernel, dim3(std::min((S->n + 255) / 256, 8192)), dim3(256), 0, stream(), S->n,
                     S->ind, S->col, d_tab, d_tab + kDiaTable);
  PSP_LAUNCH_CHECK();
  int tab[kDiaTable + 1];
  PSP_HIP(hipMemcpyAsync(tab, d_tab, sizeof(tab), hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  if (tab[kDiaTable]) return PSP_OK;
  std::vector<int> offs;
  for (int i = 0; i < kDiaTable; ++i)
    if (tab[i] != kDiaEmpty) offs.push_back(tab[i]);
  if (offs.empty() || offs.size() > 8) return PSP_OK;
  std::sort(offs.begin(), offs.end());
  const int nol = (int)offs.size();
  const size_t nblk = ((size_t)S->n + kDiaRows - 1) / kDiaRows;
  // padded lower values must stay below what the mirrored product streams for them
  if ((double)nblk * kDiaRows * nol * 8.0 > 11.0 * (double)S->nnz_lower) return PSP_OK;
  SssOffs so;
  for (int i = 0; i < 8; ++i) so.o[i] = i < nol ? offs[i] : -1;
  const size_t nval = nblk * kDiaRows * nol;
  unsigned char *low = nullptr;
  hipError_t e1 = hipMalloc((void **)&S->w4_val, sizeof(double) * nval);
  hipError_t e2 = hipMalloc((void **)&S->w4_mask, sizeof(unsigned short) * (nblk * kDiaRows + 2));
  hipError_t e3 = hipMalloc((void **)&low, (size_t)S->n);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    (void)hipGetLastError();
    if (e1 == hipSuccess) (void)hipFree(S->w4_val);
    if (e2 == hipSuccess) (void)hipFree(S->w4_mask);
    if (e3 == hipSuccess) (void)hipFree(low);
    S->w4_val = nullptr;
    S->w4_mask = nullptr;
    return PSP_OK;
  }
  PSP_HIP(hipMemsetAsync(S->w4_val, 0, sizeof(double) * nval, stream()));
  PSP_HIP(hipMemsetAsync(S->w4_mask, 0, sizeof(unsigned short) * (nblk * kDiaRows + 2), stream()));
  const int grid = std::min((S->n + 255) / 256, 65536);
  bool soa = false;
  if (const char *e = psp::tuning_env("PSP_SSS_SOA")) soa = atoi(e) != 0;  // A/B, read per handle: one value array per offset
  S->w4_soa = soa;
  hipLaunchKernelGGL(sss_lowmask_kernel, dim3(grid), dim3(256), 0, stream(), S->n, nol, so, S->ind, S->col,
                     S->val, S->w4_val, low, soa ? (long)(nblk * kDiaRows) : 0L);
  hipLaunchKernelGGL(sss_mask_kernel, dim3(grid), dim3(256), 0, stream(), S->n, nol, so, low, S->w4_mask);
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(stream()));
  PSP_HIP(hipFree(low));
  for (int i = 0; i < 8; ++i) S->w4_offs[i] = so.o[i];
  S->w4_nol = nol;
  S->w4_state = 1;
  return PSP_OK;
}

static int launch_sss_w4(const psp_sss *S, int stripe, const double *x, double *y, const double *dotv,
                         double *pbuf, const int *skip, int grid, int use_div = 0, double xdiv = 1.0,
                         const double *xdiv_dev = nullptr) {
  SssOffs so;
  for (int i = 0; i < 8; ++i) so.o[i] = S->w4_offs[i];
  const int flags = (S->full->variant >= 0 ? S->full->variant : 0) >> 23 & 3;  // A/B: 1 NT lower loads (-6 %), 2 NT shifted loads (-25 %); profiles/r1_sss_spmv_w4_timing.txt
#define PSP_SW4_F(NOL, F)                                                                            \
  hipLaunchKernelGGL((sss_spmv_w4<NOL, F>), dim3(grid), dim3(256), 0, stream(), S->n, stripe, so,     \
                     S->w4_val, S->diag, S->w4_mask, x, y, dotv, pbuf, skip, use_div, xdiv, xdiv_dev,         \
                     (dotv == x && !w4_dot_reload()) ? 1 : 0)
  static const bool shfl = [] {
    const char *e = psp::tuning_env("PSP_SSS_SHFL");  // A/B: 0 = every offset by its own loads (round 1)
    return e ? atoi(e) != 0 : true;
  }();
#define PSP_SW4(NOL)                                                                                 \
  case NOL:                                                                                          \
    if (S->w4_soa) PSP_SW4_F(NOL, 12);                                                               \
    else if (flags == 0 && shfl) PSP_SW4_F(NOL, 4);                                                  \
    else if (flags == 0) PSP_SW4_F(NOL, 0);                                                          \
    else if (flags == 1) PSP_SW4_F(NOL, 1);                                                          \
    else if (flags == 2) PSP_SW4_F(NOL, 2);                                                          \
    else PSP_SW4_F(NOL, 3);                                                                          \
    break
  switch (S->w4_nol) {
    PSP_SW4(1); PSP_SW4(2); PSP_SW4(3); PSP_SW4(4); PSP_SW4(5); PSP_SW4(6); PSP_SW4(7); PSP_SW4(8);
    default:
      return fail(PSP_EINVAL, "sss_spmv_w4: %d offsets", S->w4_nol);
  }
#undef PSP_SW4
#undef PSP_SW4_F
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// y = A^T x through csr_spmv_w4_transp; *available = 0 when A has no 16-bit-mask w4 layout
static int launch_w4_transp(const psp_csr *A, const double *x, double *y, int *available) {
  *available = 0;
  if (A->nparts) return PSP_OK;
  Variant v = decode_variant(A->variant);
  if (A->w4_only) v.w4 = true;
  if (!v.w4 || A->nrows < 1 || A->ncols < 1) return PSP_OK;
  psp::CsrExtra *ex;
  PSP_TRY(ensure_w4(A, &ex));
  if (ex->dia_state != 1 || ex->dia_no > 16) return PSP_OK;
  const int grid = (int)(((long)A->ncols + 511) / 512);
#define PSP_W4T(NO)                                                                                 \
  case NO:                                                                                          \
    hipLaunchKernelGGL((csr_spmv_w4_transp<NO>), dim3(grid), dim3(256), 0, stream(), A->nrows, A->ncols, \
                       ex->dia_offs, ex->dia_val, ex->dia_mask, x, y);                               \
    break
  switch (ex->dia_no) {
    PSP_W4T(1); PSP_W4T(2); PSP_W4T(3); PSP_W4T(4); PSP_W4T(5); PSP_W4T(6); PSP_W4T(7); PSP_W4T(8);
    PSP_W4T(9); PSP_W4T(10); PSP_W4T(11); PSP_W4T(12); PSP_W4T(13); PSP_W4T(14); PSP_W4T(15); PSP_W4T(16);
    default:
      return PSP_OK;
  }
#undef PSP_W4T
  PSP_LAUNCH_CHECK();
  *available = 1;
  return PSP_OK;
}

// T (allocated: ncols x nrows, nnz entries) = transpose of the CSR triple (device arrays): stable sort by column, so
// each row of T keeps its entries in ascending original-row order
static int transpose_into(int nrows, int ncols, int nnz, const int *ind, const int *col, const double *val,
                          psp_csr *T) {
  int rc = PSP_OK;
  int *rows = nullptr, *pos = nullptr, *keys = nullptr, *perm = nullptr;
  void *tmp = nullptr;
#define TR_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      rc = fail(e_ == hipErrorOutOfMemory ? PSP_ENOMEM : PSP_ENODEV, "%s: %s", #call,       \
                hipGetErrorString(e_));                                                    \
      goto done;                                                                           \
    }                                                                                      \
  } while (0)
  static const bool by_sort = [] {  // A/B: the stable radix sort of rounds 1-5
    const char *e = psp::tuning_env("PSP_TRANSPOSE_SORT");
    return e && atoi(e) != 0;
  }();
  if (nnz > 0 && !by_sort) {
    // counting form (kernels above); scratch from the solvers' vector pool: no hipMalloc / hipFree of 80 MB arrays
    double *kbuf = nullptr, *cbuf = nullptr;
    const size_t nk = (size_t)nnz, ncur = ((size_t)ncols + 2) / 2 + 1;
    rc = psp::scratch_get(nk, &kbuf);
    if (rc == PSP_OK) rc = psp::scratch_get(ncur, &cbuf);
    if (rc == PSP_OK) {
      unsigned long long *key = reinterpret_cast<unsigned long long *>(kbuf);
      int *cursor = reinterpret_cast<int *>(cbuf);  // ncols + 1 ints: the counts, then the cursors; [ncols] = the flag
      size_t bytes = 0;
      hipError_t e = hipMemsetAsync(cursor, 0, sizeof(int) * ((size_t)ncols + 2), stream());
      const int g = (int)std::min<long>(((long)nnz + 255) / 256, 65536);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(transp_count_kernel, dim3(g), dim3(256), 0, stream(), nnz, ncols, col, cursor, cursor + ncols + 1);
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cursor, T->ind, ncols + 1, stream());
      }
      if (e == hipSuccess) e = hipMalloc(&tmp, bytes ? bytes : 1);
      if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, bytes, cursor, T->ind, ncols + 1, stream());
      int bad = 0;
      if (e == hipSuccess) e = hipMemcpyAsync(&bad, cursor + ncols + 1, sizeof(int), hipMemcpyDeviceToHost, stream());
      if (e == hipSuccess) e = hipStreamSynchronize(stream());
      if (e == hipSuccess && bad) rc = fail(PSP_EINVAL, "transpose: a column index is out of range");
      if (e == hipSuccess && !bad) e = hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)ncols, stream());
      if (e == hipSuccess && !bad) {
        hipLaunchKernelGGL(transp_slot_kernel, dim3(std::min((nrows + 3) / 4, 65536)), dim3(256), 0, stream(), nrows, ind, col,
                           T->ind, cursor, key);
        hipLaunchKernelGGL(transp_sort_kernel, dim3(std::min((ncols + 255) / 256, 65536)), dim3(256), 0, stream(), ncols, T->ind,
                           key);
        hipLaunchKernelGGL(transp_emit_kernel, dim3(g), dim3(256), 0, stream(), nnz, key, val, T->col, T->val);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(stream());
      if (e != hipSuccess)
        rc = fail(e == hipErrorOutOfMemory ? PSP_ENOMEM : PSP_ENODEV, "transpose: %s", hipGetErrorString(e));
    }
    psp::scratch_put(kbuf, nk);
    psp::scratch_put(cbuf, ncur);
    if (rc != PSP_OK) goto done;
    rc = finalize_csr(T);
    goto done;
  }
  if (nnz > 0) {
    const size_t ib = sizeof(int) * (size_t)nnz;
    TR_HIP(hipMalloc((void **)&rows, ib));
    TR_HIP(hipMalloc((void **)&pos, ib));
    TR_HIP(hipMalloc((void **)&keys, ib));
    TR_HIP(hipMalloc((void **)&perm, ib));
    const int g = std::min((nnz + 255) / 256, 65536);
    hipLaunchKernelGGL(rows_of_nonzeros_kernel, dim3(std::min((nrows + 3) / 4, 65536)), dim3(256), 0, stream(),
                       nrows, ind, rows);
    hipLaunchKernelGGL(iota_int_kernel, dim3(g), dim3(256), 0, stream(), nnz, pos);
    int bits = 1;
    while (bits < 31 && (1L << bits) < ncols) ++bits;
    size_t bytes = 0;
    TR_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, col, keys, pos, perm, nnz, 0, bits, stream()));
    TR_HIP(hipMalloc(&tmp, bytes ? bytes : 1));
    TR_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, col, keys, pos, perm, nnz, 0, bits, stream()));  // stable
    hipLaunchKernelGGL(transp_gather_kernel, dim3(g), dim3(256), 0, stream(), nnz, perm, rows, val, T->col, T->val);
    hipLaunchKernelGGL(transp_ptr_kernel, dim3(g), dim3(256), 0, stream(), nnz, ncols, keys, T->ind);
    TR_HIP(hipGetLastError());
  } else {
    TR_HIP(hipMemsetAsync(T->ind, 0, sizeof(int) * ((size_t)ncols + 1), stream()));
  }
  TR_HIP(hipStreamSynchronize(stream()));
  rc = finalize_csr(T);
done:
#undef TR_HIP
  (void)hipFree(rows);
  (void)hipFree(pos);
  (void)hipFree(keys);
  (void)hipFree(perm);
  (void)hipFree(tmp);
  return rc;
}

// A^T as a CSR handle of its own, cached on A (irregular matrices; w4 matrices use csr_spmv_w4_transp)
static int ensure_transposed(const psp_csr *A, psp_csr **out) {
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    psp::CsrExtra *ex = &g_extra[A];
    if (ex->transposed) {
      *out = ex->transposed;
      return PSP_OK;
    }
  }
  psp_csr *T = nullptr;
  PSP_TRY(alloc_csr(A->ncols, A->nrows, A->nnz, &T));
  const int rc = transpose_into(A->nrows, A->ncols, A->nnz, A->ind, A->col, A->val, T);
  if (rc != PSP_OK) {
    psp_csr_destroy(T);
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    g_extra[A].transposed = T;
  }
  *out = T;
  return PSP_OK;
}

// renumbered copy of an irregular square operator for csr_spmv_w3 (psp_reorder.hip); built on first use
namespace psp {
int reorder_rcm_host(int n, const int *ind, const int *col, const double *val, std::vector<int> &perm,
                     std::vector<int> &rind, std::vector<int> &rcol, std::vector<double> &rval);
int reorder_rcm_device(int n, const int *ind, const int *col, int **perm_dev, int **inv_dev, int *status);
int reorder_symmetrize_device(int n, int nnz, const int *ind, const int *col, int **sind_out, int **scol_out,
                              long *snnz, int *ok);
int reorder_build_device(int n, const int *ind, const int *col, const double *val, const int *perm_dev,
                         const int *inv_dev, int *rind, int *rcol, double *rval);
int reorder_gather(int n, const int *perm_dev, const double *x, double *xp, const int *skip);
int reorder_scatter(int n, const int *idx_dev, const double *src, double *dst, const int *skip);
int reorder_back(int n, const int *inv_dev, const double *yp, double *y, const double *dotv, double *partials,
                 int *nparts, const int *skip);
}  // namespace psp

static int ensure_reordered(const psp_csr *A, psp::CsrExtra *ex, int orig_max_blocks) {
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    if (ex->reorder_state >= 0) return PSP_OK;
    ex->reorder_state = 0;
    ex->orig_max_blocks = orig_max_blocks;
  }
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_SPMV_REORDER");
    return e && atoi(e) == 0;
  }();
  // worth it when the gather pass (20 n bytes) is small against the matrix stream (12 nnz); the numbering is
  // computed on the host from a copy of the arrays (seconds and 30 bytes of host memory per nonzero): not attempted
  // beyond PSP_SPMV_REORDER_MAX_NNZ nonzeros (default 3e8)
  static const long max_nnz = [] {
    const char *e = psp::tuning_env("PSP_SPMV_REORDER_MAX_NNZ");
    return e ? atol(e) : 300000000L;
  }();
  if (off || A->no_reorder || A->w4_only || A->nrows != A->ncols || A->nrows < 1024 ||
      (long)A->nnz < 12L * A->nrows || (long)A->nnz > max_nnz)
    return PSP_OK;
  const int n = A->nrows;
  const size_t nnz = (size_t)A->nnz;
  // the numbering: on the device when the pattern is structurally symmetric with ascending rows (tens of
  // milliseconds at n = 1e6), else on the host from a copy of the arrays (a second or more; also what
  // PSP_SPMV_REORDER_HOST=1 forces -- the two give the same permutation, tests/test_gpu_spmv.py)
  static const bool host_forced = [] {
    const char *e = psp::tuning_env("PSP_SPMV_REORDER_HOST");
    return e && atoi(e) != 0;
  }();
  // The renumbered copy is a pure optimisation: whatever goes wrong while building it (no room for the copy or
  // for the scratch of the numbering, a failed copy to / from the host) means "no renumbering" -- the product
  // then runs on csr_spmv_w5 / csr_spmv_w2 -- and never fails the caller's y = A x.  An out-of-memory attempt is
  // repeated once after the work-vector pool has been emptied.
  psp_csr *R = nullptr;
  int *dperm = nullptr, *dinv = nullptr;
  double *xp = nullptr;
  int on_device = 0;
  auto release = [&]() {
    (void)hipGetLastError();
    if (R) psp_csr_destroy(R);
    if (dperm) (void)hipFree(dperm);
    if (dinv) (void)hipFree(dinv);
    if (xp) (void)hipFree(xp);
    R = nullptr;
    dperm = dinv = nullptr;
    xp = nullptr;
    on_device = 0;
  };
  auto attempt = [&]() -> int {
    psp::setup_mark("first product: before the renumbering");
    if (!host_forced) {
      PSP_TRY(psp::reorder_rcm_device(n, A->ind, A->col, &dperm, &dinv, &on_device));
      psp::setup_mark("renumbering: reorder_rcm_device");
      if (on_device < 0) {  // unsymmetric pattern or unsorted rows: number the pattern of A + A^T, built on the device
        int *sind = nullptr, *scol = nullptr, ok_sym = 0;
        long snnz = 0;
        on_device = 0;
        PSP_TRY(psp::reorder_symmetrize_device(n, A->nnz, A->ind, A->col, &sind, &scol, &snnz, &ok_sym));
        if (ok_sym) {
          const int rc_sym = psp::reorder_rcm_device(n, sind, scol, &dperm, &dinv, &on_device);
          (void)hipFree(sind);
          (void)hipFree(scol);
          PSP_TRY(rc_sym);
          if (on_device < 0) on_device = 0;
        }
      }
    }
    PSP_TRY(alloc_csr(n, n, (long)nnz, &R));
    R->no_reorder = true;
    psp::setup_mark("renumbering: allocate the copy");
    if (on_device) {
      PSP_TRY(psp::reorder_build_device(n, A->ind, A->col, A->val, dperm, dinv, R->ind, R->col, R->val));
      psp::setup_mark("renumbering: build R = P A P^T");
    } else {
      std::vector<int> ind((size_t)n + 1), col(nnz), perm, rind, rcol;
      std::vector<double> val(nnz), rval;
      PSP_HIP(hipMemcpy(ind.data(), A->ind, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost));
      PSP_HIP(hipMemcpy(col.data(), A->col, sizeof(int) * nnz, hipMemcpyDeviceToHost));
      PSP_HIP(hipMemcpy(val.data(), A->val, sizeof(double) * nnz, hipMemcpyDeviceToHost));
      PSP_TRY(psp::reorder_rcm_host(n, ind.data(), col.data(), val.data(), perm, rind, rcol, rval));
      std::vector<int> inv((size_t)n);
      for (int i = 0; i < n; ++i) inv[perm[i]] = i;
      PSP_HIP(hipMemcpy(R->ind, rind.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice));
      PSP_HIP(hipMemcpy(R->col, rcol.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
      PSP_HIP(hipMemcpy(R->val, rval.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
      if (dperm) (void)hipFree(dperm);
      if (dinv) (void)hipFree(dinv);
      dperm = dinv = nullptr;
      PSP_HIP(hipMalloc((void **)&dperm, sizeof(int) * (size_t)n));
      PSP_HIP(hipMalloc((void **)&dinv, sizeof(int) * (size_t)n));
      PSP_HIP(hipMemcpy(dperm, perm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
      PSP_HIP(hipMemcpy(dinv, inv.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    }
    PSP_HIP(hipMalloc((void **)&xp, sizeof(double) * 2 * (size_t)n));
    PSP_TRY(finalize_csr(R));
    psp::setup_mark("renumbering: finalize_csr(R)");
    ChunkTable *t = nullptr;
    PSP_TRY(get_chunk_table(R, 1024, &t));
    PSP_TRY(ensure_rowoff(R, t));
    if (t->np == 0) return PSP_EINVAL;  // the new numbering does not qualify either
    PSP_TRY(ensure_w3(R, t));
    psp::setup_mark("renumbering: chunk table + w3 tables of R");
    return t->nb > 0 ? PSP_OK : PSP_EINVAL;
  };
  auto guarded = [&]() -> int {
    try {
      return attempt();
    } catch (const std::bad_alloc &) {  // the host path keeps copies of the arrays in std::vector
      return PSP_ENOMEM;
    }
  };
  const auto t_build = std::chrono::steady_clock::now();
  int rc = guarded();
  if (rc == PSP_ENOMEM) {
    release();
    psp_trim();
    rc = guarded();
  }
  if (rc != PSP_OK) {
    release();
    return PSP_OK;
  }
  (void)hipStreamSynchronize(stream());
  std::lock_guard<std::mutex> lk(g_extra_mu);
  ex->reorder_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
  ex->reordered = R;
  ex->perm = dperm;
  ex->inv = dinv;
  ex->xp = xp;
  ex->reorder_on_device = on_device != 0;
  ex->reorder_state = 1;
  return PSP_OK;
}

static int ensure_packed(const psp_csr *A, char **out) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  psp::CsrExtra &ex = g_extra[A];
  if (ex.packed == nullptr) {
    const size_t ntiles = (A->padded + 63) / 64 + 1;
    PSP_HIP(hipMalloc((void **)&ex.packed, ntiles * 768));
    PSP_HIP(hipMemsetAsync(ex.packed, 0, ntiles * 768, stream()));
    const int grid = (int)std::min<size_t>((A->padded + 255) / 256, 65536);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, stream(), (long)A->padded, A->col, A->val,
                       ex.packed);
    PSP_LAUNCH_CHECK();
  }
  *out = ex.packed;
  return PSP_OK;
}

namespace psp {

// tuning aid, compiled in only with -DPSP_TUNING: PSP_SPMV_COLMASK=<int> ANDs every gathered column index
// (WRONG results, used only to price the x gathers).  The shipped library ignores the variable: -1 leaves
// the indices untouched.
static int colmask() {
#ifdef PSP_TUNING
  static const int m = [] {
    const char *e = psp::tuning_env("PSP_SPMV_COLMASK");
    return e ? atoi(e) : -1;
  }();
  return m;
#else
  return -1;
#endif
}

// workgroups per XCD stripe of csr_spmv_w1 (0 = plain dispatch order); PSP_SPMV_STRIPE overrides
static int spmv_stripe() {
  static const int m = [] {
    const char *e = psp::tuning_env("PSP_SPMV_STRIPE");
    return e ? atoi(e) : -1;
  }();
  return m;
}

// XCD stripe of w4 in workgroups of 512 rows: 32 measured best at 512^3 (0: -0.5 %, 64: -2 %,
// 256: -4 %; profiles/r1_spmv_w4_knobs.txt); an explicit variant or PSP_SPMV_STRIPE overrides
// XCD stripe of the index-free kernels: workgroup b runs on XCD b mod 8; the remap lets each XCD walk contiguous stripes of
// `stripe` workgroups (512 rows each).  Round 3 sweep (profiles/r3_w4_stripe.txt; interleaved rounds inside one process,
// 512^3 in eleven processes, the other grids in one each): on grids whose plane is a power of two -- every BASELINE config --
// a stripe of 128 workgroups is 2.5-5 % faster than the 32 of rounds 1-2 (512^3: 1.60-1.62 vs 1.64-1.69 ms; 4096^2: 0.132 vs
// 0.139; 1024^3: 13.03 vs 13.35) and flat elsewhere (320^3, 384^3, 640x640x300, 8192^2: +-1 %).  The one bad case measured
// is a stripe of exactly one plane (256^3, plane = 128 workgroups: 0.204 vs 0.185 ms), which falls back to an eighth of the
// plane (0.176).  sss_spmv_w4 keeps 32 (512^3: 1.20-1.22 ms at 16-32, 1.26 at 128).
static int w4_auto_stripe(const psp_csr *A) {
  if (A->sym_owner) return 32;
  int plane_wgs = 0;
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    auto it = g_extra.find(A);
    if (it != g_extra.end() && it->second.dia_state == 1) {
      int omax = 0;
      for (int i = 0; i < it->second.dia_no; ++i) omax = std::max(omax, std::abs(it->second.dia_offs.o[i]));
      plane_wgs = omax / (4 * kDiaRows);
    }
  }
  if (plane_wgs > 0) {
    const double r = 128.0 / plane_wgs;
    if (r > 0.7 && r < 1.5) {
      int s = 16;
      while (2 * s <= plane_wgs / 8) s *= 2;
      return s;
    }
  }
  return 128;
}

static int w4_stripe(const psp_csr *A, const Variant &v) {
  if (spmv_stripe() >= 0) return spmv_stripe();
  return A->variant < 0 ? w4_auto_stripe(A) : v.stripe;
}

template <int TILE, int VEC, bool NT>
static void launch_variant(int grid, int nchunks, int map_mode, const int2 *tab, const psp_csr *A,
                           const double *x, double *y, const double *dotv, double *partials) {
  hipLaunchKernelGGL((csr_spmv_stream<TILE, VEC, NT>), dim3(grid), dim3(kBlock), 0, stream(),
                     nchunks, map_mode, colmask(), tab, A->ind, A->col, A->val, x, y, dotv, partials);
}

// A/B knobs of the w3 kernel
// Form of the w3 value / column stream: 1 NT loads, 2 pair layout, 3 both.  Default = both: the
// pair layout alone changes nothing (2.075 vs 2.080 ms at 512^3), NT loads on the 4-wide form cost
// 13 % (each line is touched by two load instructions), together they give 2 % (2.038 ms).
// Variant bits 25-26 select the other three forms for A/B (stored value XOR 3).
static int w3_ab(const psp_csr *A) { return (A->variant >= 0 ? (A->variant >> 25) & 3 : 0) ^ 3; }

// csr_spmv_w3 over chunks [c0, c1) (the whole matrix: 0, nchunks)
template <int NP, int NB>
static void launch_w3_np_nb(const psp_csr *A, const ChunkTable *t, bool nts, int grid, int stripe, int c0,
                            int c1, const double *x, double *y, const double *dotv, double *pbuf,
                            const int *skip, const int *perm, const int *rowperm) {
#define PSP_W3_AB(NTL, PAIRS)                                                                          \
  hipLaunchKernelGGL((csr_spmv_w3<NP, NB, 4, true, NTL, PAIRS>), dim3(grid), dim3(256), 0, stream(), c0, c1, \
                     stripe, t->target, (int)A->padded - 4, A->ncols, t->tab, t->rowoff, t->col16, t->blist,   \
                     A->val, x, y, dotv, pbuf, skip, perm, rowperm, nullptr, colmod)
  const int ab = w3_ab(A);
  const char *cm = psp::tuning_env("PSP_W3_COLMOD");
  const int colmod = cm ? atoi(cm) : 0;
  if constexpr (NB == 32 || NB == 64) {
    if (t->outliers > 0) {  // one form only: the default stream layout, with the per-chunk fallback compiled in
      hipLaunchKernelGGL((csr_spmv_w3<NP, NB, 4, true, true, true, true>), dim3(grid), dim3(256), 0, stream(), c0, c1,
                         stripe, t->target, (int)A->padded - 4, A->ncols, t->tab, t->rowoff, t->col16, t->blist,
                         A->val, x, y, dotv, pbuf, skip, perm, rowperm, A->col, colmod);
      return;
    }
  }
  if (nts && ab == 1) PSP_W3_AB(true, false);
  else if (nts && ab == 2) PSP_W3_AB(false, true);
  else if (nts && ab == 3) PSP_W3_AB(true, true);
  else if (nts)
    hipLaunchKernelGGL((csr_spmv_w3<NP, NB, 4, true>), dim3(grid), dim3(256), 0, stream(), c0, c1, stripe,
                       t->target, (int)A->padded - 4, A->ncols, t->tab, t->rowoff, t->col16, t->blist,
                       A->val, x, y, dotv, pbuf, skip, perm, rowperm, nullptr, colmod);
  else
    hipLaunchKernelGGL((csr_spmv_w3<NP, NB, 4, false>), dim3(grid), dim3(256), 0, stream(), c0, c1, stripe,
                       t->target, (int)A->padded - 4, A->ncols, t->tab, t->rowoff, t->col16, t->blist,
                       A->val, x, y, dotv, pbuf, skip, perm, rowperm, nullptr, colmod);
}

template <int NP>
static void launch_w3_np(const psp_csr *A, const ChunkTable *t, bool nts, int grid, int stripe, int c0,
                         int c1, const double *x, double *y, const double *dotv, double *pbuf,
                         const int *skip, const int *perm, const int *rowperm) {
  if (t->nb == 32) launch_w3_np_nb<NP, 32>(A, t, nts, grid, stripe, c0, c1, x, y, dotv, pbuf, skip, perm, rowperm);
  else if (t->nb == 64) launch_w3_np_nb<NP, 64>(A, t, nts, grid, stripe, c0, c1, x, y, dotv, pbuf, skip, perm, rowperm);
  else launch_w3_np_nb<NP, 128>(A, t, nts, grid, stripe, c0, c1, x, y, dotv, pbuf, skip, perm, rowperm);
}

static void launch_w3(const psp_csr *A, const ChunkTable *t, bool nts, int grid, int stripe, int c0, int c1,
                      const double *x, double *y, const double *dotv, double *pbuf, const int *skip,
                      const int *perm = nullptr, const int *rowperm = nullptr) {
  if (t->np == 2) launch_w3_np<2>(A, t, nts, grid, stripe, c0, c1, x, y, dotv, pbuf, skip, perm, rowperm);
  else if (t->np == 3) launch_w3_np<3>(A, t, nts, grid, stripe, c0, c1, x, y, dotv, pbuf, skip, perm, rowperm);
  else launch_w3_np<4>(A, t, nts, grid, stripe, c0, c1, x, y, dotv, pbuf, skip, perm, rowperm);
}

// csr_spmv_w5 over chunks [c0, c1)
template <int NP>
static void launch_w5_np(const psp_csr *A, const ChunkTable *t, int grid, int stripe, int c0, int c1, const double *x,
                         double *y, const double *dotv, double *pbuf, const int *skip) {
#define PSP_W5(NU64)                                                                                  \
  hipLaunchKernelGGL((csr_spmv_w5<NP, NU64, 4, true>), dim3(grid), dim3(256), 0, stream(), c0, c1, stripe, \
                     t->target, (int)A->padded - 4, t->tab, t->rowoff, t->colu, t->ulist, A->val, x, y, dotv, \
                     pbuf, skip)
  if (t->nu == 256) PSP_W5(4);
  else if (t->nu == 384) PSP_W5(6);
  else PSP_W5(8);
#undef PSP_W5
}

static void launch_w5(const psp_csr *A, const ChunkTable *t, int grid, int stripe, int c0, int c1, const double *x,
                      double *y, const double *dotv, double *pbuf, const int *skip) {
  if (t->np == 2) launch_w5_np<2>(A, t, grid, stripe, c0, c1, x, y, dotv, pbuf, skip);
  else if (t->np == 3) launch_w5_np<3>(A, t, grid, stripe, c0, c1, x, y, dotv, pbuf, skip);
  else launch_w5_np<4>(A, t, grid, stripe, c0, c1, x, y, dotv, pbuf, skip);
}

// What multiplies a matrix whose stored numbering scatters a chunk's columns over more x blocks than
// csr_spmv_w3 takes (t->nb == 0): 1 = the renumbered copy through csr_spmv_w3 (measured best on the FEM-like
// stand-ins and the only form that cuts the cache-line traffic of the x gathers), 2 = csr_spmv_w5, 0 = neither
// (csr_spmv_w2).  A/B: variant bit 27 switches the renumbered copy off, bit 28 csr_spmv_w5.
// THE COST RULE OF THE RENUMBERED COPY (round 6, VERDICT r5 #4a).  Building it -- reverse Cuthill-McKee on the device, R =
// P A P^T, R's tables -- takes 17-20 ms at n = 9.3e5 / 4.1e7 nonzeros in a warm process (46-57 ms the first time a process
// does it); what it buys is 4-12 us per product against csr_spmv_w5 on the stored numbering (0.092-0.093 against
// 0.097-0.105 ms) and 16-24 us per Jacobi-MINRES iteration (106 against 122-130: the fused loops then run in the copy's
// numbering without permutation passes) -- profiles/r6_mtx_leg_standins.jsonl.  It pays for itself after 800 ... 4 000
// products, and the solve of BASELINE.json configs[4] converges in 14.  So a handle multiplies with csr_spmv_w5 until it
// HAS done kReorderAfter products (counted here: every product and every solver iteration on the stored numbering) or
// its caller announces that many (psp_csr_prepare / psp_sss_prepare); the copy is then built at the next product.  A
// fused solve that is under way keeps the numbering it started in from its first reduction to its last (its products go
// through the copy's two permutation passes meanwhile); the next solve starts in the copy's numbering.
// y = A x has the same bits either way; a solve's iterates differ at rounding level between the two numberings (its
// reductions add in the numbering it runs in), deterministically for a given sequence of calls.
// PSP_SPMV_REORDER_AFTER (tuning) moves the threshold; 0 = the copy at first use, as rounds 2-5 built it.
constexpr long kReorderAfter = 2048;
static long reorder_after() {
  static const long v = [] {
    const char *e = psp::tuning_env("PSP_SPMV_REORDER_AFTER");
    return e ? atol(e) : kReorderAfter;
  }();
  return v;
}

static int pick_scattered(const psp_csr *A, ChunkTable *t, psp::CsrExtra **ex_out, int *mode, bool count = false) {
  *mode = 0;
  if (t->nb != 0 || t->max_blocks <= 0) return PSP_OK;
  const int var = A->variant < 0 ? 0 : A->variant;
  if (((var >> 27) & 1) == 0) {
    psp::CsrExtra *exr;
    bool due;
    {
      std::lock_guard<std::mutex> lk(g_extra_mu);
      exr = &g_extra[A];
      due = exr->reorder_state >= 0 || exr->products >= reorder_after() || exr->expected_products >= reorder_after();
      if (count && !due) exr->products += 1;
    }
    if (due) {
      PSP_TRY(ensure_reordered(A, exr, t->max_blocks));
      if (exr->reorder_state == 1) {
        *ex_out = exr;
        *mode = 1;
        return PSP_OK;
      }
    }
  }
  if (((var >> 28) & 1) == 0) {
    PSP_TRY(ensure_w5(A, t));
    if (t->nu > 0) *mode = 2;
  }
  return PSP_OK;
}

// y = A x through the renumbered copy: xp = x[perm]; yp = R xp (csr_spmv_w3); y[j] = yp[inv[j]] (+ the dot)
static int launch_reordered(const psp_csr *A, psp::CsrExtra *ex, int stripe, const double *x, double *y,
                            const double *dotv, double *partials, int *nparts, const int *skip) {
  psp_csr *R = ex->reordered;
  ChunkTable *t;
  PSP_TRY(get_chunk_table(R, 1024, &t));
  int grid = (t->nchunks + 3) / 4;
  if (stripe > 0) grid = (grid + 8 * stripe - 1) / (8 * stripe) * (8 * stripe);
  const int n = A->nrows;
  double *xp = ex->xp, *yp = ex->xp + n;
  // the permutations as scatters (one load round trip instead of two dependent ones); PSP_SPMV_REORDER_GATHER=1: A/B
  static const bool gather_form = [] {
    const char *e = psp::tuning_env("PSP_SPMV_REORDER_GATHER");
    return e && atoi(e) != 0;
  }();
  if (gather_form) PSP_TRY(psp::reorder_gather(n, ex->perm, x, xp, skip));
  else PSP_TRY(psp::reorder_scatter(n, ex->inv, x, xp, skip));  // xp[inv[j]] = x[j]
  R->variant = A->variant;
  launch_w3(R, t, true, grid, stripe, 0, t->nchunks, xp, yp, nullptr, nullptr, skip);
  PSP_LAUNCH_CHECK();
  const int gback = (n + 1023) / 1024;
  double *pbuf = partials;
  if (partials && gback > kMaxParts) {
    PSP_TRY(ensure_big_partials(ex, gback));
    pbuf = ex->big_partials;
  }
  int np = 0;
  if (!partials && !gather_form) PSP_TRY(psp::reorder_scatter(n, ex->perm, yp, y, skip));  // y[perm[i]] = yp[i]
  else PSP_TRY(psp::reorder_back(n, ex->inv, yp, y, partials ? dotv : nullptr, pbuf, &np, skip));
  if (partials && pbuf != partials) {
    hipLaunchKernelGGL(fold_partials_kernel, dim3(kFold / 16), dim3(256), 0, stream(), pbuf, np, partials, kFold);
    PSP_LAUNCH_CHECK();
    np = kFold;
  }
  if (nparts) *nparts = np;
  return PSP_OK;
}

// y = A (x ./ xdiv) and the partials of (x ./ xdiv) . y, for the two index-free layouts only
// (MINRES: v = y / beta is never materialised); *available = 0 otherwise (nothing launched)
int csr_spmv_scaled_launch(const psp_csr *A, const double *x, double xdiv, double *y, double *partials,
                           int *nparts, int *available, const int *skip, const double *xdiv_dev) {
  *available = 0;
  static const bool on = [] {
    const char *e = psp::tuning_env("PSP_MINRES_SCALED");
    return e ? atoi(e) != 0 : true;
  }();
  Variant v = decode_variant(A->variant);
  if (A->w4_only) v.w4 = true;
  if (!on || !v.w4 || A->nparts || A->nrows != A->ncols || A->nrows < 2) return PSP_OK;
  const int stripe = w4_stripe(A, v);
  const int nblk = (A->nrows + kDiaRows - 1) / kDiaRows;
  const int grid = w4_grid(nblk, stripe);
  psp::CsrExtra *ex;
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    ex = &g_extra[A];
  }
  bool sss = false;
  if (A->sym_owner) {
    psp_sss *S = const_cast<psp_sss *>(A->sym_owner);
    PSP_TRY(ensure_sss_w4(S));
    sss = S->w4_state == 1;
  }
  if (!sss) {
    PSP_TRY(ensure_w4(A, &ex));
    if (ex->dia_state != 1 || ex->dia_no > 16) return PSP_OK;
  }
  double *pbuf = partials;
  if (partials && grid > kMaxParts) {
    PSP_TRY(ensure_big_partials(ex, grid));
    pbuf = ex->big_partials;
  }
  if (sss)
    PSP_TRY(launch_sss_w4(A->sym_owner, stripe, x, y, x, pbuf, skip, grid, 1, xdiv, xdiv_dev));
  else
    PSP_TRY(launch_w4(A, ex, stripe, 0, nblk, x, y, x, pbuf, skip, grid, 1, xdiv, xdiv_dev));
  int np = grid;
  if (pbuf != partials) {
    np = kFold;
    hipLaunchKernelGGL(fold_partials_kernel, dim3(np / 16), dim3(256), 0, stream(), pbuf, grid, partials, np);
    PSP_LAUNCH_CHECK();
  }
  if (nparts) *nparts = np;
  *available = 1;
  return PSP_OK;
}

// rows whose stored entries at offset -1 / +1 / -nx / +nx would couple cells that are NOT neighbours on an nx x ny x nz
// grid (k = i + nx j + nx ny l): counted into *bad
__global__ __launch_bounds__(256) void grid_wrap_check_kernel(int n, int nx, int ny, const unsigned short *__restrict__ mask,
                                                              int *__restrict__ bad) {
  int found = 0;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
    const unsigned m = mask[r];  // bits 0 .. 6: offsets -nx ny, -nx, -1, 0, +1, +nx, +nx ny
    const int i = (int)(r % nx), j = (int)((r / nx) % ny);
    if (((m >> 2) & 1u) && i == 0) found = 1;
    if (((m >> 4) & 1u) && i == nx - 1) found = 1;
    if (((m >> 1) & 1u) && j == 0) found = 1;
    if (((m >> 5) & 1u) && j == ny - 1) found = 1;
  }
  if (found) atomicAdd(bad, 1);
}

// per offset the smallest and the largest stored value (rows whose mask has the offset's bit), one pair per workgroup:
// out[(block * no + o) * 2 + {0, 1}]; +inf / -inf where a workgroup saw no entry at the offset
__global__ __launch_bounds__(256) void w4_value_range_kernel(int n, int no, const double *__restrict__ valT,
                                                             const unsigned short *__restrict__ mask, double *__restrict__ out) {
  __shared__ double smin[4][16], smax[4][16];
  double lo[16], hi[16];
  for (int o = 0; o < 16; ++o) {
    lo[o] = INFINITY;
    hi[o] = -INFINITY;
  }
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
    const unsigned m = mask[r];
    const double *vp = valT + (size_t)(r / kDiaRows) * no * kDiaRows + (size_t)(r % kDiaRows);
    for (int o = 0; o < no; ++o)
      if ((m >> o) & 1u) {
        const double v = vp[(size_t)o * kDiaRows];
        lo[o] = v < lo[o] ? v : lo[o];
        hi[o] = v > hi[o] ? v : hi[o];
        if (!(v == v)) hi[o] = INFINITY, lo[o] = -INFINITY;  // a NaN entry: never "constant"
      }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int o = 0; o < no; ++o) {
    double a = lo[o], b = hi[o];
    for (int off = 32; off > 0; off >>= 1) {
      const double a2 = __shfl_down(a, off, 64), b2 = __shfl_down(b, off, 64);
      a = a2 < a ? a2 : a;
      b = b2 > b ? b2 : b;
    }
    if (lane == 0) {
      smin[wid][o] = a;
      smax[wid][o] = b;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < no) {
    const int o = threadIdx.x;
    double a = smin[0][o], b = smax[0][o];
    for (int w = 1; w < 4; ++w) {
      a = smin[w][o] < a ? smin[w][o] : a;
      b = smax[w][o] > b ? smax[w][o] : b;
    }
    out[((size_t)blockIdx.x * no + o) * 2] = a;
    out[((size_t)blockIdx.x * no + o) * 2 + 1] = b;
  }
}

// What the single-kernel loops for mid-size systems (psp_mid.hip) need to know about an operator's index-free layout:
// the offsets, the value / mask tables, and the grid and XCD stripe the launch-per-phase product would use (its dot
// partials are indexed by workgroup, and the mid-size loops add theirs in exactly that order).  *available = 0 when the
// product of this handle is not csr_spmv_w4 with at most 8 offsets.
int csr_w4_view(const psp_csr *A, W4View *out, int *available) {
  *available = 0;
  Variant v = decode_variant(A->variant);
  if (A->w4_only) v.w4 = true;
  if (!v.w4 || A->nparts || A->multi || A->host || A->nrows != A->ncols || A->nrows < 2) return PSP_OK;
  // The full mirror of an sss_mat (sym_owner): its product is sss_spmv_w4 -- per row the lower entries in ascending column,
  // the diagonal, the mirrored entries in ascending row (sss_mat.c:45-55) -- which IS the mirror's row in ascending column
  // order, added left to right: the offset table of the mirror gives the same bits, and the workgroup order of the dot
  // partials (stripe, grid) is computed from the same handle.  The table is built here, on the first single-kernel solve.
  psp::CsrExtra *ex;
  PSP_TRY(ensure_w4(A, &ex));
  if (ex->dia_state != 1 || ex->dia_no > 9 || !ex->dia_mask) return PSP_OK;  // (9: the nine-point stencils of 2-D grids)
  out->no = ex->dia_no;
  for (int i = 0; i < 12; ++i) out->offs[i] = i < ex->dia_no ? ex->dia_offs.o[i] : 0;
  out->valT = ex->dia_val;
  out->mask = ex->dia_mask;
  out->stripe = w4_stripe(A, v);
  out->grid = w4_grid((A->nrows + kDiaRows - 1) / kDiaRows, out->stripe);
  // 3-D grid operator?  offsets {-s2, -s1, -1, 0, 1, s1, s2} with nx = s1, ny = s2 / s1, nz = n / s2 whole numbers, and no
  // entry that couples across the end of a grid line (one pass over the row masks, once per handle)
  out->grid3[0] = out->grid3[1] = out->grid3[2] = 0;
  if (ex->dia_no == 7) {
    const int *o = out->offs;
    const int s1 = o[5], s2 = o[6];
    if (ex->grid_state < 0) {
      ex->grid_state = 0;
      if (o[3] == 0 && o[4] == 1 && o[2] == -1 && o[1] == -s1 && o[0] == -s2 && s1 >= 2 && s2 % s1 == 0 && s2 / s1 >= 2 &&
          A->nrows % s2 == 0 && A->nrows / s2 >= 2) {
        int *bad = nullptr;
        PSP_HIP(hipMalloc((void **)&bad, sizeof(int)));
        PSP_HIP(hipMemsetAsync(bad, 0, sizeof(int), stream()));
        hipLaunchKernelGGL(grid_wrap_check_kernel, dim3(std::min((A->nrows + 255) / 256, 4096)), dim3(256), 0, stream(),
                           A->nrows, s1, s2 / s1, ex->dia_mask, bad);
        int hbad = 1;
        const hipError_t e1 = hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, stream());
        const hipError_t e2 = hipStreamSynchronize(stream());
        (void)hipFree(bad);
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(PSP_ENODEV, "csr_w4_view: grid check failed");
        if (hbad == 0) {
          ex->grid_state = 1;
          ex->grid_nx = s1;
          ex->grid_ny = s2 / s1;
        }
      }
    }
    if (ex->grid_state == 1) {
      out->grid3[0] = ex->grid_nx;
      out->grid3[1] = ex->grid_ny;
      out->grid3[2] = A->nrows / (ex->grid_nx * ex->grid_ny);
    }
  }
  // constant coefficients?  (one value per offset: the kernels of psp_mid.hip then keep 7 scalars instead of 7 registers
  // per grid point)  One pass over the table, once per handle.
  out->constv = 0;
  if (ex->dia_no <= 12) {
    if (ex->constv_state < 0) {
      ex->constv_state = 0;
      const int no = ex->dia_no, blocks = std::min((A->nrows + 255) / 256, 512);
      double *d = nullptr;
      PSP_HIP(hipMalloc((void **)&d, sizeof(double) * 2 * no * blocks));
      hipLaunchKernelGGL(w4_value_range_kernel, dim3(blocks), dim3(256), 0, stream(), A->nrows, no, ex->dia_val, ex->dia_mask, d);
      std::vector<double> h((size_t)2 * no * blocks);
      const hipError_t e1 = hipMemcpyAsync(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost, stream());
      const hipError_t e2 = hipStreamSynchronize(stream());
      (void)hipFree(d);
      if (e1 != hipSuccess || e2 != hipSuccess) return fail(PSP_ENODEV, "csr_w4_view: value range pass failed");
      bool all = true;
      for (int o = 0; o < no && all; ++o) {
        double lo = INFINITY, hi = -INFINITY;
        for (int b = 0; b < blocks; ++b) {
          lo = std::min(lo, h[((size_t)b * no + o) * 2]);
          hi = std::max(hi, h[((size_t)b * no + o) * 2 + 1]);
        }
        // (bitwise: +0.0 and -0.0 compare equal but multiply differently)
        if (!(lo == hi) || std::signbit(lo) != std::signbit(hi)) all = false;
        ex->constv[o] = lo;
      }
      if (all) ex->constv_state = 1;
    }
    if (ex->constv_state == 1) {
      out->constv = 1;
      for (int o = 0; o < 12; ++o) out->cval[o] = o < ex->dia_no ? ex->constv[o] : 0.0;
    }
  }
  *available = 1;
  return PSP_OK;
}

// the lazy loop's product (csr_spmv_w4_pf<.., XU = true>): pending x update + scan, p_new, q = A p_new, p_new.q in one
// pass; device-resident scalars only.  partials: slot 0 = p.q, slot 2 (partials + 2 kMaxParts) = the scan.
// *available = 0 when the operator has no index-free layout of <= 8 offsets or the grid exceeds the partial-sum slots.
int csr_spmv_pfx_launch(const psp_csr *A, const double *r, const double *dinv, const double *p_old, double *p_new,
                        double *q, double *x, double *partials, int *nparts, const PcgDev *dstate, int *available) {
  *available = 0;
  Variant v = decode_variant(A->variant);
  if (!v.w4 || A->nparts || A->sym_owner || A->nrows != A->ncols || A->nrows < 2 || !dstate || !partials) return PSP_OK;
  psp::CsrExtra *ex;
  PSP_TRY(ensure_w4(A, &ex));
  if (ex->dia_state != 1 || ex->dia_no > 8) return PSP_OK;  // register budget: up to 8 offsets
  const int stripe = w4_stripe(A, v);
  const int nblk = (A->nrows + kDiaRows - 1) / kDiaRows;
  const int grid = w4_grid(nblk, stripe);
  if (grid > kMaxParts) return PSP_OK;
  double dc = 0.0;
  const int pre = !dinv ? 0 : (dinv_constant(dinv, A->nrows, &dc) ? 2 : 1);
  double *scan = partials + 2 * (size_t)kMaxParts;
#define PSP_PFX(NO, PRE)                                                                                    \
  hipLaunchKernelGGL((csr_spmv_w4_pf<NO, PRE, true>), dim3(grid), dim3(256), 0, stream(), A->nrows, stripe, \
                     ex->dia_offs, ex->dia_val, ex->dia_mask, r, dinv, dc, p_old, p_new, q, 0.0, 0, partials, dstate, x, scan)
#define PSP_PFX_NO(NO)                \
  case NO:                            \
    if (pre == 0) PSP_PFX(NO, 0);     \
    else if (pre == 1) PSP_PFX(NO, 1); \
    else PSP_PFX(NO, 2);              \
    break
  switch (ex->dia_no) {
    PSP_PFX_NO(1); PSP_PFX_NO(2); PSP_PFX_NO(3); PSP_PFX_NO(4); PSP_PFX_NO(5); PSP_PFX_NO(6); PSP_PFX_NO(7); PSP_PFX_NO(8);
    default:
      return PSP_OK;
  }
#undef PSP_PFX_NO
#undef PSP_PFX
  PSP_LAUNCH_CHECK();
  *nparts = grid;
  *available = 1;
  return PSP_OK;
}

// q = A (z + beta p_old) with p_new written on the way (csr_spmv_w4_pf); *available = 0 when the
// operator has no w4 layout (the caller then runs pupdate + csr_spmv_launch)
int csr_spmv_pfused_launch(const psp_csr *A, const double *r, const double *dinv, const double *p_old,
                           double *p_new, double *q, double beta, bool first, double *partials, int *nparts,
                           const PcgDev *dstate, int *available) {
  *available = 0;
  // OFF by default: measured at 512^3 it changes nothing (296 / 298 iterations/s with, 291 / 299
  // without, alternating processes) -- the 8 bytes per row of DRAM traffic it saves are paid
  // back by reading two arrays instead of one at every neighbour position.  PSP_PCG_PFUSED=1 enables.
  static const bool on = [] {
    const char *e = psp::tuning_env("PSP_PCG_PFUSED");
    return e ? atoi(e) != 0 : false;
  }();
  Variant v = decode_variant(A->variant);
  if (!on || !v.w4 || A->nparts || A->sym_owner || A->nrows != A->ncols || A->nrows < 2) return PSP_OK;
  psp::CsrExtra *ex;
  PSP_TRY(ensure_w4(A, &ex));
  if (ex->dia_state != 1 || ex->dia_no > 8) return PSP_OK;  // register budget: up to 8 offsets
  const int stripe = w4_stripe(A, v);
  const int nblk = (A->nrows + kDiaRows - 1) / kDiaRows;
  const int grid = w4_grid(nblk, stripe);
  double *pbuf = partials;
  if (partials && grid > kMaxParts) {
    PSP_TRY(ensure_big_partials(ex, grid));
    pbuf = ex->big_partials;
  }
  double dc = 0.0;
  const int pre = !dinv ? 0 : (dinv_constant(dinv, A->nrows, &dc) ? 2 : 1);
#define PSP_PF(NO, PRE)                                                                             \
  hipLaunchKernelGGL((csr_spmv_w4_pf<NO, PRE>), dim3(grid), dim3(256), 0, stream(), A->nrows, stripe, \
                     ex->dia_offs, ex->dia_val, ex->dia_mask, r, dinv, dc, p_old, p_new, q, beta,     \
                     first ? 1 : 0, pbuf, dstate)
#define PSP_PF_NO(NO)                                                                               \
  case NO:                                                                                          \
    if (pre == 0) PSP_PF(NO, 0);                                                                    \
    else if (pre == 1) PSP_PF(NO, 1);                                                               \
    else PSP_PF(NO, 2);                                                                             \
    break
  switch (ex->dia_no) {
    PSP_PF_NO(1); PSP_PF_NO(2); PSP_PF_NO(3); PSP_PF_NO(4); PSP_PF_NO(5); PSP_PF_NO(6); PSP_PF_NO(7); PSP_PF_NO(8);
    default:
      return PSP_OK;
  }
#undef PSP_PF_NO
#undef PSP_PF
  PSP_LAUNCH_CHECK();
  int np = grid;
  if (pbuf != partials) {
    np = kFold;
    hipLaunchKernelGGL(fold_partials_kernel, dim3(np / 16), dim3(256), 0, stream(), pbuf, grid, partials, np);
    PSP_LAUNCH_CHECK();
  }
  if (nparts) *nparts = np;
  *available = 1;
  return PSP_OK;
}

bool csr_spmv_has_skip(const psp_csr *A) {
  if (A->nparts) {
    for (int p = 0; p < A->nparts; ++p)
      if (A->parts[p]->nrows && !csr_spmv_has_skip(A->parts[p])) return false;
    return true;
  }
  Variant v = decode_variant(A->variant);
  if (A->w4_only) v.w4 = true;
  if (v.w4 && A->sym_owner) {
    psp_sss *S = const_cast<psp_sss *>(A->sym_owner);
    if (ensure_sss_w4(S) != PSP_OK) return false;
    if (S->w4_state == 1) return true;
  }
  if (v.w4) {
    psp::CsrExtra *ex;
    if (ensure_w4(A, &ex) != PSP_OK) return false;
    if (ex->dia_state == 1) return true;
  }
  if (!v.w2 || A->max_row_nnz > v.tile / 2) return false;
  ChunkTable *t;
  if (get_chunk_table(const_cast<psp_csr *>(A), v.tile, &t) != PSP_OK) return false;
  if (ensure_rowoff(A, t) != PSP_OK) return false;
  return t->np != 0;
}

// The renumbered copy R = P A P^T when y = A x goes through it (nullptr otherwise), with perm (new -> old)
// and inv (old -> new) on the device: the fused solver loops then run entirely in the new numbering --
// b, x0 and dinv are permuted once, x is permuted back once -- instead of paying the two permutation passes
// of launch_reordered in every iteration.
int csr_reordered_view(const psp_csr *A, psp_csr **R, const int **perm, const int **inv) {
  *R = nullptr;
  psp::setup_mark(nullptr);
  Variant v = decode_variant(A->variant);
  if (A->nparts || A->w4_only || A->no_reorder || !(v.w1 && v.w2 && v.w3) || A->nrows != A->ncols ||
      A->max_row_nnz > v.tile / 2 || v.tile != 1024)
    return PSP_OK;
  if (v.w4) {
    if (A->sym_owner) {
      psp_sss *S = const_cast<psp_sss *>(A->sym_owner);
      PSP_TRY(ensure_sss_w4(S));
      if (S->w4_state == 1) return PSP_OK;
    }
    psp::CsrExtra *ex;
    PSP_TRY(ensure_w4(A, &ex));
    if (ex->dia_state == 1) return PSP_OK;
  }
  psp::setup_mark("first use: index-free (w4) layouts tried");
  ChunkTable *t;
  PSP_TRY(get_chunk_table(const_cast<psp_csr *>(A), v.tile, &t));
  PSP_TRY(ensure_rowoff(A, t));
  if (t->np == 0) return PSP_OK;
  psp::setup_mark("first use: chunk table + row offsets");
  PSP_TRY(ensure_w3(A, t));
  psp::setup_mark("first use: w3 tables on the stored numbering");
  psp::CsrExtra *exs = nullptr;
  int mode = 0;
  PSP_TRY(pick_scattered(A, t, &exs, &mode));
  if (mode != 1) return PSP_OK;
  exs->reordered->variant = A->variant;
  *R = exs->reordered;
  *perm = exs->perm;
  *inv = exs->inv;
  return PSP_OK;
}

int csr_spmv_launch(const psp_csr *A, const double *x, double *y, const double *dotv,
                    double *partials, int *nparts, const int *skip) {
  Workspace *w;
  PSP_TRY(workspace(&w));
  psp::setup_mark(nullptr);
  if (A->nparts) {
    // partitioned matrix: one product per part, rows offset; each part's dot partials are folded to kFold
    // values at partials + p*kFold (fixed order), through the workspace's last slot
    double *tmp = w->partials + (size_t)(kSlots - 1) * kMaxParts;
    if (partials && (partials == tmp || (long)A->nparts * kFold > kMaxParts))
      return fail(PSP_EINVAL, "csr_spmv_launch: partitioned matrix needs a partial-sum slot other than the last");
    for (int p = 0; p < A->nparts; ++p) {
      const psp_csr *P = A->parts[p];
      if (P->nrows == 0) continue;
      const int r0 = A->part_row0[p];
      int np = 0;
      PSP_TRY(csr_spmv_launch(P, x, y + r0, (partials && dotv) ? dotv + r0 : nullptr, partials ? tmp : nullptr, &np,
                              skip));
      if (partials) {
        hipLaunchKernelGGL(fold_partials_kernel, dim3(kFold / 16), dim3(256), 0, stream(), tmp, np,
                           partials + (size_t)p * kFold, kFold);
        PSP_LAUNCH_CHECK();
      }
    }
    if (nparts) *nparts = A->nparts * kFold;
    return PSP_OK;
  }
  Variant v = decode_variant(A->variant);
  if (A->w4_only) v.w4 = true;
  if (v.w4 && A->sym_owner) {  // the full mirror of an sss_mat: multiply with the lower triangle only
    psp_sss *S = const_cast<psp_sss *>(A->sym_owner);
    PSP_TRY(ensure_sss_w4(S));
    if (S->w4_state == 1) {
      const int stripe = w4_stripe(A, v);
      const int nblk = (S->n + kDiaRows - 1) / kDiaRows;
      const int grid = w4_grid(nblk, stripe);
      double *pbuf = partials;
      if (partials && grid > kMaxParts) {
        psp::CsrExtra *ex;
        {
          std::lock_guard<std::mutex> lk(g_extra_mu);
          ex = &g_extra[A];
        }
        PSP_TRY(ensure_big_partials(ex, grid));
        pbuf = ex->big_partials;
      }
      PSP_TRY(launch_sss_w4(S, stripe, x, y, dotv, pbuf, skip, grid));
      int np = grid;
      if (pbuf != partials) {
        np = kFold;
        hipLaunchKernelGGL(fold_partials_kernel, dim3(np / 16), dim3(256), 0, stream(), pbuf, grid,
                           partials, np);
        PSP_LAUNCH_CHECK();
      }
      if (nparts) *nparts = np;
      return PSP_OK;
    }
  }
  if (v.w4) {
    psp::CsrExtra *ex;
    PSP_TRY(ensure_w4(A, &ex));
    if (ex->dia_state == 1) {
      const int stripe = w4_stripe(A, v);
      const int nblk = (A->nrows + kDiaRows - 1) / kDiaRows;
      const int grid = w4_grid(nblk, stripe);
      double *pbuf = partials;
      if (partials && grid > kMaxParts) {
        PSP_TRY(ensure_big_partials(ex, grid));
        pbuf = ex->big_partials;
      }
      PSP_TRY(launch_w4(A, ex, stripe, 0, nblk, x, y, dotv, pbuf, skip, grid));
      int np = grid;
      if (pbuf != partials) {
        np = kFold;
        hipLaunchKernelGGL(fold_partials_kernel, dim3(np / 16), dim3(256), 0, stream(), pbuf, grid,
                           partials, np);
        PSP_LAUNCH_CHECK();
      }
      if (nparts) *nparts = np;
      return PSP_OK;
    }
  }
  if (A->w4_only) return fail(PSP_EINVAL, "this operator exists only in the w4 layout (psp_csr_poisson_big)");
  if ((v.wave || v.w1) && A->max_row_nnz > v.tile / 2) {  // a chunk would not fit one wave tile
    v.wave = v.w1 = v.w2 = false;
    v.tile = 2048;
    v.vec = 4;
    v.map_mode = 0;
    v.full_grid = false;
  }
  ChunkTable *t;
  PSP_TRY(get_chunk_table(const_cast<psp_csr *>(A), v.tile, &t));
  if (v.w1) {
    if (v.w3) v.wpb = 4;
    // the NT-store / NT-load / packed forms of w2 exist with 4 waves per workgroup only: the grid must
    // be computed for that (it was computed for 8 / 16 before: half the chunks were skipped)
    if (v.w2 && (v.full_grid || v.nt || v.layout == 1)) v.wpb = 4;
    int grid = (t->nchunks + v.wpb - 1) / v.wpb;
    const int stripe = spmv_stripe() >= 0 ? spmv_stripe() : v.stripe;
    if (stripe > 0) grid = (grid + 8 * stripe - 1) / (8 * stripe) * (8 * stripe);
    if (v.w2) {
      PSP_TRY(ensure_rowoff(A, t));
      if (t->np == 0) v.w2 = false;
    }
    bool use_w3 = false;
    const int *perm = nullptr;
    if (v.w2 && v.w3) {
      PSP_TRY(ensure_w3(A, t));
      use_w3 = t->nb > 0;
      if (use_w3 && v.sched) {
        PSP_TRY(ensure_schedule(A, t));
        if (t->sched_state == 1) {
          perm = t->perm;
          grid = t->sched_grid;
        }
      }
    }
    double *pbuf = partials;
    psp::CsrExtra *ex = nullptr;
    if (partials && grid > kMaxParts) {
      std::lock_guard<std::mutex> lk(g_extra_mu);
      ex = &g_extra[A];
      if (ex->big_cap < grid) {
        if (ex->big_partials) (void)hipFree(ex->big_partials);
        ex->big_partials = nullptr;
        ex->big_cap = 0;
        PSP_HIP(hipMalloc((void **)&ex->big_partials, sizeof(double) * (size_t)grid));
        ex->big_cap = grid;
      }
      pbuf = ex->big_partials;
    }
    if (!use_w3 && v.w2 && v.w3) {  // scattered numbering
      psp::CsrExtra *exs = nullptr;
      int mode = 0;
      PSP_TRY(pick_scattered(A, t, &exs, &mode, true));
      if (mode == 1) return launch_reordered(A, exs, stripe, x, y, dotv, partials, nparts, skip);
      if (mode == 2) {
        double *pb5 = partials;
        if (partials && grid > kMaxParts) {
          psp::CsrExtra *ex5;
          {
            std::lock_guard<std::mutex> lk(g_extra_mu);
            ex5 = &g_extra[A];
          }
          PSP_TRY(ensure_big_partials(ex5, grid));
          pb5 = ex5->big_partials;
        }
        launch_w5(A, t, grid, stripe, 0, t->nchunks, x, y, dotv, pb5, skip);
        PSP_LAUNCH_CHECK();
        int np5 = grid;
        if (pb5 != partials) {
          np5 = kFold;
          hipLaunchKernelGGL(fold_partials_kernel, dim3(np5 / 16), dim3(256), 0, stream(), pb5, grid, partials, np5);
          PSP_LAUNCH_CHECK();
        }
        if (nparts) *nparts = np5;
        return PSP_OK;
      }
    }
    if (use_w3) {
      launch_w3(A, t, v.full_grid, grid, stripe, 0, t->nchunks, x, y, dotv, pbuf, skip, perm);
      PSP_LAUNCH_CHECK();
      int np = grid;
      if (pbuf != partials) {
        np = kFold;
        hipLaunchKernelGGL(fold_partials_kernel, dim3(np / 16), dim3(256), 0, stream(), pbuf, grid,
                           partials, np);
        PSP_LAUNCH_CHECK();
      }
      if (nparts) *nparts = np;
      return PSP_OK;
    }
    if (v.w2 && v.w6) {
      PSP_TRY(ensure_w6(A, t));
      if (t->nb6 > 0) {
        const int g6 = stripe > 0 ? ((t->nchunks + 3) / 4 + 8 * stripe - 1) / (8 * stripe) * (8 * stripe) : (t->nchunks + 3) / 4;
        double *pb6 = partials;
        if (partials && g6 > kMaxParts) {
          psp::CsrExtra *ex6;
          {
            std::lock_guard<std::mutex> lk(g_extra_mu);
            ex6 = &g_extra[A];
          }
          PSP_TRY(ensure_big_partials(ex6, g6));
          pb6 = ex6->big_partials;
        }
#define PSP_W6_F(NP, NTL, PAIRS)                                                                             \
  hipLaunchKernelGGL((csr_spmv_w6<NP, 4, true, NTL, PAIRS>), dim3(g6), dim3(256), 0, stream(), 0, t->nchunks, \
                     stripe, t->target, (int)A->padded - 4, A->ncols, t->tab, t->rowoff, A->col, t->blist6, A->val, \
                     x, y, dotv, pb6, skip)
        // load form as for csr_spmv_w3 (variant bits 25-26 XOR 3; default: non-temporal pair loads)
#define PSP_W6(NP)                                                                                           \
  do {                                                                                                       \
    const int ab6 = w3_ab(A);                                                                                \
    if (ab6 == 3) PSP_W6_F(NP, true, true);                                                                  \
    else if (ab6 == 2) PSP_W6_F(NP, false, true);                                                            \
    else if (ab6 == 1) PSP_W6_F(NP, true, false);                                                            \
    else PSP_W6_F(NP, false, false);                                                                         \
  } while (0)
        if (t->np == 2) PSP_W6(2); else if (t->np == 3) PSP_W6(3); else PSP_W6(4);
#undef PSP_W6
#undef PSP_W6_F
        PSP_LAUNCH_CHECK();
        int np6 = g6;
        if (pb6 != partials) {
          np6 = kFold;
          hipLaunchKernelGGL(fold_partials_kernel, dim3(np6 / 16), dim3(256), 0, stream(), pb6, g6, partials, np6);
          PSP_LAUNCH_CHECK();
        }
        if (nparts) *nparts = np6;
        return PSP_OK;
      }
    }
    char *packed = nullptr;
    if (v.w2 && v.layout == 1) PSP_TRY(ensure_packed(A, &packed));
    if (v.w2) {
      if (v.wpb > 8) v.wpb = 8;
#define PSP_W2(WT, NP, WPB)                                                                       \
  hipLaunchKernelGGL((csr_spmv_w2<WT, NP, WPB>), dim3(grid), dim3(64 * WPB), 0, stream(), 0,      \
                     t->nchunks, colmask(), stripe, t->target, (int)A->padded - 4, t->tab, t->rowoff, \
                     A->col, A->val, x, y, dotv, pbuf, skip)
#define PSP_W2_NT(WT, NP)                                                                          \
  hipLaunchKernelGGL((csr_spmv_w2<WT, NP, 4, true, false>), dim3(grid), dim3(256), 0, stream(), 0,  \
                     t->nchunks, colmask(), stripe, t->target, (int)A->padded - 4, t->tab, t->rowoff, \
                     A->col, A->val, x, y, dotv, pbuf, skip)
#define PSP_W2_NS(WT, NP)                                                                          \
  hipLaunchKernelGGL((csr_spmv_w2<WT, NP, 4, false, true>), dim3(grid), dim3(256), 0, stream(), 0,  \
                     t->nchunks, colmask(), stripe, t->target, (int)A->padded - 4, t->tab, t->rowoff, \
                     A->col, A->val, x, y, dotv, pbuf, skip)
#define PSP_W2_PK(WT, NP)                                                                          \
  hipLaunchKernelGGL((csr_spmv_w2<WT, NP, 4, false, true, true>), dim3(grid), dim3(256), 0, stream(), 0, \
                     t->nchunks, colmask(), stripe, t->target, (int)A->padded - 4, t->tab, t->rowoff, \
                     A->col, reinterpret_cast<const double *>(packed), x, y, dotv, pbuf, skip)
#define PSP_W2_WPB(WT, NP) do { if (packed) PSP_W2_PK(WT, NP); else if (v.nt) PSP_W2_NT(WT, NP); else if (v.full_grid) PSP_W2_NS(WT, NP); else if (v.wpb == 4) PSP_W2(WT, NP, 4); else PSP_W2(WT, NP, 8); } while (0)
#define PSP_W2_NP(WT) do { if (t->np == 2) PSP_W2_WPB(WT, 2); else if (t->np == 3) PSP_W2_WPB(WT, 3); else PSP_W2_WPB(WT, 4); } while (0)
      if (v.tile == 512) PSP_W2_NP(512); else PSP_W2_NP(1024);
#undef PSP_W2_NP
#undef PSP_W2_WPB
#undef PSP_W2_NT
#undef PSP_W2_NS
#undef PSP_W2_PK
#undef PSP_W2
    } else {
#define PSP_W1(WT, WPB, LAY, NT)                                                                 \
  hipLaunchKernelGGL((csr_spmv_w1<WT, WPB, LAY, NT>), dim3(grid), dim3(64 * WPB), 0, stream(),    \
                     t->nchunks, colmask(), stripe, t->target, (int)A->padded - 4, t->tab, A->ind,  \
                     A->col, A->val, x, y, dotv, pbuf)
#define PSP_W1_NT(WT, WPB, LAY) do { if (v.nt) PSP_W1(WT, WPB, LAY, true); else PSP_W1(WT, WPB, LAY, false); } while (0)
#define PSP_W1_LAY(WT, WPB) do { if (v.layout) PSP_W1_NT(WT, WPB, 1); else PSP_W1_NT(WT, WPB, 0); } while (0)
#define PSP_W1_WPB(WT) do { if (v.wpb == 4) PSP_W1_LAY(WT, 4); else if (v.wpb == 8) PSP_W1_LAY(WT, 8); else PSP_W1_LAY(WT, 16); } while (0)
    if (v.tile == 512) PSP_W1_WPB(512); else PSP_W1_WPB(1024);
#undef PSP_W1_WPB
#undef PSP_W1_LAY
#undef PSP_W1_NT
#undef PSP_W1
    }
    PSP_LAUNCH_CHECK();
    int np = grid;
    if (pbuf != partials) {
      np = kFold;
      hipLaunchKernelGGL(fold_partials_kernel, dim3(np / 16), dim3(256), 0, stream(), pbuf, grid,
                         partials, np);
      PSP_LAUNCH_CHECK();
    }
    if (nparts) *nparts = np;
    return PSP_OK;
  }
  if (v.wave) {
    // 4 waves per workgroup, one chunk per wave at a time; residency is VGPR-bound
    const int per_cu = v.tile == 512 ? 5 : 3;
    int grid = std::min((t->nchunks + 3) / 4, std::min(w->num_cu * per_cu, 2048));
    if (v.full_grid && !partials) grid = (t->nchunks + 3) / 4;
    if (v.map_mode == 1) {
      grid = grid / 8 * 8;
      if (grid < 8) grid = 8;
    }
    if (grid < 1) grid = 1;
#define PSP_WCASE(WT, NT)                                                                       \
  hipLaunchKernelGGL((csr_spmv_wave<WT, NT>), dim3(grid), dim3(kBlock), 0, stream(), t->nchunks, \
                     v.map_mode, colmask(), t->tab, A->ind, A->col, A->val, x, y, dotv, partials)
    if (v.tile == 512) { if (v.nt) PSP_WCASE(512, true); else PSP_WCASE(512, false); }
    else { if (v.nt) PSP_WCASE(1024, true); else PSP_WCASE(1024, false); }
#undef PSP_WCASE
    PSP_LAUNCH_CHECK();
    if (nparts) *nparts = grid;
    return PSP_OK;
  }
  // persistent grid: as many workgroups as stay resident (LDS: 32 KiB -> 5/CU, 16 KiB -> 8/CU)
  const int per_cu = v.tile == 4096 ? 5 : 8;
  int grid = std::min(t->nchunks, std::min(w->num_cu * per_cu, 2048));
  if (v.full_grid && !partials) grid = t->nchunks;
  if (v.map_mode == 1) {
    grid = grid / 8 * 8;
    if (grid < 8) grid = 8;
  }
  if (grid < 1) grid = 1;
#define PSP_CASE(TILE, VEC, NT)                                                             \
  launch_variant<TILE, VEC, NT>(grid, t->nchunks, v.map_mode, t->tab, A, x, y, dotv, partials)
  if (v.tile == 4096) {
    if (v.vec == 4) { if (v.nt) PSP_CASE(4096, 4, true); else PSP_CASE(4096, 4, false); }
    else if (v.vec == 2) { if (v.nt) PSP_CASE(4096, 2, true); else PSP_CASE(4096, 2, false); }
    else { if (v.nt) PSP_CASE(4096, 1, true); else PSP_CASE(4096, 1, false); }
  } else {
    if (v.vec == 4) { if (v.nt) PSP_CASE(2048, 4, true); else PSP_CASE(2048, 4, false); }
    else if (v.vec == 2) { if (v.nt) PSP_CASE(2048, 2, true); else PSP_CASE(2048, 2, false); }
    else { if (v.nt) PSP_CASE(2048, 1, true); else PSP_CASE(2048, 1, false); }
  }
#undef PSP_CASE
  PSP_LAUNCH_CHECK();
  if (nparts) *nparts = grid;
  return PSP_OK;
}

// ---- SpMV split around a halo exchange (multi-GPU): the chunks whose rows lie inside
// [row_a, row_b) touch no ghost entry and are launched first; wait() blocks until the ghost
// entries of x have arrived (on the library's stream); then the remaining chunks run.  Every
// row is computed exactly once; dot partials of the three launches go to consecutive slots.
static int chunk_lower_bound(const ChunkTable *t, int row, int *out) {
  // first chunk c with tab[c].x >= row (binary search over the device table, a few 8-byte reads)
  int lo = 0, hi = t->nchunks;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    int2 e;
    PSP_HIP(hipMemcpy(&e, t->tab + mid, sizeof(int2), hipMemcpyDeviceToHost));
    if (e.x >= row)
      hi = mid;
    else
      lo = mid + 1;
  }
  *out = lo;
  return PSP_OK;
}

template <int WT, int NP>
static void launch_w2_range(const psp_csr *A, const ChunkTable *t, bool w3, int stripe, int c0, int c1,
                            const double *x, double *y, const double *dotv, double *pbuf, int *grid_out,
                            const int *skip) {
  int grid = (c1 - c0 + 3) / 4;
  if (stripe > 0) grid = (grid + 8 * stripe - 1) / (8 * stripe) * (8 * stripe);
  *grid_out = grid;
  if (c1 <= c0) {
    *grid_out = 0;
    return;
  }
  if (w3) {
    launch_w3(A, t, false, grid, stripe, c0, c1, x, y, dotv, pbuf, skip);
    return;
  }
  hipLaunchKernelGGL((csr_spmv_w2<WT, NP, 4>), dim3(grid), dim3(256), 0, stream(), c0, c1, colmask(),
                     stripe, t->target, (int)A->padded - 4, t->tab, t->rowoff, A->col, A->val, x, y,
                     dotv, pbuf, skip);
}

int csr_spmv_overlap(const psp_csr *A, const double *x, double *y, const double *dotv,
                     double *partials, int *nparts, int row_a, int row_b, int (*wait)(void *),
                     void *ctx, const int *skip) {
  if (A->nparts) {  // no split form: exchange first, then everything
    if (wait && wait(ctx)) return fail(PSP_ECALLBACK, "halo wait callback failed");
    return csr_spmv_launch(A, x, y, dotv, partials, nparts, skip);
  }
  Variant v = decode_variant(A->variant);
  if (A->w4_only) v.w4 = true;
  if (v.w4 && row_a < row_b) {
    psp::CsrExtra *ex;
    PSP_TRY(ensure_w4(A, &ex));
    if (ex->dia_state == 1) {
      // interior = the 128-row blocks that lie inside [row_a, row_b)
      const int nblk = (A->nrows + kDiaRows - 1) / kDiaRows;
      int ba = (row_a + kDiaRows - 1) / kDiaRows, bb = row_b / kDiaRows;
      if (row_b >= A->nrows) bb = nblk;
      if (bb < ba) bb = ba;
      const int stripe = w4_stripe(A, v);
      const int g1 = bb > ba ? w4_grid(bb - ba, stripe) : 0;
      const int g2 = ba > 0 ? w4_grid(ba, stripe) : 0;
      const int g3 = nblk > bb ? w4_grid(nblk - bb, stripe) : 0;
      double *pbuf = nullptr;
      if (partials) {
        PSP_TRY(ensure_big_partials(ex, g1 + g2 + g3 + 8));
        pbuf = ex->big_partials;
      }
      if (g1) PSP_TRY(launch_w4(A, ex, stripe, ba, bb, x, y, dotv, pbuf, skip, g1));
      if (wait && wait(ctx)) return fail(PSP_ECALLBACK, "halo wait callback failed");
      if (g2) PSP_TRY(launch_w4(A, ex, stripe, 0, ba, x, y, dotv, pbuf ? pbuf + g1 : nullptr, skip, g2));
      if (g3) PSP_TRY(launch_w4(A, ex, stripe, bb, nblk, x, y, dotv, pbuf ? pbuf + g1 + g2 : nullptr, skip, g3));
      if (partials) {
        hipLaunchKernelGGL(fold_partials_kernel, dim3(kFold / 16), dim3(256), 0, stream(), pbuf, g1 + g2 + g3,
                           partials, kFold);
        PSP_LAUNCH_CHECK();
        if (nparts) *nparts = kFold;
      }
      return PSP_OK;
    }
  }
  ChunkTable *t = nullptr;
  bool ok = v.w2 && A->max_row_nnz <= v.tile / 2 && row_a < row_b;
  if (ok) {
    PSP_TRY(get_chunk_table(const_cast<psp_csr *>(A), v.tile, &t));
    PSP_TRY(ensure_rowoff(A, t));
    ok = t->np != 0;
  }
  bool w3 = false;
  if (ok && v.w3) {
    PSP_TRY(ensure_w3(A, t));
    w3 = t->nb > 0;
  }
  if (!ok) {  // no split possible with this kernel variant: exchange first, then everything
    if (wait && wait(ctx)) return fail(PSP_ECALLBACK, "halo wait callback failed");
    return csr_spmv_launch(A, x, y, dotv, partials, nparts, skip);
  }
  // interior chunk range [ca, cb): all rows >= row_a and < row_b
  psp::CsrExtra *ex;
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    ex = &g_extra[A];
  }
  int ca, cb;
  auto key = std::make_pair(row_a, row_b);
  auto it = ex->split.find(key);
  if (it != ex->split.end() && it->second.tile == v.tile) {
    ca = it->second.ca;
    cb = it->second.cb;
  } else {
    PSP_TRY(chunk_lower_bound(t, row_a, &ca));
    PSP_TRY(chunk_lower_bound(t, row_b, &cb));  // first chunk starting at/after row_b ...
    // ... the chunk before it may straddle row_b: it is interior only if it ends at row_b
    if (cb > 0) {
      int2 e;
      PSP_HIP(hipMemcpy(&e, t->tab + cb, sizeof(int2), hipMemcpyDeviceToHost));
      if (e.x > row_b) cb -= 1;
    }
    if (cb < ca) cb = ca;
    ex->split[key] = {v.tile, ca, cb};
  }
  const int stripe = spmv_stripe() >= 0 ? spmv_stripe() : v.stripe;
  const int per = 8 * (stripe > 0 ? stripe : 1) + 4;
  const long cap_needed = (long)(t->nchunks + 3) / 4 + 3L * per;
  double *pbuf = nullptr;
  if (partials) {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    if (ex->big_cap < cap_needed) {
      if (ex->big_partials) (void)hipFree(ex->big_partials);
      ex->big_partials = nullptr;
      ex->big_cap = 0;
      PSP_HIP(hipMalloc((void **)&ex->big_partials, sizeof(double) * (size_t)cap_needed));
      ex->big_cap = (int)cap_needed;
    }
    pbuf = ex->big_partials;
  }
  int g1 = 0, g2 = 0, g3 = 0;
#define PSP_RANGE(C0, C1, OFF, G)                                                              \
  do {                                                                                         \
    if (v.tile == 512) {                                                                       \
      if (t->np == 2) launch_w2_range<512, 2>(A, t, w3, stripe, C0, C1, x, y, dotv, pbuf ? pbuf + (OFF) : nullptr, &G, skip); \
      else if (t->np == 3) launch_w2_range<512, 3>(A, t, w3, stripe, C0, C1, x, y, dotv, pbuf ? pbuf + (OFF) : nullptr, &G, skip); \
      else launch_w2_range<512, 4>(A, t, w3, stripe, C0, C1, x, y, dotv, pbuf ? pbuf + (OFF) : nullptr, &G, skip); \
    } else {                                                                                   \
      if (t->np == 2) launch_w2_range<1024, 2>(A, t, w3, stripe, C0, C1, x, y, dotv, pbuf ? pbuf + (OFF) : nullptr, &G, skip); \
      else if (t->np == 3) launch_w2_range<1024, 3>(A, t, w3, stripe, C0, C1, x, y, dotv, pbuf ? pbuf + (OFF) : nullptr, &G, skip); \
      else launch_w2_range<1024, 4>(A, t, w3, stripe, C0, C1, x, y, dotv, pbuf ? pbuf + (OFF) : nullptr, &G, skip); \
    }                                                                                          \
    PSP_LAUNCH_CHECK();                                                                        \
  } while (0)
  PSP_RANGE(ca, cb, 0, g1);
  if (wait && wait(ctx)) return fail(PSP_ECALLBACK, "halo wait callback failed");
  PSP_RANGE(0, ca, g1, g2);
  PSP_RANGE(cb, t->nchunks, g1 + g2, g3);
#undef PSP_RANGE
  if (partials) {
    const int total = g1 + g2 + g3;
    hipLaunchKernelGGL(fold_partials_kernel, dim3(kFold / 16), dim3(256), 0, stream(), pbuf, total,
                       partials, kFold);
    PSP_LAUNCH_CHECK();
    if (nparts) *nparts = kFold;
  }
  return PSP_OK;
}

}  // namespace psp

// ------------------------------------------------------------------ staging helpers

namespace {

struct DevBuf {
  double *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t n) {
    PSP_HIP(hipMalloc((void **)&p, sizeof(double) * (n ? n : 1)));
    return PSP_OK;
  }
};

int upload_strided(double *dev, const double *host, size_t n, ptrdiff_t inc) {
  if (inc == 1) {
    PSP_HIP(hipMemcpyAsync(dev, host, sizeof(double) * n, hipMemcpyHostToDevice, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    return PSP_OK;
  }
  std::vector<double> tmp(n);
  for (size_t i = 0; i < n; ++i) tmp[i] = host[(ptrdiff_t)i * inc];
  PSP_HIP(hipMemcpyAsync(dev, tmp.data(), sizeof(double) * n, hipMemcpyHostToDevice, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

int download_strided(double *host, const double *dev, size_t n, ptrdiff_t inc) {
  if (inc == 1) {
    PSP_HIP(hipMemcpyAsync(host, dev, sizeof(double) * n, hipMemcpyDeviceToHost, stream()));
    PSP_HIP(hipStreamSynchronize(stream()));
    return PSP_OK;
  }
  std::vector<double> tmp(n);
  PSP_HIP(hipMemcpyAsync(tmp.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  for (size_t i = 0; i < n; ++i) host[(ptrdiff_t)i * inc] = tmp[i];
  return PSP_OK;
}

// ---- host-pointer products at the PCIe rate (the reference boundary: csr_mat.c:141-163 hands NumPy buffers over)
//
// A product on host vectors is 8 n bytes up, one kernel, 8 n bytes down: at 512^3 two 1 GiB transfers of ~19 ms each
// (56 GB/s each way, pageable or pinned alike on this platform -- tools/pcie_probe.py) around 1.65 ms of kernel.  The
// link is full duplex (95 GB/s both ways at once), so for an offset-structured operator -- rows [r0, r1) need
// x[r0 + min offset, r1 + max offset] only -- the product is pipelined in row chunks: one helper thread uploads x
// chunk by chunk, this thread launches the row blocks of a chunk as soon as the x entries it reads have arrived,
// a second helper thread downloads each finished chunk of y while later chunks are still going up.  Same kernel,
// same rows, same bits; about 2 n * 8 / 95 GB/s instead of 2 n * 8 / 56 GB/s + kernel.
struct HostStage {
  double *x = nullptr, *y = nullptr;
  size_t nx = 0, ny = 0;
  int device = -1;
  hipStream_t up = nullptr, dn = nullptr;
};
// one staging pair per host thread (device, stream and workspace are the thread's too: psp_internal.h, "Threading
// model"); a thread that ends gives its pair back
struct HostStageOwner : HostStage {
  ~HostStageOwner() {
    if (x) (void)hipFree(x);
    if (y) (void)hipFree(y);
  }
};
thread_local HostStageOwner g_stage;

// Measured (profiles/r3_host_matvec.json): 512^3, 32 chunks of 32 MiB: 23.7 ms against 39.7 ms plain (0.94 of the link's
// full-duplex rate); 4096^2 (16.7e6 rows) loses -- 6.4-7.1 ms in 4 x 32 MiB or 16 x 8 MiB chunks against 4.9 ms plain: a
// pageable copy has ~0.2 ms of fixed cost, so the pipeline needs many large chunks.  From 2^26 rows (512 MiB per vector) on.
constexpr long kPipeMinRows = 1L << 26;
inline long pipe_chunk(long) { return 1L << 22; }  // rows per chunk: 32 MiB each way, whole 128-row blocks

int host_matvec_pipelined(psp_csr *A, const double *xh, double *yh, double *xd, double *yd, bool *done) {
  *done = false;
  if (A->nparts || A->nrows < kPipeMinRows || A->nrows != A->ncols) return PSP_OK;
  const long kPipeChunk = pipe_chunk(A->nrows);
  static const bool off = [] {
    const char *e = psp::tuning_env("PSP_HOST_PIPELINE");
    return e && atoi(e) == 0;
  }();
  if (off) return PSP_OK;
  Variant v = decode_variant(A->variant);
  if (A->w4_only) v.w4 = true;
  if (!v.w4) return PSP_OK;
  psp::CsrExtra *ex;
  PSP_TRY(ensure_w4(A, &ex));
  if (ex->dia_state != 1) return PSP_OK;
  int omax = 0;
  for (int i = 0; i < ex->dia_no; ++i) omax = std::max(omax, ex->dia_offs.o[i]);
  const long n = A->nrows;
  const int K = (int)((n + kPipeChunk - 1) / kPipeChunk);
  const int device = psp::current_device();
  if (!g_stage.up) {
    PSP_HIP(hipStreamCreateWithFlags(&g_stage.up, hipStreamNonBlocking));
    PSP_HIP(hipStreamCreateWithFlags(&g_stage.dn, hipStreamNonBlocking));
  }
  std::vector<hipEvent_t> ev_up(K, nullptr), ev_k(K, nullptr);
  for (int k = 0; k < K; ++k) {
    PSP_HIP(hipEventCreateWithFlags(&ev_up[k], hipEventDisableTiming));
    PSP_HIP(hipEventCreateWithFlags(&ev_k[k], hipEventDisableTiming));
  }
  PSP_HIP(hipStreamSynchronize(stream()));  // earlier work on the staging vectors is done
  std::atomic<int> up_done{0}, k_done{0}, err{0};
  hipStream_t s_up = g_stage.up, s_dn = g_stage.dn;
  std::thread uploader([&] {
    if (hipSetDevice(device) != hipSuccess) err = 1;
    for (int c = 0; c < K && !err; ++c) {
      const long lo = c * kPipeChunk, hi = std::min(n, lo + kPipeChunk);
      if (hipMemcpyAsync(xd + lo, xh + lo, sizeof(double) * (size_t)(hi - lo), hipMemcpyHostToDevice, s_up) != hipSuccess ||
          hipEventRecord(ev_up[c], s_up) != hipSuccess)
        err = 1;
      up_done.store(c + 1, std::memory_order_release);
    }
    up_done.store(K, std::memory_order_release);
  });
  std::thread downloader([&] {
    if (hipSetDevice(device) != hipSuccess) err = 1;
    for (int k = 0; k < K && !err; ++k) {
      while (k_done.load(std::memory_order_acquire) <= k && !err) std::this_thread::yield();
      if (err) break;
      const long lo = k * kPipeChunk, hi = std::min(n, lo + kPipeChunk);
      if (hipStreamWaitEvent(s_dn, ev_k[k], 0) != hipSuccess ||
          hipMemcpyAsync(yh + lo, yd + lo, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToHost, s_dn) != hipSuccess)
        err = 1;
    }
    if (hipStreamSynchronize(s_dn) != hipSuccess) err = 1;
  });
  const int stripe = w4_stripe(A, v);
  int rc = PSP_OK;
  for (int k = 0; k < K && rc == PSP_OK && !err; ++k) {
    const long r0 = k * kPipeChunk, r1 = std::min(n, r0 + kPipeChunk);
    const long xhi = std::min(n, r1 + omax + 2);  // a lane reads the x pair of its two rows at every offset
    const int need = (int)((xhi + kPipeChunk - 1) / kPipeChunk);
    while (up_done.load(std::memory_order_acquire) < need && !err) std::this_thread::yield();
    if (err) break;
    if (hipStreamWaitEvent(stream(), ev_up[need - 1], 0) != hipSuccess) {
      err = 1;
      break;
    }
    const int b0 = (int)(r0 / kDiaRows), b1 = (int)((r1 + kDiaRows - 1) / kDiaRows);
    rc = launch_w4(A, ex, stripe, b0, b1, xd, yd, nullptr, nullptr, nullptr, w4_grid(b1 - b0, stripe));
    if (rc == PSP_OK && hipEventRecord(ev_k[k], stream()) != hipSuccess) err = 1;
    k_done.store(k + 1, std::memory_order_release);
  }
  if (rc != PSP_OK || err) err = 1;  // releases the helper threads' waits
  k_done.store(K, std::memory_order_release);
  uploader.join();
  downloader.join();
  (void)hipStreamSynchronize(stream());
  for (int k = 0; k < K; ++k) {
    (void)hipEventDestroy(ev_up[k]);
    (void)hipEventDestroy(ev_k[k]);
  }
  if (rc != PSP_OK) return rc;
  if (err) return fail(PSP_ENODEV, "host-pointer matvec pipeline: %s", hipGetErrorString(hipGetLastError()));
  *done = true;
  return PSP_OK;
}

}  // namespace

namespace psp {
int host_stage(const psp_csr *A, size_t nx, size_t ny, double **x, double **y) {
  const int device = current_device();
  if (g_stage.device != device || g_stage.nx < nx || g_stage.ny < ny) {
    host_stage_trim();
    // the pair is the library's: where the product is HBM-bound its two vectors are drawn for their roles
    // (psp_place.hip) -- once per thread and size, the pair is kept between calls
    int rc = place_operands(A, nx, ny, 1, &g_stage.y, &g_stage.x, nullptr);
    if (rc == PSP_ENOMEM) {
      host_stage_trim();
      (void)psp_trim();
      rc = place_operands(A, nx, ny, 1, &g_stage.y, &g_stage.x, nullptr);
    }
    PSP_TRY(rc);
    // (place_operands sizes both vectors for max(nx, ny))
    g_stage.nx = std::max(nx, ny);
    g_stage.ny = std::max(nx, ny);
    g_stage.device = device;
  }
  *x = g_stage.x;
  *y = g_stage.y;
  return PSP_OK;
}
void host_stage_trim() {
  if (g_stage.x) (void)hipFree(g_stage.x);
  if (g_stage.y) (void)hipFree(g_stage.y);
  g_stage.x = g_stage.y = nullptr;
  g_stage.nx = g_stage.ny = 0;
  g_stage.device = -1;
}
}  // namespace psp

// ------------------------------------------------------------------ C ABI: csr

// *bad = the first position whose column is outside [0, ncols) (unchanged: none)
__global__ __launch_bounds__(256) void csr_validate_kernel(int nnz, int ncols, const int *__restrict__ col,
                                                           unsigned long long *bad) {
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256) {
    const int c = col[k];
    if (c < 0 || c >= ncols) atomicMin(bad, (unsigned long long)k);
  }
}

extern "C" {

int psp_csr_create(int nrows, int ncols, int nnz, const int *ind_host, const int *col_host,
                   const double *val_host, psp_csr_t **out) {
  if (psp::cpu_mode()) return psp::cpu::csr_create(nrows, ncols, nnz, ind_host, col_host, val_host, out);
  if (!out || !ind_host || (nnz > 0 && (!col_host || !val_host)))
    return fail(PSP_EINVAL, "psp_csr_create: NULL argument");
  if (nrows < 0 || ncols < 0 || nnz < 0) return fail(PSP_EINVAL, "psp_csr_create: negative size");
  // validate on the host: a malformed triple must never reach a kernel
  if (ind_host[0] != 0 || ind_host[nrows] != nnz)
    return fail(PSP_EINVAL, "psp_csr_create: ind[0] must be 0 and ind[nrows] == nnz");
  for (int i = 0; i < nrows; ++i)
    if (ind_host[i + 1] < ind_host[i])
      return fail(PSP_EINVAL, "psp_csr_create: ind not monotone at row %d", i);
  // small triples are checked here; large ones on the device once they are there (csr_validate_kernel: the loop over
  // 4e7 entries was 20 ms of host time) -- either way before any kernel indexes with a column
  const bool check_on_device = nnz >= (1 << 22);
  if (!check_on_device)
    for (int k = 0; k < nnz; ++k)
      if (col_host[k] < 0 || col_host[k] >= ncols)
        return fail(PSP_EINVAL, "psp_csr_create: column index %d out of range at %d", col_host[k], k);
  psp_csr *A;
  PSP_TRY(alloc_csr(nrows, ncols, nnz, &A));
  PSP_HIP(hipMemcpyAsync(A->ind, ind_host, sizeof(int) * ((size_t)nrows + 1),
                         hipMemcpyHostToDevice, stream()));
  if (nnz > 0) {
    PSP_HIP(hipMemcpyAsync(A->col, col_host, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, stream()));
    PSP_HIP(hipMemcpyAsync(A->val, val_host, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, stream()));
  }
  if (check_on_device) {
    unsigned long long *d_bad = nullptr, bad = ~0ull;
    hipError_t e = hipMalloc((void **)&d_bad, sizeof(bad));
    if (e == hipSuccess) e = hipMemcpyAsync(d_bad, &bad, sizeof(bad), hipMemcpyHostToDevice, stream());
    if (e == hipSuccess) {
      hipLaunchKernelGGL(csr_validate_kernel, dim3(std::min((nnz + 255) / 256, 65536)), dim3(256), 0, stream(), nnz, ncols, A->col, d_bad);
      e = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream());
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    if (d_bad) (void)hipFree(d_bad);
    if (e != hipSuccess || bad != ~0ull) {
      psp_csr_destroy(A);
      if (e != hipSuccess) return fail(PSP_ENODEV, "psp_csr_create: %s", hipGetErrorString(e));
      return fail(PSP_EINVAL, "psp_csr_create: column index %d out of range at %d", col_host[bad], (int)bad);
    }
  }
  PSP_HIP(hipStreamSynchronize(stream()));
  PSP_TRY(finalize_csr(A));
  *out = A;
  return PSP_OK;
}

int psp_csr_poisson_slab(int nx, int ny, int nz, int64_t row_lo, int64_t row_hi,
                         int64_t col_shift, int ncols_local, psp_csr_t **out) {
  if (!out || nx < 1 || ny < 1 || nz < 0) return fail(PSP_EINVAL, "psp_csr_poisson: bad grid");
  const long n = (long)nx * ny * (nz > 0 ? nz : 1);
  if (row_lo < 0 || row_hi > n || row_lo > row_hi)
    return fail(PSP_EINVAL, "psp_csr_poisson: bad row range");
  const long nloc = row_hi - row_lo;
  const long nnz = poisson_prefix(row_hi, nx, ny, nz) - poisson_prefix(row_lo, nx, ny, nz);
  if (nloc > 0x7fffffffL || nnz > 0x7fffffffL || ncols_local < 0)
    return fail(PSP_EINVAL, "psp_csr_poisson: local part exceeds 32-bit indices");
  // every local column index must land inside [0, ncols_local)
  const long reach = nz > 0 ? (long)nx * ny : nx;
  long cmin = (row_lo - reach > 0 ? row_lo - reach : 0) - col_shift;
  long cmax = (row_hi - 1 + reach < n - 1 ? row_hi - 1 + reach : n - 1) - col_shift;
  if (nloc > 0 && (cmin < 0 || cmax >= ncols_local))
    return fail(PSP_EINVAL, "psp_csr_poisson: col_shift/ncols_local do not cover the halo");
  psp_csr *A;
  PSP_TRY(alloc_csr((int)nloc, ncols_local, nnz, &A));
  int grid = (int)std::min<long>((nloc + 1 + 255) / 256, 8192);
  hipLaunchKernelGGL(poisson_csr_kernel, dim3(grid), dim3(256), 0, stream(), nx, ny, nz,
                     (long)row_lo, (long)row_hi, (long)col_shift, A->ind, A->col, A->val);
  PSP_LAUNCH_CHECK();
  PSP_TRY(finalize_csr(A));
  *out = A;
  return PSP_OK;
}

int psp_csr_poisson(int nx, int ny, int nz, psp_csr_t **out) {
  if (psp::cpu_mode()) return out ? psp::cpu::csr_poisson(nx, ny, nz, out) : fail(PSP_EINVAL, "psp_csr_poisson: NULL argument");
  const long n = (long)nx * ny * (nz > 0 ? nz : 1);
  if (n > 0x7fffffffL) return fail(PSP_EINVAL, "psp_csr_poisson: n exceeds 32-bit indices");
  return psp_csr_poisson_slab(nx, ny, nz, 0, n, 0, (int)n, out);
}

int psp_csr_poisson_big_slab(int nx, int ny, int nz, int64_t row_lo, int64_t row_hi, int64_t col_shift,
                             int ncols_local, psp_csr_t **out) {
  if (!out || nx < 2 || ny < 2 || nz < 0 || nz == 1)
    return fail(PSP_EINVAL, "psp_csr_poisson_big: grid dimensions must be >= 2 (nz = 0: 2-D)");
  const long n = (long)nx * ny * (nz > 0 ? nz : 1);
  if (row_lo < 0 || row_hi > n || row_lo >= row_hi)
    return fail(PSP_EINVAL, "psp_csr_poisson_big: bad row range");
  const long nloc = row_hi - row_lo;
  if (nloc > 0x7fffffffL - 256 || ncols_local < 2)
    return fail(PSP_EINVAL, "psp_csr_poisson_big: local rows exceed 32-bit row indices");
  const bool three_d = nz > 0;
  const int no = three_d ? 7 : 5;
  const long nxy = (long)nx * ny;
  // every local column index must land inside [0, ncols_local)
  const long reach = three_d ? nxy : nx;
  const long cmin = (row_lo - reach > 0 ? row_lo - reach : 0) - col_shift;
  const long cmax = (row_hi - 1 + reach < n - 1 ? row_hi - 1 + reach : n - 1) - col_shift;
  if (cmin < 0 || cmax >= ncols_local)
    return fail(PSP_EINVAL, "psp_csr_poisson_big: col_shift/ncols_local do not cover the halo");
  PSP_TRY(ensure_device());
  psp_csr *A = new psp_csr();
  A->nrows = (int)nloc;
  A->ncols = ncols_local;
  A->nnz64 = poisson_prefix(row_hi, nx, ny, nz) - poisson_prefix(row_lo, nx, ny, nz);
  A->nnz = A->nnz64 > 0x7fffffffL ? -1 : (int)A->nnz64;
  A->max_row_nnz = no;
  A->w4_only = true;
  psp::CsrExtra *ex;
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    ex = &g_extra[A];
  }
  // local col - local row = (global col - col_shift) - (global row - row_lo) = offset + shift
  const long shift = row_lo - col_shift;
  if (shift + nxy > 0x7fffffffL || shift - nxy < -0x7fffffffL) {
    psp_csr_destroy(A);
    return fail(PSP_EINVAL, "psp_csr_poisson_big: column shift out of range");
  }
  int b = 0;
  if (three_d) ex->dia_offs.o[b++] = (int)(shift - nxy);
  ex->dia_offs.o[b++] = (int)(shift - nx);
  ex->dia_offs.o[b++] = (int)(shift - 1);
  ex->dia_offs.o[b++] = (int)shift;
  ex->dia_offs.o[b++] = (int)(shift + 1);
  ex->dia_offs.o[b++] = (int)(shift + nx);
  if (three_d) ex->dia_offs.o[b++] = (int)(shift + nxy);
  for (; b < kDiaMaxOffs; ++b) ex->dia_offs.o[b] = 0;
  A->w4_diag_slot = three_d ? 3 : 2;
  const size_t nblk = ((size_t)nloc + kDiaRows - 1) / kDiaRows;
  const size_t nval = nblk * kDiaRows * no;
  hipError_t e1 = hipMalloc((void **)&ex->dia_val, sizeof(double) * nval);
  hipError_t e2 = hipMalloc((void **)&ex->dia_mask, sizeof(unsigned short) * (nblk * kDiaRows + 2));
  if (e1 != hipSuccess || e2 != hipSuccess) {
    (void)hipGetLastError();
    psp_csr_destroy(A);
    return fail(PSP_ENOMEM, "psp_csr_poisson_big: device allocation of %zu values failed", nval);
  }
  PSP_HIP(hipMemsetAsync(ex->dia_val, 0, sizeof(double) * nval, stream()));
  PSP_HIP(hipMemsetAsync(ex->dia_mask, 0, sizeof(unsigned short) * (nblk * kDiaRows + 2), stream()));
  hipLaunchKernelGGL(poisson_w4_kernel, dim3(65536), dim3(256), 0, stream(), nx, ny, nz, (long)row_lo, nloc, no,
                     ex->dia_val, ex->dia_mask);
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(stream()));
  ex->dia_no = no;
  ex->dia_state = 1;
  *out = A;
  return PSP_OK;
}

int psp_csr_poisson_big(int nx, int ny, int nz, psp_csr_t **out) {
  const long n = (long)nx * ny * (nz > 0 ? nz : 1);
  if (n > 0x7fffffffL - 256) return fail(PSP_EINVAL, "psp_csr_poisson_big: n exceeds 32-bit row indices");
  return psp_csr_poisson_big_slab(nx, ny, nz, 0, n, 0, (int)n, out);
}

int64_t psp_csr_nnz64(const psp_csr_t *A) {
  return A ? ((A->w4_only || A->nparts || A->multi) ? A->nnz64 : (int64_t)A->nnz) : 0;
}

// rows [r0, r1) of a host triple with 64-bit offsets as one ordinary handle
static int create_part(int ncols, int64_t r0, int64_t r1, const int64_t *ind, const int *col, const double *val,
                       psp_csr **out) {
  const int64_t base = ind[r0];
  const int64_t pn = ind[r1] - base;
  std::vector<int> pind((size_t)(r1 - r0) + 1);
  for (int64_t r = r0; r <= r1; ++r) pind[(size_t)(r - r0)] = (int)(ind[r] - base);
  return psp_csr_create((int)(r1 - r0), ncols, (int)pn, pind.data(), col + base, val + base, out);
}

// nonzeros per part of a partitioned matrix; PSP_PART_NNZ lowers it so that the tests can cut small matrices
static int64_t part_nnz() {
  static const int64_t v = [] {
    const char *e = psp::tuning_env("PSP_PART_NNZ");
    const long long t = e ? atoll(e) : 0;
    return (int64_t)((t >= 64 && t < (1LL << 30)) ? t : (1LL << 30));
  }();
  return v;
}
#define kPartNnz part_nnz()

static psp_csr *new_partitioned(int nrows, int ncols, int64_t nnz, int nparts) {
  psp_csr *A = new psp_csr();
  A->nrows = nrows;
  A->ncols = ncols;
  A->nnz = -1;
  A->nnz64 = nnz;
  A->nparts = nparts;
  A->parts = new psp_csr *[nparts]();
  A->part_row0 = new int[nparts + 1]();
  return A;
}

int psp_csr_create64(int nrows, int ncols, int64_t nnz, const int64_t *ind_host, const int *col_host,
                     const double *val_host, psp_csr_t **out) {
  if (!out || !ind_host || (nnz > 0 && (!col_host || !val_host)))
    return fail(PSP_EINVAL, "psp_csr_create64: NULL argument");
  if (nrows < 0 || ncols < 0 || nnz < 0) return fail(PSP_EINVAL, "psp_csr_create64: negative size");
  if (ind_host[0] != 0 || ind_host[nrows] != nnz)
    return fail(PSP_EINVAL, "psp_csr_create64: ind[0] must be 0 and ind[nrows] == nnz");
  for (int i = 0; i < nrows; ++i)
    if (ind_host[i + 1] < ind_host[i]) return fail(PSP_EINVAL, "psp_csr_create64: ind not monotone at row %d", i);
  if (nnz <= kPartNnz) {  // fits 32-bit offsets: an ordinary handle
    psp_csr *P = nullptr;
    PSP_TRY(create_part(ncols, 0, nrows, ind_host, col_host, val_host, &P));
    *out = P;
    return PSP_OK;
  }
  // cut at row boundaries so that every part holds at most kPartNnz nonzeros
  std::vector<int64_t> cuts{0};
  while (cuts.back() < nrows) {
    const int64_t r0 = cuts.back();
    const int64_t want = ind_host[r0] + kPartNnz;
    int64_t r1 = std::upper_bound(ind_host + r0, ind_host + nrows + 1, want) - ind_host - 1;  // last r with ind[r] <= want
    if (r1 <= r0) return fail(PSP_EINVAL, "psp_csr_create64: row %ld alone exceeds 2^30 nonzeros", (long)r0);
    cuts.push_back(std::min<int64_t>(r1, nrows));
  }
  const int np = (int)cuts.size() - 1;
  psp_csr *A = new_partitioned(nrows, ncols, nnz, np);
  for (int p = 0; p < np; ++p) {
    A->part_row0[p] = (int)cuts[p];
    int rc = create_part(ncols, cuts[p], cuts[p + 1], ind_host, col_host, val_host, &A->parts[p]);
    if (rc != PSP_OK) {
      psp_csr_destroy(A);
      return rc;
    }
    A->parts[p]->no_reorder = true;
    A->max_row_nnz = std::max(A->max_row_nnz, A->parts[p]->max_row_nnz);
  }
  A->part_row0[np] = nrows;
  *out = A;
  return PSP_OK;
}

int psp_csr_random_banded(int nrows, int ncols, int m, int stride, uint64_t seed, psp_csr_t **out) {
  if (!out || nrows < 1 || ncols < 1 || m < 1 || m > 512 || stride < 1 || (long)m * stride > ncols)
    return fail(PSP_EINVAL, "psp_csr_random_banded: bad argument (need m*stride <= ncols)");
  PSP_TRY(ensure_device());
  const int64_t nnz = (int64_t)nrows * m;
  const int64_t rows_per_part = std::max<int64_t>(1, kPartNnz / m);
  const int np = (int)((nrows + rows_per_part - 1) / rows_per_part);
  psp_csr *A = nullptr;
  if (np > 1) A = new_partitioned(nrows, ncols, nnz, np);
  for (int p = 0; p < np; ++p) {
    const int64_t r0 = (int64_t)p * rows_per_part, r1 = std::min<int64_t>(nrows, r0 + rows_per_part);
    psp_csr *P = nullptr;
    int rc = alloc_csr((int)(r1 - r0), ncols, (r1 - r0) * m, &P);
    if (rc == PSP_OK) {
      hipLaunchKernelGGL(random_banded_kernel, dim3(65536), dim3(256), 0, stream(), (int)(r1 - r0), (long)r0, ncols, m,
                         stride, (unsigned long long)seed, P->ind, P->col, P->val);
      if (hipGetLastError() != hipSuccess) rc = fail(PSP_ENODEV, "psp_csr_random_banded: launch failed");
    }
    if (rc == PSP_OK) rc = finalize_csr(P);
    if (rc != PSP_OK) {
      if (P) psp_csr_destroy(P);
      if (A) psp_csr_destroy(A);
      return rc;
    }
    if (!A) {
      *out = P;
      return PSP_OK;
    }
    P->no_reorder = true;
    A->parts[p] = P;
    A->part_row0[p] = (int)r0;
    A->max_row_nnz = std::max(A->max_row_nnz, P->max_row_nnz);
  }
  A->part_row0[np] = nrows;
  *out = A;
  return PSP_OK;
}

int psp_csr_download_rows(const psp_csr_t *A, int row_lo, int row_hi, int64_t *ind_host, int *col_host,
                          double *val_host) {
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_download_rows");
  if (!A || !ind_host) return fail(PSP_EINVAL, "psp_csr_download_rows: NULL argument");
  if (A->w4_only) return fail(PSP_EINVAL, "psp_csr_download_rows: the operator has no CSR arrays");
  if (row_lo < 0 || row_hi > A->nrows || row_lo > row_hi) return fail(PSP_EINVAL, "psp_csr_download_rows: bad row range");
  int64_t written = 0;
  ind_host[0] = 0;
  const int np = A->nparts ? A->nparts : 1;
  for (int p = 0; p < np; ++p) {
    const psp_csr *P = A->nparts ? A->parts[p] : A;
    const int p0 = A->nparts ? A->part_row0[p] : 0;
    const int a = std::max(row_lo, p0) - p0, b = std::min(row_hi, p0 + P->nrows) - p0;
    if (a >= b) continue;
    std::vector<int> pi((size_t)(b - a) + 1);
    PSP_HIP(hipMemcpy(pi.data(), P->ind + a, sizeof(int) * pi.size(), hipMemcpyDeviceToHost));
    const int k0 = pi[0], cnt = pi.back() - k0;
    for (int r = a; r < b; ++r) ind_host[(size_t)(p0 + r - row_lo) + 1] = written + (pi[(size_t)(r - a) + 1] - k0);
    if (cnt > 0) {
      if (col_host) PSP_HIP(hipMemcpy(col_host + written, P->col + k0, sizeof(int) * (size_t)cnt, hipMemcpyDeviceToHost));
      if (val_host) PSP_HIP(hipMemcpy(val_host + written, P->val + k0, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
    }
    written += cnt;
  }
  return PSP_OK;
}

int psp_csr_destroy(psp_csr_t *A) {
  if (!A) return PSP_OK;
  if (A->host) return psp::cpu::csr_destroy(A);
  if (A->multi) {  // the row blocks, streams and communicators live with the multi-device object (psp_multi.hip)
    PSP_API_GUARD_H(A);
    const int rc = psp::multi_destroy(A->multi);
    delete A;
    return rc;
  }
  psp_csr *transposed = nullptr, *reordered = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    auto it = g_extra.find(A);
    if (it != g_extra.end()) {
      transposed = it->second.transposed;
      reordered = it->second.reordered;
      if (it->second.perm) (void)hipFree(it->second.perm);
      if (it->second.inv) (void)hipFree(it->second.inv);
      if (it->second.xp) (void)hipFree(it->second.xp);
      for (auto &t : it->second.t) {
        if (t.second.tab) (void)hipFree(t.second.tab);
        if (t.second.rowoff) (void)hipFree(t.second.rowoff);
        if (t.second.blist) (void)hipFree(t.second.blist);
        if (t.second.blist6) (void)hipFree(t.second.blist6);
        if (t.second.col16) (void)hipFree(t.second.col16);
        if (t.second.ulist) (void)hipFree(t.second.ulist);
        if (t.second.colu) (void)hipFree(t.second.colu);
        if (t.second.perm) (void)hipFree(t.second.perm);
      }
      if (it->second.big_partials) (void)hipFree(it->second.big_partials);
      if (it->second.packed) (void)hipFree(it->second.packed);
      if (it->second.dia_val) (void)hipFree(it->second.dia_val);
      if (it->second.dia_mask) (void)hipFree(it->second.dia_mask);
      if (it->second.dia_mask32) (void)hipFree(it->second.dia_mask32);
      if (it->second.dia_mask64) (void)hipFree(it->second.dia_mask64);
      if (it->second.dia_offs_dev) (void)hipFree(it->second.dia_offs_dev);
      g_extra.erase(it);
    }
  }
  for (int p = 0; p < A->nparts; ++p) psp_csr_destroy(A->parts[p]);
  delete[] A->parts;
  delete[] A->part_row0;
  if (transposed) psp_csr_destroy(transposed);  // outside the lock: it has side tables of its own
  if (reordered) psp_csr_destroy(reordered);
  (void)hipFree(A->ind);
  (void)hipFree(A->col);
  (void)hipFree(A->val);
  delete A;
  return PSP_OK;
}

int psp_csr_shape(const psp_csr_t *A, int *nrows, int *ncols, int *nnz) {
  if (!A) return fail(PSP_EINVAL, "psp_csr_shape: NULL handle");
  if (nrows) *nrows = A->nrows;
  if (ncols) *ncols = A->ncols;
  if (nnz) *nnz = A->nnz;
  return PSP_OK;
}

int psp_csr_download(const psp_csr_t *A, int *ind_host, int *col_host, double *val_host) {
  if (A && A->host) return psp::cpu::csr_download(A, ind_host, col_host, val_host);
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_download");
  if (!A) return fail(PSP_EINVAL, "psp_csr_download: NULL handle");
  if (A->w4_only) return fail(PSP_EINVAL, "psp_csr_download: the operator has no CSR arrays (psp_csr_poisson_big)");
  if (A->nparts) return fail(PSP_EINVAL, "psp_csr_download: more than 2^31 nonzeros: use psp_csr_download_rows");
  if (ind_host)
    PSP_HIP(hipMemcpyAsync(ind_host, A->ind, sizeof(int) * ((size_t)A->nrows + 1),
                           hipMemcpyDeviceToHost, stream()));
  if (col_host && A->nnz)
    PSP_HIP(hipMemcpyAsync(col_host, A->col, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToHost,
                           stream()));
  if (val_host && A->nnz)
    PSP_HIP(hipMemcpyAsync(val_host, A->val, sizeof(double) * (size_t)A->nnz,
                           hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

int psp_csr_diagonal_dev(const psp_csr_t *A, double *diag_dev) {
  PSP_API_GUARD_H(A);
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_diagonal_dev");
  if (A->nrows == 0) return PSP_OK;
  if (A->nparts) {
    for (int p = 0; p < A->nparts; ++p) {
      const psp_csr *P = A->parts[p];
      if (P->nrows == 0) continue;
      hipLaunchKernelGGL(csr_diag_kernel, dim3(std::min((P->nrows + 255) / 256, 4096)), dim3(256), 0, stream(),
                         P->nrows, A->part_row0[p], P->ind, P->col, P->val, diag_dev + A->part_row0[p]);
    }
    PSP_LAUNCH_CHECK();
    return PSP_OK;
  }
  if (A->w4_only) {
    psp::CsrExtra *ex;
    PSP_TRY(ensure_w4(A, &ex));
    const int zero_slot = A->w4_diag_slot;  // the slot of A[r, r] (offsets are shifted on a slab)
    hipLaunchKernelGGL(dia_diag_kernel, dim3(std::min((A->nrows + 255) / 256, 65536)), dim3(256), 0, stream(),
                       A->nrows, ex->dia_no, zero_slot, ex->dia_val, ex->dia_mask, diag_dev);
    PSP_LAUNCH_CHECK();
    return PSP_OK;
  }
  int grid = std::min((A->nrows + 255) / 256, 4096);
  hipLaunchKernelGGL(csr_diag_kernel, dim3(grid), dim3(256), 0, stream(), A->nrows, 0, A->ind, A->col,
                     A->val, diag_dev);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

int psp_csr_diagonal(const psp_csr_t *A, double *diag_host) {
  PSP_API_GUARD_H(A);
  if (!A || !diag_host) return fail(PSP_EINVAL, "psp_csr_diagonal: NULL argument");
  if (A->host) return psp::cpu::csr_diagonal(A, diag_host);
  if (A->multi) return psp::multi_diagonal_host(A->multi, diag_host);
  DevBuf d;
  PSP_TRY(d.alloc(A->nrows));
  PSP_TRY(psp_csr_diagonal_dev(A, d.p));
  return download_strided(diag_host, d.p, A->nrows, 1);
}

int psp_csr_matvec_dev(psp_csr_t *A, const double *x_dev, double *y_dev) {
  PSP_API_GUARD_H(A);
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_matvec_dev");
  if (!A || !x_dev || !y_dev) return fail(PSP_EINVAL, "psp_csr_matvec_dev: NULL argument");
  if (A->nrows == 0) return PSP_OK;
  return csr_spmv_launch(A, x_dev, y_dev, nullptr, nullptr, nullptr);
}

int psp_csr_matvec_stride(psp_csr_t *A, const double *x_host, ptrdiff_t incx, double *y_host,
                          ptrdiff_t incy) {
  PSP_API_GUARD_H(A);
  if (!A || !x_host || !y_host) return fail(PSP_EINVAL, "psp_csr_matvec: NULL argument");
  if (A->host) return psp::cpu::csr_matvec(A, x_host, incx, y_host, incy, false);
  if (A->multi) return psp::multi_matvec_host(A->multi, x_host, incx, y_host, incy);
  PSP_TRY(ensure_device());
  // device staging for the caller's host vectors: kept between calls (hipMalloc + hipFree of two GB-sized vectors cost
  // milliseconds per product); psp_trim() releases it
  double *xd, *yd;
  PSP_TRY(psp::host_stage(A, A->ncols, A->nrows, &xd, &yd));
  if (incx == 1 && incy == 1) {
    bool done = false;
    PSP_TRY(host_matvec_pipelined(A, x_host, y_host, xd, yd, &done));
    if (done) return PSP_OK;
  }
  PSP_TRY(upload_strided(xd, x_host, A->ncols, incx));
  PSP_TRY(psp_csr_matvec_dev(A, xd, yd));
  return download_strided(y_host, yd, A->nrows, incy);
}

int psp_csr_matvec(psp_csr_t *A, const double *x_host, double *y_host) {
  PSP_API_GUARD_H(A);
  return psp_csr_matvec_stride(A, x_host, 1, y_host, 1);
}

int psp_csr_matvec_transp_dev(psp_csr_t *A, const double *x_dev, double *y_dev) {
  PSP_API_GUARD_H(A);
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_matvec_transp");
  if (!A || !x_dev || !y_dev) return fail(PSP_EINVAL, "psp_csr_matvec_transp_dev: NULL argument");
  {  // offset-structured operators: exact gather in the reference's order, no atomics
    int done = 0;
    PSP_TRY(launch_w4_transp(A, x_dev, y_dev, &done));
    if (done) return PSP_OK;
  }
  if (A->w4_only) return fail(PSP_EINVAL, "matvec_transp: the operator has no CSR arrays (psp_csr_poisson_big)");
  if (A->nparts) return fail(PSP_EINVAL, "matvec_transp: not available for a partitioned (> 2^31 nonzeros) matrix");
  if (A->ncols == 0) return PSP_OK;
  // irregular matrices: multiply with A^T stored as CSR (built once): every y[c] adds its terms by
  // ascending row, the order of csr_matvec_transp_kernel (csr_mat.c:80-87) -- exact, no atomics
  psp_csr *T;
  PSP_TRY(ensure_transposed(A, &T));
  T->variant = A->variant;
  return csr_spmv_launch(T, x_dev, y_dev, nullptr, nullptr, nullptr);
}

int psp_csr_matvec_transp_stride(psp_csr_t *A, const double *x_host, ptrdiff_t incx,
                                 double *y_host, ptrdiff_t incy) {
  PSP_API_GUARD_H(A);
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_matvec_transp");
  if (!A || !x_host || !y_host) return fail(PSP_EINVAL, "psp_csr_matvec_transp: NULL argument");
  if (A->host) return psp::cpu::csr_matvec(A, x_host, incx, y_host, incy, true);
  PSP_TRY(ensure_device());
  DevBuf x, y;
  PSP_TRY(x.alloc(A->nrows));
  PSP_TRY(y.alloc(A->ncols));
  PSP_TRY(upload_strided(x.p, x_host, A->nrows, incx));
  PSP_TRY(psp_csr_matvec_transp_dev(A, x.p, y.p));
  return download_strided(y_host, y.p, A->ncols, incy);
}

int psp_csr_matvec_transp(psp_csr_t *A, const double *x_host, double *y_host) {
  PSP_API_GUARD_H(A);
  return psp_csr_matvec_transp_stride(A, x_host, 1, y_host, 1);
}

int psp_csr_set_schedule(psp_csr_t *A, int strip_rows) {
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_set_schedule");
  if (!A) return fail(PSP_EINVAL, "psp_csr_set_schedule: NULL handle");
  A->sched_strip_rows = strip_rows;
  std::lock_guard<std::mutex> lk(g_extra_mu);
  auto it = g_extra.find(A);
  if (it != g_extra.end())
    for (auto &t : it->second.t) {  // rebuilt on the next product
      if (t.second.perm) (void)hipFree(t.second.perm);
      t.second.perm = nullptr;
      t.second.sched_state = -1;
    }
  return PSP_OK;
}

int psp_csr_renumbering(psp_csr_t *A, int *perm_host, int *available) {
  PSP_API_GUARD_H(A);
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_renumbering");
  if (!A || !perm_host || !available) return fail(PSP_EINVAL, "psp_csr_renumbering: NULL argument");
  *available = 0;
  const int *dperm = nullptr;
  bool on_device = false;
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    auto it = g_extra.find(A);
    if (it != g_extra.end() && it->second.reorder_state == 1) {
      dperm = it->second.perm;
      on_device = it->second.reorder_on_device;
    }
  }
  if (!dperm) return PSP_OK;
  PSP_HIP(hipMemcpy(perm_host, dperm, sizeof(int) * (size_t)A->nrows, hipMemcpyDeviceToHost));
  *available = on_device ? 2 : 1;
  return PSP_OK;
}

int psp_csr_prepare(psp_csr_t *A, long long expected_products) {
  PSP_API_GUARD_H(A);
  if (!A) return fail(PSP_EINVAL, "psp_csr_prepare: NULL handle");
  if (A->host || A->multi || A->nparts) return PSP_OK;  // nothing to decide for these
  std::lock_guard<std::mutex> lk(g_extra_mu);
  g_extra[A].expected_products = expected_products < 0 ? 0 : (expected_products > 0x7fffffffffffLL ? 0x7fffffffffffLL : (long)expected_products);
  return PSP_OK;
}

int psp_csr_setup_info(psp_csr_t *A, double *info4) {
  PSP_API_GUARD_H(A);
  if (!A || !info4) return fail(PSP_EINVAL, "psp_csr_setup_info: NULL argument");
  info4[0] = info4[1] = info4[2] = info4[3] = 0.0;
  if (A->host || A->multi || A->nparts) return PSP_OK;
  std::lock_guard<std::mutex> lk(g_extra_mu);
  auto it = g_extra.find(A);
  if (it == g_extra.end()) return PSP_OK;
  info4[0] = it->second.reorder_ms;
  info4[1] = (double)it->second.products;
  info4[2] = (double)reorder_after();
  info4[3] = (double)it->second.reorder_state;
  return PSP_OK;
}

int psp_csr_kernel_info(psp_csr_t *A, char *name, int name_cap, int *info) {
  PSP_API_GUARD_H(A);
  if (!A) return fail(PSP_EINVAL, "psp_csr_kernel_info: NULL handle");
  if (A->host) {
    if (name && name_cap > 0) snprintf(name, name_cap, "cpu loops (PSP_DEVICE=cpu)");
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    return PSP_OK;
  }
  if (A->multi) {
    char buf[160];
    psp::multi_describe(A->multi, buf, sizeof buf);
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", buf);
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    return PSP_OK;
  }
  if (A->nparts) return psp_csr_kernel_info(A->parts[0], name, name_cap, info);  // every part by its own rules
  Variant v = decode_variant(A->variant);
  if (A->w4_only) v.w4 = true;
  const char *k = "csr_spmv_stream";
  int vals[4] = {0, 0, 0, 0};
  bool w4 = false;
  if (v.w4 && A->sym_owner) {
    psp_sss *S = const_cast<psp_sss *>(A->sym_owner);
    PSP_TRY(ensure_sss_w4(S));
    if (S->w4_state == 1) {
      w4 = true;
      k = "sss_spmv_w4";
      vals[0] = S->w4_nol;
    }
  }
  if (!w4 && v.w4) {
    psp::CsrExtra *ex;
    PSP_TRY(ensure_w4(A, &ex));
    if (ex->dia_state == 1) {
      w4 = true;
      k = "csr_spmv_w4";
      vals[0] = ex->dia_no;
    }
  }
  if (!w4 && A->nrows > 0 && (v.wave || v.w1) && A->max_row_nnz <= v.tile / 2) {
    k = v.wave ? "csr_spmv_wave" : "csr_spmv_w1";
    if (v.w2) {
      ChunkTable *t;
      PSP_TRY(get_chunk_table(A, v.tile, &t));
      PSP_TRY(ensure_rowoff(A, t));
      if (t->np != 0) {
        k = "csr_spmv_w2";
        if (v.w3) {
          PSP_TRY(ensure_w3(A, t));
          vals[1] = t->max_blocks;
          {
            psp::CsrExtra *exs = nullptr;
            int mode = 0;
            PSP_TRY(pick_scattered(A, t, &exs, &mode));
            if (mode == 1) {
              ChunkTable *rt;
              PSP_TRY(get_chunk_table(exs->reordered, 1024, &rt));
              k = "csr_spmv_w3_rcm";
              vals[0] = rt->nb;
              vals[1] = rt->max_blocks;
              vals[3] = t->max_blocks;  // what the stored numbering needs
            } else if (mode == 2) {
              k = "csr_spmv_w5";
              vals[0] = t->nu;
              vals[3] = t->max_cols;
            }
          }
          if (t->nb > 0) {
            k = "csr_spmv_w3";
            vals[0] = t->nb;
            if (v.sched) {
              PSP_TRY(ensure_schedule(A, t));
              vals[2] = t->sched_state == 1;
              vals[3] = t->half_band;
            }
          }
        }
        if (v.w6 && !strcmp(k, "csr_spmv_w2")) {  // what csr_spmv_launch tries in front of w2
          PSP_TRY(ensure_w6(A, t));
          if (t->nb6 > 0) {
            k = "csr_spmv_w6";
            vals[0] = t->nb6;
            vals[1] = t->max_blocks;
            vals[2] = t->direct6;
          }
        }
      }
    }
  }
  if (name && name_cap > 0) {
    strncpy(name, k, (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (info)
    for (int i = 0; i < 4; ++i) info[i] = vals[i];
  return PSP_OK;
}

int psp_csr_set_variant(psp_csr_t *A, int variant) {
  if (A && A->multi) return fail(PSP_EINVAL, "%s is not available on a multi-device matrix (psp_csr_*_multi): use matvec, jacobi, pcg, minres", "psp_csr_set_variant");
  if (!A) return fail(PSP_EINVAL, "psp_csr_set_variant: NULL handle");
  A->variant = variant;
  for (int p = 0; p < A->nparts; ++p) A->parts[p]->variant = variant;
  return PSP_OK;
}

int64_t psp_csr_device_bytes(const psp_csr_t *A) {
  if (!A || A->multi || A->host) return 0;
  if (A->nparts) {
    int64_t b = 0;
    for (int p = 0; p < A->nparts; ++p) b += psp_csr_device_bytes(A->parts[p]);
    return b;
  }
  if (A->w4_only) {
    const int64_t rows = ((int64_t)A->nrows + kDiaRows - 1) / kDiaRows * kDiaRows;
    return rows * (8 * (int64_t)A->max_row_nnz + 2);
  }
  return (int64_t)(sizeof(int) * ((size_t)A->nrows + 1) + (sizeof(int) + sizeof(double)) * A->padded);
}

// ------------------------------------------------------------------ C ABI: sss

// 0 <= col < row for every stored entry of an sss_mat's lower triangle; *bad = the smallest (row << 32 | position) that
// is not (unchanged: all are)
__global__ __launch_bounds__(256) void sss_validate_kernel(int n, const int *__restrict__ ind, const int *__restrict__ col,
                                                           unsigned long long *bad) {
  const int lane = threadIdx.x & 63;
  for (long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (long)gridDim.x * 4)
    for (int k = ind[i] + lane; k < ind[i + 1]; k += 64) {
      const int c = col[k];
      if (c < 0 || c >= i) atomicMin(bad, ((unsigned long long)(unsigned)i << 32) | (unsigned)k);
    }
}

// rows of the full mirror of an sss_mat: lower entries, the diagonal, the transposed lower triangle's row
__global__ __launch_bounds__(256) void sss_full_len_kernel(int n, const int *__restrict__ lind,
                                                           const int *__restrict__ tind, int *__restrict__ flen) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flen[i] = (lind[i + 1] - lind[i]) + 1 + (tind[i + 1] - tind[i]);
  else if (i == n) flen[i] = 0;
}

__global__ __launch_bounds__(256) void sss_full_fill_kernel(
    int n, const int *__restrict__ lind, const int *__restrict__ lcol, const double *__restrict__ lval,
    const double *__restrict__ diag, const int *__restrict__ tind, const int *__restrict__ tcol,
    const double *__restrict__ tval, const int *__restrict__ find, int *__restrict__ fcol,
    double *__restrict__ fval) {
  const int lane = threadIdx.x & 63;
  // one wave per row; grid-stride: a launch may not have 2^32 threads (n = 512^3 rows would ask for 8.6e9)
  for (long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (long)gridDim.x * 4) {
    const int l0 = lind[i], ll = lind[i + 1] - l0, t0 = tind[i], tl = tind[i + 1] - t0, f0 = find[i];
    for (int k = lane; k < ll; k += 64) {
      fcol[f0 + k] = lcol[l0 + k];
      fval[f0 + k] = lval[l0 + k];
    }
    if (lane == 0) {
      fcol[f0 + ll] = (int)i;
      fval[f0 + ll] = diag[i];
    }
    for (int k = lane; k < tl; k += 64) {
      fcol[f0 + ll + 1 + k] = tcol[t0 + k];
      fval[f0 + ll + 1 + k] = tval[t0 + k];
    }
  }
}

int psp_sss_create(int n, int nnz_lower, const int *ind_host, const int *col_host,
                   const double *val_host, const double *diag_host, psp_sss_t **out) {
  if (psp::cpu_mode()) return psp::cpu::sss_create(n, nnz_lower, ind_host, col_host, val_host, diag_host, out);
  if (!out || !ind_host || !diag_host || (nnz_lower > 0 && (!col_host || !val_host)))
    return fail(PSP_EINVAL, "psp_sss_create: NULL argument");
  if (n < 0 || nnz_lower < 0) return fail(PSP_EINVAL, "psp_sss_create: negative size");
  psp::setup_mark(nullptr);
  if (ind_host[0] != 0 || ind_host[n] != nnz_lower)
    return fail(PSP_EINVAL, "psp_sss_create: ind[0] must be 0 and ind[n] == nnz");
  for (int i = 0; i < n; ++i)
    if (ind_host[i + 1] < ind_host[i])
      return fail(PSP_EINVAL, "psp_sss_create: ind not monotone at row %d", i);
  // (the columns -- 0 <= col < row for every stored entry -- are checked on the device once they are there:
  // sss_validate_kernel below; on the host the loop over 2e7 entries was 9 ms of a 60 ms upload)
  if (2L * nnz_lower + n > 0x7fffffffL)
    return fail(PSP_EINVAL, "psp_sss_create: expanded matrix exceeds 32-bit indices");
  PSP_TRY(ensure_device());
  psp::setup_mark("sss_create: host validation");

  // Expand to the full, column-sorted CSR the device multiplies with -- on the device.  Row i receives its
  // lower entries (stored order), the diagonal, then the mirrored entries (i, r) for the rows r > i that
  // reference column i, in ascending r (= row i of the stably transposed lower triangle) -- the summation
  // order of sss_matvec (sss_mat.c:45-55).
  psp_sss *S = new psp_sss();
  S->n = n;
  S->nnz_lower = nnz_lower;
  hipError_t e1 = hipMalloc((void **)&S->ind, sizeof(int) * ((size_t)n + 1));
  hipError_t e2 = hipMalloc((void **)&S->col, sizeof(int) * (size_t)(nnz_lower ? nnz_lower : 1));
  hipError_t e3 = hipMalloc((void **)&S->val, sizeof(double) * (size_t)(nnz_lower ? nnz_lower : 1));
  hipError_t e4 = hipMalloc((void **)&S->diag, sizeof(double) * (size_t)(n ? n : 1));
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    psp_sss_destroy(S);
    return fail(PSP_ENOMEM, "psp_sss_create: device allocation failed");
  }
  psp_csr *T = nullptr, *F = nullptr;
  int *flen = nullptr;
  void *tmp = nullptr;
  int rc = PSP_OK;
  auto cleanup = [&](int code) {
    if (T) psp_csr_destroy(T);
    if (flen) (void)hipFree(flen);
    if (tmp) (void)hipFree(tmp);
    if (code != PSP_OK) {
      if (F) psp_csr_destroy(F);
      psp_sss_destroy(S);
    }
    return code;
  };
#define SSS_HIP(call)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return cleanup(fail(e_ == hipErrorOutOfMemory ? PSP_ENOMEM : PSP_ENODEV, "%s: %s", #call,        \
                          hipGetErrorString(e_)));                                                     \
  } while (0)
  // (plain copies from the caller's pageable arrays: they run at the wire's 57 GB/s once a process has made its first large
  // copy -- which costs ~160 ms whatever it copies; staging through pinned buffers filled by host threads was built and
  // measured slower, 11.5 against 2.8 ms for the 79 MB of columns: profiles/r6_config5_setup.txt)
  SSS_HIP(hipMemcpyAsync(S->ind, ind_host, sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, stream()));
  if (nnz_lower) {
    SSS_HIP(hipMemcpyAsync(S->col, col_host, sizeof(int) * (size_t)nnz_lower, hipMemcpyHostToDevice, stream()));
    SSS_HIP(hipMemcpyAsync(S->val, val_host, sizeof(double) * (size_t)nnz_lower, hipMemcpyHostToDevice, stream()));
  }
  if (n) SSS_HIP(hipMemcpyAsync(S->diag, diag_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream()));
  psp::setup_mark("sss_create: allocate + copy the arrays up");
  if (nnz_lower) {  // every entry strictly below the diagonal, before anything indexes with the columns
    unsigned long long *d_bad = nullptr, bad = ~0ull;
    SSS_HIP(hipMalloc((void **)&d_bad, sizeof(bad)));
    hipError_t ev = hipMemcpyAsync(d_bad, &bad, sizeof(bad), hipMemcpyHostToDevice, stream());
    if (ev == hipSuccess) {
      hipLaunchKernelGGL(sss_validate_kernel, dim3(std::min((n + 3) / 4, 65536)), dim3(256), 0, stream(), n, S->ind, S->col, d_bad);
      ev = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream());
    }
    if (ev == hipSuccess) ev = hipStreamSynchronize(stream());
    (void)hipFree(d_bad);
    SSS_HIP(ev);
    if (bad != ~0ull) {
      const int k = (int)(bad & 0xffffffffull);
      return cleanup(fail(PSP_EINVAL, "psp_sss_create: entry (%d,%d) is not strictly lower", (int)(bad >> 32), col_host[k]));
    }
  }
  psp::setup_mark("sss_create: validate the columns (device)");
  rc = alloc_csr(n, n, nnz_lower, &T);
  if (rc != PSP_OK) return cleanup(rc);
  psp::setup_mark("sss_create: allocate the transpose");
  rc = transpose_into(n, n, nnz_lower, S->ind, S->col, S->val, T);
  if (rc != PSP_OK) return cleanup(rc);
  psp::setup_mark("sss_create: transpose (radix sort)");
  rc = alloc_csr(n, n, 2L * nnz_lower + n, &F);
  if (rc != PSP_OK) return cleanup(rc);
  psp::setup_mark("sss_create: allocate the mirror");
  {
    SSS_HIP(hipMalloc((void **)&flen, sizeof(int) * ((size_t)n + 1)));
    hipLaunchKernelGGL(sss_full_len_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, stream(), n, S->ind, T->ind, flen);
    size_t bytes = 0;
    SSS_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, flen, F->ind, n + 1, stream()));
    SSS_HIP(hipMalloc(&tmp, bytes ? bytes : 1));
    SSS_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, flen, F->ind, n + 1, stream()));
    if (n > 0)
      hipLaunchKernelGGL(sss_full_fill_kernel, dim3(std::min((n + 3) / 4, 1 << 22)), dim3(256), 0, stream(), n, S->ind, S->col, S->val,
                         S->diag, T->ind, T->col, T->val, F->ind, F->col, F->val);
    SSS_HIP(hipGetLastError());
    SSS_HIP(hipStreamSynchronize(stream()));
  }
#undef SSS_HIP
  psp::setup_mark("sss_create: fill the mirror");
  rc = finalize_csr(F);
  if (rc != PSP_OK) return cleanup(rc);
  psp::setup_mark("sss_create: finalize_csr(mirror)");
  S->full = F;
  (void)cleanup(PSP_OK);
  psp::setup_mark("sss_create: free the scratch");
  S->full->sym_owner = S;
  *out = S;
  return PSP_OK;
}

int psp_sss_poisson(int nx, int ny, int nz, psp_sss_t **out) {
  if (!out || nx < 1 || ny < 1 || nz < 0) return fail(PSP_EINVAL, "psp_sss_poisson: bad grid");
  if (psp::cpu_mode()) return psp::cpu::sss_poisson(nx, ny, nz, out);
  const long n = (long)nx * ny * (nz > 0 ? nz : 1);
  if (n > 0x7fffffffL) return fail(PSP_EINVAL, "psp_sss_poisson: n exceeds 32-bit indices");
  const long nnzl = poisson_lower_prefix(n, nx, ny, nz);
  PSP_TRY(ensure_device());
  psp_sss *S = new psp_sss();
  S->n = (int)n;
  S->nnz_lower = (int)nnzl;
  int rc = psp_csr_poisson(nx, ny, nz, &S->full);
  if (rc != PSP_OK) {
    delete S;
    return rc;
  }
  hipError_t e1 = hipMalloc((void **)&S->ind, sizeof(int) * ((size_t)n + 1));
  hipError_t e2 = hipMalloc((void **)&S->col, sizeof(int) * (size_t)(nnzl ? nnzl : 1));
  hipError_t e3 = hipMalloc((void **)&S->val, sizeof(double) * (size_t)(nnzl ? nnzl : 1));
  hipError_t e4 = hipMalloc((void **)&S->diag, sizeof(double) * (size_t)n);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    psp_sss_destroy(S);
    return fail(PSP_ENOMEM, "psp_sss_poisson: device allocation failed");
  }
  int grid = (int)std::min<long>((n + 1 + 255) / 256, 8192);
  hipLaunchKernelGGL(poisson_sss_kernel, dim3(grid), dim3(256), 0, stream(), nx, ny, nz, n, S->ind,
                     S->col, S->val, S->diag);
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(stream()));
  S->full->sym_owner = S;
  *out = S;
  return PSP_OK;
}

int psp_sss_destroy(psp_sss_t *S) {
  if (!S) return PSP_OK;
  if (S->host) return psp::cpu::sss_destroy(S);
  psp_csr_destroy(S->full);
  if (S->w4_val) (void)hipFree(S->w4_val);
  if (S->w4_mask) (void)hipFree(S->w4_mask);
  (void)hipFree(S->ind);
  (void)hipFree(S->col);
  (void)hipFree(S->val);
  (void)hipFree(S->diag);
  delete S;
  return PSP_OK;
}

int psp_sss_shape(const psp_sss_t *S, int *n, int *nnz_reported) {
  if (!S) return fail(PSP_EINVAL, "psp_sss_shape: NULL handle");
  if (n) *n = S->n;
  if (nnz_reported) *nnz_reported = S->nnz_lower + S->n;  // sss_mat.c:155
  return PSP_OK;
}

int psp_sss_download(const psp_sss_t *S, int *ind_host, int *col_host, double *val_host,
                     double *diag_host) {
  if (!S) return fail(PSP_EINVAL, "psp_sss_download: NULL handle");
  if (S->host) return psp::cpu::sss_download(S, ind_host, col_host, val_host, diag_host);
  if (ind_host)
    PSP_HIP(hipMemcpyAsync(ind_host, S->ind, sizeof(int) * ((size_t)S->n + 1),
                           hipMemcpyDeviceToHost, stream()));
  if (col_host && S->nnz_lower)
    PSP_HIP(hipMemcpyAsync(col_host, S->col, sizeof(int) * (size_t)S->nnz_lower,
                           hipMemcpyDeviceToHost, stream()));
  if (val_host && S->nnz_lower)
    PSP_HIP(hipMemcpyAsync(val_host, S->val, sizeof(double) * (size_t)S->nnz_lower,
                           hipMemcpyDeviceToHost, stream()));
  if (diag_host && S->n)
    PSP_HIP(hipMemcpyAsync(diag_host, S->diag, sizeof(double) * (size_t)S->n,
                           hipMemcpyDeviceToHost, stream()));
  PSP_HIP(hipStreamSynchronize(stream()));
  return PSP_OK;
}

int psp_sss_getitem(const psp_sss_t *S, int i, int j, double *value) {
  if (!S || !value) return fail(PSP_EINVAL, "psp_sss_getitem: NULL argument");
  if (i < 0 || j < 0 || i >= S->n || j >= S->n)
    return fail(PSP_EINVAL, "psp_sss_getitem: indices out of range");
  if (S->host) return psp::cpu::sss_getitem(S, i, j, value);
  if (i == j) {
    PSP_HIP(hipMemcpy(value, S->diag + i, sizeof(double), hipMemcpyDeviceToHost));
    return PSP_OK;
  }
  if (i < j) std::swap(i, j);
  int lohi[2];
  PSP_HIP(hipMemcpy(lohi, S->ind + i, 2 * sizeof(int), hipMemcpyDeviceToHost));
  *value = 0.0;
  const int len = lohi[1] - lohi[0];
  if (len > 0) {
    std::vector<int> c((size_t)len);
    PSP_HIP(hipMemcpy(c.data(), S->col + lohi[0], sizeof(int) * (size_t)len, hipMemcpyDeviceToHost));
    for (int k = 0; k < len; ++k)
      if (c[k] == j) {
        PSP_HIP(hipMemcpy(value, S->val + lohi[0] + k, sizeof(double), hipMemcpyDeviceToHost));
        break;
      }
  }
  return PSP_OK;
}

int psp_sss_matvec_dev(psp_sss_t *S, const double *x_dev, double *y_dev) {
  PSP_API_GUARD_H(S);
  if (!S) return fail(PSP_EINVAL, "psp_sss_matvec_dev: NULL handle");
  return psp_csr_matvec_dev(S->full, x_dev, y_dev);
}

int psp_sss_matvec_stride(psp_sss_t *S, const double *x_host, ptrdiff_t incx, double *y_host,
                          ptrdiff_t incy) {
  PSP_API_GUARD_H(S);
  if (!S) return fail(PSP_EINVAL, "psp_sss_matvec: NULL handle");
  if (S->host) return (x_host && y_host) ? psp::cpu::sss_matvec(S, x_host, incx, y_host, incy)
                                         : fail(PSP_EINVAL, "psp_sss_matvec: NULL argument");
  return psp_csr_matvec_stride(S->full, x_host, incx, y_host, incy);
}

int psp_sss_matvec(psp_sss_t *S, const double *x_host, double *y_host) {
  PSP_API_GUARD_H(S);
  return psp_sss_matvec_stride(S, x_host, 1, y_host, 1);
}

int psp_sss_kernel_info(psp_sss_t *S, char *name, int name_cap, int *info) {
  PSP_API_GUARD_H(S);
  if (!S) return fail(PSP_EINVAL, "psp_sss_kernel_info: NULL handle");
  if (S->host) {
    if (name && name_cap > 0) snprintf(name, name_cap, "cpu loops (PSP_DEVICE=cpu)");
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    return PSP_OK;
  }
  return psp_csr_kernel_info(S->full, name, name_cap, info);
}

int psp_sss_prepare(psp_sss_t *S, long long expected_products) {
  PSP_API_GUARD_H(S);
  if (!S) return fail(PSP_EINVAL, "psp_sss_prepare: NULL handle");
  if (S->host) return PSP_OK;
  return psp_csr_prepare(S->full, expected_products);
}

int psp_sss_setup_info(psp_sss_t *S, double *info4) {
  PSP_API_GUARD_H(S);
  if (!S || !info4) return fail(PSP_EINVAL, "psp_sss_setup_info: NULL argument");
  if (S->host) {
    info4[0] = info4[1] = info4[2] = info4[3] = 0.0;
    return PSP_OK;
  }
  return psp_csr_setup_info(S->full, info4);
}

int psp_sss_set_variant(psp_sss_t *S, int variant) {
  if (!S) return fail(PSP_EINVAL, "psp_sss_set_variant: NULL handle");
  return psp_csr_set_variant(S->full, variant);
}

int64_t psp_sss_device_bytes(const psp_sss_t *S) {
  if (!S || S->host) return 0;
  return psp_csr_device_bytes(S->full) + (int64_t)sizeof(int) * (S->n + 1) +
         (int64_t)(sizeof(int) + sizeof(double)) * S->nnz_lower + (int64_t)sizeof(double) * S->n;
}

}  // extern "C"
