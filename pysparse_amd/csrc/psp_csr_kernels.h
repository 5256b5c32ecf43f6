// psp_csr_kernels.h -- a FRAGMENT of psp_csr.hip (included there, in this order, into one translation unit; not a header of its
// own): the SpMV kernels: csr_spmv_stream / wave / w1 / w2 / w3 / w6 / w5, the index-free family (csr_spmv_w4, w4x, w4y,
// sss_spmv_w4, csr_spmv_w4_pf, csr_spmv_w4_transp), the kernels that build their tables, fold / transpose / diagonal kernels.

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

// the longest column (one thread sorts a column: the counting form is for columns of a few thousand entries at most)
__global__ void transp_maxlen_kernel(int ncols, const int *__restrict__ tind, int *maxlen) {
  int m = 0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x) m = max(m, tind[c + 1] - tind[c]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(maxlen, m);
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

