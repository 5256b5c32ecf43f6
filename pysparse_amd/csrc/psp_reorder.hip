// psp_reorder.hip -- bandwidth-reducing renumbering for irregular csr_mat / sss_mat operators.
//
// csr_spmv_w3 (x staged in LDS, 16-bit chunk-local columns, psp_csr.hip) needs every chunk of ~1000
// nonzeros to reference at most 64 blocks of 16 consecutive x entries.  An unstructured numbering
// (FEM meshes as they come out of a mesh generator, shuffled node ids) breaks that: the same matrix
// references 100-140 blocks per chunk and falls back to csr_spmv_w2, whose x gathers are bound by the
// per-CU L1 (0.61-0.67 of the HBM roofline, profiles/r1_fem_standin.txt).  This file computes a reverse
// Cuthill-McKee numbering once per handle -- on the device (reorder_rcm_device: 36-48 ms to the first product at
// n = 9.3e5, 41 M nonzeros; an unsymmetric pattern is symmetrised on the device first), with the host code it
// replaced (reorder_rcm_host: 1.25 s for the same matrix; same rules, same permutation) as the fallback for graphs
// with very many levels -- and builds the symmetrically permuted matrix R = P A P^T as a second device handle:
//     row i of R   = row perm[i] of A, entries in A's stored order (NOT re-sorted: the reference adds a
//                    row's products left to right, csr_mat.c:49-54, and so must we -- same products,
//                    same order, same bits in y);
//     column ids   = inv[col].
// The product y = A x then runs as  xp = x[perm] (gather pass, 20 n bytes)  ->  yp = R xp (csr_spmv_w3,
// coalesced stores)  ->  y[j] = yp[inv[j]] (a second gather pass; storing row i's sum straight to
// y[perm[i]] from the SpMV -- 16 partial writes per 128-byte line of y, from different XCDs -- measured
// no faster: 0.0987 vs 0.0967 ms at n = 9.3e5).  Nothing here changes a bit of y, and the fused dot product
// of the PCG path is formed by the last pass in the caller's numbering, so it too has the bits every other
// kernel gives.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "psp_internal.h"

using namespace psp;

namespace {

// ---- reverse Cuthill-McKee on the pattern of A + A^T (host)

struct Graph {
  std::vector<long> ptr;
  std::vector<int> adj;
  std::vector<int> deg;
  // HUBS: rows with more than hub_thr neighbours (constraint rows, a few long-range couplings) are left out of the
  // level structure -- their edges are short cuts through the mesh that blow up the level sets and with them the
  // band -- and are numbered last (first after the reversal).  No more than 1 % of the rows, else nobody is a hub.
  int hub_thr = 0x7fffffff;
  bool hub(int v) const { return deg[v] > hub_thr; }
};

// the threshold both implementations use: twice the average degree, at least 32
inline int hub_threshold(long long deg_sum, int n) {
  const long long avg = n > 0 ? (deg_sum + n - 1) / n : 0;
  const long long t = 2 * avg;
  return (int)(t < 32 ? 32 : (t > 0x7ffffffe ? 0x7ffffffe : t));
}

Graph symmetric_pattern(int n, const int *ind, const int *col) {
  Graph g;
  std::vector<long> cnt((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i)
    for (int k = ind[i]; k < ind[i + 1]; ++k) {
      const int j = col[k];
      if (j == i || j < 0 || j >= n) continue;
      ++cnt[(size_t)i + 1];
      ++cnt[(size_t)j + 1];
    }
  for (int i = 0; i < n; ++i) cnt[(size_t)i + 1] += cnt[i];
  std::vector<int> tmp((size_t)cnt[n]);
  std::vector<long> fill(cnt.begin(), cnt.end() - 1);
  for (int i = 0; i < n; ++i)
    for (int k = ind[i]; k < ind[i + 1]; ++k) {
      const int j = col[k];
      if (j == i || j < 0 || j >= n) continue;
      tmp[(size_t)fill[i]++] = j;
      tmp[(size_t)fill[j]++] = i;
    }
  // sort + unique every list ((i,j) and (j,i) are both stored in a structurally symmetric matrix)
  g.ptr.assign((size_t)n + 1, 0);
  g.adj.resize(tmp.size());
  long out = 0;
  for (int i = 0; i < n; ++i) {
    int *b = tmp.data() + cnt[i], *e = tmp.data() + cnt[(size_t)i + 1];
    std::sort(b, e);
    e = std::unique(b, e);
    g.ptr[i] = out;
    for (int *p = b; p != e; ++p) g.adj[(size_t)out++] = *p;
  }
  g.ptr[n] = out;
  g.adj.resize((size_t)out);
  g.deg.resize(n);
  long long sum = 0;
  for (int i = 0; i < n; ++i) {
    g.deg[i] = (int)(g.ptr[(size_t)i + 1] - g.ptr[i]);
    sum += g.deg[i];
  }
  const int thr = hub_threshold(sum, n);
  long hubs = 0;
  for (int i = 0; i < n; ++i) hubs += g.deg[i] > thr;
  if (hubs * 100 <= (long)n) g.hub_thr = thr;
  return g;
}

// breadth-first level structure rooted at `root` inside the component marked by comp_id in `mark`;
// returns the number of levels, the last level's nodes in `last`
int bfs_levels(const Graph &g, int root, std::vector<int> &level, std::vector<int> &queue, std::vector<int> &last,
               int stamp, std::vector<int> &seen) {
  queue.clear();
  queue.push_back(root);
  seen[root] = stamp;
  level[root] = 0;
  size_t head = 0;
  int nlev = 1;
  while (head < queue.size()) {
    const int u = queue[head++];
    for (long k = g.ptr[u]; k < g.ptr[(size_t)u + 1]; ++k) {
      const int v = g.adj[(size_t)k];
      if (seen[v] != stamp && !g.hub(v)) {
        seen[v] = stamp;
        level[v] = level[u] + 1;
        nlev = level[v] + 1;
        queue.push_back(v);
      }
    }
  }
  last.clear();
  for (size_t i = queue.size(); i-- > 0;) {
    if (level[queue[i]] != nlev - 1) break;
    last.push_back(queue[i]);
  }
  return nlev;
}

// perm[new] = old
std::vector<int> rcm_order(const Graph &g, int n) {
  std::vector<int> order;
  order.reserve(n);
  std::vector<char> placed((size_t)n, 0);
  std::vector<int> level((size_t)n, 0), queue, last, seen((size_t)n, 0), nbrs;
  int stamp = 0;
  // components in order of their lowest-numbered node; start nodes of minimal degree
  for (int s = 0; s < n; ++s) {
    if (placed[s] || g.hub(s)) continue;
    // pseudo-peripheral node (George & Liu): walk to a minimum-degree node of the last level while the
    // eccentricity grows
    int root = s;
    int nlev = bfs_levels(g, root, level, queue, last, ++stamp, seen);
    {  // lowest (degree, id) node of this component as the first guess
      int best = root;
      for (int u : queue)
        if (g.deg[u] < g.deg[best] || (g.deg[u] == g.deg[best] && u < best)) best = u;
      if (best != root) {
        root = best;
        nlev = bfs_levels(g, root, level, queue, last, ++stamp, seen);
      }
    }
    for (int iter = 0; iter < 8; ++iter) {
      int cand = last[0];
      for (int u : last)
        if (g.deg[u] < g.deg[cand] || (g.deg[u] == g.deg[cand] && u < cand)) cand = u;
      std::vector<int> q2, last2;
      const int nlev2 = bfs_levels(g, cand, level, q2, last2, ++stamp, seen);
      if (nlev2 <= nlev) break;
      root = cand;
      nlev = nlev2;
      last.swap(last2);
    }
    // Cuthill-McKee from root: neighbours by ascending degree
    const size_t first = order.size();
    order.push_back(root);
    placed[root] = 1;
    size_t head = first;
    while (head < order.size()) {
      const int u = order[head++];
      nbrs.clear();
      for (long k = g.ptr[u]; k < g.ptr[(size_t)u + 1]; ++k) {
        const int v = g.adj[(size_t)k];
        if (!placed[v] && !g.hub(v)) {
          placed[v] = 1;
          nbrs.push_back(v);
        }
      }
      std::sort(nbrs.begin(), nbrs.end(), [&](int a, int b) {
        return g.deg[a] != g.deg[b] ? g.deg[a] < g.deg[b] : a < b;
      });
      order.insert(order.end(), nbrs.begin(), nbrs.end());
    }
  }
  for (int v = 0; v < n; ++v)  // the hubs, by id
    if (g.hub(v)) order.push_back(v);
  std::reverse(order.begin(), order.end());
  return order;
}

// Both passes are latency-bound (perm -> x is a dependent pair of loads, the vectors are a few MB): each
// thread takes 4 elements with all index loads, then all value loads, in flight together, so that the
// whole pass is ONE residency round of the chip instead of three (7.7 us -> ~4 us at n = 9.3e5).
constexpr int kPermPerThread = 4;

__global__ __launch_bounds__(256) void permute_gather_kernel(int n, const int *__restrict__ perm,
                                                             const double *__restrict__ x, double *__restrict__ xp,
                                                             const int *__restrict__ skip) {
  if (skip && *skip) return;
  const long base = (long)blockIdx.x * (256 * kPermPerThread) + threadIdx.x;
  int idx[kPermPerThread];
  double v[kPermPerThread];
#pragma unroll
  for (int u = 0; u < kPermPerThread; ++u) {
    const long i = base + u * 256;
    idx[u] = i < n ? perm[i] : 0;
  }
#pragma unroll
  for (int u = 0; u < kPermPerThread; ++u) v[u] = x[idx[u]];
#pragma unroll
  for (int u = 0; u < kPermPerThread; ++u) {
    const long i = base + u * 256;
    if (i < n) xp[i] = v[u];
  }
}

// dst[idx[i]] = src[i]: the same two permutations written as scatters.  The index and the value are loaded side by
// side (one round trip instead of the two dependent ones of the gather form) and the 8-byte stores leave the kernel
// without being waited for: a pass of n = 9.3e5 is a single wave lifetime either way, and this one is shorter.
__global__ __launch_bounds__(256) void permute_scatter_kernel(int n, const int *__restrict__ idx,
                                                              const double *__restrict__ src, double *__restrict__ dst,
                                                              const int *__restrict__ skip) {
  if (skip && *skip) return;
  const long base = (long)blockIdx.x * (256 * kPermPerThread) + threadIdx.x;
  int id[kPermPerThread];
  double v[kPermPerThread];
#pragma unroll
  for (int u = 0; u < kPermPerThread; ++u) {
    const long i = base + u * 256;
    id[u] = i < n ? idx[i] : -1;
    v[u] = i < n ? src[i] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < kPermPerThread; ++u)
    if (id[u] >= 0) dst[id[u]] = v[u];
}

// y[j] = yp[inv[j]] and, optionally, one partial sum of dotv[j] * y[j] per workgroup (fixed order)
__global__ __launch_bounds__(256) void permute_back_kernel(int n, const int *__restrict__ inv,
                                                           const double *__restrict__ yp, double *__restrict__ y,
                                                           const double *__restrict__ dotv,
                                                           double *__restrict__ partials,
                                                           const int *__restrict__ skip) {
  if (skip && *skip) return;
  __shared__ double red[4];
  double dsum = 0.0;
  const long base = (long)blockIdx.x * (256 * kPermPerThread) + threadIdx.x;
  int idx[kPermPerThread];
  double v[kPermPerThread], d[kPermPerThread];
#pragma unroll
  for (int u = 0; u < kPermPerThread; ++u) {
    const long j = base + u * 256;
    idx[u] = j < n ? inv[j] : 0;
    d[u] = (dotv && j < n) ? dotv[j] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < kPermPerThread; ++u) v[u] = yp[idx[u]];
#pragma unroll
  for (int u = 0; u < kPermPerThread; ++u) {
    const long j = base + u * 256;
    if (j < n) {
      y[j] = v[u];
      dsum += d[u] * v[u];
    }
  }
  if (partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dsum += __shfl_down(dsum, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}


// ---- the same numbering computed on the device (structurally symmetric patterns with ascending rows: the full
// mirror of an sss_mat, a symmetric ll_mat's to_csr()).  Level-synchronous Cuthill-McKee: the nodes discovered from
// level l are sorted by (position of their first parent in the order, degree, id) -- exactly the order in which
// the host loop above appends them -- so both produce the SAME permutation (tests/test_gpu_spmv.py compares them).
// Every choice is an order-independent minimum or a stable sort of an id-ordered list: no atomics decide an order.

constexpr int kRcmInf = 0x7f7f7f7f;         // "no parent yet": what hipMemset(0x7f) writes
constexpr int kRcmMaxLevels = 16384;     // levels walked per handle before the device path gives up
constexpr int kRcmMaxComponents = 256;   // connected components likewise
constexpr int kRcmBatch = 32;            // BFS levels enqueued between two looks at the result

// deg[i] = entries of row i off the diagonal; *ok = 0 when a row is not strictly ascending or an entry has no
// mirror; *maxdeg = largest degree
__global__ __launch_bounds__(256) void rcm_deg_kernel(int n, const int *__restrict__ ind, const int *__restrict__ col,
                                                      int *__restrict__ deg, int *__restrict__ ok,
                                                      int *__restrict__ maxdeg,
                                                      unsigned long long *__restrict__ degsum) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int d = 0;
  bool good = true;
  int prev = -1;
  for (int k = i < n ? ind[i] : 0; k < (i < n ? ind[i + 1] : 0); ++k) {
    const int j = col[k];
    if (j <= prev || j >= n) {
      good = false;
      break;
    }
    prev = j;
    if (j == i) continue;
    ++d;
    int lo = ind[j], hi = ind[j + 1];  // is (j, i) stored?
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (col[mid] < i) lo = mid + 1;
      else hi = mid;
    }
    if (lo >= ind[j + 1] || col[lo] != i) good = false;
  }
  if (i < n) {
    deg[i] = d;
    if (!good) *ok = 0;
    atomicMax(maxdeg, d);
  }
  // sum of the degrees, one atomic per wave
  unsigned long long sum = (unsigned long long)d;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(degsum, sum);
}

__global__ __launch_bounds__(256) void rcm_count_hubs_kernel(int n, const int *__restrict__ deg, int thr,
                                                             int *__restrict__ count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long m = __ballot(i < n && deg[i] > thr);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

// level[v] = -2 (placed in an earlier component) or -1 (not reached yet)
// (hubs -- more than thr neighbours, see struct Graph -- are never reached: they look placed)
__global__ __launch_bounds__(256) void rcm_reset_kernel(int n, const int *__restrict__ pos, const int *__restrict__ deg,
                                                        int thr, int *__restrict__ level) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v < n) level[v] = (pos[v] >= 0 || deg[v] > thr) ? -2 : -1;
}

// one breadth-first level: the nodes of level `cur` give their unreached neighbours level cur + 1
__global__ __launch_bounds__(256) void rcm_bfs_level_kernel(int n, const int *__restrict__ ind,
                                                            const int *__restrict__ col, int *level, int cur,
                                                            int *__restrict__ found) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= n || level[u] != cur) return;
  bool any = false;
  for (int k = ind[u]; k < ind[u + 1]; ++k) {
    const int v = col[k];
    if (level[v] == -1) {
      level[v] = cur + 1;  // every writer stores the same value
      any = true;
    }
  }
  if (any) found[cur + 1] = 1;
}

// smallest (key[v], v) over the nodes with level[v] == want, or level[v] >= 0 when want == -3; key may be null
// (then the smallest v)
__global__ __launch_bounds__(256) void rcm_argmin_kernel(int n, const int *__restrict__ level, int want,
                                                         const int *__restrict__ key,
                                                         unsigned long long *__restrict__ out) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  unsigned long long best = ~0ull;
  if (v < n) {
    const int l = level[v];
    const bool in = want == -3 ? l >= 0 : l == want;
    if (in) best = ((unsigned long long)(unsigned)(key ? key[v] : 0) << 32) | (unsigned)v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_down(best, off, 64);
    best = o < best ? o : best;
  }
  if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(out, best);
}

// Cuthill-McKee, one level: positions [lo, hi) of `order` offer themselves as parent to their unplaced neighbours
__global__ __launch_bounds__(256) void rcm_expand_kernel(int lo, int hi, const int *__restrict__ order,
                                                         const int *__restrict__ ind, const int *__restrict__ col,
                                                         const int *__restrict__ pos, const int *__restrict__ deg,
                                                         int thr, int *parent) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int p = lo + (t >> 3), sub = t & 7;
  if (p >= hi) return;
  const int u = order[p];
  for (int k = ind[u] + sub; k < ind[u + 1]; k += 8) {
    const int v = col[k];
    if (pos[v] < 0 && deg[v] <= thr) atomicMin(parent + v, p);
  }
}

struct RcmHub {
  const int *deg;
  int thr;
  __device__ bool operator()(int v) const { return deg[v] > thr; }
};

struct RcmDiscovered {
  const int *pos, *parent;
  __device__ bool operator()(int v) const { return pos[v] < 0 && parent[v] != kRcmInf; }
};

__global__ __launch_bounds__(256) void rcm_keys_kernel(int m, const int *__restrict__ cand,
                                                       const int *__restrict__ parent, const int *__restrict__ deg,
                                                       int degbits, unsigned long long *__restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) {
    const int v = cand[i];
    keys[i] = ((unsigned long long)(unsigned)parent[v] << degbits) | (unsigned)deg[v];
  }
}

__global__ __launch_bounds__(256) void rcm_assign_kernel(int m, int hi, const int *__restrict__ sorted,
                                                         int *__restrict__ pos, int *__restrict__ order) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) {
    const int v = sorted[i];
    pos[v] = hi + i;
    order[hi + i] = v;
  }
}

__global__ __launch_bounds__(256) void rcm_place_root_kernel(int root, int at, int *pos, int *order) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    pos[root] = at;
    order[at] = root;
  }
}

// perm[i] = order[n - 1 - i] (the "reverse" of reverse Cuthill-McKee), inv = its inverse
__global__ __launch_bounds__(256) void rcm_finish_kernel(int n, const int *__restrict__ order, int *__restrict__ perm,
                                                         int *__restrict__ inv) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int o = order[n - 1 - i];
    perm[i] = o;
    inv[o] = i;
  }
}

// ---- R = P A P^T on the device: row i of R = row perm[i] of A in its stored order, columns through inv
__global__ __launch_bounds__(256) void rcm_rowlen_kernel(int n, const int *__restrict__ ind,
                                                         const int *__restrict__ perm, int *__restrict__ rlen) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int o = perm[i];
    rlen[i] = ind[o + 1] - ind[o];
  } else if (i == n) {
    rlen[i] = 0;
  }
}

__global__ __launch_bounds__(256) void rcm_copy_rows_kernel(int n, const int *__restrict__ ind,
                                                            const int *__restrict__ col,
                                                            const double *__restrict__ val,
                                                            const int *__restrict__ perm, const int *__restrict__ inv,
                                                            const int *__restrict__ rind, int *__restrict__ rcol,
                                                            double *__restrict__ rval) {
  const int lane = threadIdx.x & 63;
  // one wave per row; grid-stride: a launch may not have 2^32 threads
  for (long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (long)gridDim.x * 4) {
    const int o = perm[i];
    const int src = ind[o], len = ind[o + 1] - src, dst = rind[i];
    for (int k = lane; k < len; k += 64) {
      rcol[dst + k] = inv[col[src + k]];
      rval[dst + k] = val[src + k];
    }
  }
}

// ---- pattern of A + A^T on the device (unsymmetric patterns, unsorted rows): every entry (r, c), c != r, gives the
// keys r*2^32 + c and c*2^32 + r; sorted and made unique they are the rows of the symmetrised pattern, ascending
__global__ __launch_bounds__(256) void sym_keys_kernel(int n, const int *__restrict__ ind, const int *__restrict__ col,
                                                       unsigned long long *__restrict__ keys) {
  const int lane = threadIdx.x & 63;
  // one wave per row; grid-stride: a launch may not have 2^32 threads
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (long)gridDim.x * 4) {
    for (int k = ind[r] + lane; k < ind[r + 1]; k += 64) {
      const int c = col[k];
      // entries on the diagonal (and out-of-range columns, which a square operator does not have) map to the
      // all-ones key, which sorts last and is dropped
      const bool keep = c != r && c >= 0 && c < n;
      keys[2 * (size_t)k] = keep ? ((unsigned long long)(unsigned)r << 32) | (unsigned)c : ~0ull;
      keys[2 * (size_t)k + 1] = keep ? ((unsigned long long)(unsigned)c << 32) | (unsigned)r : ~0ull;
    }
  }
}

// from the sorted unique keys: scol[i] and the row offsets (rows without entries included)
__global__ __launch_bounds__(256) void sym_rows_kernel(long m, int n, const unsigned long long *__restrict__ keys,
                                                       int *__restrict__ sind, int *__restrict__ scol) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i > m) return;
  const int lo = i == 0 ? 0 : (int)(keys[i - 1] >> 32) + 1;  // rows (prev, cur] start at i
  const int hi = i == m ? n : (int)(keys[i] >> 32);
  for (int r = lo; r <= hi; ++r) sind[r] = (int)i;
  if (i < m) scol[i] = (int)(keys[i] & 0xffffffffull);
}

// (Round 6 measured the whole numbering as ONE cooperative kernel -- grid barriers instead of launches, children ranked
// among their siblings in LDS instead of a radix sort per level: 22 ms against the 12 ms of the launch-per-level loop
// below at n = 9.3e5 in a warm process.  What crosses workgroups there must be read with device-coherent loads, each a
// 2-3 us round trip, and a lane's dozen of them per level are dependent; the per-level kernels read through the caches
// with n threads in flight.  Backed out; profiles/r6_config5_setup.txt.)
struct DevBuf {  // frees on scope exit
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <class T>
  T *as() const {
    return static_cast<T *>(p);
  }
};

}  // namespace

namespace psp {

// The numbering on the device.  *status: 1 = perm_dev / inv_dev hold it (n ints each, hipMalloc'ed, the caller
// frees); -1 = the pattern is not structurally symmetric with ascending rows (reorder_symmetrize_device gives one
// that is); 0 = not done -- the graph has more levels / components than the device loop is willing to walk (one
// host look per level): the caller takes the host path above.
int reorder_rcm_device(int n, const int *ind, const int *col, int **perm_dev, int **inv_dev, int *status) {
  *status = 0;
  *perm_dev = *inv_dev = nullptr;
  if (n <= 0) return PSP_OK;
  hipStream_t st = stream();
  const int g = (n + 255) / 256;
  DevBuf b_deg, b_level, b_pos, b_order, b_parent, b_cand, b_sorted, b_keys, b_keys2, b_found, b_small, b_tmp;
  PSP_HIP(b_deg.alloc(sizeof(int) * (size_t)n));
  PSP_HIP(b_level.alloc(sizeof(int) * (size_t)n));
  PSP_HIP(b_pos.alloc(sizeof(int) * (size_t)n));
  PSP_HIP(b_order.alloc(sizeof(int) * (size_t)n));
  PSP_HIP(b_parent.alloc(sizeof(int) * (size_t)n));
  PSP_HIP(b_cand.alloc(sizeof(int) * (size_t)n));
  PSP_HIP(b_sorted.alloc(sizeof(int) * (size_t)n));
  PSP_HIP(b_keys.alloc(sizeof(unsigned long long) * (size_t)n));
  PSP_HIP(b_keys2.alloc(sizeof(unsigned long long) * (size_t)n));
  PSP_HIP(b_found.alloc(sizeof(int) * (size_t)(kRcmMaxLevels + 2 * kRcmBatch + 2)));
  PSP_HIP(b_small.alloc(64));
  int *deg = b_deg.as<int>(), *level = b_level.as<int>(), *pos = b_pos.as<int>(), *order = b_order.as<int>();
  int *parent = b_parent.as<int>(), *cand = b_cand.as<int>(), *sorted = b_sorted.as<int>();
  unsigned long long *keys = b_keys.as<unsigned long long>(), *keys2 = b_keys2.as<unsigned long long>();
  int *found = b_found.as<int>();
  int *d_ok = b_small.as<int>(), *d_maxdeg = d_ok + 1, *d_count = d_ok + 2;
  unsigned long long *d_min = reinterpret_cast<unsigned long long *>(d_ok + 4);
  unsigned long long *d_degsum = reinterpret_cast<unsigned long long *>(d_ok + 6);

  setup_mark("rcm: scratch allocations");
  // degrees + the symmetry / ordering check
  {
    const int init[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    PSP_HIP(hipMemcpyAsync(d_ok, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rcm_deg_kernel, dim3(g), dim3(256), 0, st, n, ind, col, deg, d_ok, d_maxdeg, d_degsum);
    PSP_LAUNCH_CHECK();
    int res[8];
    PSP_HIP(hipMemcpyAsync(res, d_ok, sizeof(res), hipMemcpyDeviceToHost, st));
    PSP_HIP(hipStreamSynchronize(st));
    setup_mark("rcm: degrees + symmetry check (rcm_deg_kernel)");
    if (!res[0]) {
      *status = -1;  // not structurally symmetric with ascending rows: reorder_symmetrize_device first
      return PSP_OK;
    }
    // hubs (see struct Graph): the same threshold and the same 1 % rule as the host code
    long long degsum = 0;
    memcpy(&degsum, res + 6, sizeof(degsum));
    int thr = hub_threshold(degsum, n);
    {
      PSP_HIP(hipMemsetAsync(d_count, 0, sizeof(int), st));
      hipLaunchKernelGGL(rcm_count_hubs_kernel, dim3(g), dim3(256), 0, st, n, deg, thr, d_count);
      int hubs = 0;
      PSP_HIP(hipMemcpyAsync(&hubs, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
      PSP_HIP(hipStreamSynchronize(st));
      if ((long)hubs * 100 > (long)n) thr = 0x7fffffff;
    }
    setup_mark("rcm: hubs");
    int degbits = 1, posbits = 1;
    while ((1L << degbits) <= res[1]) ++degbits;
    while ((1L << posbits) < n) ++posbits;
    if (degbits + posbits > 62) return PSP_OK;
    // scratch of the two hipcub primitives, sized for n items once
    size_t bytes_sel = 0, bytes_sort = 0;
    RcmDiscovered pred{pos, parent};
    hipcub::CountingInputIterator<int> ids(0);
    PSP_HIP(hipcub::DeviceSelect::If(nullptr, bytes_sel, ids, cand, d_count, n, pred, st));
    PSP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes_sort, keys, keys2, cand, sorted, n, 0,
                                               degbits + posbits, st));
    const size_t bytes_tmp = std::max(bytes_sel, bytes_sort);
    PSP_HIP(b_tmp.alloc(bytes_tmp));

    PSP_HIP(hipMemsetAsync(pos, 0xff, sizeof(int) * (size_t)n, st));      // -1: not placed
    PSP_HIP(hipMemsetAsync(parent, 0x7f, sizeof(int) * (size_t)n, st));   // kRcmInf: no parent yet
    pred = RcmDiscovered{pos, parent};
    setup_mark("rcm: degrees, symmetry check, hubs, sort scratch");

    int levels_walked = 0;
    // breadth-first level structure from `root` inside the unplaced part; returns the number of levels (0 = gave up)
    auto bfs = [&](int root, int *nlev) -> int {
      hipLaunchKernelGGL(rcm_reset_kernel, dim3(g), dim3(256), 0, st, n, pos, deg, thr, level);
      PSP_HIP(hipMemsetAsync(level + root, 0, sizeof(int), st));
      PSP_HIP(hipMemsetAsync(found, 0, sizeof(int) * (size_t)(kRcmMaxLevels + 2 * kRcmBatch + 2), st));
      int cur = 0;
      int flags[kRcmBatch];
      for (;;) {
        if (cur + kRcmBatch > kRcmMaxLevels || levels_walked > 4 * kRcmMaxLevels) {
          *nlev = 0;
          return PSP_OK;
        }
        for (int b = 0; b < kRcmBatch; ++b)
          hipLaunchKernelGGL(rcm_bfs_level_kernel, dim3(g), dim3(256), 0, st, n, ind, col, level, cur + b, found);
        PSP_LAUNCH_CHECK();
        PSP_HIP(hipMemcpyAsync(flags, found + cur + 1, sizeof(flags), hipMemcpyDeviceToHost, st));
        PSP_HIP(hipStreamSynchronize(st));
        int b = 0;
        while (b < kRcmBatch && flags[b]) ++b;
        levels_walked += b + 1;
        if (b < kRcmBatch) {  // level cur + b + 1 is empty: levels 0 .. cur + b exist
          *nlev = cur + b + 1;
          return PSP_OK;
        }
        cur += kRcmBatch;
      }
    };
    auto argmin = [&](int want, const int *key, int *node) -> int {
      PSP_HIP(hipMemsetAsync(d_min, 0xff, sizeof(unsigned long long), st));
      hipLaunchKernelGGL(rcm_argmin_kernel, dim3(g), dim3(256), 0, st, n, level, want, key, d_min);
      PSP_LAUNCH_CHECK();
      unsigned long long r;
      PSP_HIP(hipMemcpyAsync(&r, d_min, sizeof(r), hipMemcpyDeviceToHost, st));
      PSP_HIP(hipStreamSynchronize(st));
      *node = r == ~0ull ? -1 : (int)(unsigned)(r & 0xffffffffull);
      return PSP_OK;
    };

    int placed = 0, comps = 0;
    for (;;) {
      // lowest-numbered node that is neither placed nor a hub
      int s = -1;
      hipLaunchKernelGGL(rcm_reset_kernel, dim3(g), dim3(256), 0, st, n, pos, deg, thr, level);
      PSP_TRY(argmin(-1, nullptr, &s));
      if (s < 0) break;
      if (++comps > kRcmMaxComponents) return PSP_OK;
      // pseudo-peripheral root (George & Liu), as rcm_order above
      int root = s, nlev = 0;
      PSP_TRY(bfs(root, &nlev));
      if (!nlev) return PSP_OK;
      int best = -1;
      PSP_TRY(argmin(-3, deg, &best));
      if (best != root) {
        root = best;
        PSP_TRY(bfs(root, &nlev));
        if (!nlev) return PSP_OK;
      }
      for (int iter = 0; iter < 8; ++iter) {
        int c = -1, nlev2 = 0;
        PSP_TRY(argmin(nlev - 1, deg, &c));
        PSP_TRY(bfs(c, &nlev2));
        if (!nlev2) return PSP_OK;
        if (nlev2 <= nlev) break;
        root = c;
        nlev = nlev2;
      }
      // Cuthill-McKee from root, level by level
      hipLaunchKernelGGL(rcm_place_root_kernel, dim3(1), dim3(64), 0, st, root, placed, pos, order);
      int lo = placed, hi = placed + 1;
      for (;;) {
        if (++levels_walked > 5 * kRcmMaxLevels) return PSP_OK;
        const int nf = hi - lo;
        hipLaunchKernelGGL(rcm_expand_kernel, dim3((int)(((long)nf * 8 + 255) / 256)), dim3(256), 0, st, lo, hi, order,
                           ind, col, pos, deg, thr, parent);
        size_t tb = bytes_tmp;
        PSP_HIP(hipcub::DeviceSelect::If(b_tmp.p, tb, ids, cand, d_count, n, pred, st));
        int m = 0;
        PSP_HIP(hipMemcpyAsync(&m, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
        PSP_HIP(hipStreamSynchronize(st));
        if (m == 0) break;
        const int gm = (m + 255) / 256;
        hipLaunchKernelGGL(rcm_keys_kernel, dim3(gm), dim3(256), 0, st, m, cand, parent, deg, degbits, keys);
        tb = bytes_tmp;
        PSP_HIP(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, tb, keys, keys2, cand, sorted, m, 0, degbits + posbits, st));
        hipLaunchKernelGGL(rcm_assign_kernel, dim3(gm), dim3(256), 0, st, m, hi, sorted, pos, order);
        PSP_LAUNCH_CHECK();
        lo = hi;
        hi += m;
      }
      placed = hi;
    }
    setup_mark("rcm: numbering (launch per level)");
    if (placed < n) {  // the hubs, by id, behind everybody else
      size_t tb = bytes_tmp;
      RcmHub is_hub{deg, thr};
      PSP_HIP(hipcub::DeviceSelect::If(b_tmp.p, tb, ids, order + placed, d_count, n, is_hub, st));
      int m = 0;
      PSP_HIP(hipMemcpyAsync(&m, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
      PSP_HIP(hipStreamSynchronize(st));
      if (placed + m != n) return fail(PSP_EINVAL, "reorder: %d placed + %d hubs != %d rows", placed, m, n);
    }
  }
  int *perm = nullptr, *inv = nullptr;
  if (hipMalloc((void **)&perm, sizeof(int) * (size_t)n) != hipSuccess ||
      hipMalloc((void **)&inv, sizeof(int) * (size_t)n) != hipSuccess) {
    if (perm) (void)hipFree(perm);
    (void)hipGetLastError();
    return PSP_OK;
  }
  hipLaunchKernelGGL(rcm_finish_kernel, dim3(g), dim3(256), 0, st, n, order, perm, inv);
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(st));
  *perm_dev = perm;
  *inv_dev = inv;
  *status = 1;
  return PSP_OK;
}

// Pattern of A + A^T without the diagonal as a CSR pair on the device (sind: n + 1 ints, scol: *snnz ints, both
// hipMalloc'ed, the caller frees); *ok = 0 when there is no room for the scratch (8 bytes x 4 per entry)
int reorder_symmetrize_device(int n, int nnz, const int *ind, const int *col, int **sind_out, int **scol_out,
                              long *snnz, int *ok) {
  *ok = 0;
  *sind_out = *scol_out = nullptr;
  hipStream_t st = stream();
  const size_t nk = 2 * (size_t)nnz;
  if (nk >= 0x7fffffffull) return PSP_OK;
  DevBuf b_keys, b_sorted, b_uniq, b_tmp, b_cnt;
  if (b_keys.alloc(sizeof(unsigned long long) * nk) != hipSuccess ||
      b_sorted.alloc(sizeof(unsigned long long) * nk) != hipSuccess ||
      b_uniq.alloc(sizeof(unsigned long long) * nk) != hipSuccess || b_cnt.alloc(sizeof(int)) != hipSuccess) {
    (void)hipGetLastError();
    return PSP_OK;
  }
  unsigned long long *keys = b_keys.as<unsigned long long>(), *sorted = b_sorted.as<unsigned long long>();
  unsigned long long *uniq = b_uniq.as<unsigned long long>();
  if (nnz > 0) {
    hipLaunchKernelGGL(sym_keys_kernel, dim3(std::min((n + 3) / 4, 1 << 22)), dim3(256), 0, st, n, ind, col, keys);
    PSP_LAUNCH_CHECK();
  }
  int bits = 1;
  while (bits < 32 && (1L << bits) < n) ++bits;
  size_t b1 = 0, b2 = 0;
  // (the all-ones keys of dropped entries need all 64 bits to sort last)
  PSP_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, b1, keys, sorted, (int)nk, 0, 64, st));
  PSP_HIP(hipcub::DeviceSelect::Unique(nullptr, b2, sorted, uniq, b_cnt.as<int>(), (int)nk, st));
  if (b_tmp.alloc(std::max(b1, b2)) != hipSuccess) {
    (void)hipGetLastError();
    return PSP_OK;
  }
  size_t tb = std::max(b1, b2);
  PSP_HIP(hipcub::DeviceRadixSort::SortKeys(b_tmp.p, tb, keys, sorted, (int)nk, 0, 64, st));
  tb = std::max(b1, b2);
  PSP_HIP(hipcub::DeviceSelect::Unique(b_tmp.p, tb, sorted, uniq, b_cnt.as<int>(), (int)nk, st));
  int m = 0;
  PSP_HIP(hipMemcpyAsync(&m, b_cnt.as<int>(), sizeof(int), hipMemcpyDeviceToHost, st));
  PSP_HIP(hipStreamSynchronize(st));
  if (m > 0) {  // the all-ones key, if any entry was dropped, is the last one
    unsigned long long last = 0;
    PSP_HIP(hipMemcpy(&last, uniq + (m - 1), sizeof(last), hipMemcpyDeviceToHost));
    if (last == ~0ull) --m;
  }
  int *sind = nullptr, *scol = nullptr;
  if (hipMalloc((void **)&sind, sizeof(int) * ((size_t)n + 1)) != hipSuccess ||
      hipMalloc((void **)&scol, sizeof(int) * (size_t)(m ? m : 1)) != hipSuccess) {
    if (sind) (void)hipFree(sind);
    (void)hipGetLastError();
    return PSP_OK;
  }
  hipLaunchKernelGGL(sym_rows_kernel, dim3((unsigned)(((long)m + 1 + 255) / 256)), dim3(256), 0, st, (long)m, n, uniq, sind,
                     scol);
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(st));
  *sind_out = sind;
  *scol_out = scol;
  *snnz = m;
  *ok = 1;
  return PSP_OK;
}

// R = P A P^T from device arrays into device arrays (rind: n + 1, rcol / rval: nnz entries)
int reorder_build_device(int n, const int *ind, const int *col, const double *val, const int *perm_dev,
                         const int *inv_dev, int *rind, int *rcol, double *rval) {
  hipStream_t st = stream();
  DevBuf b_len, b_tmp;
  PSP_HIP(b_len.alloc(sizeof(int) * ((size_t)n + 1)));
  hipLaunchKernelGGL(rcm_rowlen_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, st, n, ind, perm_dev, b_len.as<int>());
  size_t bytes = 0;
  PSP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, b_len.as<int>(), rind, n + 1, st));
  PSP_HIP(b_tmp.alloc(bytes));
  PSP_HIP(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, bytes, b_len.as<int>(), rind, n + 1, st));
  hipLaunchKernelGGL(rcm_copy_rows_kernel, dim3(std::min((n + 3) / 4, 1 << 22)), dim3(256), 0, st, n, ind, col, val, perm_dev, inv_dev,
                     rind, rcol, rval);
  PSP_LAUNCH_CHECK();
  PSP_HIP(hipStreamSynchronize(st));
  return PSP_OK;
}

// y[j] = yp[inv[j]] (+ partial sums of dotv . y, one per workgroup of 1024 rows: *nparts of them)
int reorder_back(int n, const int *inv_dev, const double *yp, double *y, const double *dotv, double *partials,
                 int *nparts, const int *skip) {
  if (n <= 0) return PSP_OK;
  const int grid = (n + 256 * kPermPerThread - 1) / (256 * kPermPerThread);
  hipLaunchKernelGGL(permute_back_kernel, dim3(grid), dim3(256), 0, stream(), n, inv_dev, yp, y, dotv, partials,
                     skip);
  PSP_LAUNCH_CHECK();
  if (nparts) *nparts = grid;
  return PSP_OK;
}

// dst[idx[i]] = src[i] on the library stream (idx a permutation of 0..n-1)
int reorder_scatter(int n, const int *idx_dev, const double *src, double *dst, const int *skip) {
  if (n <= 0) return PSP_OK;
  const int grid = (n + 256 * kPermPerThread - 1) / (256 * kPermPerThread);
  hipLaunchKernelGGL(permute_scatter_kernel, dim3(grid), dim3(256), 0, stream(), n, idx_dev, src, dst, skip);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// xp[i] = x[perm[i]] on the library stream
int reorder_gather(int n, const int *perm_dev, const double *x, double *xp, const int *skip) {
  if (n <= 0) return PSP_OK;
  const int grid = (n + 256 * kPermPerThread - 1) / (256 * kPermPerThread);
  hipLaunchKernelGGL(permute_gather_kernel, dim3(grid), dim3(256), 0, stream(), n, perm_dev, x, xp, skip);
  PSP_LAUNCH_CHECK();
  return PSP_OK;
}

// Host side of the renumbering: perm (new -> old) and the permuted CSR triple of a square matrix whose
// arrays are given on the host.  Per-row storage order is kept.
int reorder_rcm_host(int n, const int *ind, const int *col, const double *val, std::vector<int> &perm,
                     std::vector<int> &rind, std::vector<int> &rcol, std::vector<double> &rval) {
  const Graph g = symmetric_pattern(n, ind, col);
  perm = rcm_order(g, n);
  if ((int)perm.size() != n) return fail(PSP_EINVAL, "reorder: numbering is not a permutation");
  std::vector<int> inv((size_t)n);
  for (int i = 0; i < n; ++i) inv[perm[i]] = i;
  const size_t nnz = (size_t)ind[n];
  rind.resize((size_t)n + 1);
  rcol.resize(nnz);
  rval.resize(nnz);
  size_t p = 0;
  for (int i = 0; i < n; ++i) {
    rind[i] = (int)p;
    const int o = perm[i];
    for (int k = ind[o]; k < ind[o + 1]; ++k) {
      rcol[p] = inv[col[k]];
      rval[p++] = val[k];
    }
  }
  rind[n] = (int)p;
  return PSP_OK;
}

}  // namespace psp
