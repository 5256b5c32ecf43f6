// psp_reorder.hip -- bandwidth-reducing renumbering for irregular csr_mat / sss_mat operators.
//
// csr_spmv_w3 (x staged in LDS, 16-bit chunk-local columns, psp_csr.hip) needs every chunk of ~1000
// nonzeros to reference at most 64 blocks of 16 consecutive x entries.  An unstructured numbering
// (FEM meshes as they come out of a mesh generator, shuffled node ids) breaks that: the same matrix
// references 100-140 blocks per chunk and falls back to csr_spmv_w2, whose x gathers are bound by the
// per-CU L1 (0.61-0.67 of the HBM roofline, profiles/r1_fem_standin.txt).  This file computes a reverse
// Cuthill-McKee numbering on the host (once per handle) and builds the symmetrically permuted matrix
// R = P A P^T as a second device handle:
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
#include <numeric>
#include <vector>

#include "psp_internal.h"

using namespace psp;

namespace {

// ---- reverse Cuthill-McKee on the pattern of A + A^T (host)

struct Graph {
  std::vector<long> ptr;
  std::vector<int> adj;
  std::vector<int> deg;
};

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
  for (int i = 0; i < n; ++i) g.deg[i] = (int)(g.ptr[(size_t)i + 1] - g.ptr[i]);
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
      if (seen[v] != stamp) {
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
    if (placed[s]) continue;
    // pseudo-peripheral node (George & Liu): walk to a minimum-degree node of the last level while the
    // eccentricity grows
    int root = s;
    int nlev = bfs_levels(g, root, level, queue, last, ++stamp, seen);
    {  // lowest degree node of this component as the first guess
      int best = root;
      for (int u : queue)
        if (g.deg[u] < g.deg[best]) best = u;
      if (best != root) {
        root = best;
        nlev = bfs_levels(g, root, level, queue, last, ++stamp, seen);
      }
    }
    for (int iter = 0; iter < 8; ++iter) {
      int cand = last[0];
      for (int u : last)
        if (g.deg[u] < g.deg[cand]) cand = u;
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
        if (!placed[v]) {
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

}  // namespace

namespace psp {

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
