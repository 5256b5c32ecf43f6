// psp_csr_tables.h -- a FRAGMENT of psp_csr.hip (included there, in this order, into one translation unit; not a header of its
// own): per-handle side tables and what builds them lazily: ChunkTable / CsrExtra, ensure_rowoff / w3 / w6 / w5 / w4 / sss_w4 /
// schedule, the transposes, the renumbered copy (ensure_reordered) and its cost rule.

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
    const size_t nk = (size_t)nnz, ncur = ((size_t)ncols + 3) / 2 + 1;
    constexpr int kTranspMaxColumn = 4096;  // a longer column (a dense one) goes to the radix sort below: one thread sorts a column
    bool too_long = false;
    rc = psp::scratch_get(nk, &kbuf);
    if (rc == PSP_OK) rc = psp::scratch_get(ncur, &cbuf);
    if (rc == PSP_OK) {
      unsigned long long *key = reinterpret_cast<unsigned long long *>(kbuf);
      int *cursor = reinterpret_cast<int *>(cbuf);  // ncols + 1 ints: the counts, then the cursors; [ncols] = the flag
      size_t bytes = 0;
      hipError_t e = hipMemsetAsync(cursor, 0, sizeof(int) * ((size_t)ncols + 3), stream());
      const int g = (int)std::min<long>(((long)nnz + 255) / 256, 65536);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(transp_count_kernel, dim3(g), dim3(256), 0, stream(), nnz, ncols, col, cursor, cursor + ncols + 1);
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cursor, T->ind, ncols + 1, stream());
      }
      if (e == hipSuccess) e = hipMalloc(&tmp, bytes ? bytes : 1);
      if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, bytes, cursor, T->ind, ncols + 1, stream());
      int flags[2] = {0, 0};  // {a column index out of range, the longest column}
      if (e == hipSuccess) {
        hipLaunchKernelGGL(transp_maxlen_kernel, dim3(std::min((ncols + 255) / 256, 4096)), dim3(256), 0, stream(), ncols, T->ind,
                           cursor + ncols + 2);
        e = hipMemcpyAsync(flags, cursor + ncols + 1, sizeof(flags), hipMemcpyDeviceToHost, stream());
      }
      if (e == hipSuccess) e = hipStreamSynchronize(stream());
      const int bad = flags[0];
      too_long = flags[1] > kTranspMaxColumn;
      if (e == hipSuccess && bad) rc = fail(PSP_EINVAL, "transpose: a column index is out of range");
      if (e == hipSuccess && !bad && !too_long) e = hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)ncols, stream());
      if (e == hipSuccess && !bad && !too_long) {
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
    if (!too_long) {
      rc = finalize_csr(T);
      goto done;
    }
    if (tmp) (void)hipFree(tmp);  // (the scan's scratch; the radix sort below sizes its own)
    tmp = nullptr;
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

