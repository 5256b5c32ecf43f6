// psp_csr_select.h -- a FRAGMENT of psp_csr.hip (included there, in this order, into one translation unit; not a header of its
// own): selection and launch: csr_spmv_launch (which kernel a handle gets), the fused / scaled / folded variants the solvers call, the
// halo-overlap split of the row-block drivers, the host-pointer staging and the pipelined host-pointer product.
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
