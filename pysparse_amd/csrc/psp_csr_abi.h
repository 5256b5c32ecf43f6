// psp_csr_abi.h -- a FRAGMENT of psp_csr.hip (included there, in this order, into one translation unit; not a header of its
// own): the C ABI of include/pysparse_hip.h: psp_csr_*, psp_sss_*.
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
  if (A->w4_only) return fail(PSP_EINVAL, "psp_csr_download_rows: the operator has no CSR arrays (psp_csr_poisson_big, psp_csr_release_arrays)");
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
  if (A->w4_only) return fail(PSP_EINVAL, "psp_csr_download: the operator has no CSR arrays (psp_csr_poisson_big, psp_csr_release_arrays)");
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
  if (A->w4_only) return fail(PSP_EINVAL, "matvec_transp: the operator has no CSR arrays (psp_csr_poisson_big, psp_csr_release_arrays)");
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

int psp_csr_release_arrays(psp_csr_t *A) {
  PSP_API_GUARD_H(A);
  if (!A) return fail(PSP_EINVAL, "psp_csr_release_arrays: NULL handle");
  if (A->w4_only) return PSP_OK;
  if (A->host || A->multi || A->nparts || A->sym_owner)
    return fail(PSP_EINVAL, "psp_csr_release_arrays: not available for this kind of handle");
  PSP_TRY(ensure_device());
  psp::CsrExtra *ex;
  PSP_TRY(ensure_w4(A, &ex));
  if (ex->dia_state != 1 || ex->dia_no > 16)
    return fail(PSP_EINVAL, "psp_csr_release_arrays: the operator does not multiply with the index-free layout (csr_spmv_w4, "
                            "<= 16 offsets): its CSR arrays are what it streams");
  PSP_HIP(hipStreamSynchronize(stream()));  // nothing in flight reads the arrays
  int slot = -1;
  for (int o = 0; o < ex->dia_no; ++o)
    if (ex->dia_offs.o[o] == 0) slot = o;
  psp_csr *transposed = nullptr, *reordered = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);  // side tables that index into the arrays go with them
    transposed = ex->transposed;
    reordered = ex->reordered;
    ex->transposed = ex->reordered = nullptr;
    for (auto &t : ex->t) {
      if (t.second.tab) (void)hipFree(t.second.tab);
      if (t.second.rowoff) (void)hipFree(t.second.rowoff);
      if (t.second.blist) (void)hipFree(t.second.blist);
      if (t.second.blist6) (void)hipFree(t.second.blist6);
      if (t.second.col16) (void)hipFree(t.second.col16);
      if (t.second.ulist) (void)hipFree(t.second.ulist);
      if (t.second.colu) (void)hipFree(t.second.colu);
      if (t.second.perm) (void)hipFree(t.second.perm);
    }
    ex->t.clear();
    if (ex->packed) (void)hipFree(ex->packed);
    ex->packed = nullptr;
  }
  if (transposed) psp_csr_destroy(transposed);
  if (reordered) psp_csr_destroy(reordered);
  if (A->ind) (void)hipFree(A->ind);
  if (A->col) (void)hipFree(A->col);
  if (A->val) (void)hipFree(A->val);
  A->ind = A->col = nullptr;
  A->val = nullptr;
  A->padded = 0;
  A->nnz64 = A->nnz;
  A->w4_diag_slot = slot;
  A->max_row_nnz = ex->dia_no;  // what psp_csr_device_bytes prices an index-free handle with
  A->variant = -1;
  A->w4_only = true;
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

