// psp_csr_hostutil.h -- a FRAGMENT of psp_csr.hip (included there, in this order, into one translation unit; not a header of its
// own): host helpers: variant decoding, launch wrappers of the kernels.
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

