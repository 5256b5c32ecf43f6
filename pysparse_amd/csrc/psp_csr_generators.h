// psp_csr_generators.h -- a FRAGMENT of psp_csr.hip (included there, in this order, into one translation unit; not a header of its
// own): the Poisson generators (pysparse/tools/poisson.py:22-37 extended to 3-D, on the device) and the random banded rows.
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

