/*
 * pysparse_hip.h -- C ABI of libpysparse_hip.so, the MI355X (gfx950) implementation of
 * PySparse's SpMV + Krylov hot path.
 *
 * Plain C: ints, doubles, raw pointers and opaque handles only -- no Python, NumPy or
 * torch types cross this boundary.  The CPython extension modules in pysparse_amd/
 * (spmatrix, krylov, precon), bench.py and the GPU tests all go through these entry
 * points.  Citations are file:line under the reference tree (PythonOptimizers/pysparse).
 *
 * Conventions
 *   - every function returns PSP_OK (0) or a negative PSP_E* status; psp_last_error()
 *     returns the message of the calling thread's last failure.  Nothing throws.
 *   - "host" pointers are ordinary process memory, borrowed for the call;
 *     "dev" pointers are HIP device memory on the library's current device.
 *   - handles own their device memory; destroy releases it.
 *   - all work is enqueued on ONE stream (psp_set_stream; default = the null stream);
 *     host-pointer entry points synchronise before returning, *_dev entry points that
 *     return no scalar do not.
 *   - handles are not thread-safe; there is NO CPU fallback: without a usable GPU every
 *     compute entry point fails with PSP_ENODEV.
 *   - indices are 32-bit (the reference's C int: csr_mat.h:6-13), values are double.
 */
#ifndef PYSPARSE_HIP_H
#define PYSPARSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSP_OK 0
#define PSP_EINVAL (-1)   /* bad argument (shape, NULL, range)                     */
#define PSP_ENODEV (-2)   /* no HIP device / runtime error                         */
#define PSP_ENOMEM (-3)   /* device or host allocation failed                      */
#define PSP_ESINGULAR (-4) /* jacobi: diagonal element close to zero               */
#define PSP_ECALLBACK (-5) /* a host operator callback reported failure            */

/* solver info codes: pysparse/itsolvers/src/itsolversmodule.c:625-646 */
#define PSP_INFO_CONVERGED 0
#define PSP_INFO_MAXIT (-1)
#define PSP_INFO_ILLCOND_PRECON (-2)
#define PSP_INFO_NOT_SPD_PRECON (-3)
#define PSP_INFO_STAGNATION (-5)
#define PSP_INFO_BREAKDOWN (-6)

typedef struct psp_csr psp_csr_t;       /* device mirror of CSRMatObject, csr_mat.h:6-13     */
typedef struct psp_sss psp_sss_t;       /* device mirror of SSSMatObject, sss_mat.h:6-14     */
typedef struct psp_jacobi psp_jacobi_t; /* device mirror of JacobiObject, preconmodule.c:11-19 */
typedef struct psp_ssor psp_ssor_t;     /* device mirror of SSORObject, preconmodule.c:21-31   */
typedef struct psp_op psp_op_t;         /* "anything with shape + matvec / precon":
                                           the operator protocol of spmatrixmodule.c:169-248 */

/* ------------------------------------------------------------------ runtime */

const char *psp_last_error(void);
const char *psp_version(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int psp_device_count(void);
/* Threading: device, stream, reduction workspace and host staging belong to the CALLING THREAD.  The first thread
 * that uses the library enqueues on the null stream (or on what it hands psp_set_stream); every other thread gets a
 * non-blocking stream of its own on first use, so two threads with two handles overlap on the GPU.  A handle may be
 * shared between threads: every entry point locks the handles it is given (operator, preconditioner and the matrix
 * behind it), so calls on the SAME handle take turns.  (The reference holds the GIL across its whole solve.) */
/* select the device the calling thread's later calls use (hipSetDevice); threads that first call the library later
 * start on the device selected last */
int psp_set_device(int device);
/* which context the calling thread has: its slot (0 = the first thread), its device and the stream it enqueues on */
int psp_thread_info(int *thread_slot, int *device, void **hip_stream);
/* test hook of the locking discipline (tests/test_threading_cpu.py): takes the locks of the two handles exactly as an
 * entry point that is handed them does (address order, recursive), holds them for `milliseconds`, releases them */
int psp_debug_hold_handles(const void *h1, const void *h2, int milliseconds);
/* test hooks of the multi-stream ordering (tests/test_gpu_shake.py; only in a process started with PSP_TUNING=1, else
 * PSP_EINVAL).  psp_debug_shake arms delay injection: at every cut point of the multi-device driver whose bit is set in
 * point_mask (psp_internal.h ShakePoint), for every rank whose bit is set in rank_mask, a one-wave spin kernel of
 * min_us..max_us microseconds (pseudo-random from `seed`; a fixed delay when min_us == max_us) is enqueued on the stream
 * that cut point names, so that stream's later work moves against every other stream.  revert_mask takes a known fix
 * out again (bit 0: the copy stream's wait for the receiver's own stream, round 4) -- the test that the facility
 * finds what it is there to find.  seed < 0 disarms.  psp_debug_shake_count: spin kernels injected since it was armed.
 * psp_debug_spin: one spin kernel on the calling thread's stream (the torch driver's cut points, distributed.py). */
/* how many solves the single-kernel loop for mid-size offset-structured systems (psp_mid.hip) has run in this process,
 * and how many times it handed a solve back to the launch-per-phase loops (tests) */
int psp_debug_mid_count(long long *solves, long long *fallbacks);
/* the same counters for the brick form of those loops (7-offset operators of 3-D grids; psp_mid.hip) */
int psp_debug_brick_count(long long *solves, long long *fallbacks);
int psp_debug_shake(long long seed, int min_us, int max_us, unsigned point_mask, unsigned rank_mask, int revert_mask);
int psp_debug_shake_count(long long *injected);
int psp_debug_spin(int microseconds);
/* *can_access = 1 when `device` can read / write `peer`'s memory directly (xGMI or PCIe peer path; what the
 * halo copies of a multi-device matrix need), 1 for device == peer.  Creates no context: bench.py's pre-flight
 * matrix of an N-GPU run */
int psp_peer_access(int device, int peer, int *can_access);
/* Placement of library-owned vectors (round 5; pysparse_amd/csrc/psp_place.hip).  What a
 * bandwidth-bound product takes depends on which pages of HBM its vector operands occupy (two levels ~8 % apart at
 * 512^3); for the vectors the LIBRARY allocates -- the solvers' work vectors (the reference's solvers own their work
 * array too: pysparse/itsolvers/src/itsolversmodule.c:32-118) and the staging pair of the host-pointer products -- it
 * can draw a few candidates once per (device, length), time the handle's own product on each and keep the best.  Results
 * do not change by a bit.  psp_set_placement(1) turns the draws on; the default is OFF: measured over 11 processes the
 * product on a drawn pair is 0 ... 4.6 % faster (mean 1.8 %), Jacobi-PCG +0.5 ... +1.7 %, for 60-80 ms per draw
 * (profiles/r5_placement_ab_*.jsonl).  psp_placement_info: draws made so far and the milliseconds they took.  psp_place_operands hands the same service to a caller that owns its vectors:
 * *y_dev / *x_dev receive zeroed device vectors (nrows / ncols doubles, release with psp_free) chosen for the output /
 * input role of y = A x; report6 (may be NULL): candidates drawn (0: plain allocations -- vectors under 64 MiB, or memory
 * short), best and worst output-role ms, best and worst input-role ms, ms the draw took. */
int psp_set_placement(int on);
int psp_placement_info(int *enabled, long long *draws, double *draw_ms_total);
int psp_place_operands(const psp_csr_t *A, double **y_dev, double **x_dev, double *report6);
/* the calling thread enqueues on an externally owned hipStream_t (e.g. torch's current stream); NULL = null stream */
int psp_set_stream(void *hip_stream);
int psp_synchronize(void);
/* name, CU count and HBM bytes of the current device */
int psp_device_info(char *name, int name_len, int *compute_units, int64_t *hbm_bytes);
/* free / total device memory right now (leak checks; psp_trim() first to drop the solver scratch pool) */
int psp_mem_info(int64_t *free_bytes, int64_t *total_bytes);

/* device memory + timing hooks used by bench.py and the tests */
int psp_malloc(void **dev, size_t bytes);
int psp_free(void *dev);
int psp_memcpy_h2d(void *dev, const void *host, size_t bytes);
int psp_memcpy_d2h(void *host, const void *dev, size_t bytes);
int psp_memset(void *dev, int byte, size_t bytes);
/* release the cached solver work vectors (the library keeps the last few GB-sized scratch
 * vectors between solves because hipMalloc/hipFree of them costs milliseconds) */
int psp_trim(void);
int psp_event_create(void **event);
int psp_event_destroy(void *event);
int psp_event_record(void *event);                       /* on the library stream */
int psp_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */
/* measurement hook ("timing hooks (hipEvent ms, bytes moved)"): what this GPU's memory
 * system gives a plain streaming kernel of a given shape right now, in THIS process -- `reads` (0..8) read
 * streams and `writes` (0..1) write streams of bytes_per_stream bytes each (a multiple of 4096), one 16-byte
 * element per thread and stream, one workgroup per 4 KiB span, `reps` timed launches after two untimed ones.
 * reads = 7, writes = 1 is the access shape of csr_spmv_w4 on the 7-point operator without its x re-reads:
 * bench.py prints it beside the SpMV as the ceiling of that shape.  The buffers are the call's own. */
int psp_stream_probe(int reads, int writes, size_t bytes_per_stream, int reps, float *avg_ms, float *min_ms);
/* Which loop the calling thread's LAST psp_pcg / psp_minres ran (the reference has one loop each: pcg.c:91-163,
 * minres.c:96-193; here the size and the operator pick one of several with the same semantics), so that a caller -- bench.py
 * -- derives its byte model from what ran instead of mirroring the selection rules.  name: "pcg_lazy_pf" (p and x updates
 * folded into the product: 4 launches), "pcg_lazy" (5), "pcg_eager" (6), "pcg_mid" / "pcg_brick" / "pcg_coop" (the whole
 * loop in one cooperative kernel), "pcg_host_scalars" (generic operands, Python callbacks), "minres_async", "minres_mid",
 * "minres_brick", "minres_coop", "minres_host_scalars", "pcg_multi" / "minres_multi" (device list); "" before the first
 * solve.  info[4] = {launches per iteration, vector bytes per row and iteration beside the product (0 inside the
 * single-kernel loops, -1 where not modelled), 1 when the Jacobi diagonal is streamed (not a constant), single-kernel
 * loops refused or abandoned after launch in this process so far (each fell back to the launch-per-phase loop)}. */
int psp_last_solve_info(char *name, int name_cap, int *info);
/* on = 0: psp_pcg / psp_minres never take the single-kernel loops (psp_mid.hip, psp_coop.hip) in this process.  Why one
 * would: the brick and small-system loops add their dot products in an order of their own that depends on the device's CU
 * count, so their iterates agree with the launch-per-phase loops' to rounding only, and a refused cooperative launch
 * (a shared GPU) falls back to those loops' bits -- with 0 every solve has the launch-per-phase bits on every device.
 * Default 1. */
int psp_set_single_kernel_loops(int on);
/* hash of the sources this binary was built from (__graft_entry__.source_hash(); "unstamped" for a hand build):
 * tests compare it with the hash of the sources on disk, so a stale prebuilt library cannot pass for a fresh one */
const char *psp_build_id(void);

/* ------------------------------------------------------------------ csr_mat */

/* replaces newCSRMatObject + the fill loop of LLMat_to_csr
 * (pysparse/sparse/src/csr_mat.c:259-296, ll_mat.c:1577-1648): uploads a host CSR
 * triple.  ind has nrows+1 entries, ind[0] == 0, ind[nrows] == nnz. */
int psp_csr_create(int nrows, int ncols, int nnz, const int *ind_host, const int *col_host,
                   const double *val_host, psp_csr_t **out);
/* scalable constructor: 5-/7-point Poisson operator generated ON the
 * device in the ordering of pysparse/tools/poisson.py:22-37, k = i + nx*j (+ nx*ny*l),
 * diag 4 (nz == 0, 2-D) or 6 (3-D), off-diagonals -1, columns ascending.
 * The slab form keeps only global rows [row_lo, row_hi) and shifts column indices by
 * -col_shift (row-range partition for multi-GPU; ncols_local = width of the local
 * extended x vector).  psp_csr_poisson == slab over all rows with shift 0. */
int psp_csr_poisson(int nx, int ny, int nz, psp_csr_t **out);
int psp_csr_poisson_slab(int nx, int ny, int nz, int64_t row_lo, int64_t row_hi, int64_t col_shift,
                         int ncols_local, psp_csr_t **out);
/* The same operator for grids whose nonzero count exceeds 32 bits (1024^3 on ONE GPU: 7.5e9
 * entries, the strong-scaling baseline of BASELINE.json configs[3]): generated directly in the
 * index-free offset-major layout of csr_spmv_w4, ~62 GB at 1024^3.  Supports
 * matvec, jacobi, the solvers and kernel_info; there are no CSR arrays to download or transpose.
 * psp_csr_shape reports nnz = -1 when it does not fit an int: use psp_csr_nnz64. */
int psp_csr_poisson_big(int nx, int ny, int nz, psp_csr_t **out);
/* Row slab [row_lo, row_hi) of the same index-free operator, columns shifted by -col_shift into a
 * local extended vector of ncols_local entries (same meaning as psp_csr_poisson_slab): the
 * strong-scaling legs of configs[3] at 2 and 4 GPUs hold 2^29 / 2^28 rows = 3.8e9 / 1.9e9 nonzeros
 * per rank, beyond 32-bit CSR offsets.  psp_csr_poisson_big == slab over all rows with shift 0. */
int psp_csr_poisson_big_slab(int nx, int ny, int nz, int64_t row_lo, int64_t row_hi, int64_t col_shift,
                             int ncols_local, psp_csr_t **out);
int64_t psp_csr_nnz64(const psp_csr_t *A);
/* Multi-GPU variants taking a device list (reference loops pcg.c:91-163,
 * minres.c:96-193, csr_mat.c:49-54): the rows of the operator -- and the matching slices of every solver
 * vector -- live as contiguous row blocks on devices[0..ndev), one rank per entry, ONE process (one host
 * thread enqueues for all ranks; a device may be listed more than once: ranks sharing a GPU).  The result is
 * an ordinary psp_csr_t: psp_csr_matvec(_stride), psp_csr_diagonal, psp_csr_shape, psp_csr_kernel_info,
 * psp_op_from_csr, psp_jacobi_create_csr (steps = 1), psp_pcg and psp_minres accept it -- K = NULL or the
 * jacobi of the same matrix -- and every other entry point answers PSP_EINVAL.  Ghost entries travel by peer
 * copies on a second stream per rank, overlapped with the rows that need none; the two packed reductions of
 * an iteration go through RCCL (ncclAllReduce in stream order; librccl is loaded on first use) when every
 * rank has its own device, else through a fixed-order fold kernel over peer pointers.
 *   poisson: z-slabs (y-slabs in 2-D) of whole grid planes, index-free slab operators (psp_csr_poisson_big_slab)
 *   create : any square CSR matrix from host arrays, n / ndev rows per rank, ghost lists found here */
int psp_csr_poisson_multi(int nx, int ny, int nz, const int *devices, int ndev, psp_csr_t **out);
int psp_csr_create_multi(int nrows, int ncols, int nnz, const int *ind_host, const int *col_host,
                         const double *val_host, const int *devices, int ndev, psp_csr_t **out);
/* ranks / distinct devices behind a handle (0 / 0 for a single-device matrix); uses_rccl: the reductions go
 * through RCCL rather than the fold kernel */
int psp_csr_multi_info(const psp_csr_t *A, int *nranks, int *distinct_devices, int *uses_rccl);
/* The partition psp_csr_create_multi computes for rank `rank` of `ndev`, as pure host code (no device needed: the CPU tests
 * compare it with general_halo_plan of pysparse_amd/distributed.py): row_range_out = {row_lo, row_hi}; counts_out =
 * {ghost_lo, ghost_hi, interior_a, interior_b, nlinks}; ghost_ids (may be NULL) = the sorted global ids of the ghost entries;
 * links (may be NULL) = nlinks x {sending rank, offset in the extended vector, count, send offset in the sender's owned
 * entries or -1 for a gathered index list}; col_local (may be NULL, ind[row_hi] - ind[row_lo] ints) = the block's columns in
 * [ghost_lo | owned | ghost_hi] numbering. */
int psp_multi_plan(int nrows, int ncols, const int *ind_host, const int *col_host, int ndev, int rank,
                   int64_t *row_range_out, int *counts_out, int *ghost_ids, int ghost_cap, int *links, int links_cap,
                   int *col_local);
/* timing hook: `reps` products y = A x on resident slices, each done the way a solver iteration does it (ghost
 * copies on the copy streams, the rows that need none meanwhile, then the boundary rows); *ms_per_product is the
 * slowest rank's stream time per product (HIP events on every rank's compute stream) */
int psp_csr_multi_spmv_time(psp_csr_t *A, int warmup, int reps, double *ms_per_product);
/* the pieces of that product and of an iteration on their own, same timing: what 0 = the ghost exchange alone,
 * 1 = the local product alone (no exchange), 2 = one packed reduction of two doubles (RCCL all-reduce or the fold
 * kernel).  bench.py reports them as `phases` of the single-process line; the halo time hidden behind the interior
 * rows is (t0 + t1 - t_product) / t0. */
int psp_csr_multi_phase_time(psp_csr_t *A, int what, int warmup, int reps, double *ms_per_rep);
/* General CSR beyond the reference's C int (csr_mat.h:6-13: `int nnz`, `int *ind`): row offsets are 64-bit at
 * this boundary, column indices stay 32-bit.  Above 2^30 nonzeros the rows are cut into parts of < 2^30
 * nonzeros that share x and write disjoint row ranges of y ("64-bit row offsets ... or
 * partitions resident on one device"): the SpMV kernels never see a 64-bit offset.  matvec, the diagonal,
 * jacobi and the solvers work on such a handle; download goes by row ranges; there is no transposed product. */
int psp_csr_create64(int nrows, int ncols, int64_t nnz, const int64_t *ind_host, const int *col_host,
                     const double *val_host, psp_csr_t **out);
/* Synthetic general (non-stencil) CSR generated on the device, for tests and measurements at sizes no host
 * array reaches: row r stores m entries, entry j in column (r + (j - m/2)*stride + h(r, j) mod stride) mod ncols
 * with a value in [-1, 1), h = splitmix64 of (seed, r, j) (psp_csr.hip: random_banded_kernel; the tests restate
 * the formula).  nrows*m may exceed 2^31. */
int psp_csr_random_banded(int nrows, int ncols, int m, int stride, uint64_t seed, psp_csr_t **out);
/* rows [row_lo, row_hi) of any handle with CSR arrays: ind_host gets row_hi - row_lo + 1 offsets relative to
 * row_lo; col_host / val_host (may be NULL) the entries */
int psp_csr_download_rows(const psp_csr_t *A, int row_lo, int row_hi, int64_t *ind_host, int *col_host,
                          double *val_host);
int psp_csr_destroy(psp_csr_t *A);
/* shape / nnz attributes: CSRMatType_getattr, csr_mat.c:208-231 */
int psp_csr_shape(const psp_csr_t *A, int *nrows, int *ncols, int *nnz);
/* copy the device arrays back (structure parity tests; any pointer may be NULL) */
int psp_csr_download(const psp_csr_t *A, int *ind_host, int *col_host, double *val_host);
/* A[i,i] for all rows (0.0 where no diagonal entry is stored) */
int psp_csr_diagonal(const psp_csr_t *A, double *diag_host);

/* y := A x.  CSRMat_matvec, csr_mat.c:141-163 (kernel :49-54); x has ncols, y nrows
 * entries; y is overwritten.  Per-row summation is left to right from 0.0 with separate
 * multiply and add, so results are bit-identical to the reference loop. */
int psp_csr_matvec(psp_csr_t *A, const double *x_host, double *y_host);
/* element strides of non-contiguous NumPy views: csr_matvec_kernel_stride, csr_mat.c:58-72 */
int psp_csr_matvec_stride(psp_csr_t *A, const double *x_host, ptrdiff_t incx, double *y_host,
                          ptrdiff_t incy);
/* y := A^T x.  CSRMat_matvec_transp, csr_mat.c:114-133 (kernel :74-88).  Every y[c] adds its terms by
 * ascending row like the reference's scatter loop, so the result is bit-identical: offset-structured
 * matrices gather over the w4 layout, all others multiply with A^T kept as a second CSR matrix (built
 * on the first call, + one matrix of device memory). */
int psp_csr_matvec_transp(psp_csr_t *A, const double *x_host, double *y_host);
int psp_csr_matvec_transp_stride(psp_csr_t *A, const double *x_host, ptrdiff_t incx,
                                 double *y_host, ptrdiff_t incy);
/* device-pointer flavours (no synchronisation) */
int psp_csr_matvec_dev(psp_csr_t *A, const double *x_dev, double *y_dev);
int psp_csr_matvec_transp_dev(psp_csr_t *A, const double *x_dev, double *y_dev);
/* kernel tuning knob for A/B measurements: variant < 0 restores the default */
int psp_csr_set_variant(psp_csr_t *A, int variant);
/* workgroup schedule of the SpMV (tuning / tests; results never depend on it): strip_rows < 0
 * automatic (plane-sweeping order for wide-band operators such as the 7-point stencil), 0 natural
 * row order, > 0 plane-sweeping with that many rows per strip regardless of the matrix size */
int psp_csr_set_schedule(psp_csr_t *A, int strip_rows);
/* which kernel y := A x runs for this handle (builds its tables if needed): name, and
 * info[4] = {w3: x blocks per chunk list / w4: number of distinct col-row offsets,
 * most x blocks one chunk references (w3), schedule active, half band width used by the schedule}.
 * csr_spmv_w4: offset-structured operators (<= 32 distinct col - row, ascending columns), values in
 *   offset-major blocks + 16-bit row masks; csr_spmv_w3: banded CSR, x staged in LDS, 16-bit
 *   chunk-local columns; csr_spmv_w2 / w1 / stream: any CSR.  All add each row's products left
 *   to right (csr_mat.c:49-54), so the choice never changes a bit of y. */
int psp_csr_kernel_info(psp_csr_t *A, char *name, int name_cap, int *info);
/* Set-up against steady state for irregular numberings (round 6; no reference analogue: csr_mat.c:259-296 builds nothing).
 * A csr_mat whose stored numbering scatters its columns (an FEM mesh as it comes out of a generator) multiplies 4-12 us
 * faster per product, and iterates 16-24 us faster per Jacobi-MINRES iteration, through a renumbered copy (reverse
 * Cuthill-McKee, "csr_spmv_w3_rcm") than on the stored numbering ("csr_spmv_w5") -- and the copy costs 17-57 ms to build at
 * n = 9.3e5.  The library therefore builds it only once a handle has done 2048 products (solver iterations included), or at
 * the next product after the caller has announced at least that many here.  y = A x has the same bits either way; a fused solve's iterates differ at rounding level between the two
 * numberings (its dot products add in the numbering it runs in), deterministically for a given sequence of calls.
 * psp_csr_setup_info: info4 = {ms the copy took to build (0: not built), products counted so far, the threshold,
 * state (-1 undecided, 0 examined and not built, 1 built)}. */
int psp_csr_prepare(psp_csr_t *A, long long expected_products);
/* An offset-structured csr_mat (csr_spmv_w4, <= 16 distinct col - row) multiplies with its index-free tables; the CSR arrays
 * it was created from stay for download, the variants and ABI parity with CSRMatObject (csr_mat.h:6-13) -- 1.65 x the
 * memory (512^3: 11.8 GB of arrays beside 7.8 GB of tables).  A caller that needs neither frees them here: the handle then
 * supports matvec (incl. the transposed product), diagonal / jacobi, the solvers and kernel_info, like the operator of
 * psp_csr_poisson_big; psp_csr_download fails with PSP_EINVAL and psp_csr_set_variant has nothing left to select.  Results do not change by a bit.
 * PSP_EINVAL for handles that stream their CSR arrays (w3 / w6 / w2 / w5). */
int psp_csr_release_arrays(psp_csr_t *A);
int psp_csr_setup_info(psp_csr_t *A, double *info4);
/* The renumbering behind "csr_spmv_w3_rcm" (irregular square operators): perm_host[new] = old row
 * (nrows ints) and *available = 2 (numbering computed on the device) or 1 (on the host: fallback, A/B switch) when
 * psp_csr_kernel_info / a product has built one for this handle; *available = 0 otherwise (perm_host untouched).
 * Diagnostic: no product needs it. */
int psp_csr_renumbering(psp_csr_t *A, int *perm_host, int *available);
/* bytes of device memory held by the handle */
int64_t psp_csr_device_bytes(const psp_csr_t *A);

/* ------------------------------------------------------------------ sss_mat */

/* replaces newSSSMatObject + LLMat_to_sss (sss_mat.c:243-284, ll_mat.c:1654-1708):
 * strict lower triangle in CSR form (nnz_lower entries) + dense diagonal. */
int psp_sss_create(int n, int nnz_lower, const int *ind_host, const int *col_host,
                   const double *val_host, const double *diag_host, psp_sss_t **out);
/* symmetric-skyline form of the Poisson operator (poisson2d_sym(n).to_sss()) */
int psp_sss_poisson(int nx, int ny, int nz, psp_sss_t **out);
int psp_sss_destroy(psp_sss_t *A);
/* n, and nnz as the reference reports it: strict-lower count + n (sss_mat.c:155) */
/* which kernel y := S x runs (sss_spmv_w4 for offset-structured matrices: the strict lower triangle
 * only, half the traffic of the mirrored product; otherwise the csr kernels on the full mirror),
 * and the A/B knob -- same meaning as psp_csr_kernel_info / psp_csr_set_variant */
int psp_sss_kernel_info(psp_sss_t *A, char *name, int name_cap, int *info);
/* psp_csr_prepare / psp_csr_setup_info for the mirror an irregular sss_mat multiplies with */
int psp_sss_prepare(psp_sss_t *A, long long expected_products);
int psp_sss_setup_info(psp_sss_t *A, double *info4);
int psp_sss_set_variant(psp_sss_t *A, int variant);
int psp_sss_shape(const psp_sss_t *A, int *n, int *nnz_reported);
int psp_sss_download(const psp_sss_t *A, int *ind_host, int *col_host, double *val_host,
                     double *diag_host);
/* A[i,j]: getitem, sss_mat.c:14-28 */
int psp_sss_getitem(const psp_sss_t *A, int i, int j, double *value);
/* y := A x (== A^T x).  SSSMat_matvec, sss_mat.c:78-96 (kernel :40-56).  Per row the
 * terms are added in the reference's order: lower entries by ascending column, the
 * diagonal term, then the mirrored upper entries by ascending row. */
int psp_sss_matvec(psp_sss_t *A, const double *x_host, double *y_host);
int psp_sss_matvec_stride(psp_sss_t *A, const double *x_host, ptrdiff_t incx, double *y_host,
                          ptrdiff_t incy);
int psp_sss_matvec_dev(psp_sss_t *A, const double *x_dev, double *y_dev);
int64_t psp_sss_device_bytes(const psp_sss_t *A);

/* ------------------------------------------------------------------- jacobi */

/* replaces newJacobiObject (pysparse/precon/src/preconmodule.c:352-412):
 * dinv[i] = omega / A[i,i]; PSP_ESINGULAR if 1.0 + A[i,i] == 1.0 for some i (:395).
 * The native forms read the diagonal on the device instead of n calls of A[i,i].
 * steps > 1 keeps a reference to the operator for the extra sweeps (:45-52). */
int psp_jacobi_create_csr(psp_csr_t *A, double omega, int steps, psp_jacobi_t **out);
int psp_jacobi_create_sss(psp_sss_t *A, double omega, int steps, psp_jacobi_t **out);
/* generic form: host diagonal (already fetched through A[i,i]) + the operator used by
 * the steps > 1 sweeps (may be NULL when steps == 1) */
int psp_jacobi_create_diag(int n, const double *diag_host, double omega, int steps,
                           const psp_op_t *A_or_null, psp_jacobi_t **out);
int psp_jacobi_destroy(psp_jacobi_t *K);
int psp_jacobi_shape(const psp_jacobi_t *K, int *n);
/* y := K x.  Jacobi_precon, preconmodule.c:60-80 (kernel :35-54) */
int psp_jacobi_precon(psp_jacobi_t *K, const double *x_host, double *y_host);
int psp_jacobi_precon_dev(psp_jacobi_t *K, const double *x_dev, double *y_dev);

/* --------------------------------------------------------------------- ssor */

/* replaces newSSORObject (preconmodule.c:414-459): `steps` symmetric Gauss-Seidel (omega == 1,
 * symgs_kernel :149-193) or SSOR (ssor_kernel :95-143) steps with a zero initial guess on an
 * sss_mat.  The triangular sweeps are level-scheduled; every row performs the reference's
 * operations in the reference's order, so results are bit-identical to the sequential loops.
 * The handle borrows A (keep it alive). */
int psp_ssor_create(psp_sss_t *A, double omega, int steps, psp_ssor_t **out);
int psp_ssor_destroy(psp_ssor_t *K);
/* n and the number of dependency levels of the forward / backward sweep */
int psp_ssor_info(const psp_ssor_t *K, int *n, int *levels_forward, int *levels_backward);
/* how much of the schedule runs as single-workgroup runs of narrow levels that exchange x through LDS (2-D operators,
 * the thin ends of 3-D ones) instead of one launch per level: runs per direction, levels and slots covered; no
 * reference analogue (the reference's sweeps are sequential, preconmodule.c:95-193) */
int psp_ssor_run_info(const psp_ssor_t *K, int *runs_forward, int *runs_backward, long *levels_in_runs,
                      long *slots_in_runs);
/* 3-D grid operators whose levels are too wide for such runs are swept brick by brick (bricks of `edge`^3 grid points, a
 * coarse wavefront of workgroups): the number of bricks, 0 when the handle does not use them */
int psp_ssor_brick_info(const psp_ssor_t *K, int *bricks, int *edge);
/* y := K x.  SSOR_precon, preconmodule.c:199-223 */
int psp_ssor_precon(psp_ssor_t *K, const double *x_host, double *y_host);
int psp_ssor_precon_dev(psp_ssor_t *K, const double *x_dev, double *y_dev);

/* --------------------------------------------------------- operator protocol */

/* Host callback operator: the C image of SpMatrix_Matvec / SpMatrix_Precon
 * (spmatrixmodule.c:169-201, :215-248).  Called with HOST vectors; returns 0, or
 * non-zero when the callee failed (a Python exception is pending). */
typedef int (*psp_host_apply_fn)(void *ctx, int n, const double *x_host, double *y_host);

int psp_op_from_csr(psp_csr_t *A, psp_op_t **out);
int psp_op_from_sss(psp_sss_t *A, psp_op_t **out);
int psp_op_from_jacobi(psp_jacobi_t *K, psp_op_t **out);
int psp_op_from_ssor(psp_ssor_t *K, psp_op_t **out);
int psp_op_from_callback(int n, psp_host_apply_fn fn, void *ctx, psp_op_t **out);
int psp_op_destroy(psp_op_t *op);

/* ------------------------------------------------------------------ solvers */

/* info, iter, relres = pcg(A, b, x, tol, maxit[, K]) -- ItSolvers_pcg
 * (itsolversmodule.c:32-118) + Itsolvers_pcg_kernel (pysparse/itsolvers/src/pcg.c:22-171).
 * The whole loop runs on the device; b and x cross PCIe once each way.  Semantics kept:
 * tolb = tol*||b||; b == 0 -> x := 0, info 0; iter == maxit+1 when not converged;
 * stagnation / breakdown codes as above.  K may be NULL.  hist_host (may be NULL, maxit+1
 * doubles) receives ||r|| per iteration.  Returns PSP_ECALLBACK when a callback failed
 * (the reference returns -1 from the kernel, pcg.c:8-11). */
int psp_pcg(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
            double tol, int maxit, int *info, int *iter, double *relres, double *hist_host);
int psp_pcg_dev(const psp_op_t *A, const psp_op_t *K, int n, double *x_dev, const double *b_dev,
                double tol, int maxit, int *info, int *iter, double *relres, double *hist_host);

/* info, iter, relres = minres(A, b, x, tol, maxit[, K]) -- ItSolvers_minres
 * (itsolversmodule.c:217-305) + Itsolvers_minres_kernel (pysparse/itsolvers/src/minres.c:43-200).
 * *relres is written only on the 0 / -1 exits, as in the reference. */
int psp_minres(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
               double tol, int maxit, int *info, int *iter, double *relres, double *hist_host);
int psp_minres_dev(const psp_op_t *A, const psp_op_t *K, int n, double *x_dev,
                   const double *b_dev, double tol, int maxit, int *info, int *iter,
                   double *relres, double *hist_host);

/* The four other Krylov kernels of the reference's `krylov` module on the same operator
 * protocol, unfused: cgs (pysparse/itsolvers/src/cgs.c:14-110,
 * wrapper itsolversmodule.c:503-586), bicgstab (bicgstab.c:233-320, :125-216), qmrs
 * (qmrs.c:29-154, :410-496; the initial guess is ignored), gmres(dim)
 * (gmres.c:62-175, :313-403; *relres is the true residual reduction). */
int psp_cgs(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
            double tol, int maxit, int *info, int *iter, double *relres);
int psp_bicgstab(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
                 double tol, int maxit, int *info, int *iter, double *relres);
int psp_qmrs(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
             double tol, int maxit, int *info, int *iter, double *relres);
int psp_gmres(const psp_op_t *A, const psp_op_t *K, int n, double *x_host, const double *b_host,
              double tol, int maxit, int dim, int *info, int *iter, double *relres);

/* ------------------------------------------ solver phase kernels (device pointers)
 * The building blocks of the loops above, exported so that the row-partitioned
 * multi-GPU driver (pysparse_amd/distributed.py) can interleave them with RCCL
 * collectives issued through torch.distributed.  Each reduction leaves its result(s) in
 * out_dev[0..k) on the device (no host synchronisation). */

/* Tell the vector kernels that v_dev[0..n) holds ONE value everywhere (checked on the device; a
 * vector that does not is simply not registered): psp_k_residual / psp_k_pupdate / psp_k_xr_update
 * then form z = dinv.*r as r_i * c without streaming dinv -- same product, 8 bytes per row less.
 * The vector must not change until psp_k_unhint.  precon.jacobi registers its own dinv. */
int psp_k_hint_constant(const double *v_dev, int n);
int psp_k_unhint(const double *v_dev);
/* out[0] = sum x_i*y_i */
int psp_k_dot(int n, const double *x_dev, const double *y_dev, double *out_dev);
/* r := b - r (pcg.c:73-74); out = { r.r, r.z } with z = dinv.*r (dinv may be NULL: z = r) */
int psp_k_residual(int n, const double *b_dev, double *r_dev, const double *dinv_dev,
                   double *out_dev);
/* p := z + beta*p (pcg.c:113-114), z = dinv.*r or r; first != 0: p := z (pcg.c:106) */
int psp_k_pupdate(int n, const double *r_dev, const double *dinv_dev, double beta, int first,
                  double *p_dev);
/* q := A p and out[0] = p.q in one pass (pcg.c:116-117); p has A.ncols entries of which
 * the first p_offset.. rows' worth are the owned ones: out = sum p[p_offset+i]*q[i] */
int psp_k_csr_matvec_dot(psp_csr_t *A, const double *p_dev, int p_offset, double *q_dev,
                         double *out_dev);
/* y := A x split around a halo exchange: rows [row_a, row_b) reference no ghost entry of x
 * and are multiplied first; wait(ctx) must return once the ghost entries have arrived (in
 * stream order); the remaining rows follow.  dot_out_dev != NULL additionally leaves
 * sum x[x_offset+i]*y[i] there (the fused p.q of pcg.c:116-117). */
typedef int (*psp_wait_fn)(void *ctx);
int psp_k_csr_matvec_overlap(psp_csr_t *A, const double *x_dev, int x_offset, double *y_dev,
                             int row_a, int row_b, psp_wait_fn wait, void *ctx, double *dot_out_dev);
/* The same update as psp_k_xr_update in the "lazy x" arrangement: the x update and
 * stagnation scan of the iteration that just finished ride in the p update of the next one.
 *   px: if xpend: scan + x += alpha_x p (p = previous direction); then p := z + beta p (first: p := z);
 *       out[0] = nonstag (meaningful when xpend)
 *   r : r -= alpha q; out = { r.r, r.z }
 *   x : the pending x update on its own (after the loop); out[0] = nonstag */
int psp_k_px_update(int n, const double *r_dev, const double *dinv_dev, double beta, int first, double alpha_x,
                    int xpend, double *p_dev, double *x_dev, double *out_dev);
int psp_k_r_update(int n, double alpha, const double *q_dev, const double *dinv_dev, double *r_dev,
                   double *out_dev);
int psp_k_x_update(int n, double alpha, const double *p_dev, double *x_dev, double *out_dev);
/* stagnation scan + x += alpha p, r -= alpha q (pcg.c:127-143) and
 * out = { r.r, r.z (z = dinv.*r), nonstag } where nonstag != 0 iff 1 + dmax != 1 */
int psp_k_xr_update(int n, double alpha, const double *p_dev, const double *q_dev,
                    const double *dinv_dev, double *x_dev, double *r_dev, double *out_dev);
/* y := x .* dinv (jacobi(), preconmodule.c:41-42) on device vectors */
int psp_k_jacobi(int n, const double *x_dev, const double *dinv_dev, double *y_dev);
/* gather send_dev[i] = v[idx[i]] (halo packing) */
int psp_k_gather(int count, const int *idx_dev, const double *v_dev, double *send_dev);

/* ------------------------------------------ device-resident solver state for the multi-GPU driver
 * The same loops with NO host round trip per reduction: alpha / beta / the exit tests of pcg.c:100-162
 * (resp. the Lanczos / Givens recurrences of minres.c:129-192) live in a small device-side state; the
 * psp_kd_* kernels read their coefficients from it and turn into no-ops once the loop has ended.  One
 * iteration is cut at its two reductions so that an all-reduce can be issued in stream order between
 * "finish the local partial sums" (out_dev) and "take the reference's branches on the reduced values"
 * (psp_kd_*_scalar*).  The host enqueues a batch of iterations, then calls *_fetch once. */
typedef struct psp_pcgstate psp_pcgstate_t;
typedef struct {
  int status;      /* 0 running, 1 finished */
  int info, iter;  /* valid when finished (pend_maxit: decided by the final x update, see below) */
  int it;          /* iteration the enqueued kernels are working on */
  int xpend;       /* x += alpha_x p of the last finished iteration is still to be applied (psp_k_x_update) */
  int stag0;       /* alpha == 0 in that iteration (pcg.c:124-125) */
  int pend_maxit;  /* the loop ran out: -5 or -1 is decided by the stagnation scan of the final x update */
  double relres, normr, n2b, alpha_x;
} psp_pcg_status_t;
int psp_pcgstate_create(psp_pcgstate_t **out);
int psp_pcgstate_destroy(psp_pcgstate_t *st);
/* state at the head of iteration 1: ||b||, tol*||b||, ||r0||, rho0 = r0.z0 (all already reduced over ranks) */
int psp_pcgstate_init(psp_pcgstate_t *st, double n2b, double tolb, double normr0, double rho0, int maxit,
                      int want_hist);
int psp_pcgstate_fetch(psp_pcgstate_t *st, psp_pcg_status_t *out); /* synchronises the stream */
int psp_pcgstate_hist(psp_pcgstate_t *st, int first, int count, double *hist_host);
/* lazy-x arrangement, scalars from the state:
 *   px: pending x update + scan, then p := z + beta p;  out[0] = local nonstag count
 *   matvec_overlap: q := A p around the halo wait;       dot_out[0] = local p.q
 *   scalar_xpq(scal = {p.q, nonstag} after the all-reduce): pcg.c:159-162 of the previous iteration,
 *       :101-112 of this one, alpha (:117-125)
 *   r : r -= alpha q;                                    out = local {r.r, r.z}
 *   scalar_r(scal = {r.r, r.z} after the all-reduce): pcg.c:152-157, next rho / beta */
int psp_kd_px_update(const psp_pcgstate_t *st, int n, const double *r_dev, const double *dinv_dev, double *p_dev,
                     double *x_dev, double *out_dev);
int psp_kd_csr_matvec_overlap(const psp_pcgstate_t *st, psp_csr_t *A, const double *x_dev, int x_offset,
                              double *y_dev, int row_a, int row_b, psp_wait_fn wait, void *ctx,
                              double *dot_out_dev);
int psp_kd_pcg_scalar_xpq(psp_pcgstate_t *st, const double *scal_dev);
int psp_kd_r_update(const psp_pcgstate_t *st, int n, const double *q_dev, const double *dinv_dev, double *r_dev,
                    double *out_dev);
int psp_kd_pcg_scalar_r(psp_pcgstate_t *st, const double *scal_dev);

typedef struct psp_minresstate psp_minresstate_t;
typedef struct {
  int status;  /* ended by -3 / -6 (minres.c:144-146, :160-162), or after the last w/x update */
  int stop;    /* the loop test of minres.c:114 failed: finished once the enqueued w/x update has run */
  int info, iter;
  double relres;   /* written on the 0 / -1 exits only, as in the reference */
  double norm_rmr;
} psp_minres_status_t;
int psp_minresstate_create(psp_minresstate_t **out);
int psp_minresstate_destroy(psp_minresstate_t *st);
/* state at the head of iteration 1: ||r0||, beta = sqrt(v_hat . y) (reduced over ranks), tol, maxit */
int psp_minresstate_init(psp_minresstate_t *st, double norm_r0, double beta0, double errtol, int it_max,
                         int want_hist);
int psp_minresstate_fetch(psp_minresstate_t *st, psp_minres_status_t *out);
int psp_minresstate_hist(psp_minresstate_t *st, int first, int count, double *hist_host);
/* one iteration of minres.c:96-193 on a row block:
 *   scale:   v := y / beta (:123-124)        matvec: Av := A v around the halo wait, dot_out = local v.Av
 *   scalar(0, {v.Av} reduced): alpha, the Lanczos coefficients (:129-131)
 *   lanczos: v_hat_old := Av - c1 v_hat - c2 v_hat_old (the caller swaps the names), y := dinv .* it,
 *            out = local v_hat.y
 *   scalar(1, {v_hat.y} reduced): beta, Givens rotation, the exits, the loop test of the next iteration
 *   wx:      w_old := (v - r3 w_old - r2 w) / r1 (the caller swaps the names); x += c eta w (:172-180) */
int psp_kd_minres_scale(const psp_minresstate_t *st, int n, const double *y_dev, double *v_dev);
int psp_kd_minres_matvec(const psp_minresstate_t *st, psp_csr_t *A, const double *v_dev, int v_offset,
                         double *av_dev, int row_a, int row_b, psp_wait_fn wait, void *ctx, double *dot_out_dev);
int psp_kd_minres_lanczos(const psp_minresstate_t *st, int n, const double *av_dev, const double *v_hat_dev,
                          double *v_hat_old_dev, const double *dinv_dev, double *y_dev, double *out_dev);
int psp_kd_minres_scalar(psp_minresstate_t *st, int which, const double *scal_dev);
int psp_kd_minres_wx(const psp_minresstate_t *st, int n, const double *v_dev, const double *w_dev,
                     double *w_old_dev, double *x_dev);

#ifdef __cplusplus
}
#endif
#endif /* PYSPARSE_HIP_H */
