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
// ONE translation unit in six parts (round 6; it was 5 800 lines in one file), included below in this order:
//     psp_csr_kernels.h      the SpMV kernels (stream, wave, w1, w2, w3, w6, w5; the index-free family w4 / w4x / w4y,
//                            sss_spmv_w4, csr_spmv_w4_pf, csr_spmv_w4_transp) and the kernels that build their tables
//     psp_csr_generators.h   Poisson generators, random banded rows
//     psp_csr_hostutil.h     variant decoding, launch wrappers
//     psp_csr_tables.h       per-handle side tables (ChunkTable, CsrExtra) and their lazy builders, the transposes, the
//                            renumbered copy and its cost rule
//     psp_csr_select.h       csr_spmv_launch (which kernel a handle gets) and the variants the solvers call, the
//                            halo-overlap split, host-pointer staging
//     psp_csr_abi.h          the C ABI: psp_csr_*, psp_sss_*
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
#include "psp_csr_kernels.h"
#include "psp_csr_generators.h"
#include "psp_csr_hostutil.h"
}  // namespace

#include "psp_csr_tables.h"
#include "psp_csr_select.h"
#include "psp_csr_abi.h"
