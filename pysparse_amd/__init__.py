"""pysparse_amd -- MI355X (gfx950) implementation of PySparse's SpMV + Krylov hot path.

Layout
  csrc/                 hand-written HIP kernels + the C ABI (include/pysparse_hip.h)
  libpysparse_hip.so    built by __graft_entry__.build() (hipcc --offload-arch=gfx950)
  device.py, _capi.py   ctypes handles over the C ABI (bench, GPU tests, multi-GPU driver)
  sparse/ itsolvers/ precon/   the drop-in CPython extension modules
                        (spmatrix, krylov, precon) with the reference's names
"""
__version__ = "0.1.0"
