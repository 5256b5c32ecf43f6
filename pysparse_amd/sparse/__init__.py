"""pysparse_amd.sparse -- counterpart of pysparse.sparse: the `spmatrix` extension module
(ll_mat feeder, csr_mat, sss_mat) with every matrix-vector product on the GPU."""
from . import spmatrix  # noqa: F401
