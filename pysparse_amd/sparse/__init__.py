"""pysparse_amd.sparse -- counterpart of pysparse.sparse: the `spmatrix` extension module
(ll_mat feeder, csr_mat, sss_mat) with every matrix-vector product on the GPU."""
from . import spmatrix  # noqa: F401
from . import pysparseMatrix  # noqa: F401,E402
from .pysparseMatrix import PysparseIdentityMatrix, PysparseMatrix, PysparseSpDiagsMatrix  # noqa: F401,E402
