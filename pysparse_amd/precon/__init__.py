"""pysparse_amd.precon -- counterpart of pysparse.precon: the `precon` extension module
with `jacobi(A, omega=1.0, steps=1)`."""
from . import precon  # noqa: F401
from .precon import jacobi  # noqa: F401
