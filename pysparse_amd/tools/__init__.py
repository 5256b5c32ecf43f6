"""pysparse_amd.tools -- counterpart of pysparse.tools for the SpMV + Krylov path: the Poisson
generators (element-wise and vectorised, pysparse/tools/poisson.py and poisson_vec.py) and the CPU
timer (sptime.py).  `from pysparse.tools import poisson` keeps working through the alias package."""
from . import mtx, poisson, poisson_vec, spmatrix_util, sptime  # noqa: F401
from .poisson import *  # noqa: F401,F403
from .poisson_vec import *  # noqa: F401,F403
from .sptime import cputime  # noqa: F401
from .mtx import csr_from_mtx, sss_from_mtx  # noqa: F401
