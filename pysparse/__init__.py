"""Drop-in alias: `from pysparse.sparse import spmatrix`, `from pysparse.itsolvers.krylov
import pcg`, `from pysparse.precon import precon` resolve to the MI355X implementation in
pysparse_amd (same module names as PythonOptimizers/pysparse for the SpMV + Krylov path)."""
import sys

import pysparse_amd
from pysparse_amd import itsolvers, precon, sparse, tools
from pysparse_amd.itsolvers import krylov
from pysparse_amd.precon import precon as _precon_mod
from pysparse_amd.sparse import spmatrix

for _name, _mod in (("pysparse.sparse", sparse), ("pysparse.sparse.spmatrix", spmatrix),
                    ("pysparse.itsolvers", itsolvers), ("pysparse.itsolvers.krylov", krylov),
                    ("pysparse.precon", precon), ("pysparse.precon.precon", _precon_mod),
                    ("pysparse.sparse.pysparseMatrix", sparse.pysparseMatrix), ("pysparse.tools", tools), ("pysparse.tools.poisson", tools.poisson),
                    ("pysparse.tools.poisson_vec", tools.poisson_vec), ("pysparse.tools.sptime", tools.sptime)):
    sys.modules[_name] = _mod
__version__ = pysparse_amd.__version__
