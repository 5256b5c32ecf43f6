"""pysparse_amd.itsolvers -- counterpart of pysparse.itsolvers: `krylov.pcg`, `krylov.minres`
(device-resident loops) and the thin ItSolver wrappers of itsolvers_util.py."""
from . import krylov  # noqa: F401
from .krylov import bicgstab, cgs, gmres, minres, pcg, qmrs  # noqa: F401
from .itsolvers_util import Bicgstab, Cgs, Gmres, ItSolver, Minres, Pcg, Qmrs  # noqa: F401
