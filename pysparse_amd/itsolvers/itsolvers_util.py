"""Object-oriented wrappers around the Krylov solvers, Python 3.

Counterpart of pysparse/itsolvers/itsolvers_util.py:13-125 for the two solvers of this
build: `solve()` zeroes x first (:41), keeps the counters nofCalled / totalIterations /
lastIterations / lastInfo (:24-27,43-46) and raises on info < 0 (:49-50)."""
from . import krylov

__all__ = ["ItSolver", "Pcg", "Minres", "Qmrs", "Cgs", "Bicgstab", "Gmres"]


class ItSolver:
    def __init__(self, matrix, **kwargs):
        self.matrix = matrix
        self.name = "Generic"
        self.itsolver = None
        self.nofCalled = 0
        self.totalIterations = 0
        self.lastIterations = 0
        self.lastInfo = 0
        self.relres = None
        self.debug = kwargs.get("debug", False)

    def solve(self, b, x, tol, maxit, K=None, **kwargs):
        if self.itsolver is None:
            raise NotImplementedError("This class cannot be instantiated")
        x[:] = 0.0  # itsolvers_util.py:41: the initial guess is always zero
        if K is None:
            info, it, relres = self.itsolver(self.matrix, b, x, tol, maxit)
        else:
            info, it, relres = self.itsolver(self.matrix, b, x, tol, maxit, K)
        self.nofCalled += 1
        self.totalIterations += it
        self.lastIterations = it
        self.lastInfo = info
        self.relres = relres
        if info < 0:
            raise RuntimeError("%s: info=%d, iter=%d, relres=%g" % (self.name, info, it, relres))
        return None


class Pcg(ItSolver):
    def __init__(self, matrix, **kwargs):
        ItSolver.__init__(self, matrix, **kwargs)
        self.name = "pcg"
        self.itsolver = krylov.pcg


class Minres(ItSolver):
    def __init__(self, matrix, **kwargs):
        ItSolver.__init__(self, matrix, **kwargs)
        self.name = "minres"
        self.itsolver = krylov.minres


class Qmrs(ItSolver):
    def __init__(self, matrix, **kwargs):
        ItSolver.__init__(self, matrix, **kwargs)
        self.name = "qmrs"
        self.itsolver = krylov.qmrs


class Cgs(ItSolver):
    def __init__(self, matrix, **kwargs):
        ItSolver.__init__(self, matrix, **kwargs)
        self.name = "cgs"
        self.itsolver = krylov.cgs


class Bicgstab(ItSolver):
    def __init__(self, matrix, **kwargs):
        ItSolver.__init__(self, matrix, **kwargs)
        self.name = "bicgstab"
        self.itsolver = krylov.bicgstab


class Gmres(ItSolver):
    def __init__(self, matrix, **kwargs):
        ItSolver.__init__(self, matrix, **kwargs)
        self.name = "gmres"
        self.itsolver = krylov.gmres
