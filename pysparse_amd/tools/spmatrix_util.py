"""pysparse.tools.spmatrix_util for Python 3: the byte formatter, the matrix statistics printer and the random ll_mat
generator (pysparse/tools/spmatrix_util.py:4-56).  The VTK viewers of the reference (:58-100) are not part of this build."""
import random

from ..sparse import spmatrix


def bytesToString(n):
    for unit in ("Bytes", "Kbytes", "Mbytes"):
        if n < 1024:
            return ("%d %s" if unit == "Bytes" else "%.1f %s") % (n, unit)
        n /= 1024.0
    return "%.1f Gbytes" % n


def printInfo(mat, name):
    if isinstance(mat, spmatrix.LLMatType):
        typeName = "LL symmetric" if mat.issym else "LL general"
        storage = bytesToString(4 * mat.shape[0] + 16 * mat.nnz)
    elif isinstance(mat, spmatrix.SSSMatType):
        typeName = "SSS"
        storage = bytesToString(12 * mat.shape[0] + 12 * (mat.nnz - mat.shape[0]))
    elif isinstance(mat, spmatrix.CSRMatType):
        typeName = "CSR"
        storage = bytesToString(4 * mat.shape[0] + 12 * mat.nnz)
    else:
        typeName = storage = "Unknown"
    print("Matrix statistics:")
    for key, value in (("name", name), ("type", typeName), ("dimensions", "%dx%d" % tuple(mat.shape)),
                       ("#non-zeros", "%d" % mat.nnz), ("storage", storage)):
        print("%-20s: %s" % (key, value))
    print()


def ll_mat_rand(n, m, density):
    """a general n-by-m ll_mat with at most n*m*density entries, values in [0.0, 1.0)"""
    nnz = int(density * n * m)
    A = spmatrix.ll_mat(n, m, max(nnz, 1))
    for _ in range(nnz):
        A[random.randrange(n), random.randrange(m)] = random.random()
    return A
