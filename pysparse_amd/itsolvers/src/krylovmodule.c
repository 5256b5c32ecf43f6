/*
 * pysparse_amd.itsolvers.krylov -- info, iter, relres = pcg|minres(A, b, x, tol, maxit[, K])
 *
 * Python-3 counterpart of ItSolvers_pcg / ItSolvers_minres
 * (pysparse/itsolvers/src/itsolversmodule.c:32-118, :217-305).  The loops themselves
 * (pcg.c:22-171, minres.c:43-200) run device-resident in libpysparse_hip.so; this file only
 * parses arguments the way the reference does, turns A and K into operators and builds the
 * result tuple.  Native operands keep the whole solve on the GPU with the GIL released;
 * duck-typed Python operators are called back once per application with the GIL held.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

#include <string.h>

#include "psp_pyops.h"

/* which kernel run_solver dispatches to */
enum { SOLVER_PCG, SOLVER_MINRES, SOLVER_CGS, SOLVER_BICGSTAB, SOLVER_QMRS, SOLVER_GMRES };

static int dispatch(int which, int gmres_dim, const psp_op_t *A, const psp_op_t *K, int n, double *x,
                    const double *b, double tol, int maxit, int *info, int *iter, double *relres) {
  switch (which) {
    case SOLVER_PCG: return psp_pcg(A, K, n, x, b, tol, maxit, info, iter, relres, NULL);
    case SOLVER_MINRES: return psp_minres(A, K, n, x, b, tol, maxit, info, iter, relres, NULL);
    case SOLVER_CGS: return psp_cgs(A, K, n, x, b, tol, maxit, info, iter, relres);
    case SOLVER_BICGSTAB: return psp_bicgstab(A, K, n, x, b, tol, maxit, info, iter, relres);
    case SOLVER_QMRS: return psp_qmrs(A, K, n, x, b, tol, maxit, info, iter, relres);
    default: return psp_gmres(A, K, n, x, b, tol, maxit, gmres_dim, info, iter, relres);
  }
}

static PyObject *run_solver(PyObject *args, int which, int gmres_dim, int check_positive_shape) {
  PyObject *amat, *bo, *xo, *precon = Py_None;
  PyArrayObject *b = NULL, *x = NULL;
  double tol, relres = 0.0;
  int maxit, info = 0, iter = 0, n = 0, nk = 0, rc;
  PyOpRef aref, kref;
  PyObject *result = NULL;
  int have_a = 0, have_k = 0;

  if (!PyArg_ParseTuple(args, "OOOdi|O", &amat, &bo, &xo, &tol, &maxit, &precon)) return NULL;

  /* check shape of matrix object (itsolversmodule.c:65-67 / :245-250) */
  if (pyop_acquire(amat, 0, &aref, &n)) return NULL;
  have_a = 1;
  if (check_positive_shape && n <= 0) {
    PyErr_SetString(PyExc_ValueError, "invalid matrix shape");
    goto done;
  }

  /* x and b as contiguous double arrays (itsolversmodule.c:70-82): x is solved in place only
   * when it already is one; any other input is converted and the copy discarded */
  x = (PyArrayObject *)PyArray_FROM_OTF(xo, NPY_DOUBLE, NPY_ARRAY_CARRAY);
  if (x == NULL) {
    PyErr_SetString(PyExc_ValueError, "Unable to convert x to double array");
    goto done;
  }
  b = (PyArrayObject *)PyArray_FROM_OTF(bo, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY);
  if (b == NULL) {
    PyErr_SetString(PyExc_ValueError, "Unable to convert b to double array");
    goto done;
  }
  if (PyArray_NDIM(x) != 1 || PyArray_NDIM(b) != 1 || PyArray_DIM(x, 0) != PyArray_DIM(b, 0) ||
      PyArray_DIM(x, 0) != n) {
    PyErr_SetString(PyExc_ValueError, "incompatible operand shapes");
    goto done;
  }
  if (precon != Py_None) {
    if (pyop_acquire(precon, 1, &kref, &nk)) goto done;
    have_k = 1;
    if (nk != n) {
      PyErr_SetString(PyExc_ValueError, "incompatible operand shapes");
      goto done;
    }
  }

  /* the GIL is released for every solve; callback operators take it back in pyop_trampoline */
  Py_BEGIN_ALLOW_THREADS
  rc = dispatch(which, gmres_dim, aref.op, have_k ? kref.op : NULL, n, (double *)PyArray_DATA(x),
                (const double *)PyArray_DATA(b), tol, maxit, &info, &iter, &relres);
  Py_END_ALLOW_THREADS
  if (PyErr_Occurred()) goto done; /* a callback raised (itsolversmodule.c:114-115) */
  if (rc != PSP_OK) {
    PyErr_SetString(rc == PSP_ENOMEM ? PyExc_MemoryError
                                     : (rc == PSP_EINVAL ? PyExc_ValueError : PyExc_RuntimeError),
                    psp_last_error());
    goto done;
  }
  result = Py_BuildValue("iid", info, iter, relres);
done:
  if (have_k) pyop_release(&kref);
  if (have_a) pyop_release(&aref);
  Py_XDECREF(x);
  Py_XDECREF(b);
  return result;
}

static PyObject *ItSolvers_pcg(PyObject *self, PyObject *args) {
  return run_solver(args, SOLVER_PCG, 0, 0);
}

static PyObject *ItSolvers_minres(PyObject *self, PyObject *args) {
  return run_solver(args, SOLVER_MINRES, 0, 1);
}

static PyObject *ItSolvers_cgs(PyObject *self, PyObject *args) {
  return run_solver(args, SOLVER_CGS, 0, 1);
}
static PyObject *ItSolvers_bicgstab(PyObject *self, PyObject *args) {
  return run_solver(args, SOLVER_BICGSTAB, 0, 0);
}
static PyObject *ItSolvers_qmrs(PyObject *self, PyObject *args) {
  return run_solver(args, SOLVER_QMRS, 0, 1);
}
/* gmres(A, b, x, tol, maxit[, K[, dim=20]]): itsolversmodule.c:313-403 */
static PyObject *ItSolvers_gmres(PyObject *self, PyObject *args) {
  Py_ssize_t na = PyTuple_GET_SIZE(args);
  PyObject *core, *res;
  int dim = 20;
  if (na == 7) {
    dim = (int)PyLong_AsLong(PyTuple_GET_ITEM(args, 6));
    if (PyErr_Occurred()) return NULL;
    core = PyTuple_GetSlice(args, 0, 6);
  } else {
    core = args;
    Py_INCREF(core);
  }
  if (core == NULL) return NULL;
  res = run_solver(core, SOLVER_GMRES, dim, 1);
  Py_DECREF(core);
  return res;
}

static PyMethodDef krylov_methods[] = {
    {"cgs", ItSolvers_cgs, METH_VARARGS,
     "info, iter, relres = cgs(A, b, x, tol, maxit[, K])\n\nConjugate Gradient Squared method."},
    {"bicgstab", ItSolvers_bicgstab, METH_VARARGS,
     "info, iter, relres = bicgstab(A, b, x, tol, maxit[, K])\n\nStabilized BiConjugate Gradient method."},
    {"qmrs", ItSolvers_qmrs, METH_VARARGS,
     "info, iter, relres = qmrs(A, b, x, tol, maxit[, K])\n\nQuasi-Minimal Residual Smoothing method."},
    {"gmres", ItSolvers_gmres, METH_VARARGS,
     "info, iter, relres = gmres(A, b, x, tol, maxit[, K[, dim]])\n\nGMRES(dim) of Saad and Schultz."},
    {"pcg", ItSolvers_pcg, METH_VARARGS,
     "info, iter, relres = pcg(A, b, x, tol, maxit[, K])\n\nPreconditioned Conjugate Gradient method."},
    {"minres", ItSolvers_minres, METH_VARARGS,
     "info, iter, relres = minres(A, b, x, tol, maxit[, K])\n\nMinimal Residual method."},
    {NULL, NULL, 0, NULL}};

/* itsolversmodule.c:625-646 */
static const char krylov_doc[] =
    "Iterative solvers on MI355X.  info >= 0: converged; -1: maxit reached; -2: ill-conditioned\n"
    "preconditioner; -3: preconditioner not SPD (minres); -5: stagnation; -6: breakdown.";

static struct PyModuleDef krylov_module = {PyModuleDef_HEAD_INIT, "krylov", krylov_doc, -1,
                                           krylov_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit_krylov(void) {
  import_array();
  if (import_spmatrix() < 0) return NULL;
  return PyModule_Create(&krylov_module);
}
