/*
 * pysparse_amd.sparse.spmatrix -- Python-3 extension module with the reference's names:
 * ll_mat (feeder), csr_mat, sss_mat (matvec only), the operator-protocol shims and the
 * inter-module C API table.  Written from scratch against the NumPy-2 C API; every
 * matrix-vector product runs in libpysparse_hip.so on the GPU (there is no CPU matvec
 * here -- ll_mat.matvec multiplies with a device CSR mirror of the linked list).
 *
 * Reference interfaces mirrored (file:line under the reference tree):
 *   module functions ll_mat / ll_mat_sym / ll_mat_from_mtx   spmatrixmodule.c:317-357
 *   SpMatrix_GetShape/GetOrder/GetItem/Matvec/Precon           spmatrixmodule.c:86-248
 *   ll_mat get/set item, put, to_csr, to_sss                   ll_mat.c:210-356,1577-1708,2497-2752
 *   csr_mat (shape, nnz, matvec, matvec_transp)                csr_mat.c:114-231
 *   sss_mat (shape, nnz, matvec, A[i,j])                       sss_mat.c:14-28,78-172
 * Additions (needed to feed 10^8-row problems):
 *   csr_from_arrays, sss_from_arrays, poisson_csr, poisson_sss, csr_mat.to_arrays(),
 *   sss_mat.to_arrays().
 */
#define PY_SSIZE_T_CLEAN
#include <stdarg.h>
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <unistd.h>

#define SPMATRIX_MODULE
#include "spmatrix_api.h"

static PyObject *SpMatrix_ErrorObject;
static PyTypeObject LLMatType, CSRMatType, SSSMatType;

#define INCREASE_FACTOR 1.5 /* ll_mat.c: growth of the entry arrays */

static PyObject *psp_raise(int rc) {
  const char *msg = psp_last_error();
  if (rc == PSP_ENOMEM)
    PyErr_SetString(PyExc_MemoryError, msg);
  else if (rc == PSP_EINVAL)
    PyErr_SetString(PyExc_ValueError, msg);
  else
    PyErr_SetString(PyExc_RuntimeError, msg);
  return NULL;
}

/* ------------------------------------------------------------------ vector argument parsing */

/* SPMATRIX_PARSE_ARGS_ARR_ARR_STRIDE (pysparse/include/spmatrix.h:38-54): two 1-D float64
 * ndarrays of the given lengths; strides allowed. */
static int parse_arr_arr_stride(PyObject *args, PyArrayObject **xp, PyArrayObject **yp, npy_intp n1,
                                npy_intp n2) {
  if (!PyArg_ParseTuple(args, "O!O!", &PyArray_Type, xp, &PyArray_Type, yp)) return -1;
  if (PyArray_NDIM(*xp) != 1 || PyArray_TYPE(*xp) != NPY_DOUBLE || PyArray_DIM(*xp, 0) != n1) {
    PyErr_SetString(PyExc_ValueError,
                    "arg 1 must be a 1-dimensional double array of appropriate size.");
    return -1;
  }
  if (PyArray_NDIM(*yp) != 1 || PyArray_TYPE(*yp) != NPY_DOUBLE || PyArray_DIM(*yp, 0) != n2) {
    PyErr_SetString(PyExc_ValueError,
                    "arg 2 must be a 1-dimensional double array of appropriate size.");
    return -1;
  }
  if (!PyArray_ISWRITEABLE(*yp)) {
    PyErr_SetString(PyExc_ValueError, "arg 2 must be writeable.");
    return -1;
  }
  return 0;
}

#define ELEM_STRIDE(a) ((ptrdiff_t)(PyArray_STRIDE((a), 0) / (npy_intp)sizeof(double)))

/* ------------------------------------------------------------------ C API shims */

/* spmatrixmodule.c:86-104 */
static int SpMatrix_GetShape(PyObject *op, int dim[]) {
  PyObject *sh, *elem;
  long v;
  if ((sh = PyObject_GetAttrString(op, "shape")) == NULL) return -1;
  if (PySequence_Size(sh) != 2) {
    Py_DECREF(sh);
    PyErr_SetString(PyExc_ValueError, "invalid matrix shape");
    return -1;
  }
  for (int k = 0; k < 2; k++) {
    elem = PySequence_GetItem(sh, k);
    v = elem ? PyLong_AsLong(elem) : -1;
    Py_XDECREF(elem);
    dim[k] = (int)v;
  }
  Py_DECREF(sh);
  if (PyErr_Occurred() != NULL) {
    PyErr_SetString(PyExc_ValueError, "invalid matrix shape");
    return -1;
  }
  return 0;
}

/* spmatrixmodule.c:118-132 */
static int SpMatrix_GetOrder(PyObject *op, int *n) {
  int dim[2];
  if (SpMatrix_GetShape(op, dim) == -1) return -1;
  if (dim[0] != dim[1]) {
    PyErr_SetString(PyExc_ValueError, "matrix is not square");
    return -1;
  }
  *n = dim[0];
  return 0;
}

/* spmatrixmodule.c:141-156: op[i,j] as a double; 0.0 with the exception pending on failure */
static double SpMatrix_GetItem(PyObject *op, int i, int j) {
  PyObject *index = Py_BuildValue("(ii)", i, j), *fo;
  double d;
  if (index == NULL) return 0.0;
  fo = PyObject_GetItem(op, index);
  Py_DECREF(index);
  if (fo == NULL) return 0.0;
  d = PyFloat_AsDouble(fo);
  Py_DECREF(fo);
  return d;
}

static int call_vec_method(PyObject *obj, const char *name, int nx, double *x, int ny, double *y) {
  PyObject *xa = NULL, *ya = NULL, *res;
  npy_intp dims[1];
  dims[0] = nx;
  xa = PyArray_SimpleNewFromData(1, dims, NPY_DOUBLE, (void *)x);
  if (xa == NULL) goto fail;
  dims[0] = ny;
  ya = PyArray_SimpleNewFromData(1, dims, NPY_DOUBLE, (void *)y);
  if (ya == NULL) goto fail;
  res = PyObject_CallMethod(obj, name, "OO", xa, ya);
  if (res == NULL) goto fail;
  Py_DECREF(res);
  Py_DECREF(xa);
  Py_DECREF(ya);
  return 0;
fail:
  Py_XDECREF(xa);
  Py_XDECREF(ya);
  return -1;
}

/* spmatrixmodule.c:169-201 */
static int SpMatrix_Matvec(PyObject *matrix, int nx, double *x, int ny, double *y) {
  return call_vec_method(matrix, "matvec", nx, x, ny, y);
}

/* spmatrixmodule.c:215-248 */
static int SpMatrix_Precon(PyObject *prec, int n, double *x, double *y) {
  return call_vec_method(prec, "precon", n, x, n, y);
}

/* spmatrix.h:18-36 as a function: two CONTIGUOUS float64 vectors of length n */
static int SpMatrix_ParseVecOpArgs(PyObject *args, double **x_data, double **y_data, int n) {
  PyArrayObject *xp, *yp;
  if (!PyArg_ParseTuple(args, "O!O!", &PyArray_Type, &xp, &PyArray_Type, &yp)) return -1;
  if (PyArray_NDIM(xp) != 1 || PyArray_TYPE(xp) != NPY_DOUBLE || PyArray_DIM(xp, 0) != n ||
      !PyArray_IS_C_CONTIGUOUS(xp)) {
    PyErr_SetString(PyExc_ValueError,
                    "arg 1 must be a contiguous 1-dimensional double array of appropriate size.");
    return -1;
  }
  if (PyArray_NDIM(yp) != 1 || PyArray_TYPE(yp) != NPY_DOUBLE || PyArray_DIM(yp, 0) != n ||
      !PyArray_IS_C_CONTIGUOUS(yp) || !PyArray_ISWRITEABLE(yp)) {
    PyErr_SetString(PyExc_ValueError,
                    "arg 2 must be a contiguous 1-dimensional double array of appropriate size.");
    return -1;
  }
  *x_data = (double *)PyArray_DATA(xp);
  *y_data = (double *)PyArray_DATA(yp);
  return 0;
}

/* spmatrixmodule.c:262-310: info, iter, relres = linsolver(A, b, x, tol, itmax[, K]) */
static int ItSolvers_Solve(PyObject *linsolver, PyObject *A, int n, double *b, double *x, double tol,
                           int itmax, PyObject *K, int *info, int *iter, double *relres) {
  PyObject *ba = NULL, *xa = NULL, *res = NULL;
  npy_intp dims[1];
  int rc = -1;
  dims[0] = n;
  ba = PyArray_SimpleNewFromData(1, dims, NPY_DOUBLE, (void *)b);
  xa = PyArray_SimpleNewFromData(1, dims, NPY_DOUBLE, (void *)x);
  if (ba == NULL || xa == NULL) goto done;
  if (K == NULL)
    res = PyObject_CallFunction(linsolver, "OOOdi", A, ba, xa, tol, itmax);
  else
    res = PyObject_CallFunction(linsolver, "OOOdiO", A, ba, xa, tol, itmax, K);
  if (res == NULL) goto done;
  if (PyArg_ParseTuple(res, "iid", info, iter, relres)) rc = 0;
done:
  Py_XDECREF(res);
  Py_XDECREF(ba);
  Py_XDECREF(xa);
  return rc;
}

/* ------------------------------------------------------------------ ll_mat core */

static void ll_invalidate(LLMatObject *a) {
  if (a->op) {
    psp_op_destroy(a->op);
    a->op = NULL;
  }
  if (a->mirror) {
    psp_csr_destroy(a->mirror);
    a->mirror = NULL;
  }
}

/* ll_mat.c:3332-3388 */
static PyObject *SpMatrix_NewLLMatObject(int dim[], int sym, int sizeHint, int storeZeros) {
  LLMatObject *op;
  int i;
  if (dim[0] < 0 || dim[1] < 0) {
    PyErr_SetString(PyExc_ValueError, "matrix dimension must be non-negative");
    return NULL;
  }
  if (sizeHint < 1) sizeHint = 1;
  op = PyObject_New(LLMatObject, &LLMatType);
  if (op == NULL) return PyErr_NoMemory();
  op->val = NULL;
  op->col = op->link = op->root = NULL;
  op->mirror = NULL;
  op->op = NULL;
  op->root = PyMem_New(int, dim[0] > 0 ? dim[0] : 1);
  op->val = PyMem_New(double, sizeHint);
  op->col = PyMem_New(int, sizeHint);
  op->link = PyMem_New(int, sizeHint);
  if (!op->root || !op->val || !op->col || !op->link) {
    Py_DECREF(op);
    return PyErr_NoMemory();
  }
  for (i = 0; i < dim[0]; i++) op->root[i] = -1;
  op->dim[0] = dim[0];
  op->dim[1] = dim[1];
  op->issym = sym;
  op->storeZeros = storeZeros;
  op->nnz = 0;
  op->nalloc = sizeHint;
  op->free = -1;
  return (PyObject *)op;
}

/* ll_mat.c:210-244 */
static double SpMatrix_LLMatGetItem(LLMatObject *a, int i, int j) {
  int k, t;
  if (i < 0 || i >= a->dim[0] || j < 0 || j >= a->dim[1]) {
    PyErr_SetString(PyExc_IndexError, "indices out of range");
    return 0.0;
  }
  if (a->issym && i < j) {
    t = i;
    i = j;
    j = t;
  }
  for (k = a->root[i]; k != -1; k = a->link[k])
    if (a->col[k] == j) return a->val[k];
  return 0.0;
}

static int ll_grow(LLMatObject *a) {
  int nalloc_new = (int)(INCREASE_FACTOR * a->nalloc) + 1;
  void *t;
  if ((t = PyMem_Resize(a->col, int, nalloc_new)) == NULL) return -1;
  a->col = (int *)t;
  if ((t = PyMem_Resize(a->link, int, nalloc_new)) == NULL) return -1;
  a->link = (int *)t;
  if ((t = PyMem_Resize(a->val, double, nalloc_new)) == NULL) return -1;
  a->val = (double *)t;
  a->nalloc = nalloc_new;
  return 0;
}

/* shared by set (add == 0) and update-add (add == 1): ll_mat.c:250-356, :362-460.
 * Rows stay sorted by ascending column; a zero result deletes the entry unless storeZeros. */
static int ll_store(LLMatObject *a, int i, int j, double x, int add) {
  int k, new_elem, last, col;
  if (a->issym && i < j) {
    PyErr_SetString(PyExc_IndexError, "write operation to upper triangle of symmetric matrix");
    return -1;
  }
  if (i < 0 || i >= a->dim[0] || j < 0 || j >= a->dim[1]) {
    PyErr_SetString(PyExc_IndexError, "indices out of range");
    return -1;
  }
  ll_invalidate(a);
  col = last = -1;
  k = a->root[i];
  while (k != -1) {
    col = a->col[k];
    if (col >= j) break;
    last = k;
    k = a->link[k];
  }
  if (add) {
    if (x == 0.0 && a->storeZeros == 0) return 0; /* ll_mat.c:374 */
    if (col == j && k != -1) x += a->val[k];
  }
  if (x != 0.0 || a->storeZeros == 1) {
    if (col == j && k != -1) {
      a->val[k] = x;
    } else {
      if (a->free != -1) {
        new_elem = a->free;
        a->free = a->link[new_elem];
      } else {
        new_elem = a->nnz;
        if (a->nnz == a->nalloc && ll_grow(a) < 0) {
          PyErr_NoMemory();
          return -1;
        }
      }
      a->val[new_elem] = x;
      a->col[new_elem] = j;
      a->link[new_elem] = k;
      if (last == -1)
        a->root[i] = new_elem;
      else
        a->link[last] = new_elem;
      a->nnz++;
    }
  } else if (col == j && k != -1) {
    if (last == -1)
      a->root[i] = a->link[k];
    else
      a->link[last] = a->link[k];
    a->link[k] = a->free;
    a->free = k;
    a->nnz--;
  }
  return 0;
}

static int SpMatrix_LLMatSetItem(LLMatObject *a, int i, int j, double x) {
  return ll_store(a, i, j, x, 0);
}

static int SpMatrix_LLMatUpdateItemAdd(LLMatObject *a, int i, int j, double x) {
  return ll_store(a, i, j, x, 1);
}

/* ll_mat.c:135-184: (root,row,link) lists per column, built bottom-up so that every
 * column list is sorted by ascending row */
static int SpMatrix_LLMatBuildColIndex(struct llColIndex **idx, LLMatObject *self,
                                       int includeDiagonal) {
  int i, j, k;
  struct llColIndex *c = (struct llColIndex *)calloc(1, sizeof(struct llColIndex));
  if (c == NULL) goto fail;
  c->link = PyMem_New(int, self->nalloc > 0 ? self->nalloc : 1);
  c->row = PyMem_New(int, self->nalloc > 0 ? self->nalloc : 1);
  c->root = PyMem_New(int, self->dim[1] > 0 ? self->dim[1] : 1);
  if (!c->link || !c->row || !c->root) goto fail;
  for (i = 0; i < self->dim[1]; i++) c->root[i] = -1;
  for (i = self->dim[0] - 1; i >= 0; i--)
    for (k = self->root[i]; k != -1; k = self->link[k]) {
      j = self->col[k];
      if (i > j)
        c->nzLo++;
      else if (i == j)
        c->nzDiag++;
      else
        c->nzUp++;
      if (includeDiagonal || i != j) {
        c->link[k] = c->root[j];
        c->root[j] = k;
        c->row[k] = i;
      }
    }
  *idx = c;
  return 0;
fail:
  if (c) {
    PyMem_Del(c->link);
    PyMem_Del(c->row);
    PyMem_Del(c->root);
    free(c);
  }
  *idx = NULL;
  PyErr_NoMemory();
  return 1;
}

static void SpMatrix_LLMatDestroyColIndex(struct llColIndex **idx) {
  if (*idx != NULL) {
    PyMem_Del((*idx)->link);
    PyMem_Del((*idx)->row);
    PyMem_Del((*idx)->root);
    free(*idx);
    *idx = NULL;
  }
}

/* ------------------------------------------------------------------ csr / sss construction */

static PyObject *newCSRMatObject(int dim[], int nnz, int host_arrays) {
  CSRMatObject *op = PyObject_New(CSRMatObject, &CSRMatType);
  if (op == NULL) return PyErr_NoMemory();
  op->dim[0] = dim[0];
  op->dim[1] = dim[1];
  op->nnz = nnz;
  op->val = NULL;
  op->col = op->ind = NULL;
  op->dev = NULL;
  op->op = NULL;
  if (host_arrays) {
    op->ind = PyMem_New(int, dim[0] + 1);
    op->val = PyMem_New(double, nnz > 0 ? nnz : 1);
    op->col = PyMem_New(int, nnz > 0 ? nnz : 1);
    if (!op->ind || !op->val || !op->col) {
      Py_DECREF(op);
      return PyErr_NoMemory();
    }
  }
  return (PyObject *)op;
}

static PyObject *newSSSMatObject(int n, int nnz, int host_arrays) {
  SSSMatObject *op = PyObject_New(SSSMatObject, &SSSMatType);
  if (op == NULL) return PyErr_NoMemory();
  op->n = n;
  op->nnz = nnz;
  op->val = op->diag = NULL;
  op->col = op->ind = NULL;
  op->dev = NULL;
  op->op = NULL;
  if (host_arrays) {
    op->ind = PyMem_New(int, n + 1);
    op->diag = PyMem_New(double, n > 0 ? n : 1);
    op->val = PyMem_New(double, nnz > 0 ? nnz : 1);
    op->col = PyMem_New(int, nnz > 0 ? nnz : 1);
    if (!op->ind || !op->diag || !op->val || !op->col) {
      Py_DECREF(op);
      return PyErr_NoMemory();
    }
  }
  return (PyObject *)op;
}

/* fill host CSR arrays from the linked list: LLMat_to_csr, ll_mat.c:1577-1648 */
static int ll_fill_csr(LLMatObject *self, double *val, int *col, int *ind) {
  int i, k, r = 0;
  ind[0] = 0;
  if (self->issym) {
    struct llColIndex *ci;
    if (SpMatrix_LLMatBuildColIndex(&ci, self, 0)) return -1;
    for (i = 0; i < self->dim[0]; i++) {
      for (k = self->root[i]; k != -1; k = self->link[k]) { /* stored lower part + diagonal */
        val[r] = self->val[k];
        col[r] = self->col[k];
        r++;
      }
      for (k = ci->root[i]; k != -1; k = ci->link[k]) { /* mirrored entries, ascending */
        val[r] = self->val[k];
        col[r] = ci->row[k];
        r++;
      }
      ind[i + 1] = r;
    }
    SpMatrix_LLMatDestroyColIndex(&ci);
  } else {
    for (i = 0; i < self->dim[0]; i++) {
      for (k = self->root[i]; k != -1; k = self->link[k]) {
        val[r] = self->val[k];
        col[r] = self->col[k];
        r++;
      }
      ind[i + 1] = r;
    }
  }
  return r;
}

static int ll_csr_nnz(LLMatObject *self) {
  int i, k, lo = 0, dg = 0;
  if (!self->issym) return self->nnz;
  for (i = 0; i < self->dim[0]; i++)
    for (k = self->root[i]; k != -1; k = self->link[k]) {
      if (i > self->col[k])
        lo++;
      else if (i == self->col[k])
        dg++;
    }
  return 2 * lo + dg; /* ll_mat.c:1592-1593 */
}

static int parse_devices(PyObject *o, int *dev, int cap);

/* to_csr() as in the reference (ll_mat.c:1577-1648); to_csr(devices=[...]) puts the rows on several GPUs as row blocks
 * (psp_csr_create_multi): matvec, precon.jacobi, krylov.pcg and krylov.minres then run on all of them */
static PyObject *LLMat_to_csr(LLMatObject *self, PyObject *args, PyObject *kwds) {
  CSRMatObject *op;
  int rc, ndev, devs[64];
  PyObject *odev = NULL;
  static char *kwlist[] = {"devices", NULL};
  if (!PyArg_ParseTupleAndKeywords(args, kwds, "|O", kwlist, &odev)) return NULL;
  if ((ndev = parse_devices(odev, devs, 64)) < 0) return NULL;
  op = (CSRMatObject *)newCSRMatObject(self->dim, ll_csr_nnz(self), 1);
  if (op == NULL) return NULL;
  if (ll_fill_csr(self, op->val, op->col, op->ind) < 0) {
    Py_DECREF(op);
    return NULL;
  }
  Py_BEGIN_ALLOW_THREADS
  if (ndev > 0)
    rc = psp_csr_create_multi(op->dim[0], op->dim[1], op->nnz, op->ind, op->col, op->val, devs, ndev, &op->dev);
  else
    rc = psp_csr_create(op->dim[0], op->dim[1], op->nnz, op->ind, op->col, op->val, &op->dev);
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) {
    Py_DECREF(op);
    return psp_raise(rc);
  }
  return (PyObject *)op;
}

/* LLMat_to_sss, ll_mat.c:1654-1708 */
static PyObject *LLMat_to_sss(LLMatObject *self, PyObject *args) {
  SSSMatObject *op;
  int i, j, k, r, n, nnz, rc;
  if (!PyArg_ParseTuple(args, "")) return NULL;
  n = self->dim[0];
  if (n != self->dim[1]) {
    PyErr_SetString(PyExc_ValueError, "Matrix must be square");
    return NULL;
  }
  nnz = 0;
  for (i = 0; i < n; i++)
    for (k = self->root[i]; k != -1; k = self->link[k])
      if (i > self->col[k]) nnz++;
  op = (SSSMatObject *)newSSSMatObject(n, nnz, 1);
  if (op == NULL) return NULL;
  for (i = 0; i < n; i++) op->diag[i] = 0.0;
  r = 0;
  op->ind[0] = 0;
  for (i = 0; i < n; i++) {
    for (k = self->root[i]; k != -1; k = self->link[k]) {
      j = self->col[k];
      if (i > j) {
        op->val[r] = self->val[k];
        op->col[r] = j;
        r++;
      } else if (i == j)
        op->diag[i] = self->val[k];
    }
    op->ind[i + 1] = r;
  }
  Py_BEGIN_ALLOW_THREADS
  rc = psp_sss_create(n, nnz, op->ind, op->col, op->val, op->diag, &op->dev);
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) {
    Py_DECREF(op);
    return psp_raise(rc);
  }
  return (PyObject *)op;
}

/* host-only exports of the two conversions (no device involved): the arrays to_csr() /
 * to_sss() upload, as NumPy arrays */
static PyObject *LLMat_to_csr_arrays(LLMatObject *self, PyObject *args) {
  npy_intp d;
  PyObject *ind, *col, *val;
  int nnz;
  if (!PyArg_ParseTuple(args, "")) return NULL;
  nnz = ll_csr_nnz(self);
  d = self->dim[0] + 1;
  ind = PyArray_SimpleNew(1, &d, NPY_INT32);
  d = nnz;
  col = PyArray_SimpleNew(1, &d, NPY_INT32);
  val = PyArray_SimpleNew(1, &d, NPY_DOUBLE);
  if (!ind || !col || !val ||
      ll_fill_csr(self, (double *)PyArray_DATA((PyArrayObject *)val),
                  (int *)PyArray_DATA((PyArrayObject *)col),
                  (int *)PyArray_DATA((PyArrayObject *)ind)) < 0) {
    Py_XDECREF(ind);
    Py_XDECREF(col);
    Py_XDECREF(val);
    return NULL;
  }
  return Py_BuildValue("(NNN)", ind, col, val);
}

static PyObject *LLMat_to_sss_arrays(LLMatObject *self, PyObject *args) {
  npy_intp d;
  PyObject *ind, *col, *val, *diag;
  int i, j, k, r = 0, n = self->dim[0], nnz = 0;
  int *pi, *pc;
  double *pv, *pd;
  if (!PyArg_ParseTuple(args, "")) return NULL;
  if (n != self->dim[1]) {
    PyErr_SetString(PyExc_ValueError, "Matrix must be square");
    return NULL;
  }
  for (i = 0; i < n; i++)
    for (k = self->root[i]; k != -1; k = self->link[k])
      if (i > self->col[k]) nnz++;
  d = n + 1;
  ind = PyArray_SimpleNew(1, &d, NPY_INT32);
  d = nnz;
  col = PyArray_SimpleNew(1, &d, NPY_INT32);
  val = PyArray_SimpleNew(1, &d, NPY_DOUBLE);
  d = n;
  diag = PyArray_SimpleNew(1, &d, NPY_DOUBLE);
  if (!ind || !col || !val || !diag) {
    Py_XDECREF(ind);
    Py_XDECREF(col);
    Py_XDECREF(val);
    Py_XDECREF(diag);
    return NULL;
  }
  pi = (int *)PyArray_DATA((PyArrayObject *)ind);
  pc = (int *)PyArray_DATA((PyArrayObject *)col);
  pv = (double *)PyArray_DATA((PyArrayObject *)val);
  pd = (double *)PyArray_DATA((PyArrayObject *)diag);
  pi[0] = 0;
  for (i = 0; i < n; i++) {
    pd[i] = 0.0;
    for (k = self->root[i]; k != -1; k = self->link[k]) {
      j = self->col[k];
      if (i > j) {
        pv[r] = self->val[k];
        pc[r] = j;
        r++;
      } else if (i == j)
        pd[i] = self->val[k];
    }
    pi[i + 1] = r;
  }
  return Py_BuildValue("(NNNN)", ind, col, val, diag);
}

/* device mirror of the linked list (full, column-sorted CSR): built on first use after a
 * modification.  The per-row summation order of the mirror equals that of the reference's
 * ll_matvec kernels (general: ll_mat.c:1262-1277; symmetric: :1300-1320). */
static int ll_ensure_mirror(LLMatObject *self) {
  int nnz, rc;
  double *val;
  int *col, *ind;
  if (self->mirror) return 0;
  nnz = ll_csr_nnz(self);
  val = PyMem_New(double, nnz > 0 ? nnz : 1);
  col = PyMem_New(int, nnz > 0 ? nnz : 1);
  ind = PyMem_New(int, self->dim[0] + 1);
  if (!val || !col || !ind) {
    PyMem_Del(val);
    PyMem_Del(col);
    PyMem_Del(ind);
    PyErr_NoMemory();
    return -1;
  }
  if (ll_fill_csr(self, val, col, ind) < 0) {
    rc = PSP_ENOMEM;
  } else {
    rc = psp_csr_create(self->dim[0], self->dim[1], nnz, ind, col, val, &self->mirror);
  }
  PyMem_Del(val);
  PyMem_Del(col);
  PyMem_Del(ind);
  if (rc != PSP_OK) {
    if (!PyErr_Occurred()) psp_raise(rc);
    return -1;
  }
  return 0;
}

static PyObject *LLMat_matvec(LLMatObject *self, PyObject *args) {
  PyArrayObject *xp, *yp;
  int rc;
  if (parse_arr_arr_stride(args, &xp, &yp, self->dim[1], self->dim[0])) return NULL;
  if (ll_ensure_mirror(self)) return NULL;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_csr_matvec_stride(self->mirror, (double *)PyArray_DATA(xp), ELEM_STRIDE(xp),
                             (double *)PyArray_DATA(yp), ELEM_STRIDE(yp));
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) return psp_raise(rc);
  Py_RETURN_NONE;
}

static PyObject *LLMat_matvec_transp(LLMatObject *self, PyObject *args) {
  PyArrayObject *xp, *yp;
  int rc;
  if (self->issym) return LLMat_matvec(self, args); /* ll_mat.c:1474-1477 */
  if (parse_arr_arr_stride(args, &xp, &yp, self->dim[0], self->dim[1])) return NULL;
  if (ll_ensure_mirror(self)) return NULL;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_csr_matvec_transp_stride(self->mirror, (double *)PyArray_DATA(xp), ELEM_STRIDE(xp),
                                    (double *)PyArray_DATA(yp), ELEM_STRIDE(yp));
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) return psp_raise(rc);
  Py_RETURN_NONE;
}

/* a.put(b[, id1[, id2]]): a[id1[i], id2[i]] = b[i]  (ll_mat.c:2497-2752).  b scalar or
 * sequence; id1 defaults to 0..len-1, id2 defaults to id1; symmetric matrices store the
 * entry in the lower triangle (:2709-2716). */
static PyObject *LLMat_put(LLMatObject *self, PyObject *args) {
  PyObject *bIn, *id1in = NULL, *id2in = NULL;
  PyArrayObject *b = NULL, *id1 = NULL, *id2 = NULL;
  npy_intp len = -1, i;
  double bval = 0.0;
  int b_is_scalar = 0;
  PyObject *ret = NULL;

  if (!PyArg_ParseTuple(args, "O|OO", &bIn, &id1in, &id2in)) return NULL;
  if (id1in == Py_None) id1in = NULL;
  if (id2in == Py_None) id2in = NULL;
  if (PyLong_Check(bIn) || PyFloat_Check(bIn)) {
    bval = PyFloat_AsDouble(bIn);
    if (PyErr_Occurred()) return NULL;
    b_is_scalar = 1;
  } else {
    b = (PyArrayObject *)PyArray_FROM_OTF(bIn, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY);
    if (b == NULL) goto done;
    if (PyArray_NDIM(b) != 1) {
      PyErr_SetString(PyExc_ValueError, "b must be a scalar or a 1-dimensional sequence");
      goto done;
    }
    len = PyArray_DIM(b, 0);
  }
  if (id1in) {
    id1 = (PyArrayObject *)PyArray_FROM_OTF(id1in, NPY_INTP, NPY_ARRAY_IN_ARRAY);
    if (id1 == NULL) goto done;
    if (PyArray_NDIM(id1) != 1 || (len >= 0 && PyArray_DIM(id1, 0) != len)) {
      PyErr_SetString(PyExc_IndexError, "Not as many row indices as values");
      goto done;
    }
    len = PyArray_DIM(id1, 0);
  }
  if (id2in) {
    id2 = (PyArrayObject *)PyArray_FROM_OTF(id2in, NPY_INTP, NPY_ARRAY_IN_ARRAY);
    if (id2 == NULL) goto done;
    if (PyArray_NDIM(id2) != 1 || (len >= 0 && PyArray_DIM(id2, 0) != len)) {
      PyErr_SetString(PyExc_IndexError, "Not as many column indices as values");
      goto done;
    }
    len = PyArray_DIM(id2, 0);
  }
  if (len < 0) len = 1; /* scalar without index lists: a[0,0] = b */
  for (i = 0; i < len; i++) {
    npy_intp i1 = id1 ? ((npy_intp *)PyArray_DATA(id1))[i] : i;
    npy_intp j1 = id2 ? ((npy_intp *)PyArray_DATA(id2))[i] : i1;
    double v = b_is_scalar ? bval : ((double *)PyArray_DATA(b))[i];
    int rc;
    if (i1 > j1 || !self->issym)
      rc = SpMatrix_LLMatSetItem(self, (int)i1, (int)j1, v);
    else
      rc = SpMatrix_LLMatSetItem(self, (int)j1, (int)i1, v);
    if (rc == -1) goto done;
  }
  ret = Py_None;
  Py_INCREF(ret);
done:
  Py_XDECREF(b);
  Py_XDECREF(id1);
  Py_XDECREF(id2);
  return ret;
}

static PyObject *LLMat_update_add_at(LLMatObject *self, PyObject *args) {
  PyObject *bIn, *id1in, *id2in;
  PyArrayObject *b, *id1 = NULL, *id2 = NULL;
  npy_intp len, i;
  PyObject *ret = NULL;
  if (!PyArg_ParseTuple(args, "OOO", &bIn, &id1in, &id2in)) return NULL;
  b = (PyArrayObject *)PyArray_FROM_OTF(bIn, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY);
  if (b == NULL) return NULL;
  id1 = (PyArrayObject *)PyArray_FROM_OTF(id1in, NPY_INTP, NPY_ARRAY_IN_ARRAY);
  id2 = id1 ? (PyArrayObject *)PyArray_FROM_OTF(id2in, NPY_INTP, NPY_ARRAY_IN_ARRAY) : NULL;
  if (id1 == NULL || id2 == NULL) goto done;
  len = PyArray_SIZE(b);
  if (PyArray_SIZE(id1) != len || PyArray_SIZE(id2) != len) {
    PyErr_SetString(PyExc_ValueError, "id1 and id2 must have the same length as b");
    goto done;
  }
  for (i = 0; i < len; i++) {
    npy_intp i1 = ((npy_intp *)PyArray_DATA(id1))[i], j1 = ((npy_intp *)PyArray_DATA(id2))[i];
    if (self->issym && i1 < j1) {
      npy_intp t = i1;
      i1 = j1;
      j1 = t;
    }
    if (SpMatrix_LLMatUpdateItemAdd(self, (int)i1, (int)j1, ((double *)PyArray_DATA(b))[i]) == -1)
      goto done;
  }
  ret = Py_None;
  Py_INCREF(ret);
done:
  Py_DECREF(b);
  Py_XDECREF(id1);
  Py_XDECREF(id2);
  return ret;
}

/* an (int, int) key with negative indices counting from the end (sss_mat[i, j]) */
static int parse_int_index(PyObject *key, int dim0, int dim1, int *i, int *j) {
  long a, b;
  if (!PyTuple_Check(key) || PyTuple_GET_SIZE(key) != 2 ||
      !PyIndex_Check(PyTuple_GET_ITEM(key, 0)) || !PyIndex_Check(PyTuple_GET_ITEM(key, 1))) {
    PyErr_SetString(PyExc_IndexError, "only integer index pairs [i,j] are supported");
    return -1;
  }
  a = PyLong_AsLong(PyTuple_GET_ITEM(key, 0));
  b = PyLong_AsLong(PyTuple_GET_ITEM(key, 1));
  if (PyErr_Occurred()) return -1;
  if (a < 0) a += dim0;
  if (b < 0) b += dim1;
  if (a < 0 || a >= dim0 || b < 0 || b >= dim1) {
    PyErr_SetString(PyExc_IndexError, "indices out of range");
    return -1;
  }
  *i = (int)a;
  *j = (int)b;
  return 0;
}

/* sub-matrix read / write, copy, shift, scale, norms, deletion, compress, exports, assembly updates, matrix products */
#include "ll_mat_edit.c"

static void LLMat_dealloc(LLMatObject *a) {
  ll_invalidate(a);
  PyMem_Del(a->root);
  PyMem_Del(a->val);
  PyMem_Del(a->col);
  PyMem_Del(a->link);
  PyObject_Del(a);
}

static PyObject *LLMat_get_shape(LLMatObject *a, void *c) {
  return Py_BuildValue("(i,i)", a->dim[0], a->dim[1]);
}
static PyObject *LLMat_get_nnz(LLMatObject *a, void *c) { return PyLong_FromLong(a->nnz); }
static PyObject *LLMat_get_issym(LLMatObject *a, void *c) { return PyLong_FromLong(a->issym); }
static PyObject *LLMat_get_storezeros(LLMatObject *a, void *c) { return PyLong_FromLong(a->storeZeros); }

static PyObject *LLMat_get_psp_op(LLMatObject *a, void *c) {
  if (a->dim[0] != a->dim[1]) {
    PyErr_SetString(PyExc_ValueError, "matrix is not square");
    return NULL;
  }
  if (ll_ensure_mirror(a)) return NULL;
  if (a->op == NULL) {
    int rc = psp_op_from_csr(a->mirror, &a->op);
    if (rc != PSP_OK) return psp_raise(rc);
  }
  return PyCapsule_New(a->op, PSP_OP_CAPSULE_NAME, NULL);
}

static PyObject *LLMat_repr(LLMatObject *a) {
  return PyUnicode_FromFormat("<ll_mat%s object, shape (%d,%d), %d stored entries>",
                              a->issym ? "_sym" : "", a->dim[0], a->dim[1], a->nnz);
}

static PyMethodDef LLMat_methods[] = {
    {"matvec", (PyCFunction)LLMat_matvec, METH_VARARGS, "a.matvec(x, y): y := a * x (on the GPU)"},
    {"matvec_transp", (PyCFunction)LLMat_matvec_transp, METH_VARARGS, "a.matvec_transp(x, y): y := a^T * x"},
    {"to_csr", (PyCFunction)(void (*)(void))LLMat_to_csr, METH_VARARGS | METH_KEYWORDS,
     "A.to_csr(): new csr_mat from the data of A; A.to_csr(devices=[0, 1, ...]): its rows on several GPUs"},
    {"to_sss", (PyCFunction)LLMat_to_sss, METH_VARARGS, "a.to_sss(): new sss_mat from the lower triangle of a"},
    {"to_csr_arrays", (PyCFunction)LLMat_to_csr_arrays, METH_VARARGS, "(indptr, indices, data) of to_csr(), on the host"},
    {"to_sss_arrays", (PyCFunction)LLMat_to_sss_arrays, METH_VARARGS, "(indptr, indices, data, diag) of to_sss(), on the host"},
    {"put", (PyCFunction)LLMat_put, METH_VARARGS, "a.put(b[, id1[, id2]]): a[id1[i], id2[i]] = b[i]"},
    {"update_add_at", (PyCFunction)LLMat_update_add_at, METH_VARARGS, "a.update_add_at(b, id1, id2): a[id1[i], id2[i]] += b[i]"},
    {"generalize", (PyCFunction)LLMat_generalize, METH_VARARGS, "convert from symmetric to non-symmetric form (in place)"},
    {"compress", (PyCFunction)LLMat_compress, METH_VARARGS, "A.compress(): reclaim unused space; returns the number of elements freed"},
    {"export_mtx", (PyCFunction)LLMat_export_mtx, METH_VARARGS, "A.export_mtx(fileName, precision=16): write A in MatrixMarket format"},
    {"copy", (PyCFunction)LLMat_copy, METH_VARARGS, "A.copy(): a (deep) copy of A"},
    {"norm", (PyCFunction)LLMat_norm, METH_VARARGS, "A.norm(p): p = '1', 'inf' (general storage) or 'fro'"},
    {"shift", (PyCFunction)LLMat_shift, METH_VARARGS, "A.shift(sigma, B): A = A + sigma * B"},
    {"scale", (PyCFunction)LLMat_scale, METH_VARARGS, "A.scale(sigma): every element times sigma"},
    {"col_scale", (PyCFunction)LLMat_col_scale, METH_VARARGS, "A.col_scale(v): column i times v[i]"},
    {"row_scale", (PyCFunction)LLMat_row_scale, METH_VARARGS, "A.row_scale(v): row i times v[i]"},
    {"keys", (PyCFunction)LLMat_keys, METH_VARARGS, "A.keys(): list of the (i, j) of the stored entries"},
    {"values", (PyCFunction)LLMat_values, METH_VARARGS, "A.values(): list of the stored values"},
    {"items", (PyCFunction)LLMat_items, METH_VARARGS, "A.items(): list of ((i, j), value)"},
    {"take", (PyCFunction)LLMat_take, METH_VARARGS, "A.take(b[, id1[, id2]]): b[i] = A[id1[i], id2[i]]"},
    {"find", (PyCFunction)LLMat_find, METH_VARARGS, "A.find(): (val, irow, jcol) of the stored entries"},
    {"update_add_mask", (PyCFunction)LLMat_update_add_mask, METH_VARARGS,
     "A.update_add_mask(b, ind0, ind1, mask0, mask1): a[ind0[i], ind1[j]] += b[i, j] where both masks are set"},
    {"update_add_mask_sym", (PyCFunction)LLMat_update_add_mask_sym, METH_VARARGS,
     "A.update_add_mask_sym(b, ind, mask): the symmetric assembly update (pairs j <= i)"},
    {"delete_rows", (PyCFunction)LLMat_delete_rows, METH_VARARGS, "A.delete_rows(mask): keep the rows whose mask is non-zero"},
    {"delete_cols", (PyCFunction)LLMat_delete_cols, METH_VARARGS, "A.delete_cols(mask): keep the columns whose mask is non-zero"},
    {"delete_rowcols", (PyCFunction)LLMat_delete_rowcols, METH_VARARGS, "A.delete_rowcols(mask): both (square matrices)"},
    {NULL, NULL, 0, NULL}};

static PyGetSetDef LLMat_getset[] = {{"shape", (getter)LLMat_get_shape, NULL, "(rows, cols)", NULL},
                                     {"nnz", (getter)LLMat_get_nnz, NULL, "stored entries", NULL},
                                     {"issym", (getter)LLMat_get_issym, NULL, "symmetric storage", NULL},
                                     {"storeZeros", (getter)LLMat_get_storezeros, NULL, "explicit zeros are kept", NULL},
                                     {"_psp_op", (getter)LLMat_get_psp_op, NULL, "device operator", NULL},
                                     {NULL, NULL, NULL, NULL, NULL}};

static PyMappingMethods LLMat_as_mapping = {(lenfunc)LLMat_length, (binaryfunc)LLMat_subscript,
                                            (objobjargproc)LLMat_ass_subscript};

/* ------------------------------------------------------------------ csr_mat */

static PyObject *CSRMat_matvec(CSRMatObject *self, PyObject *args) {
  PyArrayObject *xp, *yp;
  int rc;
  if (parse_arr_arr_stride(args, &xp, &yp, self->dim[1], self->dim[0])) return NULL;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_csr_matvec_stride(self->dev, (double *)PyArray_DATA(xp), ELEM_STRIDE(xp),
                             (double *)PyArray_DATA(yp), ELEM_STRIDE(yp));
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) return psp_raise(rc);
  Py_RETURN_NONE;
}

static PyObject *CSRMat_matvec_transp(CSRMatObject *self, PyObject *args) {
  PyArrayObject *xp, *yp;
  int rc;
  if (parse_arr_arr_stride(args, &xp, &yp, self->dim[0], self->dim[1])) return NULL;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_csr_matvec_transp_stride(self->dev, (double *)PyArray_DATA(xp), ELEM_STRIDE(xp),
                                    (double *)PyArray_DATA(yp), ELEM_STRIDE(yp));
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) return psp_raise(rc);
  Py_RETURN_NONE;
}

static PyObject *CSRMat_to_arrays(CSRMatObject *self, PyObject *args) {
  npy_intp d;
  PyObject *ind, *col, *val;
  int rc;
  if (!PyArg_ParseTuple(args, "")) return NULL;
  d = self->dim[0] + 1;
  ind = PyArray_SimpleNew(1, &d, NPY_INT32);
  d = self->nnz;
  col = PyArray_SimpleNew(1, &d, NPY_INT32);
  val = PyArray_SimpleNew(1, &d, NPY_DOUBLE);
  if (!ind || !col || !val) goto fail;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_csr_download(self->dev, (int *)PyArray_DATA((PyArrayObject *)ind),
                        (int *)PyArray_DATA((PyArrayObject *)col),
                        (double *)PyArray_DATA((PyArrayObject *)val));
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) {
    psp_raise(rc);
    goto fail;
  }
  return Py_BuildValue("(NNN)", ind, col, val);
fail:
  Py_XDECREF(ind);
  Py_XDECREF(col);
  Py_XDECREF(val);
  return NULL;
}

static void CSRMat_dealloc(CSRMatObject *a) {
  if (a->op) psp_op_destroy(a->op);
  if (a->dev) psp_csr_destroy(a->dev);
  PyMem_Del(a->ind);
  PyMem_Del(a->val);
  PyMem_Del(a->col);
  PyObject_Del(a);
}

static PyObject *CSRMat_get_shape(CSRMatObject *a, void *c) {
  return Py_BuildValue("(i,i)", a->dim[0], a->dim[1]);
}
static PyObject *CSRMat_get_nnz(CSRMatObject *a, void *c) {
  if (a->nnz < 0 && a->dev != NULL) return PyLong_FromLongLong((long long)psp_csr_nnz64(a->dev)); /* beyond the struct's int */
  return PyLong_FromLong(a->nnz);
}

static PyObject *CSRMat_get_psp_op(CSRMatObject *a, void *c) {
  if (a->op == NULL) {
    int rc = psp_op_from_csr(a->dev, &a->op);
    if (rc != PSP_OK) return psp_raise(rc);
  }
  return PyCapsule_New(a->op, PSP_OP_CAPSULE_NAME, NULL);
}

/* where the products of this process run: the GPU, or -- PSP_DEVICE=cpu, the opt-in host mode -- the host loops */
static const char *psp_where(void) {
  return strstr(psp_version(), "PSP_DEVICE=cpu") != NULL ? "in host memory (PSP_DEVICE=cpu)" : "on the GPU";
}

static PyObject *CSRMat_repr(CSRMatObject *a) {
  return PyUnicode_FromFormat("<csr_mat object %s, shape (%d,%d), nnz %d>", psp_where(), a->dim[0],
                              a->dim[1], a->nnz);
}

/* print(A): the text of the reference's tp_print (csr_mat.c:186-207) -- "csr_mat([m,n], [(i,j): v, ...])" -- for matrices
 * a person would print; beyond 10 000 stored entries the one-line repr (the arrays live on the device) */
static PyObject *csr_like_str(const char *name, int rows, int cols, const int *ind, const int *col, const double *val,
                              const double *diag, long nnz) {
  LLText t = {NULL, 0, 0};
  PyObject *ret;
  int i, k, bad = 0, first = 1;
  if (nnz == 0) {
    bad |= ll_text_add(&t, "%s([%d,%d])", name, rows, cols);
  } else {
    bad |= ll_text_add(&t, "%s([%d,%d], [", name, rows, cols);
    for (i = 0; i < rows && !bad; i++) {
      for (k = ind[i]; k < ind[i + 1] && !bad; k++) {
        bad |= ll_text_add(&t, "%s(%d,%d): %g", first ? "" : ", ", i, col[k], val[k]);
        first = 0;
      }
      if (diag) { /* sss_mat.c:136-144: the row's lower entries, then its diagonal entry */
        bad |= ll_text_add(&t, "%s(%d,%d): %g", first ? "" : ", ", i, i, diag[i]);
        first = 0;
      }
    }
    bad |= ll_text_add(&t, "])");
  }
  if (bad) {
    free(t.p);
    return PyErr_NoMemory();
  }
  ret = PyUnicode_FromStringAndSize(t.p, (Py_ssize_t)t.len);
  free(t.p);
  return ret;
}

static PyObject *CSRMat_repr(CSRMatObject *a);
static PyObject *CSRMat_str(CSRMatObject *a) {
  PyObject *empty, *arrs, *ret;
  if (a->nnz > 10000) return CSRMat_repr(a);
  if ((empty = PyTuple_New(0)) == NULL) return NULL;
  arrs = CSRMat_to_arrays(a, empty);
  Py_DECREF(empty);
  if (arrs == NULL) return NULL;
  ret = csr_like_str("csr_mat", a->dim[0], a->dim[1], (const int *)PyArray_DATA((PyArrayObject *)PyTuple_GET_ITEM(arrs, 0)),
                     (const int *)PyArray_DATA((PyArrayObject *)PyTuple_GET_ITEM(arrs, 1)),
                     (const double *)PyArray_DATA((PyArrayObject *)PyTuple_GET_ITEM(arrs, 2)), NULL, a->nnz);
  Py_DECREF(arrs);
  return ret;
}

static PyMethodDef CSRMat_methods[] = {
    {"matvec", (PyCFunction)CSRMat_matvec, METH_VARARGS, "a.matvec(x, y): y := a * x"},
    {"matvec_transp", (PyCFunction)CSRMat_matvec_transp, METH_VARARGS, "a.matvec_transp(x, y): y := a^T * x"},
    {"to_arrays", (PyCFunction)CSRMat_to_arrays, METH_VARARGS, "(indptr, indices, data) copied back from the device"},
    {NULL, NULL, 0, NULL}};

static PyGetSetDef CSRMat_getset[] = {{"shape", (getter)CSRMat_get_shape, NULL, "(rows, cols)", NULL},
                                      {"nnz", (getter)CSRMat_get_nnz, NULL, "stored entries", NULL},
                                      {"_psp_op", (getter)CSRMat_get_psp_op, NULL, "device operator", NULL},
                                      {NULL, NULL, NULL, NULL, NULL}};

/* ------------------------------------------------------------------ sss_mat */

static PyObject *SSSMat_matvec(SSSMatObject *self, PyObject *args) {
  PyArrayObject *xp, *yp;
  int rc;
  if (parse_arr_arr_stride(args, &xp, &yp, self->n, self->n)) return NULL;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_sss_matvec_stride(self->dev, (double *)PyArray_DATA(xp), ELEM_STRIDE(xp),
                             (double *)PyArray_DATA(yp), ELEM_STRIDE(yp));
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) return psp_raise(rc);
  Py_RETURN_NONE;
}

/* A[i,j]: the evident intent of SSSMat_subscript -> getitem (sss_mat.c:14-28,189-205); in the
 * reference snapshot LLMat_parse_index is a stub that always raises (ll_mat.c:3798-3806) */
static PyObject *SSSMat_subscript(SSSMatObject *self, PyObject *key) {
  int i, j, rc;
  double v;
  if (!PyTuple_Check(key) || PyTuple_GET_SIZE(key) != 2 ||
      !PyIndex_Check(PyTuple_GET_ITEM(key, 0)) || !PyIndex_Check(PyTuple_GET_ITEM(key, 1))) {
    PyErr_SetString(PyExc_IndexError, "slices not supported");
    return NULL;
  }
  if (parse_int_index(key, self->n, self->n, &i, &j)) return NULL;
  if (self->diag) { /* host copy present */
    int k, t;
    if (i == j) return PyFloat_FromDouble(self->diag[i]);
    if (i < j) {
      t = i;
      i = j;
      j = t;
    }
    for (k = self->ind[i]; k < self->ind[i + 1]; k++)
      if (self->col[k] == j) return PyFloat_FromDouble(self->val[k]);
    return PyFloat_FromDouble(0.0);
  }
  rc = psp_sss_getitem(self->dev, i, j, &v);
  if (rc != PSP_OK) return psp_raise(rc);
  return PyFloat_FromDouble(v);
}

static PyObject *SSSMat_to_arrays(SSSMatObject *self, PyObject *args) {
  npy_intp d;
  PyObject *ind, *col, *val, *diag;
  int rc;
  if (!PyArg_ParseTuple(args, "")) return NULL;
  d = self->n + 1;
  ind = PyArray_SimpleNew(1, &d, NPY_INT32);
  d = self->nnz;
  col = PyArray_SimpleNew(1, &d, NPY_INT32);
  val = PyArray_SimpleNew(1, &d, NPY_DOUBLE);
  d = self->n;
  diag = PyArray_SimpleNew(1, &d, NPY_DOUBLE);
  if (!ind || !col || !val || !diag) goto fail;
  rc = psp_sss_download(self->dev, (int *)PyArray_DATA((PyArrayObject *)ind),
                        (int *)PyArray_DATA((PyArrayObject *)col),
                        (double *)PyArray_DATA((PyArrayObject *)val),
                        (double *)PyArray_DATA((PyArrayObject *)diag));
  if (rc != PSP_OK) {
    psp_raise(rc);
    goto fail;
  }
  return Py_BuildValue("(NNNN)", ind, col, val, diag);
fail:
  Py_XDECREF(ind);
  Py_XDECREF(col);
  Py_XDECREF(val);
  Py_XDECREF(diag);
  return NULL;
}

static void SSSMat_dealloc(SSSMatObject *a) {
  if (a->op) psp_op_destroy(a->op);
  if (a->dev) psp_sss_destroy(a->dev);
  PyMem_Del(a->ind);
  PyMem_Del(a->val);
  PyMem_Del(a->col);
  PyMem_Del(a->diag);
  PyObject_Del(a);
}

static PyObject *SSSMat_get_shape(SSSMatObject *a, void *c) {
  return Py_BuildValue("(i,i)", a->n, a->n);
}
/* sss_mat.c:155: the attribute reports strict-lower count + n */
static PyObject *SSSMat_get_nnz(SSSMatObject *a, void *c) { return PyLong_FromLong(a->nnz + a->n); }

static PyObject *SSSMat_get_psp_op(SSSMatObject *a, void *c) {
  if (a->op == NULL) {
    int rc = psp_op_from_sss(a->dev, &a->op);
    if (rc != PSP_OK) return psp_raise(rc);
  }
  return PyCapsule_New(a->op, PSP_OP_CAPSULE_NAME, NULL);
}

static PyObject *SSSMat_repr(SSSMatObject *a) {
  return PyUnicode_FromFormat("<sss_mat object %s, order %d, %d stored lower entries>", psp_where(),
                              a->n, a->nnz);
}

static PyObject *SSSMat_repr(SSSMatObject *a);
static PyObject *SSSMat_str(SSSMatObject *a) {
  PyObject *empty, *arrs, *ret;
  if (a->nnz + a->n > 10000) return SSSMat_repr(a);
  if ((empty = PyTuple_New(0)) == NULL) return NULL;
  arrs = SSSMat_to_arrays(a, empty);
  Py_DECREF(empty);
  if (arrs == NULL) return NULL;
  ret = csr_like_str("sss_mat", a->n, a->n, (const int *)PyArray_DATA((PyArrayObject *)PyTuple_GET_ITEM(arrs, 0)),
                     (const int *)PyArray_DATA((PyArrayObject *)PyTuple_GET_ITEM(arrs, 1)),
                     (const double *)PyArray_DATA((PyArrayObject *)PyTuple_GET_ITEM(arrs, 2)),
                     (const double *)PyArray_DATA((PyArrayObject *)PyTuple_GET_ITEM(arrs, 3)), (long)a->nnz + a->n);
  Py_DECREF(arrs);
  return ret;
}

static PyMethodDef SSSMat_methods[] = {
    {"matvec", (PyCFunction)SSSMat_matvec, METH_VARARGS, "a.matvec(x, y): y := a * x"},
    {"matvec_transp", (PyCFunction)SSSMat_matvec, METH_VARARGS, "a.matvec_transp(x, y): y := a^T * x (== a * x)"},
    {"to_arrays", (PyCFunction)SSSMat_to_arrays, METH_VARARGS, "(indptr, indices, data, diag) copied back from the device"},
    {NULL, NULL, 0, NULL}};

static PyGetSetDef SSSMat_getset[] = {{"shape", (getter)SSSMat_get_shape, NULL, "(n, n)", NULL},
                                      {"nnz", (getter)SSSMat_get_nnz, NULL, "stored entries + n", NULL},
                                      {"_psp_op", (getter)SSSMat_get_psp_op, NULL, "device operator", NULL},
                                      {NULL, NULL, NULL, NULL, NULL}};

static PyMappingMethods SSSMat_as_mapping = {NULL, (binaryfunc)SSSMat_subscript, NULL};

/* ------------------------------------------------------------------ module functions */

/* ll_mat(n, m, sizeHint=1000, storeZeros=0): spmatrixmodule.c:317-328 */
static PyObject *LLMat_zeros(PyObject *self, PyObject *args) {
  int dim[2], sizeHint = 1000, storeZeros = 0;
  if (!PyArg_ParseTuple(args, "ii|ii", dim, dim + 1, &sizeHint, &storeZeros)) return NULL;
  return SpMatrix_NewLLMatObject(dim, 0, sizeHint, storeZeros);
}

/* ll_mat_sym(n, sizeHint=1000, storeZeros=0): spmatrixmodule.c:330-342 */
static PyObject *LLMat_sym_zeros(PyObject *self, PyObject *args) {
  int dim[2], n, sizeHint = 1000, storeZeros = 0;
  if (!PyArg_ParseTuple(args, "i|ii", &n, &sizeHint, &storeZeros)) return NULL;
  dim[0] = dim[1] = n;
  return SpMatrix_NewLLMatObject(dim, 1, sizeHint, storeZeros);
}

/* ll_mat_from_mtx(fileName): coordinate real [general|symmetric] MatrixMarket files
 * (LLMat_from_mtx, ll_mat.c:3390-3456; banner rules of mmio: "%%MatrixMarket matrix
 * coordinate real|integer general|symmetric") */
static PyObject *LLMat_from_mtx(PyObject *module, PyObject *args) {
  const char *fileName;
  char line[1100], banner[64], mtx[64], crd[64], dtype[64], sym[64];
  int dim[2], nz, i, row, col, is_sym;
  double val;
  LLMatObject *self = NULL;
  FILE *f;
  char *p;
  if (!PyArg_ParseTuple(args, "s", &fileName)) return NULL;
  f = fopen(fileName, "r");
  if (f == NULL) return PyErr_SetFromErrnoWithFilename(PyExc_IOError, fileName);
  if (fgets(line, sizeof line, f) == NULL ||
      sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, dtype, sym) != 5 ||
      strcmp(banner, "%%MatrixMarket") != 0) {
    PyErr_SetString(PyExc_IOError, "error reading MTX file header");
    goto fail;
  }
  for (p = mtx; *p; p++) *p = (char)tolower(*p);
  for (p = crd; *p; p++) *p = (char)tolower(*p);
  for (p = dtype; *p; p++) *p = (char)tolower(*p);
  for (p = sym; *p; p++) *p = (char)tolower(*p);
  if (strcmp(mtx, "matrix") != 0 || strcmp(crd, "coordinate") != 0 ||
      (strcmp(dtype, "real") != 0 && strcmp(dtype, "integer") != 0)) {
    PyErr_SetString(SpMatrix_ErrorObject, "must be real, sparse matrix");
    goto fail;
  }
  is_sym = strcmp(sym, "symmetric") == 0;
  if (!is_sym && strcmp(sym, "general") != 0) {
    PyErr_SetString(SpMatrix_ErrorObject, "must be real, sparse matrix");
    goto fail;
  }
  do { /* skip comments */
    if (fgets(line, sizeof line, f) == NULL) {
      PyErr_SetString(PyExc_IOError, "error reading MTX file size information");
      goto fail;
    }
  } while (line[0] == '%');
  if (sscanf(line, "%d %d %d", dim, dim + 1, &nz) != 3) {
    PyErr_SetString(PyExc_IOError, "error reading MTX file size information");
    goto fail;
  }
  self = (LLMatObject *)SpMatrix_NewLLMatObject(dim, is_sym, nz, 0);
  if (self == NULL) goto fail;
  for (i = 0; i < nz; i++) {
    if (fscanf(f, "%d %d %lg\n", &row, &col, &val) != 3) {
      PyErr_SetString(PyExc_IOError, "error reading MTX file data");
      goto fail;
    }
    row--;
    col--;
    if (!(0 <= row && row < dim[0] && 0 <= col && col < dim[1])) {
      PyErr_SetString(PyExc_IndexError, "matrix indices out of range");
      goto fail;
    }
    if (SpMatrix_LLMatSetItem(self, row, col, val)) goto fail;
  }
  fclose(f);
  return (PyObject *)self;
fail:
  fclose(f);
  Py_XDECREF(self);
  return NULL;
}

/* mtx_read_coordinate(fileName, threads=0) -> (m, n, symmetric, rows, cols, vals): the entries of a coordinate
 * MatrixMarket file in file order, 0-based int64 indices and float64 values, parsed by `threads` threads (0: up to 8).
 * Same banner rules as ll_mat_from_mtx (LLMat_from_mtx, ll_mat.c:3390-3456).  What pysparse_amd.tools.mtx builds
 * csr_mat / sss_mat from without one sorted list insertion per entry: SuiteSparse
 * Emilia_923 has 2.1e7 entries in its file; the values go through strtod, i.e. correctly rounded like the
 * reference's fscanf("%lg"). */
typedef struct {
  const char *p, *e;
  long cnt, off, m, n;
  npy_int64 *ri, *ci;
  double *v;
  int err; /* 1: malformed line, 2: index out of range */
} MtxChunk;

static const char *mtx_skip_blank(const char *p, const char *e) {
  while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
  return p;
}

static void *mtx_count_worker(void *arg) {
  MtxChunk *c = (MtxChunk *)arg;
  const char *p = c->p;
  long cnt = 0;
  while (p < c->e) {
    const char *q = mtx_skip_blank(p, c->e);
    const char *nl = (const char *)memchr(q, '\n', (size_t)(c->e - q));
    if (q < c->e && *q != '\n' && *q != '%') cnt++;
    p = nl ? nl + 1 : c->e;
  }
  c->cnt = cnt;
  return NULL;
}

static void *mtx_parse_worker(void *arg) {
  MtxChunk *c = (MtxChunk *)arg;
  const char *p = c->p;
  long k = c->off;
  while (p < c->e) {
    const char *q = mtx_skip_blank(p, c->e);
    const char *nl = (const char *)memchr(q, '\n', (size_t)(c->e - q));
    if (q < c->e && *q != '\n' && *q != '%') {
      char *end;
      long r, cc;
      double val;
      r = strtol(q, &end, 10);
      if (end == q) { c->err = 1; return NULL; }
      q = end;
      cc = strtol(q, &end, 10);
      if (end == q) { c->err = 1; return NULL; }
      q = end;
      val = strtod(q, &end); /* the buffer ends with a NUL behind the last line */
      if (end == q) { c->err = 1; return NULL; }
      if (r < 1 || r > c->m || cc < 1 || cc > c->n) { c->err = 2; return NULL; }
      c->ri[k] = r - 1;
      c->ci[k] = cc - 1;
      c->v[k] = val;
      k++;
    }
    p = nl ? nl + 1 : c->e;
  }
  return NULL;
}

static PyObject *Mtx_read_coordinate(PyObject *module, PyObject *args) {
  const char *fileName;
  int threads = 0, T, t, is_sym, err = 0;
  char banner[64], mtx[64], crd[64], dtype[64], sym[64];
  char *buf = NULL, *p, *data;
  long size, got, m, n, nz, total = 0;
  FILE *f;
  MtxChunk ch[64];
  pthread_t th[64];
  PyArrayObject *ri = NULL, *ci = NULL, *va = NULL;
  npy_intp dims[1];
  PyObject *res;
  if (!PyArg_ParseTuple(args, "s|i", &fileName, &threads)) return NULL;
  f = fopen(fileName, "rb");
  if (f == NULL) return PyErr_SetFromErrnoWithFilename(PyExc_IOError, fileName);
  if (fseek(f, 0, SEEK_END) != 0 || (size = ftell(f)) < 0 || fseek(f, 0, SEEK_SET) != 0) {
    fclose(f);
    PyErr_SetString(PyExc_IOError, "error reading MTX file");
    return NULL;
  }
  buf = (char *)malloc((size_t)size + 2);
  if (buf == NULL) {
    fclose(f);
    return PyErr_NoMemory();
  }
  Py_BEGIN_ALLOW_THREADS
  got = (long)fread(buf, 1, (size_t)size, f);
  Py_END_ALLOW_THREADS
  fclose(f);
  if (got != size) {
    free(buf);
    PyErr_SetString(PyExc_IOError, "error reading MTX file");
    return NULL;
  }
  buf[size] = '\n';
  buf[size + 1] = '\0';
  if (sscanf(buf, "%63s %63s %63s %63s %63s", banner, mtx, crd, dtype, sym) != 5 ||
      strcmp(banner, "%%MatrixMarket") != 0) {
    free(buf);
    PyErr_SetString(PyExc_IOError, "error reading MTX file header");
    return NULL;
  }
  for (p = mtx; *p; p++) *p = (char)tolower(*p);
  for (p = crd; *p; p++) *p = (char)tolower(*p);
  for (p = dtype; *p; p++) *p = (char)tolower(*p);
  for (p = sym; *p; p++) *p = (char)tolower(*p);
  is_sym = strcmp(sym, "symmetric") == 0;
  if (strcmp(mtx, "matrix") != 0 || strcmp(crd, "coordinate") != 0 ||
      (strcmp(dtype, "real") != 0 && strcmp(dtype, "integer") != 0) || (!is_sym && strcmp(sym, "general") != 0)) {
    free(buf);
    PyErr_SetString(SpMatrix_ErrorObject, "must be real, sparse matrix");
    return NULL;
  }
  /* the size line: the first line behind the banner that is neither a comment nor blank */
  p = (char *)memchr(buf, '\n', (size_t)size + 1);
  for (;;) {
    const char *q;
    if (p == NULL || p + 1 >= buf + size) {
      free(buf);
      PyErr_SetString(PyExc_IOError, "error reading MTX file size information");
      return NULL;
    }
    p++;
    q = mtx_skip_blank(p, buf + size);
    if (*q != '%' && *q != '\n') break;
    p = (char *)memchr(p, '\n', (size_t)(buf + size + 1 - p));
  }
  if (sscanf(p, "%ld %ld %ld", &m, &n, &nz) != 3 || m < 0 || n < 0 || nz < 0) {
    free(buf);
    PyErr_SetString(PyExc_IOError, "error reading MTX file size information");
    return NULL;
  }
  data = (char *)memchr(p, '\n', (size_t)(buf + size + 1 - p)) + 1;
  T = threads > 0 ? threads : (int)sysconf(_SC_NPROCESSORS_ONLN);
  if (threads <= 0 && T > 8) T = 8;
  if (T < 1) T = 1;
  if (T > 64) T = 64;
  if ((buf + size + 1) - data < (1 << 16)) T = 1;
  { /* chunks end behind a newline */
    const char *e = buf + size + 1;
    const char *cur = data;
    for (t = 0; t < T; t++) {
      const char *stop = t == T - 1 ? e : data + (long)((double)(e - data) * (t + 1) / T);
      if (stop < cur) stop = cur;
      if (t < T - 1) {
        const char *nl = (const char *)memchr(stop, '\n', (size_t)(e - stop));
        stop = nl ? nl + 1 : e;
      }
      memset(&ch[t], 0, sizeof ch[t]);
      ch[t].p = cur;
      ch[t].e = stop;
      ch[t].m = m;
      ch[t].n = n;
      cur = stop;
    }
  }
  Py_BEGIN_ALLOW_THREADS
  for (t = 1; t < T; t++)
    if (pthread_create(&th[t], NULL, mtx_count_worker, &ch[t]) != 0) {
      mtx_count_worker(&ch[t]); /* no thread: count this chunk here; buf stays alive until every started thread is joined */
      th[t] = 0;
    }
  mtx_count_worker(&ch[0]);
  for (t = 1; t < T; t++)
    if (th[t]) pthread_join(th[t], NULL);
  Py_END_ALLOW_THREADS
  for (t = 0; t < T; t++) {
    ch[t].off = total;
    total += ch[t].cnt;
  }
  if (total != nz) {
    free(buf);
    PyErr_Format(SpMatrix_ErrorObject, "file holds %ld entries, the size line promises %ld", total, nz);
    return NULL;
  }
  dims[0] = (npy_intp)nz;
  ri = (PyArrayObject *)PyArray_SimpleNew(1, dims, NPY_INT64);
  ci = (PyArrayObject *)PyArray_SimpleNew(1, dims, NPY_INT64);
  va = (PyArrayObject *)PyArray_SimpleNew(1, dims, NPY_DOUBLE);
  if (ri == NULL || ci == NULL || va == NULL) {
    free(buf);
    Py_XDECREF(ri);
    Py_XDECREF(ci);
    Py_XDECREF(va);
    return NULL;
  }
  for (t = 0; t < T; t++) {
    ch[t].ri = (npy_int64 *)PyArray_DATA(ri);
    ch[t].ci = (npy_int64 *)PyArray_DATA(ci);
    ch[t].v = (double *)PyArray_DATA(va);
  }
  Py_BEGIN_ALLOW_THREADS
  for (t = 1; t < T; t++)
    if (pthread_create(&th[t], NULL, mtx_parse_worker, &ch[t]) != 0) {
      mtx_parse_worker(&ch[t]); /* no thread: parse it here */
      th[t] = 0;
    }
  mtx_parse_worker(&ch[0]);
  for (t = 1; t < T; t++)
    if (th[t]) pthread_join(th[t], NULL);
  Py_END_ALLOW_THREADS
  free(buf);
  for (t = 0; t < T; t++)
    if (ch[t].err > err) err = ch[t].err;
  if (err) {
    Py_DECREF(ri);
    Py_DECREF(ci);
    Py_DECREF(va);
    if (err == 2)
      PyErr_SetString(PyExc_IndexError, "matrix indices out of range");
    else
      PyErr_SetString(PyExc_IOError, "error reading MTX file data");
    return NULL;
  }
  res = Py_BuildValue("llONNN", m, n, is_sym ? Py_True : Py_False, (PyObject *)ri, (PyObject *)ci, (PyObject *)va);
  return res;
}

/* coo_sort_unique(rows, cols, vals, nrows) -> (rows, cols, vals) sorted by (row, col); of entries with the same (row, col)
 * the LAST in input order survives -- what repeated ll_mat assignments leave behind (SpMatrix_LLMatSetItem, ll_mat.c:250-356).
 * A counting sort by row (stable), then every row's entries by column (stable insertion / merge sort: rows are short),
 * rows split over threads. */
typedef struct {
  long r0, r1;
  const npy_int64 *start; /* nrows + 1 */
  npy_int64 *c;
  double *v;
  npy_int64 *tc;
  double *tv;
  long *keep; /* per row: entries kept */
} CooRows;

static void coo_merge_sort(npy_int64 *c, double *v, npy_int64 *tc, double *tv, long n) {
  long w, i;
  if (n <= 32) { /* insertion sort, stable */
    for (i = 1; i < n; i++) {
      npy_int64 kc = c[i];
      double kv = v[i];
      long j = i - 1;
      while (j >= 0 && c[j] > kc) {
        c[j + 1] = c[j];
        v[j + 1] = v[j];
        j--;
      }
      c[j + 1] = kc;
      v[j + 1] = kv;
    }
    return;
  }
  for (w = 1; w < n; w *= 2) { /* bottom-up merge sort, stable */
    for (i = 0; i < n; i += 2 * w) {
      long a = i, am = i + w < n ? i + w : n, b = am, bm = i + 2 * w < n ? i + 2 * w : n, k = i;
      while (a < am && b < bm) {
        if (c[b] < c[a]) { tc[k] = c[b]; tv[k++] = v[b++]; }
        else { tc[k] = c[a]; tv[k++] = v[a++]; }
      }
      while (a < am) { tc[k] = c[a]; tv[k++] = v[a++]; }
      while (b < bm) { tc[k] = c[b]; tv[k++] = v[b++]; }
    }
    memcpy(c, tc, sizeof(npy_int64) * (size_t)n);
    memcpy(v, tv, sizeof(double) * (size_t)n);
  }
}

static void *coo_rows_worker(void *arg) {
  CooRows *w = (CooRows *)arg;
  long r;
  for (r = w->r0; r < w->r1; r++) {
    const long a = (long)w->start[r], n = (long)w->start[r + 1] - a;
    npy_int64 *c = w->c + a;
    double *v = w->v + a;
    long i, k = 0;
    if (n > 1) coo_merge_sort(c, v, w->tc + a, w->tv + a, n);
    for (i = 0; i < n; i++) /* of equal columns the last one (= last in input order) stays */
      if (i + 1 == n || c[i + 1] != c[i]) {
        c[k] = c[i];
        v[k++] = v[i];
      }
    w->keep[r] = k;
  }
  return NULL;
}

static PyObject *Coo_sort_unique(PyObject *module, PyObject *args) {
  PyObject *orows, *ocols, *ovals;
  PyArrayObject *rows = NULL, *cols = NULL, *vals = NULL, *R = NULL, *Cc = NULL, *V = NULL;
  long nrows, nz, i, r, T, t, total;
  npy_int64 *start = NULL, *fill = NULL, *c = NULL, *tc = NULL;
  double *v = NULL, *tv = NULL;
  long *keep = NULL;
  const npy_int64 *ri, *ci;
  const double *vi;
  CooRows ws[64];
  pthread_t th[64];
  npy_intp dims[1];
  int bad = 0;
  if (!PyArg_ParseTuple(args, "OOOl", &orows, &ocols, &ovals, &nrows)) return NULL;
  rows = (PyArrayObject *)PyArray_FROM_OTF(orows, NPY_INT64, NPY_ARRAY_IN_ARRAY);
  cols = (PyArrayObject *)PyArray_FROM_OTF(ocols, NPY_INT64, NPY_ARRAY_IN_ARRAY);
  vals = (PyArrayObject *)PyArray_FROM_OTF(ovals, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY);
  if (rows == NULL || cols == NULL || vals == NULL) goto fail;
  nz = (long)PyArray_SIZE(rows);
  if (PyArray_NDIM(rows) != 1 || PyArray_SIZE(cols) != nz || PyArray_SIZE(vals) != nz || nrows < 0) {
    PyErr_SetString(PyExc_ValueError, "coo_sort_unique: rows, cols, vals must be 1-D arrays of one length");
    goto fail;
  }
  ri = (const npy_int64 *)PyArray_DATA(rows);
  ci = (const npy_int64 *)PyArray_DATA(cols);
  vi = (const double *)PyArray_DATA(vals);
  start = (npy_int64 *)calloc((size_t)nrows + 2, sizeof(npy_int64));
  fill = (npy_int64 *)malloc(sizeof(npy_int64) * ((size_t)nrows + 1));
  c = (npy_int64 *)malloc(sizeof(npy_int64) * (size_t)(nz ? nz : 1));
  tc = (npy_int64 *)malloc(sizeof(npy_int64) * (size_t)(nz ? nz : 1));
  v = (double *)malloc(sizeof(double) * (size_t)(nz ? nz : 1));
  tv = (double *)malloc(sizeof(double) * (size_t)(nz ? nz : 1));
  keep = (long *)calloc((size_t)nrows + 1, sizeof(long));
  if (!start || !fill || !c || !tc || !v || !tv || !keep) {
    PyErr_NoMemory();
    goto fail;
  }
  Py_BEGIN_ALLOW_THREADS
  for (i = 0; i < nz; i++) {
    if (ri[i] < 0 || ri[i] >= nrows) { bad = 1; break; }
    start[ri[i] + 1]++;
  }
  if (!bad) {
    for (r = 0; r < nrows; r++) start[r + 1] += start[r];
    memcpy(fill, start, sizeof(npy_int64) * (size_t)nrows);
    for (i = 0; i < nz; i++) { /* stable: input order inside a row */
      const npy_int64 k = fill[ri[i]]++;
      c[k] = ci[i];
      v[k] = vi[i];
    }
    T = sysconf(_SC_NPROCESSORS_ONLN);
    if (T > 8) T = 8;
    if (T < 1 || nz < (1 << 16)) T = 1;
    for (t = 0; t < T; t++) { /* rows split by entry count */
      ws[t].start = start;
      ws[t].c = c;
      ws[t].v = v;
      ws[t].tc = tc;
      ws[t].tv = tv;
      ws[t].keep = keep;
    }
    {
      long rr = 0;
      for (t = 0; t < T; t++) {
        const long want = (long)((double)nz * (t + 1) / T);
        ws[t].r0 = rr;
        while (rr < nrows && (t == T - 1 || (long)start[rr + 1] <= want)) rr++;
        ws[t].r1 = rr;
      }
    }
    for (t = 1; t < T; t++)
      if (pthread_create(&th[t], NULL, coo_rows_worker, &ws[t]) != 0) {
        coo_rows_worker(&ws[t]);
        th[t] = 0;
      }
    coo_rows_worker(&ws[0]);
    for (t = 1; t < T; t++)
      if (th[t]) pthread_join(th[t], NULL);
  }
  Py_END_ALLOW_THREADS
  if (bad) {
    PyErr_SetString(PyExc_IndexError, "indices out of range");
    goto fail;
  }
  total = 0;
  for (r = 0; r < nrows; r++) total += keep[r];
  dims[0] = (npy_intp)total;
  R = (PyArrayObject *)PyArray_SimpleNew(1, dims, NPY_INT64);
  Cc = (PyArrayObject *)PyArray_SimpleNew(1, dims, NPY_INT64);
  V = (PyArrayObject *)PyArray_SimpleNew(1, dims, NPY_DOUBLE);
  if (R == NULL || Cc == NULL || V == NULL) goto fail;
  {
    npy_int64 *ro = (npy_int64 *)PyArray_DATA(R), *co = (npy_int64 *)PyArray_DATA(Cc);
    double *vo = (double *)PyArray_DATA(V);
    long k = 0;
    for (r = 0; r < nrows; r++) {
      const long a = (long)start[r];
      for (i = 0; i < keep[r]; i++, k++) {
        ro[k] = r;
        co[k] = c[a + i];
        vo[k] = v[a + i];
      }
    }
  }
  free(start); free(fill); free(c); free(tc); free(v); free(tv); free(keep);
  Py_DECREF(rows);
  Py_DECREF(cols);
  Py_DECREF(vals);
  return Py_BuildValue("NNN", (PyObject *)R, (PyObject *)Cc, (PyObject *)V);
fail:
  free(start); free(fill); free(c); free(tc); free(v); free(tv); free(keep);
  Py_XDECREF(rows);
  Py_XDECREF(cols);
  Py_XDECREF(vals);
  Py_XDECREF(R);
  Py_XDECREF(Cc);
  Py_XDECREF(V);
  return NULL;
}

/* csr_from_arrays(indptr, indices, data, shape): scalable constructor without an ll_mat */
/* devices=[...] of poisson_csr / csr_from_arrays: a sequence of device ordinals (one rank per entry; may repeat).
 * Returns the count (0: None / absent -> single device), -1 with an exception set. */
static int parse_devices(PyObject *o, int *dev, int cap) {
  PyObject *seq, *it;
  Py_ssize_t i, n;
  if (o == NULL || o == Py_None) return 0;
  seq = PySequence_Fast(o, "devices must be a sequence of device ordinals");
  if (seq == NULL) return -1;
  n = PySequence_Fast_GET_SIZE(seq);
  if (n < 1 || n > cap) {
    Py_DECREF(seq);
    PyErr_Format(PyExc_ValueError, "devices must list 1..%d device ordinals", cap);
    return -1;
  }
  for (i = 0; i < n; i++) {
    it = PySequence_Fast_GET_ITEM(seq, i);
    dev[i] = (int)PyLong_AsLong(it);
    if (dev[i] == -1 && PyErr_Occurred()) {
      Py_DECREF(seq);
      return -1;
    }
  }
  Py_DECREF(seq);
  return (int)n;
}

static PyObject *CSR_from_arrays(PyObject *module, PyObject *args, PyObject *kwds) {
  PyObject *oi, *oc, *ov;
  PyArrayObject *ind = NULL, *col = NULL, *val = NULL;
  int dim[2], rc, keep_host = 0, ndev, devs[64];
  CSRMatObject *op = NULL;
  npy_intp nnz;
  PyObject *odev = NULL;
  static char *kwlist[] = {"indptr", "indices", "data", "shape", "keep_host", "devices", NULL};
  if (!PyArg_ParseTupleAndKeywords(args, kwds, "OOO(ii)|iO", kwlist, &oi, &oc, &ov, dim, dim + 1, &keep_host, &odev))
    return NULL;
  if ((ndev = parse_devices(odev, devs, 64)) < 0) return NULL;
  ind = (PyArrayObject *)PyArray_FROM_OTF(oi, NPY_INT32, NPY_ARRAY_IN_ARRAY);
  col = ind ? (PyArrayObject *)PyArray_FROM_OTF(oc, NPY_INT32, NPY_ARRAY_IN_ARRAY) : NULL;
  val = col ? (PyArrayObject *)PyArray_FROM_OTF(ov, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY) : NULL;
  if (!val) goto done;
  nnz = PyArray_SIZE(val);
  if (PyArray_NDIM(ind) != 1 || PyArray_SIZE(ind) != dim[0] + 1 || PyArray_SIZE(col) != nnz ||
      nnz > 0x7fffffff) {
    PyErr_SetString(PyExc_ValueError, "inconsistent CSR arrays");
    goto done;
  }
  op = (CSRMatObject *)newCSRMatObject(dim, (int)nnz, keep_host);
  if (op == NULL) goto done;
  if (keep_host) {
    memcpy(op->ind, PyArray_DATA(ind), sizeof(int) * (size_t)(dim[0] + 1));
    memcpy(op->col, PyArray_DATA(col), sizeof(int) * (size_t)nnz);
    memcpy(op->val, PyArray_DATA(val), sizeof(double) * (size_t)nnz);
  }
  Py_BEGIN_ALLOW_THREADS
  if (ndev > 0) /* row blocks on a list of devices: matvec / jacobi / pcg / minres (psp_csr_create_multi) */
    rc = psp_csr_create_multi(dim[0], dim[1], (int)nnz, (int *)PyArray_DATA(ind), (int *)PyArray_DATA(col),
                              (double *)PyArray_DATA(val), devs, ndev, &op->dev);
  else
    rc = psp_csr_create(dim[0], dim[1], (int)nnz, (int *)PyArray_DATA(ind), (int *)PyArray_DATA(col),
                        (double *)PyArray_DATA(val), &op->dev);
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) {
    Py_DECREF(op);
    op = NULL;
    psp_raise(rc);
  }
done:
  Py_XDECREF(ind);
  Py_XDECREF(col);
  Py_XDECREF(val);
  return (PyObject *)op;
}

/* sss_from_arrays(indptr, indices, data, diag) */
static PyObject *SSS_from_arrays(PyObject *module, PyObject *args) {
  PyObject *oi, *oc, *ov, *od;
  PyArrayObject *ind = NULL, *col = NULL, *val = NULL, *diag = NULL;
  int rc, n;
  npy_intp nnz;
  SSSMatObject *op = NULL;
  if (!PyArg_ParseTuple(args, "OOOO", &oi, &oc, &ov, &od)) return NULL;
  ind = (PyArrayObject *)PyArray_FROM_OTF(oi, NPY_INT32, NPY_ARRAY_IN_ARRAY);
  col = ind ? (PyArrayObject *)PyArray_FROM_OTF(oc, NPY_INT32, NPY_ARRAY_IN_ARRAY) : NULL;
  val = col ? (PyArrayObject *)PyArray_FROM_OTF(ov, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY) : NULL;
  diag = val ? (PyArrayObject *)PyArray_FROM_OTF(od, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY) : NULL;
  if (!diag) goto done;
  n = (int)PyArray_SIZE(diag);
  nnz = PyArray_SIZE(val);
  if (PyArray_SIZE(ind) != n + 1 || PyArray_SIZE(col) != nnz || nnz > 0x7fffffff) {
    PyErr_SetString(PyExc_ValueError, "inconsistent SSS arrays");
    goto done;
  }
  op = (SSSMatObject *)newSSSMatObject(n, (int)nnz, 1);
  if (op == NULL) goto done;
  memcpy(op->ind, PyArray_DATA(ind), sizeof(int) * (size_t)(n + 1));
  memcpy(op->col, PyArray_DATA(col), sizeof(int) * (size_t)nnz);
  memcpy(op->val, PyArray_DATA(val), sizeof(double) * (size_t)nnz);
  memcpy(op->diag, PyArray_DATA(diag), sizeof(double) * (size_t)n);
  Py_BEGIN_ALLOW_THREADS
  rc = psp_sss_create(n, (int)nnz, op->ind, op->col, op->val, op->diag, &op->dev);
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) {
    Py_DECREF(op);
    op = NULL;
    psp_raise(rc);
  }
done:
  Py_XDECREF(ind);
  Py_XDECREF(col);
  Py_XDECREF(val);
  Py_XDECREF(diag);
  return (PyObject *)op;
}

/* poisson_csr(nx, ny, nz=0) / poisson_sss(...): generated on the device in the ordering of
 * pysparse/tools/poisson.py:22-50 (k = i + nx*j + nx*ny*l) */
static PyObject *Poisson_csr(PyObject *module, PyObject *args, PyObject *kwds) {
  int nx, ny, nz = 0, rc, dim[2], nnz, ndev, devs[64];
  CSRMatObject *op;
  PyObject *odev = NULL;
  static char *kwlist[] = {"nx", "ny", "nz", "devices", NULL};
  if (!PyArg_ParseTupleAndKeywords(args, kwds, "ii|iO", kwlist, &nx, &ny, &nz, &odev)) return NULL;
  if ((ndev = parse_devices(odev, devs, 64)) < 0) return NULL;
  dim[0] = dim[1] = 0;
  op = (CSRMatObject *)newCSRMatObject(dim, 0, 0);
  if (op == NULL) return NULL;
  Py_BEGIN_ALLOW_THREADS
  if (ndev > 0) /* z-slabs on a list of devices, one process (psp_csr_poisson_multi) */
    rc = psp_csr_poisson_multi(nx, ny, nz, devs, ndev, &op->dev);
  else if (7.0 * (double)nx * (double)ny * (double)(nz > 0 ? nz : 1) > 2147483647.0)
    /* more than 2^31 stored entries (1024^3: 7.5e9) on ONE device: the index-free operator with 64-bit row offsets
     * (configs[3]'s one-GPU baseline); nnz then reads through psp_csr_nnz64 */
    rc = psp_csr_poisson_big(nx, ny, nz, &op->dev);
  else
    rc = psp_csr_poisson(nx, ny, nz, &op->dev);
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) {
    Py_DECREF(op);
    return psp_raise(rc);
  }
  psp_csr_shape(op->dev, &op->dim[0], &op->dim[1], &nnz);
  op->nnz = nnz;
  return (PyObject *)op;
}

static PyObject *Poisson_sss(PyObject *module, PyObject *args) {
  int nx, ny, nz = 0, rc, rep;
  SSSMatObject *op;
  if (!PyArg_ParseTuple(args, "ii|i", &nx, &ny, &nz)) return NULL;
  op = (SSSMatObject *)newSSSMatObject(0, 0, 0);
  if (op == NULL) return NULL;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_sss_poisson(nx, ny, nz, &op->dev);
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) {
    Py_DECREF(op);
    return psp_raise(rc);
  }
  psp_sss_shape(op->dev, &op->n, &rep);
  op->nnz = rep - op->n;
  return (PyObject *)op;
}

static PyObject *Device_count(PyObject *module, PyObject *args) {
  return PyLong_FromLong(psp_device_count());
}

static PyMethodDef spmatrix_methods[] = {
    {"ll_mat", LLMat_zeros, METH_VARARGS, "ll_mat(n, m, sizeHint=1000, storeZeros=0): empty n x m linked-list matrix"},
    {"ll_mat_sym", LLMat_sym_zeros, METH_VARARGS, "ll_mat_sym(n, sizeHint=1000, storeZeros=0): empty symmetric matrix"},
    {"ll_mat_from_mtx", LLMat_from_mtx, METH_VARARGS, "ll_mat_from_mtx(fileName): read a MatrixMarket coordinate file"},
    {"matrixmultiply", LLMat_matrixmultiply, METH_VARARGS, "matrixmultiply(A, B): new ll_mat A * B"},
    {"dot", LLMat_dot, METH_VARARGS, "dot(A, B): new ll_mat transpose(A) * B"},
    {"symdot", LLMat_symdot, METH_VARARGS, "symdot(A[, d]): new symmetric ll_mat transpose(A) * A or transpose(A) * diag(d) * A"},
    {"coo_sort_unique", Coo_sort_unique, METH_VARARGS,
     "coo_sort_unique(rows, cols, vals, nrows) -> (rows, cols, vals) sorted by (row, col); a repeated (row, col) keeps its last value"},
    {"mtx_read_coordinate", Mtx_read_coordinate, METH_VARARGS,
     "mtx_read_coordinate(fileName, threads=0) -> (m, n, symmetric, rows, cols, vals): the entries of a coordinate MatrixMarket "
     "file in file order (0-based int64 / float64), parsed in parallel"},
    {"csr_from_arrays", (PyCFunction)(void (*)(void))CSR_from_arrays, METH_VARARGS | METH_KEYWORDS,
     "csr_from_arrays(indptr, indices, data, shape[, keep_host, devices=[...]]) -> csr_mat (devices: row blocks on several GPUs)"},
    {"sss_from_arrays", SSS_from_arrays, METH_VARARGS, "sss_from_arrays(indptr, indices, data, diag) -> sss_mat"},
    {"poisson_csr", (PyCFunction)(void (*)(void))Poisson_csr, METH_VARARGS | METH_KEYWORDS,
     "poisson_csr(nx, ny, nz=0, devices=None) -> csr_mat of the 5-/7-point operator, built on the GPU(s); devices=[0, 1, ...] "
     "partitions it into slabs over those GPUs: matvec, precon.jacobi, krylov.pcg and krylov.minres then run on all of them"},
    {"poisson_sss", Poisson_sss, METH_VARARGS, "poisson_sss(nx, ny, nz=0) -> sss_mat of the 5-/7-point operator, built on the GPU"},
    {"device_count", Device_count, METH_NOARGS, "number of visible GPUs"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef spmatrix_module = {PyModuleDef_HEAD_INIT, "spmatrix",
                                             "ll_mat / csr_mat / sss_mat on MI355X", -1,
                                             spmatrix_methods, NULL, NULL, NULL, NULL};

static void init_type(PyTypeObject *t, const char *name, size_t size, destructor dealloc,
                      reprfunc repr, PyMethodDef *methods, PyGetSetDef *getset,
                      PyMappingMethods *mapping) {
  t->tp_name = name;
  t->tp_basicsize = (Py_ssize_t)size;
  t->tp_dealloc = dealloc;
  t->tp_repr = repr;
  t->tp_flags = Py_TPFLAGS_DEFAULT;
  t->tp_methods = methods;
  t->tp_getset = getset;
  t->tp_as_mapping = mapping;
}

PyMODINIT_FUNC PyInit_spmatrix(void) {
  static void *api[SpMatrix_API_pointers];
  PyObject *m, *cap;
  import_array();
  {
    PyTypeObject zero = {PyVarObject_HEAD_INIT(NULL, 0)};
    LLMatType = zero;
    CSRMatType = zero;
    SSSMatType = zero;
  }
  init_type(&LLMatType, "pysparse_amd.sparse.spmatrix.ll_mat", sizeof(LLMatObject),
            (destructor)LLMat_dealloc, (reprfunc)LLMat_repr, LLMat_methods, LLMat_getset,
            &LLMat_as_mapping);
  init_type(&CSRMatType, "pysparse_amd.sparse.spmatrix.csr_mat", sizeof(CSRMatObject),
            (destructor)CSRMat_dealloc, (reprfunc)CSRMat_repr, CSRMat_methods, CSRMat_getset, NULL);
  init_type(&SSSMatType, "pysparse_amd.sparse.spmatrix.sss_mat", sizeof(SSSMatObject),
            (destructor)SSSMat_dealloc, (reprfunc)SSSMat_repr, SSSMat_methods, SSSMat_getset,
            &SSSMat_as_mapping);
  LLMatType.tp_str = (reprfunc)LLMat_str; /* print(A): the text of the reference's tp_print slots */
  CSRMatType.tp_str = (reprfunc)CSRMat_str;
  SSSMatType.tp_str = (reprfunc)SSSMat_str;
  if (PyType_Ready(&LLMatType) < 0 || PyType_Ready(&CSRMatType) < 0 || PyType_Ready(&SSSMatType) < 0)
    return NULL;
  m = PyModule_Create(&spmatrix_module);
  if (m == NULL) return NULL;
  Py_INCREF(&LLMatType);
  PyModule_AddObject(m, "LLMatType", (PyObject *)&LLMatType);
  Py_INCREF(&CSRMatType);
  PyModule_AddObject(m, "CSRMatType", (PyObject *)&CSRMatType);
  Py_INCREF(&SSSMatType);
  PyModule_AddObject(m, "SSSMatType", (PyObject *)&SSSMatType);
  SpMatrix_ErrorObject = PyErr_NewException("pysparse_amd.sparse.spmatrix.error", NULL, NULL);
  Py_XINCREF(SpMatrix_ErrorObject);
  PyModule_AddObject(m, "error", SpMatrix_ErrorObject);

  api[LLMatType_NUM] = (void *)&LLMatType;
  api[CSRMatType_NUM] = (void *)&CSRMatType;
  api[SSSMatType_NUM] = (void *)&SSSMatType;
  api[SpMatrix_ParseVecOpArgs_NUM] = (void *)SpMatrix_ParseVecOpArgs;
  api[SpMatrix_GetShape_NUM] = (void *)SpMatrix_GetShape;
  api[SpMatrix_GetOrder_NUM] = (void *)SpMatrix_GetOrder;
  api[SpMatrix_GetItem_NUM] = (void *)SpMatrix_GetItem;
  api[SpMatrix_Matvec_NUM] = (void *)SpMatrix_Matvec;
  api[SpMatrix_Precon_NUM] = (void *)SpMatrix_Precon;
  api[SpMatrix_NewLLMatObject_NUM] = (void *)SpMatrix_NewLLMatObject;
  api[SpMatrix_LLMatGetItem_NUM] = (void *)SpMatrix_LLMatGetItem;
  api[SpMatrix_LLMatSetItem_NUM] = (void *)SpMatrix_LLMatSetItem;
  api[SpMatrix_LLMatUpdateItemAdd_NUM] = (void *)SpMatrix_LLMatUpdateItemAdd;
  api[SpMatrix_LLMatBuildColIndex_NUM] = (void *)SpMatrix_LLMatBuildColIndex;
  api[SpMatrix_LLMatDestroyColIndex_NUM] = (void *)SpMatrix_LLMatDestroyColIndex;
  api[ItSolvers_Solve_NUM] = (void *)ItSolvers_Solve;
  cap = PyCapsule_New((void *)api, SPMATRIX_CAPSULE_NAME, NULL);
  if (cap == NULL) return NULL;
  PyModule_AddObject(m, "_C_API", cap);
  return m;
}
