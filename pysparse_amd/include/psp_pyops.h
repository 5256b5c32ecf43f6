/*
 * psp_pyops.h -- turning Python objects into psp_op_t operators (shared by the krylov and
 * precon extension modules).
 *
 *   native objects (csr_mat, sss_mat, ll_mat, jacobi) expose a private attribute
 *   `_psp_op` = PyCapsule("psp_op_t") over a handle the object owns  -> device operator;
 *   anything else with `shape` + `matvec` / `precon` (the reference's duck-typed protocol,
 *   spmatrixmodule.c:86-132, :169-248) -> host-callback operator whose trampoline wraps the
 *   raw host pointers in temporary ndarrays and calls the method.
 */
#ifndef PSP_PYOPS_H
#define PSP_PYOPS_H

#include "spmatrix_api.h"

typedef struct {
  psp_op_t *op;      /* operator handed to the solver */
  int owned;         /* 1: created here (callback op), destroy after the solve */
  int is_callback;
  PyObject *keep;    /* capsule reference keeping a native handle alive */
  PyObject *obj;     /* borrowed: the Python object behind a callback op */
  int is_precon;
} PyOpRef;

/* Called by libpysparse_hip.so from inside a solve.  The extension modules ALWAYS release the GIL before they
 * enter the library (its compute entry points take a library-wide lock, and nobody may wait for that lock while
 * holding the GIL: a callback of the lock's holder would never get it), so the callback takes the GIL itself. */
static int pyop_trampoline(void *ctx, int n, const double *x, double *y) {
  PyOpRef *r = (PyOpRef *)ctx;
  PyGILState_STATE g = PyGILState_Ensure();
  int rc = r->is_precon ? SpMatrix_Precon(r->obj, n, (double *)x, y)
                        : SpMatrix_Matvec(r->obj, n, (double *)x, n, y);
  PyGILState_Release(g);
  return rc;
}

/* returns 0 and fills *r, or -1 with a Python exception set.  *n_out = operator order. */
static int pyop_acquire(PyObject *obj, int is_precon, PyOpRef *r, int *n_out) {
  PyObject *cap;
  int rc, n;
  memset(r, 0, sizeof *r);
  r->obj = obj;
  r->is_precon = is_precon;
  if (SpMatrix_GetOrder(obj, &n)) return -1; /* shape must exist and be square */
  *n_out = n;
  cap = PyObject_GetAttrString(obj, "_psp_op");
  if (cap != NULL && PyCapsule_IsValid(cap, PSP_OP_CAPSULE_NAME)) {
    r->op = (psp_op_t *)PyCapsule_GetPointer(cap, PSP_OP_CAPSULE_NAME);
    r->keep = cap;
    return 0;
  }
  if (cap == NULL) {
    if (!PyErr_ExceptionMatches(PyExc_AttributeError)) return -1; /* a native getter failed */
    PyErr_Clear();
  } else {
    Py_DECREF(cap);
  }
  if (n <= 0) {
    PyErr_SetString(PyExc_ValueError, "invalid matrix shape");
    return -1;
  }
  rc = psp_op_from_callback(n, pyop_trampoline, (void *)r, &r->op);
  if (rc != PSP_OK) {
    PyErr_SetString(rc == PSP_ENOMEM ? PyExc_MemoryError : PyExc_RuntimeError, psp_last_error());
    return -1;
  }
  r->owned = 1;
  r->is_callback = 1;
  return 0;
}

static void pyop_release(PyOpRef *r) {
  if (r->owned && r->op) psp_op_destroy(r->op);
  Py_XDECREF(r->keep);
  memset(r, 0, sizeof *r);
}

#endif
