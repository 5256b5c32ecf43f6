/*
 * pysparse_amd.precon.precon -- jacobi(A, omega=1.0, steps=1) and ssor(A, omega=1.0, steps=1):
 * objects with `shape` and `precon(x, y)` (pysparse/precon/src/preconmodule.c:11-80, 352-412,
 * 470-485 for jacobi; :21-31, 95-223, 414-459, 487-510 for ssor).
 *
 * dinv[i] = omega / A[i,i] lives on the GPU.  Native matrices hand over their diagonal on
 * the device (csr_mat / sss_mat; the reference can only subscript ll_mat, so jacobi(csr_mat)
 * is an extension); ll_mat and foreign objects are read through A[i,i] exactly like
 * newJacobiObject does (:389-401).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

#include <string.h>

#include "psp_pyops.h"

typedef struct {
  PyObject_VAR_HEAD
  int n;
  PyObject *matrix;
  double omega;
  int steps;
  psp_jacobi_t *dev;
  psp_op_t *op;    /* operator over dev, handed to the solvers */
  PyOpRef aref;    /* operator of the matrix (steps > 1 sweeps) */
  int have_aref;
} JacobiObject;

static PyTypeObject JacobiType;

static PyObject *raise_psp(int rc) {
  if (rc == PSP_ESINGULAR)
    PyErr_SetString(PyExc_ValueError, "diagonal element close to zero"); /* :396 */
  else if (rc == PSP_ENOMEM)
    PyErr_SetString(PyExc_MemoryError, psp_last_error());
  else if (rc == PSP_EINVAL)
    PyErr_SetString(PyExc_ValueError, psp_last_error());
  else
    PyErr_SetString(PyExc_RuntimeError, psp_last_error());
  return NULL;
}

static PyObject *newJacobiObject(PyObject *matrix, double omega, int steps) {
  JacobiObject *op;
  int n, rc, i;
  if (steps < 1) {
    PyErr_SetString(PyExc_ValueError, "steps must be >= 1");
    return NULL;
  }
  if (SpMatrix_GetOrder(matrix, &n)) return NULL;
  op = PyObject_New(JacobiObject, &JacobiType);
  if (op == NULL) return PyErr_NoMemory();
  op->n = n;
  op->matrix = NULL;
  op->omega = omega;
  op->steps = steps;
  op->dev = NULL;
  op->op = NULL;
  op->have_aref = 0;

  if (PyObject_TypeCheck(matrix, &CSRMatType)) {
    Py_BEGIN_ALLOW_THREADS
    rc = psp_jacobi_create_csr(((CSRMatObject *)matrix)->dev, omega, steps, &op->dev);
    Py_END_ALLOW_THREADS
  } else if (PyObject_TypeCheck(matrix, &SSSMatType)) {
    Py_BEGIN_ALLOW_THREADS
    rc = psp_jacobi_create_sss(((SSSMatObject *)matrix)->dev, omega, steps, &op->dev);
    Py_END_ALLOW_THREADS
  } else {
    double *diag = PyMem_New(double, n > 0 ? n : 1);
    int nn;
    if (diag == NULL) {
      Py_DECREF(op);
      return PyErr_NoMemory();
    }
    for (i = 0; i < n; i++) {
      if (PyObject_TypeCheck(matrix, &LLMatType))
        diag[i] = SpMatrix_LLMatGetItem((LLMatObject *)matrix, i, i);
      else
        diag[i] = SpMatrix_GetItem(matrix, i, i); /* :391 */
      if (PyErr_Occurred()) {
        PyMem_Del(diag);
        Py_DECREF(op);
        return NULL;
      }
    }
    if (steps > 1) {
      if (pyop_acquire(matrix, 0, &op->aref, &nn)) {
        PyMem_Del(diag);
        Py_DECREF(op);
        return NULL;
      }
      op->have_aref = 1;
    }
    Py_BEGIN_ALLOW_THREADS
    rc = psp_jacobi_create_diag(n, diag, omega, steps, op->have_aref ? op->aref.op : NULL, &op->dev);
    Py_END_ALLOW_THREADS
    PyMem_Del(diag);
  }
  if (rc != PSP_OK) {
    Py_DECREF(op);
    return raise_psp(rc);
  }
  Py_INCREF(matrix);
  op->matrix = matrix;
  return (PyObject *)op;
}

/* jacobi(A, omega=1.0, steps=1): preconmodule.c:470-485 */
static PyObject *jacobi_prec(PyObject *self, PyObject *args, PyObject *kw) {
  static char *kwlist[] = {"A", "omega", "steps", NULL};
  PyObject *matrix;
  double omega = 1.0;
  int steps = 1;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "O|di", kwlist, &matrix, &omega, &steps)) return NULL;
  return newJacobiObject(matrix, omega, steps);
}

/* self.precon(x, y): Jacobi_precon, preconmodule.c:60-80 -- contiguous arrays only */
static PyObject *Jacobi_precon(JacobiObject *self, PyObject *args) {
  double *x, *y;
  int rc;
  if (SpMatrix_ParseVecOpArgs(args, &x, &y, self->n)) return NULL;
  Py_BEGIN_ALLOW_THREADS /* sweeps of a callback operator take the GIL back in pyop_trampoline */
  rc = psp_jacobi_precon(self->dev, x, y);
  Py_END_ALLOW_THREADS
  if (PyErr_Occurred()) return NULL;
  if (rc != PSP_OK) {
    PyErr_SetString(PyExc_RuntimeError, "unknown error in Jacobi iteration"); /* :74 */
    return NULL;
  }
  Py_RETURN_NONE;
}

static void Jacobi_dealloc(JacobiObject *self) {
  if (self->op) psp_op_destroy(self->op);
  if (self->dev) psp_jacobi_destroy(self->dev);
  if (self->have_aref) pyop_release(&self->aref);
  Py_XDECREF(self->matrix);
  PyObject_Del(self);
}

static PyObject *Jacobi_get_shape(JacobiObject *self, void *c) {
  return Py_BuildValue("(i,i)", self->n, self->n); /* :245-266 */
}

static PyObject *Jacobi_get_psp_op(JacobiObject *self, void *c) {
  if (self->op == NULL) {
    int rc = psp_op_from_jacobi(self->dev, &self->op);
    if (rc != PSP_OK) return raise_psp(rc);
  }
  return PyCapsule_New(self->op, PSP_OP_CAPSULE_NAME, NULL);
}

static PyMethodDef Jacobi_methods[] = {
    {"precon", (PyCFunction)Jacobi_precon, METH_VARARGS,
     "self.precon(x, y)\n\napply preconditioner self on x, store result in y. x is unchanged."},
    {NULL, NULL, 0, NULL}};

static PyGetSetDef Jacobi_getset[] = {{"shape", (getter)Jacobi_get_shape, NULL, "(n, n)", NULL},
                                      {"_psp_op", (getter)Jacobi_get_psp_op, NULL, "device operator", NULL},
                                      {NULL, NULL, NULL, NULL, NULL}};

/* ------------------------------------------------------------------------------ ssor */

typedef struct {
  PyObject_VAR_HEAD
  int n;
  PyObject *matrix; /* the sss_mat: the device handle borrows it */
  double omega;
  int steps;
  psp_ssor_t *dev;
  psp_op_t *op;
} SSORObject;

static PyTypeObject SSORType;

/* ssor(A, omega=1.0, steps=1): preconmodule.c:487-510 -- A must be an sss_mat ("O!") */
static PyObject *ssor_prec(PyObject *self, PyObject *args) {
  PyObject *matrix;
  double omega = 1.0;
  int steps = 1, n, rc;
  SSORObject *op;
  if (!PyArg_ParseTuple(args, "O!|di", &SSSMatType, &matrix, &omega, &steps)) return NULL;
  if (SpMatrix_GetOrder(matrix, &n)) return NULL; /* :421 */
  op = PyObject_New(SSORObject, &SSORType);
  if (op == NULL) return PyErr_NoMemory();
  op->n = n;
  op->matrix = NULL;
  op->omega = omega;
  op->steps = steps;
  op->dev = NULL;
  op->op = NULL;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_ssor_create(((SSSMatObject *)matrix)->dev, omega, steps, &op->dev);
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) {
    Py_DECREF(op);
    return raise_psp(rc);
  }
  Py_INCREF(matrix);
  op->matrix = matrix;
  return (PyObject *)op;
}

/* self.precon(x, y): SSOR_precon, preconmodule.c:199-223 -- contiguous arrays only */
static PyObject *SSOR_precon(SSORObject *self, PyObject *args) {
  double *x, *y;
  int rc;
  if (SpMatrix_ParseVecOpArgs(args, &x, &y, self->n)) return NULL;
  Py_BEGIN_ALLOW_THREADS
  rc = psp_ssor_precon(self->dev, x, y);
  Py_END_ALLOW_THREADS
  if (rc != PSP_OK) return raise_psp(rc);
  Py_RETURN_NONE;
}

static void SSOR_dealloc(SSORObject *self) {
  if (self->op) psp_op_destroy(self->op);
  if (self->dev) psp_ssor_destroy(self->dev);
  Py_XDECREF(self->matrix);
  PyObject_Del(self);
}

static PyObject *SSOR_get_shape(SSORObject *self, void *c) {
  return Py_BuildValue("(i,i)", self->n, self->n); /* :297-318 */
}

static PyObject *SSOR_get_psp_op(SSORObject *self, void *c) {
  if (self->op == NULL) {
    int rc = psp_op_from_ssor(self->dev, &self->op);
    if (rc != PSP_OK) return raise_psp(rc);
  }
  return PyCapsule_New(self->op, PSP_OP_CAPSULE_NAME, NULL);
}

static PyMethodDef SSOR_methods[] = {
    {"precon", (PyCFunction)SSOR_precon, METH_VARARGS,
     "self.precon(x, y)\n\napply preconditioner self on x, store result in y. x is unchanged."},
    {NULL, NULL, 0, NULL}};

static PyGetSetDef SSOR_getset[] = {{"shape", (getter)SSOR_get_shape, NULL, "(n, n)", NULL},
                                    {"_psp_op", (getter)SSOR_get_psp_op, NULL, "device operator", NULL},
                                    {NULL, NULL, NULL, NULL, NULL}};

static PyMethodDef precon_methods[] = {
    {"jacobi", (PyCFunction)jacobi_prec, METH_VARARGS | METH_KEYWORDS,
     "jacobi(A, omega=1.0, steps=1)\n\nnew Jacobi preconditioner object"},
    {"ssor", (PyCFunction)ssor_prec, METH_VARARGS,
     "ssor(A, omega, steps) -- return SSOR preconditioner object\n\n"
     "This preconditioner executes 'steps' SSOR steps with a zero initial guess.\n\n"
     "A      'sss_mat' object, symmetric sparse matrix\n"
     "omega  relaxation parameter (default value: 1.0)\n"
     "steps  number of SSOR steps"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef precon_module = {PyModuleDef_HEAD_INIT, "precon",
                                           "preconditioners on MI355X", -1, precon_methods,
                                           NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit_precon(void) {
  PyObject *m;
  import_array();
  if (import_spmatrix() < 0) return NULL;
  {
    PyTypeObject zero = {PyVarObject_HEAD_INIT(NULL, 0)};
    JacobiType = zero;
  }
  JacobiType.tp_name = "pysparse_amd.precon.precon.jacobi";
  JacobiType.tp_basicsize = sizeof(JacobiObject);
  JacobiType.tp_dealloc = (destructor)Jacobi_dealloc;
  JacobiType.tp_flags = Py_TPFLAGS_DEFAULT;
  JacobiType.tp_methods = Jacobi_methods;
  JacobiType.tp_getset = Jacobi_getset;
  if (PyType_Ready(&JacobiType) < 0) return NULL;
  {
    PyTypeObject zero = {PyVarObject_HEAD_INIT(NULL, 0)};
    SSORType = zero;
  }
  SSORType.tp_name = "pysparse_amd.precon.precon.ssor";
  SSORType.tp_basicsize = sizeof(SSORObject);
  SSORType.tp_dealloc = (destructor)SSOR_dealloc;
  SSORType.tp_flags = Py_TPFLAGS_DEFAULT;
  SSORType.tp_methods = SSOR_methods;
  SSORType.tp_getset = SSOR_getset;
  if (PyType_Ready(&SSORType) < 0) return NULL;
  m = PyModule_Create(&precon_module);
  if (m == NULL) return NULL;
  Py_INCREF(&JacobiType);
  PyModule_AddObject(m, "JacobiType", (PyObject *)&JacobiType);
  Py_INCREF(&SSORType);
  PyModule_AddObject(m, "SSORType", (PyObject *)&SSORType);
  return m;
}
