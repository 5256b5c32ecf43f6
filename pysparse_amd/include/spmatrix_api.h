/*
 * spmatrix_api.h -- inter-module C API of pysparse_amd.sparse.spmatrix (Python 3).
 *
 * Same slot numbers and prototypes as the reference's table
 * (pysparse/include/spmatrix_api.h:10-72), exported as a PyCapsule named
 * "pysparse_amd.sparse.spmatrix._C_API" instead of the Python-2 PyCObject (:114).
 * Object struct layouts start with the reference's fields in the reference's order
 * (ll_mat.h:6-18, csr_mat.h:6-13, sss_mat.h:6-14) so that C consumers written against
 * them keep working; the device handles are appended after them.
 */
#ifndef PSP_SPMATRIX_API_H
#define PSP_SPMATRIX_API_H

#include <Python.h>

#include "pysparse_hip.h"

typedef struct {
  PyObject_VAR_HEAD
  int dim[2];     /* array dimension */
  int issym;      /* non-zero, if obj represents a symmetric matrix */
  int storeZeros; /* whether to store zero values */
  int nnz;        /* number of stored items */
  int nalloc;     /* allocated size of value and index arrays */
  int free;       /* index to first element in free chain */
  double *val;    /* values */
  int *col;       /* column indices */
  int *link;      /* next entry of the same row, -1 terminates */
  int *root;      /* first entry of each row, -1 when empty */
  /* --- additions --- */
  psp_csr_t *mirror; /* device CSR of the current contents (NULL when stale) */
  psp_op_t *op;      /* operator handle over the mirror */
} LLMatObject;

struct llColIndex {
  int *root; /* first element of each column */
  int *row;  /* row index of each element */
  int *link; /* next element in the column */
  int nzLo, nzDiag, nzUp;
};

typedef struct {
  PyObject_VAR_HEAD
  int dim[2];
  int nnz;
  double *val; /* host copies: may be NULL for matrices generated on the device */
  int *col;
  int *ind;
  /* --- additions --- */
  psp_csr_t *dev;
  psp_op_t *op;
} CSRMatObject;

typedef struct {
  PyObject_VAR_HEAD
  int n;
  int nnz; /* strict-lower count */
  double *val;
  double *diag;
  int *col;
  int *ind;
  /* --- additions --- */
  psp_sss_t *dev;
  psp_op_t *op;
} SSSMatObject;

#define LLMatType_NUM 0
#define CSRMatType_NUM 1
#define SSSMatType_NUM 2
#define SpMatrix_ParseVecOpArgs_NUM 3
#define SpMatrix_GetShape_NUM 4
#define SpMatrix_GetOrder_NUM 5
#define SpMatrix_GetItem_NUM 6
#define SpMatrix_Matvec_NUM 7
#define SpMatrix_Precon_NUM 8
#define SpMatrix_NewLLMatObject_NUM 9
#define SpMatrix_LLMatGetItem_NUM 10
#define SpMatrix_LLMatSetItem_NUM 11
#define SpMatrix_LLMatUpdateItemAdd_NUM 12
#define SpMatrix_LLMatBuildColIndex_NUM 13
#define SpMatrix_LLMatDestroyColIndex_NUM 14
#define ItSolvers_Solve_NUM 15
#define SpMatrix_API_pointers 16

#define SPMATRIX_CAPSULE_NAME "pysparse_amd.sparse.spmatrix._C_API"
/* capsule returned by the private `_psp_op` attribute of native operators */
#define PSP_OP_CAPSULE_NAME "psp_op_t"

#ifndef SPMATRIX_MODULE
/* consumers: static void **SpMatrix_API; call import_spmatrix() in module init */
static void **SpMatrix_API;

#define LLMatType (*(PyTypeObject *)SpMatrix_API[LLMatType_NUM])
#define CSRMatType (*(PyTypeObject *)SpMatrix_API[CSRMatType_NUM])
#define SSSMatType (*(PyTypeObject *)SpMatrix_API[SSSMatType_NUM])
#define SpMatrix_ParseVecOpArgs \
  (*(int (*)(PyObject *, double **, double **, int))SpMatrix_API[SpMatrix_ParseVecOpArgs_NUM])
#define SpMatrix_GetShape (*(int (*)(PyObject *, int[]))SpMatrix_API[SpMatrix_GetShape_NUM])
#define SpMatrix_GetOrder (*(int (*)(PyObject *, int *))SpMatrix_API[SpMatrix_GetOrder_NUM])
#define SpMatrix_GetItem (*(double (*)(PyObject *, int, int))SpMatrix_API[SpMatrix_GetItem_NUM])
#define SpMatrix_Matvec \
  (*(int (*)(PyObject *, int, double *, int, double *))SpMatrix_API[SpMatrix_Matvec_NUM])
#define SpMatrix_Precon \
  (*(int (*)(PyObject *, int, double *, double *))SpMatrix_API[SpMatrix_Precon_NUM])
#define SpMatrix_NewLLMatObject \
  (*(PyObject * (*)(int[], int, int, int)) SpMatrix_API[SpMatrix_NewLLMatObject_NUM])
#define SpMatrix_LLMatGetItem \
  (*(double (*)(LLMatObject *, int, int))SpMatrix_API[SpMatrix_LLMatGetItem_NUM])
#define SpMatrix_LLMatSetItem \
  (*(int (*)(LLMatObject *, int, int, double))SpMatrix_API[SpMatrix_LLMatSetItem_NUM])
#define SpMatrix_LLMatUpdateItemAdd \
  (*(int (*)(LLMatObject *, int, int, double))SpMatrix_API[SpMatrix_LLMatUpdateItemAdd_NUM])
#define SpMatrix_LLMatBuildColIndex                       \
  (*(int (*)(struct llColIndex **, LLMatObject *, int)) \
       SpMatrix_API[SpMatrix_LLMatBuildColIndex_NUM])
#define SpMatrix_LLMatDestroyColIndex \
  (*(void (*)(struct llColIndex **))SpMatrix_API[SpMatrix_LLMatDestroyColIndex_NUM])
#define ItSolvers_Solve                                                                      \
  (*(int (*)(PyObject *, PyObject *, int, double *, double *, double, int, PyObject *, int *, \
             int *, double *))SpMatrix_API[ItSolvers_Solve_NUM])

static int import_spmatrix(void) {
  PyObject *m = PyImport_ImportModule("pysparse_amd.sparse.spmatrix");
  if (m == NULL) return -1;
  Py_DECREF(m);
  SpMatrix_API = (void **)PyCapsule_Import(SPMATRIX_CAPSULE_NAME, 0);
  return SpMatrix_API ? 0 : -1;
}
#endif /* !SPMATRIX_MODULE */

#endif
