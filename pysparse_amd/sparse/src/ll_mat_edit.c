/* ll_mat_edit.c -- the EDITING half of ll_mat: everything a script does to the feeder matrix before it hands it to the
 * SpMV / Krylov path (sub-matrix read / write, copy, shift, scale, norms, deletion of rows / columns, compress, coordinate
 * export, MatrixMarket export, the FEM assembly updates, matrix * matrix).  Host code only; textually included by
 * spmatrixmodule.c (one translation unit: the storage primitives ll_store / SpMatrix_LLMat{Get,Set,UpdateAdd}Item,
 * the column index and ll_invalidate are its statics).
 *
 * Reference: pysparse/sparse/src/ll_mat.c -- the citations at each function.  The storage is the reference's
 * (root / link / col / val lists, rows sorted by column, free list), so every method below is a walk over those lists;
 * summation orders (shift, matrixmultiply, dot, symdot) are the reference's, entry by entry, hence the same bits.
 * Where the reference's sub-matrix code contradicts its own intent (ll_mat.c:660-700 tests `col % step` instead of
 * `(col - start) % step`; :1036-1038 and :686-688 overwrite the row counter when they mirror an entry; :1097-1101 mirrors
 * a symmetric right-hand side into the WRONG block of a general matrix) the evident intent is implemented and
 * tests/test_ll_mat_edit.py pins it against dense NumPy.
 *
 * Every method that changes entries drops the device mirror first (ll_invalidate): the next matvec rebuilds it. */

/* ------------------------------------------------------------------ small helpers */

static PyArrayObject *ll_vec_arg(PyObject *o, npy_intp want, const char *what) {
  PyArrayObject *v = (PyArrayObject *)PyArray_FROM_OTF(o, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY);
  if (v == NULL) {
    PyErr_Format(SpMatrix_ErrorObject, "Supply %s as input.", what);
    return NULL;
  }
  if (PyArray_NDIM(v) != 1 || PyArray_DIM(v, 0) != want) {
    PyErr_Format(SpMatrix_ErrorObject, "%s has wrong dimension.", what);
    Py_DECREF(v);
    return NULL;
  }
  return v;
}

/* mask argument of the delete_* methods: 1-D integer (or bool) array of the given length (ll_mat.c:2771-2775 insists on
 * dtype 'l'; any integer type is taken here) */
static PyArrayObject *ll_mask_arg(PyObject *o, npy_intp want) {
  PyArrayObject *m;
  if (!PyArray_Check(o) || PyArray_NDIM((PyArrayObject *)o) != 1 || PyArray_DIM((PyArrayObject *)o, 0) != want ||
      !(PyArray_ISINTEGER((PyArrayObject *)o) || PyArray_ISBOOL((PyArrayObject *)o))) {
    PyErr_SetString(PyExc_ValueError, "mask must be a 1D integer NumPy array of appropriate length");
    return NULL;
  }
  m = (PyArrayObject *)PyArray_FROM_OTF(o, NPY_LONG, NPY_ARRAY_IN_ARRAY);
  return m;
}

/* ------------------------------------------------------------------ whole-matrix methods */

/* ll_mat.c:1713-1735: symmetric -> general storage in place (the mirrored entries are inserted) */
static PyObject *LLMat_generalize(LLMatObject *self, PyObject *args) {
  int i, k;
  if (!PyArg_ParseTuple(args, "")) return NULL;
  if (self->issym) {
    ll_invalidate(self);
    self->issym = 0;
    for (i = 0; i < self->dim[0]; i++)
      /* an insertion into row j < i never touches row i's list; the arrays may move (ll_grow): indices, not pointers */
      for (k = self->root[i]; k != -1; k = self->link[k]) {
        const int j = self->col[k];
        if (i > j && SpMatrix_LLMatSetItem(self, j, i, self->val[k])) return NULL;
      }
  }
  Py_RETURN_NONE;
}

/* ll_mat.c:463-516: the entries packed to the front of the arrays (row by row here), arrays shrunk to nnz */
static int ll_compress(LLMatObject *self, int *freed) {
  const int nnz = self->nnz, cap = nnz > 0 ? nnz : 1;
  double *val = PyMem_New(double, cap);
  int *col = PyMem_New(int, cap), *link = PyMem_New(int, cap);
  int i, k, t = 0;
  if (!val || !col || !link) {
    PyMem_Del(val);
    PyMem_Del(col);
    PyMem_Del(link);
    PyErr_NoMemory();
    return -1;
  }
  for (i = 0; i < self->dim[0]; i++) {
    int last = -1;
    for (k = self->root[i]; k != -1; k = self->link[k]) {
      val[t] = self->val[k];
      col[t] = self->col[k];
      link[t] = -1;
      if (last == -1)
        self->root[i] = t;
      else
        link[last] = t;
      last = t++;
    }
  }
  PyMem_Del(self->val);
  PyMem_Del(self->col);
  PyMem_Del(self->link);
  self->val = val;
  self->col = col;
  self->link = link;
  self->free = -1;
  *freed = self->nalloc - cap; /* (one slot stays allocated for an empty matrix) */
  self->nalloc = cap;
  return 0;
}

static PyObject *LLMat_compress(LLMatObject *self, PyObject *args) {
  int freed;
  if (!PyArg_ParseTuple(args, "")) return NULL;
  if (ll_compress(self, &freed)) return NULL;
  return PyLong_FromLong(freed < 0 ? 0 : freed);
}

/* ll_mat.c:1757-1810: coordinate real general / symmetric, one-based, the STORED entries row by row */
static PyObject *LLMat_export_mtx(LLMatObject *self, PyObject *args) {
  const char *name;
  int precision = 16, i, k;
  FILE *f;
  if (!PyArg_ParseTuple(args, "s|i", &name, &precision)) return NULL;
  if (precision < 1) precision = 1;
  if (!(f = fopen(name, "w"))) return PyErr_SetFromErrnoWithFilename(PyExc_IOError, name);
  if (fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n%% file created by pysparse module\n%d %d %d\n",
              self->issym ? "symmetric" : "general", self->dim[0], self->dim[1], self->nnz) < 0)
    goto ioerr;
  for (i = 0; i < self->dim[0]; i++)
    for (k = self->root[i]; k != -1; k = self->link[k])
      if (fprintf(f, "%d %d %.*e\n", i + 1, self->col[k] + 1, precision - 1, self->val[k]) < 0) goto ioerr;
  if (fclose(f)) return PyErr_SetFromErrnoWithFilename(PyExc_IOError, name);
  Py_RETURN_NONE;
ioerr:
  fclose(f);
  PyErr_SetString(PyExc_IOError, "Error writing matrix data");
  return NULL;
}

/* ll_mat.c:1817-1839 */
static PyObject *ll_copy(LLMatObject *self) {
  LLMatObject *c = (LLMatObject *)SpMatrix_NewLLMatObject(self->dim, self->issym, self->nnz, self->storeZeros);
  int i, k;
  if (c == NULL) return NULL;
  for (i = 0; i < self->dim[0]; i++)
    for (k = self->root[i]; k != -1; k = self->link[k])
      if (SpMatrix_LLMatSetItem(c, i, self->col[k], self->val[k]) == -1) {
        Py_DECREF(c);
        return NULL;
      }
  return (PyObject *)c;
}

static PyObject *LLMat_copy(LLMatObject *self, PyObject *args) {
  if (!PyArg_ParseTuple(args, "")) return NULL;
  return ll_copy(self);
}

/* ll_mat.c:1916-1976: '1' and 'inf' for general storage only (NotImplementedError otherwise), 'fro' for both */
static PyObject *LLMat_norm(LLMatObject *self, PyObject *args) {
  const char *p;
  double norm = 0.0, s;
  int i, k;
  if (!PyArg_ParseTuple(args, "s", &p)) return NULL;
  if (strcmp(p, "1") == 0 || strcmp(p, "inf") == 0) {
    if (self->issym) {
      PyErr_SetString(PyExc_NotImplementedError, "Not implemented for symmetric matrices");
      return NULL;
    }
    if (p[0] == '1') {
      struct llColIndex *ci;
      if (SpMatrix_LLMatBuildColIndex(&ci, self, 1)) return NULL;
      for (i = 0; i < self->dim[1]; i++) {
        for (s = 0.0, k = ci->root[i]; k != -1; k = ci->link[k]) s += fabs(self->val[k]);
        norm = s > norm ? s : norm;
      }
      SpMatrix_LLMatDestroyColIndex(&ci);
    } else {
      for (i = 0; i < self->dim[0]; i++) {
        for (s = 0.0, k = self->root[i]; k != -1; k = self->link[k]) s += fabs(self->val[k]);
        norm = s > norm ? s : norm;
      }
    }
  } else if (strcmp(p, "fro") == 0) {
    for (i = 0; i < self->dim[0]; i++)
      for (k = self->root[i]; k != -1; k = self->link[k]) {
        const double v = self->val[k];
        norm += v * v;
        if (self->issym && self->col[k] != i) norm += v * v;
      }
    norm = sqrt(norm);
  } else {
    PyErr_SetString(PyExc_ValueError, "unknown norm type");
    return NULL;
  }
  return PyFloat_FromDouble(norm);
}

/* ll_mat.c:1984-2031: A += sigma * B, entry by entry in B's storage order */
static PyObject *LLMat_shift(LLMatObject *self, PyObject *args) {
  LLMatObject *B;
  PyObject *held = NULL;
  double sigma;
  int i, k;
  if (!PyArg_ParseTuple(args, "dO!", &sigma, &LLMatType, &B)) return NULL;
  if (self->dim[0] != B->dim[0] || self->dim[1] != B->dim[1]) {
    PyErr_SetString(PyExc_ValueError, "matrix shapes do not match");
    return NULL;
  }
  if (self->issym && !B->issym) {
    PyErr_SetString(PyExc_NotImplementedError, "Cannot shift symmetric matrix by non-symmetric matrix.");
    return NULL;
  }
  if (B == self) { /* A.shift(s, A): walk a copy, the lists change under the updates */
    if ((held = ll_copy(self)) == NULL) return NULL;
    B = (LLMatObject *)held;
  }
  for (i = 0; i < B->dim[0]; i++)
    for (k = B->root[i]; k != -1; k = B->link[k]) {
      const int j = B->col[k];
      const double v = sigma * B->val[k];
      if (SpMatrix_LLMatUpdateItemAdd(self, i, j, v) == -1) goto fail;
      if (B->issym && !self->issym && i != j && SpMatrix_LLMatUpdateItemAdd(self, j, i, v) == -1) goto fail;
    }
  Py_XDECREF(held);
  Py_RETURN_NONE;
fail:
  Py_XDECREF(held);
  return NULL;
}

/* ll_mat.c:2174-2189 */
static PyObject *LLMat_scale(LLMatObject *self, PyObject *args) {
  double sigma;
  int i, k;
  if (!PyArg_ParseTuple(args, "d", &sigma)) return NULL;
  ll_invalidate(self);
  for (i = 0; i < self->dim[0]; i++)
    for (k = self->root[i]; k != -1; k = self->link[k]) self->val[k] *= sigma;
  Py_RETURN_NONE;
}

/* ll_mat.c:1530-1571 / :1473-1522: stored entries times v[row] / v[col] */
static PyObject *LLMat_row_scale(LLMatObject *self, PyObject *args) {
  PyObject *vin;
  PyArrayObject *v;
  int i, k;
  if (!PyArg_ParseTuple(args, "O", &vin)) return NULL;
  if ((v = ll_vec_arg(vin, self->dim[0], "Row scaling vector")) == NULL) return NULL;
  ll_invalidate(self);
  for (i = 0; i < self->dim[0]; i++)
    for (k = self->root[i]; k != -1; k = self->link[k]) self->val[k] *= ((double *)PyArray_DATA(v))[i];
  Py_DECREF(v);
  Py_RETURN_NONE;
}

static PyObject *LLMat_col_scale(LLMatObject *self, PyObject *args) {
  PyObject *vin;
  PyArrayObject *v;
  int i, k;
  if (!PyArg_ParseTuple(args, "O", &vin)) return NULL;
  if ((v = ll_vec_arg(vin, self->dim[1], "Column scaling vector")) == NULL) return NULL;
  ll_invalidate(self);
  for (i = 0; i < self->dim[0]; i++)
    for (k = self->root[i]; k != -1; k = self->link[k]) self->val[k] *= ((double *)PyArray_DATA(v))[self->col[k]];
  Py_DECREF(v);
  Py_RETURN_NONE;
}

/* ll_mat.c:2038-2061, :2112-2139, :2150-2168: row by row, ascending column; keys / values refuse symmetric storage */
static PyObject *ll_listing(LLMatObject *a, int what) {
  PyObject *list, *item;
  int i, k;
  Py_ssize_t pos = 0;
  if (a->issym && what != 2) {
    PyErr_Format(PyExc_NotImplementedError, "%s() doesn't yet support symmetric matrices", what ? "values" : "keys");
    return NULL;
  }
  if ((list = PyList_New(a->nnz)) == NULL) return NULL;
  for (i = 0; i < a->dim[0]; i++)
    for (k = a->root[i]; k != -1; k = a->link[k]) {
      item = what == 0   ? Py_BuildValue("ii", i, a->col[k])
             : what == 1 ? PyFloat_FromDouble(a->val[k])
                         : Py_BuildValue("((ii)d)", i, a->col[k], a->val[k]);
      if (item == NULL) {
        Py_DECREF(list);
        return NULL;
      }
      PyList_SET_ITEM(list, pos++, item);
    }
  return list;
}

static PyObject *LLMat_keys(LLMatObject *a, PyObject *args) {
  return PyArg_ParseTuple(args, "") ? ll_listing(a, 0) : NULL;
}
static PyObject *LLMat_values(LLMatObject *a, PyObject *args) {
  return PyArg_ParseTuple(args, "") ? ll_listing(a, 1) : NULL;
}
static PyObject *LLMat_items(LLMatObject *a, PyObject *args) {
  return PyArg_ParseTuple(args, "") ? ll_listing(a, 2) : NULL;
}

/* ll_mat.c:2999-3035: (val, irow, jcol) of the stored entries */
static PyObject *LLMat_find(LLMatObject *self, PyObject *args) {
  npy_intp d = self->nnz;
  PyObject *ar = PyArray_SimpleNew(1, &d, NPY_INT), *ac = PyArray_SimpleNew(1, &d, NPY_INT);
  PyObject *av = PyArray_SimpleNew(1, &d, NPY_DOUBLE);
  int i, k, t = 0;
  if (!ar || !ac || !av) {
    Py_XDECREF(ar);
    Py_XDECREF(ac);
    Py_XDECREF(av);
    return NULL;
  }
  for (i = 0; i < self->dim[0]; i++)
    for (k = self->root[i]; k != -1; k = self->link[k]) {
      ((int *)PyArray_DATA((PyArrayObject *)ar))[t] = i;
      ((int *)PyArray_DATA((PyArrayObject *)ac))[t] = self->col[k];
      ((double *)PyArray_DATA((PyArrayObject *)av))[t++] = self->val[k];
    }
  return Py_BuildValue("NNN", av, ar, ac);
}

/* ll_mat.c:2398-2487: b[i] = A[id1[i], id2[i]]; id1 defaults to 0..len-1, id2 to id1; b must be a contiguous float64
 * array to receive the values (the reference writes into a converted temporary otherwise -- here that is a TypeError) */
static PyObject *LLMat_take(LLMatObject *self, PyObject *args) {
  PyObject *bin, *id1in = NULL, *id2in = NULL, *ret = NULL;
  PyArrayObject *b, *id1 = NULL, *id2 = NULL;
  npy_intp len, i;
  if (!PyArg_ParseTuple(args, "O|OO", &bin, &id1in, &id2in)) return NULL;
  if (id1in == Py_None) id1in = NULL;
  if (id2in == Py_None) id2in = NULL;
  if (!PyArray_Check(bin) || PyArray_TYPE((PyArrayObject *)bin) != NPY_DOUBLE || PyArray_NDIM((PyArrayObject *)bin) != 1 ||
      !PyArray_ISCARRAY((PyArrayObject *)bin)) {
    PyErr_SetString(PyExc_TypeError, "b must be a contiguous 1-dimensional float64 array (it receives the values)");
    return NULL;
  }
  b = (PyArrayObject *)bin;
  len = PyArray_DIM(b, 0);
  if (id1in && !(id1 = (PyArrayObject *)PyArray_FROM_OTF(id1in, NPY_INTP, NPY_ARRAY_IN_ARRAY))) goto done;
  if (id2in && !(id2 = (PyArrayObject *)PyArray_FROM_OTF(id2in, NPY_INTP, NPY_ARRAY_IN_ARRAY))) goto done;
  if (id1 && (PyArray_NDIM(id1) != 1 || PyArray_DIM(id1, 0) != len)) {
    PyErr_SetString(PyExc_IndexError, "id1 does not have the same size as b");
    goto done;
  }
  if (id2 && (PyArray_NDIM(id2) != 1 || PyArray_DIM(id2, 0) != len)) {
    PyErr_SetString(PyExc_IndexError, "id2 does not have the same size as b");
    goto done;
  }
  for (i = 0; i < len; i++) {
    const npy_intp i1 = id1 ? ((npy_intp *)PyArray_DATA(id1))[i] : i;
    const npy_intp j1 = id2 ? ((npy_intp *)PyArray_DATA(id2))[i] : i1;
    const double v = SpMatrix_LLMatGetItem(self, (int)i1, (int)j1); /* mirrors a symmetric matrix itself */
    if (PyErr_Occurred()) goto done;
    ((double *)PyArray_DATA(b))[i] = v;
  }
  ret = Py_None;
  Py_INCREF(ret);
done:
  Py_XDECREF(id1);
  Py_XDECREF(id2);
  return ret;
}

/* ll_mat.c:2202-2291: FEM assembly, a[ind0[i], ind1[j]] += b[i, j] where mask0[i] and mask1[j]; general storage only.
 * The element of b is read at offset i + len0 * j of its C-contiguous data, as the reference does (:2262) -- for the
 * square element matrices this is called with, b transposed. */
static PyObject *LLMat_update_add_mask(LLMatObject *self, PyObject *args) {
  PyObject *bin, *i0in, *i1in, *m0in, *m1in, *ret = NULL;
  PyArrayObject *b = NULL, *i0 = NULL, *i1 = NULL, *m0 = NULL, *m1 = NULL;
  npy_intp len0, len1, i, j;
  if (self->issym) {
    PyErr_SetString(SpMatrix_ErrorObject, "Method not allowed for symmetric matrices");
    return NULL;
  }
  if (!PyArg_ParseTuple(args, "OOOOO", &bin, &i0in, &i1in, &m0in, &m1in)) return NULL;
  b = (PyArrayObject *)PyArray_FROM_OTF(bin, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY);
  i0 = (PyArrayObject *)PyArray_FROM_OTF(i0in, NPY_LONG, NPY_ARRAY_IN_ARRAY);
  i1 = (PyArrayObject *)PyArray_FROM_OTF(i1in, NPY_LONG, NPY_ARRAY_IN_ARRAY);
  m0 = (PyArrayObject *)PyArray_FROM_OTF(m0in, NPY_LONG, NPY_ARRAY_IN_ARRAY);
  m1 = (PyArrayObject *)PyArray_FROM_OTF(m1in, NPY_LONG, NPY_ARRAY_IN_ARRAY);
  if (!b || !i0 || !i1 || !m0 || !m1) goto done;
  if (PyArray_NDIM(b) != 2 || PyArray_NDIM(i0) != 1 || PyArray_NDIM(i1) != 1 || PyArray_NDIM(m0) != 1 ||
      PyArray_NDIM(m1) != 1) {
    PyErr_SetString(PyExc_ValueError, "b must be 2-dimensional, the index and mask arrays 1-dimensional");
    goto done;
  }
  len0 = PyArray_DIM(i0, 0);
  len1 = PyArray_DIM(i1, 0);
  if (PyArray_DIM(m0, 0) != len0 || PyArray_DIM(m1, 0) != len1) {
    PyErr_SetString(PyExc_ValueError, "Shapes of index and mask array do not match");
    goto done;
  }
  if (PyArray_DIM(b, 0) != len0 || PyArray_DIM(b, 1) != len1) {
    PyErr_SetString(PyExc_ValueError, "Shapes of input matrix and index arrays do not match");
    goto done;
  }
  for (i = 0; i < len0; i++) {
    long r;
    if (!((long *)PyArray_DATA(m0))[i]) continue;
    r = ((long *)PyArray_DATA(i0))[i];
    if (r < 0) r += self->dim[0];
    if (r < 0 || r >= self->dim[0]) {
      PyErr_SetString(PyExc_IndexError, "element of arg 2 out of range");
      goto done;
    }
    for (j = 0; j < len1; j++) {
      long c;
      if (!((long *)PyArray_DATA(m1))[j]) continue;
      c = ((long *)PyArray_DATA(i1))[j];
      if (c < 0) c += self->dim[1];
      if (c < 0 || c >= self->dim[1]) {
        PyErr_SetString(PyExc_IndexError, "element of arg 3 out of range");
        goto done;
      }
      if (SpMatrix_LLMatUpdateItemAdd(self, (int)r, (int)c, ((double *)PyArray_DATA(b))[i + len0 * j]) == -1) goto done;
    }
  }
  ret = Py_None;
  Py_INCREF(ret);
done:
  Py_XDECREF(b);
  Py_XDECREF(i0);
  Py_XDECREF(i1);
  Py_XDECREF(m0);
  Py_XDECREF(m1);
  return ret;
}

/* ll_mat.c:2306-2391: the symmetric assembly, pairs j <= i of the masked positions; a symmetric matrix receives the
 * entry in its lower triangle, a general one at (i1, j1) and, off the diagonal, at (j1, i1) */
static PyObject *LLMat_update_add_mask_sym(LLMatObject *self, PyObject *args) {
  PyObject *bin, *iin, *min, *ret = NULL;
  PyArrayObject *b = NULL, *ind = NULL, *mask = NULL;
  npy_intp len, i, j;
  if (!PyArg_ParseTuple(args, "OOO", &bin, &iin, &min)) return NULL;
  b = (PyArrayObject *)PyArray_FROM_OTF(bin, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY);
  ind = (PyArrayObject *)PyArray_FROM_OTF(iin, NPY_LONG, NPY_ARRAY_IN_ARRAY);
  mask = (PyArrayObject *)PyArray_FROM_OTF(min, NPY_LONG, NPY_ARRAY_IN_ARRAY);
  if (!b || !ind || !mask) goto done;
  if (PyArray_NDIM(b) != 2 || PyArray_NDIM(ind) != 1 || PyArray_NDIM(mask) != 1) {
    PyErr_SetString(PyExc_ValueError, "b must be 2-dimensional, the index and mask arrays 1-dimensional");
    goto done;
  }
  len = PyArray_DIM(ind, 0);
  if (PyArray_DIM(mask, 0) != len) {
    PyErr_SetString(PyExc_ValueError, "Shapes of index and mask array do not match");
    goto done;
  }
  if (PyArray_DIM(b, 0) != len || PyArray_DIM(b, 1) != len) {
    PyErr_SetString(PyExc_ValueError, "Shapes of input matrix and index arrays do not match");
    goto done;
  }
  for (i = 0; i < len; i++) {
    long r;
    if (!((long *)PyArray_DATA(mask))[i]) continue;
    r = ((long *)PyArray_DATA(ind))[i];
    if (r < 0) r += self->dim[0];
    if (r < 0 || r >= self->dim[0]) {
      PyErr_SetString(PyExc_IndexError, "element of arg 2 out of range");
      goto done;
    }
    for (j = 0; j <= i; j++) {
      long c;
      double v;
      if (!((long *)PyArray_DATA(mask))[j]) continue;
      c = ((long *)PyArray_DATA(ind))[j]; /* j <= i and masked: checked when it was i */
      if (c < 0) c += self->dim[1];
      v = ((double *)PyArray_DATA(b))[i + len * j];
      if (self->issym) {
        if (SpMatrix_LLMatUpdateItemAdd(self, (int)(r > c ? r : c), (int)(r > c ? c : r), v) == -1) goto done;
      } else {
        if (SpMatrix_LLMatUpdateItemAdd(self, (int)r, (int)c, v) == -1) goto done;
        if (r != c && SpMatrix_LLMatUpdateItemAdd(self, (int)c, (int)r, v) == -1) goto done;
      }
    }
  }
  ret = Py_None;
  Py_INCREF(ret);
done:
  Py_XDECREF(b);
  Py_XDECREF(ind);
  Py_XDECREF(mask);
  return ret;
}

/* ------------------------------------------------------------------ deletion of rows / columns (ll_mat.c:2764-2990) */

/* rows whose mask is 0 go to the free list, the kept ones move up */
static void ll_drop_rows(LLMatObject *self, const long *mask) {
  int row, kept = 0;
  for (row = 0; row < self->dim[0]; row++) {
    if (mask[row]) {
      self->root[kept++] = self->root[row];
    } else {
      int k = self->root[row];
      while (k != -1) {
        const int next = self->link[k];
        self->link[k] = self->free;
        self->free = k;
        self->nnz--;
        k = next;
      }
    }
  }
  self->dim[0] = kept;
}

/* entries in columns whose mask is 0 go to the free list, the others get their new column number */
static int ll_drop_cols(LLMatObject *self, const long *mask) {
  const int ncol = self->dim[1];
  int *newcol = (int *)malloc(sizeof(int) * (size_t)(ncol > 0 ? ncol : 1));
  int c, kept = 0, row;
  if (newcol == NULL) {
    PyErr_NoMemory();
    return -1;
  }
  for (c = 0; c < ncol; c++) newcol[c] = mask[c] ? kept++ : -1;
  for (row = 0; row < self->dim[0]; row++) {
    int last = -1, k = self->root[row];
    while (k != -1) {
      const int next = self->link[k];
      if (newcol[self->col[k]] >= 0) {
        self->col[k] = newcol[self->col[k]];
        last = k;
      } else {
        if (last == -1)
          self->root[row] = next;
        else
          self->link[last] = next;
        self->link[k] = self->free;
        self->free = k;
        self->nnz--;
      }
      k = next;
    }
  }
  self->dim[1] = kept;
  free(newcol);
  return 0;
}

static PyObject *ll_delete(LLMatObject *self, PyObject *args, int rows, int cols) {
  PyObject *min;
  PyArrayObject *mask;
  int rc = 0;
  if (!PyArg_ParseTuple(args, "O", &min)) return NULL;
  if ((mask = ll_mask_arg(min, rows ? self->dim[0] : self->dim[1])) == NULL) return NULL;
  if (rows && cols) {
    if (self->dim[0] != self->dim[1]) {
      PyErr_SetString(SpMatrix_ErrorObject, "method only allowed for square matrices");
      Py_DECREF(mask);
      return NULL;
    }
  } else if (self->issym) {
    PyErr_SetString(SpMatrix_ErrorObject, "method not allowed for symmetric matrices");
    Py_DECREF(mask);
    return NULL;
  }
  ll_invalidate(self);
  if (cols) rc = ll_drop_cols(self, (const long *)PyArray_DATA(mask)); /* before the rows: the mask indexes old columns */
  if (rc == 0 && rows) ll_drop_rows(self, (const long *)PyArray_DATA(mask));
  Py_DECREF(mask);
  if (rc) return NULL;
  Py_RETURN_NONE;
}

static PyObject *LLMat_delete_rows(LLMatObject *self, PyObject *args) { return ll_delete(self, args, 1, 0); }
static PyObject *LLMat_delete_cols(LLMatObject *self, PyObject *args) { return ll_delete(self, args, 0, 1); }
static PyObject *LLMat_delete_rowcols(LLMatObject *self, PyObject *args) { return ll_delete(self, args, 1, 1); }

/* ------------------------------------------------------------------ sub-matrices: A[I, J] and A[I, J] = value
 *
 * One subscript position: an integer (negative counts from the end), a slice, a list of integers or a 1-D integer array
 * (ll_mat.c:525-598).  List / array elements are taken as they are (out of range -> IndexError). */

typedef struct {
  long *idx;
  long n;
  int is_int, is_slice;
  Py_ssize_t start, step; /* of a slice */
} LLIndexSet;

static void ll_index_free(LLIndexSet *s) {
  free(s->idx);
  s->idx = NULL;
}

static int ll_index_set(PyObject *o, int dim, LLIndexSet *s) {
  long i;
  memset(s, 0, sizeof(*s));
  if (PyArray_Check(o) && PyArray_NDIM((PyArrayObject *)o) >= 1) {
    PyArrayObject *a;
    if (PyArray_NDIM((PyArrayObject *)o) != 1 || !(PyArray_ISINTEGER((PyArrayObject *)o))) {
      PyErr_SetString(PyExc_IndexError, "an index array must be 1-dimensional and of integer type");
      return -1;
    }
    if ((a = (PyArrayObject *)PyArray_FROM_OTF(o, NPY_LONG, NPY_ARRAY_IN_ARRAY)) == NULL) return -1;
    s->n = (long)PyArray_DIM(a, 0);
    s->idx = (long *)malloc(sizeof(long) * (size_t)(s->n > 0 ? s->n : 1));
    if (s->idx) memcpy(s->idx, PyArray_DATA(a), sizeof(long) * (size_t)s->n);
    Py_DECREF(a);
  } else if (PyIndex_Check(o)) {
    long v = PyLong_AsLong(o);
    if (v == -1 && PyErr_Occurred()) {
      PyObject *n = PyNumber_Index(o);
      PyErr_Clear();
      if (n == NULL) return -1;
      v = PyLong_AsLong(n);
      Py_DECREF(n);
      if (v == -1 && PyErr_Occurred()) return -1;
    }
    if (v < 0) v += dim; /* test/test_spmatrix.py:58-65 */
    if (v < 0 || v >= dim) {
      PyErr_SetString(PyExc_IndexError, "indices out of range");
      return -1;
    }
    s->n = 1;
    s->is_int = 1;
    if ((s->idx = (long *)malloc(sizeof(long)))) s->idx[0] = v;
  } else if (PySlice_Check(o)) {
    Py_ssize_t start, stop, step, len;
    if (PySlice_Unpack(o, &start, &stop, &step) < 0) return -1;
    len = PySlice_AdjustIndices(dim, &start, &stop, step);
    s->n = (long)len;
    s->is_slice = 1;
    s->start = start;
    s->step = step;
    if ((s->idx = (long *)malloc(sizeof(long) * (size_t)(len > 0 ? len : 1))))
      for (i = 0; i < len; i++) s->idx[i] = (long)(start + i * step);
  } else if (PyList_Check(o)) {
    s->n = (long)PyList_GET_SIZE(o);
    if ((s->idx = (long *)malloc(sizeof(long) * (size_t)(s->n > 0 ? s->n : 1))))
      for (i = 0; i < s->n; i++) {
        PyObject *e = PyList_GET_ITEM(o, i);
        if (!PyIndex_Check(e)) {
          PyErr_SetString(PyExc_ValueError, "Index must be a list of integers");
          ll_index_free(s);
          return -1;
        }
        s->idx[i] = PyLong_AsLong(e);
        if (s->idx[i] == -1 && PyErr_Occurred()) {
          ll_index_free(s);
          return -1;
        }
      }
  } else {
    PyErr_SetString(PyExc_TypeError, "Invalid index type");
    return -1;
  }
  if (s->idx == NULL) {
    PyErr_NoMemory();
    return -1;
  }
  if (!s->is_int && !s->is_slice)
    for (i = 0; i < s->n; i++)
      if (s->idx[i] < 0 || s->idx[i] >= dim) {
        PyErr_SetString(PyExc_IndexError, "indices out of range");
        ll_index_free(s);
        return -1;
      }
  return 0;
}

static int ll_parse_key(LLMatObject *self, PyObject *key, LLIndexSet *RI, LLIndexSet *CJ) {
  PyObject *k0, *k1;
  int rc;
  if (!PySequence_Check(key) || PyUnicode_Check(key)) {
    PyErr_SetString(PyExc_IndexError, "Index must be a sequence");
    return -1;
  }
  if (PySequence_Length(key) != 2) {
    PyErr_SetString(PyExc_IndexError, "There must be exactly two indices");
    return -1;
  }
  if ((k0 = PySequence_GetItem(key, 0)) == NULL) return -1;
  if ((k1 = PySequence_GetItem(key, 1)) == NULL) {
    Py_DECREF(k0);
    return -1;
  }
  rc = ll_index_set(k0, self->dim[0], RI);
  if (rc == 0 && (rc = ll_index_set(k1, self->dim[1], CJ)) != 0) ll_index_free(RI);
  Py_DECREF(k0);
  Py_DECREF(k1);
  return rc;
}

/* position of column c in CJ, or -1: by formula for a slice, through `pos` (dim[1] entries, -1 = absent) otherwise */
static long ll_col_pos(const LLIndexSet *CJ, const long *pos, long c) {
  if (CJ->is_slice) {
    const long d = c - (long)CJ->start;
    long q;
    if (CJ->step == 0 || d % (long)CJ->step != 0) return -1;
    q = d / (long)CJ->step;
    return (q >= 0 && q < CJ->n) ? q : -1;
  }
  return pos[c];
}

/* ll_mat.c:632-884: always a GENERAL matrix (:637); dst[i, j] = self[I[i], J[j]] */
static PyObject *ll_get_submatrix(LLMatObject *self, const LLIndexSet *RI, const LLIndexSet *CJ) {
  int dim[2];
  LLMatObject *dst;
  long *pos = NULL, i, j;
  int dup = 0, k;
  struct llColIndex *ci = NULL;
  double hint = (double)RI->n * (double)CJ->n;
  dim[0] = (int)RI->n;
  dim[1] = (int)CJ->n;
  if (hint > (double)self->nnz) hint = (double)self->nnz;
  dst = (LLMatObject *)SpMatrix_NewLLMatObject(dim, 0, hint < 1.0 ? 1 : (int)hint, self->storeZeros);
  if (dst == NULL) return NULL;
  if (!CJ->is_slice) {
    pos = (long *)malloc(sizeof(long) * (size_t)(self->dim[1] > 0 ? self->dim[1] : 1));
    if (pos == NULL) {
      Py_DECREF(dst);
      return PyErr_NoMemory();
    }
    for (j = 0; j < self->dim[1]; j++) pos[j] = -1;
    for (j = 0; j < CJ->n; j++) {
      if (pos[CJ->idx[j]] != -1) dup = 1;
      pos[CJ->idx[j]] = j;
    }
  }
  if (dup) { /* a column listed twice: element by element (ll_mat.c:607-630) */
    for (i = 0; i < RI->n; i++)
      for (j = 0; j < CJ->n; j++) {
        const double v = SpMatrix_LLMatGetItem(self, (int)RI->idx[i], (int)CJ->idx[j]);
        if ((v != 0.0 || PyErr_Occurred()) && (PyErr_Occurred() || SpMatrix_LLMatSetItem(dst, (int)i, (int)j, v))) goto fail;
      }
  } else {
    if (self->issym && SpMatrix_LLMatBuildColIndex(&ci, self, 0)) goto fail;
    for (i = 0; i < RI->n; i++) {
      const int row = (int)RI->idx[i];
      for (k = self->root[row]; k != -1; k = self->link[k])
        if ((j = ll_col_pos(CJ, pos, self->col[k])) >= 0 && SpMatrix_LLMatSetItem(dst, (int)i, (int)j, self->val[k])) goto fail;
      if (ci) /* the mirrored part of the row: stored entries (r, row) with r > row */
        for (k = ci->root[row]; k != -1; k = ci->link[k])
          if ((j = ll_col_pos(CJ, pos, ci->row[k])) >= 0 && SpMatrix_LLMatSetItem(dst, (int)i, (int)j, self->val[k])) goto fail;
    }
  }
  if (ci) SpMatrix_LLMatDestroyColIndex(&ci);
  free(pos);
  return (PyObject *)dst;
fail:
  if (ci) SpMatrix_LLMatDestroyColIndex(&ci);
  free(pos);
  Py_DECREF(dst);
  return NULL;
}

static PyObject *LLMat_subscript(LLMatObject *self, PyObject *key) {
  LLIndexSet RI, CJ;
  PyObject *ret;
  if (ll_parse_key(self, key, &RI, &CJ)) return NULL;
  if (RI.is_int && CJ.is_int) {
    const double v = SpMatrix_LLMatGetItem(self, (int)RI.idx[0], (int)CJ.idx[0]);
    ret = PyErr_Occurred() ? NULL : PyFloat_FromDouble(v);
  } else {
    ret = ll_get_submatrix(self, &RI, &CJ);
  }
  ll_index_free(&RI);
  ll_index_free(&CJ);
  return ret;
}

/* the stored entries that stand for the block (rows RI, columns CJ; both slices) go to the free list: the reference keeps
 * this step as clear_submatrix (ll_mat.c:890-921, no longer called there) and its own test expects a block to BE the
 * assigned matrix afterwards (test/test_spmatrix.py:97-103) */
static void ll_unlink_in(LLMatObject *self, int row, const LLIndexSet *cols, long below) {
  int last = -1, k = self->root[row];
  while (k != -1) {
    const int next = self->link[k];
    if (self->col[k] < below && ll_col_pos(cols, NULL, self->col[k]) >= 0) {
      if (last == -1)
        self->root[row] = next;
      else
        self->link[last] = next;
      self->link[k] = self->free;
      self->free = k;
      self->nnz--;
    } else {
      last = k;
    }
    k = next;
  }
}

static void ll_clear_block(LLMatObject *self, const LLIndexSet *RI, const LLIndexSet *CJ) {
  long i;
  ll_invalidate(self);
  for (i = 0; i < RI->n; i++) ll_unlink_in(self, (int)RI->idx[i], CJ, (long)self->dim[1]);
  if (self->issym) /* (r, c) with r < c lives at (c, r) */
    for (i = 0; i < CJ->n; i++) ll_unlink_in(self, (int)CJ->idx[i], RI, CJ->idx[i]);
}

/* one element of an assignment: a symmetric target takes it in its lower triangle when the value comes from a symmetric
 * matrix (its mirror image arrives too), otherwise writing above the diagonal is an IndexError (ll_mat.c:989-993) */
static int ll_assign_one(LLMatObject *self, long row, long col, double v, int value_is_sym) {
  if (self->issym && row < col) {
    long t;
    if (!value_is_sym) {
      PyErr_SetString(PyExc_IndexError, "Writing to upper triangle of symmetric matrix");
      return -1;
    }
    t = row;
    row = col;
    col = t;
  }
  return SpMatrix_LLMatSetItem(self, (int)row, (int)col, v);
}

/* ll_mat.c:927-1255.  value: a number (every element of the block) or an ll_mat of the block's shape; afterwards the block
 * IS the value.  Two slices and a matrix: the block is emptied, then the entries the matrix stores are written (cost ~
 * the entries involved); every other combination writes every element of the block, zeros included (they delete). */
static int LLMat_ass_subscript(LLMatObject *self, PyObject *key, PyObject *value) {
  LLIndexSet RI, CJ;
  LLMatObject *mat = NULL;
  PyObject *held = NULL;
  double x = 0.0;
  long i, j;
  int rc = -1, k, is_num;
  if (value == NULL) {
    PyErr_SetString(PyExc_IndexError, "cannot delete matrix entries");
    return -1;
  }
  if (ll_parse_key(self, key, &RI, &CJ)) return -1;
  is_num = !PyObject_TypeCheck(value, &LLMatType);
  if (is_num) {
    x = PyFloat_AsDouble(value);
    if (x == -1.0 && PyErr_Occurred()) {
      PyErr_Clear();
      PyErr_SetString(PyExc_ValueError, RI.is_int && CJ.is_int ? "Value must be double" : "Value must be a number or an ll_mat");
      goto done;
    }
  }
  if (RI.is_int && CJ.is_int) {
    if (!is_num) {
      PyErr_SetString(PyExc_ValueError, "Value must be double");
      goto done;
    }
    rc = SpMatrix_LLMatSetItem(self, (int)RI.idx[0], (int)CJ.idx[0], x); /* (above the diagonal of a symmetric matrix: IndexError) */
    goto done;
  }
  if (is_num) {
    for (i = 0; i < RI.n; i++)
      for (j = 0; j < CJ.n; j++)
        if (ll_assign_one(self, RI.idx[i], CJ.idx[j], x, 0)) goto done;
    rc = 0;
    goto done;
  }
  mat = (LLMatObject *)value;
  if (mat->dim[0] != RI.n || mat->dim[1] != CJ.n) {
    PyErr_SetString(PyExc_ValueError, "Matrix shapes are different");
    goto done;
  }
  if (mat == self) { /* A[I, J] = A: read from a copy */
    if ((held = ll_copy(self)) == NULL) goto done;
    mat = (LLMatObject *)held;
  }
  if (RI.is_slice && CJ.is_slice) {
    /* nothing is changed when an entry cannot be written */
    if (self->issym && !mat->issym)
      for (i = 0; i < RI.n; i++)
        for (k = mat->root[i]; k != -1; k = mat->link[k])
          if (RI.idx[i] < CJ.idx[mat->col[k]]) {
            PyErr_SetString(PyExc_IndexError, "Writing to upper triangle of symmetric matrix");
            goto done;
          }
    ll_clear_block(self, &RI, &CJ);
    for (i = 0; i < RI.n; i++)
      for (k = mat->root[i]; k != -1; k = mat->link[k]) {
        const double v = mat->val[k];
        j = mat->col[k];
        if (ll_assign_one(self, RI.idx[i], CJ.idx[j], v, mat->issym)) goto done;
        if (mat->issym && i != j && ll_assign_one(self, RI.idx[j], CJ.idx[i], v, 1)) goto done;
      }
  } else {
    for (i = 0; i < RI.n; i++)
      for (j = 0; j < CJ.n; j++) {
        const double v = SpMatrix_LLMatGetItem(mat, (int)i, (int)j);
        if (PyErr_Occurred() || ll_assign_one(self, RI.idx[i], CJ.idx[j], v, mat->issym)) goto done;
      }
  }
  rc = 0;
done:
  Py_XDECREF(held);
  ll_index_free(&RI);
  ll_index_free(&CJ);
  return rc;
}

/* ------------------------------------------------------------------ module functions: products of two ll_mat */

/* one row of C = (row of A) * B accumulated in a dense work row: the same additions in the same order as the reference's
 * element-wise update-adds (ll_mat.c:3488-3568), stored once per entry in ascending column order */
typedef struct {
  double *acc;
  int *mark, *cols, ncols;
} LLRowAcc;

static int ll_acc_init(LLRowAcc *w, int n) {
  int i;
  w->acc = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  w->mark = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  w->cols = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  w->ncols = 0;
  if (!w->acc || !w->mark || !w->cols) {
    free(w->acc);
    free(w->mark);
    free(w->cols);
    PyErr_NoMemory();
    return -1;
  }
  for (i = 0; i < n; i++) w->mark[i] = 0;
  return 0;
}

static void ll_acc_free(LLRowAcc *w) {
  free(w->acc);
  free(w->mark);
  free(w->cols);
}

static void ll_acc_add(LLRowAcc *w, int c, double v) {
  if (!w->mark[c]) {
    w->mark[c] = 1;
    w->acc[c] = v; /* the reference's first update-add of an absent entry stores v itself */
    w->cols[w->ncols++] = c;
  } else {
    w->acc[c] += v;
  }
}

static int ll_int_cmp(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

static int ll_acc_flush(LLRowAcc *w, LLMatObject *C, int row) {
  int t;
  qsort(w->cols, (size_t)w->ncols, sizeof(int), ll_int_cmp);
  for (t = 0; t < w->ncols; t++) {
    const int c = w->cols[t];
    w->mark[c] = 0;
    if (SpMatrix_LLMatSetItem(C, row, c, w->acc[c])) return -1;
  }
  w->ncols = 0;
  return 0;
}

/* ll_mat.c:3461-3660: C = A * B; general * general and symmetric * general (the other two: NotImplementedError) */
static PyObject *LLMat_matrixmultiply(PyObject *module, PyObject *args) {
  LLMatObject *A, *B, *C;
  int dim[2], i, kA, kB;
  if (!PyArg_ParseTuple(args, "O!O!", &LLMatType, &A, &LLMatType, &B)) return NULL;
  if (A->dim[1] != B->dim[0]) {
    PyErr_SetString(PyExc_ValueError, "matrix dimensions must agree");
    return NULL;
  }
  if (B->issym) {
    PyErr_SetString(PyExc_NotImplementedError, A->issym ? "multiply of two symmetric matrices not supported"
                                                        : "multiply of an unsymmetric and a symmetric matrix not supported");
    return NULL;
  }
  dim[0] = A->dim[0];
  dim[1] = B->dim[1];
  C = (LLMatObject *)SpMatrix_NewLLMatObject(dim, 0, 1000, A->storeZeros == 1 && B->storeZeros == 1);
  if (C == NULL) return NULL;
  if (!A->issym) {
    LLRowAcc w;
    if (ll_acc_init(&w, dim[1])) goto fail;
    for (i = 0; i < A->dim[0]; i++) {
      for (kA = A->root[i]; kA != -1; kA = A->link[kA])
        for (kB = B->root[A->col[kA]]; kB != -1; kB = B->link[kB]) ll_acc_add(&w, B->col[kB], A->val[kA] * B->val[kB]);
      if (ll_acc_flush(&w, C, i)) {
        ll_acc_free(&w);
        goto fail;
      }
    }
    ll_acc_free(&w);
  } else { /* a stored entry (i, j), j <= i, sends row j of B to row i of C and, off the diagonal, row i to row j (:3581-3612) */
    for (i = 0; i < A->dim[0]; i++)
      for (kA = A->root[i]; kA != -1; kA = A->link[kA]) {
        const int j = A->col[kA];
        const double a = A->val[kA];
        for (kB = B->root[j]; kB != -1; kB = B->link[kB])
          if (SpMatrix_LLMatUpdateItemAdd(C, i, B->col[kB], a * B->val[kB]) == -1) goto fail;
        if (i == j) continue;
        for (kB = B->root[i]; kB != -1; kB = B->link[kB])
          if (SpMatrix_LLMatUpdateItemAdd(C, j, B->col[kB], a * B->val[kB]) == -1) goto fail;
      }
  }
  return (PyObject *)C;
fail:
  Py_DECREF(C);
  return NULL;
}

/* ll_mat.c:3673-3722: C = A^T * B, both general */
static PyObject *LLMat_dot(PyObject *module, PyObject *args) {
  LLMatObject *A, *B, *C;
  int dim[2], i, kA, kB;
  if (!PyArg_ParseTuple(args, "O!O!", &LLMatType, &A, &LLMatType, &B)) return NULL;
  if (A->dim[0] != B->dim[0]) {
    PyErr_SetString(PyExc_ValueError, "matrix dimensions must agree");
    return NULL;
  }
  if (A->issym || B->issym) {
    PyErr_SetString(PyExc_NotImplementedError, "ddot operation with symmetric matrices not supported");
    return NULL;
  }
  dim[0] = A->dim[1];
  dim[1] = B->dim[1];
  C = (LLMatObject *)SpMatrix_NewLLMatObject(dim, 0, 1000, 1); /* storeZeros starts at 1 in the reference (:3674) */
  if (C == NULL) return NULL;
  for (i = 0; i < A->dim[0]; i++)
    for (kA = A->root[i]; kA != -1; kA = A->link[kA])
      for (kB = B->root[i]; kB != -1; kB = B->link[kB])
        if (SpMatrix_LLMatUpdateItemAdd(C, A->col[kA], B->col[kB], A->val[kA] * B->val[kB]) == -1) {
          Py_DECREF(C);
          return NULL;
        }
  return (PyObject *)C;
}

/* ll_mat.c:3730-3796: the symmetric matrix A^T * A, or A^T * diag(d) * A */
static PyObject *LLMat_symdot(PyObject *module, PyObject *args) {
  LLMatObject *A, *C;
  PyObject *din = NULL;
  PyArrayObject *d = NULL;
  int dim[2], i, kA, k2;
  if (!PyArg_ParseTuple(args, "O!|O", &LLMatType, &A, &din)) return NULL;
  if (din == Py_None) din = NULL;
  if (A->issym) {
    PyErr_SetString(PyExc_NotImplementedError, "symdot operation with symmetric matrices not supported");
    return NULL;
  }
  if (din && (d = ll_vec_arg(din, A->dim[0], "Scaling vector")) == NULL) return NULL;
  dim[0] = dim[1] = A->dim[1];
  C = (LLMatObject *)SpMatrix_NewLLMatObject(dim, 1, 1000, A->storeZeros == 1);
  if (C == NULL) {
    Py_XDECREF(d);
    return NULL;
  }
  for (i = 0; i < A->dim[0]; i++)
    for (kA = A->root[i]; kA != -1; kA = A->link[kA]) {
      double a = A->val[kA];
      const int r = A->col[kA];
      if (d) a *= ((double *)PyArray_DATA(d))[i];
      for (k2 = A->root[i]; k2 != -1; k2 = A->link[k2])
        if (r >= A->col[k2] && SpMatrix_LLMatUpdateItemAdd(C, r, A->col[k2], a * A->val[k2]) == -1) {
          Py_XDECREF(d);
          Py_DECREF(C);
          return NULL;
        }
    }
  Py_XDECREF(d);
  return (PyObject *)C;
}

/* ------------------------------------------------------------------ str(A): the text the reference's tp_print writes
 * (ll_mat.c:3085-3151; PPRINT thresholds :16-17): a dense picture up to 500 x 20, the entry list beyond */

typedef struct {
  char *p;
  size_t len, cap;
} LLText;

static int ll_text_add(LLText *t, const char *fmt, ...) {
  va_list ap;
  int n;
  if (t->cap - t->len < 64) {
    size_t cap = t->cap ? 2 * t->cap : 1024;
    char *q = (char *)realloc(t->p, cap);
    if (q == NULL) return -1;
    t->p = q;
    t->cap = cap;
  }
  va_start(ap, fmt);
  n = vsnprintf(t->p + t->len, t->cap - t->len, fmt, ap);
  va_end(ap);
  if (n < 0 || (size_t)n >= t->cap - t->len) return -1; /* (every piece is far below 64 characters) */
  t->len += (size_t)n;
  return 0;
}

static PyObject *LLMat_str(LLMatObject *a) {
  const char *sym = a->issym ? "symmetric" : "general";
  LLText t = {NULL, 0, 0};
  PyObject *ret;
  int i, j, k, bad = 0;
  if (a->dim[1] <= 20 && a->dim[0] <= 500) {
    double *row = (double *)malloc(sizeof(double) * (size_t)(a->dim[1] > 0 ? a->dim[1] : 1));
    if (row == NULL) return PyErr_NoMemory();
    bad |= ll_text_add(&t, "ll_mat(%s, [%d,%d]):\n", sym, a->dim[0], a->dim[1]);
    for (i = 0; i < a->dim[0] && !bad; i++) {
      for (j = 0; j < a->dim[1]; j++) row[j] = 0.0;
      for (k = a->root[i]; k != -1; k = a->link[k]) row[a->col[k]] = a->val[k];
      for (j = 0; j < a->dim[1] && !bad; j++) {
        const double v = row[j];
        if (v != 0.0) {
          const int e = (int)log10(fabs(v));
          if (abs(e) <= 4)
            bad |= ll_text_add(&t, "%9.*f ", e < 0 ? 6 : 6 - e, v);
          else
            bad |= ll_text_add(&t, "%9.1e ", v);
        } else if (!a->issym || i > j) {
          bad |= ll_text_add(&t, " -------- ");
        }
      }
      bad |= ll_text_add(&t, "\n");
    }
    free(row);
  } else if (a->nnz == 0) {
    bad |= ll_text_add(&t, "ll_mat(%s, [%d,%d])", sym, a->dim[0], a->dim[1]);
  } else {
    int first = 1;
    bad |= ll_text_add(&t, "ll_mat(%s, [%d,%d], [", sym, a->dim[0], a->dim[1]);
    for (i = 0; i < a->dim[0] && !bad; i++)
      for (k = a->root[i]; k != -1 && !bad; k = a->link[k]) {
        bad |= ll_text_add(&t, "%s(%d,%d): %g", first ? "" : ", ", i, a->col[k], a->val[k]);
        first = 0;
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

/* ll_mat.c:3193-3197: len(A) = rows * columns */
static Py_ssize_t LLMat_length(LLMatObject *a) { return (Py_ssize_t)a->dim[0] * (Py_ssize_t)a->dim[1]; }
