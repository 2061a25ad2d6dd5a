/* _tq_fastcall -- a ~100-line CPython stub over the SAME C ABI as ctypes uses (include/tq_hip.h): the foreign call of a
 * launch-bound quantizer forward without ctypes' argument marshalling (~2 us per call for 8 arguments, x 161 / 1333
 * quantizer calls per BERT-base / MobileBERT forward).
 *
 * Plain C against Python.h only: no torch headers, no pybind11.  It never looks inside a tensor -- the Python side
 * passes `tensor.data_ptr()` integers and the address of a descriptor it keeps alive -- and it never dlopen()s anything:
 * the entry point is handed over as an address taken from the ctypes handle of libtq_hip.so (`ctypes.cast(fn,
 * c_void_p).value`), so both call routes end in the same function of the same mapped library.  ctypes stays the reference
 * route (and the fallback when this module is not built); tests/test_abi.py runs the two against each other.
 *
 *   fake_quant_fwd(fn, x, y, idx, idx_dtype, n, dtype, q, stream) -> int
 *       int tq_fake_quant_fwd(const void* x, void* y, void* idx, int idx_dtype, uint64_t n, int dtype,
 *                             const tq_quantizer* q, tq_stream_t stream)            (include/tq_hip.h:81)
 *   call_ptrs(fn, a0, ..., a11) -> int
 *       any `int f(uintptr_t x 12)` shaped entry point whose arguments are all pointers / 64-bit sizes passed in
 *       integer registers (System V x86-64: the first six in registers, the rest on the stack, unused ones ignored)
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*fq_fwd_t)(const void*, void*, void*, int, uint64_t, int, const void*, void*);

static int as_u64(PyObject* o, uint64_t* out) {
  if (o == Py_None) { *out = 0; return 0; }
  unsigned long long v = PyLong_AsUnsignedLongLong(o);
  if (v == (unsigned long long)-1 && PyErr_Occurred()) return -1;
  *out = (uint64_t)v;
  return 0;
}

static PyObject* fake_quant_fwd(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  (void)self;
  if (nargs != 9) {
    PyErr_SetString(PyExc_TypeError, "fake_quant_fwd(fn, x, y, idx, idx_dtype, n, dtype, q, stream)");
    return NULL;
  }
  uint64_t v[9];
  for (int i = 0; i < 9; ++i)
    if (as_u64(args[i], &v[i])) return NULL;
  if (v[0] == 0) {
    PyErr_SetString(PyExc_ValueError, "fake_quant_fwd: NULL entry point");
    return NULL;
  }
  fq_fwd_t fn = (fq_fwd_t)(uintptr_t)v[0];
  /* the launch is asynchronous and returns in microseconds: the GIL is kept (releasing + re-taking it costs more) */
  int rc = fn((const void*)(uintptr_t)v[1], (void*)(uintptr_t)v[2], (void*)(uintptr_t)v[3], (int)v[4], v[5], (int)v[6],
              (const void*)(uintptr_t)v[7], (void*)(uintptr_t)v[8]);
  return PyLong_FromLong(rc);
}

typedef int (*ptr12_t)(uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t,
                       uintptr_t, uintptr_t, uintptr_t);

static PyObject* call_ptrs(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  (void)self;
  if (nargs < 1 || nargs > 13) {
    PyErr_SetString(PyExc_TypeError, "call_ptrs(fn, up to 12 integer / pointer arguments)");
    return NULL;
  }
  uint64_t v[13] = {0};
  for (Py_ssize_t i = 0; i < nargs; ++i)
    if (as_u64(args[i], &v[i])) return NULL;
  if (v[0] == 0) {
    PyErr_SetString(PyExc_ValueError, "call_ptrs: NULL entry point");
    return NULL;
  }
  int rc = ((ptr12_t)(uintptr_t)v[0])(v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12]);
  return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"fake_quant_fwd", (PyCFunction)(void (*)(void))fake_quant_fwd, METH_FASTCALL, "tq_fake_quant_fwd through a raw address"},
    {"call_ptrs", (PyCFunction)(void (*)(void))call_ptrs, METH_FASTCALL, "int f(uintptr_t...) through a raw address"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_tq_fastcall", "fast foreign calls into libtq_hip.so", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__tq_fastcall(void) { return PyModule_Create(&moddef); }
