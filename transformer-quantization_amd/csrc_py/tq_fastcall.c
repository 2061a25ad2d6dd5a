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
 *   calibrate_tensor(fn, <21 arguments>) -> int
 *       int tq_calibrate_tensor(...)   the calibrating call of a per-tensor quantizer                (include/tq_hip.h:275)
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

/* int tq_calibrate_tensor(const void* x, uint64_t n, int dtype, int mode, const float* prev_min, const float* prev_max,
 *                         float* cur_min, float* cur_max, double momentum, int n_bits, int symmetric, float eps,
 *                         int log_domain, float* delta, float* zero_float, uint8_t* signed_flag, void* y, void* workspace,
 *                         size_t workspace_bytes, uint32_t* counter, tq_stream_t stream)           (include/tq_hip.h:275)
 * the launch-bound CALIBRATING call of a per-tensor quantizer (161 per BERT-base calibration batch): arguments in the
 * order of the C prototype, after the entry point's address. */
typedef int (*calib_tensor_t)(const void*, uint64_t, int, int, const float*, const float*, float*, float*, double, int, int,
                              float, int, float*, float*, uint8_t*, void*, void*, size_t, uint32_t*, void*);

static PyObject* calibrate_tensor(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  (void)self;
  if (nargs != 22) {
    PyErr_SetString(PyExc_TypeError, "calibrate_tensor(fn, <the 21 arguments of tq_calibrate_tensor>)");
    return NULL;
  }
  uint64_t v[22];
  double momentum = 0.0, eps = 0.0;
  for (int i = 0; i < 22; ++i) {
    if (i == 9 || i == 12) {                      /* momentum (double), eps (float) */
      const double d = PyFloat_AsDouble(args[i]);
      if (d == -1.0 && PyErr_Occurred()) return NULL;
      if (i == 9) momentum = d; else eps = d;
      v[i] = 0;
    } else if (as_u64(args[i], &v[i])) {
      return NULL;
    }
  }
  if (v[0] == 0) {
    PyErr_SetString(PyExc_ValueError, "calibrate_tensor: NULL entry point");
    return NULL;
  }
  calib_tensor_t fn = (calib_tensor_t)(uintptr_t)v[0];
#define P(i) ((void*)(uintptr_t)v[i])
  int rc = fn(P(1), v[2], (int)v[3], (int)v[4], (const float*)P(5), (const float*)P(6), (float*)P(7), (float*)P(8), momentum,
              (int)v[10], (int)v[11], (float)eps, (int)v[13], (float*)P(14), (float*)P(15), (uint8_t*)P(16), P(17), P(18),
              (size_t)v[19], (uint32_t*)P(20), P(21));
#undef P
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
    {"calibrate_tensor", (PyCFunction)(void (*)(void))calibrate_tensor, METH_FASTCALL, "tq_calibrate_tensor through a raw address"},
    {"call_ptrs", (PyCFunction)(void (*)(void))call_ptrs, METH_FASTCALL, "int f(uintptr_t...) through a raw address"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_tq_fastcall", "fast foreign calls into libtq_hip.so", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__tq_fastcall(void) { return PyModule_Create(&moddef); }
