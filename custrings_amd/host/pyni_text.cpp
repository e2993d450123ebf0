// pyniNVText -- CPython glue of the nvtext Python module (python/cpp/pytext.cpp in the reference, method
// table :653-666) for tokenize, n-grams and the token counters, over libNVText.so.  The strings arguments
// are nvstrings Python objects (their m_cptr is read), as in the reference.
#include "nvstrings/NVText.h"
#include "pyni_common.h"

using namespace pyni;

static NVStrings* strs_arg(PyObject* args, int i) {
  NVStrings* s = handle_of<NVStrings>(arg(args, i));
  if (!s) PyErr_SetString(PyExc_ValueError, "nvtext: parameter must be nvstrings object");
  return s;
}
static PyObject* n_tokenize(PyObject*, PyObject* args) {
  NVStrings* s = strs_arg(args, 0);
  if (!s) return nullptr;
  const char* d = str_arg(args, 1);
  return make_instance([&] { return NVText::tokenize(*s, d); });
}
static PyObject* n_tokenize_multi(PyObject*, PyObject* args) {  // (strs, delimiters: nvstrings object)
  NVStrings *s = strs_arg(args, 0), *d = s ? strs_arg(args, 1) : nullptr;
  if (!s || !d) return nullptr;
  return make_instance([&] { return NVText::tokenize(*s, *d); });
}
static PyObject* n_unique_tokens(PyObject*, PyObject* args) {
  NVStrings* s = strs_arg(args, 0);
  if (!s) return nullptr;
  const char* d = str_arg(args, 1);
  return make_instance([&] { return NVText::unique_tokens(*s, d); });
}
static PyObject* n_token_count(PyObject*, PyObject* args) {  // (strs, delimiter, devptr)
  NVStrings* s = strs_arg(args, 0);
  if (!s) return nullptr;
  const char* d = str_arg(args, 1);
  return int_results<unsigned int>(s, ptr_arg<unsigned int>(args, 2), 0, [&](unsigned int* out, bool dev) { NVText::token_count(*s, d, out, dev); });
}
static PyObject* n_tokens_counts(PyObject*, PyObject* args) {  // (strs, tgts, delimiter, devptr) -> list of rows
  NVStrings *s = strs_arg(args, 0), *t = s ? strs_arg(args, 1) : nullptr;
  if (!s || !t) return nullptr;
  const char* d = str_arg(args, 2);
  unsigned int* devptr = ptr_arg<unsigned int>(args, 3);
  if (devptr) {
    if (!guarded([&] { NVText::tokens_counts(*s, *t, d, devptr, true); })) return PyErr_Occurred() ? nullptr : none();
    return PyLong_FromVoidPtr(devptr);
  }
  const unsigned int rows = s->size(), tc = t->size();
  std::vector<unsigned int> host((size_t)rows * tc + 1);
  if (!guarded([&] { NVText::tokens_counts(*s, *t, d, host.data(), false); })) return PyErr_Occurred() ? nullptr : none();
  PyObject* ret = PyList_New(rows);
  for (unsigned int r = 0; r < rows; ++r) {
    PyObject* row = PyList_New(tc);
    for (unsigned int k = 0; k < tc; ++k) PyList_SetItem(row, k, PyLong_FromLong((long)host[(size_t)r * tc + k]));
    PyList_SetItem(ret, r, row);
  }
  return ret;
}
static PyObject* n_replace_tokens(PyObject*, PyObject* args) {  // (strs, tgts, repls, delimiter)
  NVStrings *s = strs_arg(args, 0), *t = s ? strs_arg(args, 1) : nullptr, *r = t ? strs_arg(args, 2) : nullptr;
  if (!s || !t || !r) return nullptr;
  const char* d = str_arg(args, 3);
  return make_instance([&] { return NVText::replace_tokens(*s, *t, *r, d); });
}
static PyObject* n_normalize_spaces(PyObject*, PyObject* args) {
  NVStrings* s = strs_arg(args, 0);
  if (!s) return nullptr;
  return make_instance([&] { return NVText::normalize_spaces(*s); });
}
static PyObject* n_create_ngrams(PyObject*, PyObject* args) {  // (strs, N, sep)
  NVStrings* s = strs_arg(args, 0);
  if (!s) return nullptr;
  const unsigned int n = (unsigned int)int_arg(args, 1, 2);
  const char* sep = str_arg(args, 2);
  return make_instance([&] { return NVText::create_ngrams(*s, n, sep ? sep : "_"); });
}

static PyMethodDef s_Methods[] = {
#define M(n) {#n, n, METH_VARARGS, ""}
    M(n_tokenize), M(n_tokenize_multi), M(n_unique_tokens), M(n_token_count), M(n_tokens_counts), M(n_replace_tokens), M(n_normalize_spaces), M(n_create_ngrams),
#undef M
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef s_Module = {PyModuleDef_HEAD_INIT, "pyniNVText", "CPython glue of nvtext over the MI355X back-end", -1, s_Methods};
PyMODINIT_FUNC PyInit_pyniNVText(void) { return PyModule_Create(&s_Module); }
