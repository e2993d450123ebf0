// pyniNVCategory -- CPython glue of the nvcategory Python class (python/cpp/pycategory.cpp in the reference,
// method table :900-930) for the string-category members, over libNVCategory.so.
#include "nvstrings/NVCategory.h"
#include "pyni_common.h"

using namespace pyni;

namespace pyni {
template <>
struct Bridge<NVCategory> {
  static NVCategory* wrap(void* h) { return NVCategory::adopt(static_cast<cs_category*>(h)); }
  static void drop(NVCategory* c) {
    c->release();
    NVCategory::destroy(c);
  }
};
}  // namespace pyni

#define SELF(args) ptr_arg<NVCategory>(args, 0)

static PyObject* n_createCategoryFromHostStrings(PyObject*, PyObject* args) {  // pycategory.cpp:89-130
  PyObject* strs = arg(args, 0);
  std::vector<const char*> list;
  if (PyUnicode_Check(strs)) list.push_back(PyUnicode_AsUTF8(strs));
  else if (PyList_Check(strs)) list_strings(strs, list);
  else {
    PyErr_SetString(PyExc_ValueError, "nvcategory: a list of str is required");
    return nullptr;
  }
  return make_instance([&] { return NVCategory::create_from_array(list.data(), (unsigned int)list.size()); });
}
static PyObject* n_createCategoryFromNVStrings(PyObject*, PyObject* args) {  // one nvstrings object or a list of them
  PyObject* o = arg(args, 0);
  std::vector<NVStrings*> all;
  if (PyList_Check(o))
    for (Py_ssize_t i = 0; i < PyList_Size(o); ++i) all.push_back(handle_of<NVStrings>(PyList_GetItem(o, i)));
  else all.push_back(handle_of<NVStrings>(o));
  for (auto* p : all)
    if (!p) {
      PyErr_SetString(PyExc_ValueError, "nvcategory: argument must be nvstrings object(s)");
      return nullptr;
    }
  return make_instance([&] { return all.size() == 1 ? NVCategory::create_from_strings(*all[0]) : NVCategory::create_from_strings(all); });
}
static PyObject* n_createFromOffsets(PyObject*, PyObject* args) {  // (sbuf, obuf, scount, nbuf, ncount, bdevmem)
  Region chars(arg(args, 0)), offs(arg(args, 1)), nulls(arg(args, 3));
  const unsigned int count = (unsigned int)int_arg(args, 2, 0);
  const int ncount = (int)int_arg(args, 4, 0);
  const bool dev = bool_arg(args, 5);
  return make_instance(
      [&] { return NVCategory::create_from_offsets((const char*)chars.p, count, (const int*)offs.p, (const unsigned char*)nulls.p, ncount, dev); });
}
static PyObject* n_destroyCategory(PyObject*, PyObject* args) {
  NVCategory* c = SELF(args);
  guarded([&] { NVCategory::destroy(c); });
  return PyLong_FromLong(0);
}
static PyObject* n_size(PyObject*, PyObject* args) { return PyLong_FromLong((long)SELF(args)->size()); }
static PyObject* n_keys_size(PyObject*, PyObject* args) { return PyLong_FromLong((long)SELF(args)->keys_size()); }
static PyObject* n_keys_type(PyObject*, PyObject*) { return PyUnicode_FromString("str"); }
static PyObject* n_get_keys(PyObject*, PyObject* args) {
  NVCategory* c = SELF(args);
  return make_instance([&] { return c->get_keys(); });
}
static PyObject* n_get_value_for_index(PyObject*, PyObject* args) {
  NVCategory* c = SELF(args);
  const unsigned int i = (unsigned int)int_arg(args, 1, 0);
  int v = -1;
  if (!guarded([&] { v = c->get_value(i); })) return PyErr_Occurred() ? nullptr : none();
  return PyLong_FromLong(v);
}
static PyObject* n_get_value_for_string(PyObject*, PyObject* args) {
  NVCategory* c = SELF(args);
  const char* s = str_arg(args, 1);
  int v = -1;
  if (!guarded([&] { v = c->get_value(s); })) return PyErr_Occurred() ? nullptr : none();
  return PyLong_FromLong(v);
}
static PyObject* n_get_values(PyObject*, PyObject* args) {  // (self, devptr) -> devptr | list
  NVCategory* c = SELF(args);
  int* devptr = ptr_arg<int>(args, 1);
  if (devptr) {
    if (!guarded([&] { c->get_values(devptr, true); })) return PyErr_Occurred() ? nullptr : none();
    return PyLong_FromVoidPtr(devptr);
  }
  const unsigned int n = c->size();
  std::vector<int> v(n ? n : 1);
  if (n && !guarded([&] { c->get_values(v.data(), false); })) return nullptr;
  PyObject* ret = PyList_New(n);
  for (unsigned int i = 0; i < n; ++i) PyList_SetItem(ret, i, PyLong_FromLong(v[i]));
  return ret;
}
static PyObject* n_get_values_cpointer(PyObject*, PyObject* args) { return from_ptr(SELF(args)->values_cptr()); }
static PyObject* n_get_indexes_for_key(PyObject*, PyObject* args) {  // (self, key, devptr) -> list of rows
  NVCategory* c = SELF(args);
  const char* key = str_arg(args, 1);
  const unsigned int n = c->size();
  std::vector<int> rows(n ? n : 1);
  int found = 0;
  if (!guarded([&] { found = c->get_indexes_for(key, rows.data(), false); })) return PyErr_Occurred() ? nullptr : none();
  if (found < 0) found = 0;
  PyObject* ret = PyList_New(found);
  for (int i = 0; i < found; ++i) PyList_SetItem(ret, i, PyLong_FromLong(rows[(size_t)i]));
  return ret;
}
#define WITH_STRINGS(NAME, CALL)                                                          \
  static PyObject* NAME(PyObject*, PyObject* args) {                                      \
    NVCategory* c = SELF(args);                                                           \
    NVStrings* s = handle_of<NVStrings>(arg(args, 1));                                    \
    if (!s) {                                                                             \
      PyErr_SetString(PyExc_ValueError, "nvcategory: parameter must be nvstrings object"); \
      return nullptr;                                                                     \
    }                                                                                     \
    return make_instance([&] { return c->CALL(*s); });                                    \
  }
WITH_STRINGS(n_add_strings, add_strings)
WITH_STRINGS(n_remove_strings, remove_strings)
WITH_STRINGS(n_add_keys, add_keys_and_remap)
WITH_STRINGS(n_remove_keys, remove_keys_and_remap)
WITH_STRINGS(n_set_keys, set_keys_and_remap)
#define WITH_CATEGORY(NAME, CALL)                                                          \
  static PyObject* NAME(PyObject*, PyObject* args) {                                       \
    NVCategory* c = SELF(args);                                                            \
    NVCategory* o = handle_of<NVCategory>(arg(args, 1));                                   \
    if (!o) {                                                                              \
      PyErr_SetString(PyExc_ValueError, "nvcategory: parameter must be nvcategory object"); \
      return nullptr;                                                                      \
    }                                                                                      \
    return make_instance([&] { return c->CALL(*o); });                                     \
  }
WITH_CATEGORY(n_merge_category, merge_category)
WITH_CATEGORY(n_merge_and_remap, merge_and_remap)
static PyObject* n_remove_unused_keys(PyObject*, PyObject* args) {
  NVCategory* c = SELF(args);
  return make_instance([&] { return c->remove_unused_keys_and_remap(); });
}
static PyObject* n_to_strings(PyObject*, PyObject* args) {
  NVCategory* c = SELF(args);
  return make_instance([&] { return c->to_strings(); });
}
#define WITH_INDEXES(NAME, CALL)                                                                     \
  static PyObject* NAME(PyObject*, PyObject* args) {                                                 \
    NVCategory* c = SELF(args);                                                                      \
    Array<int> a(arg(args, 1));                                                                      \
    const unsigned int count = a.on_device ? (unsigned int)int_arg(args, 2, 0) : (unsigned int)a.count; \
    return make_instance([&] { return c->CALL(a.data, count, a.on_device); });                       \
  }
WITH_INDEXES(n_gather_strings, gather_strings)
WITH_INDEXES(n_gather, gather)
WITH_INDEXES(n_gather_and_remap, gather_and_remap)

static PyObject* n_dropWrapper(PyObject*, PyObject* args) { return drop_wrapper<NVCategory>(args); }

static PyMethodDef s_Methods[] = {
#define M(n) {#n, n, METH_VARARGS, ""}
    M(n_dropWrapper),
    M(n_createCategoryFromHostStrings), M(n_createCategoryFromNVStrings), M(n_createFromOffsets), M(n_destroyCategory), M(n_size), M(n_keys_size),
    M(n_keys_type), M(n_get_keys), M(n_get_indexes_for_key), M(n_get_value_for_index), M(n_get_value_for_string), M(n_get_values),
    M(n_get_values_cpointer), M(n_add_strings), M(n_remove_strings), M(n_to_strings), M(n_gather_strings), M(n_gather), M(n_gather_and_remap),
    M(n_merge_category), M(n_merge_and_remap), M(n_add_keys), M(n_remove_keys), M(n_remove_unused_keys), M(n_set_keys),
#undef M
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef s_Module = {PyModuleDef_HEAD_INIT, "pyniNVCategory", "CPython glue of nvcategory over the MI355X back-end", -1, s_Methods};
PyMODINIT_FUNC PyInit_pyniNVCategory(void) { return PyModule_Create(&s_Module); }
