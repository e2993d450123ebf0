// pyniNVStrings -- CPython glue of the nvstrings Python class (python/cpp/pystrings.cpp in the reference,
// method table :3860-3973) for the members SURVEY.md section 8 names, over libNVStrings.so.
#include "pyni_common.h"
#include "nvstrings/ipc_transfer.h"
#include <cstring>

using namespace pyni;

#define SELF(args) ptr_arg<NVStrings>(args, 0)

// ---- construction / export ----------------------------------------------------------------------------
static PyObject* n_createFromHostStrings(PyObject*, PyObject* args) {  // pystrings.cpp:212-250
  PyObject* strs = arg(args, 0);
  std::vector<const char*> list;
  if (PyUnicode_Check(strs)) list.push_back(PyUnicode_AsUTF8(strs));
  else if (PyList_Check(strs)) list_strings(strs, list);
  else {
    PyErr_SetString(PyExc_ValueError, "nvstrings: a list of str is required");
    return nullptr;
  }
  return make_instance([&] { return NVStrings::create_from_array(list.data(), (unsigned int)list.size()); });
}
static PyObject* n_destroyStrings(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  guarded([&] { NVStrings::destroy(s); });
  return PyLong_FromLong(0);
}
static PyObject* n_createHostStrings(PyObject*, PyObject* args) { return host_strings(SELF(args)); }
static PyObject* n_createFromOffsets(PyObject*, PyObject* args) {  // (sbuf, obuf, scount, nbuf, ncount, bdevmem)
  if (arg(args, 0) == Py_None || arg(args, 1) == Py_None) {
    PyErr_SetString(PyExc_ValueError, "nvstrings: missing parameter");
    return nullptr;
  }
  Region chars(arg(args, 0)), offs(arg(args, 1)), nulls(arg(args, 3));
  const int count = (int)int_arg(args, 2, 0), ncount = (int)int_arg(args, 4, 0);
  const bool dev = bool_arg(args, 5);
  return make_instance([&] {
    return NVStrings::create_from_offsets((const char*)chars.p, count, (const int*)offs.p, (const unsigned char*)nulls.p, ncount, dev);
  });
}
static PyObject* n_create_offsets(PyObject*, PyObject* args) {  // (self, sbuf, obuf, nbuf, bdevmem)
  NVStrings* s = SELF(args);
  if (arg(args, 1) == Py_None || arg(args, 2) == Py_None) {
    PyErr_SetString(PyExc_ValueError, "nvstrings: missing parameter");
    return nullptr;
  }
  Region chars(arg(args, 1)), offs(arg(args, 2)), nulls(arg(args, 3));
  const bool dev = bool_arg(args, 4);
  if (!guarded([&] { s->create_offsets((char*)chars.p, (int*)offs.p, (unsigned char*)nulls.p, dev); })) return nullptr;
  return none();
}
static PyObject* n_createFromNVStrings(PyObject*, PyObject* args) {  // one instance or a list of them -> their rows in order
  PyObject* o = arg(args, 0);
  std::vector<NVStrings*> all;
  if (PyList_Check(o))
    for (Py_ssize_t i = 0; i < PyList_Size(o); ++i) all.push_back(handle_of<NVStrings>(PyList_GetItem(o, i)));
  else all.push_back(handle_of<NVStrings>(o));
  for (auto* p : all)
    if (!p) {
      PyErr_SetString(PyExc_ValueError, "nvstrings: argument list must contain nvstrings objects");
      return nullptr;
    }
  return make_instance([&] { return NVStrings::create_from_strings(all); });
}
static PyObject* n_getIPCData(PyObject*, PyObject* args) {  // pystrings.cpp: n_getIPCData -- here the transfer record as bytes
  NVStrings* s = SELF(args);
  nvstrings_ipc_transfer rec;
  if (!guarded([&] { s->create_ipc_transfer(rec); })) return nullptr;
  return PyBytes_FromStringAndSize(reinterpret_cast<const char*>(&rec), (Py_ssize_t)sizeof(rec));
}
static PyObject* n_createFromIPC(PyObject*, PyObject* args) {  // pystrings.cpp: n_createFromIPC
  PyObject* o = arg(args, 0);
  char* data = nullptr;
  Py_ssize_t len = 0;
  if (!PyBytes_Check(o) || PyBytes_AsStringAndSize(o, &data, &len) != 0 || len != (Py_ssize_t)sizeof(nvstrings_ipc_transfer)) {
    PyErr_SetString(PyExc_ValueError, "nvstrings: the bytes of get_ipc_data() are required");
    return nullptr;
  }
  nvstrings_ipc_transfer rec;
  memcpy(&rec, data, sizeof(rec));
  return make_instance([&] { return NVStrings::create_from_ipc(rec); });
}
static PyObject* n_size(PyObject*, PyObject* args) { return PyLong_FromLong((long)SELF(args)->size()); }
static PyObject* n_len(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  return int_results<int>(s, ptr_arg<int>(args, 1), 0, [&](int* out, bool dev) { s->len(out, dev); });
}
static PyObject* n_byte_count(PyObject*, PyObject* args) {  // (self, memptr, bdevmem) -> total bytes
  NVStrings* s = SELF(args);
  int* mem = ptr_arg<int>(args, 1);
  const bool dev = bool_arg(args, 2);
  size_t total = 0;
  if (!guarded([&] { total = s->byte_count(mem, dev); })) return PyErr_Occurred() ? nullptr : none();
  return PyLong_FromLong((long)total);
}
static PyObject* n_null_count(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const bool empty_is_null = bool_arg(args, 1);
  unsigned int n = 0;
  std::vector<unsigned char> bits(((size_t)s->size() + 7) / 8 + 1);
  if (!guarded([&] { n = s->set_null_bitarray(bits.data(), empty_is_null, false); })) return PyErr_Occurred() ? nullptr : none();
  return PyLong_FromLong((long)n);
}
static PyObject* n_set_null_bitmask(PyObject*, PyObject* args) {  // (self, nbuf, bdevmem) -> null count
  NVStrings* s = SELF(args);
  Region nulls(arg(args, 1));
  const bool dev = bool_arg(args, 2);
  unsigned int n = 0;
  if (!guarded([&] { n = s->set_null_bitarray((unsigned char*)nulls.p, false, dev); })) return PyErr_Occurred() ? nullptr : none();
  return PyLong_FromLong((long)n);
}
static PyObject* n_copy(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  return make_instance([&] { return s->copy(); });
}

// ---- split family ---------------------------------------------------------------------------------------------
template <class F>
static PyObject* columns_of(F&& f) {
  std::vector<NVStrings*> results;
  if (!guarded([&] { f(results); })) return PyErr_Occurred() ? nullptr : none();
  return instance_list(results);
}
static PyObject* n_split(PyObject*, PyObject* args) {  // pystrings.cpp:1619-1642: (self, delimiter|None, n|None)
  NVStrings* s = SELF(args);
  const char* d = str_arg(args, 1);
  const int n = (int)int_arg(args, 2, -1);
  return columns_of([&](std::vector<NVStrings*>& r) { s->split(d, n, r); });
}
static PyObject* n_rsplit(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* d = str_arg(args, 1);
  const int n = (int)int_arg(args, 2, -1);
  return columns_of([&](std::vector<NVStrings*>& r) { s->rsplit(d, n, r); });
}
static PyObject* n_split_record(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* d = str_arg(args, 1);
  const int n = (int)int_arg(args, 2, -1);
  return columns_of([&](std::vector<NVStrings*>& r) { s->split_record(d, n, r); });
}
static PyObject* n_rsplit_record(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* d = str_arg(args, 1);
  const int n = (int)int_arg(args, 2, -1);
  return columns_of([&](std::vector<NVStrings*>& r) { s->rsplit_record(d, n, r); });
}
static PyObject* n_partition(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* d = str_arg(args, 1);
  return columns_of([&](std::vector<NVStrings*>& r) { s->partition(d, r); });
}
static PyObject* n_rpartition(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* d = str_arg(args, 1);
  return columns_of([&](std::vector<NVStrings*>& r) { s->rpartition(d, r); });
}

// ---- replace / strip / case -------------------------------------------------------------------------------------
static PyObject* n_replace(PyObject*, PyObject* args) {  // pystrings.cpp:1902-1931: "Oszip"
  PyObject* vo = nullptr;
  const char *pat = nullptr, *repl = nullptr;
  int maxrepl = -1, regex = 1;
  if (!PyArg_ParseTuple(args, "Oszip", &vo, &pat, &repl, &maxrepl, &regex)) {
    PyErr_Clear();
    PyErr_SetString(PyExc_ValueError, "nvstrings.replace: invalid parameters");
    return nullptr;
  }
  NVStrings* s = reinterpret_cast<NVStrings*>(PyLong_AsVoidPtr(vo));
  return make_instance([&] { return regex ? s->replace_re(pat, repl, maxrepl) : s->replace(pat, repl, maxrepl); });
}
static std::string quote_literal(const char* t) {  // a literal as a pattern of the regex dialect (regcomp.cpp:314-539)
  std::string out;
  for (; *t; ++t) {
    if (strchr("\\^$.|?*+()[]{}", *t)) out.push_back('\\');
    out.push_back(*t);
  }
  return out;
}
static PyObject* n_replace_multi(PyObject*, PyObject* args) {  // (self, pats: list of str | nvstrings, repls ptr, regex)
  NVStrings* s = SELF(args);
  PyObject* pats = arg(args, 1);
  NVStrings* repls = handle_of<NVStrings>(arg(args, 2));
  const bool regex = bool_arg(args, 3);
  std::vector<std::string> owned;
  std::vector<const char*> list;
  if (PyList_Check(pats)) {
    list_strings(pats, list);
  } else if (NVStrings* p = handle_of<NVStrings>(pats)) {  // literal targets held on the device
    PyObject* host = host_strings(p);
    if (host == Py_None) return host;
    list_strings(host, list);
    for (auto& q : list) owned.push_back(q ? q : "");
    Py_DECREF(host);
    for (size_t i = 0; i < list.size(); ++i) list[i] = list[i] ? owned[i].c_str() : nullptr;
  } else {
    PyErr_SetString(PyExc_ValueError, "nvstrings.replace_multi: pats must be list of str");
    return nullptr;
  }
  std::vector<std::string> quoted;
  if (!regex) {
    quoted.reserve(list.size());
    for (auto& q : list) {
      quoted.push_back(q ? quote_literal(q) : std::string());
      if (q) q = quoted.back().c_str();
    }
  }
  return make_instance([&] { return s->replace_re(list, *repls); });
}
static PyObject* n_replace_with_backrefs(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char *pat = str_arg(args, 1), *repl = str_arg(args, 2);
  return make_instance([&] { return s->replace_with_backrefs(pat, repl); });
}
#define STRIP_FN(NAME, CALL)                                  \
  static PyObject* NAME(PyObject*, PyObject* args) {          \
    NVStrings* s = SELF(args);                                \
    const char* t = str_arg(args, 1);                         \
    return make_instance([&] { return s->CALL(t); });         \
  }
STRIP_FN(n_lstrip, lstrip)
STRIP_FN(n_strip, strip)
STRIP_FN(n_rstrip, rstrip)
static PyObject* n_lower(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  return make_instance([&] { return s->lower(); });
}
static PyObject* n_upper(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  return make_instance([&] { return s->upper(); });
}

// ---- search -----------------------------------------------------------------------------------------------------------
static PyObject* n_find(PyObject*, PyObject* args) {  // (self, str, start, end|None, devptr): values < -1 -> None
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  const int start = (int)int_arg(args, 2, 0), end = (int)int_arg(args, 3, -1);
  return int_results<int>(s, ptr_arg<int>(args, 4), -1, [&](int* out, bool dev) { s->find(str, start, end, out, dev); });
}
// the rest of the find family (pystrings.cpp:1005-1044, 2234-2360, 2738-2916)
static PyObject* n_compare(PyObject*, PyObject* args) {  // (self, str, devptr): None for null rows in the host list
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  int* devptr = ptr_arg<int>(args, 2);
  if (devptr) return int_results<int>(s, devptr, 0, [&](int* out, bool dev) { s->compare(str, out, dev); });
  const unsigned int count = s->size();
  if (count == 0) return PyList_New(0);
  std::vector<int> host(count);
  std::vector<unsigned char> nulls((count + 7) / 8, 0);
  unsigned int ncount = 0;
  if (!guarded([&] {
        s->compare(str, host.data(), false);
        ncount = s->set_null_bitarray(nulls.data(), false, false);
      }))
    return nullptr;
  PyObject* ret = PyList_New(count);
  for (unsigned int i = 0; i < count; ++i) {
    if (ncount && !((nulls[i / 8] >> (i % 8)) & 1)) {
      Py_INCREF(Py_None);
      PyList_SetItem(ret, i, Py_None);
    } else {
      PyList_SetItem(ret, i, PyLong_FromLong((long)host[i]));
    }
  }
  return ret;
}
static PyObject* n_rfind(PyObject*, PyObject* args) {  // (self, str, start, end|None, devptr)
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  const int start = (int)int_arg(args, 2, 0), end = (int)int_arg(args, 3, -1);
  return int_results<int>(s, ptr_arg<int>(args, 4), -1, [&](int* out, bool dev) { s->rfind(str, start, end, out, dev); });
}
static PyObject* n_find_from(PyObject*, PyObject* args) {  // (self, str, starts devptr, ends devptr, devptr)
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  int *starts = ptr_arg<int>(args, 2), *ends = ptr_arg<int>(args, 3);
  return int_results<int>(s, ptr_arg<int>(args, 4), -1, [&](int* out, bool dev) { s->find_from(str, starts, ends, out, dev); });
}
static PyObject* n_startswith(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  return bool_results(s, ptr_arg<bool>(args, 2), [&](bool* out, bool dev) { return (int)s->startswith(str, out, dev); });
}
static PyObject* n_endswith(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  return bool_results(s, ptr_arg<bool>(args, 2), [&](bool* out, bool dev) { return (int)s->endswith(str, out, dev); });
}
// the other column: an nvstrings object or a list of str / None
struct OtherStrings {
  NVStrings* p = nullptr;
  bool own = false;
  OtherStrings(PyObject* o, const char* what) {
    if (o == Py_None) {
      PyErr_Format(PyExc_ValueError, "nvstrings.%s: parameter required", what);
    } else if (PyList_Check(o)) {
      if (PyList_Size(o) == 0) {
        PyErr_Format(PyExc_ValueError, "nvstrings.%s empty argument list", what);
        return;
      }
      std::vector<const char*> list;
      list_strings(o, list);
      guarded([&] { p = NVStrings::create_from_array(list.data(), (unsigned int)list.size()); });
      own = p != nullptr;
    } else {
      p = handle_of<NVStrings>(o);
      if (!p) PyErr_Format(PyExc_ValueError, "nvstrings.%s: argument must be nvstrings object", what);
    }
  }
  ~OtherStrings() {
    if (own) guarded([&] { NVStrings::destroy(p); });
  }
};
static PyObject* n_match_strings(PyObject*, PyObject* args) {  // (self, strs, devptr): plain bools (two nulls are equal)
  NVStrings* s = SELF(args);
  OtherStrings other(arg(args, 1), "match_strings");
  if (!other.p) return nullptr;
  if (other.p->size() != s->size()) {
    PyErr_SetString(PyExc_ValueError, "nvstrings.match_strings list size must match");
    return nullptr;
  }
  bool* devptr = ptr_arg<bool>(args, 2);
  if (devptr) {
    if (!guarded([&] { s->match_strings(*other.p, devptr, true); })) return nullptr;
    return PyLong_FromVoidPtr(devptr);
  }
  const unsigned int count = s->size();
  if (count == 0) return PyList_New(0);
  std::vector<unsigned char> host(count);
  if (!guarded([&] { s->match_strings(*other.p, reinterpret_cast<bool*>(host.data()), false); })) return nullptr;
  PyObject* ret = PyList_New(count);
  for (unsigned int i = 0; i < count; ++i) PyList_SetItem(ret, i, PyBool_FromLong(host[i]));
  return ret;
}
static PyObject* n_find_multiple(PyObject*, PyObject* args) {  // (self, strs, devptr): a list of positions per row
  NVStrings* s = SELF(args);
  OtherStrings other(arg(args, 1), "find_multiple");
  if (!other.p) return nullptr;
  int* devptr = ptr_arg<int>(args, 2);
  if (devptr) {
    if (!guarded([&] { s->find_multiple(*other.p, devptr, true); })) return nullptr;
    return PyLong_FromVoidPtr(devptr);
  }
  const unsigned int rows = s->size(), tc = other.p->size();
  PyObject* ret = PyList_New(rows);
  if (rows == 0) return ret;
  std::vector<int> host((size_t)rows * tc);
  if (!guarded([&] { s->find_multiple(*other.p, host.data(), false); })) {
    Py_DECREF(ret);
    return nullptr;
  }
  for (unsigned int r = 0; r < rows; ++r) {
    PyObject* row = PyList_New(tc);
    for (unsigned int j = 0; j < tc; ++j) {
      const int v = host[(size_t)r * tc + j];
      if (v < -1) {
        Py_INCREF(Py_None);
        PyList_SetItem(row, j, Py_None);
      } else {
        PyList_SetItem(row, j, PyLong_FromLong((long)v));
      }
    }
    PyList_SetItem(ret, r, row);
  }
  return ret;
}
static PyObject* n_contains(PyObject*, PyObject* args) {  // pystrings.cpp:2588-2666: (self, str, regex, devptr)
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  const bool regex = bool_arg(args, 2);
  return bool_results(s, ptr_arg<bool>(args, 3), [&](bool* out, bool dev) { return regex ? s->contains_re(str, out, dev) : s->contains(str, out, dev); });
}
static PyObject* n_match(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  return bool_results(s, ptr_arg<bool>(args, 2), [&](bool* out, bool dev) { return s->match(str, out, dev); });
}
static PyObject* n_count(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  int* devptr = ptr_arg<int>(args, 2);
  if (devptr) return int_results<int>(s, devptr, 0, [&](int* out, bool dev) { s->count_re(str, out, dev); });
  // host list: None for null rows (pystrings.cpp:2949-2968)
  const unsigned int count = s->size();
  if (count == 0) return PyList_New(0);
  std::vector<int> host(count);
  std::vector<unsigned char> nulls((count + 7) / 8, 0);
  unsigned int ncount = 0;
  if (!guarded([&] {
        s->count_re(str, host.data(), false);
        ncount = s->set_null_bitarray(nulls.data(), false, false);
      }))
    return nullptr;
  PyObject* ret = PyList_New(count);
  for (unsigned int i = 0; i < count; ++i) {
    if (ncount && !((nulls[i / 8] >> (i % 8)) & 1)) {
      Py_INCREF(Py_None);
      PyList_SetItem(ret, i, Py_None);
    } else {
      PyList_SetItem(ret, i, PyLong_FromLong(host[i]));
    }
  }
  return ret;
}
#define PATTERN_COLUMNS(NAME, CALL)                                                   \
  static PyObject* NAME(PyObject*, PyObject* args) {                                  \
    NVStrings* s = SELF(args);                                                        \
    const char* pat = str_arg(args, 1);                                               \
    return columns_of([&](std::vector<NVStrings*>& r) { s->CALL(pat, r); });          \
  }
PATTERN_COLUMNS(n_findall, findall)
PATTERN_COLUMNS(n_findall_record, findall_record)
PATTERN_COLUMNS(n_extract, extract)
PATTERN_COLUMNS(n_extract_record, extract_record)

// ---- re-arrangement / combine ---------------------------------------------------------------------------------------------
static PyObject* n_sort(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const NVStrings::sorttype st = (NVStrings::sorttype)int_arg(args, 1, 2);
  const bool asc = bool_arg(args, 2), nf = bool_arg(args, 3);
  return make_instance([&] { return s->sort(st, asc, nf); });
}
static PyObject* n_order(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const NVStrings::sorttype st = (NVStrings::sorttype)int_arg(args, 1, 2);
  const bool asc = bool_arg(args, 2), nf = bool_arg(args, 3);
  return int_results<unsigned int>(s, ptr_arg<unsigned int>(args, 4), 0, [&](unsigned int* out, bool dev) { s->order(st, asc, out, nf, dev); });
}
static PyObject* n_gather(PyObject*, PyObject* args) {  // (self, indexes: list of int | list of bool | address, count)
  NVStrings* s = SELF(args);
  PyObject* idx = arg(args, 1);
  if (PyList_Check(idx) && PyList_Size(idx) > 0 && PyBool_Check(PyList_GetItem(idx, 0))) {
    Array<unsigned char> mask(idx);
    if (mask.count != s->size()) {
      PyErr_SetString(PyExc_ValueError, "nvstrings.gather: the mask must have one entry per string");
      return nullptr;
    }
    return make_instance([&] { return s->gather(reinterpret_cast<const bool*>(mask.data), false); });
  }
  Array<int> a(idx);
  if (a.bad) {
    PyErr_SetString(PyExc_ValueError, "nvstrings.gather: unknown type of indexes");
    return nullptr;
  }
  if (a.all_bool && !a.on_device) {  // (a numpy bool array: the mask form, as a list of bool)
    if (a.count != s->size()) {
      PyErr_SetString(PyExc_ValueError, "nvstrings.gather: the mask must have one entry per string");
      return nullptr;
    }
    std::vector<unsigned char> m(a.count);
    for (size_t i = 0; i < a.count; ++i) m[i] = a.data[i] != 0;
    return make_instance([&] { return s->gather(reinterpret_cast<const bool*>(m.data()), false); });
  }
  const unsigned int count = a.on_device ? (unsigned int)int_arg(args, 2, 0) : (unsigned int)a.count;
  return make_instance([&] { return s->gather(a.data, count, a.on_device); });
}
static PyObject* n_sublist(PyObject*, PyObject* args) {  // (self, start|None, end|None, step|None)
  NVStrings* s = SELF(args);
  const unsigned int start = (unsigned int)int_arg(args, 1, 0), end = (unsigned int)int_arg(args, 2, (long)s->size());
  const int step = (int)int_arg(args, 3, 1);
  return make_instance([&] { return s->sublist(start, end, step); });
}
static PyObject* n_scatter(PyObject*, PyObject* args) {  // (self, nvstrings object, indexes)
  NVStrings* s = SELF(args);
  NVStrings* strs = handle_of<NVStrings>(arg(args, 1));
  if (!strs) {
    PyErr_SetString(PyExc_ValueError, "nvstrings.scatter: parameter must be nvstrings object");
    return nullptr;
  }
  Array<int> a(arg(args, 2));
  if (!a.on_device && a.count < strs->size()) {
    PyErr_SetString(PyExc_ValueError, "nvstrings.scatter: number of indexes must match the number of strings");
    return nullptr;
  }
  return make_instance([&] { return s->scatter(*strs, a.data, a.on_device); });
}
static PyObject* n_scalar_scatter(PyObject*, PyObject* args) {  // (self, str, indexes, count)
  NVStrings* s = SELF(args);
  const char* str = str_arg(args, 1);
  Array<int> a(arg(args, 2));
  const unsigned int count = a.on_device ? (unsigned int)int_arg(args, 3, 0) : (unsigned int)a.count;
  return make_instance([&] { return s->scatter(str, a.data, count, a.on_device); });
}
static PyObject* n_remove_strings(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  Array<int> a(arg(args, 1));
  const unsigned int count = a.on_device ? (unsigned int)int_arg(args, 2, 0) : (unsigned int)a.count;
  return make_instance([&] { return s->remove_strings(a.data, count, a.on_device); });
}
static PyObject* n_add_strings(PyObject*, PyObject* args) {  // (self, nvstrings | list of nvstrings)
  std::vector<NVStrings*> all{SELF(args)};
  PyObject* o = arg(args, 1);
  if (PyList_Check(o))
    for (Py_ssize_t i = 0; i < PyList_Size(o); ++i) all.push_back(handle_of<NVStrings>(PyList_GetItem(o, i)));
  else all.push_back(handle_of<NVStrings>(o));
  for (auto* p : all)
    if (!p) {
      PyErr_SetString(PyExc_ValueError, "nvstrings.add_strings: argument must be nvstrings object(s)");
      return nullptr;
    }
  return make_instance([&] { return NVStrings::create_from_strings(all); });
}
static PyObject* n_cat(PyObject*, PyObject* args) {  // (self, others: None | nvstrings | list of nvstrings, sep|None, na_rep|None)
  NVStrings* s = SELF(args);
  PyObject* others = arg(args, 1);
  const char *sep = str_arg(args, 2), *narep = str_arg(args, 3);
  if (others == Py_None)  // no others: all rows joined into one string (pystrings.cpp:1454-1473)
    return make_instance([&] { return s->join(sep ? sep : "", narep); });
  if (PyList_Check(others)) {
    std::vector<NVStrings*> list;
    for (Py_ssize_t i = 0; i < PyList_Size(others); ++i) list.push_back(handle_of<NVStrings>(PyList_GetItem(others, i)));
    return make_instance([&] { return s->cat(list, sep, narep); });
  }
  NVStrings* one = handle_of<NVStrings>(others);
  return make_instance([&] { return s->cat(one, sep, narep); });
}
static PyObject* n_join(PyObject*, PyObject* args) {
  NVStrings* s = SELF(args);
  const char* sep = str_arg(args, 1);
  return make_instance([&] { return s->join(sep ? sep : ""); });
}
static PyObject* n_device_memory(PyObject*, PyObject* args) { return PyLong_FromLong((long)SELF(args)->memsize()); }

static PyObject* n_dropWrapper(PyObject*, PyObject* args) { return drop_wrapper<NVStrings>(args); }

static PyMethodDef s_Methods[] = {
#define M(n) {#n, n, METH_VARARGS, ""}
    M(n_dropWrapper), M(n_getIPCData), M(n_createFromIPC),
    M(n_createFromHostStrings), M(n_destroyStrings), M(n_createHostStrings), M(n_createFromOffsets), M(n_createFromNVStrings), M(n_create_offsets),
    M(n_size), M(n_len), M(n_byte_count), M(n_null_count), M(n_set_null_bitmask), M(n_copy), M(n_split), M(n_rsplit), M(n_split_record),
    M(n_rsplit_record), M(n_partition), M(n_rpartition), M(n_replace), M(n_replace_multi), M(n_replace_with_backrefs), M(n_lstrip), M(n_strip),
    M(n_rstrip), M(n_lower), M(n_upper), M(n_find), M(n_rfind), M(n_find_from), M(n_find_multiple), M(n_compare), M(n_match_strings), M(n_startswith), M(n_endswith), M(n_contains), M(n_match), M(n_count), M(n_findall), M(n_findall_record), M(n_extract),
    M(n_extract_record), M(n_sort), M(n_order), M(n_gather), M(n_sublist), M(n_scatter), M(n_scalar_scatter), M(n_remove_strings),
    M(n_add_strings), M(n_cat), M(n_join), M(n_device_memory),
#undef M
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef s_Module = {PyModuleDef_HEAD_INIT, "pyniNVStrings", "CPython glue of nvstrings over the MI355X back-end", -1, s_Methods};
PyMODINIT_FUNC PyInit_pyniNVStrings(void) { return PyModule_Create(&s_Module); }
