// Shared helpers of the CPython glue modules pyniNVStrings / pyniNVCategory / pyniNVText: the module
// names, function names (n_*), positional-argument conventions and return conventions of the reference's
// python/cpp/pystrings.cpp, pycategory.cpp and pytext.cpp, so the Python classes written for those
// modules drive this back-end unchanged.  Conventions kept (pystrings.cpp:212-330,1619-1642,1902-1931):
//   * instances travel as integers (the C++ pointer); a method that makes an instance returns its
//     pointer; a C++ exception is turned into ValueError (raised through CPython's NULL-return protocol:
//     the reference returns None with the error indicator set, which CPython >= 3.5 reports as
//     SystemError -- ValueError is what its callers were meant to see);
//   * the GIL is released around every C++ call;
//   * array arguments may be a Python list, a numpy array / buffer (host memory) or an int (device
//     address); array results go to the caller's device pointer when one is given, else come back as
//     a Python list with None for null rows.
#pragma once
#include <Python.h>

#include <exception>
#include <string>
#include <vector>

#include "nvstrings/NVStrings.h"

namespace pyni {

inline PyObject* arg(PyObject* args, int i) { return i < PyTuple_Size(args) ? PyTuple_GetItem(args, i) : Py_None; }
template <class T>
T* ptr_arg(PyObject* args, int i) {
  PyObject* o = arg(args, i);
  return o == Py_None ? nullptr : reinterpret_cast<T*>(PyLong_AsVoidPtr(o));
}
inline const char* str_arg(PyObject* args, int i) {  // None -> nullptr
  PyObject* o = arg(args, i);
  return o == Py_None ? nullptr : PyUnicode_AsUTF8(o);
}
inline long int_arg(PyObject* args, int i, long dflt) {
  PyObject* o = arg(args, i);
  return o == Py_None ? dflt : PyLong_AsLong(o);
}
inline bool bool_arg(PyObject* args, int i) { return PyObject_IsTrue(arg(args, i)) == 1; }
inline PyObject* none() { Py_RETURN_NONE; }
inline PyObject* from_ptr(const void* p) { return p ? PyLong_FromVoidPtr(const_cast<void*>(p)) : none(); }
// The instance behind an argument: an integer is the C++ pointer itself; a Python object written for
// the reference's glue carries it in m_cptr.  The ctypes classes of this repository (custrings_amd/
// nvstrings.py, nvcategory.py: `_cs_abi = True`) hold a C-ABI handle in m_cptr instead: such an object
// gets a C++ instance wrapped around its handle on first use, remembered on the object as _nv_cptr and
// dropped (without freeing the handle) by the object's own destructor through n_dropWrapper.
template <class T>
struct Bridge;  // Bridge<T>::wrap(handle) -> T* around a C-ABI handle the Python object keeps owning
template <>
struct Bridge<NVStrings> {
  static NVStrings* wrap(void* h) { return NVStrings::adopt(static_cast<cs_column*>(h)); }
  static void drop(NVStrings* s) {
    s->release();
    NVStrings::destroy(s);
  }
};
template <class T>
T* handle_of(PyObject* o) {
  if (o == Py_None) return nullptr;
  if (PyLong_Check(o)) return reinterpret_cast<T*>(PyLong_AsVoidPtr(o));
  PyObject* a = PyObject_GetAttrString(o, "m_cptr");
  if (!a) {
    PyErr_Clear();
    return nullptr;
  }
  void* p = a == Py_None ? nullptr : PyLong_AsVoidPtr(a);
  Py_DECREF(a);
  if (!p || !PyObject_HasAttrString(o, "_cs_abi")) return static_cast<T*>(p);
  if (PyObject* w = PyObject_GetAttrString(o, "_nv_cptr")) {
    void* have = w == Py_None ? nullptr : PyLong_AsVoidPtr(w);
    Py_DECREF(w);
    if (have) return static_cast<T*>(have);
  } else {
    PyErr_Clear();
  }
  T* made = Bridge<T>::wrap(p);
  PyObject* v = PyLong_FromVoidPtr(made);
  const int rc = PyObject_SetAttrString(o, "_nv_cptr", v);
  Py_DECREF(v);
  if (rc != 0) {
    PyErr_Clear();
    Bridge<T>::drop(made);
    return nullptr;
  }
  return made;
}
template <class T>
PyObject* drop_wrapper(PyObject* args) {  // n_dropWrapper(ptr): frees the C++ instance, not the handle inside it
  PyObject* o = PyTuple_Size(args) > 0 ? PyTuple_GetItem(args, 0) : Py_None;
  if (o != Py_None && PyLong_Check(o))
    if (T* p = reinterpret_cast<T*>(PyLong_AsVoidPtr(o))) Bridge<T>::drop(p);
  Py_RETURN_NONE;
}

// Runs `f` without the GIL; a C++ exception becomes ValueError (and `false`).
template <class F>
bool guarded(F&& f) {
  std::string message;
  bool failed = false;
  Py_BEGIN_ALLOW_THREADS
  try {
    f();
  } catch (const std::exception& e) {
    message = e.what();
    failed = true;
  }
  Py_END_ALLOW_THREADS
  if (failed) PyErr_SetString(PyExc_ValueError, message.empty() ? "nvstrings: the operation failed" : message.c_str());
  return !failed;
}
// the usual "call, return the new instance" shape
template <class F>
PyObject* make_instance(F&& f) {
  const void* r = nullptr;
  if (!guarded([&] { r = f(); })) return nullptr;
  return from_ptr(r);
}
inline PyObject* instance_list(const std::vector<NVStrings*>& v) {
  PyObject* ret = PyList_New((Py_ssize_t)v.size());
  for (size_t i = 0; i < v.size(); ++i) PyList_SetItem(ret, (Py_ssize_t)i, from_ptr(v[i]));
  return ret;
}

// An integer / bool array argument: Python list, numpy array or buffer (host), or int address (device).
template <class T>
struct Array {
  std::vector<T> own;
  Py_buffer view{};
  bool has_view = false;
  T* data = nullptr;
  size_t count = 0;
  bool on_device = false, all_bool = false, bad = false;
  explicit Array(PyObject* o) {
    if (o == Py_None) return;
    if (PyList_Check(o)) {
      count = (size_t)PyList_Size(o);
      own.resize(count ? count : 1);
      all_bool = count > 0;
      for (size_t i = 0; i < count; ++i) {
        PyObject* e = PyList_GetItem(o, (Py_ssize_t)i);
        all_bool = all_bool && PyBool_Check(e);
        own[i] = e == Py_None ? T(0) : (PyBool_Check(e) ? T(e == Py_True) : T(PyLong_AsLong(e)));
      }
      data = own.data();
    } else if (PyLong_Check(o)) {
      data = reinterpret_cast<T*>(PyLong_AsVoidPtr(o));
      on_device = true;
    } else if (PyObject_CheckBuffer(o) && PyObject_GetBuffer(o, &view, PyBUF_FORMAT | PyBUF_ND | PyBUF_C_CONTIGUOUS) == 0) {
      // A typed buffer (numpy array, array.array, memoryview): its ITEM size decides how it is read -- numpy's default
      // int64 index arrays handed to a 4-byte parameter used to be reinterpreted as pairs of int32 (twice the count, wrong
      // rows).  Same width: taken in place; another integer width or a bool array: converted element by element (a bool
      // array also sets all_bool, which routes gather / remove_strings to the mask overloads).
      has_view = true;
      const size_t isz = (size_t)view.itemsize, n = isz ? (size_t)view.len / isz : 0;
      const char f = view.format ? view.format[view.format[0] == '<' || view.format[0] == '=' || view.format[0] == '@' ? 1 : 0] : 'B';
      const bool is_bool = f == '?', is_signed = f == 'b' || f == 'h' || f == 'i' || f == 'l' || f == 'q' || f == 'n';
      const bool is_int = is_bool || is_signed || f == 'B' || f == 'H' || f == 'I' || f == 'L' || f == 'Q' || f == 'N';
      if (!is_int || !(isz == 1 || isz == 2 || isz == 4 || isz == 8)) {
        bad = true;
      } else if (isz == sizeof(T) && !is_bool) {
        data = static_cast<T*>(view.buf);
        count = n;
      } else {
        count = n;
        own.resize(n ? n : 1);
        const unsigned char* p = static_cast<const unsigned char*>(view.buf);
        for (size_t i = 0; i < n; ++i) {
          long long v = 0;
          if (isz == 1) v = is_signed ? (long long)*reinterpret_cast<const signed char*>(p + i) : (long long)p[i];
          else if (isz == 2) v = is_signed ? (long long)*reinterpret_cast<const short*>(p + 2 * i) : (long long)*reinterpret_cast<const unsigned short*>(p + 2 * i);
          else if (isz == 4) v = is_signed ? (long long)*reinterpret_cast<const int*>(p + 4 * i) : (long long)*reinterpret_cast<const unsigned int*>(p + 4 * i);
          else v = *reinterpret_cast<const long long*>(p + 8 * i);
          own[i] = is_bool ? T(v != 0) : T(v);
        }
        data = own.data();
        all_bool = is_bool && n > 0;
      }
    } else {
      PyErr_Clear();
      bad = true;
    }
  }
  ~Array() {
    if (has_view) PyBuffer_Release(&view);
  }
};

// Per-row integer results: to the caller's device pointer, or back as a list with None below `null_below`.
template <class T, class Call>
PyObject* int_results(NVStrings* s, T* devptr, long null_below, Call&& call) {
  if (devptr) {
    if (!guarded([&] { call(devptr, true); })) return PyErr_Occurred() ? nullptr : none();
    return PyLong_FromVoidPtr(devptr);
  }
  const unsigned int count = s->size();
  PyObject* ret = PyList_New(count);
  if (count == 0) return ret;
  std::vector<T> host(count);
  if (!guarded([&] { call(host.data(), false); })) {
    Py_DECREF(ret);
    return nullptr;
  }
  for (unsigned int i = 0; i < count; ++i) {
    if ((long)host[i] < null_below) {
      Py_INCREF(Py_None);
      PyList_SetItem(ret, i, Py_None);
    } else {
      PyList_SetItem(ret, i, PyLong_FromLong((long)host[i]));
    }
  }
  return ret;
}
// Per-row bool results; null rows come back as None in the host list (pystrings.cpp:2654-2662).
template <class Call>
PyObject* bool_results(NVStrings* s, bool* devptr, Call&& call) {
  if (devptr) {
    int rc = 0;
    if (!guarded([&] { rc = call(devptr, true); })) return nullptr;
    if (rc < 0) return none();
    return PyLong_FromVoidPtr(devptr);
  }
  const unsigned int count = s->size();
  if (count == 0) return PyList_New(0);
  std::vector<unsigned char> host(count);
  int rc = 0;
  if (!guarded([&] { rc = call(reinterpret_cast<bool*>(host.data()), false); })) return nullptr;
  if (rc < 0) return none();
  std::vector<unsigned char> nulls((count + 7) / 8, 0);
  unsigned int ncount = 0;
  guarded([&] { ncount = s->set_null_bitarray(nulls.data(), false, false); });
  PyObject* ret = PyList_New(count);
  for (unsigned int i = 0; i < count; ++i) {
    if (ncount && !((nulls[i / 8] >> (i % 8)) & 1)) {
      Py_INCREF(Py_None);
      PyList_SetItem(ret, i, Py_None);
    } else {
      PyList_SetItem(ret, i, PyBool_FromLong(host[i]));
    }
  }
  return ret;
}
// the rows of an instance as a Python list of str / None (n_createHostStrings)
inline PyObject* host_strings(NVStrings* s) {
  const unsigned int count = s->size();
  if (count == 0) return PyList_New(0);
  std::vector<int> lens(count);
  std::vector<char> buffer;
  std::vector<char*> rows(count);
  if (!guarded([&] {
        const size_t total = s->byte_count(lens.data(), false);
        buffer.assign(total + count + 1, 0);
        size_t off = 0;
        for (unsigned int i = 0; i < count; ++i) {
          rows[i] = buffer.data() + off;
          off += (size_t)(lens[i] > 0 ? lens[i] : 0) + 1;
        }
        s->to_host(rows.data(), 0, (int)count);
      }))
    return nullptr;
  PyObject* ret = PyList_New(count);
  for (unsigned int i = 0; i < count; ++i) {
    if (lens[i] >= 0) {
      PyList_SetItem(ret, i, PyUnicode_DecodeUTF8(rows[i], lens[i], "surrogateescape"));
    } else {
      Py_INCREF(Py_None);
      PyList_SetItem(ret, i, Py_None);
    }
  }
  return ret;
}
// host strings out of a Python list (None / non-str -> null row); the list keeps the bytes alive
inline void list_strings(PyObject* list, std::vector<const char*>& out) {
  const Py_ssize_t n = PyList_Size(list);
  out.resize((size_t)n);
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* e = PyList_GetItem(list, i);
    out[(size_t)i] = (e == Py_None || !PyUnicode_Check(e)) ? nullptr : PyUnicode_AsUTF8(e);
  }
}
// a buffer-or-address argument (create_from_offsets / create_offsets)
struct Region {
  Py_buffer view{};
  bool has_view = false;
  void* p = nullptr;
  explicit Region(PyObject* o) {
    if (o == Py_None) return;
    if (PyLong_Check(o)) p = PyLong_AsVoidPtr(o);
    else if (PyObject_CheckBuffer(o) && PyObject_GetBuffer(o, &view, PyBUF_SIMPLE) == 0) {
      has_view = true;
      p = view.buf;
    } else PyErr_Clear();
  }
  ~Region() {
    if (has_view) PyBuffer_Release(&view);
  }
};

}  // namespace pyni
