/* base_category_type -- the polymorphic root the reference gives its category classes
 * (/root/reference/cpp/include/base_category.h:18-23): callers that hold a category by this base -- the reference's
 * python/cpp/numeric_category.cpp:190-236 casts the handle to it and dispatches on get_type_name() -- read the object's
 * first word as a vtable pointer, so NVCategory must have one in the same place. */
#ifndef NVSTRINGS_AMD_BASE_CATEGORY_H
#define NVSTRINGS_AMD_BASE_CATEGORY_H

class base_category_type {
 public:
  virtual const char* get_type_name() = 0;
  virtual ~base_category_type() {}
};

#endif
