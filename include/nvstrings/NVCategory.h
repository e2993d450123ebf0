/* NVCategory -- dictionary encoding (sorted unique keys + one int32 value per row), the members of
 * /root/reference/cpp/include/NVCategory.h that SURVEY.md section 8 names, over the C ABI.  Out-of-line,
 * exported by libNVCategory.so (custrings_amd/host/NVCategory.cpp) under the reference's mangled names. */
#ifndef NVSTRINGS_AMD_NVCATEGORY_H
#define NVSTRINGS_AMD_NVCATEGORY_H

#include <cstddef>
#include <utility>
#include <vector>

#include "base_category.h"

struct cs_category;
struct nvcategory_ipc_transfer; /* nvstrings/ipc_transfer.h */
class NVStrings;

class NVCategory : base_category_type { /* NVCategory.h:48: the object begins with the vtable pointer */
  cs_category* m_cat;
  NVCategory();
  NVCategory(const NVCategory&);
  ~NVCategory();
  NVCategory& operator=(const NVCategory&) = delete;

 public:
  /* NVCategory.h:71-138 */
  static NVCategory* create_from_array(const char** strs, unsigned int count);
  static NVCategory* create_from_index(std::pair<const char*, size_t>* strs, unsigned int count, bool devmem = true);
  static NVCategory* create_from_offsets(const char* strs, unsigned int count, const int* offsets, const unsigned char* nullbitmask = 0, int nulls = 0,
                                         bool devmem = true);
  static NVCategory* create_from_strings(NVStrings& strs);
  static NVCategory* create_from_strings(std::vector<NVStrings*>& strs);
  static NVCategory* create_from_categories(std::vector<NVCategory*>& cats);
  static NVCategory* create_from_ipc(nvcategory_ipc_transfer& ipc); /* NVCategory.h:128 */
  int create_ipc_transfer(nvcategory_ipc_transfer& ipc);            /* NVCategory.h:176 */
  static void destroy(NVCategory* inst);
  const char* get_type_name(); /* NVCategory.h:143, NVCategory.cu:581: "custring" */
  /* NVCategory.h:148-249 */
  unsigned int size();
  unsigned int keys_size();
  bool has_nulls();
  NVCategory* copy();
  NVStrings* get_keys();
  int get_value(unsigned int index);
  int get_value(const char* str);
  int get_values(int* results, bool devmem = true);
  const int* values_cptr();
  int get_indexes_for(unsigned int index, int* results, bool devmem = true);
  int get_indexes_for(const char* str, int* results, bool devmem = true);
  /* NVCategory.h:258-350: the remap family (NVCategory.cu:926-1822) */
  NVCategory* add_strings(NVStrings& strs);
  NVCategory* remove_strings(NVStrings& strs);
  NVCategory* add_keys_and_remap(NVStrings& strs);
  NVCategory* remove_keys_and_remap(NVStrings& strs);
  NVCategory* set_keys_and_remap(NVStrings& strs);
  NVCategory* remove_unused_keys_and_remap();
  NVCategory* merge_category(NVCategory& cat);
  NVCategory* merge_and_remap(NVCategory& cat);
  NVStrings* to_strings();
  NVStrings* gather_strings(const int* pos, unsigned int elems, bool devmem = true);
  NVCategory* gather_and_remap(const int* pos, unsigned int elems, bool devmem = true);
  NVCategory* gather(const int* pos, unsigned int elems, bool devmem = true);

  /* not in the reference: the bridge to the C ABI */
  static NVCategory* adopt(cs_category* cat);
  cs_category* handle() const;
  cs_category* release(); /* gives the handle back to the caller: the instance is empty afterwards */
};

#endif
