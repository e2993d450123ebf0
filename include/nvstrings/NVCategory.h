/* NVCategory -- dictionary encoding (sorted unique keys + int32 values), hot-path
 * subset of /root/reference/cpp/include/NVCategory.h over the C ABI.  Header-only. */
#ifndef NVSTRINGS_AMD_NVCATEGORY_H
#define NVSTRINGS_AMD_NVCATEGORY_H

#include <vector>

#include "NVStrings.h"

class NVCategory {
  cs_category* m_cat;
  explicit NVCategory(cs_category* c) : m_cat(c) {}
  ~NVCategory() { cs_category_destroy(m_cat); }
  NVCategory(const NVCategory&) = delete;

 public:
  /* NVCategory.h:107 */
  static NVCategory* create_from_strings(NVStrings& strs) {
    cs_category* c = nullptr;
    NVStrings::check(cs_category_build(strs.handle(), nullptr, &c));
    return new NVCategory(c);
  }
  /* NVCategory.h:114 -- one category over the rows of all instances, in order */
  static NVCategory* create_from_strings(std::vector<NVStrings*>& strs) {
    NVStrings* all = NVStrings::create_from_strings(strs);
    NVCategory* r = create_from_strings(*all);
    NVStrings::destroy(all);
    return r;
  }
  /* NVCategory.h:101 */
  static NVCategory* create_from_offsets(const char* strs, unsigned int count, const int* offsets,
                                         const unsigned char* nullbitmask = 0, int nulls = 0, bool devmem = true) {
    NVStrings* s = NVStrings::create_from_offsets(strs, (int)count, offsets, nullbitmask, nulls, devmem);
    NVCategory* r = create_from_strings(*s);
    NVStrings::destroy(s);
    return r;
  }
  /* NVCategory.h:121 -- merged key set, concatenated remapped values */
  static NVCategory* create_from_categories(std::vector<NVCategory*>& cats) {
    std::vector<const cs_category*> h;
    for (auto* c : cats) h.push_back(c->m_cat);
    cs_category* out = nullptr;
    NVStrings::check(cs_category_merge(h.data(), (int)h.size(), nullptr, &out));
    return new NVCategory(out);
  }
  /* NVCategory.h:138 */
  static void destroy(NVCategory* inst) { delete inst; }
  /* NVCategory.h:148,152 */
  unsigned int size() { return (unsigned int)cs_category_size(m_cat); }
  unsigned int keys_size() { return (unsigned int)cs_category_keys_size(m_cat); }
  /* NVCategory.h:197 -- new instance, caller destroys */
  NVStrings* get_keys() {
    cs_column* k = nullptr;
    NVStrings::check(cs_category_keys(m_cat, &k));
    return NVStrings::adopt(k);
  }
  /* NVCategory.h:225 -- returns the number of values */
  int get_values(int* results, bool devmem = true) {
    NVStrings::check(cs_category_get_values(m_cat, results, devmem ? 1 : 0, nullptr));
    return (int)size();
  }
  /* NVCategory.h:232 -- device pointer into the instance */
  const int* values_cptr() { return cs_category_values_ptr(m_cat); }
};

#endif
