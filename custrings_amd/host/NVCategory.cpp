// libNVCategory.so -- the NVCategory class (include/nvstrings/NVCategory.h), out of line, over the C ABI.
#include "nvstrings/NVCategory.h"

#include <stdexcept>
#include <string>

#include "custrings_amd.h"
#include "nvstrings/NVStrings.h"
#include "nvstrings/ipc_transfer.h"

namespace {
void check_range(int status) {  // the gather family throws std::out_of_range (NVCategory.cu:1067,1101,1161)
  if (status == CS_ERR_RANGE) throw std::out_of_range(cs_last_error());
  NVStrings::check(status);
}
}  // namespace

NVCategory::NVCategory() : m_cat(nullptr) {}
NVCategory::NVCategory(const NVCategory&) : m_cat(nullptr) {}
NVCategory::~NVCategory() { cs_category_destroy(m_cat); }
NVCategory* NVCategory::adopt(cs_category* cat) {
  NVCategory* c = new NVCategory();
  c->m_cat = cat;
  return c;
}
cs_category* NVCategory::handle() const { return m_cat; }
const char* NVCategory::get_type_name() { return "custring"; }  // NVCategory.cu:581 (the key function: the vtable is emitted here)
NVCategory* NVCategory::create_from_ipc(nvcategory_ipc_transfer& ipc) {  // NVCategory.cu:373
  NVStrings::ensure_device();
  cs_category* c = nullptr;
  NVStrings::check(cs_category_ipc_import(&ipc.category, &c));
  return adopt(c);
}
int NVCategory::create_ipc_transfer(nvcategory_ipc_transfer& ipc) {  // NVCategory.cu:715
  NVStrings::check(cs_category_ipc_export(m_cat, &ipc.category));
  return 0;
}
cs_category* NVCategory::release() {
  cs_category* c = m_cat;
  m_cat = nullptr;
  return c;
}

NVCategory* NVCategory::create_from_strings(NVStrings& strs) {
  cs_category* c = nullptr;
  NVStrings::check(cs_category_build(strs.handle(), nullptr, &c));
  return adopt(c);
}
NVCategory* NVCategory::create_from_strings(std::vector<NVStrings*>& strs) {  // one category over all rows, in order
  NVStrings* all = NVStrings::create_from_strings(strs);
  NVCategory* r = create_from_strings(*all);
  NVStrings::destroy(all);
  return r;
}
NVCategory* NVCategory::create_from_array(const char** strs, unsigned int count) {
  NVStrings* s = NVStrings::create_from_array(strs, count);
  NVCategory* r = create_from_strings(*s);
  NVStrings::destroy(s);
  return r;
}
NVCategory* NVCategory::create_from_index(std::pair<const char*, size_t>* strs, unsigned int count, bool devmem) {
  NVStrings* s = NVStrings::create_from_index(strs, count, devmem);
  NVCategory* r = create_from_strings(*s);
  NVStrings::destroy(s);
  return r;
}
NVCategory* NVCategory::create_from_offsets(const char* strs, unsigned int count, const int* offsets, const unsigned char* nullbitmask, int nulls,
                                            bool devmem) {
  NVStrings* s = NVStrings::create_from_offsets(strs, (int)count, offsets, nullbitmask, nulls, devmem);
  NVCategory* r = create_from_strings(*s);
  NVStrings::destroy(s);
  return r;
}
NVCategory* NVCategory::create_from_categories(std::vector<NVCategory*>& cats) {  // NVCategory.cu:430-514
  std::vector<const cs_category*> h;
  for (auto* c : cats) h.push_back(c->m_cat);
  cs_category* out = nullptr;
  NVStrings::check(cs_category_merge(h.data(), (int)h.size(), nullptr, &out));
  return adopt(out);
}
void NVCategory::destroy(NVCategory* inst) { delete inst; }

unsigned int NVCategory::size() { return (unsigned int)cs_category_size(m_cat); }
unsigned int NVCategory::keys_size() { return (unsigned int)cs_category_keys_size(m_cat); }
bool NVCategory::has_nulls() {  // the null key, when present, is key 0 (NVCategory.cu:605-615)
  NVStrings* k = get_keys();
  unsigned char bits[1] = {0xFF};
  bool r = false;
  if (k->size()) {
    std::vector<unsigned char> all((k->size() + 7) / 8);
    r = k->set_null_bitarray(all.data(), false, false) > 0;
    (void)bits;
  }
  NVStrings::destroy(k);
  return r;
}
NVCategory* NVCategory::copy() {
  std::vector<NVCategory*> one{this};
  return create_from_categories(one);
}
NVStrings* NVCategory::get_keys() {
  cs_column* k = nullptr;
  NVStrings::check(cs_category_keys(m_cat, &k));
  return NVStrings::adopt(k);
}
int NVCategory::get_values(int* results, bool devmem) {
  NVStrings::check(cs_category_get_values(m_cat, results, devmem ? 1 : 0, nullptr));
  return (int)size();
}
const int* NVCategory::values_cptr() { return cs_category_values_ptr(m_cat); }
int NVCategory::get_value(unsigned int index) {  // NVCategory.cu:754-764
  if (index >= size()) return -1;
  std::vector<int> v(size());
  get_values(v.data(), false);
  return v[index];
}
int NVCategory::get_value(const char* str) {  // NVCategory.cu:766-864: index of the key, -1 when absent
  NVStrings* k = get_keys();
  const unsigned int n = k->size();
  int found = -1;
  if (n) {
    std::vector<int> lens(n);
    const size_t total = k->byte_count(lens.data(), false);
    std::vector<char> buf(total + n + 1, 0);
    std::vector<char*> ptrs(n);
    size_t off = 0;
    for (unsigned int i = 0; i < n; ++i) {
      ptrs[i] = buf.data() + off;
      off += (size_t)(lens[i] > 0 ? lens[i] : 0) + 1;
    }
    k->to_host(ptrs.data(), 0, (int)n);
    for (unsigned int i = 0; i < n && found < 0; ++i) {
      if (!str) found = lens[i] < 0 ? (int)i : -1;
      else if (lens[i] >= 0 && std::string(ptrs[i], (size_t)lens[i]) == str) found = (int)i;
    }
  }
  NVStrings::destroy(k);
  return found;
}
int NVCategory::get_indexes_for(unsigned int index, int* results, bool devmem) {  // NVCategory.cu:885-915
  if (devmem) throw std::invalid_argument("get_indexes_for: pass devmem=false");
  std::vector<int> v(size());
  if (!v.empty()) get_values(v.data(), false);
  int n = 0;
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == (int)index) {
      if (results) results[n] = (int)i;
      ++n;
    }
  return n;
}
int NVCategory::get_indexes_for(const char* str, int* results, bool devmem) {
  const int id = get_value(str);
  if (id < 0) return id;
  return get_indexes_for((unsigned int)id, results, devmem);
}

#define CAT_OP(fn, ...)                                  \
  cs_category* out = nullptr;                            \
  check_range(fn(m_cat, __VA_ARGS__, nullptr, &out));    \
  return adopt(out)
NVCategory* NVCategory::add_strings(NVStrings& strs) { CAT_OP(cs_category_add_strings, strs.handle()); }
NVCategory* NVCategory::remove_strings(NVStrings& strs) { CAT_OP(cs_category_remove_strings, strs.handle()); }
NVCategory* NVCategory::add_keys_and_remap(NVStrings& strs) { CAT_OP(cs_category_add_keys, strs.handle()); }
NVCategory* NVCategory::remove_keys_and_remap(NVStrings& strs) { CAT_OP(cs_category_remove_keys, strs.handle()); }
NVCategory* NVCategory::set_keys_and_remap(NVStrings& strs) { CAT_OP(cs_category_set_keys, strs.handle()); }
NVCategory* NVCategory::merge_category(NVCategory& cat) { CAT_OP(cs_category_merge_category, cat.m_cat); }
NVCategory* NVCategory::gather_and_remap(const int* pos, unsigned int elems, bool devmem) { CAT_OP(cs_category_gather_and_remap, pos, elems, devmem ? 1 : 0); }
NVCategory* NVCategory::gather(const int* pos, unsigned int elems, bool devmem) { CAT_OP(cs_category_gather, pos, elems, devmem ? 1 : 0); }
#undef CAT_OP
NVCategory* NVCategory::remove_unused_keys_and_remap() {
  cs_category* out = nullptr;
  NVStrings::check(cs_category_remove_unused_keys(m_cat, nullptr, &out));
  return adopt(out);
}
NVCategory* NVCategory::merge_and_remap(NVCategory& cat) {  // NVCategory.cu:1339-1345
  std::vector<NVCategory*> two{this, &cat};
  return create_from_categories(two);
}
NVStrings* NVCategory::to_strings() {
  cs_column* c = nullptr;
  NVStrings::check(cs_category_to_strings(m_cat, nullptr, &c));
  return c ? NVStrings::adopt(c) : nullptr;
}
NVStrings* NVCategory::gather_strings(const int* pos, unsigned int elems, bool devmem) {
  cs_column* c = nullptr;
  check_range(cs_category_gather_strings(m_cat, pos, elems, devmem ? 1 : 0, nullptr, &c));
  return NVStrings::adopt(c);
}
