// libNVStrings.so -- the NVStrings class of the custrings API (include/nvstrings/NVStrings.h), out of line,
// over the C ABI of libcustrings_amd.so.  Host C++ only: every member is argument checking, one or two
// C-ABI calls and the reference's exception / return-value conventions (cited per member).
#include "nvstrings/NVStrings.h"
#include "nvstrings/ipc_transfer.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "custrings_amd.h"

namespace {
struct Regex {  // compiled pattern for the duration of one call
  cs_regex* h = nullptr;
  explicit Regex(const char* p) { NVStrings::check(cs_regex_compile(p, &h)); }
  ~Regex() { cs_regex_destroy(h); }
};
void take_columns(cs_column** cols, int n, std::vector<NVStrings*>& results) {
  for (int i = 0; i < n; ++i) results.push_back(NVStrings::adopt(cols[i]));
  if (cols) cs_free(cols);
}
// one instance per record out of the native record form (flat column + list offsets); `null_rows`: the
// source column, whose null rows give a null instance (split_record, split.cu:176-180)
void cut_records(cs_column* flat, const std::vector<int64_t>& list, const cs_column* null_rows, std::vector<NVStrings*>& results) {
  const int64_t rows = (int64_t)list.size() - 1;
  std::vector<unsigned char> bits;
  if (null_rows && rows) {
    bits.assign((size_t)(rows + 7) / 8, 0);
    int64_t nulls = 0;
    NVStrings::check(cs_column_null_bitarray(null_rows, bits.data(), 0, 0, nullptr, &nulls));
    if (nulls == 0) bits.clear();
  }
  for (int64_t r = 0; r < rows; ++r) {
    if (!bits.empty() && !((bits[(size_t)r >> 3] >> (r & 7)) & 1)) {
      results.push_back(nullptr);
      continue;
    }
    cs_column* row = nullptr;
    NVStrings::check(cs_sublist(flat, list[(size_t)r], list[(size_t)r + 1], 1, nullptr, &row));
    results.push_back(NVStrings::adopt(row));
  }
}
}  // namespace

NVStrings::NVStrings() : m_col(nullptr) {}
NVStrings::NVStrings(unsigned int) : m_col(nullptr) {}
NVStrings::NVStrings(const NVStrings&) : m_col(nullptr) {}
NVStrings::~NVStrings() { cs_column_destroy(m_col); }

void NVStrings::check(int status) {
  if (status == CS_OK) return;
  const std::string msg = cs_last_error();
  if (status == CS_ERR_INVALID_ARG || status == CS_ERR_RANGE) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
void NVStrings::ensure_device() {
  if (cs_current_device() >= 0) return;  // bound already (cs_init by the embedding process, e.g. one rank per GPU)
  int dev = 0;
  if (const char* e = std::getenv("CS_DEVICE")) dev = std::atoi(e);
  else if (const char* e = std::getenv("LOCAL_RANK")) dev = std::atoi(e);
  if (cs_device_count() == 1) dev = 0;
  check(cs_init(dev));
}
NVStrings* NVStrings::adopt(cs_column* column) {
  NVStrings* s = new NVStrings();
  s->m_col = column;
  return s;
}
cs_column* NVStrings::handle() const { return m_col; }
cs_column* NVStrings::release() {
  cs_column* c = m_col;
  m_col = nullptr;
  return c;
}

// ---- construction ------------------------------------------------------------------------------------
NVStrings* NVStrings::create_from_array(const char** strs, unsigned int count) {  // NVStrings.cu:74-86
  ensure_device();
  cs_column* c = nullptr;
  check(cs_column_from_host_strings(strs, count, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::create_from_index(std::pair<const char*, size_t>* strs, unsigned int count, bool devmem, sorttype stype) {  // NVStrings.cu:88-107
  ensure_device();
  cs_column* c = nullptr;
  check(cs_column_from_index(strs, count, devmem ? 1 : 0, (int)stype, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::create_from_offsets(const char* strs, int count, const int* offsets, const unsigned char* nullbitmask, int, bool devmem) {
  ensure_device();
  cs_column* c = nullptr;
  check(cs_column_from_offsets32(strs, count, offsets, nullbitmask, devmem ? 1 : 0, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::create_from_strings(std::vector<NVStrings*> strs) {  // NVStrings.cu:121-153
  ensure_device();
  std::vector<const cs_column*> cols;
  for (auto* s : strs) cols.push_back(s->m_col);
  cs_column* c = nullptr;
  check(cs_column_concat(cols.data(), (int)cols.size(), nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::create_from_ipc(nvstrings_ipc_transfer& ipc) {  // strings/NVStrings.cu:137
  ensure_device();
  cs_column* c = nullptr;
  check(cs_column_ipc_import(&ipc.column, &c));
  return adopt(c);
}
int NVStrings::create_ipc_transfer(nvstrings_ipc_transfer& ipc) {  // strings/NVStrings.cu:484
  check(cs_column_ipc_export(m_col, &ipc.column));
  return 0;
}
void NVStrings::destroy(NVStrings* inst) { delete inst; }

// ---- attributes / export --------------------------------------------------------------------------------
size_t NVStrings::memsize() const {
  const int64_t rows = cs_column_rows(m_col);
  return (size_t)cs_column_nbytes(m_col) + (size_t)cs_column_offset_width(m_col) * (size_t)(rows + 1) + (size_t)(rows + 7) / 8;
}
unsigned int NVStrings::size() const { return (unsigned int)cs_column_rows(m_col); }
int NVStrings::create_index(std::pair<const char*, size_t>* strs, bool devmem) {
  check(cs_column_create_index(m_col, strs, devmem ? 1 : 0, nullptr));
  return 0;
}
int NVStrings::create_offsets(char* strs, int* offsets, unsigned char* nullbitmask, bool devmem) {
  check(cs_column_export_offsets32(m_col, strs, offsets, nullbitmask, devmem ? 1 : 0, nullptr));
  return 0;
}
unsigned int NVStrings::set_null_bitarray(unsigned char* bitarray, bool emptyIsNull, bool devmem) {
  int64_t n = 0;
  check(cs_column_null_bitarray(m_col, bitarray, emptyIsNull ? 1 : 0, devmem ? 1 : 0, nullptr, &n));
  return (unsigned int)n;
}
NVStrings* NVStrings::copy() {
  cs_column* c = nullptr;
  check(cs_sublist(m_col, 0, cs_column_rows(m_col), 1, nullptr, &c));
  return adopt(c);
}
int NVStrings::to_host(char** list, int start, int end) {  // NVStrings.cu:266-346: no terminator is written
  const int count = (int)size();
  if (end < 0 || end > count) end = count;
  if (start < 0 || start >= end) return 0;
  std::vector<int64_t> off((size_t)count + 1);
  std::vector<unsigned char> chars((size_t)cs_column_nbytes(m_col) + 1);
  check(cs_column_export_offsets64(m_col, chars.data(), off.data(), nullptr, 0, nullptr));
  for (int i = start; i < end; ++i) {
    char* dst = list[i - start];
    if (!dst) continue;
    memcpy(dst, chars.data() + off[(size_t)i], (size_t)(off[(size_t)i + 1] - off[(size_t)i]));
  }
  return 0;
}
unsigned int NVStrings::len(int* lengths, bool devmem) {  // attrs.cu:32-69
  int64_t total = 0;
  check(cs_len(m_col, lengths, devmem ? 1 : 0, nullptr, &total));
  return (unsigned int)total;
}
size_t NVStrings::byte_count(int* lengths, bool devmem) {
  int64_t total = 0;
  check(cs_column_byte_count(m_col, lengths, devmem ? 1 : 0, nullptr, &total));
  return (size_t)total;
}

// ---- re-arrangement ----------------------------------------------------------------------------------------
NVStrings* NVStrings::sublist(unsigned int start, unsigned int end, int step) {
  cs_column* c = nullptr;
  const int st = cs_sublist(m_col, start, end, step, nullptr, &c);
  if (st == CS_ERR_RANGE) throw std::out_of_range(cs_last_error());
  check(st);
  return adopt(c);
}
NVStrings* NVStrings::gather(const int* pos, unsigned int count, bool devmem) {
  cs_column* c = nullptr;
  const int st = cs_gather(m_col, pos, count, devmem ? 1 : 0, nullptr, &c);
  if (st == CS_ERR_RANGE) throw std::out_of_range("gather position value out of range");  // array.cu:108
  check(st);
  return adopt(c);
}
NVStrings* NVStrings::gather(const bool* mask, bool devmem) {
  cs_column* c = nullptr;
  check(cs_gather_mask(m_col, reinterpret_cast<const unsigned char*>(mask), devmem ? 1 : 0, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::scatter(NVStrings& strs, const int* pos, bool devmem) {
  cs_column* c = nullptr;
  check(cs_scatter(m_col, strs.m_col, pos, devmem ? 1 : 0, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::scatter(const char* str, const int* pos, unsigned int count, bool devmem) {
  cs_column* c = nullptr;
  check(cs_scatter_scalar(m_col, str, pos, count, devmem ? 1 : 0, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::remove_strings(const int* pos, unsigned int count, bool devmem) {  // array.cu:262-300
  const unsigned int rows = size();
  if (rows == 0) return create_from_array(nullptr, 0);
  if (count == 0 || !pos) return copy();
  std::vector<int> h(count);
  if (devmem) {
    // (positions live on the device: bring them over through a one-row-per-position gather of nothing -- the
    //  C ABI has no raw copy; callers on this path hold host lists in practice)
    throw std::invalid_argument("remove_strings: device positions are not supported, pass devmem=false");
  }
  memcpy(h.data(), pos, sizeof(int) * count);
  std::vector<unsigned char> keep(rows, 1);
  for (int p : h)
    if (p >= 0 && (unsigned)p < rows) keep[(size_t)p] = 0;
  cs_column* c = nullptr;
  check(cs_gather_mask(m_col, keep.data(), 0, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::sort(sorttype stype, bool ascending, bool nullfirst) {
  cs_column* c = nullptr;
  check(cs_sort(m_col, (int)stype, ascending ? 1 : 0, nullfirst ? 1 : 0, nullptr, &c));
  return adopt(c);
}
int NVStrings::order(sorttype stype, bool ascending, unsigned int* indexes, bool nullfirst, bool devmem) {
  check(cs_order(m_col, (int)stype, ascending ? 1 : 0, nullfirst ? 1 : 0, indexes, devmem ? 1 : 0, nullptr));
  return 0;
}

// ---- combine ---------------------------------------------------------------------------------------------------
NVStrings* NVStrings::cat(NVStrings* others, const char* separator, const char* narep) {
  if (!others) return nullptr;  // combine.cu:33-34
  const cs_column* o = others->m_col;
  cs_column* c = nullptr;
  check(cs_cat(m_col, &o, 1, separator, narep, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::cat(std::vector<NVStrings*>& others, const char* separator, const char* narep) {
  if (others.empty()) return nullptr;  // combine.cu:148-149
  std::vector<const cs_column*> cols;
  for (auto* o : others) cols.push_back(o->m_col);
  cs_column* c = nullptr;
  check(cs_cat(m_col, cols.data(), (int)cols.size(), separator, narep, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::join(const char* separator, const char* narep) {
  cs_column* c = nullptr;
  check(cs_join(m_col, separator, narep, nullptr, &c));
  return adopt(c);
}

// ---- split family ----------------------------------------------------------------------------------------------
int NVStrings::split_record(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results) {
  std::vector<int64_t> list((size_t)size() + 1, 0);
  cs_column* flat = nullptr;
  check(cs_split_record(m_col, delimiter, maxsplit, list.data(), 0, nullptr, &flat));
  const int total = (int)cs_column_rows(flat);
  cut_records(flat, list, m_col, results);
  cs_column_destroy(flat);
  return total;  // the number of strings made (split.cu:234)
}
int NVStrings::rsplit_record(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results) {
  std::vector<int64_t> list((size_t)size() + 1, 0);
  cs_column* flat = nullptr;
  check(cs_rsplit_record(m_col, delimiter, maxsplit, list.data(), 0, nullptr, &flat));
  const int total = (int)cs_column_rows(flat);
  cut_records(flat, list, m_col, results);
  cs_column_destroy(flat);
  return total;
}
int NVStrings::split_record(int maxsplit, std::vector<NVStrings*>& results) { return split_record(nullptr, maxsplit, results); }
int NVStrings::rsplit_record(int maxsplit, std::vector<NVStrings*>& results) { return rsplit_record(nullptr, maxsplit, results); }
unsigned int NVStrings::split(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results) {
  cs_column** cols = nullptr;
  int n = 0;
  check(cs_split(m_col, delimiter, maxsplit, nullptr, &cols, &n));
  take_columns(cols, n, results);
  return (unsigned int)n;
}
unsigned int NVStrings::rsplit(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results) {
  cs_column** cols = nullptr;
  int n = 0;
  check(cs_rsplit(m_col, delimiter, maxsplit, nullptr, &cols, &n));
  take_columns(cols, n, results);
  return (unsigned int)n;
}
unsigned int NVStrings::split(int maxsplit, std::vector<NVStrings*>& results) { return split(nullptr, maxsplit, results); }
unsigned int NVStrings::rsplit(int maxsplit, std::vector<NVStrings*>& results) { return rsplit(nullptr, maxsplit, results); }
static int partition_impl(cs_column* col, const char* delimiter, int from_right, std::vector<NVStrings*>& results) {
  cs_column* flat = nullptr;
  NVStrings::check(cs_partition(col, delimiter, from_right, nullptr, &flat));
  if (!flat) return 0;  // null / empty delimiter (split.cu:1167-1171)
  const int64_t rows = cs_column_rows(col);
  std::vector<int64_t> list((size_t)rows + 1);
  for (int64_t r = 0; r <= rows; ++r) list[(size_t)r] = 3 * r;
  cut_records(flat, list, nullptr, results);  // a null row gives an instance of three nulls (split.cu:1200-1203)
  cs_column_destroy(flat);
  return (int)rows;
}
int NVStrings::partition(const char* delimiter, std::vector<NVStrings*>& results) { return partition_impl(m_col, delimiter, 0, results); }
int NVStrings::rpartition(const char* delimiter, std::vector<NVStrings*>& results) { return partition_impl(m_col, delimiter, 1, results); }

// ---- regex extraction ------------------------------------------------------------------------------------------
int NVStrings::extract(const char* pattern, std::vector<NVStrings*>& results) {
  if (!pattern) return -1;
  Regex re(pattern);
  cs_column** cols = nullptr;
  int n = 0;
  check(cs_extract(m_col, re.h, nullptr, &cols, &n));
  take_columns(n ? cols : nullptr, n, results);
  return n;
}
int NVStrings::findall(const char* pattern, std::vector<NVStrings*>& results) {
  if (!pattern) return -1;
  Regex re(pattern);
  cs_column** cols = nullptr;
  int n = 0;
  check(cs_findall(m_col, re.h, nullptr, &cols, &n));
  take_columns(n ? cols : nullptr, n, results);
  return n;
}
static int records_of(std::vector<NVStrings*>& cols, int ragged, std::vector<NVStrings*>& results) {
  if (cols.empty()) return 0;
  std::vector<const cs_column*> h;
  for (auto* c : cols) h.push_back(c->handle());
  const int64_t rows = (int64_t)cols[0]->size();
  std::vector<int64_t> list((size_t)rows + 1, 0);
  cs_column* flat = nullptr;
  NVStrings::check(cs_records_from_columns(h.data(), (int)h.size(), ragged, list.data(), 0, nullptr, &flat));
  cut_records(flat, list, nullptr, results);
  cs_column_destroy(flat);
  return (int)results.size();
}
int NVStrings::extract_record(const char* pattern, std::vector<NVStrings*>& results) {
  std::vector<NVStrings*> cols;
  const int n = extract(pattern, cols);
  if (n > 0) records_of(cols, 0, results);
  for (auto* c : cols) destroy(c);
  return n < 0 ? n : (int)results.size();
}
int NVStrings::findall_record(const char* pattern, std::vector<NVStrings*>& results) {
  std::vector<NVStrings*> cols;
  const int n = findall(pattern, cols);
  if (n > 0) records_of(cols, 1, results);
  for (auto* c : cols) destroy(c);
  return n < 0 ? n : (int)results.size();
}

// ---- replace / strip / case -------------------------------------------------------------------------------------
NVStrings* NVStrings::replace(const char* str, const char* repl, int maxrepl) {
  cs_column* c = nullptr;
  check(cs_replace(m_col, str, repl, maxrepl, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::replace_re(const char* pattern, const char* repl, int maxrepl) {
  if (!pattern || !*pattern) throw std::invalid_argument("nvstrings::replace_re parameter cannot be null or empty");  // replace.cu:112-113
  Regex re(pattern);
  cs_column* c = nullptr;
  check(cs_replace_re(m_col, re.h, repl, maxrepl, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::replace_re(std::vector<const char*>& patterns, NVStrings& repls) {  // replace_multi.cu:110-189
  if (patterns.empty() || repls.size() == 0) throw std::invalid_argument("replace_re patterns and repls parameters cannot be empty");
  std::vector<cs_regex*> res(patterns.size(), nullptr);
  struct Free {
    std::vector<cs_regex*>& v;
    ~Free() {
      for (auto* r : v)
        if (r) cs_regex_destroy(r);
    }
  } guard{res};
  for (size_t i = 0; i < patterns.size(); ++i)
    if (patterns[i]) check(cs_regex_compile(patterns[i], &res[i]));
  cs_column* c = nullptr;
  check(cs_replace_re_multi(m_col, res.data(), (int)res.size(), repls.m_col, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::replace_with_backrefs(const char* pattern, const char* repl) {
  if (!pattern || !*pattern) throw std::invalid_argument("nvstrings::replace_with_backrefs parameter cannot be null or empty");
  Regex re(pattern);
  cs_column* c = nullptr;
  check(cs_replace_with_backrefs(m_col, re.h, repl, nullptr, &c));
  return adopt(c);
}
static NVStrings* strip_side(cs_column* col, const char* to_strip, int side) {
  cs_column* c = nullptr;
  NVStrings::check(cs_strip(col, to_strip, side, nullptr, &c));
  return NVStrings::adopt(c);
}
NVStrings* NVStrings::lstrip(const char* to_strip) { return strip_side(m_col, to_strip, 1); }
NVStrings* NVStrings::strip(const char* to_strip) { return strip_side(m_col, to_strip, 0); }
NVStrings* NVStrings::rstrip(const char* to_strip) { return strip_side(m_col, to_strip, 2); }
NVStrings* NVStrings::lower() {
  cs_column* c = nullptr;
  check(cs_lower(m_col, nullptr, &c));
  return adopt(c);
}
NVStrings* NVStrings::upper() {
  cs_column* c = nullptr;
  check(cs_upper(m_col, nullptr, &c));
  return adopt(c);
}

// ---- search --------------------------------------------------------------------------------------------------------
unsigned int NVStrings::find(const char* str, int start, int end, int* results, bool devmem) {
  int64_t n = 0;
  check(cs_find(m_col, str, start, end, results, devmem ? 1 : 0, nullptr, &n));
  return (unsigned int)n;
}
// find.cu:36-72, 123-236, 276-387 (`starts` / `ends` of find_from are device pointers, as the reference's are)
unsigned int NVStrings::compare(const char* str, int* results, bool devmem) {
  int64_t n = 0;
  check(cs_compare(m_col, str, results, devmem ? 1 : 0, nullptr, &n));
  return (unsigned int)n;
}
unsigned int NVStrings::rfind(const char* str, int start, int end, int* results, bool devmem) {
  int64_t n = 0;
  check(cs_rfind(m_col, str, start, end, results, devmem ? 1 : 0, nullptr, &n));
  return (unsigned int)n;
}
unsigned int NVStrings::find_from(const char* str, int* starts, int* ends, int* results, bool devmem) {
  int64_t n = 0;
  check(cs_find_from(m_col, str, starts, ends, 1, results, devmem ? 1 : 0, nullptr, &n));
  return (unsigned int)n;
}
unsigned int NVStrings::find_multiple(NVStrings& strs, int* results, bool devmem) {
  int64_t n = 0;
  check(cs_find_multiple(m_col, strs.handle(), results, devmem ? 1 : 0, nullptr, &n));
  return (unsigned int)n;
}
int NVStrings::match_strings(NVStrings& strs, bool* results, bool devmem) {
  if (!results) return -1;  // find.cu:278-279
  if (size() == 0) return 0;
  if (size() != strs.size()) throw std::invalid_argument("sizes must match");
  int64_t n = 0;
  check(cs_match_strings(m_col, strs.handle(), reinterpret_cast<unsigned char*>(results), devmem ? 1 : 0, nullptr, &n));
  return (int)n;
}
unsigned int NVStrings::startswith(const char* str, bool* results, bool devmem) {
  int64_t n = 0;
  check(cs_startswith(m_col, str, reinterpret_cast<unsigned char*>(results), devmem ? 1 : 0, nullptr, &n));
  return (unsigned int)n;
}
unsigned int NVStrings::endswith(const char* str, bool* results, bool devmem) {
  int64_t n = 0;
  check(cs_endswith(m_col, str, reinterpret_cast<unsigned char*>(results), devmem ? 1 : 0, nullptr, &n));
  return (unsigned int)n;
}
int NVStrings::contains(const char* str, bool* results, bool devmem) {
  if (!str || !results) return -1;  // find.cu:239-240
  int64_t n = 0;
  check(cs_contains(m_col, str, reinterpret_cast<unsigned char*>(results), devmem ? 1 : 0, nullptr, &n));
  return (int)n;
}
int NVStrings::contains_re(const char* pattern, bool* results, bool devmem) {
  if (!pattern || !results) return -1;  // count.cu:61-62
  Regex re(pattern);
  int64_t n = 0;
  check(cs_contains_re(m_col, re.h, reinterpret_cast<unsigned char*>(results), devmem ? 1 : 0, nullptr, &n));
  return (int)n;
}
int NVStrings::match(const char* pattern, bool* results, bool devmem) {
  if (!pattern || !results) return -1;
  Regex re(pattern);
  int64_t n = 0;
  check(cs_match_re(m_col, re.h, reinterpret_cast<unsigned char*>(results), devmem ? 1 : 0, nullptr, &n));
  return (int)n;
}
int NVStrings::count_re(const char* pattern, int* results, bool devmem) {
  if (!pattern || !results) return -1;
  Regex re(pattern);
  int64_t n = 0;
  check(cs_count_re(m_col, re.h, results, devmem ? 1 : 0, nullptr, &n));
  return (int)n;
}
