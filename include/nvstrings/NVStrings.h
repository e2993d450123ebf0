/* NVStrings -- host C++ class of the custrings API, hot-path subset, implemented
 * over the C ABI of the MI355X back-end (include/custrings_amd.h).
 *
 * Same class name, method names, argument order / defaults and ownership rules
 * as the reference's /root/reference/cpp/include/NVStrings.h (line cited per
 * method), so code written against that header for the hot path recompiles
 * against this one.  Header-only: link libcustrings_amd.so.  Instances are
 * immutable; every transforming method returns a new heap instance that the
 * caller frees with NVStrings::destroy (NVStrings.h:52-57,156).
 * Error mapping: CS_ERR_INVALID_ARG -> std::invalid_argument,
 * anything else -> std::runtime_error; integer returns where the reference
 * returns integers.  Methods outside the hot path are not declared.
 */
#ifndef NVSTRINGS_AMD_NVSTRINGS_H
#define NVSTRINGS_AMD_NVSTRINGS_H

#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#include "../custrings_amd.h"

class NVStrings {
  cs_column* m_col;
  explicit NVStrings(cs_column* c) : m_col(c) {}
  ~NVStrings() { cs_column_destroy(m_col); }
  NVStrings(const NVStrings&) = delete;
  NVStrings& operator=(const NVStrings&) = delete;

 public:
  static void check(int status) {
    if (status == CS_OK) return;
    std::string msg = cs_last_error();
    if (status == CS_ERR_INVALID_ARG || status == CS_ERR_RANGE) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
  }
  static void ensure_device() {
    static bool ok = false;
    if (!ok) {
      check(cs_init(0));
      ok = true;
    }
  }
  /* adopts a handle produced by the C ABI (used by NVCategory / NVText) */
  static NVStrings* adopt(cs_column* c) { return new NVStrings(c); }
  cs_column* handle() const { return m_col; }

  /* NVStrings.h:86 */
  static NVStrings* create_from_array(const char** strs, unsigned int count) {
    ensure_device();
    cs_column* c = nullptr;
    check(cs_column_from_host_strings(strs, count, nullptr, &c));
    return new NVStrings(c);
  }
  /* NVStrings.h:116 */
  static NVStrings* create_from_offsets(const char* strs, int count, const int* offsets,
                                        const unsigned char* nullbitmask = 0, int nulls = 0, bool devmem = true) {
    (void)nulls;
    ensure_device();
    cs_column* c = nullptr;
    check(cs_column_from_offsets32(strs, count, offsets, nullbitmask, devmem ? 1 : 0, nullptr, &c));
    return new NVStrings(c);
  }
  /* NVStrings.h:125 */
  static NVStrings* create_from_strings(std::vector<NVStrings*> strs) {
    ensure_device();
    std::vector<const cs_column*> cols;
    for (auto* s : strs) cols.push_back(s->m_col);
    cs_column* c = nullptr;
    check(cs_column_concat(cols.data(), (int)cols.size(), nullptr, &c));
    return new NVStrings(c);
  }
  /* NVStrings.h:156 */
  static void destroy(NVStrings* inst) { delete inst; }

  /* NVStrings.h:167 */
  unsigned int size() const { return (unsigned int)cs_column_rows(m_col); }
  /* NVStrings.h:207 -- returns 0 */
  int create_offsets(char* strs, int* offsets, unsigned char* nullbitmask = 0, bool devmem = true) {
    check(cs_column_export_offsets32(m_col, strs, offsets, nullbitmask, devmem ? 1 : 0, nullptr));
    return 0;
  }
  /* NVStrings.h:225 -- returns the number of nulls */
  unsigned int set_null_bitarray(unsigned char* bitarray, bool emptyIsNull = false, bool devmem = true) {
    int64_t n = 0;
    check(cs_column_null_bitarray(m_col, bitarray, emptyIsNull ? 1 : 0, devmem ? 1 : 0, nullptr, &n));
    return (unsigned int)n;
  }
  /* NVStrings.h:354 -- bytes per row (-1 for null), returns the total */
  size_t byte_count(int* lengths, bool devmem = true) {
    int64_t total = 0;
    check(cs_column_byte_count(m_col, lengths, devmem ? 1 : 0, nullptr, &total));
    return (size_t)total;
  }
  /* NVStrings.h:251 -- copies rows [start,end) into caller-allocated host buffers
   * (no terminator is written, as in the reference); end < 0 = all. Returns 0. */
  int to_host(char** list, int start, int end) {
    const int count = (int)size();
    if (end < 0 || end > count) end = count;
    if (start < 0 || start >= end) return 0;
    std::vector<int64_t> off((size_t)count + 1);
    std::vector<unsigned char> chars((size_t)cs_column_nbytes(m_col) + 1);
    check(cs_column_export_offsets64(m_col, chars.data(), off.data(), nullptr, 0, nullptr));
    for (int i = start; i < end; ++i) {
      char* dst = list[i - start];
      if (!dst) continue;
      for (int64_t k = off[i]; k < off[i + 1]; ++k) *dst++ = (char)chars[(size_t)k];
    }
    return 0;
  }

  /* columns -> one NVStrings per row (ragged: each row's leading non-null columns) */
  static void records(const std::vector<NVStrings*>& cols, int ragged, std::vector<NVStrings*>& results) {
    if (cols.empty()) return;
    std::vector<const cs_column*> h;
    for (auto* c : cols) h.push_back(c->m_col);
    const int64_t rows = (int64_t)cols[0]->size();
    std::vector<int64_t> list((size_t)rows + 1, 0);
    cs_column* flat = nullptr;
    check(cs_records_from_columns(h.data(), (int)h.size(), ragged, list.data(), 0, nullptr, &flat));
    for (int64_t r = 0; r < rows; ++r) {
      cs_column* row = nullptr;
      check(cs_column_slice(flat, list[(size_t)r], list[(size_t)r + 1] - list[(size_t)r], nullptr, &row));
      results.push_back(new NVStrings(row));
    }
    cs_column_destroy(flat);
  }

  /* NVStrings.h:504 -- column-major split on a delimiter; returns the column count */
  unsigned int split(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results) {
    cs_column** cols = nullptr;
    int n = 0;
    check(cs_split(m_col, delimiter, maxsplit, nullptr, &cols, &n));
    for (int i = 0; i < n; ++i) results.push_back(new NVStrings(cols[i]));
    cs_free(cols);
    return (unsigned int)n;
  }
  /* NVStrings.h:524 -- whitespace split */
  unsigned int split(int maxsplit, std::vector<NVStrings*>& results) { return split(nullptr, maxsplit, results); }

  /* NVStrings.h:514,534 -- column-major split with the tokens located from the right */
  unsigned int rsplit(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results) {
    cs_column** cols = nullptr;
    int n = 0;
    check(cs_rsplit(m_col, delimiter, maxsplit, nullptr, &cols, &n));
    for (int i = 0; i < n; ++i) results.push_back(new NVStrings(cols[i]));
    cs_free(cols);
    return (unsigned int)n;
  }
  unsigned int rsplit(int maxsplit, std::vector<NVStrings*>& results) { return rsplit(nullptr, maxsplit, results); }

  /* NVStrings.h:714 */
  NVStrings* replace(const char* str, const char* repl, int maxrepl = -1) {
    cs_column* c = nullptr;
    check(cs_replace(m_col, str, repl, maxrepl, nullptr, &c));
    return new NVStrings(c);
  }
  /* NVStrings.h:766 */
  NVStrings* replace_re(const char* pattern, const char* repl, int maxrepl = -1) {
    if (!pattern || !*pattern) throw std::invalid_argument("nvstrings::replace_re parameter cannot be null or empty");
    Regex re(pattern);
    cs_column* c = nullptr;
    check(cs_replace_re(m_col, re.h, repl, maxrepl, nullptr, &c));
    return new NVStrings(c);
  }
  /* NVStrings.h:682 -- one instance per capture group; returns the group count (-1: null pattern) */
  int extract(const char* pattern, std::vector<NVStrings*>& results) {
    if (!pattern) return -1;
    Regex re(pattern);
    cs_column** cols = nullptr;
    int n = 0;
    check(cs_extract(m_col, re.h, nullptr, &cols, &n));
    for (int i = 0; i < n; ++i) results.push_back(new NVStrings(cols[i]));
    if (n) cs_free(cols);
    return n;
  }
  /* NVStrings.h:943 -- column k = every row's k-th match; returns the column count (-1: null pattern) */
  int findall(const char* pattern, std::vector<NVStrings*>& results) {
    if (!pattern) return -1;
    Regex re(pattern);
    cs_column** cols = nullptr;
    int n = 0;
    check(cs_findall(m_col, re.h, nullptr, &cols, &n));
    for (int i = 0; i < n; ++i) results.push_back(new NVStrings(cols[i]));
    if (n) cs_free(cols);
    return n;
  }
  /* NVStrings.h:693,952 -- row-major forms: one instance per row, cut out of the native record column
   * (cs_records_from_columns: one flat column + list offsets) */
  int extract_record(const char* pattern, std::vector<NVStrings*>& results) {
    std::vector<NVStrings*> cols;
    const int n = extract(pattern, cols);
    if (n > 0) records(cols, 0, results);
    for (auto* c : cols) destroy(c);
    return n < 0 ? n : (int)results.size();
  }
  int findall_record(const char* pattern, std::vector<NVStrings*>& results) {
    std::vector<NVStrings*> cols;
    const int n = findall(pattern, cols);
    if (n > 0) records(cols, 1, results);
    for (auto* c : cols) destroy(c);
    return n < 0 ? n : (int)results.size();
  }
  /* NVStrings.h:788 */
  NVStrings* replace_with_backrefs(const char* pattern, const char* repl) {
    if (!pattern || !*pattern) throw std::invalid_argument("nvstrings::replace_with_backrefs parameter cannot be null or empty");
    Regex re(pattern);
    cs_column* c = nullptr;
    check(cs_replace_with_backrefs(m_col, re.h, repl, nullptr, &c));
    return new NVStrings(c);
  }
  /* NVStrings.h:796-808 */
  NVStrings* lstrip(const char* to_strip) { return strip_side(to_strip, 1); }
  NVStrings* strip(const char* to_strip) { return strip_side(to_strip, 0); }
  NVStrings* rstrip(const char* to_strip) { return strip_side(to_strip, 2); }
  /* NVStrings.h:815,822 */
  NVStrings* lower() {
    cs_column* c = nullptr;
    check(cs_lower(m_col, nullptr, &c));
    return new NVStrings(c);
  }
  NVStrings* upper() {
    cs_column* c = nullptr;
    check(cs_upper(m_col, nullptr, &c));
    return new NVStrings(c);
  }
  /* NVStrings.h:861 -- returns the number of rows whose result is not -1 */
  unsigned int find(const char* str, int start, int end, int* results, bool todevice = true) {
    int64_t n = 0;
    check(cs_find(m_col, str, start, end, results, todevice ? 1 : 0, nullptr, &n));
    return (unsigned int)n;
  }
  /* NVStrings.h:907 -- returns the number of matches, -1 on null arguments */
  int contains(const char* str, bool* results, bool todevice = true) {
    if (!str || !results) return -1;
    int64_t n = 0;
    check(cs_contains(m_col, str, reinterpret_cast<uint8_t*>(results), todevice ? 1 : 0, nullptr, &n));
    return (int)n;
  }
  /* NVStrings.h:963,975,988 */
  int contains_re(const char* pattern, bool* results, bool todevice = true) {
    if (!pattern || !results) return -1;
    Regex re(pattern);
    int64_t n = 0;
    check(cs_contains_re(m_col, re.h, reinterpret_cast<uint8_t*>(results), todevice ? 1 : 0, nullptr, &n));
    return (int)n;
  }
  int match(const char* pattern, bool* results, bool todevice = true) {
    if (!pattern || !results) return -1;
    Regex re(pattern);
    int64_t n = 0;
    check(cs_match_re(m_col, re.h, reinterpret_cast<uint8_t*>(results), todevice ? 1 : 0, nullptr, &n));
    return (int)n;
  }
  int count_re(const char* pattern, int* results, bool todevice = true) {
    if (!pattern || !results) return -1;
    Regex re(pattern);
    int64_t n = 0;
    check(cs_count_re(m_col, re.h, results, todevice ? 1 : 0, nullptr, &n));
    return (int)n;
  }

 private:
  struct Regex {
    cs_regex* h = nullptr;
    explicit Regex(const char* p) { check(cs_regex_compile(p, &h)); }
    ~Regex() { cs_regex_destroy(h); }
  };
  NVStrings* strip_side(const char* to_strip, int side) {
    cs_column* c = nullptr;
    check(cs_strip(m_col, to_strip, side, nullptr, &c));
    return new NVStrings(c);
  }
};

#endif
