/* NVStrings -- host C++ class of the custrings API over the MI355X back-end.
 *
 * Same class name, nested types, method names, argument types / defaults and ownership rules as
 * the reference's /root/reference/cpp/include/NVStrings.h (line cited per method), so that code
 * compiled against EITHER header links against libNVStrings.so built from this repository
 * (custrings_amd/host/NVStrings.cpp): the members are out-of-line and exported under the same
 * mangled names.  Instances are immutable; every transforming method returns a new heap instance
 * that the caller frees with NVStrings::destroy (NVStrings.h:52-57,156).
 * Error mapping: CS_ERR_INVALID_ARG -> std::invalid_argument, CS_ERR_RANGE ->
 * std::out_of_range where the reference throws that (gather), else std::invalid_argument; anything
 * else -> std::runtime_error; integer returns where the reference returns integers.
 * Members of the reference outside SURVEY.md section 8 are not declared.
 */
#ifndef NVSTRINGS_AMD_NVSTRINGS_H
#define NVSTRINGS_AMD_NVSTRINGS_H

#include <cstddef>
#include <utility>
#include <vector>

struct cs_column;
struct nvstrings_ipc_transfer; /* nvstrings/ipc_transfer.h */

class NVStrings {
  cs_column* m_col; /* (the reference holds an NVStringsImpl* here; one pointer either way) */

  /* ctors / dtor are private to control allocation (NVStrings.h:52-57) */
  NVStrings();
  NVStrings(unsigned int count);
  NVStrings(const NVStrings&);
  NVStrings& operator=(const NVStrings&) = delete;
  ~NVStrings();

 public:
  /* NVStrings.h:66-70 */
  enum sorttype { none = 0, length = 1, name = 2 };

  /* ---- construction (NVStrings.h:86-156) ---- */
  static NVStrings* create_from_array(const char** strs, unsigned int count);
  static NVStrings* create_from_index(std::pair<const char*, size_t>* strs, unsigned int count, bool devmem = true, sorttype stype = none);
  static NVStrings* create_from_offsets(const char* strs, int count, const int* offsets, const unsigned char* nullbitmask = 0, int nulls = 0,
                                        bool devmem = true);
  static NVStrings* create_from_strings(std::vector<NVStrings*> strs);
  static NVStrings* create_from_ipc(nvstrings_ipc_transfer& ipc); /* NVStrings.h:132: maps the exporter's buffers, no copy */
  static void destroy(NVStrings* inst);

  /* ---- attributes / export (NVStrings.h:162-354) ---- */
  size_t memsize() const;
  unsigned int size() const;
  int create_index(std::pair<const char*, size_t>* strs, bool devmem = true);
  int create_offsets(char* strs, int* offsets, unsigned char* nullbitmask = 0, bool devmem = true);
  int create_ipc_transfer(nvstrings_ipc_transfer& ipc); /* NVStrings.h:214 */
  unsigned int set_null_bitarray(unsigned char* bitarray, bool emptyIsNull = false, bool devmem = true);
  NVStrings* copy();
  int to_host(char** list, int start, int end);
  unsigned int len(int* lengths, bool devmem = true);
  size_t byte_count(int* lengths, bool devmem = true);

  /* ---- re-arrangement (NVStrings.h:261-334; array.cu) ---- */
  NVStrings* sublist(unsigned int start, unsigned int end, int step = 0);
  NVStrings* gather(const int* pos, unsigned int count, bool devmem = true);
  NVStrings* gather(const bool* mask, bool devmem = true);
  NVStrings* scatter(NVStrings& strs, const int* pos, bool devmem = true);
  NVStrings* scatter(const char* str, const int* pos, unsigned int count, bool devmem = true);
  NVStrings* remove_strings(const int* pos, unsigned int count, bool devmem = true);
  NVStrings* sort(sorttype stype = sorttype::name, bool ascending = true, bool nullfirst = true);
  int order(sorttype stype, bool ascending, unsigned int* indexes, bool nullfirst = true, bool devmem = true);

  /* ---- combine (NVStrings.h:437-452; combine.cu) ---- */
  NVStrings* cat(NVStrings* others, const char* separator, const char* narep = nullptr);
  NVStrings* cat(std::vector<NVStrings*>& others, const char* separator, const char* narep = nullptr);
  NVStrings* join(const char* separator = "", const char* narep = nullptr);

  /* ---- split family (NVStrings.h:464-552; split.cu) ---- */
  int split_record(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results);
  int rsplit_record(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results);
  int split_record(int maxsplit, std::vector<NVStrings*>& results);
  int rsplit_record(int maxsplit, std::vector<NVStrings*>& results);
  unsigned int split(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results);
  unsigned int rsplit(const char* delimiter, int maxsplit, std::vector<NVStrings*>& results);
  unsigned int split(int maxsplit, std::vector<NVStrings*>& results);
  unsigned int rsplit(int maxsplit, std::vector<NVStrings*>& results);
  int partition(const char* delimiter, std::vector<NVStrings*>& results);
  int rpartition(const char* delimiter, std::vector<NVStrings*>& results);

  /* ---- regex extraction (NVStrings.h:682-693,943-952) ---- */
  int extract(const char* pattern, std::vector<NVStrings*>& results);
  int extract_record(const char* pattern, std::vector<NVStrings*>& results);
  int findall(const char* pattern, std::vector<NVStrings*>& results);
  int findall_record(const char* pattern, std::vector<NVStrings*>& results);

  /* ---- replace / strip / case (NVStrings.h:714-820) ---- */
  NVStrings* replace(const char* str, const char* repl, int maxrepl = -1);
  NVStrings* replace_re(const char* pattern, const char* repl, int maxrepl = -1);
  NVStrings* replace_re(std::vector<const char*>& patterns, NVStrings& repls);
  NVStrings* replace_with_backrefs(const char* pattern, const char* repl);
  NVStrings* lstrip(const char* to_strip);
  NVStrings* strip(const char* to_strip);
  NVStrings* rstrip(const char* to_strip);
  NVStrings* lower();
  NVStrings* upper();

  /* ---- search (NVStrings.h:861-981) ---- */
  unsigned int find(const char* str, int start, int end, int* results, bool devmem = true);
  int contains(const char* str, bool* results, bool devmem = true);
  // the rest of the find family (reference NVStrings.h:849-934; find.cu)
  unsigned int compare(const char* str, int* results, bool devmem = true);
  unsigned int rfind(const char* str, int start, int end, int* results, bool devmem = true);
  unsigned int find_from(const char* str, int* starts, int* ends, int* results, bool devmem = true);
  unsigned int find_multiple(NVStrings& strs, int* results, bool devmem = true);
  int match_strings(NVStrings& strs, bool* results, bool devmem = true);
  unsigned int startswith(const char* str, bool* results, bool devmem = true);
  unsigned int endswith(const char* str, bool* results, bool devmem = true);
  int contains_re(const char* pattern, bool* results, bool devmem = true);
  int match(const char* pattern, bool* results, bool devmem = true);
  int count_re(const char* pattern, int* results, bool devmem = true);

  /* ---- not in the reference: the bridge to the C ABI (used by libNVCategory / libNVText and by callers that
   * want the native record form or the engine's handles) ---- */
  static NVStrings* adopt(cs_column* column); /* takes ownership of a handle produced by the C ABI */
  cs_column* handle() const;
  cs_column* release();              /* gives the handle back to the caller: the instance is empty afterwards */
  static void check(int status);     /* cs_status -> the exceptions listed above */
  static void ensure_device();       /* binds the process to a GPU once (the device cs_init chose, else LOCAL_RANK, else 0) */
};

#endif
