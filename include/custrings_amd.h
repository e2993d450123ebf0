/* custrings_amd.h -- C ABI of the MI355X-native columnar string engine.
 *
 * This is the drop-in boundary for the hot path of rapidsai/custrings: the
 * entry points below are what the reference's host classes (NVStrings /
 * NVCategory / NVText, /root/reference/cpp/include/) need from a device
 * back-end for split / find / contains / replace / strip / lower,
 * contains_re / replace_re, the category key build and tokenize / ngrams.
 * Each function cites the reference interface it stands in for.
 *
 * Conventions
 *  - Strings live in an Arrow-style device column: chars (u8), offsets
 *    (rows+1 entries; int64, or int32 for columns an op produced with less
 *    than 2 GiB of chars -- cs_column_get_view always hands out int64,
 *    widening once on demand), validity bitmask (LSB-first, bit=1 valid,
 *    NULL pointer = all valid).  Columns are immutable and reference counted
 *    internally; every producing call returns a new handle the caller must
 *    release with cs_column_destroy (the reference's "new instance, caller
 *    destroys" rule, NVStrings.h:52-57,156).
 *  - Scalars (patterns, delimiters, replacement text) are host NUL-terminated
 *    UTF-8, exactly as in the reference.
 *  - Array arguments carry an `on_device` flag like the reference's trailing
 *    `bool devmem` (NVStrings.h:861,907,963): non-zero = device pointer.
 *  - Every call returns a cs_status; no C++ exception crosses the boundary.
 *    cs_last_error() returns the thread-local message of the last failure.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *    that return host-visible values synchronise that stream before returning.
 *  - There is no CPU fallback: without a usable gfx950 device every compute
 *    call fails with CS_ERR_NO_DEVICE.
 */
#ifndef CUSTRINGS_AMD_H
#define CUSTRINGS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cs_column cs_column;     /* device strings column            */
typedef struct cs_regex cs_regex;       /* compiled pattern (host + device) */
typedef struct cs_category cs_category; /* sorted unique keys + int32 codes */
typedef void* cs_stream;                /* hipStream_t                      */

typedef enum cs_status {
  CS_OK = 0,
  CS_ERR_INVALID_ARG = 1, /* std::invalid_argument in the reference */
  CS_ERR_ALLOC = 2,       /* std::runtime_error("allocate error")   */
  CS_ERR_HIP = 3,
  CS_ERR_NO_DEVICE = 4,
  CS_ERR_RANGE = 5, /* column does not fit an int32-offset export */
  CS_ERR_INTERNAL = 6
} cs_status;

/* Borrowed, read-only view of a column's device buffers. */
typedef struct cs_column_view {
  const uint8_t* chars;
  const int64_t* offsets; /* rows + 1 entries, offsets[0] == 0 */
  const uint8_t* validity; /* may be NULL */
  int64_t rows;
  int64_t nbytes;
  int64_t null_count;
} cs_column_view;

/* ---- library ----------------------------------------------------------- */
int cs_version(void);
/* 1 when the library was built with the experiments (`make exp`: the one-pass split and strip kernels -- bit-exact,
 * measured slower than the product's two passes, kept out of the product library), else 0. */
int cs_has_experiments(void);
const char* cs_last_error(void);
int cs_device_count(void);
/* Binds the calling thread to `device`, creates the stream-ordered memory
 * pool and uploads the unicode tables (reference: lazy get_unicode_flags /
 * get_charcases, NVStringsImpl.cu:69-91). Idempotent. */
int cs_init(int device);
/* A stream the caller is about to destroy: the buffer cache stops recording release events on it (blocks released after
 * a second stream appeared are ordered behind every stream that allocated so far; a destroyed stream is otherwise only
 * noticed when recording on its stale handle fails, which HIP does not promise).  The null stream cannot be forgotten. */
int cs_stream_forget(cs_stream stream);
/* Device the process is bound to (cs_init), or -1.  Compute calls from any host
 * thread run on it (the library re-binds the calling thread when needed). */
int cs_current_device(void);
/* Number of times a persistent single-pass kernel gave up (its grid was not
 * fully resident, e.g. a co-tenant on the device) and the column was recomputed
 * with the two-pass kernels.  Correct either way; a benchmark should see 0. */
int64_t cs_fallback_count(void);
/* Diagnostics / tests: which route the calling thread's last regex stream launch took -- "bits" (the bit-parallel form,
 * regex_bits.h), "chain", "units", "literal", "wide", "plain", "brefs", "brefs-chain" -- or "" when the call used other kernels. */
const char* cs_debug_last_route(void);
/* The CS_* switches (measurement aids, route overrides, opt-in experiments) are read from the environment ONCE, at the
 * library's first look (cs_init); the dispatch paths never call getenv.  This changes one at run time, thread-safely
 * (value NULL: unset) -- for tests and tools; a production host sets its environment before cs_init. */
int cs_config_set(const char* name, const char* value);
/* Bytes currently held by live columns/categories on this device. */
int64_t cs_device_bytes_in_use(void);
/* The buffer pool (reference: the RMM pool behind device_alloc, cpp/src/util.inl:90-106): released blocks are kept and
 * re-used; a new block's capacity is rounded up to a geometric size class (eight per octave, 3 % of headroom) so that a
 * column a little larger than the last one finds a block.  cs_debug_malloc_count: hipMalloc calls so far (tests: a
 * pipeline on columns of slightly different sizes allocates once); cs_pool_cached_bytes: bytes idle in the pool (bounded
 * by half the device's memory, CS_POOL_MAX_MB); cs_pool_trim: give idle blocks back down to `keep_bytes`. */
int64_t cs_debug_malloc_count(void);
/* What this box's memory delivers to hand-written streaming kernels (custrings_amd/csrc/box_rates.h), in TB/s (bytes read
 * + bytes written over the median of `reps` launches): tbps[0] a 16-byte-a-lane copy, [1] a read-only stream, [2] a
 * write-only stream, [3] the split emit kernel's shape -- one read stream into 20 x (256 + 192 + 8)-byte pieces per
 * 64-row sub-tile -- with plain stores, [4] the same with non-temporal stores; tbps[5]: the shader clock in GHz under an
 * all-CU integer + LDS load (six doubles in all).  `mbytes`: buffer size.  bench.py prints them as the `box` block of
 * its line (the headline's kernels follow the box's mixed read / write rate and its clock). */
int cs_box_rates(int64_t mbytes, int reps, cs_stream stream, double* tbps);
int64_t cs_pool_cached_bytes(void);
int cs_pool_trim(int64_t keep_bytes);

/* ---- column construction / export -------------------------------------- */
/* NVStrings::create_from_array (NVStrings.h:86): NULL entry = null row. */
int cs_column_from_host_strings(const char* const* strs, int64_t rows, cs_stream stream,
                                cs_column** out);
/* NVStrings::create_from_offsets (NVStrings.h:116): int32 Arrow offsets,
 * optional validity bitmask; buffers are copied. */
int cs_column_from_offsets32(const char* chars, int64_t rows, const int32_t* offsets,
                             const uint8_t* validity, int on_device, cs_stream stream,
                             cs_column** out);
/* Native ingest: int64 offsets. copy=0 wraps caller-owned DEVICE buffers that
 * must outlive the handle (zero-copy); copy=1 copies (host or device). */
int cs_column_from_offsets64(const uint8_t* chars, int64_t rows, const int64_t* offsets,
                             const uint8_t* validity, int on_device, int copy,
                             cs_stream stream, cs_column** out);
/* NVStrings::create_ipc_transfer / create_from_ipc (NVStrings.h:132,214; cpp/include/ipc_transfer.h:31-107):
 * a column handed to another process of the same node without a copy.  The exporter fills a plain
 * struct with the HIP IPC handles of the column's buffers (chars, offsets, validity) and their sizes;
 * the struct travels by any byte channel (pipe, socket, shared memory); the importer maps the buffers
 * and gets a read-only column over them.  The exporter's column must stay alive until every importer
 * has destroyed its handle.  (The reference ships an array of custring_view pointers plus the memory
 * they point into, rebased by the importer; the native layout has no pointers to rebase.) */
typedef struct cs_ipc_column {
  unsigned char chars[64], offsets[64], validity[64]; /* hipIpcMemHandle_t each */
  int64_t rows, nbytes, null_count;
  int32_t offset_width; /* 4 or 8 */
  int32_t has_validity;
  int32_t device;       /* the exporter's device ordinal (the importer must be able to reach it) */
  int32_t reserved;
} cs_ipc_column;
int cs_column_ipc_export(const cs_column* col, cs_ipc_column* out);
int cs_column_ipc_import(const cs_ipc_column* ipc, cs_column** out);
/* NVStrings::create_from_strings(std::vector<NVStrings*>) (NVStrings.h:125):
 * row-wise concatenation of `n` columns into a new column. */
int cs_column_concat(const cs_column* const* cols, int n, cs_stream stream, cs_column** out);
/* NVStrings::sublist(start, end) with step 1 (NVStrings.h:261): rows
 * [first, first + rows) as a new column (offsets rebased to 0).  This is also
 * how a column is cut into row-range shards for the multi-GPU path. */
int cs_column_slice(const cs_column* col, int64_t first, int64_t rows, cs_stream stream,
                    cs_column** out);
/* NVStrings::destroy (NVStrings.h:156). NULL is ignored. */
int cs_column_destroy(cs_column* col);
/* NVStrings::size (NVStrings.h:167) and friends. */
int64_t cs_column_rows(const cs_column* col);
int64_t cs_column_nbytes(const cs_column* col);
/* Bytes per offset the column was produced with: 4 (int32, an op's output below
 * 2 GiB of chars) or 8.  Either way cs_column_get_view hands out int64 offsets. */
int cs_column_offset_width(const cs_column* col);
int64_t cs_column_null_count(const cs_column* col);
/* What the column already knows about itself, without computing anything (diagnostics / tests): out[0] largest byte
 * span of 64 consecutive rows starting at a multiple of 64, out[1] longest row in bytes -- both upper bounds when an op
 * derived them as a by-product, exact when a pass measured them --, out[2] "plain bytes" (1 / 0), out[3] the non-ASCII
 * sample (1 / 0); -1 = not known yet (the first op that needs the number pays a pass for it).  The reference sizes its
 * strings once at ingest (NVStringsImpl.cu:399-444); here producers hand the numbers on so that ops in a chain do not. */
int cs_column_cached_meta(const cs_column* col, int64_t out[4]);
int cs_column_get_view(const cs_column* col, cs_column_view* view);
/* NVStrings::create_offsets (NVStrings.h:207): int32 offsets (rows+1), chars,
 * optional bitmask.  CS_ERR_RANGE when nbytes >= 2^31. */
int cs_column_export_offsets32(const cs_column* col, char* chars, int32_t* offsets,
                               uint8_t* validity, int on_device, cs_stream stream);
/* Native egress, int64 offsets. */
int cs_column_export_offsets64(const cs_column* col, uint8_t* chars, int64_t* offsets,
                               uint8_t* validity, int on_device, cs_stream stream);
/* NVStrings::byte_count (NVStrings.h:354): bytes per row, -1 for null rows;
 * *total receives the sum over non-null rows. */
int cs_column_byte_count(const cs_column* col, int32_t* lengths, int on_device,
                         cs_stream stream, int64_t* total);
/* NVStrings::set_null_bitarray (NVStrings.h:225): bit=1 valid, LSB first;
 * *null_count receives the number of cleared bits. */
int cs_column_null_bitarray(const cs_column* col, uint8_t* bitarray, int empty_is_null,
                            int on_device, cs_stream stream, int64_t* null_count);

/* ---- re-arrangement / combination -------------- */
/* NVStrings::create_from_index (NVStrings.h:98; NVStringsImpl.cu:209-325): `pairs`
 * is an array of `count` std::pair<const char*, size_t> (pointer, byte length;
 * NULL pointer = null row, zero length = empty string).  The pointers address
 * device-visible memory; on_device says where the PAIR ARRAY lives (the
 * reference's `devmem`).  sorttype: NVStrings::sorttype bits (0 none, 1 length,
 * 2 name): ascending, nulls first.  An unreadable pointer surfaces as
 * CS_ERR_INVALID_ARG ("bad_device_ptr"). */
int cs_column_from_index(const void* pairs, int64_t count, int on_device, int sorttype,
                         cs_stream stream, cs_column** out);
/* NVStrings::create_index (NVStrings.h:180; NVStrings.cu:348-400): one (pointer,
 * byte length) pair per row, the pointers addressing the column's own device
 * chars (valid until the handle is destroyed); null row = (NULL, 0). */
int cs_column_create_index(const cs_column* col, void* pairs, int on_device, cs_stream stream);
/* NVStrings::len (NVStrings.h:343; attrs.cu:32-69): characters per row, -1 for
 * null rows; *total = sum over non-null rows (the row count when `lengths` is
 * NULL, as in the reference). */
int cs_len(const cs_column* col, int32_t* lengths, int on_device, cs_stream stream,
           int64_t* total);
/* NVStrings::gather(pos, count) (NVStrings.h:278; array.cu:73-118): rows in the
 * given order, repeats allowed; a position outside [0, rows) -> CS_ERR_RANGE
 * (std::out_of_range in the reference). */
int cs_gather(const cs_column* col, const int32_t* pos, int64_t n, int on_device,
              cs_stream stream, cs_column** out);
/* NVStrings::gather(mask) (NVStrings.h:288; array.cu:121-146): rows whose mask
 * byte is non-zero. */
int cs_gather_mask(const cs_column* col, const uint8_t* mask, int on_device,
                   cs_stream stream, cs_column** out);
/* NVStrings::sublist(start, end, step) (NVStrings.h:261; array.cu:238-260). */
int cs_sublist(const cs_column* col, int64_t start, int64_t end, int64_t step,
               cs_stream stream, cs_column** out);
/* NVStrings::scatter(strs, pos) / scatter(str, pos, count) (NVStrings.h:302,318;
 * array.cu:157-236): a copy of `col` with row pos[j] replaced by row j of
 * `strs` (or by the scalar; NULL = null row).  Positions outside the column are
 * ignored; when a row is named more than once the last entry wins. */
int cs_scatter(const cs_column* col, const cs_column* strs, const int32_t* pos, int on_device,
               cs_stream stream, cs_column** out);
int cs_scatter_scalar(const cs_column* col, const char* str, const int32_t* pos, int64_t n,
                      int on_device, cs_stream stream, cs_column** out);
/* NVStrings::sort / order (NVStrings.h:1102,1114; array.cu:303-360): sorttype
 * bits 1 = byte length, 2 = name (unsigned bytewise, custring.inl:240-261);
 * nulls first or last regardless of direction.  Equal rows keep their index
 * order (the reference's comparator sort leaves their order unspecified). */
int cs_sort(const cs_column* col, int sorttype, int ascending, int nullfirst, cs_stream stream,
            cs_column** out);
int cs_order(const cs_column* col, int sorttype, int ascending, int nullfirst, uint32_t* indexes,
             int on_device, cs_stream stream);
/* NVStrings::cat(others, separator, narep) (NVStrings.h:384,397; combine.cu:31-291):
 * row-wise concatenation of `col` and up to 15 other columns of the same row
 * count; a null element is replaced by narep, or makes the row null when narep
 * is NULL. */
int cs_cat(const cs_column* col, const cs_column* const* others, int nothers,
           const char* separator, const char* narep, cs_stream stream, cs_column** out);
/* NVStrings::join(delimiter, narep) (NVStrings.h:561; combine.cu:293-420): all
 * rows joined into a column of ONE row; a null row contributes narep, or nothing
 * (and no delimiter) when narep is NULL.  delimiter NULL -> CS_ERR_INVALID_ARG. */
int cs_join(const cs_column* col, const char* delimiter, const char* narep, cs_stream stream,
            cs_column** out);

/* ---- per-row string ops ------------------------------------------------ */
/* NVStrings::lower / upper (NVStrings.h:815,822; case.cu:31-97,100-170). */
int cs_lower(const cs_column* col, cs_stream stream, cs_column** out);
int cs_upper(const cs_column* col, cs_stream stream, cs_column** out);
/* NVStrings::lstrip / strip / rstrip (NVStrings.h:796-808; strip.cu:30-199).
 * to_strip NULL = " \n\t".  side: 0 both, 1 left, 2 right. */
int cs_strip(const cs_column* col, const char* to_strip, int side, cs_stream stream,
             cs_column** out);
/* NVStrings::find (NVStrings.h:861; find.cu:75-120): char position of the
 * first occurrence in [start,end), -1 not found, -2 null row.
 * *found = number of rows with result != -1 (null rows included, as in the
 * reference). */
int cs_find(const cs_column* col, const char* str, int start, int end, int32_t* results,
            int on_device, cs_stream stream, int64_t* found);
/* NVStrings::contains (NVStrings.h:907; find.cu:237-272): 1 byte per row. */
/* The rest of NVStrings' find family (NVStrings.h:837-934; find.cu:36-72, 123-236, 276-387).  Positions are character
 * positions; null rows give -2 (positions), -1 (compare), false (predicates); the count out is what the reference returns. */
/* NVStrings::rfind: the last occurrence inside characters [start, end) (end < 0: to the end). */
int cs_rfind(const cs_column* col, const char* str, int start, int end, int32_t* results, int on_device, cs_stream stream, int64_t* found);
/* NVStrings::find_from: per-row windows; `starts` / `ends` hold one int32 per row (host memory unless bounds_on_device), either may be null. */
int cs_find_from(const cs_column* col, const char* str, const int32_t* starts, const int32_t* ends, int bounds_on_device, int32_t* results,
                 int on_device, cs_stream stream, int64_t* found);
/* NVStrings::find_multiple: results[row * targets + j] = position of targets[j] in the row. */
int cs_find_multiple(const cs_column* col, const cs_column* targets, int32_t* results, int on_device, cs_stream stream, int64_t* found);
/* NVStrings::compare: bytewise difference (custring.inl:240-261); *matches = rows equal to `str`.  An empty `str` is
 * compared like any other (1 for a non-empty row, 0 for an empty one) -- the reference returns without writing its
 * results (find.cu) and its callers hand back an uninitialised buffer. */
int cs_compare(const cs_column* col, const char* str, int32_t* results, int on_device, cs_stream stream, int64_t* matches);
/* NVStrings::match_strings: row-wise equality of two columns of the same size (two nulls are equal). */
int cs_match_strings(const cs_column* col, const cs_column* other, uint8_t* results, int on_device, cs_stream stream, int64_t* matches);
/* NVStrings::startswith / endswith. */
int cs_startswith(const cs_column* col, const char* str, uint8_t* results, int on_device, cs_stream stream, int64_t* matches);
int cs_endswith(const cs_column* col, const char* str, uint8_t* results, int on_device, cs_stream stream, int64_t* matches);
int cs_contains(const cs_column* col, const char* str, uint8_t* results, int on_device,
                cs_stream stream, int64_t* found);
/* NVStrings::replace (NVStrings.h:714; modify.cu:109-192). str NULL/empty ->
 * CS_ERR_INVALID_ARG. */
int cs_replace(const cs_column* col, const char* str, const char* repl, int maxrepl,
               cs_stream stream, cs_column** out);
/* NVStrings::split(delimiter,maxsplit,results) and split(maxsplit,results)
 * (NVStrings.h:504,524; split.cu:734-956). delimiter NULL = whitespace.
 * Column-major result: *out_cols is a malloc'd array of *ncols handles
 * (release each handle, then cs_free the array). */
int cs_split(const cs_column* col, const char* delimiter, int maxsplit, cs_stream stream,
             cs_column*** out_cols, int* ncols);
/* NVStrings::rsplit(delimiter,maxsplit,results) and rsplit(maxsplit,results)
 * (NVStrings.h:514,534; split.cu:960-1148): as cs_split with the tokens
 * located from the right (the per-row token count is split's).  Same
 * ownership of *out_cols. */
int cs_rsplit(const cs_column* col, const char* delimiter, int maxsplit, cs_stream stream,
              cs_column*** out_cols, int* ncols);
void cs_free(void* p);

/* ---- regex -------------------------------------------------------------- */
/* Reprog::create_from + dreprog::create_from (regcomp.cpp:954-960,
 * regexec.cpp:12-73). Host-only compile; usable without a GPU.  The library keeps
 * the last 32 compiled patterns: a handle is a counted reference to an immutable
 * object that several handles (and host threads) may share; every handle is
 * released with cs_regex_destroy.  CS_REGEX_NO_CACHE=1 compiles afresh per call. */
int cs_regex_compile(const char* pattern, cs_regex** out);
int cs_regex_destroy(cs_regex* re);
int cs_regex_inst_count(const cs_regex* re);
/* Which executor runs the pattern (no reference counterpart: the reference has one executor, regexec.inl:204-442).
 * Bit 0: the tagged DFA (otherwise the ordered-list simulator); bit 1: the unit decomposition is offered; bit 2: capture
 * groups are tracked on the DFA; bits 8..11: live threads the automaton keeps at most; bits 16..31: DFA states. */
int cs_regex_engine(const cs_regex* re);
/* Flat int32 program image (layout: custrings_amd/csrc/regex_program.h). */
int cs_regex_blob(const cs_regex* re, const int32_t** words, int* nwords);
/* NVStrings::contains_re / match / count_re (NVStrings.h:963,975,988;
 * count.cu:59-250). */
int cs_contains_re(const cs_column* col, const cs_regex* re, uint8_t* results, int on_device,
                   cs_stream stream, int64_t* found);
int cs_match_re(const cs_column* col, const cs_regex* re, uint8_t* results, int on_device,
                cs_stream stream, int64_t* found);
int cs_count_re(const cs_column* col, const cs_regex* re, int32_t* results, int on_device,
                cs_stream stream, int64_t* found);
/* NVStrings::replace_re (NVStrings.h:766; replace.cu:110-189). */
int cs_replace_re(const cs_column* col, const cs_regex* re, const char* repl, int maxrepl,
                  cs_stream stream, cs_column** out);
/* NVStrings::replace_re(patterns, repls) (NVStrings.h:777; replace_multi.cu:110-189):
 * at every character position the patterns are tried in order, anchored there;
 * the first that matches is replaced by repls[its index] (or by repls[0] when
 * `repls` holds one row; a null replacement removes the match) and the walk
 * continues behind the match.  A pattern that can match the empty string is
 * refused with CS_ERR_INVALID_ARG (the reference does not terminate on it). */
int cs_replace_re_multi(const cs_column* col, const cs_regex* const* patterns, int npatterns,
                        const cs_column* repls, cs_stream stream, cs_column** out);
/* NVStrings::replace_with_backrefs(pattern, repl) (NVStrings.h:788;
 * replace_backref.cu:36-207): every match is replaced by `repl` with \N
 * (backslash + digits) standing for capture group N of that match (0 = the
 * whole match; a group that took no part contributes nothing).  repl NULL ->
 * all rows null.  At most 16 references.  A pattern that can match the empty
 * string is refused with CS_ERR_INVALID_ARG (the reference does not terminate
 * on it). */
int cs_replace_with_backrefs(const cs_column* col, const cs_regex* re, const char* repl,
                             cs_stream stream, cs_column** out);
/* NVStrings::extract(pattern, results) (NVStrings.h:682; extract.cu:69-151):
 * column-major, one column per capture group; a row is null unless the
 * pattern matches and the group's span is non-empty.  *out_cols is a malloc'd
 * array of *ncols handles (release each, then cs_free the array); a pattern
 * without capture groups or a column of zero rows yields *ncols = 0. */
int cs_extract(const cs_column* col, const cs_regex* re, cs_stream stream,
               cs_column*** out_cols, int* ncols);
/* NVStrings::findall(pattern, results) (NVStrings.h:943; findall.cu:99-179):
 * column-major, column k holds every row's k-th match in count_re order; rows
 * with fewer matches (and null rows) are null, an empty match is an empty
 * string.  No match in any row yields one all-null column; a column of zero
 * rows yields *ncols = 0.  Same ownership as cs_extract. */
int cs_findall(const cs_column* col, const cs_regex* re, cs_stream stream,
               cs_column*** out_cols, int* ncols);

/* Record (row-major) forms -- NVStrings::split_record / rsplit_record /
 * extract_record / findall_record (NVStrings.h:443-494,693,952;
 * extract_record.cu:146-152, findall_record.cu:144-151) return one instance
 * per row.  Natively a record result is ONE column holding the records'
 * strings in row-major order plus rows+1 list offsets (record r = flat rows
 * [list_offsets[r], list_offsets[r+1])).  This call transposes a column-major
 * result (`ncols` columns of equal row count): ragged = 0 keeps
 * every column per record, nulls included (extract_record); ragged = 1 keeps
 * each row's leading non-null columns (findall_record, split_record: a null
 * row or a row without matches has an empty record).  list_offsets: rows+1
 * int64, device memory when on_device. */
int cs_records_from_columns(const cs_column* const* cols, int ncols, int ragged,
                            int64_t* list_offsets, int on_device, cs_stream stream,
                            cs_column** out);

/* NVStrings::split_record / rsplit_record (NVStrings.h:443-494; split.cu:125-700)
 * natively: ONE flat column of every row's tokens in row-major order plus rows+1
 * list offsets (record r = flat rows [list_offsets[r], list_offsets[r+1]); a
 * null row has no entries, where the reference pushes a null instance).
 * delimiter NULL (or "") = whitespace: a valid row without tokens then yields one
 * empty string (split.cu:393-397).  list_offsets: int64, device memory when
 * on_device. */
int cs_split_record(const cs_column* col, const char* delimiter, int maxsplit,
                    int64_t* list_offsets, int on_device, cs_stream stream, cs_column** out);
int cs_rsplit_record(const cs_column* col, const char* delimiter, int maxsplit,
                     int64_t* list_offsets, int on_device, cs_stream stream, cs_column** out);
/* NVStrings::partition / rpartition (NVStrings.h:537,549; split.cu:1165-1361):
 * three strings per row -- head, delimiter, tail around the first (last)
 * occurrence; no occurrence: (row, "", "") / ("", "", row); a null row: three
 * nulls.  Flat column of 3 * rows entries (row r = entries 3r .. 3r+2).  A NULL
 * or empty delimiter yields *out = NULL (the reference returns no results). */
int cs_partition(const cs_column* col, const char* delimiter, int from_right, cs_stream stream,
                 cs_column** out);

/* ---- category (dictionary encoding) ------------------------------------ */
/* NVCategory::create_from_strings (NVCategory.h:107; NVCategory.cu:220-304):
 * keys = sorted unique rows (null first, then unsigned bytewise order,
 * custring.inl:240-261), codes[r] = index of row r's key. */
int cs_category_build(const cs_column* col, cs_stream stream, cs_category** out);
/* NVCategory::create_from_categories (NVCategory.h:121; NVCategory.cu:430-514):
 * merged sorted-unique key set, concatenated remapped codes. */
int cs_category_merge(const cs_category* const* cats, int ncats, cs_stream stream,
                      cs_category** out);
/* The merge step of a distributed category build (one process per GPU): after the ranks all-gathered their key sets --
 * the transport is the caller's: RCCL, MPI, torch.distributed -- every rank merges them and maps its local codes into the
 * merged key set.  keysets[r] = rank r's key set (this rank's own at index `rank`); *merged_keys is the same column on every
 * rank; values = device memory for cs_category_size(local) int32 codes.  Composes NVCategory::create_from_categories
 * (NVCategory.cu:430-514); the reference itself is single-GPU. */
int cs_category_merge_gathered(const cs_category* local, const cs_column* const* keysets, int nranks, int rank, cs_stream stream,
                               cs_column** merged_keys, int32_t* values);
/* The whole distributed build behind the C ABI (BASELINE.json north_star; what custrings_amd/dist.py does through
 * torch.distributed): local build (NVCategory.cu:220-304), all-gather of the ranks' sorted key sets, merge
 * (NVCategory::create_from_categories, NVCategory.cu:430-514), remap.  One process per GPU.  `nccl_comm`: the caller's
 * ncclComm_t of `nranks` ranks (RCCL is resolved in the process at run time: dlsym, else dlopen of librccl.so);
 * *out: keys = the same column on every rank, values = this rank's rows' codes.  nranks == 1 is the local build. */
int cs_category_build_distributed(const cs_column* col, void* nccl_comm, int nranks, int rank, cs_stream stream, cs_category** out);
/* The same over a caller-supplied all-gather (another transport, a test harness): `allgather(ctx, send, recv, bytes,
 * stream)` puts every rank's `bytes` bytes at `send` (device memory) behind each other, in rank order, at `recv`
 * (device memory, nranks * bytes), ordered after the work queued on `stream`; returns 0 on success. */
typedef int (*cs_allgather_fn)(void* ctx, const void* send, void* recv, size_t bytes, void* stream);
int cs_category_build_distributed_with(const cs_column* col, cs_allgather_fn allgather, void* ctx, int nranks, int rank, cs_stream stream,
                                       cs_category** out);
/* ... and a caller-supplied all-to-all of variable pieces, which opens the second route: from 2^21 keys in all (K close to
 * N) the merge is partitioned by key RANGE -- splitters from a sample of every rank's keys, every key to its range's
 * owner, the merged ranges all-gathered: every rank merges 1/nranks of the keys instead of all of them.
 * `alltoallv(ctx, send, send_bytes, send_off, recv, recv_bytes, recv_off, nranks, stream)`: this rank's piece for rank d
 * is send_bytes[d] bytes at send + send_off[d] (device memory); what rank d sent to this rank arrives at recv + recv_off[d]
 * (recv_bytes[d] bytes: the library exchanged the sizes beforehand); ordered after the work queued on `stream`; 0 on
 * success.  cs_category_build_distributed passes grouped ncclSend / ncclRecv.  NULL: the all-gather route only.
 * A rank whose local build fails still takes part in the first exchange and every rank returns an error together. */
typedef int (*cs_alltoallv_fn)(void* ctx, const void* send, const size_t* send_bytes, const size_t* send_off, void* recv, const size_t* recv_bytes,
                               const size_t* recv_off, int nranks, void* stream);
int cs_category_build_distributed_with2(const cs_column* col, cs_allgather_fn allgather, cs_alltoallv_fn alltoallv, void* ctx, int nranks, int rank,
                                        cs_stream stream, cs_category** out);
int cs_category_destroy(cs_category* cat);
/* NVCategory::create_ipc_transfer / create_from_ipc (NVCategory.h:128,176; ipc_transfer.h:109-200): the key column
 * as above plus the handle of the int32 values. */
typedef struct cs_ipc_category {
  cs_ipc_column keys;
  unsigned char values[64]; /* hipIpcMemHandle_t */
  int64_t rows;
} cs_ipc_category;
int cs_category_ipc_export(const cs_category* cat, cs_ipc_category* out);
int cs_category_ipc_import(const cs_ipc_category* ipc, cs_category** out);
int64_t cs_category_size(const cs_category* cat);      /* NVCategory::size      */
int64_t cs_category_keys_size(const cs_category* cat); /* NVCategory::keys_size */
/* NVCategory::get_keys (new handle on the shared key column). */
int cs_category_keys(const cs_category* cat, cs_column** out);
/* NVCategory::values_cptr: borrowed device pointer to the int32 codes. */
const int32_t* cs_category_values_ptr(const cs_category* cat);
/* NVCategory::get_values (NVCategory.h:225). */
int cs_category_get_values(const cs_category* cat, int32_t* out, int on_device,
                           cs_stream stream);
/* The remap family (NVCategory.h:247-420; NVCategory.cu:926-1822).  Keys stay
 * sorted unique (merge_category alone appends the new keys behind the old ones);
 * values of keys that disappear become -1. */
int cs_category_to_strings(const cs_category* cat, cs_stream stream, cs_column** out);
int cs_category_gather_strings(const cs_category* cat, const int32_t* pos, int64_t n, int on_device,
                               cs_stream stream, cs_column** out); /* CS_ERR_RANGE: std::out_of_range */
int cs_category_gather(const cs_category* cat, const int32_t* pos, int64_t n, int on_device,
                       cs_stream stream, cs_category** out);
int cs_category_gather_and_remap(const cs_category* cat, const int32_t* pos, int64_t n, int on_device,
                                 cs_stream stream, cs_category** out);
int cs_category_add_strings(const cs_category* cat, const cs_column* strs, cs_stream stream,
                            cs_category** out);
int cs_category_remove_strings(const cs_category* cat, const cs_column* strs, cs_stream stream,
                               cs_category** out);
int cs_category_merge_category(const cs_category* cat, const cs_category* cat2, cs_stream stream,
                               cs_category** out);
int cs_category_add_keys(const cs_category* cat, const cs_column* strs, cs_stream stream,
                         cs_category** out); /* add_keys_and_remap */
int cs_category_remove_keys(const cs_category* cat, const cs_column* strs, cs_stream stream,
                            cs_category** out); /* remove_keys_and_remap */
int cs_category_remove_unused_keys(const cs_category* cat, cs_stream stream, cs_category** out);
int cs_category_set_keys(const cs_category* cat, const cs_column* strs, cs_stream stream,
                         cs_category** out); /* set_keys_and_remap */
/* Multi-GPU key-set merge helper: out[i] = table[codes[i]] (codes < 0 kept). */
int cs_remap_codes(const int32_t* codes, int64_t n, const int32_t* table, int32_t* out,
                   cs_stream stream);

/* ---- text --------------------------------------------------------------- */
/* NVText::tokenize(strs, delimiter) (NVText.h:40; tokens.cu:123-155):
 * delimiter NULL = whitespace (char <= ' '), else any char of `delimiter`. */
int cs_tokenize(const cs_column* col, const char* delimiter, cs_stream stream,
                cs_column** out);
/* NVText::tokenize(strs, delimiters) (NVText.h:48; tokens.cu:158-260): every row of
 * `delimiters` is a whole-string delimiter (tried in order at each byte; null and
 * empty rows skipped); empty tokens are dropped.  No delimiters = whitespace. */
int cs_tokenize_multi(const cs_column* col, const cs_column* delimiters, cs_stream stream,
                      cs_column** out);
/* NVText::create_ngrams (NVText.h:153; ngram.cu:32-110). */
int cs_ngrams(const cs_column* tokens, unsigned ngrams, const char* separator,
              cs_stream stream, cs_column** out);

/* NVText counters (NVText.h:51-141; tokens.cu:262-716).  delimiter NULL (or "")
 * = whitespace, else any character of the string. */
int cs_token_count(const cs_column* col, const char* delimiter, uint32_t* results, int on_device,
                   cs_stream stream); /* tokens per row, 0 for a null row */
int cs_unique_tokens(const cs_column* col, const char* delimiter, cs_stream stream,
                     cs_column** out); /* sorted distinct tokens */
/* results[r * tokens_rows + t] = how many of row r's tokens equal tokens[t]. */
int cs_tokens_counts(const cs_column* col, const cs_column* tokens, const char* delimiter,
                     uint32_t* results, int on_device, cs_stream stream);
/* Every token equal to targets[t] becomes repls[t] (repls[0] when `repls` has one
 * row; a null replacement removes the token).  *out = NULL when the result holds
 * no bytes at all (tokens.cu:612-613). */
int cs_replace_tokens(const cs_column* col, const cs_column* targets, const cs_column* repls,
                      const char* delimiter, cs_stream stream, cs_column** out);
/* Tokens of each row joined by one space; *out = NULL when no row holds a token
 * (tokens.cu:703-704). */
int cs_normalize_spaces(const cs_column* col, cs_stream stream, cs_column** out);

/* ---- synthetic workloads (bench / parity inputs) -- */
/* kind: 2 = C2 word rows, 3 = C3 log lines, 4 = C4 16-char tokens,
 * 5 = C5 tweet-like text.  Row r is a pure function of (seed, first_row + r). */
int cs_synth_column(int kind, int64_t first_row, int64_t rows, uint64_t seed,
                    int64_t param, cs_stream stream, cs_column** out);
/* Order-sensitive 64-bit digest of a column (offsets, chars, validity);
 * used for full-size parity ("checksum of checksums"). */
int cs_column_digest(const cs_column* col, cs_stream stream, uint64_t* digest);

/* ---- measurement -------------------------------------------------------- */
/* Device time (ms, HIP events on `stream`) and launch count of the dominant
 * kernel accumulated since the last reset; see bench.py "roofline". */
int cs_prof_reset(void);
int cs_prof_enable(int on);
int cs_prof_get(const char* kernel, double* total_ms, int64_t* launches);
/* Test aid: `blocks` workgroups of 256 threads and `lds_bytes` of LDS each that do nothing for `milliseconds`, queued on
 * `stream` -- a co-tenant that keeps part of the GPU occupied while another stream's call runs (the persistent kernels
 * size their grids for an empty device). */
int cs_debug_spin(int blocks, int lds_bytes, int milliseconds, cs_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* CUSTRINGS_AMD_H */
