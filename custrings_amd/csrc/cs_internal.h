// Internal host-side plumbing shared by the library's translation units:
// error handling, the device buffer cache, the column object and the
// lengths -> offsets scan.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "../../include/custrings_amd.h"
#include "cs_config.h"

namespace cs {

struct Error {
  int code;
  std::string msg;
};
[[noreturn]] void fail(int code, const std::string& msg);
void set_last_error(const std::string& msg);

#define CS_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      ::cs::fail(CS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

// Runs `f`, mapping exceptions to status codes (nothing throws across the ABI).
template <class F>
int guard(F&& f) {
  try {
    f();
    return CS_OK;
  } catch (const Error& e) {
    set_last_error(e.msg);
    return e.code;
  } catch (const std::bad_alloc&) {
    set_last_error("host allocation failed");
    return CS_ERR_ALLOC;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return CS_ERR_INTERNAL;
  }
}

// Throws CS_ERR_NO_DEVICE unless cs_init succeeded; binds the calling thread to the library's
// device when its current device differs (hipSetDevice is per thread).
void require_device();
int bound_device();  // -1 before cs_init
// One message per process on stderr and a counter (cs_fallback_count): a persistent single-pass
// kernel gave up and the host recomputed the column with the two-pass kernels.
void note_fallback(const char* what);
void note_route(const char* route);  // cs_debug_last_route (per thread)
void note_route_put_off();            // ... of a scan that left its rows with bytes >= 0x80 to a second launch
void note_route_pieces();             // ... of an op that ran on the column's pieces (cs_virtual.hip)

// ---- device memory ----------------------------------------------------------
// Buffers come from a size-bucketed cache over hipMalloc (one output allocation
// per produced buffer, no allocation inside kernels).  Every buffer carries 64
// bytes of slack so 16-byte vector loads may run past the logical end.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;     // usable bytes requested
  size_t capacity = 0;  // bytes actually held (0 for wrapped buffers)
  hipStream_t stream = nullptr;
  bool ipc_mapped = false;  // opened with hipIpcOpenMemHandle: closed, not freed (cs_column_ipc_import)
  ~DevBuf();
};
using Buf = std::shared_ptr<DevBuf>;
Buf dev_alloc(size_t bytes, hipStream_t stream);
Buf dev_wrap(const void* p, size_t bytes);  // caller-owned, never freed
template <class T>
T* ptr(const Buf& b) {
  return b ? static_cast<T*>(b->p) : nullptr;
}
int64_t dev_bytes_in_use();

// pinned host scratch for small device->host results (per thread)
void* pinned_scratch(size_t bytes);

// unicode tables on the device (uploaded by cs_init)
const uint8_t* d_unicode_flags();
const uint16_t* d_charcases();
const uint8_t* h_unicode_flags();
const uint16_t* h_charcases();

}  // namespace cs

namespace cs {
struct OddRows;      // cs_virtual.hip: the rows that hold a byte >= 0x80 or a NUL, as a list and as a mask per 64-row tile
struct VirtualRows;  // cs_virtual.hip: the column's rows cut into pieces of at most 92 bytes (a second column over the same chars)
}
// ---- the column ---------------------------------------------------------------
struct cs_column {
  int64_t rows = 0;
  int64_t nbytes = 0;
  mutable int64_t null_count = -1;  // -1 = not counted yet
  mutable int64_t max_span64 = -1;  // max bytes spanned by 64 consecutive rows (tile kernels); -1 = unknown
  mutable int64_t max_row = -1;     // longest row in bytes; -1 = unknown
  mutable int drops = -1;           // 1: some row is null or empty (rows create_ngrams drops), 0: none; -1 = unknown
  mutable int plain_bytes = -1;     // 1: no NUL byte and no lead byte announcing over an ASCII byte; -1 = unknown
  mutable int high_sample = -1;     // 1: a sample of the chars (three 64 KiB windows) holds a byte >= 0x80; -1 = not looked at
  mutable std::shared_ptr<std::array<uint32_t, 256>> byte_hist;  // byte counts of the same sample (a hint for kernel choice); null = not taken
  mutable std::shared_ptr<cs::VirtualRows> virt;  // the view of a long-row column as pieces (built on first use; cs_virtual.hip)
  mutable int virt_state = 0;                     // 0: not looked at, 1: `virt` holds it, -1: the column has none
  mutable std::shared_ptr<cs::OddRows> odd;       // the rows with a byte >= 0x80 or a NUL (built on first use; cs_virtual.hip)
  cs::Buf chars, validity;  // validity may be null (all valid)
  // Row extents: int64 offsets (`offsets`) and / or int32 offsets (`offsets32`, columns whose
  // chars stay below 2 GiB -- what split produces: half the bytes written per output row).  A
  // column born with int32 offsets gets its int64 form on first use by a kernel that reads
  // int64 (d_offsets(), under a lock; the widened buffer is kept).
  mutable cs::Buf offsets;
  cs::Buf offsets32;
  const uint8_t* d_chars() const { return cs::ptr<const uint8_t>(chars); }
  const int64_t* d_offsets() const;
  const int32_t* d_offsets32() const { return cs::ptr<const int32_t>(offsets32); }
  const uint8_t* d_validity() const { return cs::ptr<const uint8_t>(validity); }
  // the other column gets the same (immutable) extents
  void share_extents_with(cs_column* o) const;
};

// ---- the category: sorted unique keys + one int32 code per row (-1 / key 0 conventions: cs_category.hip)
struct cs_category {
  std::unique_ptr<cs_column> keys;
  cs::Buf values;  // int32[rows]
  int64_t rows = 0;
};

namespace cs {

struct ColView {  // passed to kernels by value
  const uint8_t* chars;
  const int64_t* offsets;
  const uint8_t* validity;
  int64_t rows;
};
inline ColView view_of(const cs_column* c) {
  return ColView{c->d_chars(), c->d_offsets(), c->d_validity(), c->rows};
}
inline hipStream_t S(cs_stream s) { return static_cast<hipStream_t>(s); }
inline unsigned blocks_for(int64_t rows) { return (unsigned)((rows + 255) / 256); }
inline size_t validity_bytes(int64_t rows) { return (size_t)((rows + 63) / 64) * 8; }

// Empty column with `rows` rows, all null (NVStrings(count)) or zero rows.
cs_column* make_all_null(int64_t rows, hipStream_t s);

// Exclusive scan of int32 lengths (negative = null row, counts as 0) into int64
// offsets[n+1]; returns the total (synchronises `s`).  When `block_sums` is
// given it must hold the per-256-row sums already (fused into the caller's size
// kernel) and the lengths are read only once.
// `meta` (optional): the column metadata the tile kernels ask for, as a by-product of the same pass -- the longest row and
// the largest byte span of 64 consecutive rows starting at a multiple of 64 (what max_row_bytes / max_span64 would compute
// with passes of their own over the finished offsets).
struct LenMeta {
  int64_t max_row = -1, max_span64 = -1;
  void give(cs_column* c) const {
    if (max_row >= 0) c->max_row = max_row;
    if (max_span64 >= 0) c->max_span64 = max_span64;
  }
};
int64_t offsets_from_lengths(const int32_t* lens, int64_t n, int64_t* offsets, hipStream_t s,
                             Buf block_sums = nullptr, LenMeta* meta = nullptr);
void offsets_from_lengths_async(const int32_t* lens, int64_t n, int64_t* offsets, hipStream_t s);  // nothing read back, no wait: the total is offsets[n] on the device
// the same and the validity mask (length >= 0) from one pass over the lengths
int64_t offsets_and_validity_from_lengths(const int32_t* lens, int64_t n, int64_t* offsets, Buf* validity, hipStream_t s, LenMeta* meta = nullptr);
// Segmented variant: `segs` independent arrays of n lengths laid out back to
// back (lens[seg * n + i]); offsets[seg * (n + 1) + i]; totals[seg] on the host.
// (`largest_host`, optional: the largest length of every segment -- for per-tile byte counts that is the column's largest
// 64-row span, by-product of the same pass)
void offsets_from_lengths_segmented(const int32_t* lens, int64_t n, int segs, int64_t* offsets,
                                    int64_t* totals_host, hipStream_t s, int64_t* largest_host = nullptr);
// Validity bitmask from int32 lengths (bit set when len >= 0).
Buf validity_from_lengths(const int32_t* lens, int64_t n, hipStream_t s);
int64_t count_nulls(const cs_column* c, hipStream_t s);
// Largest byte span of 64 consecutive rows starting at a multiple of 64 (cached
// in the column; sizes the LDS staging buffers of the tile kernels).
int64_t max_span64(const cs_column* c, hipStream_t s);
int64_t max_row_bytes(const cs_column* c, hipStream_t s);
bool bytes_plain(const cs_column* c, hipStream_t s);
// Same for tiles of `per` consecutive rows (per = 64 is the cached one).
int64_t max_span_rows(const cs_column* c, int per, hipStream_t s);
bool sample_has_high_bytes(const cs_column* c, hipStream_t s);
const uint32_t* sample_byte_hist(const cs_column* c, hipStream_t s);  // 256 counts over the same three windows (a hint, cached on the column)  // a hint (kernel choice only): non-ASCII text, by three windows of the chars
int64_t count_spans64_over(const cs_column* c, int64_t limit, hipStream_t s);
bool few_spans64_over(const cs_column* c, int64_t limit, hipStream_t s);  // all but a few 64-row tiles fit `limit` bytes
// Workgroups (256 threads, `lds` dynamic bytes) of `kern` resident at once on the device,
// capped by `wanted`: grid size of the persistent tile kernels.
unsigned resident_grid(const void* kern, size_t lds, int64_t wanted);
}  // namespace cs
namespace csrow {
struct CharSet;
}
namespace cs {
// cs_ops.hip: the character set of a UTF-8 string of any length (strip, tokenize with a delimiter set, the NVText counters);
// `more` keeps what does not fit the struct alive for the caller's kernels
csrow::CharSet make_charset(const char* s, Buf& more, hipStream_t st);
// Row-wise concatenation of columns into one new column.
cs_column* concat_columns(const std::vector<const cs_column*>& cols, hipStream_t s);

// cs_virtual.hip: a long-row column as a column of pieces that fit the 96-bit masks
struct VirtualRows {
  std::unique_ptr<cs_column> col;  // the pieces: the same chars buffer, offsets of their own
  Buf first;                       // int64[rows + 1]: row r's pieces are first[r] .. first[r + 1] - 1
};
const VirtualRows* virtual_rows(const cs_column* col, hipStream_t s);
// the rows that hold a byte >= 0x80 or a NUL byte -- the rows the 96-bit-mask forms of the regex kernels do not take: replace_re
// leaves them holes in its single pass and fills them from a thread a row (cs_regex.hip); column metadata, one pass over the chars
struct OddRows {
  int64_t count = 0;
  Buf list;   // int32[count]: their indices, ascending
  Buf mask;   // uint64[tiles]: bit j of word t = row 64 t + j is such a row
  Buf first;  // int64[tiles + 1]: list[first[t]] is tile t's first
};
const OddRows* odd_rows(const cs_column* col, hipStream_t s);
int virtual_piece_bytes();
int64_t virtual_reduce_u8(const VirtualRows* vr, const uint8_t* piece_res, int64_t rows, uint8_t* out, hipStream_t s);
int64_t virtual_reduce_i32(const VirtualRows* vr, const int32_t* piece_res, int64_t rows, int32_t* out, hipStream_t s);
cs_column* virtual_rows_to_rows(const cs_column* col, const VirtualRows* vr, std::unique_ptr<cs_column> pieces_out, hipStream_t s);
// cs_radix.hip: stable LSD radix sort of n (64-bit key, 32-bit item) pairs by key, ascending, in place (synchronises `s`)
void radix_sort_pairs64(uint64_t* keys, int32_t* items, int64_t n, hipStream_t s);
// cs_category.hip: keys = sorted unique rows (null first), values[r] = index of row r's key
cs_category* category_build(const cs_column* col, hipStream_t s);
// cs_array.hip: rows of `col` at the given device positions; with `null_when_negative` a negative
// position yields a null row instead of CS_ERR_RANGE
cs_column* gather_rows(const cs_column* col, const int32_t* d_pos, int64_t n, hipStream_t s, bool null_when_negative = false);
// finishes a column from per-row lengths (-1 = null): offsets, validity, an empty chars buffer of the
// right size; the caller's copy kernel fills the chars, then prefer_offsets32 keeps the narrow offsets
struct Built {
  std::unique_ptr<cs_column> col;
  const int64_t* off;
};
Built column_from_lengths(const int32_t* lens, int64_t rows, bool any_null_possible, hipStream_t s);
void prefer_offsets32(cs_column* c, hipStream_t s);

// profiling hooks (cs_prof_*): time a named kernel launch with HIP events
struct ProfScope {
  const char* name;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(const char* name, hipStream_t s);
  ~ProfScope();
};

}  // namespace cs
