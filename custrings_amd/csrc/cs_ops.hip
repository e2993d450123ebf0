// Row-parallel string ops: lower/upper, strip, find, contains, replace, split,
// tokenize, ngrams.  One thread per row (256 rows per workgroup), every op is
//   size kernel (+ fused block sums) -> block-sum scan -> offsets -> write kernel
// with the per-row logic in row_ops.h.  Reference call sites are cited at each
// entry point.
#include <hip/hip_runtime.h>

#include <cstring>

#include "cs_internal.h"
#include "device_utils.h"
#include "row_ops.h"

using namespace cs;
using namespace csdev;

namespace cs {
bool tokenize_fast(const cs_column* col, const unsigned char* delims, int ndel, hipStream_t s, cs_column** out);
bool change_case_fast(const cs_column* col, unsigned bit, bool ascii_rule_ok, hipStream_t s, cs_column** out);
bool ngrams_fast(const cs_column* tokens, int n, const unsigned char* sep, int sepn, hipStream_t s, cs_column** out);
}
namespace cs {
extern thread_local int g_replace_plain_only;
extern thread_local unsigned long long g_replace_literal;
extern thread_local int g_replace_literal_len;
}
namespace csrow {
struct CharSet;
}
namespace cs {
bool strip_single(const cs_column* in, const csrow::CharSet& set, int side, hipStream_t s, cs_column* o);
bool strip_write_tiles(const cs_column* in, const csrow::CharSet& set, int side, const int64_t* out_off, uint8_t* out_chars,
                       hipStream_t s);
bool find_tiles(const cs_column* in, const unsigned char* needle, int nb, int mode, int start, int end, int32_t* out32,
                uint8_t* out8, unsigned long long* found, hipStream_t s);
}
using namespace csrow;

namespace cs {
// cs_split.hip: tile kernels for a single-byte delimiter; false = not applicable
bool split_fast(const cs_column* col, const unsigned char* delim, int dlen, int tokens, hipStream_t s,
                std::vector<std::unique_ptr<cs_column>>& cols, bool reverse = false);
}

namespace {

struct Needle {  // small host string copied to the device
  Buf buf;
  int n = 0;
  const uint8_t* d() const { return ptr<const uint8_t>(buf); }
};
Needle upload(const char* s, hipStream_t st) {
  Needle nd;
  nd.n = (int)strlen(s);
  nd.buf = dev_alloc((size_t)nd.n + 1, st);
  CS_HIP(hipMemcpyAsync(nd.buf->p, s, (size_t)nd.n + 1, hipMemcpyHostToDevice, st));
  return nd;
}
}  // namespace
namespace cs {
// the character set of a UTF-8 string of any length; `more` keeps what does not fit the struct alive for the caller's kernels
CharSet make_charset(const char* s, Buf& more, hipStream_t st) {
  std::vector<Char> overflow;
  CharSet cs = charset_from_utf8(reinterpret_cast<const uint8_t*>(s), (int)strlen(s), overflow);
  if (!overflow.empty()) {
    more = dev_alloc(sizeof(Char) * overflow.size(), st);
    CS_HIP(hipMemcpyAsync(more->p, overflow.data(), sizeof(Char) * overflow.size(), hipMemcpyHostToDevice, st));
    CS_HIP(hipStreamSynchronize(st));  // (`overflow` is pageable and leaves scope)
    cs.more = ptr<const Char>(more);
    cs.nmore = (int)overflow.size();
  }
  return cs;
}
}  // namespace cs
namespace {

// ---- generic two-pass driver -------------------------------------------------
// SizeFn:  int  operator()(const uint8_t* row, int len, int64_t r) const
// WriteFn: void operator()(const uint8_t* row, int len, int64_t r, uint8_t* dst) const
template <class SizeFn>
__global__ void k_row_sizes(ColView in, SizeFn f, int32_t* __restrict__ lens,
                            int64_t* __restrict__ block_sums) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int len = -1;
  if (r < in.rows && row_is_valid(in.validity, r)) {
    int64_t b = in.offsets[r];
    len = f(in.chars + b, (int)(in.offsets[r + 1] - b), r);
  }
  if (r < in.rows) lens[r] = len;
  long long t = block_reduce_sum(len < 0 ? 0 : len);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = t;
}
template <class WriteFn>
__global__ void k_row_write(ColView in, WriteFn f, const int64_t* __restrict__ out_off,
                            uint8_t* __restrict__ out_chars) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows || !row_is_valid(in.validity, r)) return;
  int64_t b = in.offsets[r];
  f(in.chars + b, (int)(in.offsets[r + 1] - b), r, out_chars + out_off[r]);
}

template <class SizeFn, class WriteFn>
cs_column* two_pass(const cs_column* in, SizeFn sf, WriteFn wf, hipStream_t s, const char* size_name,
                    const char* write_name) {
  if (in->rows == 0) return make_all_null(0, s);
  auto* out = new cs_column;
  std::unique_ptr<cs_column> holder(out);
  out->rows = in->rows;
  out->validity = in->validity;  // null rows stay null; columns are immutable, so share
  out->null_count = in->null_count;
  unsigned nb = blocks_for(in->rows);
  Buf lens = dev_alloc(sizeof(int32_t) * in->rows, s);
  Buf sums = dev_alloc(sizeof(int64_t) * nb, s);
  {
    ProfScope ps(size_name, s);
    hipLaunchKernelGGL(k_row_sizes<SizeFn>, dim3(nb), dim3(kBlock), 0, s, view_of(in), sf,
                       ptr<int32_t>(lens), ptr<int64_t>(sums));
  }
  out->offsets = dev_alloc(sizeof(int64_t) * (in->rows + 1), s);
  LenMeta meta;
  out->nbytes = offsets_from_lengths(ptr<int32_t>(lens), in->rows, ptr<int64_t>(out->offsets), s, sums, &meta);
  meta.give(out);
  out->chars = dev_alloc((size_t)out->nbytes, s);
  {
    ProfScope ps(write_name, s);
    hipLaunchKernelGGL(k_row_write<WriteFn>, dim3(nb), dim3(kBlock), 0, s, view_of(in), wf,
                       out->d_offsets(), ptr<uint8_t>(out->chars));
  }
  return holder.release();
}

// ---- functors ---------------------------------------------------------------------
struct CaseSize {
  const uint8_t* flags;
  const uint16_t* cases;
  unsigned bit;
  __device__ int operator()(const uint8_t* p, int n, int64_t) const { return row_case_size(p, n, flags, cases, bit); }
};
struct CaseWrite {
  const uint8_t* flags;
  const uint16_t* cases;
  unsigned bit;
  __device__ void operator()(const uint8_t* p, int n, int64_t, uint8_t* o) const { row_case_write(p, n, flags, cases, bit, o); }
};
struct StripSize {
  CharSet set;
  int side;
  __device__ int operator()(const uint8_t* p, int n, int64_t) const {
    int lo, hi;
    row_strip(p, n, set, side, lo, hi);
    return hi - lo;
  }
};
struct StripWrite {
  CharSet set;
  int side;
  __device__ void operator()(const uint8_t* p, int n, int64_t, uint8_t* o) const {
    int lo, hi;
    row_strip(p, n, set, side, lo, hi);
    for (int i = lo; i < hi; ++i) *o++ = p[i];
  }
};
struct ReplaceSize {
  const uint8_t* needle;
  int nb, rb, maxrepl;
  __device__ int operator()(const uint8_t* p, int n, int64_t) const { return row_replace_size(p, n, needle, nb, rb, maxrepl); }
};
struct ReplaceWrite {
  const uint8_t* needle;
  const uint8_t* repl;
  int nb, rb, maxrepl;
  __device__ void operator()(const uint8_t* p, int n, int64_t, uint8_t* o) const { row_replace_write(p, n, needle, nb, repl, rb, maxrepl, o); }
};

// ---- find / contains ------------------------------------------------------------------
__global__ void k_find(ColView in, const uint8_t* __restrict__ needle, int nb, int start, int end,
                       int32_t* __restrict__ out, unsigned long long* __restrict__ found) {
  // (grid-stride: one same-address atomic per workgroup of a capped grid, not per 256 rows)
  int hit = 0;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < in.rows; r += (int64_t)gridDim.x * kBlock) {
    int v = -2;  // null row (find.cu:108)
    if (row_is_valid(in.validity, r)) {
      int64_t b = in.offsets[r];
      v = row_find(in.chars + b, (int)(in.offsets[r + 1] - b), needle, nb, start, end);
    }
    out[r] = v;
    hit += v != -1;  // null rows are counted too (find.cu:112)
  }
  long long t = block_reduce_sum(hit);
  if (threadIdx.x == 0 && t) atomicAdd(found, (unsigned long long)t);
}
__global__ void k_contains(ColView in, const uint8_t* __restrict__ needle, int nb,
                           uint8_t* __restrict__ out, unsigned long long* __restrict__ found) {
  int hits = 0;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < in.rows; r += (int64_t)gridDim.x * kBlock) {
    int hit = 0;
    if (nb > 0 && row_is_valid(in.validity, r)) {
      int64_t b = in.offsets[r];
      hit = find_bytes(in.chars + b, 0, (int)(in.offsets[r + 1] - b), needle, nb) >= 0;
    }
    out[r] = (uint8_t)hit;
    hits += hit;
  }
  long long t = block_reduce_sum(hits);
  if (threadIdx.x == 0 && t) atomicAdd(found, (unsigned long long)t);
}

// ---- split ---------------------------------------------------------------------------------
struct SplitArgs {
  const uint8_t* delim;  // nullptr = whitespace
  int nb;
  int tokens;
  int reverse;  // rsplit: tokens located from the right
  int ncols;    // rsplit on whitespace: the column count of the whole call (known after k_split_count)
};
// the row's tokens through emit(k, lo, hi), forward or (rsplit) from the right
template <class Emit>
__device__ __forceinline__ void split_row_tokens(const SplitArgs& a, const uint8_t* p, int n, int c, Emit&& emit) {
  if (!a.reverse) {
    if (a.delim) row_split_tokens(p, n, a.delim, a.nb, c, emit);
    else row_ws_tokens(p, n, a.tokens, emit);
  } else if (a.delim) {
    row_rsplit_tokens(p, n, a.delim, a.nb, c, emit);
  } else {
    for (int k = 0; k < c; ++k) {
      int lo, hi;
      if (row_ws_rtoken(p, n, a.tokens, c, a.ncols, k, lo, hi)) emit(k, lo, hi);
    }
  }
}
__global__ void k_split_count(ColView in, SplitArgs a, int32_t* __restrict__ counts, int* __restrict__ max_out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int c = 0;
  if (r < in.rows) {
    if (row_is_valid(in.validity, r)) {
      int64_t b = in.offsets[r];
      int n = (int)(in.offsets[r + 1] - b);
      c = a.delim ? row_split_count(in.chars + b, n, a.delim, a.nb, a.tokens)
                  : row_wssplit_count(in.chars + b, n, a.tokens);
    }
    counts[r] = c;
  }
  int m = block_reduce_max(c);
  // (only a workgroup that would raise the maximum issues the same-address atomic)
  if (threadIdx.x == 0 && m > __hip_atomic_load(max_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_out, m);
}
// lens[k * rows + r] = byte length of token k of row r, -1 when the row has no
// such token (null in that column); block sums per column are fused in.
__global__ void k_split_sizes(ColView in, SplitArgs a, const int32_t* __restrict__ counts, int ncols,
                              int32_t* __restrict__ lens, int64_t* __restrict__ block_sums) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t rows = in.rows;
  if (r < rows) {
    for (int k = 0; k < ncols; ++k) lens[(int64_t)k * rows + r] = -1;
    int c = counts[r];
    if (c > 0) {
      int64_t b = in.offsets[r];
      int n = (int)(in.offsets[r + 1] - b);
      auto emit = [&](int k, int lo, int hi) {
        if (k < ncols) lens[(int64_t)k * rows + r] = hi - lo;
      };
      split_row_tokens(a, in.chars + b, n, c, emit);
    }
  }
  for (int k = 0; k < ncols; ++k) {
    int v = (r < rows) ? lens[(int64_t)k * rows + r] : 0;
    long long t = block_reduce_sum(v < 0 ? 0 : v);
    if (threadIdx.x == 0) block_sums[(int64_t)k * gridDim.x + blockIdx.x] = t;
  }
}
struct SplitOut {
  uint8_t* chars;
  const int64_t* offsets;
};
__global__ void k_split_write(ColView in, SplitArgs a, const int32_t* __restrict__ counts, int ncols,
                              const SplitOut* __restrict__ outs) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  int c = counts[r];
  if (c <= 0) return;
  int64_t b = in.offsets[r];
  int n = (int)(in.offsets[r + 1] - b);
  const uint8_t* p = in.chars + b;
  auto emit = [&](int k, int lo, int hi) {
    if (k >= ncols) return;
    uint8_t* o = outs[k].chars + outs[k].offsets[r];
    for (int i = lo; i < hi; ++i) *o++ = p[i];
  };
  split_row_tokens(a, p, n, c, emit);
}

// ---- tokenize --------------------------------------------------------------------------------
struct TokArgs {
  CharSet set;
  int use_set;  // 0 = whitespace
};
__global__ void k_tok_count(ColView in, TokArgs a, int32_t* __restrict__ counts,
                            int64_t* __restrict__ block_sums) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int c = 0;
  if (r < in.rows && row_is_valid(in.validity, r)) {
    int64_t b = in.offsets[r];
    int n = (int)(in.offsets[r + 1] - b);
    if (a.use_set) c = row_set_tokens(in.chars + b, n, a.set, [](int, int, int) {});
    else c = row_ws_count(in.chars + b, n);
  }
  if (r < in.rows) counts[r] = c;
  long long t = block_reduce_sum(c);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = t;
}
// phase 0: token lengths into tok_lens[tok_base[r] + k]; phase 1: copy bytes
template <int PHASE>
__global__ void k_tok_emit(ColView in, TokArgs a, const int64_t* __restrict__ tok_base,
                           int32_t* __restrict__ tok_lens, const int64_t* __restrict__ out_off,
                           uint8_t* __restrict__ out_chars) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows || !row_is_valid(in.validity, r)) return;
  int64_t t0 = tok_base[r];
  if (tok_base[r + 1] == t0) return;
  int64_t b = in.offsets[r];
  int n = (int)(in.offsets[r + 1] - b);
  const uint8_t* p = in.chars + b;
  auto emit = [&](int k, int lo, int hi) {
    if (PHASE == 0) {
      tok_lens[t0 + k] = hi - lo;
    } else {
      uint8_t* o = out_chars + out_off[t0 + k];
      for (int i = lo; i < hi; ++i) *o++ = p[i];
    }
  };
  if (a.use_set) row_set_tokens(p, n, a.set, emit);
  else row_ws_tokens(p, n, 0, emit);
}

// ---- ngrams -----------------------------------------------------------------------------------
__global__ void k_keep_flags(ColView in, int32_t* __restrict__ flags) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < in.rows) flags[r] = row_is_valid(in.validity, r) && in.offsets[r + 1] > in.offsets[r];
}
__global__ void k_keep_scatter(const int32_t* __restrict__ flags, const int64_t* __restrict__ pos,
                               int64_t rows, int32_t* __restrict__ kept) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < rows && flags[r]) kept[pos[r]] = (int32_t)r;
}
__global__ void k_ngram_sizes(ColView in, const int32_t* __restrict__ kept, int64_t count, int n,
                              int sep, int32_t* __restrict__ lens) {
  int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= count) return;
  int sz = (n - 1) * sep;
  for (int k = 0; k < n; ++k) {
    int64_t r = kept[g + k];
    sz += (int)(in.offsets[r + 1] - in.offsets[r]);
  }
  lens[g] = sz;
}
__global__ void k_ngram_write(ColView in, const int32_t* __restrict__ kept, int64_t count, int n,
                              const uint8_t* __restrict__ sep, int sepn,
                              const int64_t* __restrict__ out_off, uint8_t* __restrict__ out_chars) {
  int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= count) return;
  uint8_t* o = out_chars + out_off[g];
  for (int k = 0; k < n; ++k) {
    int64_t r = kept[g + k];
    const uint8_t* p = in.chars + in.offsets[r];
    int len = (int)(in.offsets[r + 1] - in.offsets[r]);
    for (int i = 0; i < len; ++i) *o++ = p[i];
    if (k + 1 < n)
      for (int i = 0; i < sepn; ++i) *o++ = sep[i];
  }
}
// join(sep, narep="") of every row into one string (combine.cu:291-420 as used by ngram.cu:51-52)
__global__ void k_join_sizes(ColView in, int sepn, int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  int len = row_is_valid(in.validity, r) ? (int)(in.offsets[r + 1] - in.offsets[r]) : 0;
  lens[r] = len + (r + 1 < in.rows ? sepn : 0);
}
__global__ void k_join_write(ColView in, const uint8_t* __restrict__ sep, int sepn,
                             const int64_t* __restrict__ pos, uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  uint8_t* o = out + pos[r];
  if (row_is_valid(in.validity, r)) {
    const uint8_t* p = in.chars + in.offsets[r];
    int len = (int)(in.offsets[r + 1] - in.offsets[r]);
    for (int i = 0; i < len; ++i) *o++ = p[i];
  }
  if (r + 1 < in.rows)
    for (int i = 0; i < sepn; ++i) *o++ = sep[i];
}

cs_column* share(const cs_column* in) { return new cs_column(*in); }

template <class T>
T read_back(const Buf& b, hipStream_t s) {
  T* host = (T*)pinned_scratch(sizeof(T));
  CS_HIP(hipMemcpyAsync(host, b->p, sizeof(T), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return *host;
}

}  // namespace

extern "C" {

// NVStrings::lower / upper -- case.cu:31-97 / :100-170
static int change_case(const cs_column* col, unsigned bit, cs_stream stream, cs_column** out, const char* nm) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    // tile kernel (cs_case.hip) when the tables map ASCII the plain way (A-Z <-> a-z only)
    static const bool ascii_rule_ok = [] {
      const uint8_t* f = h_unicode_flags();
      const uint16_t* c = h_charcases();
      for (unsigned b = 0; b < 128; ++b) {
        const unsigned lower = (b >= 'A' && b <= 'Z') ? b + 32 : b, upper = (b >= 'a' && b <= 'z') ? b - 32 : b;
        if (((f[b] & 32) ? c[b] : b) != lower || ((f[b] & 64) ? c[b] : b) != upper) return false;
      }
      return true;
    }();
    if (change_case_fast(col, bit, ascii_rule_ok, S(stream), out)) return;
    *out = two_pass(col, CaseSize{d_unicode_flags(), d_charcases(), bit},
                    CaseWrite{d_unicode_flags(), d_charcases(), bit}, S(stream),
                    bit == 32 ? "k_lower_size" : "k_upper_size", nm);
  });
}
int cs_lower(const cs_column* col, cs_stream stream, cs_column** out) { return change_case(col, 32, stream, out, "k_lower_write"); }
int cs_upper(const cs_column* col, cs_stream stream, cs_column** out) { return change_case(col, 64, stream, out, "k_upper_write"); }

// NVStrings::strip family -- strip.cu:30-199
int cs_strip(const cs_column* col, const char* to_strip, int side, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out || side < 0 || side > 2) fail(CS_ERR_INVALID_ARG, "strip: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    Buf set_more;
    CharSet set = make_charset(to_strip ? to_strip : " \n\t", set_more, s);
    // size pass + scan as for every row-wise op; the write pass runs on row tiles (cs_rows.hip)
    if (col->rows > 0 && !cs::cfg("CS_STRIP_ROWWISE")) {
      auto o = std::make_unique<cs_column>();
      o->rows = col->rows;
      o->validity = col->validity;
      o->null_count = col->null_count;
      if (strip_single(col, set, side, s, o.get())) {  // one pass (cs_rows.hip: k_strip_stream)
        *out = o.release();
        return;
      }
      const unsigned nb = blocks_for(col->rows);
      Buf lens = dev_alloc(sizeof(int32_t) * col->rows, s);
      Buf sums = dev_alloc(sizeof(int64_t) * nb, s);
      {
        ProfScope ps("k_strip_size", s);
        hipLaunchKernelGGL(k_row_sizes<StripSize>, dim3(nb), dim3(kBlock), 0, s, view_of(col), StripSize{set, side},
                           ptr<int32_t>(lens), ptr<int64_t>(sums));
      }
      o->offsets = dev_alloc(sizeof(int64_t) * (col->rows + 1), s);
      LenMeta meta;
      o->nbytes = offsets_from_lengths(ptr<int32_t>(lens), col->rows, ptr<int64_t>(o->offsets), s, sums, &meta);
      meta.give(o.get());
      if (col->plain_bytes == 1) o->plain_bytes = 1;  // (rows cut at character boundaries of a plain column stay plain)
      if (col->high_sample == 0) o->high_sample = 0;
      o->chars = dev_alloc((size_t)o->nbytes, s);
      if (strip_write_tiles(col, set, side, o->d_offsets(), ptr<uint8_t>(o->chars), s)) {
        *out = o.release();
        return;
      }
      {
        ProfScope ps("k_strip_write", s);
        hipLaunchKernelGGL(k_row_write<StripWrite>, dim3(nb), dim3(kBlock), 0, s, view_of(col), StripWrite{set, side},
                           o->d_offsets(), ptr<uint8_t>(o->chars));
      }
      *out = o.release();
      return;
    }
    *out = two_pass(col, StripSize{set, side}, StripWrite{set, side}, S(stream), "k_strip_size", "k_strip_write");
  });
}

// NVStrings::find -- find.cu:75-120
int cs_find(const cs_column* col, const char* str, int start, int end, int32_t* results, int on_device,
            cs_stream stream, int64_t* found) {
  return guard([&] {
    if (found) *found = 0;
    if (!col || !str || !results || col->rows == 0) return;  // reference returns 0 (find.cu:78-79)
    require_device();
    hipStream_t s = S(stream);
    Needle nd = upload(str, s);
    Buf tmp, cnt = dev_alloc(8, s);
    CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
    int32_t* d_out = results;
    if (!on_device) {
      tmp = dev_alloc(sizeof(int32_t) * col->rows, s);
      d_out = ptr<int32_t>(tmp);
    }
    {
      ProfScope ps("k_find", s);
      if (!find_tiles(col, reinterpret_cast<const unsigned char*>(str), nd.n, 0, start, end, d_out, nullptr,
                      ptr<unsigned long long>(cnt), s))
        hipLaunchKernelGGL(k_find, dim3(std::min(blocks_for(col->rows), 8192u)), dim3(kBlock), 0, s, view_of(col), nd.d(),
                           nd.n, start, end, d_out, ptr<unsigned long long>(cnt));
    }
    if (!on_device)
      CS_HIP(hipMemcpyAsync(results, d_out, sizeof(int32_t) * col->rows, hipMemcpyDeviceToHost, s));
    int64_t n = read_back<int64_t>(cnt, s);
    if (found) *found = n;
  });
}

// ---- rfind, find_from, find_multiple, compare, match_strings, startswith, endswith (find.cu:36-72, 123-236, 276-387) ----
// One thread a row (these are not on the benchmarked path).  `op`: 0 rfind(start, end), 1 find_from(starts[], ends[]),
// 2 compare, 3 startswith, 4 endswith; null rows: -2 for the positions, -1 for compare, false for the predicates.
__global__ void k_find_family(ColView in, int op, const uint8_t* __restrict__ needle, int nb, int start, int end, const int32_t* __restrict__ starts,
                              const int32_t* __restrict__ ends, int32_t* __restrict__ out32, uint8_t* __restrict__ out8, unsigned long long* __restrict__ count) {
  int hit = 0;  // (a capped grid, rows a grid apart: one addition to `count` per workgroup -- see k_find)
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < in.rows; r += (int64_t)gridDim.x * kBlock) {
    const bool valid = row_is_valid(in.validity, r);
    const int64_t b = in.offsets[r];
    const uint8_t* p = in.chars + b;
    const int n = valid ? (int)(in.offsets[r + 1] - b) : 0;
    if (op == 0) {
      const int v = valid ? row_rfind_count(p, n, needle, nb, (unsigned)start, end - start) : -2;
      out32[r] = v;
      hit += v != -1;
    } else if (op == 1) {
      int v = -2;
      if (valid) {
        const int pos = starts ? starts[r] : 0;
        v = row_find_count(p, n, needle, nb, (unsigned)pos, ends ? ends[r] - pos : -1);
      }
      out32[r] = v;
      hit += v != -1;
    } else if (op == 2) {
      const int v = valid ? row_compare(p, n, needle, nb) : -1;
      out32[r] = v;
      hit += v == 0;
    } else {
      const bool v = valid && (op == 3 ? row_starts_with(p, n, needle, nb) : row_ends_with(p, n, needle, nb));
      out8[r] = v ? 1 : 0;
      hit += v;
    }
  }
  const long long t = block_reduce_sum(hit);
  if (threadIdx.x == 0 && t) atomicAdd(count, (unsigned long long)t);
}
// match_strings: equal rows (two nulls are equal); find_multiple: out[r * tcount + j] = position of target j in row r
__global__ void k_match_strings(ColView a, ColView b, uint8_t* __restrict__ out, unsigned long long* __restrict__ count) {
  int hit = 0;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < a.rows; r += (int64_t)gridDim.x * kBlock) {
    const bool va = row_is_valid(a.validity, r), vb = row_is_valid(b.validity, r);
    bool same = va == vb;
    if (va && vb) {
      const int64_t oa = a.offsets[r], ob = b.offsets[r];
      same = row_compare(a.chars + oa, (int)(a.offsets[r + 1] - oa), b.chars + ob, (int)(b.offsets[r + 1] - ob)) == 0;
    }
    out[r] = same ? 1 : 0;
    hit += same;
  }
  const long long t = block_reduce_sum(hit);
  if (threadIdx.x == 0 && t) atomicAdd(count, (unsigned long long)t);
}
__global__ void k_find_multiple(ColView in, ColView targets, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= in.rows * targets.rows) return;
  const int64_t r = i / targets.rows, j = i - r * targets.rows;
  int v = -2;
  if (row_is_valid(in.validity, r) && row_is_valid(targets.validity, j)) {
    const int64_t b = in.offsets[r], tb = targets.offsets[j];
    v = row_find_count(in.chars + b, (int)(in.offsets[r + 1] - b), targets.chars + tb, (int)(targets.offsets[j + 1] - tb), 0u, -1);
  }
  out[i] = v;
}
__global__ void k_count_not_minus_one(const int32_t* __restrict__ v, int64_t n, unsigned long long* __restrict__ count) {
  int hit = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) hit += v[i] != -1;
  const long long t = block_reduce_sum(hit);
  if (threadIdx.x == 0 && t) atomicAdd(count, (unsigned long long)t);
}
// shared driver of the one-needle ops
static void find_family(const cs_column* col, int op, const char* str, int start, int end, const int32_t* starts, const int32_t* ends, int bounds_on_device,
                        void* results, int on_device, hipStream_t s, int64_t* found) {
  const int64_t rows = col->rows;
  const bool bools = op >= 3;
  Needle nd = upload(str, s);
  Buf tmp, cnt = dev_alloc(8, s), sbuf, ebuf;
  CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
  void* d_out = results;
  const size_t esz = bools ? 1 : sizeof(int32_t);
  if (!on_device) {
    tmp = dev_alloc(esz * rows, s);
    d_out = tmp->p;
  }
  if (starts && !bounds_on_device) {
    sbuf = dev_alloc(sizeof(int32_t) * rows, s);
    CS_HIP(hipMemcpyAsync(sbuf->p, starts, sizeof(int32_t) * rows, hipMemcpyHostToDevice, s));
    starts = ptr<const int32_t>(sbuf);
  }
  if (ends && !bounds_on_device) {
    ebuf = dev_alloc(sizeof(int32_t) * rows, s);
    CS_HIP(hipMemcpyAsync(ebuf->p, ends, sizeof(int32_t) * rows, hipMemcpyHostToDevice, s));
    ends = ptr<const int32_t>(ebuf);
  }
  {
    ProfScope ps("k_find_family", s);
    hipLaunchKernelGGL(k_find_family, dim3(std::min(blocks_for(rows), 8192u)), dim3(kBlock), 0, s, view_of(col), op, nd.d(), nd.n, start, end, starts, ends,
                       bools ? nullptr : static_cast<int32_t*>(d_out), bools ? static_cast<uint8_t*>(d_out) : nullptr, ptr<unsigned long long>(cnt));
  }
  CS_HIP(hipGetLastError());
  if (!on_device) CS_HIP(hipMemcpyAsync(results, d_out, esz * rows, hipMemcpyDeviceToHost, s));
  const int64_t n = read_back<int64_t>(cnt, s);  // (synchronises: the staged bounds and the needle may go)
  if (found) *found = n;
}

// NVStrings::rfind -- find.cu:163-200
int cs_rfind(const cs_column* col, const char* str, int start, int end, int32_t* results, int on_device, cs_stream stream, int64_t* found) {
  return guard([&] {
    if (found) *found = 0;
    if (!col || !str || !results || col->rows == 0) return;  // the reference returns 0
    require_device();
    find_family(col, 0, str, start < 0 ? 0 : start, end, nullptr, nullptr, 1, results, on_device, S(stream), found);
  });
}
// NVStrings::find_from -- find.cu:123-160 (`starts` / `ends`: one int32 per row, either may be null)
int cs_find_from(const cs_column* col, const char* str, const int32_t* starts, const int32_t* ends, int bounds_on_device, int32_t* results, int on_device,
                 cs_stream stream, int64_t* found) {
  return guard([&] {
    if (found) *found = 0;
    if (!col || !str || !results || col->rows == 0) return;
    require_device();
    find_family(col, 1, str, 0, 0, starts, ends, bounds_on_device, results, on_device, S(stream), found);
  });
}
// NVStrings::compare -- find.cu:36-72 (an empty `str` leaves the results alone and returns 0)
int cs_compare(const cs_column* col, const char* str, int32_t* results, int on_device, cs_stream stream, int64_t* matches) {
  return guard([&] {
    if (matches) *matches = 0;
    // (an empty `str`: the reference returns without writing -- find.cu -- and its callers read an uninitialised buffer; here
    // the comparison with the empty string is computed like any other: 1 for a non-empty row, 0 for an empty one)
    if (!col || !str || !results || col->rows == 0) return;
    require_device();
    find_family(col, 2, str, 0, 0, nullptr, nullptr, 1, results, on_device, S(stream), matches);
  });
}
// NVStrings::startswith / endswith -- find.cu:316-387
int cs_startswith(const cs_column* col, const char* str, uint8_t* results, int on_device, cs_stream stream, int64_t* matches) {
  return guard([&] {
    if (matches) *matches = 0;
    if (!col || !str || !results || col->rows == 0) return;
    require_device();
    find_family(col, 3, str, 0, 0, nullptr, nullptr, 1, results, on_device, S(stream), matches);
  });
}
int cs_endswith(const cs_column* col, const char* str, uint8_t* results, int on_device, cs_stream stream, int64_t* matches) {
  return guard([&] {
    if (matches) *matches = 0;
    if (!col || !str || !results || col->rows == 0) return;
    require_device();
    find_family(col, 4, str, 0, 0, nullptr, nullptr, 1, results, on_device, S(stream), matches);
  });
}
// NVStrings::match_strings -- find.cu:276-314 (sizes must match: std::invalid_argument there)
int cs_match_strings(const cs_column* col, const cs_column* other, uint8_t* results, int on_device, cs_stream stream, int64_t* matches) {
  return guard([&] {
    if (matches) *matches = -1;
    if (!col || !other || !results) fail(CS_ERR_INVALID_ARG, "match_strings: null argument");
    if (matches) *matches = 0;
    if (col->rows == 0) return;
    if (col->rows != other->rows) fail(CS_ERR_INVALID_ARG, "sizes must match");
    require_device();
    hipStream_t s = S(stream);
    Buf tmp, cnt = dev_alloc(8, s);
    CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
    uint8_t* d_out = results;
    if (!on_device) {
      tmp = dev_alloc((size_t)col->rows, s);
      d_out = ptr<uint8_t>(tmp);
    }
    hipLaunchKernelGGL(k_match_strings, dim3(std::min(blocks_for(col->rows), 8192u)), dim3(kBlock), 0, s, view_of(col), view_of(other), d_out, ptr<unsigned long long>(cnt));
    CS_HIP(hipGetLastError());
    if (!on_device) CS_HIP(hipMemcpyAsync(results, d_out, (size_t)col->rows, hipMemcpyDeviceToHost, s));
    const int64_t n = read_back<int64_t>(cnt, s);
    if (matches) *matches = n;
  });
}
// NVStrings::find_multiple -- find.cu:202-234: results[r * targets + j]; the returned count looks at the first rows()
// entries only, as the reference's does
int cs_find_multiple(const cs_column* col, const cs_column* targets, int32_t* results, int on_device, cs_stream stream, int64_t* found) {
  return guard([&] {
    if (found) *found = 0;
    if (!col || !targets || !results || col->rows == 0 || targets->rows == 0) return;
    require_device();
    hipStream_t s = S(stream);
    const int64_t total = col->rows * targets->rows;
    Buf tmp, cnt = dev_alloc(8, s);
    CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
    int32_t* d_out = results;
    if (!on_device) {
      tmp = dev_alloc(sizeof(int32_t) * total, s);
      d_out = ptr<int32_t>(tmp);
    }
    hipLaunchKernelGGL(k_find_multiple, dim3(blocks_for(total)), dim3(kBlock), 0, s, view_of(col), view_of(targets), d_out);
    hipLaunchKernelGGL(k_count_not_minus_one, dim3(std::min(blocks_for(col->rows), 8192u)), dim3(kBlock), 0, s, d_out, col->rows, ptr<unsigned long long>(cnt));
    CS_HIP(hipGetLastError());
    if (!on_device) CS_HIP(hipMemcpyAsync(results, d_out, sizeof(int32_t) * total, hipMemcpyDeviceToHost, s));
    const int64_t n = read_back<int64_t>(cnt, s);
    if (found) *found = n;
  });
}

// NVStrings::contains -- find.cu:237-272
int cs_contains(const cs_column* col, const char* str, uint8_t* results, int on_device, cs_stream stream,
                int64_t* found) {
  return guard([&] {
    if (found) *found = -1;
    if (!col || !str || !results) fail(CS_ERR_INVALID_ARG, "contains: null argument");  // reference returns -1
    if (found) *found = 0;
    if (col->rows == 0) return;
    require_device();
    hipStream_t s = S(stream);
    Needle nd = upload(str, s);
    Buf tmp, cnt = dev_alloc(8, s);
    CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
    uint8_t* d_out = results;
    if (!on_device) {
      tmp = dev_alloc((size_t)col->rows, s);
      d_out = ptr<uint8_t>(tmp);
    }
    {
      ProfScope ps("k_contains", s);
      if (!find_tiles(col, reinterpret_cast<const unsigned char*>(str), nd.n, 1, 0, 0, nullptr, d_out,
                      ptr<unsigned long long>(cnt), s))
        hipLaunchKernelGGL(k_contains, dim3(std::min(blocks_for(col->rows), 8192u)), dim3(kBlock), 0, s, view_of(col), nd.d(),
                           nd.n, d_out, ptr<unsigned long long>(cnt));
    }
    if (!on_device) CS_HIP(hipMemcpyAsync(results, d_out, (size_t)col->rows, hipMemcpyDeviceToHost, s));
    int64_t n = read_back<int64_t>(cnt, s);
    if (found) *found = n;
  });
}

// NVStrings::replace -- modify.cu:109-192
int cs_replace(const cs_column* col, const char* str, const char* repl, int maxrepl, cs_stream stream,
               cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    if (!str || !*str) fail(CS_ERR_INVALID_ARG, "replace parameter cannot be null or empty");
    require_device();
    hipStream_t s = S(stream);
    if (!repl) repl = "";
    // An ASCII needle and a replacement of at most 16 bytes run on the persistent single-pass
    // replace_re kernel (same leftmost, non-overlapping semantics; modify.cu:109-192 restarts at
    // pos + nchars(str), replace.cu:91-92 at the match end): the needle becomes a pattern with
    // its regex metacharacters escaped.
    {
      bool plain = true;
      std::string pattern;
      for (const char* p = str; *p && plain; ++p) {
        const unsigned char c = (unsigned char)*p;
        plain = c >= 0x20 && c < 0x7F;
        if (strchr(".*+?{|()^$[\\", c)) pattern.push_back('\\');
        pattern.push_back((char)c);
      }
      plain = plain && pattern.size() <= 128;
      const size_t rb = strlen(repl);
      if (plain && rb <= 64 && col->rows > 0 && !cs::cfg("CS_REPLACE_ROWWISE") && bytes_plain(col, S(stream))) {
        cs_regex* re = nullptr;
        if (cs_regex_compile(pattern.c_str(), &re) == CS_OK) {
          cs::g_replace_plain_only = 1;  // the single-pass kernel or nothing: this function's own kernels are the fallback
          {  // a needle of up to eight bytes without a border is matched by byte comparison (cs_regex.hip)
            const size_t m = strlen(str);
            bool border_free = m >= 1 && m <= 8;
            for (size_t b = 1; border_free && b < m; ++b) border_free = memcmp(str, str + m - b, b) != 0;
            cs::g_replace_literal = 0;
            cs::g_replace_literal_len = border_free ? (int)m : 0;
            for (size_t i = 0; border_free && i < m; ++i) cs::g_replace_literal |= (unsigned long long)(unsigned char)str[i] << (8 * i);
          }
          const int rc = cs_replace_re(col, re, repl, maxrepl, stream, out);
          cs::g_replace_plain_only = 0;
          cs::g_replace_literal_len = 0;
          cs_regex_destroy(re);
          if (rc == CS_OK) return;
        }
      }
    }
    Needle nd = upload(str, s), rp = upload(repl, s);
    *out = two_pass(col, ReplaceSize{nd.d(), nd.n, rp.n, maxrepl},
                    ReplaceWrite{nd.d(), rp.d(), nd.n, rp.n, maxrepl}, s, "k_replace_size", "k_replace_write");
  });
}

// NVStrings::split(delimiter,maxsplit,results) / split(maxsplit,results) -- split.cu:734-956
static int split_impl(const cs_column* col, const char* delimiter, int maxsplit, cs_stream stream, cs_column*** out_cols,
                      int* ncols_out, bool reverse) {
  return guard([&] {
    if (!col || !out_cols || !ncols_out) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    SplitArgs a{nullptr, 0, maxsplit > 0 ? maxsplit + 1 : 0, 0, 0};
    if (reverse) {
      // Without a split limit the walk from the right meets the same tokens as the walk from the left
      // when the delimiter is whitespace, or ASCII and border-free (no proper prefix that is also a
      // suffix: occurrences cannot overlap, and the character/byte quirk of the token count
      // (row_ops.h) does not arise): those calls take the split kernels.  Everything else -- a
      // limit, a delimiter that can overlap itself, a multi-byte delimiter -- is located from the
      // right by the row-wise kernels below.
      bool same = maxsplit <= 0;
      if (same && delimiter) {
        const int n = (int)strlen(delimiter);
        same = n > 0;
        for (int i = 0; same && i < n; ++i) same = (unsigned char)delimiter[i] < 128;
        for (int b = 1; same && b < n; ++b) same = memcmp(delimiter, delimiter + n - b, (size_t)b) != 0;
      }
      a.reverse = same ? 0 : 1;
    }
    Needle nd;
    if (delimiter) {
      nd = upload(delimiter, s);
      a.delim = nd.d();
      a.nb = nd.n;
    }
    bool ascii_delim = delimiter && nd.n >= 1 && nd.n <= 8;
    for (int i = 0; ascii_delim && i < nd.n; ++i) ascii_delim = (unsigned char)delimiter[i] < 128;
    // rsplit with a limit on a one-byte ASCII delimiter rides the split tile kernels too (the first delimiters of a
    // row are struck from its mask: cs_split.hip TokensT)
    const bool reverse_fast = a.reverse && ascii_delim && nd.n == 1 && a.tokens > 0 && !cs::cfg("CS_RSPLIT_ROWWISE");
    if ((!a.reverse && (!delimiter || ascii_delim)) || reverse_fast) {
      std::vector<std::unique_ptr<cs_column>> fast;
      if (split_fast(col, reinterpret_cast<const unsigned char*>(delimiter), delimiter ? nd.n : 0, a.tokens, s, fast, reverse_fast)) {
        cs_column** arr = (cs_column**)malloc(sizeof(cs_column*) * fast.size());
        if (!arr) fail(CS_ERR_ALLOC, "host allocation failed");
        for (size_t k = 0; k < fast.size(); ++k) arr[k] = fast[k].release();
        *out_cols = arr;
        *ncols_out = (int)fast.size();
        return;
      }
    }
    note_route("split-rowwise");
    int ncols = 0;
    Buf counts;
    const unsigned nb = blocks_for(rows);
    if (rows) {
      counts = dev_alloc(sizeof(int32_t) * rows, s);
      Buf mx = dev_alloc(sizeof(int), s);
      CS_HIP(hipMemsetAsync(mx->p, 0, sizeof(int), s));
      {
        ProfScope ps("k_split_count", s);
        hipLaunchKernelGGL(k_split_count, dim3(nb), dim3(kBlock), 0, s, view_of(col), a,
                           ptr<int32_t>(counts), ptr<int>(mx));
      }
      ncols = read_back<int>(mx, s);
      a.ncols = ncols;
    }
    std::vector<std::unique_ptr<cs_column>> cols;
    if (ncols == 0) {
      // no columns: one all-null column (split.cu:756-757)
      cols.emplace_back(make_all_null(rows, s));
    } else {
      Buf lens = dev_alloc(sizeof(int32_t) * rows * ncols, s);
      Buf sums = dev_alloc(sizeof(int64_t) * (size_t)nb * ncols, s);
      {
        ProfScope ps("k_split_sizes", s);
        hipLaunchKernelGGL(k_split_sizes, dim3(nb), dim3(kBlock), 0, s, view_of(col), a,
                           ptr<int32_t>(counts), ncols, ptr<int32_t>(lens), ptr<int64_t>(sums));
      }
      // per-column offsets: one segmented scan over the fused block sums
      Buf offs = dev_alloc(sizeof(int64_t) * (rows + 1) * ncols, s);
      std::vector<int64_t> totals(ncols);
      offsets_from_lengths_segmented(ptr<int32_t>(lens), rows, ncols, ptr<int64_t>(offs), totals.data(), s);
      std::vector<SplitOut> outs(ncols);
      for (int k = 0; k < ncols; ++k) {
        auto* c = new cs_column;
        cols.emplace_back(c);
        c->rows = rows;
        c->nbytes = totals[k];
        c->chars = dev_alloc((size_t)totals[k], s);
        // each column owns its offsets: copy its segment out of the scan buffer
        c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
        CS_HIP(hipMemcpyAsync(c->offsets->p, ptr<int64_t>(offs) + (int64_t)k * (rows + 1),
                              sizeof(int64_t) * (rows + 1), hipMemcpyDeviceToDevice, s));
        c->validity = validity_from_lengths(ptr<int32_t>(lens) + (int64_t)k * rows, rows, s);
        outs[k] = SplitOut{ptr<uint8_t>(c->chars), c->d_offsets()};
      }
      Buf d_outs = dev_alloc(sizeof(SplitOut) * ncols, s);
      CS_HIP(hipMemcpyAsync(d_outs->p, outs.data(), sizeof(SplitOut) * ncols, hipMemcpyHostToDevice, s));
      {
        ProfScope ps("k_split_write", s);
        hipLaunchKernelGGL(k_split_write, dim3(nb), dim3(kBlock), 0, s, view_of(col), a,
                           ptr<int32_t>(counts), ncols, ptr<const SplitOut>(d_outs));
      }
      CS_HIP(hipStreamSynchronize(s));  // `outs` staging is on the host stack
    }
    cs_column** arr = (cs_column**)malloc(sizeof(cs_column*) * cols.size());
    if (!arr) fail(CS_ERR_ALLOC, "host allocation failed");
    for (size_t k = 0; k < cols.size(); ++k) arr[k] = cols[k].release();
    *out_cols = arr;
    *ncols_out = (int)cols.size();
  });
}

int cs_split(const cs_column* col, const char* delimiter, int maxsplit, cs_stream stream, cs_column*** out_cols,
             int* ncols_out) {
  return split_impl(col, delimiter, maxsplit, stream, out_cols, ncols_out, false);
}
// NVStrings::rsplit(delimiter, maxsplit, results) / rsplit(maxsplit, results) -- split.cu:960-1148
int cs_rsplit(const cs_column* col, const char* delimiter, int maxsplit, cs_stream stream, cs_column*** out_cols,
              int* ncols_out) {
  return split_impl(col, delimiter, maxsplit, stream, out_cols, ncols_out, true);
}

// NVText::tokenize(strs, delimiter) -- tokens.cu:123-155
int cs_tokenize(const cs_column* col, const char* delimiter, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    // byte-parallel tile kernels (cs_tokenize.hip) for whitespace or a few ASCII delimiters
    if (!delimiter || (*delimiter && strlen(delimiter) <= 4)) {
      const unsigned char* d = reinterpret_cast<const unsigned char*>(delimiter ? delimiter : "");
      if (tokenize_fast(col, d, (int)strlen(reinterpret_cast<const char*>(d)), s, out)) return;
    }
    TokArgs a;
    a.use_set = delimiter != nullptr;
    Buf set_more;
    a.set = make_charset(delimiter ? delimiter : "", set_more, s);
    const unsigned nb = blocks_for(rows);
    Buf counts = dev_alloc(sizeof(int32_t) * rows, s);
    Buf sums = dev_alloc(sizeof(int64_t) * nb, s);
    {
      ProfScope ps("k_tok_count", s);
      hipLaunchKernelGGL(k_tok_count, dim3(nb), dim3(kBlock), 0, s, view_of(col), a, ptr<int32_t>(counts),
                         ptr<int64_t>(sums));
    }
    Buf tok_base = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    int64_t ntok = offsets_from_lengths(ptr<int32_t>(counts), rows, ptr<int64_t>(tok_base), s, sums);
    auto* c = new cs_column;
    std::unique_ptr<cs_column> holder(c);
    c->rows = ntok;
    c->null_count = 0;
    c->offsets = dev_alloc(sizeof(int64_t) * (ntok + 1), s);
    if (ntok == 0) {
      CS_HIP(hipMemsetAsync(c->offsets->p, 0, sizeof(int64_t), s));
      c->chars = dev_alloc(0, s);
      *out = holder.release();
      return;
    }
    Buf tok_lens = dev_alloc(sizeof(int32_t) * ntok, s);
    {
      ProfScope ps("k_tok_sizes", s);
      hipLaunchKernelGGL(k_tok_emit<0>, dim3(nb), dim3(kBlock), 0, s, view_of(col), a,
                         ptr<const int64_t>(tok_base), ptr<int32_t>(tok_lens), (const int64_t*)nullptr,
                         (uint8_t*)nullptr);
    }
    LenMeta meta;
    c->nbytes = offsets_from_lengths(ptr<int32_t>(tok_lens), ntok, ptr<int64_t>(c->offsets), s, nullptr, &meta);
    meta.give(c);
    c->chars = dev_alloc((size_t)c->nbytes, s);
    {
      ProfScope ps("k_tok_write", s);
      hipLaunchKernelGGL(k_tok_emit<1>, dim3(nb), dim3(kBlock), 0, s, view_of(col), a,
                         ptr<const int64_t>(tok_base), (int32_t*)nullptr, c->d_offsets(),
                         ptr<uint8_t>(c->chars));
    }
    *out = holder.release();
  });
}

// NVText::create_ngrams -- ngram.cu:32-110
int cs_ngrams(const cs_column* tokens, unsigned ngrams, const char* separator, cs_stream stream,
              cs_column** out) {
  return guard([&] {
    if (!tokens || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    hipStream_t s = S(stream);
    if (ngrams == 0) ngrams = 2;
    if (!separator) separator = "";
    const int64_t rows = tokens->rows;
    if (rows == 0) {
      *out = share(tokens);
      return;
    }
    // tile kernel with closed-form offsets when no row is dropped (cs_ngram.hip)
    if (ngrams_fast(tokens, (int)ngrams, reinterpret_cast<const unsigned char*>(separator), (int)strlen(separator), s, out))
      return;
    Needle sep = upload(separator, s);
    // drop null and empty rows (ngram.cu:48-50)
    Buf flags = dev_alloc(sizeof(int32_t) * rows, s);
    hipLaunchKernelGGL(k_keep_flags, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(tokens),
                       ptr<int32_t>(flags));
    Buf pos = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    int64_t count = offsets_from_lengths(ptr<int32_t>(flags), rows, ptr<int64_t>(pos), s);
    if (count <= (int64_t)ngrams) {
      // join(separator, "") over ALL rows, nulls contributing "" (ngram.cu:51-52)
      Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
      hipLaunchKernelGGL(k_join_sizes, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(tokens), sep.n,
                         ptr<int32_t>(lens));
      Buf jpos = dev_alloc(sizeof(int64_t) * (rows + 1), s);
      int64_t total = offsets_from_lengths(ptr<int32_t>(lens), rows, ptr<int64_t>(jpos), s);
      auto* c = new cs_column;
      std::unique_ptr<cs_column> holder(c);
      c->rows = 1;
      c->nbytes = total;
      c->null_count = 0;
      c->chars = dev_alloc((size_t)total, s);
      c->offsets = dev_alloc(sizeof(int64_t) * 2, s);
      int64_t two[2] = {0, total};
      CS_HIP(hipMemcpyAsync(c->offsets->p, two, sizeof(two), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_join_write, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(tokens), sep.d(),
                         sep.n, ptr<const int64_t>(jpos), ptr<uint8_t>(c->chars));
      CS_HIP(hipStreamSynchronize(s));
      *out = holder.release();
      return;
    }
    if (ngrams == 1) {
      *out = share(tokens);
      return;
    }
    Buf kept = dev_alloc(sizeof(int32_t) * count, s);
    hipLaunchKernelGGL(k_keep_scatter, dim3(blocks_for(rows)), dim3(kBlock), 0, s, ptr<const int32_t>(flags),
                       ptr<const int64_t>(pos), rows, ptr<int32_t>(kept));
    const int64_t ng = count - ngrams + 1;
    Buf lens = dev_alloc(sizeof(int32_t) * ng, s);
    {
      ProfScope ps("k_ngram_sizes", s);
      hipLaunchKernelGGL(k_ngram_sizes, dim3(blocks_for(ng)), dim3(kBlock), 0, s, view_of(tokens),
                         ptr<const int32_t>(kept), ng, (int)ngrams, sep.n, ptr<int32_t>(lens));
    }
    auto* c = new cs_column;
    std::unique_ptr<cs_column> holder(c);
    c->rows = ng;
    c->null_count = 0;
    c->offsets = dev_alloc(sizeof(int64_t) * (ng + 1), s);
    LenMeta meta;
    c->nbytes = offsets_from_lengths(ptr<int32_t>(lens), ng, ptr<int64_t>(c->offsets), s, nullptr, &meta);
    meta.give(c);
    c->chars = dev_alloc((size_t)c->nbytes, s);
    {
      ProfScope ps("k_ngram_write", s);
      hipLaunchKernelGGL(k_ngram_write, dim3(blocks_for(ng)), dim3(kBlock), 0, s, view_of(tokens),
                         ptr<const int32_t>(kept), ng, (int)ngrams, sep.d(), sep.n, c->d_offsets(),
                         ptr<uint8_t>(c->chars));
    }
    *out = holder.release();
  });
}

}  // extern "C"
