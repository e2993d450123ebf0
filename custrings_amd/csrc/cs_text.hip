// NVText counters and token rewriting: token_count, unique_tokens,
// tokens_counts, replace_tokens, normalize_spaces (tokens.cu:262-716).  All share the
// tokenizer of NVText::tokenize (tokens.cu:41-76): a delimiter is any character of the
// delimiter string, or any byte <= ' ' when there is none; runs of delimiters collapse.
// Thread per row, size pass + scan + write pass (first generation).
#include <hip/hip_runtime.h>

#include <cstring>

#include "cs_internal.h"
#include "device_utils.h"
#include "row_ops.h"

using namespace cs;
using namespace csdev;
using namespace csrow;

namespace {

struct Tokenizer {
  int has_set;
  CharSet set;
};
Tokenizer make_tokenizer(const char* delimiter, Buf& more, hipStream_t st) {
  Tokenizer t;
  t.has_set = delimiter != nullptr && *delimiter != 0;
  t.set = make_charset(t.has_set ? delimiter : "", more, st);
  return t;
}
template <class Emit>
__device__ __forceinline__ int row_tokens(const Tokenizer& t, const uint8_t* p, int n, Emit&& emit) {
  if (t.has_set) return row_set_tokens(p, n, t.set, emit);
  int k = 0;
  row_ws_tokens(p, n, 0, [&](int i, int lo, int hi) {
    emit(i, lo, hi);
    k = i + 1;
  });
  return k;
}
__device__ __forceinline__ bool same_bytes(const uint8_t* a, const uint8_t* b, int n) {
  for (int i = 0; i < n; ++i)
    if (a[i] != b[i]) return false;
  return true;
}
// index of the first row of `t` equal to the n bytes at p, or -1
__device__ __forceinline__ int match_token(const ColView& t, const uint8_t* p, int n) {
  for (int64_t k = 0; k < t.rows; ++k) {
    if (!row_is_valid(t.validity, k)) continue;
    const int64_t b = t.offsets[k];
    if ((int)(t.offsets[k + 1] - b) == n && same_bytes(t.chars + b, p, n)) return (int)k;
  }
  return -1;
}

__global__ void k_token_count(ColView in, Tokenizer t, uint32_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  int c = 0;
  if (row_is_valid(in.validity, r)) {
    const int64_t b = in.offsets[r];
    c = row_tokens(t, in.chars + b, (int)(in.offsets[r + 1] - b), [](int, int, int) {});
  }
  out[r] = (uint32_t)c;
}
__global__ void k_tokens_counts(ColView in, ColView tk, Tokenizer t, uint32_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  uint32_t* mine = out + r * tk.rows;
  for (int64_t k = 0; k < tk.rows; ++k) mine[k] = 0;
  if (!row_is_valid(in.validity, r)) return;
  const int64_t b = in.offsets[r];
  const uint8_t* p = in.chars + b;
  row_tokens(t, p, (int)(in.offsets[r + 1] - b), [&](int, int lo, int hi) {
    const int k = match_token(tk, p + lo, hi - lo);
    if (k >= 0) ++mine[k];
  });
}
// replace_tokens: WRITE = false sizes, true bytes
template <bool WRITE>
__global__ void k_replace_tokens(ColView in, ColView tg, ColView rp, Tokenizer t, int32_t* __restrict__ lens, const int64_t* __restrict__ off,
                                 uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  if (!row_is_valid(in.validity, r)) {
    if (!WRITE) lens[r] = -1;
    return;
  }
  const int64_t b = in.offsets[r];
  const int n = (int)(in.offsets[r + 1] - b);
  const uint8_t* p = in.chars + b;
  uint8_t* o = WRITE ? out + off[r] : nullptr;
  int total = n, copied = 0;
  row_tokens(t, p, n, [&](int, int lo, int hi) {
    const int k = match_token(tg, p + lo, hi - lo);
    if (k < 0) return;
    const int64_t rr = rp.rows == 1 ? 0 : k;
    const bool has = row_is_valid(rp.validity, rr);
    const int rn = has ? (int)(rp.offsets[rr + 1] - rp.offsets[rr]) : 0;
    total += rn - (hi - lo);
    if (WRITE) {
      copy_bytes(o, p + copied, lo - copied);
      o += lo - copied;
      if (rn) copy_bytes(o, rp.chars + rp.offsets[rr], rn);
      o += rn;
      copied = hi;
    }
  });
  if (WRITE) copy_bytes(o, p + copied, n - copied);
  else lens[r] = total;
}
template <bool WRITE>
__global__ void k_normalize_spaces(ColView in, Tokenizer t, int32_t* __restrict__ lens, const int64_t* __restrict__ off, uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  if (!row_is_valid(in.validity, r)) {
    if (!WRITE) lens[r] = -1;
    return;
  }
  const int64_t b = in.offsets[r];
  const uint8_t* p = in.chars + b;
  uint8_t* o = WRITE ? out + off[r] : nullptr;
  int total = 0;
  row_tokens(t, p, (int)(in.offsets[r + 1] - b), [&](int k, int lo, int hi) {
    if (k > 0) {
      if (WRITE) *o++ = ' ';
      ++total;
    }
    if (WRITE) {
      copy_bytes(o, p + lo, hi - lo);
      o += hi - lo;
    }
    total += hi - lo;
  });
  if (!WRITE) lens[r] = total;
}

// tokenize with whole-string delimiters (NVText::tokenize(strs, delims), tokens.cu:158-260): at every byte the
// delimiters are tried in order (null and empty ones skipped); the text between two delimiter occurrences is a
// token, empty pieces are dropped.  Pass 0 counts the kept pieces, pass 1 leaves (pointer, length) pairs.
struct BytePair {
  const char* p;
  size_t n;
};
template <bool WRITE>
__global__ void k_tokenize_multi(ColView in, ColView dl, int32_t* __restrict__ counts, const int64_t* __restrict__ base, BytePair* __restrict__ pairs) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  int kept = 0;
  if (row_is_valid(in.validity, r)) {
    const int64_t b = in.offsets[r];
    const int n = (int)(in.offsets[r + 1] - b);
    const uint8_t* p = in.chars + b;
    BytePair* mine = WRITE ? pairs + base[r] : nullptr;
    int spos = 0, i = 0;
    auto piece = [&](int lo, int hi) {
      if (hi <= lo) return;
      if (WRITE) mine[kept] = BytePair{reinterpret_cast<const char*>(p + lo), (size_t)(hi - lo)};
      ++kept;
    };
    while (i < n) {
      int step = 1;
      for (int64_t k = 0; k < dl.rows; ++k) {
        if (!row_is_valid(dl.validity, k)) continue;
        const int64_t db = dl.offsets[k];
        const int dn = (int)(dl.offsets[k + 1] - db);
        if (dn == 0 || i + dn > n || !same_bytes(dl.chars + db, p + i, dn)) continue;
        piece(spos, i);
        step = dn;
        spos = i + dn;
        break;
      }
      i += step;
    }
    piece(spos, n);
  }
  if (!WRITE) counts[r] = kept;
}

}  // namespace

extern "C" {

int cs_token_count(const cs_column* col, const char* delimiter, uint32_t* results, int on_device, cs_stream stream) {
  return guard([&] {
    if (!col || !results) fail(CS_ERR_INVALID_ARG, "token_count: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    if (rows == 0) return;
    Buf set_more;
    Tokenizer t = make_tokenizer(delimiter, set_more, s);
    Buf tmp;
    uint32_t* d = results;
    if (!on_device) {
      tmp = dev_alloc(sizeof(uint32_t) * rows, s);
      d = ptr<uint32_t>(tmp);
    }
    hipLaunchKernelGGL(k_token_count, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), t, d);
    CS_HIP(hipGetLastError());
    if (!on_device) CS_HIP(hipMemcpyAsync(results, d, sizeof(uint32_t) * rows, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
  });
}

int cs_unique_tokens(const cs_column* col, const char* delimiter, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "unique_tokens: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    cs_column* toks = nullptr;
    int st = cs_tokenize(col, (delimiter && *delimiter) ? delimiter : nullptr, stream, &toks);
    if (st != CS_OK) fail(st, cs_last_error());
    std::unique_ptr<cs_column> hold(toks);
    std::unique_ptr<cs_category> cat(category_build(toks, s));
    *out = cat->keys.release();  // sorted unique tokens (tokens are never null, so there is no null key)
  });
}

int cs_tokens_counts(const cs_column* col, const cs_column* tokens, const char* delimiter, uint32_t* results, int on_device, cs_stream stream) {
  return guard([&] {
    if (!col || !tokens) fail(CS_ERR_INVALID_ARG, "tokens_counts: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows, tc = tokens->rows;
    if (!results || rows == 0 || tc == 0) return;  // tokens.cu:442-443
    Buf set_more;
    Tokenizer t = make_tokenizer(delimiter, set_more, s);
    Buf tmp;
    uint32_t* d = results;
    if (!on_device) {
      tmp = dev_alloc(sizeof(uint32_t) * rows * tc, s);
      d = ptr<uint32_t>(tmp);
    }
    hipLaunchKernelGGL(k_tokens_counts, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), view_of(tokens), t, d);
    CS_HIP(hipGetLastError());
    if (!on_device) CS_HIP(hipMemcpyAsync(results, d, sizeof(uint32_t) * rows * tc, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
  });
}

int cs_replace_tokens(const cs_column* col, const cs_column* targets, const cs_column* repls, const char* delimiter, cs_stream stream,
                      cs_column** out) {
  return guard([&] {
    if (!col || !targets || !repls || !out) fail(CS_ERR_INVALID_ARG, "replace_tokens: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    if (rows == 0 || targets->rows == 0) {  // tokens.cu:567-568: a copy
      *out = new cs_column(*col);
      return;
    }
    if (repls->rows == 0) fail(CS_ERR_INVALID_ARG, "replace-tokens: no replacement given");
    if (repls->rows > 1 && repls->rows != targets->rows)
      fail(CS_ERR_INTERNAL, "replace-tokens tokens and replacements must have the same number of strings");  // (std::runtime_error, tokens.cu:570)
    Buf set_more;
    Tokenizer t = make_tokenizer(delimiter, set_more, s);
    Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
    hipLaunchKernelGGL(k_replace_tokens<false>, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), view_of(targets), view_of(repls), t,
                       ptr<int32_t>(lens), (const int64_t*)nullptr, (uint8_t*)nullptr);
    Built b = column_from_lengths(ptr<int32_t>(lens), rows, col->validity != nullptr, s);
    if (b.col->nbytes == 0) {  // tokens.cu:612-613: nothing to hold -> no instance
      *out = nullptr;
      return;
    }
    hipLaunchKernelGGL(k_replace_tokens<true>, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), view_of(targets), view_of(repls), t,
                       (int32_t*)nullptr, b.off, ptr<uint8_t>(b.col->chars));
    CS_HIP(hipGetLastError());
    prefer_offsets32(b.col.get(), s);
    *out = b.col.release();
  });
}

int cs_normalize_spaces(const cs_column* col, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "normalize_spaces: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    if (rows == 0) {
      *out = new cs_column(*col);
      return;
    }
    Buf set_more;
    Tokenizer t = make_tokenizer(nullptr, set_more, s);
    Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
    hipLaunchKernelGGL(k_normalize_spaces<false>, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), t, ptr<int32_t>(lens),
                       (const int64_t*)nullptr, (uint8_t*)nullptr);
    Built b = column_from_lengths(ptr<int32_t>(lens), rows, col->validity != nullptr, s);
    if (b.col->nbytes == 0) {  // tokens.cu:703-704
      *out = nullptr;
      return;
    }
    hipLaunchKernelGGL(k_normalize_spaces<true>, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), t, (int32_t*)nullptr, b.off,
                       ptr<uint8_t>(b.col->chars));
    CS_HIP(hipGetLastError());
    prefer_offsets32(b.col.get(), s);
    *out = b.col.release();
  });
}


int cs_tokenize_multi(const cs_column* col, const cs_column* delimiters, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !delimiters || !out) fail(CS_ERR_INVALID_ARG, "tokenize: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    if (delimiters->rows == 0) {  // tokens.cu:160-162: whitespace tokenize
      int st = cs_tokenize(col, nullptr, stream, out);
      if (st != CS_OK) fail(st, cs_last_error());
      return;
    }
    const int64_t rows = col->rows;
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    Buf counts = dev_alloc(sizeof(int32_t) * rows, s);
    hipLaunchKernelGGL(k_tokenize_multi<false>, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), view_of(delimiters), ptr<int32_t>(counts),
                       (const int64_t*)nullptr, (BytePair*)nullptr);
    Buf base = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    const int64_t total = offsets_from_lengths(ptr<int32_t>(counts), rows, ptr<int64_t>(base), s);
    if (total == 0) {
      *out = make_all_null(0, s);
      return;
    }
    Buf pairs = dev_alloc(sizeof(BytePair) * total, s);
    hipLaunchKernelGGL(k_tokenize_multi<true>, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), view_of(delimiters), (int32_t*)nullptr,
                       ptr<const int64_t>(base), ptr<BytePair>(pairs));
    CS_HIP(hipGetLastError());
    int st = cs_column_from_index(pairs->p, total, 1, 0, stream, out);
    if (st != CS_OK) fail(st, cs_last_error());
  });
}

}  // extern "C"
