// Column re-arrangement and combination -- the callers either side of the hot path
// (SURVEY.md section 8f-3): create_from_index (NVStringsImpl.cu:209-325), len
// (attrs.cu:32-69), gather / scatter / sublist(step) / sort / order (array.cu:73-360),
// cat / join (combine.cu:31-420).
//
// Every producing op here is the same three steps over OUTPUT rows: a length kernel
// (-1 = null row), the shared lengths -> offsets scan, and a copy kernel in which each
// row is assembled from a few (pointer, length) pieces.  Output columns below 2 GiB of
// chars get int32 offsets.
#include <hip/hip_runtime.h>

#include <cstring>

#include "cs_internal.h"
#include "device_utils.h"
#include "row_ops.h"

using namespace cs;
using namespace csdev;

namespace cs {

// Finishes a column from per-row lengths: offsets (int32 when the chars stay below 2 GiB),
// validity from the negative lengths, the chars buffer; returns the int64 offsets the copy
// kernel should use (always produced: the scan writes int64) -- the narrow form is derived.
__global__ void k_narrow(const int64_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}
Built column_from_lengths(const int32_t* lens, int64_t rows, bool any_null_possible, hipStream_t s) {
  Built b;
  b.col = std::make_unique<cs_column>();
  cs_column* c = b.col.get();
  c->rows = rows;
  c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
  LenMeta meta;  // (the longest row and the largest 64-row span come out of the same pass: no op on the new column pays for them)
  c->nbytes = offsets_from_lengths(lens, rows, ptr<int64_t>(c->offsets), s, nullptr, &meta);
  meta.give(c);
  c->chars = dev_alloc((size_t)c->nbytes, s);
  if (any_null_possible) c->validity = validity_from_lengths(lens, rows, s);
  else c->null_count = 0;
  b.off = ptr<const int64_t>(c->offsets);
  return b;
}
// (after the copy kernel ran) keep the narrow offsets only, when they suffice
void prefer_offsets32(cs_column* c, hipStream_t s) {
  if (c->nbytes >= ((int64_t)1 << 31) || c->rows == 0) return;
  Buf o32 = dev_alloc(sizeof(int32_t) * (c->rows + 1), s);
  hipLaunchKernelGGL(k_narrow, dim3(blocks_for(c->rows + 1)), dim3(kBlock), 0, s, ptr<const int64_t>(c->offsets), c->rows + 1,
                     ptr<int32_t>(o32));
  CS_HIP(hipStreamSynchronize(s));
  c->offsets32 = o32;
  c->offsets = nullptr;
}

template <class T>
struct DevArray {  // caller array made device-visible (copied when it lives on the host)
  Buf tmp;
  const T* d = nullptr;
  DevArray(const T* p, int64_t n, int on_device, hipStream_t s) {
    if (on_device || !p || n == 0) {
      d = p;
      return;
    }
    tmp = dev_alloc(sizeof(T) * (size_t)n, s);
    CS_HIP(hipMemcpyAsync(tmp->p, p, sizeof(T) * (size_t)n, hipMemcpyHostToDevice, s));
    d = ptr<const T>(tmp);
  }
};

// ---- create_from_index ------------------------------------------------------------------------
struct IndexPair {
  const char* p;
  size_t n;
};
__global__ void k_index_lengths(const IndexPair* __restrict__ ix, int64_t rows, int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < rows) lens[r] = ix[r].p ? (int32_t)ix[r].n : -1;
}
__global__ void k_index_copy(const IndexPair* __restrict__ ix, int64_t rows, const int64_t* __restrict__ off, uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < rows && ix[r].p) copy_bytes(out + off[r], reinterpret_cast<const uint8_t*>(ix[r].p), (int)ix[r].n);
}

__global__ void k_index_make(ColView in, IndexPair* __restrict__ ix) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  const bool ok = row_is_valid(in.validity, r);
  ix[r].p = ok ? reinterpret_cast<const char*>(in.chars + in.offsets[r]) : nullptr;
  ix[r].n = ok ? (size_t)(in.offsets[r + 1] - in.offsets[r]) : 0;
}

// ---- len ------------------------------------------------------------------------------------------
__global__ void k_len(ColView in, int32_t* __restrict__ out, unsigned long long* __restrict__ total) {
  // (a capped grid, rows a grid apart: one addition to `total` per workgroup of the grid, not per 256 rows -- same-address
  // atomics retire one per 12 ns, 390K of them for 100M rows)
  long long v = 0;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < in.rows; r += (int64_t)gridDim.x * kBlock) {
    int n = -1;
    if (row_is_valid(in.validity, r)) {
      const uint8_t* p = in.chars + in.offsets[r];
      const int nb = (int)(in.offsets[r + 1] - in.offsets[r]);
      n = 0;
      for (int i = 0; i < nb; ++i) n += (p[i] & 0xC0) != 0x80;
    }
    out[r] = n;
    v += n < 0 ? 0 : n;
  }
  const long long t = block_reduce_sum_ll(v);
  if (threadIdx.x == 0 && t) atomicAdd(total, (unsigned long long)t);
}

// ---- gather -------------------------------------------------------------------------------------
__global__ void k_gather_lengths(ColView in, const int32_t* __restrict__ pos, int64_t n, int neg_ok, int32_t* __restrict__ lens, unsigned* __restrict__ bad) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool oob = false;
  if (i < n) {
    const int64_t p = pos[i];
    oob = p < 0 || p >= in.rows;
    lens[i] = (!oob && row_is_valid(in.validity, p)) ? (int32_t)(in.offsets[p + 1] - in.offsets[p]) : -1;
    if (neg_ok && p < 0) oob = false;
  }
  if (__any(oob) && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}
__global__ void k_gather_copy(ColView in, const int32_t* __restrict__ pos, int64_t n, const int64_t* __restrict__ off, uint8_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t p = pos[i];
  const int len = (int)(off[i + 1] - off[i]);
  if (len > 0) copy_bytes(out + off[i], in.chars + in.offsets[p], len);
}
__global__ void k_mask_flags(const uint8_t* __restrict__ mask, int64_t n, int32_t* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) flags[i] = mask[i] ? 1 : 0;
}
__global__ void k_mask_positions(const uint8_t* __restrict__ mask, int64_t n, const int64_t* __restrict__ slot, int32_t* __restrict__ pos) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n && mask[i]) pos[slot[i]] = (int32_t)i;
}
__global__ void k_sequence(int32_t* __restrict__ out, int64_t n, int64_t start, int64_t step) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = (int32_t)(start + i * step);
}

cs_column* gather_rows(const cs_column* col, const int32_t* d_pos, int64_t n, hipStream_t s, bool null_when_negative) {
  if (n == 0) return make_all_null(0, s);
  if (col->rows == 0 && !null_when_negative) return make_all_null(0, s);
  Buf lens = dev_alloc(sizeof(int32_t) * n, s);
  Buf bad = dev_alloc(sizeof(unsigned), s);
  CS_HIP(hipMemsetAsync(bad->p, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(k_gather_lengths, dim3(blocks_for(n)), dim3(kBlock), 0, s, view_of(col), d_pos, n, null_when_negative ? 1 : 0, ptr<int32_t>(lens), ptr<unsigned>(bad));
  unsigned* hb = (unsigned*)pinned_scratch(sizeof(unsigned));
  CS_HIP(hipMemcpyAsync(hb, bad->p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (*hb) fail(CS_ERR_RANGE, "gather position value out of range");
  Built b = column_from_lengths(ptr<int32_t>(lens), n, col->validity != nullptr || null_when_negative, s);
  hipLaunchKernelGGL(k_gather_copy, dim3(blocks_for(n)), dim3(kBlock), 0, s, view_of(col), d_pos, n, b.off, ptr<uint8_t>(b.col->chars));
  CS_HIP(hipGetLastError());
  prefer_offsets32(b.col.get(), s);
  return b.col.release();
}

// ---- order / sort -------------------------------------------------------------------------------
// The rows' ranks among the distinct strings come from the category build (hash de-dup + sort of
// the distinct keys); the row order is then a stable radix sort of (null class, [byte length,]
// rank) keys with the row index as payload -- equal strings keep their index order, one of the
// orders the reference's (unstable) comparator sort may produce.
__global__ void k_order_keys(ColView in, const int32_t* __restrict__ ranks, int by_len, int by_name, int nulls_low,
                             unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  idx[r] = (uint32_t)r;
  if (!row_is_valid(in.validity, r)) {
    keys[r] = nulls_low ? 0ull : ~0ull;
    return;
  }
  const unsigned long long len = by_len ? (unsigned long long)(in.offsets[r + 1] - in.offsets[r]) + 1ull : 1ull;
  const unsigned long long rk = by_name ? (unsigned long long)(uint32_t)ranks[r] : 0ull;
  keys[r] = (len << 32) | rk;  // (len >= 1 keeps valid rows above the all-zero null key and below the all-ones one)
}
__global__ void k_complement_keys(unsigned long long* __restrict__ keys, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) keys[i] = ~keys[i];
}

}  // namespace cs

extern "C" {

int cs_column_from_index(const void* pairs, int64_t count, int on_device, int sorttype, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!out || count < 0 || (count > 0 && !pairs)) fail(CS_ERR_INVALID_ARG, "create_from_index: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    if (count == 0) {
      *out = make_all_null(0, s);
      return;
    }
    DevArray<IndexPair> ix(static_cast<const IndexPair*>(pairs), count, on_device, s);
    Buf lens = dev_alloc(sizeof(int32_t) * count, s);
    hipLaunchKernelGGL(k_index_lengths, dim3(blocks_for(count)), dim3(kBlock), 0, s, ix.d, count, ptr<int32_t>(lens));
    Built b = column_from_lengths(ptr<int32_t>(lens), count, true, s);
    hipLaunchKernelGGL(k_index_copy, dim3(blocks_for(count)), dim3(kBlock), 0, s, ix.d, count, b.off, ptr<uint8_t>(b.col->chars));
    CS_HIP(hipGetLastError());
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) fail(CS_ERR_INVALID_ARG, "nvstrings::create_from_index bad_device_ptr");
    prefer_offsets32(b.col.get(), s);
    if (sorttype) {  // NVStringsImpl.cu:252-266: ascending, nulls first
      cs_column* sorted = nullptr;
      int st = cs_sort(b.col.get(), sorttype, 1, 1, stream, &sorted);
      if (st != CS_OK) fail(st, cs_last_error());
      *out = sorted;
      return;
    }
    *out = b.col.release();
  });
}

int cs_column_create_index(const cs_column* col, void* pairs, int on_device, cs_stream stream) {
  return guard([&] {
    if (!col || !pairs) fail(CS_ERR_INVALID_ARG, "create_index: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    if (rows == 0) return;
    Buf tmp;
    IndexPair* d = static_cast<IndexPair*>(pairs);
    if (!on_device) {
      tmp = dev_alloc(sizeof(IndexPair) * rows, s);
      d = ptr<IndexPair>(tmp);
    }
    hipLaunchKernelGGL(k_index_make, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), d);
    CS_HIP(hipGetLastError());
    if (!on_device) CS_HIP(hipMemcpyAsync(pairs, d, sizeof(IndexPair) * rows, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
  });
}

int cs_len(const cs_column* col, int32_t* lengths, int on_device, cs_stream stream, int64_t* total) {
  return guard([&] {
    if (!col) fail(CS_ERR_INVALID_ARG, "null column");
    hipStream_t s = S(stream);
    if (total) *total = col->rows;  // attrs.cu:35-36: without an output array the call returns the row count
    if (!lengths || col->rows == 0) return;
    Buf tmp;
    int32_t* d_out = lengths;
    if (!on_device) {
      tmp = dev_alloc(sizeof(int32_t) * col->rows, s);
      d_out = ptr<int32_t>(tmp);
    }
    Buf acc = dev_alloc(8, s);
    CS_HIP(hipMemsetAsync(acc->p, 0, 8, s));
    hipLaunchKernelGGL(k_len, dim3(std::min(blocks_for(col->rows), 8192u)), dim3(kBlock), 0, s, view_of(col), d_out, ptr<unsigned long long>(acc));
    if (!on_device) CS_HIP(hipMemcpyAsync(lengths, d_out, sizeof(int32_t) * col->rows, hipMemcpyDeviceToHost, s));
    int64_t* host = (int64_t*)pinned_scratch(8);
    CS_HIP(hipMemcpyAsync(host, acc->p, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (total) *total = host[0];
  });
}

int cs_gather(const cs_column* col, const int32_t* pos, int64_t n, int on_device, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out || n < 0) fail(CS_ERR_INVALID_ARG, "gather: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    if (!pos || n == 0 || col->rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    DevArray<int32_t> p(pos, n, on_device, s);
    *out = gather_rows(col, p.d, n, s);
  });
}

int cs_gather_mask(const cs_column* col, const uint8_t* mask, int on_device, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "gather: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    if (!mask || rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    DevArray<uint8_t> m(mask, rows, on_device, s);
    Buf flags = dev_alloc(sizeof(int32_t) * rows, s);
    hipLaunchKernelGGL(k_mask_flags, dim3(blocks_for(rows)), dim3(kBlock), 0, s, m.d, rows, ptr<int32_t>(flags));
    Buf slot = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    const int64_t kept = offsets_from_lengths(ptr<int32_t>(flags), rows, ptr<int64_t>(slot), s);
    if (kept == 0) {
      *out = make_all_null(0, s);
      return;
    }
    Buf pos = dev_alloc(sizeof(int32_t) * kept, s);
    hipLaunchKernelGGL(k_mask_positions, dim3(blocks_for(rows)), dim3(kBlock), 0, s, m.d, rows, ptr<const int64_t>(slot), ptr<int32_t>(pos));
    *out = gather_rows(col, ptr<const int32_t>(pos), kept, s);
  });
}

int cs_sublist(const cs_column* col, int64_t start, int64_t end, int64_t step, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "sublist: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    // array.cu:238-260
    const int64_t count = col->rows;
    if (start < 0) start = 0;
    if (end < 0) end = 0;
    if (end > count) end = count;
    if (start > count) start = count;
    if (step == 0) step = 1;
    if (start == end || (step > 0 && start > end) || (step < 0 && start < end)) {
      *out = make_all_null(0, s);
      return;
    }
    if (step == 1) {
      cs_column* c = nullptr;
      int st = cs_column_slice(col, start, end - start, stream, &c);
      if (st != CS_OK) fail(st, cs_last_error());
      *out = c;
      return;
    }
    const int64_t span = end > start ? end - start : start - end;
    const int64_t astep = step > 0 ? step : -step;
    const int64_t n = (span + astep - 1) / astep;
    Buf pos = dev_alloc(sizeof(int32_t) * n, s);
    hipLaunchKernelGGL(k_sequence, dim3(blocks_for(n)), dim3(kBlock), 0, s, ptr<int32_t>(pos), n, start, step);
    *out = gather_rows(col, ptr<const int32_t>(pos), n, s);
  });
}

int cs_order(const cs_column* col, int sorttype, int ascending, int nullfirst, uint32_t* indexes, int on_device, cs_stream stream) {
  return guard([&] {
    if (!col || !indexes) fail(CS_ERR_INVALID_ARG, "order: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    if (rows == 0) return;
    if (rows >= ((int64_t)1 << 31)) fail(CS_ERR_RANGE, "order: more than 2^31 rows");
    const int by_len = (sorttype & 1) != 0, by_name = (sorttype & 2) != 0;  // NVStrings::sorttype: length = 1, name = 2
    cs_category* cat = nullptr;
    const int32_t* ranks = nullptr;
    if (by_name) {
      int st = cs_category_build(col, stream, &cat);
      if (st != CS_OK) fail(st, cs_last_error());
      ranks = cs_category_values_ptr(cat);
    }
    struct CatGuard {
      cs_category* c;
      ~CatGuard() {
        if (c) cs_category_destroy(c);
      }
    } cg{cat};
    Buf keys = dev_alloc(sizeof(unsigned long long) * rows, s);
    Buf idx = dev_alloc(sizeof(uint32_t) * rows, s);
    unsigned long long* k0 = ptr<unsigned long long>(keys);
    uint32_t* i0 = ptr<uint32_t>(idx);
    // nulls come first (or last) whatever the direction: their key is the extreme the direction puts there
    const int nulls_low = (nullfirst != 0) == (ascending != 0);
    hipLaunchKernelGGL(k_order_keys, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), ranks, by_len, by_name, nulls_low, k0, i0);
    // the library's own stable radix sort (cs_radix.hip); a descending order is the ascending order of the
    // complemented keys (equal keys keep their input order either way, as a stable descending sort leaves them)
    if (!ascending) hipLaunchKernelGGL(k_complement_keys, dim3(blocks_for(rows)), dim3(kBlock), 0, s, k0, rows);
    radix_sort_pairs64(reinterpret_cast<uint64_t*>(k0), reinterpret_cast<int32_t*>(i0), rows, s);
    CS_HIP(hipMemcpyAsync(indexes, i0, sizeof(uint32_t) * rows, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
  });
}

int cs_sort(const cs_column* col, int sorttype, int ascending, int nullfirst, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "sort: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    if (col->rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    Buf idx = dev_alloc(sizeof(uint32_t) * col->rows, s);
    int st = cs_order(col, sorttype, ascending, nullfirst, ptr<uint32_t>(idx), 1, stream);
    if (st != CS_OK) fail(st, cs_last_error());
    *out = gather_rows(col, reinterpret_cast<const int32_t*>(ptr<uint32_t>(idx)), col->rows, s);
  });
}

}  // extern "C"

// ---- scatter / cat / join ---------------------------------------------------------------------
namespace cs {

struct PieceCols {
  static constexpr int kMax = 16;
  ColView col[kMax];
};
// scatter: sel[r] = -1 keeps the row, j >= 0 takes row j of `src` (array.cu:157-193), or the scalar (array.cu:202-236)
__global__ void k_scatter_select(const int32_t* __restrict__ pos, int64_t n, int64_t rows, int32_t* __restrict__ sel) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  const int64_t p = pos[j];
  if (p >= 0 && p < rows) atomicMax(sel + p, (int32_t)j);  // several writers of one row: the last one in the list wins
}
__global__ void k_scatter_lengths(ColView in, ColView src, const int32_t* __restrict__ sel, int scalar_len, int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  const int32_t j = sel[r];
  if (j < 0) lens[r] = row_is_valid(in.validity, r) ? (int32_t)(in.offsets[r + 1] - in.offsets[r]) : -1;
  else if (src.offsets) lens[r] = row_is_valid(src.validity, j) ? (int32_t)(src.offsets[j + 1] - src.offsets[j]) : -1;
  else lens[r] = scalar_len;  // (-1: a null scalar)
}
__global__ void k_scatter_copy(ColView in, ColView src, const int32_t* __restrict__ sel, const uint8_t* __restrict__ scalar,
                               const int64_t* __restrict__ off, uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  const int len = (int)(off[r + 1] - off[r]);
  if (len <= 0) return;
  const int32_t j = sel[r];
  const uint8_t* p = j < 0 ? in.chars + in.offsets[r] : (src.offsets ? src.chars + src.offsets[j] : scalar);
  copy_bytes(out + off[r], p, len);
}

// cat (combine.cu:31-291): row = s0 [sep s1 [sep s2 ...]], a null element is replaced by narep; without
// narep a null element makes the row null
struct CatArgs {
  int ncols;
  const uint8_t* sep;
  int sepn;  // -1: no separator
  const uint8_t* narep;
  int narn;  // -1: no replacement
};
__global__ void k_cat_lengths(PieceCols c, CatArgs a, int64_t rows, int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  int total = 0;
  bool isnull = false;
  for (int k = 0; k < a.ncols && !isnull; ++k) {
    const ColView& v = c.col[k];
    if (k > 0 && a.sepn > 0) total += a.sepn;
    if (row_is_valid(v.validity, r)) total += (int)(v.offsets[r + 1] - v.offsets[r]);
    else if (a.narn >= 0) total += a.narn;
    else isnull = true;
  }
  lens[r] = isnull ? -1 : total;
}
__global__ void k_cat_copy(PieceCols c, CatArgs a, int64_t rows, const int32_t* __restrict__ lens, const int64_t* __restrict__ off, uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows || lens[r] < 0) return;
  uint8_t* d = out + off[r];
  for (int k = 0; k < a.ncols; ++k) {
    const ColView& v = c.col[k];
    if (k > 0 && a.sepn > 0) {
      copy_bytes(d, a.sep, a.sepn);
      d += a.sepn;
    }
    if (row_is_valid(v.validity, r)) {
      const int n = (int)(v.offsets[r + 1] - v.offsets[r]);
      copy_bytes(d, v.chars + v.offsets[r], n);
      d += n;
    } else if (a.narn > 0) {
      copy_bytes(d, a.narep, a.narn);
      d += a.narn;
    }
  }
}
// join (combine.cu:293-420): one row; a null row contributes narep or nothing (and then no delimiter either)
__global__ void k_join_lengths(ColView in, int dn, int narn, int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  int n = 0;
  bool delim = r + 1 < in.rows;
  if (row_is_valid(in.validity, r)) n = (int)(in.offsets[r + 1] - in.offsets[r]);
  else if (narn >= 0) n = narn;
  else delim = false;
  lens[r] = n + (delim ? dn : 0);
}
__global__ void k_join_copy(ColView in, const uint8_t* __restrict__ delim, int dn, const uint8_t* __restrict__ narep, int narn,
                            const int64_t* __restrict__ off, uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  uint8_t* d = out + off[r];
  bool put = r + 1 < in.rows;
  if (row_is_valid(in.validity, r)) {
    const int n = (int)(in.offsets[r + 1] - in.offsets[r]);
    copy_bytes(d, in.chars + in.offsets[r], n);
    d += n;
  } else if (narn >= 0) {
    copy_bytes(d, narep, narn);
    d += narn;
  } else {
    put = false;
  }
  if (put && dn > 0) copy_bytes(d, delim, dn);
}

struct HostText {  // small host string on the device (nullptr stays nullptr)
  Buf buf;
  int n = -1;
  HostText(const char* t, hipStream_t s) {
    if (!t) return;
    n = (int)strlen(t);
    buf = dev_alloc((size_t)n + 1, s);
    CS_HIP(hipMemcpyAsync(buf->p, t, (size_t)n + 1, hipMemcpyHostToDevice, s));
  }
  const uint8_t* d() const { return ptr<const uint8_t>(buf); }
};

}  // namespace cs

extern "C" {

static cs_column* scatter_impl(const cs_column* col, const cs_column* src, const char* scalar, const int32_t* pos, int64_t n, int on_device,
                               hipStream_t s) {
  const int64_t rows = col->rows;
  if (rows == 0) return make_all_null(0, s);
  DevArray<int32_t> p(pos, n, on_device, s);
  Buf sel = dev_alloc(sizeof(int32_t) * rows, s);
  CS_HIP(hipMemsetAsync(sel->p, 0xFF, sizeof(int32_t) * rows, s));
  if (n) hipLaunchKernelGGL(k_scatter_select, dim3(blocks_for(n)), dim3(kBlock), 0, s, p.d, n, rows, ptr<int32_t>(sel));
  HostText sc(scalar, s);
  ColView sv{nullptr, nullptr, nullptr, 0};
  if (src) sv = view_of(src);
  Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
  hipLaunchKernelGGL(k_scatter_lengths, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), sv, ptr<const int32_t>(sel), sc.n, ptr<int32_t>(lens));
  Built b = column_from_lengths(ptr<int32_t>(lens), rows, true, s);
  hipLaunchKernelGGL(k_scatter_copy, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), sv, ptr<const int32_t>(sel), sc.d(), b.off,
                     ptr<uint8_t>(b.col->chars));
  CS_HIP(hipGetLastError());
  prefer_offsets32(b.col.get(), s);
  return b.col.release();
}

int cs_scatter(const cs_column* col, const cs_column* strs, const int32_t* pos, int on_device, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !strs || !out) fail(CS_ERR_INVALID_ARG, "scatter: bad arguments");
    if (!pos) fail(CS_ERR_INVALID_ARG, "position parameter cannot be null");
    require_device();
    *out = scatter_impl(col, strs, nullptr, pos, strs->rows, on_device, S(stream));
  });
}
int cs_scatter_scalar(const cs_column* col, const char* str, const int32_t* pos, int64_t n, int on_device, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out || n < 0) fail(CS_ERR_INVALID_ARG, "scatter: bad arguments");
    if (!pos) fail(CS_ERR_INVALID_ARG, "parameter cannot be null");
    require_device();
    *out = scatter_impl(col, nullptr, str, pos, n, on_device, S(stream));
  });
}

int cs_cat(const cs_column* col, const cs_column* const* others, int nothers, const char* separator, const char* narep, cs_stream stream,
           cs_column** out) {
  return guard([&] {
    if (!col || !out || nothers < 0 || (nothers > 0 && !others)) fail(CS_ERR_INVALID_ARG, "cat: bad arguments");
    if (nothers + 1 > PieceCols::kMax) fail(CS_ERR_INVALID_ARG, "cat: more than 15 other columns");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    PieceCols pc{};
    pc.col[0] = view_of(col);
    for (int k = 0; k < nothers; ++k) {
      if (!others[k]) fail(CS_ERR_INVALID_ARG, "cat: null column");
      if (others[k]->rows != rows) fail(CS_ERR_INVALID_ARG, "nvstrings::cat sizes do not match");
      pc.col[k + 1] = view_of(others[k]);
    }
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    HostText sep(separator, s), nar(narep, s);
    CatArgs a{nothers + 1, sep.d(), sep.n, nar.d(), nar.n};
    Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
    hipLaunchKernelGGL(k_cat_lengths, dim3(blocks_for(rows)), dim3(kBlock), 0, s, pc, a, rows, ptr<int32_t>(lens));
    Built b = column_from_lengths(ptr<int32_t>(lens), rows, true, s);
    hipLaunchKernelGGL(k_cat_copy, dim3(blocks_for(rows)), dim3(kBlock), 0, s, pc, a, rows, ptr<const int32_t>(lens), b.off, ptr<uint8_t>(b.col->chars));
    CS_HIP(hipGetLastError());
    prefer_offsets32(b.col.get(), s);
    *out = b.col.release();
  });
}

int cs_join(const cs_column* col, const char* delimiter, const char* narep, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "join: bad arguments");
    if (!delimiter) fail(CS_ERR_INVALID_ARG, "nvstrings::join delimiter cannot be null");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    auto o = std::make_unique<cs_column>();
    o->rows = 1;
    o->null_count = 0;
    Buf two = dev_alloc(sizeof(int64_t) * 2, s);
    int64_t total = 0;
    HostText del(delimiter, s), nar(narep, s);
    Buf off;
    if (rows) {
      Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
      hipLaunchKernelGGL(k_join_lengths, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), del.n, nar.n, ptr<int32_t>(lens));
      off = dev_alloc(sizeof(int64_t) * (rows + 1), s);
      total = offsets_from_lengths(ptr<int32_t>(lens), rows, ptr<int64_t>(off), s);
    }
    o->nbytes = total;
    o->chars = dev_alloc((size_t)total, s);
    if (rows)
      hipLaunchKernelGGL(k_join_copy, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(col), del.d(), del.n, nar.d(), nar.n, ptr<const int64_t>(off),
                         ptr<uint8_t>(o->chars));
    const int64_t h[2] = {0, total};
    CS_HIP(hipMemcpyAsync(two->p, h, sizeof(h), hipMemcpyHostToDevice, s));
    CS_HIP(hipStreamSynchronize(s));
    o->offsets = two;
    *out = o.release();
  });
}

}  // extern "C"
