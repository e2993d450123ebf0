// Row-major ("record") forms of split and the partition pair:
// NVStrings::split_record / rsplit_record (split.cu:125-700), partition / rpartition
// (split.cu:1165-1361).  The reference returns one NVStrings instance -- and makes one
// device allocation -- per row.  Natively a record result is ONE column holding every
// row's strings in row-major order plus rows+1 list offsets (record r = flat rows
// [list[r], list[r+1])): a count kernel, a scan, a per-token length kernel, a scan, a copy
// kernel -- no allocation inside kernels, five launches whatever the row count.
// The per-row token rules are the reference's, restated on byte offsets (row_ops.h).
#include <hip/hip_runtime.h>

#include <cstring>

#include "cs_internal.h"
#include "device_utils.h"
#include "row_ops.h"

using namespace cs;
using namespace csdev;
using namespace csrow;

namespace {

struct RecArgs {
  ColView in;
  const uint8_t* delim;  // nullptr: whitespace
  int nb;                // delimiter bytes
  int tokens;            // maxsplit + 1, or 0
  int from_right;
};

// emit(k, lo, hi) for the row's record entries; returns how many there are (>= 1 for a valid row).
// Whitespace forms: a row without any token yields ONE empty string (split.cu:393-397, :660-664).
template <class Emit>
__device__ __forceinline__ int record_tokens(const RecArgs& a, const uint8_t* p, int n, Emit&& emit) {
  if (a.delim) {
    const int cnt = row_split_count(p, n, a.delim, a.nb, a.tokens);
    if (a.from_right) {
      // custring_view::rsplit (custring_view.inl:1281-1336): from the right, the leftmost entry takes the rest;
      // when the delimiters run out before the entries do, the unreached entries stay empty
      int hi = n, k = cnt - 1;
      while (k > 0) {
        const int m = rfind_bytes(p, hi, a.delim, a.nb);
        if (m < 0) break;
        emit(k, m + a.nb, m + a.nb < hi ? hi : m + a.nb);
        hi = m;
        --k;
      }
      for (int j = k; j > 0; --j) emit(j, 0, 0);
      emit(0, 0, hi);
    } else {
      row_split_tokens(p, n, a.delim, a.nb, cnt, emit);
    }
    return cnt;
  }
  int made = 0;
  if (!a.from_right) {
    row_ws_tokens(p, n, a.tokens, [&](int k, int lo, int hi) {
      emit(k, lo, hi);
      made = k + 1;
    });
  } else {
    // split.cu:598-640: tokens from the right; the entry that exhausts the limit keeps everything to its left.
    // Entry indices count from the right first: the row's entry count fixes them afterwards.
    const int cnt = row_wssplit_count(p, n, a.tokens);
    int sidx = cnt - 1, epos = n;
    bool spaces = true, any = false;
    for (int pos = n; pos > 0 && sidx >= 0; --pos) {
      const bool sp = p[pos - 1] <= 0x20;
      if (spaces == sp) {
        if (spaces) epos = pos - 1;
        continue;
      }
      if (!spaces) {
        if (cnt - sidx == a.tokens) break;
        emit(sidx--, pos, epos);
        any = true;
        epos = pos - 1;
      }
      spaces = !spaces;
    }
    if (sidx >= 0 && epos > 0) {
      emit(sidx, 0, epos);
      any = true;
    }
    made = any ? cnt : 0;
  }
  if (made == 0) {
    emit(0, 0, 0);
    made = 1;
  }
  return made;
}

__global__ void k_rec_counts(RecArgs a, int32_t* __restrict__ counts) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= a.in.rows) return;
  int c = 0;
  if (row_is_valid(a.in.validity, r)) {
    const int64_t b = a.in.offsets[r];
    c = record_tokens(a, a.in.chars + b, (int)(a.in.offsets[r + 1] - b), [](int, int, int) {});
  }
  counts[r] = c;
}
__global__ void k_rec_lengths(RecArgs a, const int64_t* __restrict__ list, int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= a.in.rows || !row_is_valid(a.in.validity, r)) return;
  const int64_t b = a.in.offsets[r], e0 = list[r];
  const int cnt = (int)(list[r + 1] - e0);
  for (int k = 0; k < cnt; ++k) lens[e0 + k] = 0;  // entries a walk does not reach are empty strings
  record_tokens(a, a.in.chars + b, (int)(a.in.offsets[r + 1] - b), [&](int k, int lo, int hi) {
    if (k < cnt) lens[e0 + k] = hi - lo;
  });
}
__global__ void k_rec_copy(RecArgs a, const int64_t* __restrict__ list, const int64_t* __restrict__ off, uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= a.in.rows || !row_is_valid(a.in.validity, r)) return;
  const int64_t b = a.in.offsets[r], e0 = list[r];
  const int cnt = (int)(list[r + 1] - e0);
  const uint8_t* p = a.in.chars + b;
  record_tokens(a, p, (int)(a.in.offsets[r + 1] - b), [&](int k, int lo, int hi) {
    if (k < cnt && hi > lo) copy_bytes(out + off[e0 + k], p + lo, hi - lo);
  });
}

// partition: entry 3r = head, 3r+1 = delimiter, 3r+2 = tail (split.cu:1165-1361)
struct PartArgs {
  ColView in;
  const uint8_t* delim;
  int nb, from_right;
};
__device__ __forceinline__ void part_of_row(const PartArgs& a, const uint8_t* p, int n, int lo[3], int hi[3]) {
  const int m = n == 0 ? -1 : (a.from_right ? rfind_bytes(p, n, a.delim, a.nb) : find_bytes(p, 0, n, a.delim, a.nb));
  if (m < 0) {
    const int w = a.from_right ? 2 : 0;  // the whole row goes last (rpartition) or first (partition)
    for (int k = 0; k < 3; ++k) lo[k] = hi[k] = 0;
    hi[w] = n;
    return;
  }
  lo[0] = 0, hi[0] = m;
  lo[1] = m, hi[1] = m + a.nb;
  lo[2] = m + a.nb, hi[2] = n;
}
__global__ void k_part_lengths(PartArgs a, int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= a.in.rows) return;
  if (!row_is_valid(a.in.validity, r)) {
    lens[3 * r] = lens[3 * r + 1] = lens[3 * r + 2] = -1;
    return;
  }
  const int64_t b = a.in.offsets[r];
  int lo[3], hi[3];
  part_of_row(a, a.in.chars + b, (int)(a.in.offsets[r + 1] - b), lo, hi);
  for (int k = 0; k < 3; ++k) lens[3 * r + k] = hi[k] - lo[k];
}
__global__ void k_part_copy(PartArgs a, const int64_t* __restrict__ off, uint8_t* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= a.in.rows || !row_is_valid(a.in.validity, r)) return;
  const int64_t b = a.in.offsets[r];
  const uint8_t* p = a.in.chars + b;
  int lo[3], hi[3];
  part_of_row(a, p, (int)(a.in.offsets[r + 1] - b), lo, hi);
  for (int k = 0; k < 3; ++k)
    if (hi[k] > lo[k]) copy_bytes(out + off[3 * r + k], p + lo[k], hi[k] - lo[k]);
}

struct HostBytes {
  Buf buf;
  int n = 0;
  HostBytes(const char* t, hipStream_t s) {
    if (!t) return;
    n = (int)strlen(t);
    buf = dev_alloc((size_t)n + 1, s);
    CS_HIP(hipMemcpyAsync(buf->p, t, (size_t)n + 1, hipMemcpyHostToDevice, s));
  }
  const uint8_t* d() const { return ptr<const uint8_t>(buf); }
};

void records(const cs_column* col, const char* delimiter, int maxsplit, int from_right, int64_t* list_offsets, int on_device, hipStream_t s,
             cs_column** out) {
  const int64_t rows = col->rows;
  if (delimiter && !*delimiter) delimiter = nullptr;  // (an empty delimiter string splits on whitespace, as a null one)
  HostBytes d(delimiter, s);
  RecArgs a{view_of(col), delimiter ? d.d() : nullptr, d.n, maxsplit > 0 ? maxsplit + 1 : 0, from_right};
  Buf list = dev_alloc(sizeof(int64_t) * (rows + 1), s);
  int64_t total = 0;
  if (rows) {
    Buf counts = dev_alloc(sizeof(int32_t) * rows, s);
    hipLaunchKernelGGL(k_rec_counts, dim3(blocks_for(rows)), dim3(kBlock), 0, s, a, ptr<int32_t>(counts));
    total = offsets_from_lengths(ptr<int32_t>(counts), rows, ptr<int64_t>(list), s);
  } else {
    CS_HIP(hipMemsetAsync(list->p, 0, sizeof(int64_t), s));
  }
  if (total >= ((int64_t)1 << 31)) fail(CS_ERR_RANGE, "split_record: more than 2^31 strings");
  if (total == 0) {
    *out = make_all_null(0, s);
  } else {
    Buf lens = dev_alloc(sizeof(int32_t) * total, s);
    hipLaunchKernelGGL(k_rec_lengths, dim3(blocks_for(rows)), dim3(kBlock), 0, s, a, ptr<const int64_t>(list), ptr<int32_t>(lens));
    Built b = column_from_lengths(ptr<int32_t>(lens), total, false, s);
    hipLaunchKernelGGL(k_rec_copy, dim3(blocks_for(rows)), dim3(kBlock), 0, s, a, ptr<const int64_t>(list), b.off, ptr<uint8_t>(b.col->chars));
    CS_HIP(hipGetLastError());
    prefer_offsets32(b.col.get(), s);
    *out = b.col.release();
  }
  CS_HIP(hipMemcpyAsync(list_offsets, list->p, sizeof(int64_t) * (rows + 1), on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
}

}  // namespace

extern "C" {

int cs_split_record(const cs_column* col, const char* delimiter, int maxsplit, int64_t* list_offsets, int on_device, cs_stream stream,
                    cs_column** out) {
  return guard([&] {
    if (!col || !out || !list_offsets) fail(CS_ERR_INVALID_ARG, "split_record: bad arguments");
    require_device();
    records(col, delimiter, maxsplit, 0, list_offsets, on_device, S(stream), out);
  });
}
int cs_rsplit_record(const cs_column* col, const char* delimiter, int maxsplit, int64_t* list_offsets, int on_device, cs_stream stream,
                     cs_column** out) {
  return guard([&] {
    if (!col || !out || !list_offsets) fail(CS_ERR_INVALID_ARG, "rsplit_record: bad arguments");
    require_device();
    records(col, delimiter, maxsplit, 1, list_offsets, on_device, S(stream), out);
  });
}
int cs_partition(const cs_column* col, const char* delimiter, int from_right, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "partition: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    *out = nullptr;
    if (!delimiter || !*delimiter) return;  // the reference returns no results (split.cu:1167-1171)
    const int64_t rows = col->rows;
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    if (3 * rows >= ((int64_t)1 << 31)) fail(CS_ERR_RANGE, "partition: more than 2^31 strings");
    HostBytes d(delimiter, s);
    PartArgs a{view_of(col), d.d(), d.n, from_right};
    Buf lens = dev_alloc(sizeof(int32_t) * 3 * rows, s);
    hipLaunchKernelGGL(k_part_lengths, dim3(blocks_for(rows)), dim3(kBlock), 0, s, a, ptr<int32_t>(lens));
    Built b = column_from_lengths(ptr<int32_t>(lens), 3 * rows, col->validity != nullptr, s);
    hipLaunchKernelGGL(k_part_copy, dim3(blocks_for(rows)), dim3(kBlock), 0, s, a, b.off, ptr<uint8_t>(b.col->chars));
    CS_HIP(hipGetLastError());
    prefer_offsets32(b.col.get(), s);
    *out = b.col.release();
  });
}

}  // extern "C"
