// Category (dictionary) build: sorted unique keys + int32 code per row.
//
// Contract (NVCategory.cu:220-304): keys are the distinct rows in ascending
// order -- null first, then unsigned bytewise order with "shorter is less"
// (custring.inl:240-261) -- and values[r] is the index of row r's key.  The
// reference gets there by comparator-sorting every row; the outputs depend only
// on the sorted set of distinct keys and each row's rank in it, so here rows are
// first de-duplicated through an open-addressing hash table in HBM (one CAS per
// row, full byte compare on every hit, so there are no false merges), and only
// the U distinct representatives are sorted (bitonic network on an 8-byte
// big-endian prefix with a full-compare tie break).
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "device_utils.h"

using namespace cs;
using namespace csdev;

struct cs_category {
  std::unique_ptr<cs_column> keys;
  Buf values;  // int32[rows]
  int64_t rows = 0;
};

namespace {

// little-endian dword k of a row, zero beyond its end (aligned rows load whole dwords)
__device__ __forceinline__ uint32_t row_word(const uint8_t* p, int n, int k, bool aligned) {
  const int i = 4 * k;
  if (aligned && i + 4 <= n) return *reinterpret_cast<const uint32_t*>(p + i);
  uint32_t w = 0;
  for (int j = 0; j < 4; ++j)
    if (i + j < n) w |= (uint32_t)p[i + j] << (8 * j);
  return w;
}
// Any well-mixed hash will do: keys and codes depend only on the sorted key set.
__device__ __forceinline__ uint32_t hash_bytes(const uint8_t* p, int n) {
  const bool aligned = ((uintptr_t)p & 3) == 0;
  uint32_t h = 0x9E3779B9u ^ (uint32_t)n;
  for (int k = 0; 4 * k < n; ++k) {
    uint32_t w = row_word(p, n, k, aligned) * 0xCC9E2D51u;
    w = (w << 15) | (w >> 17);
    h ^= w * 0x1B873593u;
    h = ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
  }
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  return h ^ (h >> 16);
}
__device__ __forceinline__ bool same_bytes(const uint8_t* a, const uint8_t* b, int n) {
  if ((((uintptr_t)a | (uintptr_t)b) & 3) == 0) {
    int i = 0;
    for (; i + 4 <= n; i += 4)
      if (*reinterpret_cast<const uint32_t*>(a + i) != *reinterpret_cast<const uint32_t*>(b + i)) return false;
    for (; i < n; ++i)
      if (a[i] != b[i]) return false;
    return true;
  }
  for (int i = 0; i < n; ++i)
    if (a[i] != b[i]) return false;
  return true;
}
// custr::compare: unsigned bytewise, shorter is less
__device__ __forceinline__ int compare_rows(const ColView& in, int64_t ra, int64_t rb) {
  int64_t oa = in.offsets[ra], ob = in.offsets[rb];
  int la = (int)(in.offsets[ra + 1] - oa), lb = (int)(in.offsets[rb + 1] - ob);
  const uint8_t *pa = in.chars + oa, *pb = in.chars + ob;
  int m = la < lb ? la : lb;
  for (int i = 0; i < m; ++i)
    if (pa[i] != pb[i]) return (int)pa[i] - (int)pb[i];
  return la - lb;
}

constexpr int kProbeLimit = 128;  // longer probe runs mean the table is too small: the host retries with a bigger one
__global__ void k_cat_insert(ColView in, int32_t* __restrict__ table, uint32_t mask,
                             int32_t* __restrict__ slot_of_row, int* __restrict__ has_null, int* __restrict__ overflow,
                             int probe_limit) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  if (!row_is_valid(in.validity, r)) {
    slot_of_row[r] = -1;
    *has_null = 1;
    return;
  }
  int64_t b = in.offsets[r];
  int n = (int)(in.offsets[r + 1] - b);
  const uint8_t* p = in.chars + b;
  uint32_t slot = hash_bytes(p, n) & mask;
  int probes = 0;
  for (;;) {
    // Look before the CAS: a slot only ever changes from -1 to its final row, so a non-empty
    // value read here is final and the common case (key already present) needs no atomic at
    // all -- with a skewed key distribution the CASes of a hot key would otherwise queue on one
    // address (about 10 ns each: 100 ms for a key that 7 % of 125M rows share).
    int32_t cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == -1) cur = atomicCAS(&table[slot], -1, (int32_t)r);
    if (cur == -1 || cur == (int32_t)r) break;  // this row represents the key
    int64_t cb = in.offsets[cur];
    if ((int)(in.offsets[cur + 1] - cb) == n && same_bytes(in.chars + cb, p, n)) break;
    slot = (slot + 1) & mask;
    if (++probes > probe_limit) {
      *overflow = 1;
      return;
    }
  }
  slot_of_row[r] = (int32_t)slot;
}
__global__ void k_cat_flags(const int32_t* __restrict__ table, int64_t cap, int32_t* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < cap) flags[i] = table[i] >= 0;
}
// sort records: prefix[i] = first 8 key bytes big-endian, item[i] = slot id;
// padding up to the power of two sorts last
__global__ void k_cat_records(ColView in, const int32_t* __restrict__ table, const int32_t* __restrict__ flags,
                              const int64_t* __restrict__ pos, int64_t cap, int64_t padded, int64_t uniq,
                              uint64_t* __restrict__ prefix, int32_t* __restrict__ item) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < cap && flags[i]) {
    int64_t row = table[i];
    int64_t b = in.offsets[row];
    int n = (int)(in.offsets[row + 1] - b);
    uint64_t k = 0;
    for (int j = 0; j < 8; ++j) k = (k << 8) | (j < n ? in.chars[b + j] : 0);
    prefix[pos[i]] = k;
    item[pos[i]] = (int32_t)i;
  }
  if (i >= uniq && i < padded) {
    prefix[i] = ~0ull;
    item[i] = -1;
  }
}
__global__ void k_bitonic_step(ColView in, const int32_t* __restrict__ table, uint64_t* __restrict__ prefix,
                               int32_t* __restrict__ item, int64_t padded, int64_t j, int64_t k) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= padded) return;
  int64_t l = i ^ j;
  if (l <= i) return;
  uint64_t pa = prefix[i], pb = prefix[l];
  int32_t ia = item[i], ib = item[l];
  int cmp;
  if (ia < 0 || ib < 0) cmp = (ia < 0) - (ib < 0);  // padding is greatest
  else if (pa != pb) cmp = pa < pb ? -1 : 1;
  else cmp = compare_rows(in, table[ia], table[ib]);
  bool ascending = (i & k) == 0;
  if ((cmp > 0) == ascending && cmp != 0) {
    prefix[i] = pb;
    prefix[l] = pa;
    item[i] = ib;
    item[l] = ia;
  }
}
__global__ void k_cat_ranks(const int32_t* __restrict__ item, int64_t uniq, int shift,
                            int32_t* __restrict__ rank_of_slot) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < uniq) rank_of_slot[item[i]] = (int32_t)i + shift;
}
__global__ void k_cat_values(const int32_t* __restrict__ slot_of_row, const int32_t* __restrict__ rank_of_slot,
                             int64_t rows, int32_t* __restrict__ values) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  int32_t s = slot_of_row[r];
  values[r] = s < 0 ? 0 : rank_of_slot[s];  // a null row maps to key 0 (the null key)
}
__global__ void k_key_sizes(ColView in, const int32_t* __restrict__ table, const int32_t* __restrict__ item,
                            int64_t nkeys, int shift, int32_t* __restrict__ lens) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nkeys) return;
  if (i < shift) {
    lens[i] = -1;  // the null key
    return;
  }
  int64_t row = table[item[i - shift]];
  lens[i] = (int32_t)(in.offsets[row + 1] - in.offsets[row]);
}
__global__ void k_key_copy(ColView in, const int32_t* __restrict__ table, const int32_t* __restrict__ item,
                           int64_t nkeys, int shift, const int64_t* __restrict__ out_off,
                           uint8_t* __restrict__ out_chars) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nkeys || i < shift) return;
  int64_t row = table[item[i - shift]];
  const uint8_t* p = in.chars + in.offsets[row];
  int n = (int)(in.offsets[row + 1] - in.offsets[row]);
  uint8_t* o = out_chars + out_off[i];
  for (int k = 0; k < n; ++k) o[k] = p[k];
}
__global__ void k_remap(const int32_t* __restrict__ codes, int64_t n, const int32_t* __restrict__ table,
                        int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int32_t v = codes[i];
  out[i] = v < 0 ? v : table[v];
}

template <class T>
T read_back(const void* d, hipStream_t s) {
  T* host = (T*)pinned_scratch(sizeof(T));
  CS_HIP(hipMemcpyAsync(host, d, sizeof(T), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return *host;
}

cs_category* build(const cs_column* col, hipStream_t s) {
  auto cat = std::make_unique<cs_category>();
  const int64_t rows = col->rows;
  cat->rows = rows;
  if (rows == 0) {
    cat->keys.reset(make_all_null(0, s));
    cat->values = dev_alloc(0, s);
    return cat.release();
  }
  if (rows >= (1LL << 31) - 1) fail(CS_ERR_RANGE, "category: more than 2^31 rows in one column");
  // Table capacity: most columns have far fewer distinct keys than rows, and a table that stays
  // in the caches makes every probe and every later pass over it cheap -- start with 4M slots
  // and retry with room for all-distinct rows only if a probe run gets long.
  int64_t full = 256;
  while (full < 2 * rows) full <<= 1;
  ColView in = view_of(col);
  Buf slot_of_row = dev_alloc(sizeof(int32_t) * rows, s);
  Buf flags_d = dev_alloc(2 * sizeof(int), s);  // [0] has_null, [1] overflow
  Buf table;
  int first_log2 = 22;
  if (const char* e = getenv("CS_CAT_FIRST_LOG2")) first_log2 = std::max(4, std::min(30, atoi(e)));  // tests: force retries
  int64_t cap = std::min<int64_t>(full, getenv("CS_CAT_FULL_TABLE") ? full : (int64_t)1 << first_log2);
  for (;;) {
    table = dev_alloc(sizeof(int32_t) * cap, s);
    CS_HIP(hipMemsetAsync(table->p, 0xFF, sizeof(int32_t) * cap, s));
    CS_HIP(hipMemsetAsync(flags_d->p, 0, 2 * sizeof(int), s));
    {
      ProfScope ps("k_cat_insert", s);
      hipLaunchKernelGGL(k_cat_insert, dim3(blocks_for(rows)), dim3(kBlock), 0, s, in, ptr<int32_t>(table),
                         (uint32_t)(cap - 1), ptr<int32_t>(slot_of_row), ptr<int>(flags_d), ptr<int>(flags_d) + 1,
                         cap == full ? 0x7fffffff : kProbeLimit);
    }
    if (cap == full) break;  // room for all-distinct rows: no limit applied, nothing to retry
    int* h = (int*)pinned_scratch(2 * sizeof(int));
    CS_HIP(hipMemcpyAsync(h, flags_d->p, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (!h[1]) break;
    cap = std::min<int64_t>(full, cap * 16);
  }
  Buf has_null_d = flags_d;
  // compact the occupied slots
  Buf flags = dev_alloc(sizeof(int32_t) * cap, s);
  hipLaunchKernelGGL(k_cat_flags, dim3(blocks_for(cap)), dim3(kBlock), 0, s, ptr<const int32_t>(table), cap,
                     ptr<int32_t>(flags));
  Buf pos = dev_alloc(sizeof(int64_t) * (cap + 1), s);
  const int64_t uniq = offsets_from_lengths(ptr<int32_t>(flags), cap, ptr<int64_t>(pos), s);
  const int shift = read_back<int>(has_null_d->p, s) ? 1 : 0;
  int64_t padded = 1;
  while (padded < uniq) padded <<= 1;
  Buf prefix = dev_alloc(sizeof(uint64_t) * padded, s);
  Buf item = dev_alloc(sizeof(int32_t) * padded, s);
  hipLaunchKernelGGL(k_cat_records, dim3(blocks_for(std::max(cap, padded))), dim3(kBlock), 0, s, in,
                     ptr<const int32_t>(table), ptr<const int32_t>(flags), ptr<const int64_t>(pos), cap, padded,
                     uniq, ptr<uint64_t>(prefix), ptr<int32_t>(item));
  {
    ProfScope ps("k_cat_sort", s);
    for (int64_t k = 2; k <= padded; k <<= 1)
      for (int64_t j = k >> 1; j > 0; j >>= 1)
        hipLaunchKernelGGL(k_bitonic_step, dim3(blocks_for(padded)), dim3(kBlock), 0, s, in,
                           ptr<const int32_t>(table), ptr<uint64_t>(prefix), ptr<int32_t>(item), padded, j, k);
  }
  Buf rank_of_slot = dev_alloc(sizeof(int32_t) * cap, s);
  if (uniq)
    hipLaunchKernelGGL(k_cat_ranks, dim3(blocks_for(uniq)), dim3(kBlock), 0, s, ptr<const int32_t>(item), uniq,
                       shift, ptr<int32_t>(rank_of_slot));
  cat->values = dev_alloc(sizeof(int32_t) * rows, s);
  {
    ProfScope ps("k_cat_values", s);
    hipLaunchKernelGGL(k_cat_values, dim3(blocks_for(rows)), dim3(kBlock), 0, s, ptr<const int32_t>(slot_of_row),
                       ptr<const int32_t>(rank_of_slot), rows, ptr<int32_t>(cat->values));
  }
  // keys column
  const int64_t nkeys = uniq + shift;
  auto keys = std::make_unique<cs_column>();
  keys->rows = nkeys;
  keys->null_count = shift;
  Buf lens = dev_alloc(sizeof(int32_t) * nkeys, s);
  hipLaunchKernelGGL(k_key_sizes, dim3(blocks_for(nkeys)), dim3(kBlock), 0, s, in, ptr<const int32_t>(table),
                     ptr<const int32_t>(item), nkeys, shift, ptr<int32_t>(lens));
  keys->offsets = dev_alloc(sizeof(int64_t) * (nkeys + 1), s);
  keys->nbytes = offsets_from_lengths(ptr<int32_t>(lens), nkeys, ptr<int64_t>(keys->offsets), s);
  keys->chars = dev_alloc((size_t)keys->nbytes, s);
  if (shift) keys->validity = validity_from_lengths(ptr<int32_t>(lens), nkeys, s);
  hipLaunchKernelGGL(k_key_copy, dim3(blocks_for(nkeys)), dim3(kBlock), 0, s, in, ptr<const int32_t>(table),
                     ptr<const int32_t>(item), nkeys, shift, keys->d_offsets(), ptr<uint8_t>(keys->chars));
  CS_HIP(hipStreamSynchronize(s));
  cat->keys = std::move(keys);
  return cat.release();
}

}  // namespace

extern "C" {

// NVCategory::create_from_strings -- NVCategory.cu:327-337 -> NVCategoryImpl_init :220-304
int cs_category_build(const cs_column* col, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    *out = build(col, S(stream));
  });
}

// NVCategory::create_from_categories -- NVCategory.cu:430-514
int cs_category_merge(const cs_category* const* cats, int ncats, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!out || ncats < 0 || (ncats > 0 && !cats)) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    hipStream_t s = S(stream);
    std::vector<const cs_column*> keysets;
    int64_t total_rows = 0;
    for (int i = 0; i < ncats; ++i) {
      if (!cats[i]) fail(CS_ERR_INVALID_ARG, "null category");
      keysets.push_back(cats[i]->keys.get());
      total_rows += cats[i]->rows;
    }
    std::unique_ptr<cs_column> all_keys(concat_columns(keysets, s));
    // category of the concatenated key sets: its keys are the merged key set and
    // its codes are, per input category, the old-code -> new-code table
    std::unique_ptr<cs_category> merged(build(all_keys.get(), s));
    auto res = std::make_unique<cs_category>();
    res->rows = total_rows;
    res->values = dev_alloc(sizeof(int32_t) * total_rows, s);
    int64_t key_base = 0, row_base = 0;
    for (int i = 0; i < ncats; ++i) {
      const cs_category* c = cats[i];
      if (c->rows)
        hipLaunchKernelGGL(k_remap, dim3(blocks_for(c->rows)), dim3(kBlock), 0, s, ptr<const int32_t>(c->values),
                           c->rows, ptr<const int32_t>(merged->values) + key_base,
                           ptr<int32_t>(res->values) + row_base);
      key_base += c->keys->rows;
      row_base += c->rows;
    }
    CS_HIP(hipStreamSynchronize(s));
    res->keys = std::move(merged->keys);
    *out = res.release();
  });
}

int cs_category_destroy(cs_category* cat) {
  return guard([&] { delete cat; });
}
int64_t cs_category_size(const cs_category* cat) { return cat ? cat->rows : 0; }
int64_t cs_category_keys_size(const cs_category* cat) { return cat && cat->keys ? cat->keys->rows : 0; }
int cs_category_keys(const cs_category* cat, cs_column** out) {
  return guard([&] {
    if (!cat || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    *out = new cs_column(*cat->keys);  // shares the immutable buffers
  });
}
const int32_t* cs_category_values_ptr(const cs_category* cat) { return cat ? ptr<const int32_t>(cat->values) : nullptr; }
int cs_category_get_values(const cs_category* cat, int32_t* out, int on_device, cs_stream stream) {
  return guard([&] {
    if (!cat || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    if (cat->rows == 0) return;
    hipStream_t s = S(stream);
    CS_HIP(hipMemcpyAsync(out, cat->values->p, sizeof(int32_t) * cat->rows,
                          on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
  });
}
int cs_remap_codes(const int32_t* codes, int64_t n, const int32_t* table, int32_t* out, cs_stream stream) {
  return guard([&] {
    if (n < 0 || (n > 0 && (!codes || !table || !out))) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    if (n) hipLaunchKernelGGL(k_remap, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), codes, n, table, out);
    CS_HIP(hipGetLastError());
  });
}

}  // extern "C"
