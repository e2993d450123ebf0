// NVCategory remap family (NVCategory.cu:926-1822): to_strings,
// gather_strings, gather, gather_and_remap, add_strings, remove_strings, merge_category,
// merge_and_remap, add / remove / set keys and remove_unused_keys (all "_and_remap").
//
// A category is (sorted unique keys column, int32 value per row).  Every member of the
// family is "compute a new key set, then send each value through an old-key -> new-key
// table".  The table comes from ONE category build over the concatenation new keys ++ old
// keys: equal strings get equal codes there, so a scatter of the new keys' codes followed
// by a gather with the old keys' codes matches the two sets (no pairwise string compares).
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "device_utils.h"

using namespace cs;
using namespace csdev;

namespace {

__global__ void k_fill(int32_t* __restrict__ a, int64_t n, int32_t v) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) a[i] = v;
}
__global__ void k_iota(int32_t* __restrict__ a, int64_t n, int32_t base) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) a[i] = base + (int32_t)i;
}
// where[code of new key j] = j
__global__ void k_scatter_index(const int32_t* __restrict__ codes, int64_t n, int32_t* __restrict__ where) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < n) where[codes[j]] = (int32_t)j;
}
__global__ void k_gather_index(const int32_t* __restrict__ codes, int64_t n, const int32_t* __restrict__ where, int32_t* __restrict__ table) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) table[i] = where[codes[i]];
}
// out[i] = v < 0 ? v : table[v]
__global__ void k_remap_values(const int32_t* __restrict__ values, int64_t n, const int32_t* __restrict__ table, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t v = values[i];
  out[i] = v < 0 ? v : table[v];
}
__global__ void k_mark_used(const int32_t* __restrict__ values, int64_t n, int64_t nkeys, int lo_ok, int32_t* __restrict__ used, unsigned* __restrict__ bad) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool oob = false;
  if (i < n) {
    const int32_t v = values[i];
    if (v >= 0 && v < nkeys) used[v] = 1;
    else oob = v < lo_ok || v >= nkeys;
  }
  if (__any(oob) && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}
__global__ void k_flag_not(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = in[i] ? 0 : 1;
}
// compaction: pos[slot[i]] = i for flagged i
__global__ void k_compact(const int32_t* __restrict__ flags, const int64_t* __restrict__ slot, int64_t n, int32_t* __restrict__ pos) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n && flags[i]) pos[slot[i]] = (int32_t)i;
}
// table[i] = flags[i] ? base + slot[i] : keep[i]
__global__ void k_table_from_slots(const int32_t* __restrict__ flags, const int64_t* __restrict__ slot, int64_t n, int32_t base, const int32_t* __restrict__ keep,
                                   int32_t* __restrict__ table) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) table[i] = flags[i] ? base + (int32_t)slot[i] : (keep ? keep[i] : -1);
}
__global__ void k_flag_negative(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = in[i] < 0 ? 1 : 0;
}

struct Compacted {
  Buf pos;  // indices of the flagged entries, ascending
  Buf slot; // exclusive scan of the flags
  int64_t n = 0;
};
Compacted compact(const int32_t* d_flags, int64_t n, hipStream_t s) {
  Compacted c;
  if (n == 0) return c;
  c.slot = dev_alloc(sizeof(int64_t) * (n + 1), s);
  c.n = offsets_from_lengths(d_flags, n, ptr<int64_t>(c.slot), s);
  c.pos = dev_alloc(sizeof(int32_t) * std::max<int64_t>(c.n, 1), s);
  if (c.n) hipLaunchKernelGGL(k_compact, dim3(blocks_for(n)), dim3(kBlock), 0, s, d_flags, ptr<const int64_t>(c.slot), n, ptr<int32_t>(c.pos));
  return c;
}
unsigned read_flag(const Buf& b, hipStream_t s) {
  unsigned* h = (unsigned*)pinned_scratch(sizeof(unsigned));
  CS_HIP(hipMemcpyAsync(h, b->p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return *h;
}
Buf zeros32(int64_t n, hipStream_t s) {
  Buf b = dev_alloc(sizeof(int32_t) * std::max<int64_t>(n, 1), s);
  CS_HIP(hipMemsetAsync(b->p, 0, sizeof(int32_t) * std::max<int64_t>(n, 1), s));
  return b;
}

// table[i] = index of old key i in `neu`, or -1 (both key columns hold distinct rows)
Buf match_table(const cs_column* old_keys, const cs_column* neu, hipStream_t s) {
  const int64_t ko = old_keys->rows, kn = neu->rows;
  Buf table = dev_alloc(sizeof(int32_t) * std::max<int64_t>(ko, 1), s);
  if (ko == 0) return table;
  if (kn == 0) {
    hipLaunchKernelGGL(k_fill, dim3(blocks_for(ko)), dim3(kBlock), 0, s, ptr<int32_t>(table), ko, -1);
    return table;
  }
  std::vector<const cs_column*> both{neu, old_keys};
  std::unique_ptr<cs_column> all(concat_columns(both, s));
  std::unique_ptr<cs_category> u(category_build(all.get(), s));
  const int64_t nu = u->keys->rows;
  Buf where = dev_alloc(sizeof(int32_t) * nu, s);
  hipLaunchKernelGGL(k_fill, dim3(blocks_for(nu)), dim3(kBlock), 0, s, ptr<int32_t>(where), nu, -1);
  hipLaunchKernelGGL(k_scatter_index, dim3(blocks_for(kn)), dim3(kBlock), 0, s, ptr<const int32_t>(u->values), kn, ptr<int32_t>(where));
  hipLaunchKernelGGL(k_gather_index, dim3(blocks_for(ko)), dim3(kBlock), 0, s, ptr<const int32_t>(u->values) + kn, ko, ptr<const int32_t>(where),
                     ptr<int32_t>(table));
  CS_HIP(hipStreamSynchronize(s));
  return table;
}
// a category with the given keys and the old values sent through `table` (nullptr: values unchanged)
cs_category* remapped(const cs_category* cat, std::unique_ptr<cs_column> keys, const Buf& table, hipStream_t s) {
  auto out = std::make_unique<cs_category>();
  out->rows = cat->rows;
  out->keys = std::move(keys);
  out->values = dev_alloc(sizeof(int32_t) * std::max<int64_t>(cat->rows, 1), s);
  if (cat->rows) {
    if (table)
      hipLaunchKernelGGL(k_remap_values, dim3(blocks_for(cat->rows)), dim3(kBlock), 0, s, ptr<const int32_t>(cat->values), cat->rows, ptr<const int32_t>(table),
                         ptr<int32_t>(out->values));
    else
      CS_HIP(hipMemcpyAsync(out->values->p, cat->values->p, sizeof(int32_t) * cat->rows, hipMemcpyDeviceToDevice, s));
  }
  CS_HIP(hipStreamSynchronize(s));
  return out.release();
}
std::unique_ptr<cs_column> copy_handle(const cs_column* c) { return std::unique_ptr<cs_column>(new cs_column(*c)); }
std::unique_ptr<cs_column> unique_sorted(const cs_column* strs, hipStream_t s) {
  std::unique_ptr<cs_category> c(category_build(strs, s));
  return std::move(c->keys);
}
cs_category* copy_category(const cs_category* cat, hipStream_t s) { return remapped(cat, copy_handle(cat->keys.get()), nullptr, s); }

template <class T>
struct DevIn {
  Buf tmp;
  const T* d = nullptr;
  DevIn(const T* p, int64_t n, int on_device, hipStream_t s) {
    if (on_device || !p || n == 0) {
      d = p;
      return;
    }
    tmp = dev_alloc(sizeof(T) * (size_t)n, s);
    CS_HIP(hipMemcpyAsync(tmp->p, p, sizeof(T) * (size_t)n, hipMemcpyHostToDevice, s));
    d = ptr<const T>(tmp);
  }
};

}  // namespace

extern "C" {

// NVCategory::to_strings (NVCategory.cu:977-1009): the row strings back, value < 0 or a null key -> null row
int cs_category_to_strings(const cs_category* cat, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!cat || !out) fail(CS_ERR_INVALID_ARG, "to_strings: bad arguments");
    require_device();
    *out = nullptr;
    if (cat->rows == 0) return;  // the reference returns no instance
    *out = gather_rows(cat->keys.get(), ptr<const int32_t>(cat->values), cat->rows, S(stream), true);
  });
}
// NVCategory::gather_strings (NVCategory.cu:1011-1082): keys[pos[i]]; a position outside [0, keys) -> CS_ERR_RANGE
int cs_category_gather_strings(const cs_category* cat, const int32_t* pos, int64_t n, int on_device, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!cat || !out || n < 0 || (n > 0 && !pos)) fail(CS_ERR_INVALID_ARG, "gather_strings: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    DevIn<int32_t> p(pos, n, on_device, s);
    if (n && cat->keys->rows == 0) fail(CS_ERR_RANGE, "gather_strings: position out of range");
    *out = gather_rows(cat->keys.get(), p.d, n, s, false);
  });
}
// NVCategory::gather (NVCategory.cu:1142-1170): same keys, the given positions as values (-1 allowed)
int cs_category_gather(const cs_category* cat, const int32_t* pos, int64_t n, int on_device, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !out || n < 0 || (n > 0 && !pos)) fail(CS_ERR_INVALID_ARG, "gather: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    DevIn<int32_t> p(pos, n, on_device, s);
    const int64_t nk = cat->keys->rows;
    if (n) {
      Buf used = zeros32(nk, s), bad = zeros32(1, s);
      hipLaunchKernelGGL(k_mark_used, dim3(blocks_for(n)), dim3(kBlock), 0, s, p.d, n, nk, -1, ptr<int32_t>(used), ptr<unsigned>(bad));
      if (read_flag(bad, s)) fail(CS_ERR_RANGE, "gather: position out of range");
    }
    auto res = std::make_unique<cs_category>();
    res->rows = n;
    res->keys = copy_handle(cat->keys.get());
    res->values = dev_alloc(sizeof(int32_t) * std::max<int64_t>(n, 1), s);
    if (n) CS_HIP(hipMemcpyAsync(res->values->p, p.d, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, s));
    CS_HIP(hipStreamSynchronize(s));
    *out = res.release();
  });
}
// NVCategory::gather_and_remap (NVCategory.cu:1084-1140): only the keys the positions name, values renumbered
int cs_category_gather_and_remap(const cs_category* cat, const int32_t* pos, int64_t n, int on_device, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !out || n < 0 || (n > 0 && !pos)) fail(CS_ERR_INVALID_ARG, "gather_and_remap: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    DevIn<int32_t> p(pos, n, on_device, s);
    const int64_t nk = cat->keys->rows;
    Buf used = zeros32(nk, s), bad = zeros32(1, s);
    if (n) hipLaunchKernelGGL(k_mark_used, dim3(blocks_for(n)), dim3(kBlock), 0, s, p.d, n, nk, 0, ptr<int32_t>(used), ptr<unsigned>(bad));
    if (n && read_flag(bad, s)) fail(CS_ERR_RANGE, "gather_and_remap: position out of range");
    Compacted c = compact(ptr<const int32_t>(used), nk, s);
    Buf table = dev_alloc(sizeof(int32_t) * std::max<int64_t>(nk, 1), s);
    if (nk) hipLaunchKernelGGL(k_table_from_slots, dim3(blocks_for(nk)), dim3(kBlock), 0, s, ptr<const int32_t>(used), ptr<const int64_t>(c.slot), nk, 0,
                               (const int32_t*)nullptr, ptr<int32_t>(table));
    auto res = std::make_unique<cs_category>();
    res->rows = n;
    res->keys.reset(c.n ? gather_rows(cat->keys.get(), ptr<const int32_t>(c.pos), c.n, s) : make_all_null(0, s));
    res->values = dev_alloc(sizeof(int32_t) * std::max<int64_t>(n, 1), s);
    if (n) hipLaunchKernelGGL(k_remap_values, dim3(blocks_for(n)), dim3(kBlock), 0, s, p.d, n, ptr<const int32_t>(table), ptr<int32_t>(res->values));
    CS_HIP(hipStreamSynchronize(s));
    *out = res.release();
  });
}
// NVCategory::add_strings (NVCategory.cu:926-940): the category of (this category's rows ++ strs)
int cs_category_add_strings(const cs_category* cat, const cs_column* strs, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !strs || !out) fail(CS_ERR_INVALID_ARG, "add_strings: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    std::unique_ptr<cs_column> mine(cat->rows ? gather_rows(cat->keys.get(), ptr<const int32_t>(cat->values), cat->rows, s, true) : make_all_null(0, s));
    std::vector<const cs_column*> both{mine.get(), strs};
    std::unique_ptr<cs_column> all(concat_columns(both, s));
    *out = category_build(all.get(), s);
  });
}
// NVCategory::remove_strings (NVCategory.cu:942-975): the category of this category's rows without those equal to a row of strs
int cs_category_remove_strings(const cs_category* cat, const cs_column* strs, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !strs || !out) fail(CS_ERR_INVALID_ARG, "remove_strings: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    // keys hit by strs -> rows whose value names such a key go away
    std::unique_ptr<cs_column> del = unique_sorted(strs, s);
    Buf table = match_table(cat->keys.get(), del.get(), s);  // >= 0: key is to be removed
    const int64_t n = cat->rows;
    Buf row_key_hit = dev_alloc(sizeof(int32_t) * std::max<int64_t>(n, 1), s);
    if (n) hipLaunchKernelGGL(k_remap_values, dim3(blocks_for(n)), dim3(kBlock), 0, s, ptr<const int32_t>(cat->values), n, ptr<const int32_t>(table),
                              ptr<int32_t>(row_key_hit));
    Buf keep = dev_alloc(sizeof(int32_t) * std::max<int64_t>(n, 1), s);
    if (n) hipLaunchKernelGGL(k_flag_negative, dim3(blocks_for(n)), dim3(kBlock), 0, s, ptr<const int32_t>(row_key_hit), n, ptr<int32_t>(keep));
    Compacted c = compact(ptr<const int32_t>(keep), n, s);
    std::unique_ptr<cs_column> mine(n ? gather_rows(cat->keys.get(), ptr<const int32_t>(cat->values), n, s, true) : make_all_null(0, s));
    std::unique_ptr<cs_column> left(c.n ? gather_rows(mine.get(), ptr<const int32_t>(c.pos), c.n, s) : make_all_null(0, s));
    *out = category_build(left.get(), s);
  });
}
// NVCategory::merge_category (NVCategory.cu:1223-1337): keys = this category's keys followed by cat2's keys that are new
// (in sorted order), values = this category's values followed by cat2's, renumbered
int cs_category_merge_category(const cs_category* cat, const cs_category* cat2, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !cat2 || !out) fail(CS_ERR_INVALID_ARG, "merge_category: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t k1 = cat->keys->rows, k2 = cat2->keys->rows;
    if (k1 == 0 || k2 == 0) {  // NVCategory.cu:1231-1239: a copy of the one that has keys (or an empty category)
      *out = copy_category(k1 == 0 ? cat2 : cat, s);
      return;
    }
    Buf in1 = match_table(cat2->keys.get(), cat->keys.get(), s);  // index in keys1 of each key of cat2, or -1
    Buf is_new = dev_alloc(sizeof(int32_t) * k2, s);
    hipLaunchKernelGGL(k_flag_negative, dim3(blocks_for(k2)), dim3(kBlock), 0, s, ptr<const int32_t>(in1), k2, ptr<int32_t>(is_new));
    Compacted c = compact(ptr<const int32_t>(is_new), k2, s);
    Buf table2 = dev_alloc(sizeof(int32_t) * k2, s);
    hipLaunchKernelGGL(k_table_from_slots, dim3(blocks_for(k2)), dim3(kBlock), 0, s, ptr<const int32_t>(is_new), ptr<const int64_t>(c.slot), k2, (int32_t)k1,
                       ptr<const int32_t>(in1), ptr<int32_t>(table2));
    auto res = std::make_unique<cs_category>();
    if (c.n) {
      std::unique_ptr<cs_column> fresh(gather_rows(cat2->keys.get(), ptr<const int32_t>(c.pos), c.n, s));
      std::vector<const cs_column*> both{cat->keys.get(), fresh.get()};
      res->keys.reset(concat_columns(both, s));
    } else {
      res->keys = copy_handle(cat->keys.get());
    }
    res->rows = cat->rows + cat2->rows;
    res->values = dev_alloc(sizeof(int32_t) * std::max<int64_t>(res->rows, 1), s);
    if (cat->rows) CS_HIP(hipMemcpyAsync(res->values->p, cat->values->p, sizeof(int32_t) * cat->rows, hipMemcpyDeviceToDevice, s));
    if (cat2->rows)
      hipLaunchKernelGGL(k_remap_values, dim3(blocks_for(cat2->rows)), dim3(kBlock), 0, s, ptr<const int32_t>(cat2->values), cat2->rows,
                         ptr<const int32_t>(table2), ptr<int32_t>(res->values) + cat->rows);
    CS_HIP(hipStreamSynchronize(s));
    *out = res.release();
  });
}
// NVCategory::add_keys_and_remap (NVCategory.cu:1375-1480)
int cs_category_add_keys(const cs_category* cat, const cs_column* strs, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !strs || !out) fail(CS_ERR_INVALID_ARG, "add_keys: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    if (strs->rows == 0) {
      *out = copy_category(cat, s);
      return;
    }
    if (cat->keys->rows == 0) {  // :1391-1403: the new keys, the values as they are
      *out = remapped(cat, unique_sorted(strs, s), nullptr, s);
      return;
    }
    std::vector<const cs_column*> both{cat->keys.get(), strs};
    std::unique_ptr<cs_column> all(concat_columns(both, s));
    std::unique_ptr<cs_category> u(category_build(all.get(), s));
    // the first keys->rows codes of the union ARE the old-key -> new-key table
    Buf table = dev_alloc(sizeof(int32_t) * cat->keys->rows, s);
    CS_HIP(hipMemcpyAsync(table->p, u->values->p, sizeof(int32_t) * cat->keys->rows, hipMemcpyDeviceToDevice, s));
    *out = remapped(cat, std::move(u->keys), table, s);
  });
}
// NVCategory::remove_keys_and_remap (NVCategory.cu:1482-1565): values of removed keys become -1
int cs_category_remove_keys(const cs_category* cat, const cs_column* strs, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !strs || !out) fail(CS_ERR_INVALID_ARG, "remove_keys: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t nk = cat->keys->rows;
    if (nk == 0 || strs->rows == 0) {
      *out = copy_category(cat, s);
      return;
    }
    std::unique_ptr<cs_column> del = unique_sorted(strs, s);
    Buf hit = match_table(cat->keys.get(), del.get(), s);  // >= 0: removed
    Buf keep = dev_alloc(sizeof(int32_t) * nk, s);
    hipLaunchKernelGGL(k_flag_negative, dim3(blocks_for(nk)), dim3(kBlock), 0, s, ptr<const int32_t>(hit), nk, ptr<int32_t>(keep));
    Compacted c = compact(ptr<const int32_t>(keep), nk, s);
    Buf table = dev_alloc(sizeof(int32_t) * nk, s);
    hipLaunchKernelGGL(k_table_from_slots, dim3(blocks_for(nk)), dim3(kBlock), 0, s, ptr<const int32_t>(keep), ptr<const int64_t>(c.slot), nk, 0,
                       (const int32_t*)nullptr, ptr<int32_t>(table));
    std::unique_ptr<cs_column> keys(c.n ? gather_rows(cat->keys.get(), ptr<const int32_t>(c.pos), c.n, s) : make_all_null(0, s));
    *out = remapped(cat, std::move(keys), table, s);
  });
}
// NVCategory::remove_unused_keys_and_remap (NVCategory.cu:1567-1706)
int cs_category_remove_unused_keys(const cs_category* cat, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !out) fail(CS_ERR_INVALID_ARG, "remove_unused_keys: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t nk = cat->keys->rows;
    if (nk == 0) {
      *out = copy_category(cat, s);
      return;
    }
    Buf used = zeros32(nk, s), bad = zeros32(1, s);
    if (cat->rows)
      hipLaunchKernelGGL(k_mark_used, dim3(blocks_for(cat->rows)), dim3(kBlock), 0, s, ptr<const int32_t>(cat->values), cat->rows, nk, INT32_MIN, ptr<int32_t>(used),
                         ptr<unsigned>(bad));
    Compacted c = compact(ptr<const int32_t>(used), nk, s);
    if (c.n == nk) {
      *out = copy_category(cat, s);
      return;
    }
    Buf table = dev_alloc(sizeof(int32_t) * nk, s);
    hipLaunchKernelGGL(k_table_from_slots, dim3(blocks_for(nk)), dim3(kBlock), 0, s, ptr<const int32_t>(used), ptr<const int64_t>(c.slot), nk, 0,
                       (const int32_t*)nullptr, ptr<int32_t>(table));
    std::unique_ptr<cs_column> keys(c.n ? gather_rows(cat->keys.get(), ptr<const int32_t>(c.pos), c.n, s) : make_all_null(0, s));
    *out = remapped(cat, std::move(keys), table, s);
  });
}
// NVCategory::set_keys_and_remap (NVCategory.cu:1708-1822): the keys become sorted-unique(strs); values of keys that are gone become -1
int cs_category_set_keys(const cs_category* cat, const cs_column* strs, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!cat || !strs || !out) fail(CS_ERR_INVALID_ARG, "set_keys: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t nk = cat->keys->rows;
    if (strs->rows == 0) {  // :1715-1719: no keys, every value -1
      auto res = std::make_unique<cs_category>();
      res->rows = nk == 0 ? 0 : cat->rows;
      res->keys.reset(make_all_null(0, s));
      res->values = dev_alloc(sizeof(int32_t) * std::max<int64_t>(res->rows, 1), s);
      if (res->rows) hipLaunchKernelGGL(k_fill, dim3(blocks_for(res->rows)), dim3(kBlock), 0, s, ptr<int32_t>(res->values), res->rows, -1);
      CS_HIP(hipStreamSynchronize(s));
      *out = res.release();
      return;
    }
    std::unique_ptr<cs_column> neu = unique_sorted(strs, s);
    if (nk == 0) {
      *out = remapped(cat, std::move(neu), nullptr, s);
      return;
    }
    Buf table = match_table(cat->keys.get(), neu.get(), s);
    *out = remapped(cat, std::move(neu), table, s);
  });
}

}  // extern "C"
