// Category (dictionary) build: sorted unique keys + int32 code per row.
//
// Contract (NVCategory.cu:220-304): keys are the distinct rows in ascending
// order -- null first, then unsigned bytewise order with "shorter is less"
// (custring.inl:240-261) -- and values[r] is the index of row r's key.  The
// reference gets there by comparator-sorting every row; the outputs depend only
// on the sorted set of distinct keys and each row's rank in it, so here rows are
// first de-duplicated through an open-addressing hash table in HBM (one CAS per
// new key, full byte compare on every hit, so there are no false merges), and only
// the U distinct representatives are sorted (bitonic network on an 8-byte
// big-endian prefix with a full-compare tie break).
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "device_utils.h"

using namespace cs;
using namespace csdev;


namespace {

// A table entry names a key by where its bytes are: (byte offset into chars) << 24 | length.
// The probe then goes table -> key bytes, without a detour through the offsets array.
typedef unsigned long long Entry;
constexpr Entry kEmpty = ~0ull;
constexpr int kMaxKeyLen = (1 << 24) - 2;
__device__ __forceinline__ Entry make_entry(int64_t off, int n) { return ((Entry)off << 24) | (Entry)(unsigned)n; }
__device__ __forceinline__ int64_t entry_off(Entry e) { return (int64_t)(e >> 24); }
__device__ __forceinline__ int entry_len(Entry e) { return (int)(e & 0xFFFFFFu); }

// Bytes [i, i + 32) of the row at `p` (n bytes) as eight little-endian dwords, zero beyond the
// row's end.  Read as the 16-byte-aligned pieces that cover them (three loads at most) and
// shifted into place: the memory pipe sees a few wide requests per row instead of one per
// byte, which is what bounds this kernel (every lane of a probe touches a different line).
// Reads stay inside the aligned pieces that hold at least one byte of the row.
__device__ __forceinline__ void row_block32(const uint8_t* p, int n, int i, uint32_t w[8]) {
  const int rem = n - i;  // > 0
  const uint8_t* q = p + i;
  const int sh = (int)((uintptr_t)q & 15);
  const uint4* a = reinterpret_cast<const uint4*>(q - sh);
  const int last = (sh + (rem < 32 ? rem : 32) - 1) >> 4;  // index of the last aligned piece in use: 0..2
  // (wave-uniform branches; a lane that does not need the piece re-reads one it does need)
  const uint4 x = a[0];
  uint4 y = x, z = x;
  if (__any(last >= 1)) y = a[last >= 1 ? 1 : 0];
  if (__any(last >= 2)) z = a[last >= 2 ? 2 : 0];
  // shift right by sh bytes, one binary digit of sh at a time (selects, then a funnel shift).  Written
  // out on scalars: with arrays indexed in an unrolled loop the optimizer turns the selects into a
  // dynamically indexed private array, i.e. a round trip through scratch memory.
  const bool by8 = (sh & 8) != 0, by4 = (sh & 4) != 0;
  const int bits = (sh & 3) * 8;
  const uint32_t u0 = by8 ? x.z : x.x;
  const uint32_t u1 = by8 ? x.w : x.y;
  const uint32_t u2 = by8 ? y.x : x.z;
  const uint32_t u3 = by8 ? y.y : x.w;
  const uint32_t u4 = by8 ? y.z : y.x;
  const uint32_t u5 = by8 ? y.w : y.y;
  const uint32_t u6 = by8 ? z.x : y.z;
  const uint32_t u7 = by8 ? z.y : y.w;
  const uint32_t u8 = by8 ? z.z : z.x;
  const uint32_t u9 = by8 ? z.w : z.y;
  const uint32_t t0 = by4 ? u1 : u0;
  const uint32_t t1 = by4 ? u2 : u1;
  const uint32_t t2 = by4 ? u3 : u2;
  const uint32_t t3 = by4 ? u4 : u3;
  const uint32_t t4 = by4 ? u5 : u4;
  const uint32_t t5 = by4 ? u6 : u5;
  const uint32_t t6 = by4 ? u7 : u6;
  const uint32_t t7 = by4 ? u8 : u7;
  const uint32_t t8 = by4 ? u9 : u8;
  auto keep = [&](int k, uint32_t d) {  // zero beyond the row's end
    const int left = rem - 4 * k;
    return d & (left >= 4 ? 0xFFFFFFFFu : (left <= 0 ? 0u : (1u << (8 * left)) - 1u));
  };
  w[0] = keep(0, __funnelshift_r(t0, t1, bits));
  w[1] = keep(1, __funnelshift_r(t1, t2, bits));
  w[2] = keep(2, __funnelshift_r(t2, t3, bits));
  w[3] = keep(3, __funnelshift_r(t3, t4, bits));
  w[4] = keep(4, __funnelshift_r(t4, t5, bits));
  w[5] = keep(5, __funnelshift_r(t5, t6, bits));
  w[6] = keep(6, __funnelshift_r(t6, t7, bits));
  w[7] = keep(7, __funnelshift_r(t7, t8, bits));
}
// Any well-mixed hash will do: keys and codes depend only on the sorted key set.
__device__ __forceinline__ uint32_t hash_mix(uint32_t h, uint32_t w) {
  w *= 0xCC9E2D51u;
  w = (w << 15) | (w >> 17);
  h ^= w * 0x1B873593u;
  return ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
}
__device__ __forceinline__ uint32_t hash_final(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  return h ^ (h >> 16);
}
// are the n bytes at a and b equal?  (rows longer than the 32 bytes the caller holds in registers)
__device__ __forceinline__ bool same_tail(const uint8_t* a, const uint8_t* b, int n) {
  for (int i = 32; i < n; i += 32) {
    uint32_t x[8], y[8];
    row_block32(a, n, i, x);
    row_block32(b, n, i, y);
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d |= x[k] ^ y[k];
    if (d) return false;
  }
  return true;
}
// custr::compare: unsigned bytewise, shorter is less
__device__ __forceinline__ int compare_keys(const uint8_t* chars, Entry ea, Entry eb) {
  const uint8_t *pa = chars + entry_off(ea), *pb = chars + entry_off(eb);
  const int la = entry_len(ea), lb = entry_len(eb);
  int m = la < lb ? la : lb;
  for (int i = 0; i < m; ++i)
    if (pa[i] != pb[i]) return (int)pa[i] - (int)pb[i];
  return la - lb;
}

constexpr int kProbeLimit = 128;  // longer probe runs mean the table is too small: the host retries with a bigger one
// Slots with the key inside (`sw` == 4: 32-byte slots; tables of up to 2^24 slots).  A probe that meets another row's slot used
// to make TWO dependent trips -- the slot, then the representative row's bytes somewhere in gigabytes of chars (a cache and TLB
// miss each: 1.0 of the kernel's 2.6 ms at K = 1M) -- although most keys are short.  Now the row that claims a slot also
// leaves the key's first 21 bytes in the slot's other three words, seven bytes a word under a tag byte, and a probe compares
// in registers: one trip.  Each word is written once, by one 8-byte store, from "all ones" to its final value, so a reader
// sees a word either absent or final -- never torn, whatever the order the stores become visible in; a key whose words are
// not (yet) all there, or one beyond 21 bytes, is compared through its bytes in chars as before.
constexpr int kSlotKeyBytes = 21;
constexpr unsigned long long kSlotTag = 0x01ull << 56;
__device__ __forceinline__ bool slot_word_there(unsigned long long w) { return (w >> 56) == 0x01ull; }
// the key's bytes 7 j .. 7 j + 6 (zero beyond its end: row_block32 pads) under the tag
__device__ __forceinline__ unsigned long long slot_word(const uint32_t w[8], int j) {
  // bytes [7j, 7j+7) of the little-endian dword array
  const int b0 = 7 * j, d = b0 >> 2, sh = (b0 & 3) * 8;
  const unsigned long long lo = (unsigned long long)w[d] | ((unsigned long long)w[d + 1] << 32);
  unsigned long long v = lo >> sh;
  if (sh > 8) v |= (unsigned long long)w[d + 2] << (64 - sh);  // (the seven bytes reach into a third dword)
  return (v & 0x00FFFFFFFFFFFFFFull) | kSlotTag;
}
// `dbg` (CS_CAT_DEBUG, measurement only -- wrong results): 1 skips the byte compare, 2 the table.
// (eight waves a SIMD: the kernel is three dependent trips a row -- offsets, key bytes, slot -- and what bounds it is how many of
// those chains the resident threads keep in flight: 2.56 -> 2.31 ms at K = 1M against six waves, eleven registers spilled; two
// rows a thread, phase by phase, at six waves: 2.90 -- the registers of the second row cost more than its trips overlap)
__global__ void __launch_bounds__(256, 8) k_cat_insert(ColView in, Entry* table, uint32_t mask,
                                                    int32_t* __restrict__ slot_of_row, int* __restrict__ has_null,
                                                    int* __restrict__ overflow, int probe_limit, int dbg, int sw, int64_t stride = 1, int64_t count = -1) {
  // (`stride` / `count`: the sampling launch takes rows 0, stride, 2 stride, ... -- `count` of them; the slot ids it writes
  // for those rows are overwritten by the full pass that follows)
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= (count >= 0 ? count : in.rows)) return;
  const int64_t r = idx * stride;
  if (r >= in.rows) return;
  // a table that turned out too small: the launch is lost, the workgroups that have not begun yet leave at once
  if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) return;
  if (!row_is_valid(in.validity, r)) {
    slot_of_row[r] = -1;
    *has_null = 1;
    return;
  }
  const int64_t b = in.offsets[r];
  const int n = (int)min(in.offsets[r + 1] - b, (int64_t)kMaxKeyLen + 1);
  if (n > kMaxKeyLen) {
    atomicOr(overflow, 2);  // a key the table entry cannot name
    return;
  }
  const uint8_t* p = in.chars + b;
  // the first 32 bytes stay in registers for the probe's compare; longer rows hash the rest from memory
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = 0;
  if (n > 0) row_block32(p, n, 0, w);
  uint32_t h = 0x9E3779B9u ^ (uint32_t)n;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (4 * k < n) h = hash_mix(h, w[k]);
  for (int i = 32; i < n; i += 32) {
    uint32_t x[8];
    row_block32(p, n, i, x);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i + 4 * k < n) h = hash_mix(h, x[k]);
  }
  uint32_t slot = hash_final(h) & mask;
  const Entry mine = make_entry(b, n);
  const bool in_slot = sw == 4 && n <= kSlotKeyBytes;
  unsigned long long k0 = 0, k1 = 0, k2 = 0;
  if (in_slot) {
    k0 = slot_word(w, 0);
    k1 = slot_word(w, 1);
    k2 = slot_word(w, 2);
  }
  int probes = 0;
  if (dbg & 2) {  // measurement only: no table
    slot_of_row[r] = (int32_t)slot;
    return;
  }
  for (;;) {
    // Look before the CAS: a slot only ever changes from empty to its final key, so a non-empty
    // value read here is final and the common case (key already present) needs no atomic at
    // all -- with a skewed key distribution the CASes of a hot key would otherwise queue on one
    // address (about 10 ns each: 100 ms for a key that 7 % of 125M rows share).
    Entry* sp = table + (size_t)slot * (size_t)sw;
    Entry cur;
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    if (sw == 4) {
      // the slot's 32 bytes in two 16-byte loads that go to the L2 (sc1: what a relaxed agent-scope atomic load is on gfx950 --
      // a slot cached as empty in a CU's L1 would send every later row of a hot key to the CAS); every 8-byte word of the
      // slot is written once, whole: wider loads see each word absent or final
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      u32x4_t lo, hi;
      asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(lo), "=&v"(hi)
                   : "v"(sp)
                   : "memory");
      cur = (Entry)lo.x | ((Entry)lo.y << 32);
      s0 = (Entry)lo.z | ((Entry)lo.w << 32);
      s1 = (Entry)hi.x | ((Entry)hi.y << 32);
      s2 = (Entry)hi.z | ((Entry)hi.w << 32);
    } else {
      cur = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (cur == kEmpty) {
      cur = atomicCAS(sp, kEmpty, mine);
      if (cur == kEmpty) {  // this row represents the key: leave it in the slot
        if (in_slot) {
          __hip_atomic_store(sp + 1, k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(sp + 2, k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(sp + 3, k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
      }
      s0 = s1 = s2 = 0;  // (somebody else's slot after all: its key words were read before it was claimed)
    }
    if (cur == mine) break;
    if (entry_len(cur) == n) {
      bool same = true;
      if (in_slot && slot_word_there(s0) && slot_word_there(s1) && slot_word_there(s2) && !(dbg & 1)) {
        same = s0 == k0 && s1 == k1 && s2 == k2;
      } else if (n > 0 && !(dbg & 1)) {
        const uint8_t* q = in.chars + entry_off(cur);
        uint32_t y[8];
        row_block32(q, n, 0, y);
        uint32_t d = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) d |= w[k] ^ y[k];
        same = d == 0 && (n <= 32 || same_tail(p, q, n));
      }
      if (same) break;
    }
    slot = (slot + 1) & mask;
    if (++probes > probe_limit) {
      atomicOr(overflow, 1);
      return;
    }
  }
  slot_of_row[r] = (int32_t)slot;
}
// (slots of `sw` words: the later passes read the slot words from `dense`, one per slot)
__global__ void k_cat_flags(const Entry* __restrict__ table, int64_t cap, int sw, int32_t* __restrict__ flags, Entry* __restrict__ dense) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  const Entry e = table[(size_t)i * (size_t)sw];
  flags[i] = e != kEmpty;
  if (dense) dense[i] = e;
}
// sort records: prefix[i] = first 8 key bytes big-endian, item[i] = slot id;
// padding up to the power of two sorts last
__global__ void k_cat_records(ColView in, const Entry* __restrict__ table, const int32_t* __restrict__ flags,
                              const int64_t* __restrict__ pos, int64_t cap, int64_t padded, int64_t uniq,
                              uint64_t* __restrict__ prefix, int32_t* __restrict__ item) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < cap && flags[i]) {
    const Entry e = table[i];
    int64_t b = entry_off(e);
    int n = entry_len(e);
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
__global__ void k_bitonic_step(ColView in, const Entry* __restrict__ table, uint64_t* __restrict__ prefix,
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
  else cmp = compare_keys(in.chars, table[ia], table[ib]);
  bool ascending = (i & k) == 0;
  if ((cmp > 0) == ascending && cmp != 0) {
    prefix[i] = pb;
    prefix[l] = pa;
    item[i] = ib;
    item[l] = ia;
  }
}
// The steps of the network whose partners lie inside one chunk of kSortChunk records, run out
// of LDS in a single launch: every j < kSortChunk of the stages k_first..k_last (the whole
// network for the first kSortChunk-sized stages, the tail of each later stage).  Directions
// depend on the global index, so chunks sort independently.
constexpr int kSortChunk = 2048;
__global__ void __launch_bounds__(256) k_bitonic_local(ColView in, const Entry* __restrict__ table,
                                                       uint64_t* __restrict__ prefix, int32_t* __restrict__ item,
                                                       int64_t padded, int64_t k_first, int64_t k_last) {
  __shared__ uint64_t s_prefix[kSortChunk];
  __shared__ int32_t s_item[kSortChunk];
  const int n = (int)(padded < kSortChunk ? padded : kSortChunk);
  const int64_t base = (int64_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += 256) {
    s_prefix[i] = prefix[base + i];
    s_item[i] = item[base + i];
  }
  __syncthreads();
  for (int64_t k = k_first; k <= k_last; k <<= 1) {
    for (int j = (int)(k / 2 < n / 2 ? k / 2 : n / 2); j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < n / 2; t += 256) {
        const int i = ((t / j) * 2 * j) + (t % j), l = i + j;
        const uint64_t pa = s_prefix[i], pb = s_prefix[l];
        const int32_t ia = s_item[i], ib = s_item[l];
        int cmp;
        if (ia < 0 || ib < 0) cmp = (ia < 0) - (ib < 0);  // padding is greatest
        else if (pa != pb) cmp = pa < pb ? -1 : 1;
        else cmp = compare_keys(in.chars, table[ia], table[ib]);
        const bool ascending = ((base + i) & k) == 0;
        if ((cmp > 0) == ascending && cmp != 0) {
          s_prefix[i] = pb;
          s_prefix[l] = pa;
          s_item[i] = ib;
          s_item[l] = ia;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    prefix[base + i] = s_prefix[i];
    item[base + i] = s_item[i];
  }
}
// ---- distinct keys in order: radix sort on the 8-byte prefix, full compares only among equal prefixes ----------
// flags[i] = 1 when record i shares its prefix with a neighbour of the prefix-sorted sequence (such records keep their
// SET of positions; only the order among them is open)
__global__ void k_tie_flags(const uint64_t* __restrict__ prefix, int64_t n, int32_t* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint64_t p = prefix[i];
  flags[i] = ((i > 0 && prefix[i - 1] == p) || (i + 1 < n && prefix[i + 1] == p)) ? 1 : 0;
}
// the tied records, in sequence order, into their own (padded) arrays; where[j] = position of the j-th of them
__global__ void k_tie_gather(const uint64_t* __restrict__ prefix, const int32_t* __restrict__ item, const int32_t* __restrict__ flags,
                             const int64_t* __restrict__ pos, int64_t n, int64_t tied, int64_t padded, uint64_t* __restrict__ tprefix,
                             int32_t* __restrict__ titem, int32_t* __restrict__ where) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n && flags[i]) {
    const int64_t j = pos[i];
    tprefix[j] = prefix[i];
    titem[j] = item[i];
    where[j] = (int32_t)i;
  }
  if (i >= tied && i < padded) {
    tprefix[i] = ~0ull;
    titem[i] = -1;
  }
}
__global__ void k_tie_scatter(const int32_t* __restrict__ titem, const int32_t* __restrict__ where, int64_t tied, int32_t* __restrict__ item) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < tied) item[where[j]] = titem[j];
}
// ---- the order among tied records: rounds of radix sorts on the keys' next seven bytes --------------------------------
// The compare network over ALL records that share an 8-byte prefix with a neighbour costs log^2 steps of full key
// compares through the table: URL-like keys (`https://example.com/...`: every key tied) took 0.76 s at one million keys
// and 18 s at ten million.  Instead the tied records are refined group by group: a round keys every record by its
// group (records equal so far) and its next seven key bytes (+ how many there were: a key that ends is smaller than its
// extensions), sorts by (group, chunk) -- two stable radix sorts, chunk then group -- writes the new order into the
// groups' positions, and keeps only the records that still tie with a neighbour, in finer groups.  A shared prefix of
// p bytes costs p / 7 rounds over the tied records, each a few passes of 32 bytes per record.
__global__ void k_tie_groups0(const uint64_t* __restrict__ tprefix, int64_t tied, int32_t* __restrict__ head) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < tied) head[j] = (j == 0 || tprefix[j] != tprefix[j - 1]) ? 1 : 0;
}
__global__ void k_tie_gid(const int32_t* __restrict__ head, const int64_t* __restrict__ hpos, int64_t tied, int32_t* __restrict__ gid) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < tied) gid[j] = (int32_t)(hpos[j] + head[j] - 1);  // (number of group heads up to and including j) - 1
}
__global__ void k_tie_keys(ColView in, const Entry* __restrict__ table, const int32_t* __restrict__ titem, int64_t tied, int depth,
                           uint64_t* __restrict__ key, int32_t* __restrict__ perm) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= tied) return;
  const Entry e = table[titem[j]];
  const int64_t b = entry_off(e);
  const int n = entry_len(e);
  const int rem = n - depth;
  const int cnt = rem < 0 ? (rem < -8 ? -8 : rem) : (rem > 7 ? 7 : rem);  // bytes of the key inside this chunk (negative: it ended before)
  uint64_t k = 0;
  for (int i = 0; i < 7; ++i) k = (k << 8) | (i < cnt ? in.chars[b + depth + i] : 0);
  key[j] = (k << 8) | (uint64_t)(cnt + 8);
  perm[j] = (int32_t)j;
}
__global__ void k_tie_group_keys(const int32_t* __restrict__ gid, const int32_t* __restrict__ perm, int64_t tied, uint64_t* __restrict__ gk,
                                 int32_t* __restrict__ perm2) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= tied) return;
  gk[j] = (uint64_t)(uint32_t)gid[perm[j]];
  perm2[j] = (int32_t)j;
}
// the records in their new order (group-major, chunk-minor), written into the groups' positions of the full sequence
__global__ void k_tie_apply(const int32_t* __restrict__ titem, const int32_t* __restrict__ gid, const uint64_t* __restrict__ key1,
                            const int32_t* __restrict__ perm, const int32_t* __restrict__ perm2, const int32_t* __restrict__ where, int64_t tied,
                            int32_t* __restrict__ titem2, int32_t* __restrict__ gid2, uint64_t* __restrict__ key2, int32_t* __restrict__ item) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= tied) return;
  const int32_t p2 = perm2[j], src = perm[p2];
  const int32_t it = titem[src];
  titem2[j] = it;
  gid2[j] = gid[src];
  key2[j] = key1[p2];
  item[where[j]] = it;
}
// which records still tie with a neighbour (same group, same chunk), which of them begin a finer group, and whether any
// of them has key bytes left (chunk count 7: the key goes on)
__global__ void k_tie_flags2(const int32_t* __restrict__ gid2, const uint64_t* __restrict__ key2, int64_t tied, int32_t* __restrict__ flags,
                             int32_t* __restrict__ head, int* __restrict__ more) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= tied) return;
  const int32_t g = gid2[j];
  const uint64_t k = key2[j];
  const bool prev = j > 0 && gid2[j - 1] == g && key2[j - 1] == k;
  const bool next = j + 1 < tied && gid2[j + 1] == g && key2[j + 1] == k;
  const bool t = prev || next;
  flags[j] = t ? 1 : 0;
  head[j] = t && !prev ? 1 : 0;
  if (t && (k & 255u) == 15u) *more = 1;
}
__global__ void k_tie_compact(const int32_t* __restrict__ flags, const int64_t* __restrict__ pos, const int32_t* __restrict__ titem2,
                              const int32_t* __restrict__ where, const int32_t* __restrict__ head, const int64_t* __restrict__ hpos, int64_t tied,
                              int32_t* __restrict__ titem, int32_t* __restrict__ where_out, int32_t* __restrict__ gid) {
  int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= tied || !flags[j]) return;
  const int64_t d = pos[j];
  titem[d] = titem2[j];
  where_out[d] = where[j];
  gid[d] = (int32_t)(hpos[j] + head[j] - 1);
}
__global__ void k_cat_ranks(const int32_t* __restrict__ item, int64_t uniq, int shift,
                            int32_t* __restrict__ rank_of_slot) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < uniq) rank_of_slot[item[i]] = (int32_t)i + shift;
}
// (four rows a thread: one 16-byte load, four look-ups in flight, one 16-byte store -- a row a thread left the memory pipe
// with 4-byte requests and one look-up a lane outstanding: 0.75 ms for 125M rows)
__global__ void k_cat_values(const int32_t* __restrict__ slot_of_row, const int32_t* __restrict__ rank_of_slot,
                             int64_t rows, int32_t* __restrict__ values) {
  const int64_t r = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
  if (r >= rows) return;
  if (r + 4 <= rows) {
    const int4 s = *reinterpret_cast<const int4*>(slot_of_row + r);
    int4 v;
    v.x = s.x < 0 ? 0 : rank_of_slot[s.x];  // a null row maps to key 0 (the null key)
    v.y = s.y < 0 ? 0 : rank_of_slot[s.y];
    v.z = s.z < 0 ? 0 : rank_of_slot[s.z];
    v.w = s.w < 0 ? 0 : rank_of_slot[s.w];
    *reinterpret_cast<int4*>(values + r) = v;
    return;
  }
  for (int64_t i = r; i < rows; ++i) {
    const int32_t s = slot_of_row[i];
    values[i] = s < 0 ? 0 : rank_of_slot[s];
  }
}
__global__ void k_key_sizes(ColView in, const Entry* __restrict__ table, const int32_t* __restrict__ item,
                            int64_t nkeys, int shift, int32_t* __restrict__ lens) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nkeys) return;
  if (i < shift) {
    lens[i] = -1;  // the null key
    return;
  }
  lens[i] = entry_len(table[item[i - shift]]);
}
__global__ void k_key_copy(ColView in, const Entry* __restrict__ table, const int32_t* __restrict__ item,
                           int64_t nkeys, int shift, const int64_t* __restrict__ out_off,
                           uint8_t* __restrict__ out_chars) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nkeys || i < shift) return;
  const Entry e = table[item[i - shift]];
  const uint8_t* p = in.chars + entry_off(e);
  int n = entry_len(e);
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

}  // namespace
namespace cs {
cs_category* category_build(const cs_column* col, hipStream_t s);
}
namespace {
cs_category* build(const cs_column* col, hipStream_t s) { return cs::category_build(col, s); }
}  // namespace
cs_category* cs::category_build(const cs_column* col, hipStream_t s) {
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
  // slot ids are int32 (slot_of_row, rank_of_slot; a negative id means "null row"): a table for all-distinct rows
  // beyond 2^30 rows would need ids from 2^31 on -- such a column is refused when (and only when) it needs that table
  const bool full_out_of_range = full > ((int64_t)1 << 31);
  ColView in = view_of(col);
  Buf slot_of_row = dev_alloc(sizeof(int32_t) * rows, s);
  Buf flags_d = dev_alloc(2 * sizeof(int), s);  // [0] has_null, [1] overflow
  Buf table;
  int first_log2 = 22;
  if (const char* e = cs::cfg("CS_CAT_FIRST_LOG2")) first_log2 = std::max(4, std::min(30, atoi(e)));  // tests: force retries
  int64_t cap = std::min<int64_t>(full, cs::cfg("CS_CAT_FULL_TABLE") && !full_out_of_range ? full : (int64_t)1 << first_log2);
  constexpr int64_t kSampleRows = 1 << 21;  // (that many rows fit the small table whatever they hold)
  bool sampled = cs::cfg("CS_CAT_NO_SAMPLE") != nullptr;
  int sw = 1;  // words per slot: 4 = the key's first bytes inside the slot (k_cat_insert)
  for (;;) {
    sw = cap <= ((int64_t)1 << 24) && !cs::cfg("CS_CAT_PLAIN_SLOTS") ? 4 : 1;
    table = dev_alloc(sizeof(Entry) * cap * sw, s);
    CS_HIP(hipMemsetAsync(table->p, 0xFF, sizeof(Entry) * cap * sw, s));
    CS_HIP(hipMemsetAsync(flags_d->p, 0, 2 * sizeof(int), s));
    {
      ProfScope ps("k_cat_insert", s);
      const int limit = cap == full ? 0x7fffffff : kProbeLimit;
      const int dbg = cs::cfg_int("CS_CAT_DEBUG", 0);
      hipLaunchKernelGGL(k_cat_insert, dim3(blocks_for(rows)), dim3(kBlock), 0, s, in, ptr<Entry>(table),
                         (uint32_t)(cap - 1), ptr<int32_t>(slot_of_row), ptr<int>(flags_d), ptr<int>(flags_d) + 1, limit,
                         dbg, sw);
    }
    int* h = (int*)pinned_scratch(2 * sizeof(int));
    CS_HIP(hipMemcpyAsync(h, flags_d->p, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (h[1] & 2) fail(CS_ERR_RANGE, "category: a key of 16 MiB or more");
    if (cap == full || !h[1]) break;  // (room for all-distinct rows: no limit applied, nothing to retry)
    int64_t next_cap = std::min<int64_t>(full, cap * 16);
    if (!sampled && rows > kSampleRows && cap >= 2 * kSampleRows) {
      // The small table overflowed.  How many distinct keys is this?  A sample of rows taken at a fixed stride across the
      // WHOLE column (the first rows alone say nothing about a column sorted or clustered by key) gives the share of
      // distinct keys among that many rows -- for a shuffled column an upper bound of the column's share (it only falls as
      // more rows come), in general a heuristic: the table is sized from it instead of growing 16-fold per lost pass over
      // all rows (K = 0.5 N: the 64M-slot attempt ran to 90 % before it gave up), and a table that still turns out too
      // small grows as before.
      sampled = true;
      CS_HIP(hipMemsetAsync(table->p, 0xFF, sizeof(Entry) * cap * sw, s));
      CS_HIP(hipMemsetAsync(flags_d->p, 0, 2 * sizeof(int), s));
      // (a stride rounded UP, so that the sample spans the whole column -- rounded down it stopped short of the column's end,
      // by almost half of it when the column has just under 2 x kSampleRows rows)
      const int64_t sample_stride = (rows + kSampleRows - 1) / kSampleRows, sample_rows = (rows + sample_stride - 1) / sample_stride;
      hipLaunchKernelGGL(k_cat_insert, dim3(blocks_for(sample_rows)), dim3(kBlock), 0, s, in, ptr<Entry>(table), (uint32_t)(cap - 1),
                         ptr<int32_t>(slot_of_row), ptr<int>(flags_d), ptr<int>(flags_d) + 1, 0x7fffffff, 0, sw, sample_stride, sample_rows);
      Buf occ = dev_alloc(sizeof(int32_t) * cap, s);
      hipLaunchKernelGGL(k_cat_flags, dim3(blocks_for(cap)), dim3(kBlock), 0, s, ptr<const Entry>(table), cap, sw, ptr<int32_t>(occ), (Entry*)nullptr);
      Buf occ_pos = dev_alloc(sizeof(int64_t) * (cap + 1), s);
      const int64_t distinct = offsets_from_lengths(ptr<int32_t>(occ), cap, ptr<int64_t>(occ_pos), s);
      const double share = (double)distinct / (double)sample_rows;
      int64_t want = 256;
      while (want < (int64_t)(2.5 * share * (double)rows)) want <<= 1;
      next_cap = std::min<int64_t>(full, std::max<int64_t>(next_cap, want));
    }
    cap = next_cap;
    if (cap == full && full_out_of_range) fail(CS_ERR_RANGE, "category: more than 2^30 rows with more distinct keys than a 2^31-slot table holds");
    // slot ids travel as int32 (negative = null row): a table of 2^31 slots or more cannot be addressed
    if (cap > ((int64_t)1 << 30)) fail(CS_ERR_RANGE, "category: more than 2^29 rows with mostly distinct keys in one column");
  }
  Buf has_null_d = flags_d;
  // compact the occupied slots
  Buf flags = dev_alloc(sizeof(int32_t) * cap, s);
  {
    // (32-byte slots: the slot words move to a dense array in the same pass -- every later kernel reads that)
    Buf dense = sw == 1 ? table : dev_alloc(sizeof(Entry) * cap, s);
    hipLaunchKernelGGL(k_cat_flags, dim3(blocks_for(cap)), dim3(kBlock), 0, s, ptr<const Entry>(table), cap, sw, ptr<int32_t>(flags),
                       sw == 1 ? (Entry*)nullptr : ptr<Entry>(dense));
    table = dense;
  }
  Buf pos = dev_alloc(sizeof(int64_t) * (cap + 1), s);
  const int64_t uniq = offsets_from_lengths(ptr<int32_t>(flags), cap, ptr<int64_t>(pos), s);
  const int shift = read_back<int>(has_null_d->p, s) ? 1 : 0;
  // Distinct keys in order.  The records (8-byte big-endian prefix, slot) are radix-sorted on the prefix -- linear in the
  // key count, where the bitonic network this replaces took log^2 K steps over the padded records and compared whole
  // keys through the table on every tie (K = 100M: hundreds of global steps).  Keys that share a prefix with a
  // neighbour keep their set of positions; the order among them comes from full compares: those records alone go
  // through the compare network (a handful for ordinary columns; every one only when all keys share 8 bytes).
  auto bitonic = [&](uint64_t* pfx, int32_t* itm, int64_t padded) {
    // stages that fit a chunk run in LDS in one launch; a later stage takes its long-distance
    // steps one launch each and finishes in LDS
    const int64_t chunk = std::min<int64_t>(padded, kSortChunk);
    const unsigned nchunks = (unsigned)(padded / chunk);
    if (padded >= 2)
      hipLaunchKernelGGL(k_bitonic_local, dim3(nchunks), dim3(256), 0, s, in, ptr<const Entry>(table), pfx, itm, padded, (int64_t)2, chunk);
    for (int64_t k = 2 * chunk; k <= padded; k <<= 1) {
      for (int64_t j = k >> 1; j >= chunk; j >>= 1)
        hipLaunchKernelGGL(k_bitonic_step, dim3(blocks_for(padded)), dim3(kBlock), 0, s, in, ptr<const Entry>(table), pfx, itm, padded, j, k);
      hipLaunchKernelGGL(k_bitonic_local, dim3(nchunks), dim3(256), 0, s, in, ptr<const Entry>(table), pfx, itm, padded, k, k);
    }
  };
  // (a key set that one LDS chunk holds sorts in a single launch of the compare network: the radix sort's histogram and
  // scan round trips would cost more than they save there)
  const bool use_radix = !cs::cfg("CS_CAT_BITONIC") && (uniq > kSortChunk || cs::cfg("CS_CAT_RADIX"));
  int64_t padded = 1;
  while (padded < uniq) padded <<= 1;
  if (use_radix) padded = std::max<int64_t>(uniq, 1);
  Buf prefix = dev_alloc(sizeof(uint64_t) * padded, s);
  Buf item = dev_alloc(sizeof(int32_t) * padded, s);
  hipLaunchKernelGGL(k_cat_records, dim3(blocks_for(std::max(cap, padded))), dim3(kBlock), 0, s, in,
                     ptr<const Entry>(table), ptr<const int32_t>(flags), ptr<const int64_t>(pos), cap, padded,
                     uniq, ptr<uint64_t>(prefix), ptr<int32_t>(item));
  {
    ProfScope ps("k_cat_sort", s);
    if (!use_radix) {
      bitonic(ptr<uint64_t>(prefix), ptr<int32_t>(item), padded);
    } else if (uniq > 1) {
      radix_sort_pairs64(ptr<uint64_t>(prefix), ptr<int32_t>(item), uniq, s);
      Buf tflags = dev_alloc(sizeof(int32_t) * uniq, s);
      hipLaunchKernelGGL(k_tie_flags, dim3(blocks_for(uniq)), dim3(kBlock), 0, s, ptr<const uint64_t>(prefix), uniq, ptr<int32_t>(tflags));
      Buf tpos = dev_alloc(sizeof(int64_t) * (uniq + 1), s);
      const int64_t tied = offsets_from_lengths(ptr<int32_t>(tflags), uniq, ptr<int64_t>(tpos), s);
      if (tied > 0) {
        int64_t tpad = 1;
        while (tpad < tied) tpad <<= 1;
        Buf tprefix = dev_alloc(sizeof(uint64_t) * tpad, s), titem = dev_alloc(sizeof(int32_t) * tpad, s), where = dev_alloc(sizeof(int32_t) * tied, s);
        hipLaunchKernelGGL(k_tie_gather, dim3(blocks_for(std::max(uniq, tpad))), dim3(kBlock), 0, s, ptr<const uint64_t>(prefix), ptr<const int32_t>(item),
                           ptr<const int32_t>(tflags), ptr<const int64_t>(tpos), uniq, tied, tpad, ptr<uint64_t>(tprefix), ptr<int32_t>(titem),
                           ptr<int32_t>(where));
        if (tied <= kSortChunk || cs::cfg("CS_CAT_TIE_NETWORK")) {
          // (a handful of ties: one launch of the compare network in LDS)
          bitonic(ptr<uint64_t>(tprefix), ptr<int32_t>(titem), tpad);
          hipLaunchKernelGGL(k_tie_scatter, dim3(blocks_for(tied)), dim3(kBlock), 0, s, ptr<const int32_t>(titem), ptr<const int32_t>(where), tied,
                             ptr<int32_t>(item));
        } else {
          // rounds on the next seven key bytes (see above)
          int64_t nt = tied;
          Buf gid = dev_alloc(sizeof(int32_t) * nt, s), head = dev_alloc(sizeof(int32_t) * nt, s), hpos = dev_alloc(sizeof(int64_t) * (nt + 1), s);
          hipLaunchKernelGGL(k_tie_groups0, dim3(blocks_for(nt)), dim3(kBlock), 0, s, ptr<const uint64_t>(tprefix), nt, ptr<int32_t>(head));
          offsets_from_lengths(ptr<int32_t>(head), nt, ptr<int64_t>(hpos), s);
          hipLaunchKernelGGL(k_tie_gid, dim3(blocks_for(nt)), dim3(kBlock), 0, s, ptr<const int32_t>(head), ptr<const int64_t>(hpos), nt, ptr<int32_t>(gid));
          Buf key1 = dev_alloc(sizeof(uint64_t) * nt, s), key2 = dev_alloc(sizeof(uint64_t) * nt, s), gk = dev_alloc(sizeof(uint64_t) * nt, s);
          Buf perm = dev_alloc(sizeof(int32_t) * nt, s), perm2 = dev_alloc(sizeof(int32_t) * nt, s);
          Buf titem2 = dev_alloc(sizeof(int32_t) * nt, s), gid2 = dev_alloc(sizeof(int32_t) * nt, s), where2 = dev_alloc(sizeof(int32_t) * nt, s);
          Buf flags2 = dev_alloc(sizeof(int32_t) * nt, s), fpos = dev_alloc(sizeof(int64_t) * (nt + 1), s), more_d = dev_alloc(sizeof(int), s);
          int32_t* cur_where = ptr<int32_t>(where);
          int32_t* alt_where = ptr<int32_t>(where2);
          for (int depth = 8; nt > 0; depth += 7) {
            const unsigned g = blocks_for(nt);
            hipLaunchKernelGGL(k_tie_keys, dim3(g), dim3(kBlock), 0, s, in, ptr<const Entry>(table), ptr<const int32_t>(titem), nt, depth, ptr<uint64_t>(key1),
                               ptr<int32_t>(perm));
            radix_sort_pairs64(ptr<uint64_t>(key1), ptr<int32_t>(perm), nt, s);
            hipLaunchKernelGGL(k_tie_group_keys, dim3(g), dim3(kBlock), 0, s, ptr<const int32_t>(gid), ptr<const int32_t>(perm), nt, ptr<uint64_t>(gk),
                               ptr<int32_t>(perm2));
            radix_sort_pairs64(ptr<uint64_t>(gk), ptr<int32_t>(perm2), nt, s);
            hipLaunchKernelGGL(k_tie_apply, dim3(g), dim3(kBlock), 0, s, ptr<const int32_t>(titem), ptr<const int32_t>(gid), ptr<const uint64_t>(key1),
                               ptr<const int32_t>(perm), ptr<const int32_t>(perm2), cur_where, nt, ptr<int32_t>(titem2), ptr<int32_t>(gid2), ptr<uint64_t>(key2),
                               ptr<int32_t>(item));
            CS_HIP(hipMemsetAsync(more_d->p, 0, sizeof(int), s));
            hipLaunchKernelGGL(k_tie_flags2, dim3(g), dim3(kBlock), 0, s, ptr<const int32_t>(gid2), ptr<const uint64_t>(key2), nt, ptr<int32_t>(flags2),
                               ptr<int32_t>(head), ptr<int>(more_d));
            const int64_t left = offsets_from_lengths(ptr<int32_t>(flags2), nt, ptr<int64_t>(fpos), s);
            if (left == 0 || !read_back<int>(more_d->p, s)) break;  // (what still ties has no bytes left: equal keys, in input order)
            offsets_from_lengths(ptr<int32_t>(head), nt, ptr<int64_t>(hpos), s);
            hipLaunchKernelGGL(k_tie_compact, dim3(g), dim3(kBlock), 0, s, ptr<const int32_t>(flags2), ptr<const int64_t>(fpos), ptr<const int32_t>(titem2),
                               cur_where, ptr<const int32_t>(head), ptr<const int64_t>(hpos), nt, ptr<int32_t>(titem), alt_where, ptr<int32_t>(gid));
            std::swap(cur_where, alt_where);
            nt = left;
          }
        }
        CS_HIP(hipStreamSynchronize(s));  // (the tie buffers' lifetime)
      }
    }
  }
  Buf rank_of_slot = dev_alloc(sizeof(int32_t) * cap, s);
  if (uniq)
    hipLaunchKernelGGL(k_cat_ranks, dim3(blocks_for(uniq)), dim3(kBlock), 0, s, ptr<const int32_t>(item), uniq,
                       shift, ptr<int32_t>(rank_of_slot));
  cat->values = dev_alloc(sizeof(int32_t) * rows, s);
  {
    ProfScope ps("k_cat_values", s);
    hipLaunchKernelGGL(k_cat_values, dim3(blocks_for((rows + 3) / 4)), dim3(kBlock), 0, s, ptr<const int32_t>(slot_of_row),
                       ptr<const int32_t>(rank_of_slot), rows, ptr<int32_t>(cat->values));
  }
  // keys column
  const int64_t nkeys = uniq + shift;
  auto keys = std::make_unique<cs_column>();
  keys->rows = nkeys;
  keys->null_count = shift;
  Buf lens = dev_alloc(sizeof(int32_t) * nkeys, s);
  hipLaunchKernelGGL(k_key_sizes, dim3(blocks_for(nkeys)), dim3(kBlock), 0, s, in, ptr<const Entry>(table),
                     ptr<const int32_t>(item), nkeys, shift, ptr<int32_t>(lens));
  keys->offsets = dev_alloc(sizeof(int64_t) * (nkeys + 1), s);
  LenMeta meta;
  keys->nbytes = offsets_from_lengths(ptr<int32_t>(lens), nkeys, ptr<int64_t>(keys->offsets), s, nullptr, &meta);
  meta.give(keys.get());
  keys->chars = dev_alloc((size_t)keys->nbytes, s);
  if (shift) keys->validity = validity_from_lengths(ptr<int32_t>(lens), nkeys, s);
  hipLaunchKernelGGL(k_key_copy, dim3(blocks_for(nkeys)), dim3(kBlock), 0, s, in, ptr<const Entry>(table),
                     ptr<const int32_t>(item), nkeys, shift, keys->d_offsets(), ptr<uint8_t>(keys->chars));
  CS_HIP(hipStreamSynchronize(s));
  cat->keys = std::move(keys);
  return cat.release();
}

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

// The step of a DISTRIBUTED category build that follows the exchange (BASELINE.json north_star: "an RCCL all-gather ...
// only to merge NVCategory's global key set"; the reference has no multi-GPU form: this composes create_from_categories,
// NVCategory.cu:430-514).  Every rank built the category of its own row range and all-gathered the ranks' key sets --
// with RCCL, MPI or anything else: the transport stays with the caller, so a C++ host reaches the global category without
// the Python layer (custrings_amd/dist.py does the same composition for torch.distributed).  `keysets[r]` = rank r's
// sorted key set as a column (this rank's own at index `rank`, equal to local->keys).  Out: the merged key set (the same
// on every rank) and, in `values` (device memory, local->rows int32), the local rows' codes in the merged key set.
int cs_category_merge_gathered(const cs_category* local, const cs_column* const* keysets, int nranks, int rank, cs_stream stream,
                               cs_column** merged_keys, int32_t* values) {
  return guard([&] {
    if (!local || !keysets || !merged_keys || nranks < 1 || rank < 0 || rank >= nranks) fail(CS_ERR_INVALID_ARG, "merge_gathered: bad arguments");
    if (local->rows > 0 && !values) fail(CS_ERR_INVALID_ARG, "merge_gathered: no room for the values");
    require_device();
    hipStream_t s = S(stream);
    std::vector<const cs_column*> sets;
    int64_t before = 0;
    for (int r = 0; r < nranks; ++r) {
      if (!keysets[r]) fail(CS_ERR_INVALID_ARG, "merge_gathered: null key set");
      if (r < rank) before += keysets[r]->rows;
      sets.push_back(keysets[r]);
    }
    if (keysets[rank]->rows != local->keys->rows) fail(CS_ERR_INVALID_ARG, "merge_gathered: keysets[rank] is not the local key set");
    // the category of the concatenated key sets: its keys are the merged key set, its codes the old-code -> new-code
    // tables of the ranks, back to back
    std::unique_ptr<cs_column> all_keys(concat_columns(sets, s));
    std::unique_ptr<cs_category> merged(build(all_keys.get(), s));
    if (local->rows)
      hipLaunchKernelGGL(k_remap, dim3(blocks_for(local->rows)), dim3(kBlock), 0, s, ptr<const int32_t>(local->values), local->rows,
                         ptr<const int32_t>(merged->values) + before, values);
    CS_HIP(hipGetLastError());
    CS_HIP(hipStreamSynchronize(s));
    *merged_keys = merged->keys.release();
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
