// split() on a single-byte delimiter, tile formulation (NVStrings::split,
// split.cu:734-822; token rules: custring_view.inl:1223-1250).
//
// Two passes over the chars buffer, both one wave per sub-tile of 64 consecutive
// rows with the sub-tile's contiguous chars span staged in LDS by coalesced
// 16-byte loads:
//   pass 1  k_split_measure: tokens per row (SWAR delimiter search), the global
//           maximum (= number of output columns) and, per sub-tile and column,
//           the bytes that column receives.  A segmented scan of those sums
//           gives every (sub-tile, column) its position in the column's chars.
//   pass 2  k_split_emit: walks the tokens again; column k's tokens of the 64
//           rows are contiguous in column k's chars buffer, so they are
//           assembled in LDS and flushed with 16-byte stores; offsets (one wave
//           scan per column) and validity words (one ballot per column) are
//           written coalesced.  All columns come out of this single pass.
// Rows with more than kMaxCols tokens, multi-byte delimiters and whitespace
// splitting use the generic kernels in cs_ops.hip.
#include <hip/hip_runtime.h>

#include <vector>

#include "cs_internal.h"
#include "device_utils.h"
#include "tile_utils.h"

using namespace cs;
using namespace csdev;

namespace cs {
bool split_fast(const cs_column* col, unsigned char delim, int tokens, hipStream_t s,
                std::vector<std::unique_ptr<cs_column>>& cols);
}

namespace {

constexpr int kSub = 64;
constexpr int kMaxCols = 32;

struct RowWords {  // a row inside an LDS buffer, read through aligned 32-bit words
  const uint8_t* base;  // 4-byte aligned
  int beg;              // byte index of the row's first byte
  int n;
  int cwi;
  uint32_t cw;
  __device__ __forceinline__ RowWords(const uint8_t* b, int begin, int len) : base(b), beg(begin), n(len), cwi(-1), cw(0) {}
  __device__ __forceinline__ uint32_t word(int widx) {
    if (widx != cwi) {
      cwi = widx;
      cw = reinterpret_cast<const uint32_t*>(base)[widx];
    }
    return cw;
  }
  // first position p >= pos (row-relative) holding the delimiter, or n
  __device__ __forceinline__ int find(int pos, uint32_t dpat) {
    while (pos < n) {
      const int j = beg + pos;
      const uint32_t x = word(j >> 2) ^ dpat;
      // exact per-byte zero test (no borrow between bytes: a flag below the masked-off
      // part must not create a false hit above it)
      const uint32_t m = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u & (0xFFFFFFFFu << (8 * (j & 3)));
      if (m) {
        pos = (j & ~3) - beg + (__builtin_ctz(m) >> 3);
        return pos < n ? pos : n;
      }
      pos = (j & ~3) + 4 - beg;
    }
    return n;
  }
};

__device__ __forceinline__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
__device__ __forceinline__ long long rl64(long long v, int k) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), k);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), k);
  return ((long long)hi << 32) | (unsigned int)lo;
}

struct SubTile {
  long long r0, g0;
  int nrows, rbeg, n, lead;
  bool live;
};
// loads the sub-tile's row extents and stages its chars span into `lds_in`
__device__ __forceinline__ SubTile load_subtile(const ColView& in, long long sub, uint8_t* lds_in, int lane) {
  SubTile t;
  t.r0 = sub * kSub;
  t.nrows = (int)min((long long)kSub, in.rows - t.r0);
  const long long o0 = in.offsets[t.r0 + min(lane, t.nrows)];
  const long long o1 = in.offsets[t.r0 + min(lane + 1, t.nrows)];
  t.g0 = rl64(o0, 0);
  const long long g1 = rl64(o1, 63);
  t.live = lane < t.nrows && row_is_valid(in.validity, t.r0 + lane);
  t.rbeg = (int)(o0 - t.g0);
  t.n = t.live ? (int)(o1 - o0) : 0;
  t.lead = (int)((uintptr_t)(in.chars + t.g0) & 15);
  const uint8_t* src = in.chars + (t.g0 - t.lead);  // 16-byte aligned
  const int span = (int)(g1 - t.g0) + t.lead;
  for (int i = lane * 16; i < span; i += 64 * 16)
    *reinterpret_cast<uint4*>(lds_in + i) = *reinterpret_cast<const uint4*>(src + i);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return t;
}

// token walker: next() yields [lo, hi) of the row's next token, or false.
// Rows of up to 96 bytes (from their aligned start) get the positions of all their
// delimiters as a 96-bit mask held in registers, built in straight-line code from
// 24 aligned words; the walk is then ctz + clear-lowest-bit per token.  Longer rows
// search word by word.
struct Tokens {
  RowWords w;
  uint32_t dpat;
  int cursor, k, limit;  // limit: token index that swallows the rest (maxsplit), or -1
  bool more, masked;
  unsigned long long m_lo;  // delimiter bits 0..63 (bit q = byte at row offset q - sa)
  uint32_t m_hi;            // bits 64..95
  int sa;
  __device__ __forceinline__ Tokens(const uint8_t* base, int beg, int n, bool live, uint32_t d, int tokens)
      : w(base, beg, n), dpat(d), cursor(0), k(0), limit(tokens > 0 ? tokens - 1 : -1), more(live), masked(false),
        m_lo(0), m_hi(0), sa(beg & 3) {
    if (__all(!live || n + sa <= 96)) {  // wave-uniform choice keeps the unrolled build convergent
      masked = true;
      const uint32_t* words = reinterpret_cast<const uint32_t*>(base) + (beg >> 2);
      uint32_t r[3] = {0, 0, 0};
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        const uint32_t x = words[i] ^ d;
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
        const uint32_t nib = ((((z >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 15u;
        r[i >> 3] |= nib << (4 * (i & 7));
      }
      const int hi = sa + n;  // keep bits sa .. hi - 1
      r[0] &= 0xFFFFFFFFu << sa;
      r[0] &= hi >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (hi & 31));
      r[1] &= hi >= 64 ? 0xFFFFFFFFu : (hi <= 32 ? 0u : ~(0xFFFFFFFFu << (hi & 31)));
      r[2] &= hi >= 96 ? 0xFFFFFFFFu : (hi <= 64 ? 0u : ~(0xFFFFFFFFu << (hi & 31)));
      m_lo = ((unsigned long long)r[1] << 32) | r[0];
      m_hi = r[2];
    }
  }
  __device__ __forceinline__ int next_delim() {  // masked: position of the next delimiter, or n
    if (m_lo) {
      const int q = __builtin_ctzll(m_lo);
      m_lo &= m_lo - 1;
      return q - sa;
    }
    if (m_hi) {
      const int q = 64 + __builtin_ctz(m_hi);
      m_hi &= m_hi - 1;
      return q - sa;
    }
    return w.n;
  }
  __device__ __forceinline__ bool next(int& lo, int& hi) {
    if (!more) return false;
    lo = cursor;
    if (k == limit) {
      hi = w.n;
      more = false;
    } else {
      hi = masked ? next_delim() : w.find(cursor, dpat);
      if (hi >= w.n) more = false;
      else cursor = hi + 1;
    }
    ++k;
    return true;
  }
};

struct MeasureArgs {
  ColView in;
  uint32_t dpat;
  int tokens, cap;
  long long nsub;
  int32_t* colsum;  // [kMaxCols][nsub]
  int* max_count;
};
__global__ void __launch_bounds__(256) k_split_measure(MeasureArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap + 32);
  const long long sub = (long long)blockIdx.x * 4 + wv;
  if (sub >= a.nsub) return;
  SubTile t = load_subtile(a.in, sub, lds_in, lane);
  Tokens tk(lds_in, t.lead + t.rbeg, t.n, t.live, a.dpat, a.tokens);
  int count = 0;
  for (int k = 0;; ++k) {
    int lo, hi;
    const bool has = tk.next(lo, hi);
    if (!__any(has)) break;
    count += has;
    if (k < kMaxCols) {
      const int sum = wave_reduce_sum(has ? hi - lo : 0);
      if (lane == 0) a.colsum[(long long)k * a.nsub + sub] = sum;
    }
  }
  const int m = wave_reduce_max(count);
  // same-address atomics serialise in L2 (about 10 ns each): only waves that would
  // raise the maximum issue one; a stale (smaller) read merely costs an extra atomic
  if (lane == 0 && m > __hip_atomic_load(a.max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.max_count, m);
}

struct ColOut {
  uint8_t* chars;
  int64_t* offsets;
  uint8_t* validity;
  const int64_t* base;  // base[sub] = bytes of this column before sub-tile `sub` (nsub + 1 entries)
};
struct EmitArgs {
  ColView in;
  uint32_t dpat;
  int tokens, cap_in, cap_out, ncols;
  long long nsub;
  const ColOut* cols;
};
__global__ void __launch_bounds__(256) k_split_emit(EmitArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap_in + a.cap_out + 64);
  uint8_t* lds_out = lds_in + a.cap_in + 32;
  const long long sub = (long long)blockIdx.x * 4 + wv;
  if (sub >= a.nsub) return;
  SubTile t = load_subtile(a.in, sub, lds_in, lane);
  // lane k holds column k's destination for this sub-tile
  uint8_t* my_chars = nullptr;
  int64_t* my_off = nullptr;
  uint8_t* my_valid = nullptr;
  long long my_base = 0;
  int my_sum = 0, my_lead = 0;
  if (lane < a.ncols) {
    const ColOut c = a.cols[lane];
    my_chars = c.chars;
    my_off = c.offsets;
    my_valid = c.validity;
    my_base = c.base[sub];
    my_sum = (int)(c.base[sub + 1] - my_base);
    my_lead = (int)((uintptr_t)(my_chars + my_base) & 15);
  }
  // LDS regions: column k's bytes start at region_k + lead_k so that 16-byte chunks
  // of the region line up with 16-byte chunks of the destination
  const int padded = lane < a.ncols ? ((my_lead + my_sum + 15) & ~15) : 0;
  const int region = wave_inclusive_scan(padded) - padded;

  Tokens tk(lds_in, t.lead + t.rbeg, t.n, t.live, a.dpat, a.tokens);
  const bool last_tile = t.r0 + t.nrows == a.in.rows;
  for (int k = 0; k < a.ncols; ++k) {
    int lo = 0, hi = 0;
    const bool has = tk.next(lo, hi);
    const int len = has ? hi - lo : 0;
    const int incl = wave_inclusive_scan(len);
    const int pre = incl - len;
    const long long cbase = rl64(my_base, k);
    int64_t* coff = reinterpret_cast<int64_t*>(rl64((long long)(uintptr_t)my_off, k));
    uint8_t* cvalid = reinterpret_cast<uint8_t*>(rl64((long long)(uintptr_t)my_valid, k));
    const int cstart = rl(region, k) + rl(my_lead, k);
    if (lane < t.nrows) coff[t.r0 + lane] = cbase + pre;
    if (last_tile && lane == t.nrows - 1) coff[a.in.rows] = cbase + incl;
    const unsigned long long vmask = __ballot(has);
    if (lane == 0) *reinterpret_cast<unsigned long long*>(cvalid + sub * 8) = vmask;
    if (has) cstile::lds_copy_short(lds_out, cstart + pre, lds_in, t.lead + t.rbeg + lo, len);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int k = 0; k < a.ncols; ++k) {
    const int rstart = rl(region, k);
    const int lead = rl(my_lead, k);
    const int end = lead + rl(my_sum, k);
    uint8_t* dst = reinterpret_cast<uint8_t*>(rl64((long long)(uintptr_t)my_chars, k)) + rl64(my_base, k);
    cstile::wave_flush(dst, end - lead, lds_out + rstart, lead, lane);
  }
}

}  // namespace

namespace cs {

bool split_fast(const cs_column* col, unsigned char delim, int tokens, hipStream_t s,
                std::vector<std::unique_ptr<cs_column>>& cols) {
  const int64_t rows = col->rows;
  if (rows == 0 || getenv("CS_SPLIT_GENERIC")) return false;
  const int64_t span = max_span64(col, s);
  const int cap_in = (int)((span + 15 + 32 + 127) & ~(int64_t)127);
  const int cap_out = cap_in + 32 * kMaxCols;
  if ((size_t)(cap_in + cap_out + 64) * 4 > 150 * 1024) return false;
  const int64_t nsub = (rows + kSub - 1) / kSub;
  const unsigned grid = (unsigned)((nsub + 3) / 4);
  const uint32_t dpat = 0x01010101u * delim;

  Buf colsum = dev_alloc(sizeof(int32_t) * nsub * kMaxCols, s);
  CS_HIP(hipMemsetAsync(colsum->p, 0, sizeof(int32_t) * nsub * kMaxCols, s));  // columns a sub-tile never reaches
  Buf mx = dev_alloc(sizeof(int), s);
  CS_HIP(hipMemsetAsync(mx->p, 0, sizeof(int), s));
  MeasureArgs ma{view_of(col), dpat, tokens, cap_in, nsub, ptr<int32_t>(colsum), ptr<int>(mx)};
  {
    ProfScope ps("k_split_measure", s);
    hipLaunchKernelGGL(k_split_measure, dim3(grid), dim3(256), (size_t)(cap_in + 32) * 4, s, ma);
  }
  CS_HIP(hipGetLastError());
  int* hmx = (int*)pinned_scratch(sizeof(int));
  CS_HIP(hipMemcpyAsync(hmx, mx->p, sizeof(int), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  const int ncols = *hmx;
  if (ncols == 0 || ncols > kMaxCols) return false;  // all-null column / too many columns: generic path

  // per column: position of every sub-tile in the column's chars buffer
  Buf base = dev_alloc(sizeof(int64_t) * (nsub + 1) * ncols, s);
  std::vector<int64_t> totals(ncols);
  offsets_from_lengths_segmented(ptr<int32_t>(colsum), nsub, ncols, ptr<int64_t>(base), totals.data(), s);

  std::vector<ColOut> outs(ncols);
  for (int k = 0; k < ncols; ++k) {
    auto c = std::make_unique<cs_column>();
    c->rows = rows;
    c->nbytes = totals[k];
    c->chars = dev_alloc((size_t)totals[k], s);
    c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    c->validity = dev_alloc(validity_bytes(rows), s);
    outs[k] = ColOut{ptr<uint8_t>(c->chars), ptr<int64_t>(c->offsets), ptr<uint8_t>(c->validity),
                     ptr<const int64_t>(base) + (int64_t)k * (nsub + 1)};
    cols.push_back(std::move(c));
  }
  Buf d_outs = dev_alloc(sizeof(ColOut) * ncols, s);
  CS_HIP(hipMemcpyAsync(d_outs->p, outs.data(), sizeof(ColOut) * ncols, hipMemcpyHostToDevice, s));
  EmitArgs ea{view_of(col), dpat, tokens, cap_in, cap_out, ncols, nsub, ptr<const ColOut>(d_outs)};
  const size_t lds = (size_t)(cap_in + cap_out + 64) * 4;
  if (lds > 48 * 1024)
    CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_split_emit),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  {
    ProfScope ps("k_split_emit", s);
    hipLaunchKernelGGL(k_split_emit, dim3(grid), dim3(256), lds, s, ea);
  }
  CS_HIP(hipGetLastError());
  CS_HIP(hipStreamSynchronize(s));  // `outs` / `base` lifetime
  return true;
}

}  // namespace cs
