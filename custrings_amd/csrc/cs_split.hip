// split() on a single-byte delimiter, tile formulation (NVStrings::split,
// split.cu:734-822; token rules: custring_view.inl:1223-1250).
//
// Two passes over the chars buffer, both one wave per sub-tile of 64 consecutive
// rows with the sub-tile's contiguous chars span staged in LDS by coalesced
// 16-byte loads:
//   pass 1  k_split_measure2: tokens per row (SWAR delimiter search), the global
//           maximum (= number of output columns) and, per SEGMENT of consecutive
//           sub-tiles and column, the bytes that column receives (every row lane
//           keeps its own sum per column; one reduction per column per segment).
//           A small scan over the segments gives every emit wave the position of
//           its run of sub-tiles in every column's chars.
//   pass 2  k_split_emit2: walks the tokens again; each wave owns a contiguous run
//           of sub-tiles and carries its position in every column along; column
//           k's tokens of the 64 rows are contiguous in column k's chars buffer,
//           so they are assembled in LDS and flushed with 16-byte stores; offsets
//           (one wave scan per column; int32 when every column stays below 2 GiB,
//           which halves the bytes written per output row) and validity words
//           (one ballot per column) are written coalesced.  All columns come out
//           of this single pass.
//   (k_split_measure / k_split_emit: the first tile generation, one sub-tile per
//   wave with per-sub-tile column sums; kept for rows beyond 93 bytes.)
// Rows with more than kMaxCols tokens, multi-byte delimiters and whitespace
// splitting use the generic kernels in cs_ops.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "cs_internal.h"
#include "device_utils.h"
#include "tile_utils.h"

using namespace cs;
using namespace csdev;

namespace cs {
bool split_fast(const cs_column* col, const unsigned char* delim, int dlen, int tokens, hipStream_t s,
                std::vector<std::unique_ptr<cs_column>>& cols, bool reverse = false);
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
// WS (whitespace splitting, split.cu:863-956): a token is a maximal run of bytes above ' '
// (every byte <= ' ' separates, no non-ASCII byte does); runs of separators collapse, there are
// no empty tokens, and the token that exhausts maxsplit takes the rest of the row as it is.
// The masks then hold token STARTS (m_*) and token ENDS (e_*); only the masked form exists.
// MULTI (a delimiter of 2..8 ASCII bytes): the mask first holds the positions of the delimiter's
// first byte; each row lane then keeps those where the whole delimiter stands and that do not
// overlap the previous occurrence (the search continues after an occurrence, custring_view.inl:
// 1223-1279), and a token ends where the next kept position begins.  Masked form only.
template <bool MASKED_ONLY, bool WS = false, bool MULTI = false>
struct TokensT {
  RowWords w;
  uint32_t dpat;
  int cursor, k, limit;  // limit: token index that swallows the rest (maxsplit), or -1
  bool more, masked;
  unsigned long long m_lo;  // delimiter bits 0..63 (bit q = byte at row offset q - sa); WS: token starts
  uint32_t m_hi;            // bits 64..95
  unsigned long long e_lo;  // WS: last byte of each token
  uint32_t e_hi;
  int sa;
  int dlen;  // delimiter bytes (1 unless MULTI)
  // `reverse` (one-byte delimiter, a split limit, masked rows): rsplit -- the LAST `limit` delimiters of the row are the
  // ones that split (split.cu:1006-1021 finds them from the right), so the first ones are struck from the mask and
  // the forward walk over what is left yields rsplit's tokens, left-aligned in the columns as the reference has them.
  __device__ __forceinline__ TokensT(const uint8_t* base, int beg, int n, bool live, uint32_t d, int tokens,
                                     unsigned long long d64 = 0, int delim_len = 1, bool reverse = false)
      : w(base, beg, n), dpat(d), cursor(0), k(0), limit(tokens > 0 ? tokens - 1 : -1), more(live), masked(false),
        m_lo(0), m_hi(0), e_lo(0), e_hi(0), sa(beg & 3), dlen(MULTI ? delim_len : 1) {
    // (MASKED_ONLY: the caller guarantees that every row fits the 96-bit mask)
    if (MASKED_ONLY || __all(!live || n + sa <= 96)) {  // wave-uniform choice keeps the unrolled build convergent
      masked = true;
      const uint32_t* words = reinterpret_cast<const uint32_t*>(base) + (beg >> 2);
      uint32_t r[3] = {0, 0, 0};
      auto flags = [&](int i) -> uint32_t {  // bit 7 of every byte lane that separates
        if (WS) {
          const uint32_t x = words[i];  // byte <= 0x20: bit 7 clear and the low seven bits below 0x21
          return ~(((x & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | x) & 0x80808080u;
        }
        const uint32_t x = words[i] ^ d;
        return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
      };
#pragma unroll
      for (int i = 0; i < 24; i += 4)  // sixteen bytes -> sixteen bits (byte-wise dot products, tile_utils.h)
        r[i >> 3] |= cstile::gather16_bit7(flags(i), flags(i + 1), flags(i + 2), flags(i + 3)) << (4 * (i & 7));
      const int hi = sa + n;  // keep bits sa .. hi - 1
      uint32_t in0 = 0xFFFFFFFFu << sa;
      in0 &= hi >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (hi & 31));
      const uint32_t in1 = hi >= 64 ? 0xFFFFFFFFu : (hi <= 32 ? 0u : ~(0xFFFFFFFFu << (hi & 31)));
      const uint32_t in2 = hi >= 96 ? 0xFFFFFFFFu : (hi <= 64 ? 0u : ~(0xFFFFFFFFu << (hi & 31)));
      if (WS) {
        // token bytes of the row, then starts (byte before is not a token byte) and ends (byte after)
        const uint32_t t0 = ~r[0] & in0, t1 = ~r[1] & in1, t2 = ~r[2] & in2;
        const uint32_t up0 = t0 << 1, up1 = (t1 << 1) | (t0 >> 31), up2 = (t2 << 1) | (t1 >> 31);
        const uint32_t dn0 = (t0 >> 1) | (t1 << 31), dn1 = (t1 >> 1) | (t2 << 31), dn2 = t2 >> 1;
        m_lo = ((unsigned long long)(t1 & ~up1) << 32) | (t0 & ~up0);
        m_hi = t2 & ~up2;
        e_lo = ((unsigned long long)(t1 & ~dn1) << 32) | (t0 & ~dn0);
        e_hi = t2 & ~dn2;
        more = live && (m_lo != 0 || m_hi != 0);
      } else {
        m_lo = ((unsigned long long)(r[1] & in1) << 32) | (r[0] & in0);
        m_hi = r[2] & in2;
        if (!MULTI && reverse && limit >= 0) {
          int drop = __builtin_popcountll(m_lo) + __builtin_popcount(m_hi) - limit;  // delimiters that do not split
          while (__any(drop > 0)) {
            if (drop > 0) {
              if (m_lo) m_lo &= m_lo - 1;
              else m_hi &= m_hi - 1;
              --drop;
            }
          }
        }
        if (MULTI) {
          unsigned long long c_lo = m_lo, k_lo = 0;
          uint32_t c_hi = m_hi, k_hi = 0;
          int free_from = 0;  // row offset where the next occurrence may begin
          const uint8_t* row = base + beg;
          while (c_lo != 0 || c_hi != 0) {
            int q;
            if (c_lo) {
              q = __builtin_ctzll(c_lo);
              c_lo &= c_lo - 1;
            } else {
              q = 64 + __builtin_ctz(c_hi);
              c_hi &= c_hi - 1;
            }
            const int pos = q - sa;
            if (pos < free_from || pos + dlen > n) continue;
            int j = 1;
            while (j < dlen && row[pos + j] == (uint8_t)(d64 >> (8 * j))) ++j;
            if (j < dlen) continue;
            if (q < 64) k_lo |= 1ull << q;
            else k_hi |= 1u << (q - 64);
            free_from = pos + dlen;
          }
          m_lo = k_lo;
          m_hi = k_hi;
        }
      }
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
  __device__ __forceinline__ int next_end() {  // WS: position of the last byte of the next token
    if (e_lo) {
      const int q = __builtin_ctzll(e_lo);
      e_lo &= e_lo - 1;
      return q - sa;
    }
    const int q = 64 + __builtin_ctz(e_hi);
    e_hi &= e_hi - 1;
    return q - sa;
  }
  __device__ __forceinline__ bool next(int& lo, int& hi) {
    if (!more) return false;
    if (WS) {
      lo = next_delim();  // (the next token start)
      if (k == limit) {
        hi = w.n;
        more = false;
      } else {
        hi = next_end() + 1;
        more = m_lo != 0 || m_hi != 0;
      }
      ++k;
      return true;
    }
    lo = cursor;
    if (k == limit) {
      hi = w.n;
      more = false;
    } else {
      hi = (MASKED_ONLY || masked) ? next_delim() : w.find(cursor, dpat);
      if (hi >= w.n) more = false;
      else cursor = hi + dlen;
    }
    ++k;
    return true;
  }
};
using Tokens = TokensT<false>;

struct MeasureArgs {
  ColView in;
  uint32_t dpat;
  unsigned long long d64;  // the delimiter's bytes, first byte lowest (multi-byte delimiters)
  int dlen;
  int tokens, cap;
  long long nsub;
  int32_t* colsum;  // [kMaxCols][nsub]
  int* max_count;   // [0] most tokens in a row, [1] most bytes one column receives from one sub-tile, [2] longest row,
                    // [3] set when a sub-tile needs the generic kernels (whitespace mode: a row beyond the 96-bit masks)
};
// MODE 0: one-byte delimiter, 1: whitespace, 2: delimiter of 2..8 ASCII bytes
template <int MODE>
__global__ void __launch_bounds__(256) k_split_measure(MeasureArgs a) {
  constexpr bool WS = MODE == 1, MULTI = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap + 32);
  const long long sub = (long long)blockIdx.x * 4 + wv;
  if (sub >= a.nsub) return;
  SubTile t = load_subtile(a.in, sub, lds_in, lane);
  TokensT<false, WS, MULTI> tk(lds_in, t.lead + t.rbeg, t.n, t.live, a.dpat, a.tokens, a.d64, a.dlen);
  if ((WS || MULTI) && !tk.masked) {
    if (lane == 0) atomicMax(a.max_count + 3, 1);
    return;
  }
  int count = 0, widest = 0;
  for (int k = 0;; ++k) {
    int lo, hi;
    const bool has = tk.next(lo, hi);
    if (!__any(has)) break;
    count += has;
    if (k < kMaxCols) {
      const int sum = wave_reduce_sum(has ? hi - lo : 0);
      if (lane == 0) a.colsum[(long long)k * a.nsub + sub] = sum;
      widest = max(widest, sum);
    }
  }
  if (lane == 0 && widest > __hip_atomic_load(a.max_count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(a.max_count + 1, widest);
  const int longest = wave_reduce_max(t.n);
  if (lane == 0 && longest > __hip_atomic_load(a.max_count + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(a.max_count + 2, longest);
  const int m = wave_reduce_max(count);
  // same-address atomics serialise in L2 (about 10 ns each): only waves that would
  // raise the maximum issue one; a stale (smaller) read merely costs an extra atomic
  if (lane == 0 && m > __hip_atomic_load(a.max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.max_count, m);
}


// ---- measure, second generation: per-segment column sums ----------------------------------
// A wave walks a SEGMENT of consecutive sub-tiles (a fixed fraction of an emit wave's run); every
// row lane keeps the bytes its rows give to each column in registers and the 64 partial sums
// are reduced once per segment, so the per-sub-tile work is the token walk alone.
struct Measure2Args {
  ColView in;
  uint32_t dpat;
  unsigned long long d64;
  int dlen;
  int tokens, cap;
  long long nsub, per, seg, nseg;  // emit run length, segment length (sub-tiles), segments
  int segs_per_run;
  int reverse;  // rsplit with a limit (TokensT)
  int32_t* colsum;  // [kMaxCols][nseg], zeroed by the host
  int* max_count;   // [0] most tokens in a row, [1] bound on the bytes one column receives from one sub-tile
                    // (sum over its rows of the row's longest token), [2] longest row, [3] a sub-tile needs the generic kernels
};
template <int MODE>
__global__ void __launch_bounds__(256, 8) k_split_measure2(Measure2Args a) {  // (8 waves per SIMD: 2.05 -> 1.94 ms)
  constexpr bool WS = MODE == 1, MULTI = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap + 32);
  const long long sid = (long long)blockIdx.x * 4 + wv;
  if (sid >= a.nseg) return;
  const long long run = sid / a.segs_per_run;
  const long long t0 = run * a.per + (sid - run * a.segs_per_run) * a.seg;
  const long long t1 = min(min(t0 + a.seg, (run + 1) * a.per), a.nsub);
  int acc[kMaxCols];
#pragma unroll
  for (int k = 0; k < kMaxCols; ++k) acc[k] = 0;
  int most = 0, widest = 0, longest = 0;
  bool generic = false;
  for (long long sub = t0; sub < t1; ++sub) {
    SubTile t = load_subtile(a.in, sub, lds_in, lane);
    TokensT<false, WS, MULTI> tk(lds_in, t.lead + t.rbeg, t.n, t.live, a.dpat, a.tokens, a.d64, a.dlen, a.reverse != 0);
    if ((WS || MULTI || a.reverse) && !tk.masked) {  // (wave-uniform)
      generic = true;
      break;
    }
    int count = 0, rowmax = 0;
    bool any_more = true;
#pragma unroll
    for (int k = 0; k < kMaxCols; ++k) {
      if (any_more) {
        int lo = 0, hi = 0;
        const bool has = tk.next(lo, hi);
        any_more = __any(has);
        const int len = has ? hi - lo : 0;
        count += has;
        acc[k] += len;
        rowmax = max(rowmax, len);
      }
    }
    while (any_more) {  // rows with more than kMaxCols tokens: the host takes the generic path
      int lo = 0, hi = 0;
      const bool has = tk.next(lo, hi);
      any_more = __any(has);
      count += has;
    }
    most = max(most, count);
    longest = max(longest, t.n);
    widest = max(widest, wave_reduce_sum(rowmax));
    __builtin_amdgcn_wave_barrier();  // the next sub-tile overwrites the staged rows
  }
  if (generic) {
    if (lane == 0) atomicMax(a.max_count + 3, 1);
    return;
  }
  const int m = wave_reduce_max(most);
#pragma unroll
  for (int k = 0; k < kMaxCols; ++k) {
    if (k < m) {
      const int sum = wave_reduce_sum(acc[k]);
      if (lane == 0) a.colsum[(long long)k * a.nseg + sid] = sum;
    }
  }
  // same-address atomics serialise in L2: only waves that would raise a maximum issue one
  if (lane == 0 && widest > __hip_atomic_load(a.max_count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.max_count + 1, widest);
  const int lg = wave_reduce_max(longest);
  if (lane == 0 && lg > __hip_atomic_load(a.max_count + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.max_count + 2, lg);
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


// ---- persistent emit (tile_utils.h: sub-tile stream) ------------------------------------
// Same outputs as k_split_emit.  Differences: persistent waves with the next sub-tile's chars
// and column positions prefetched; one column at a time is assembled in a small LDS region
// (sized by the largest per-sub-tile column contribution the measure pass saw) and flushed
// right away, two regions alternating, so a wave needs cap_in + 2 * cap_col bytes of LDS
// instead of 2 * cap_in + 1 KB and twice as many waves are resident; short tokens reach the
// region as three ds_or_b32 of the funnel-shifted token instead of byte stores.
struct ColOut2 {
  uint8_t* chars;
  void* offsets;  // int32 or int64 (OFF32)
  uint8_t* validity;
  const int64_t* seg_base;  // seg_base[j] = bytes of this column before measure segment j
};
struct Emit2Args {
  ColView in;
  uint32_t dpat;
  unsigned long long d64;
  int dlen;
  int tokens, cap_in, cap_col, ncols;
  long long nsub;
  long long per;  // sub-tiles per wave (run length); wave w owns [w * per, (w + 1) * per)
  int segs_per_run;
  const ColOut2* cols;
  unsigned long long* prof;  // instrumented builds: 6 cycle counters
  int reverse;  // rsplit with a limit (TokensT)
  int debug;  // CS_SPLIT_DEBUG bit mask: 1 no offset stores, 2 no chars stores, 4 no assembly, 8 no column loop (measurement only)
};
__device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v) {
  __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
#ifndef CS_EMIT2_WAVES
#define CS_EMIT2_WAVES 4
#endif
template <int MODE, bool OFF32>
__global__ void __launch_bounds__(256, CS_EMIT2_WAVES) k_split_emit2(Emit2Args a) {
  constexpr bool WS = MODE == 1, MULTI = MODE == 2;
  typedef typename std::conditional<OFF32, int32_t, int64_t>::type off_t;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap_in + 32 + 2 * a.cap_col + (WS ? 2 : 1) * a.ncols * 64);
  uint8_t* region0 = lds_in + a.cap_in + 32;
  uint8_t* dpos = region0 + 2 * a.cap_col;  // dpos[j * 64 + lane] = row offset of the lane's j-th delimiter (WS: j-th token start)
  uint8_t* epos = dpos + a.ncols * 64;      // WS: last byte of the lane's j-th token
  // each wave owns a contiguous run of sub-tiles: its pieces of every output column are
  // contiguous too, so the cache lines that two neighbouring sub-tiles share are completed in
  // one L2 instead of being written half-filled from two XCDs
  const long long per = a.per;
  constexpr long long W = 1;
  const long long run = (long long)blockIdx.x * 4 + wv;
  long long tile = run * per;
  const long long tile_end = min(a.nsub, tile + per);
  if (tile >= tile_end) return;
  const ColView& in = a.in;
  // lane k keeps column k's destination and the wave's running position in column k's chars
  uint8_t* my_chars = nullptr;
  off_t* my_off = nullptr;
  uint8_t* my_valid = nullptr;
  long long my_pos = 0;
  if (lane < a.ncols) {
    const ColOut2 c = a.cols[lane];
    my_chars = c.chars;
    my_off = reinterpret_cast<off_t*>(c.offsets);
    my_valid = c.validity;
    my_pos = c.seg_base[run * a.segs_per_run];
  }
  cstile::TileOffs cur = cstile::load_tile_offsets(in.offsets, in.rows, tile, lane);
  cstile::TileOffs nxt = cur;
  if (tile + W < tile_end) nxt = cstile::load_tile_offsets(in.offsets, in.rows, tile + W, lane);
  cstile::TileChars pf;
#pragma unroll
  for (int j = 0; j < cstile::kPfChunks; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
#if defined(CS_PHASE_PROF)
  unsigned long long phase_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long phase_t = __builtin_readcyclecounter();
#endif
  for (;;) {
    const long long r0 = tile * 64;
    const int nrows = (int)min(64ll, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool live = lane < nrows && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n = live ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const int want = (int)(g1 - g0) + lead;
    cstile::stage_chars(lds_in, want, lane, pf);
    const long long my_base = my_pos;
    const int my_lead = (int)((uintptr_t)(my_chars + my_base) & 15);
    const bool has_next = tile + W < tile_end;
    // The offsets two sub-tiles ahead are fetched unconditionally (clamped index) and become `nxt` only at the bottom
    // of the iteration: assigned under a condition, a value still in flight is copied at the join of the branch, and
    // the compiler waited for it there -- s_waitcnt vmcnt(0) right behind the issue of the whole prefetch, every
    // iteration (the prefetch never overlapped the work it was meant to hide behind).
    const cstile::TileOffs nn = cstile::load_tile_offsets(in.offsets, in.rows, tile + 2 * W < tile_end ? tile + 2 * W : tile_end - 1, lane);
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
    }
    cstile::wave_lds_fence();
    CS_PHASE_MARK(0);

    TokensT<true, WS, MULTI> tk(lds_in, lead + rbeg, n, live, a.dpat, a.tokens, a.d64, a.dlen, a.reverse != 0);
    // every row's delimiter positions go to LDS once, in a loop that does nothing else; the
    // column loop then needs one byte load per token instead of the bit-mask walk
    const int nd = __builtin_popcountll(tk.m_lo) + __builtin_popcount(tk.m_hi);
    auto fill = [&](unsigned long long lo64, uint32_t hi32, uint8_t* table, int entries) {
      // word by word, lowest set bit first: ffbl, clear, one byte store per position
      uint32_t w[3] = {(uint32_t)lo64, (uint32_t)(lo64 >> 32), hi32};
      uint8_t* slot = table + lane;  // advances one table row (64 bytes) per position found
      const uint8_t* last = table + entries * 64;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        uint32_t v = w[i];
        while (__any(v != 0)) {
          if (v != 0) {
            const int q = 32 * i + __builtin_ctz(v) - tk.sa;
            v &= v - 1;
            if (slot < last) *slot = (uint8_t)q;
            slot += 64;
          }
        }
      }
    };
    if (WS) {
      fill(tk.m_lo, tk.m_hi, dpos, a.ncols);
      fill(tk.e_lo, tk.e_hi, epos, a.ncols);
    } else {
      fill(tk.m_lo, tk.m_hi, dpos, a.ncols - 1);
    }
    const int ntok = !live ? 0 : WS ? min(nd, a.tokens > 0 ? a.tokens : 1 << 20) : min(nd, a.tokens > 0 ? a.tokens - 1 : 1 << 20) + 1;
    cstile::wave_lds_fence();
    CS_PHASE_MARK(1);
    const bool last_tile = r0 + nrows == in.rows;
    unsigned long long my_vmask = 0;
    int cursor = 0;
    for (int k = 0; k < ((a.debug & 8) ? 0 : a.ncols); ++k) {
      uint8_t* region = region0 + (k & 1) * a.cap_col;
      const int dk = dpos[k * 64 + lane];
      const bool has = k < ntok;
      int lo, hi;
      if (WS) {  // the token that exhausts maxsplit keeps the rest of the row
        lo = dk;
        hi = (a.tokens > 0 && k == a.tokens - 1) ? n : epos[k * 64 + lane] + 1;
      } else {
        lo = cursor;
        hi = k == ntok - 1 ? n : dk;
        cursor = hi + (MULTI ? a.dlen : 1);
      }
      const int len = has ? hi - lo : 0;
      const int incl = wave_inclusive_scan(len);
      const int pre = incl - len;
      const long long cbase = cstile::rl64(my_base, k);
      cstile::gptr<off_t> coff = cstile::as_global(reinterpret_cast<off_t*>(cstile::rl64((long long)(uintptr_t)my_off, k)));
      const int clead = rl(my_lead, k);
      const int csum = rl(incl, 63);  // bytes this sub-tile adds to column k
      if (lane < nrows && !(a.debug & 1)) coff[r0 + lane] = (off_t)(cbase + pre);
      if (last_tile && lane == nrows - 1) coff[in.rows] = (off_t)(cbase + incl);
      const unsigned long long vmask = __ballot(has);
      if (lane == k) {
        my_vmask = vmask;
        my_pos += csum;
      }
      CS_PHASE_MARK(2);
      if (csum == 0 || (a.debug & 4)) continue;  // no row of this sub-tile reaches column k (or all its tokens are empty): offsets only
      if (has) {
        // the token goes to its place in the column's region with at most five stores of 16 / 8 / 4 / 2 / 1 bytes at
        // whatever alignment the position has (gfx950 takes DS accesses at any alignment): one 16-byte read of the
        // staged row, no zeroing of the region, no funnel shifts, no OR-assembly (round 1 composed the token from five
        // aligned dwords, shifted it to the destination's byte phase and OR-ed it into a zeroed region)
        const int ti = lead + rbeg + lo;
        const int di = clead + pre;
        cstile::lds_put16(region + di, *reinterpret_cast<const cstile::lds_u32x4u*>(lds_in + ti), len);
        if (len > 16) cstile::lds_copy(region, di + 16, lds_in, ti + 16, len - 16);
      }
      cstile::wave_lds_fence();
      CS_PHASE_MARK(3);
      uint8_t* dst = reinterpret_cast<uint8_t*>(cstile::rl64((long long)(uintptr_t)my_chars, k)) + cbase;
      if (!(a.debug & 2)) cstile::wave_flush(dst, csum, region, clead, lane);
      CS_PHASE_MARK(4);
    }
    if (lane < a.ncols) *cstile::as_global(reinterpret_cast<unsigned long long*>(my_valid + tile * 8)) = my_vmask;
    if (!has_next) break;
    tile += W;
    nxt = nn;
  }
#if defined(CS_PHASE_PROF)
  CS_PHASE_MARK(5);
  if (lane == 0 && a.prof)
    for (int k = 0; k < 6; ++k) atomicAdd(a.prof + k, phase_acc[k]);
#endif
}

}  // namespace

namespace cs {

// `delim`: 1..8 ASCII bytes, or nullptr for whitespace splitting.
// `reverse`: rsplit with a limit on a one-byte delimiter (the masked kernels only; anything else returns false).
bool split_fast(const cs_column* col, const unsigned char* delim, int dlen, int tokens, hipStream_t s,
                std::vector<std::unique_ptr<cs_column>>& cols, bool reverse) {
  const bool ws = delim == nullptr;
  const int mode = ws ? 1 : (dlen > 1 ? 2 : 0);
  if (reverse && (mode != 0 || tokens <= 0)) return false;
  unsigned long long d64 = 0;
  for (int i = 0; !ws && i < dlen; ++i) d64 |= (unsigned long long)delim[i] << (8 * i);
  const int64_t rows = col->rows;
  if (rows == 0 || getenv("CS_SPLIT_GENERIC")) return false;
  const int64_t span = max_span64(col, s);
  const int cap_in = (int)((span + 15 + 32 + 127) & ~(int64_t)127);
  const int cap_out = cap_in + 32 * kMaxCols;
  if ((size_t)(cap_in + cap_out + 64) * 4 > 150 * 1024) return false;
  const int64_t nsub = (rows + kSub - 1) / kSub;
  const uint32_t dpat = 0x01010101u * (ws ? 0u : (uint32_t)delim[0]);
  Buf mx = dev_alloc(4 * sizeof(int), s);
  int* hmx = (int*)pinned_scratch(4 * sizeof(int));

  // ---- second generation: runs of sub-tiles per wave (rows up to 93 bytes, 64-row spans up to 6 KB)
  if (cap_in <= cstile::kPfBytes && !getenv("CS_SPLIT_OLD_EMIT")) {
    // The run decomposition is a function of the row count alone (not of the emit kernel's
    // residency, which depends on what the measure pass finds): emit needs no co-residency.
    int dev = 0, cus = 0;
    CS_HIP(hipGetDevice(&dev));
    CS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int64_t runs = std::min<int64_t>(nsub, (int64_t)cus * 16);
    const int64_t per = (nsub + runs - 1) / runs;
    runs = (nsub + per - 1) / per;
    const int segs_per_run = (int)std::min<int64_t>(4, per);
    const int64_t seg = (per + segs_per_run - 1) / segs_per_run;
    const int64_t nseg = runs * segs_per_run;
    Buf colsum = dev_alloc(sizeof(int32_t) * nseg * kMaxCols, s);
    CS_HIP(hipMemsetAsync(colsum->p, 0, sizeof(int32_t) * nseg * kMaxCols, s));
    CS_HIP(hipMemsetAsync(mx->p, 0, 4 * sizeof(int), s));
    Measure2Args ma{view_of(col), dpat, d64, dlen, tokens, cap_in, nsub, per, seg, nseg, segs_per_run, reverse ? 1 : 0, ptr<int32_t>(colsum), ptr<int>(mx)};
    {
      ProfScope ps("k_split_measure", s);
      const unsigned g = (unsigned)((nseg + 3) / 4);
      const size_t lds = (size_t)(cap_in + 32) * 4;
      if (mode == 1) hipLaunchKernelGGL(k_split_measure2<1>, dim3(g), dim3(256), lds, s, ma);
      else if (mode == 2) hipLaunchKernelGGL(k_split_measure2<2>, dim3(g), dim3(256), lds, s, ma);
      else hipLaunchKernelGGL(k_split_measure2<0>, dim3(g), dim3(256), lds, s, ma);
    }
    CS_HIP(hipGetLastError());
    CS_HIP(hipMemcpyAsync(hmx, mx->p, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    const int ncols = hmx[0], bound = hmx[1], longest_row = hmx[2];
    if (ncols == 0 || ncols > kMaxCols) return false;  // all-null column / too many columns: generic path
    const bool emit2_ok = !hmx[3] && longest_row + 3 <= 96;
    const int cap_col = (bound + 64 + 15) & ~15;
    const size_t lds2 = (size_t)(cap_in + 32 + 2 * cap_col + (ws ? 2 : 1) * ncols * 64) * 4;
    if (emit2_ok && lds2 <= 150 * 1024) {
      // per column: position of every segment in the column's chars buffer
      Buf base = dev_alloc(sizeof(int64_t) * (nseg + 1) * ncols, s);
      std::vector<int64_t> totals(ncols);
      offsets_from_lengths_segmented(ptr<int32_t>(colsum), nseg, ncols, ptr<int64_t>(base), totals.data(), s);
      bool off32 = !getenv("CS_SPLIT_OFF64");
      for (int k = 0; k < ncols; ++k) off32 = off32 && totals[k] < ((int64_t)1 << 31);
      std::vector<ColOut2> outs(ncols);
      for (int k = 0; k < ncols; ++k) {
        auto c = std::make_unique<cs_column>();
        c->rows = rows;
        c->nbytes = totals[k];
        c->chars = dev_alloc((size_t)totals[k], s);
        if (off32) c->offsets32 = dev_alloc(sizeof(int32_t) * (rows + 1), s);
        else c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
        c->validity = dev_alloc(validity_bytes(rows), s);
        outs[k] = ColOut2{ptr<uint8_t>(c->chars), off32 ? c->offsets32->p : c->offsets->p, ptr<uint8_t>(c->validity),
                          ptr<const int64_t>(base) + (int64_t)k * (nseg + 1)};
        cols.push_back(std::move(c));
      }
      Buf d_outs = dev_alloc(sizeof(ColOut2) * ncols, s);
      CS_HIP(hipMemcpyAsync(d_outs->p, outs.data(), sizeof(ColOut2) * ncols, hipMemcpyHostToDevice, s));
      Emit2Args e2{view_of(col), dpat, d64, dlen, tokens, cap_in, cap_col, ncols, nsub, per, segs_per_run, ptr<const ColOut2>(d_outs), nullptr,
                   reverse ? 1 : 0, getenv("CS_SPLIT_DEBUG") ? atoi(getenv("CS_SPLIT_DEBUG")) : 0};
#if defined(CS_PHASE_PROF)
      Buf profbuf = dev_alloc(64, s);
      CS_HIP(hipMemsetAsync(profbuf->p, 0, 64, s));
      e2.prof = ptr<unsigned long long>(profbuf);
#endif
      typedef void (*EmitKernel)(Emit2Args);
      static const EmitKernel kerns[2][3] = {{k_split_emit2<0, false>, k_split_emit2<1, false>, k_split_emit2<2, false>},
                                             {k_split_emit2<0, true>, k_split_emit2<1, true>, k_split_emit2<2, true>}};
      const EmitKernel kern = kerns[off32 ? 1 : 0][mode];
      if (lds2 > 48 * 1024) CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      const unsigned g2 = (unsigned)((runs + 3) / 4);
      {
        ProfScope ps("k_split_emit", s);
        hipLaunchKernelGGL(kern, dim3(g2), dim3(256), lds2, s, e2);
      }
      CS_HIP(hipGetLastError());
      CS_HIP(hipStreamSynchronize(s));  // `outs` / `base` lifetime
#if defined(CS_PHASE_PROF)
      {
        unsigned long long ph[6];
        CS_HIP(hipMemcpy(ph, e2.prof, sizeof(ph), hipMemcpyDeviceToHost));
        const double it = (double)nsub;
        fprintf(stderr, "emit2 cycles/wave-iteration: stage %.0f masks %.0f col-walk+scan+offsets %.0f col-assemble %.0f col-flush %.0f tail %.0f | grid %u lds %zu cap_col %d\n",
                ph[0] / it, ph[1] / it, ph[2] / it, ph[3] / it, ph[4] / it, ph[5] / it, g2, lds2, cap_col);
      }
#endif
      return true;
    }
    if (mode != 0 || reverse) return false;  // whitespace, multi-byte delimiters and rsplit exist in the masked kernels only
  } else if (mode != 0 || reverse) {
    return false;
  }

  // ---- first generation (one-byte delimiter; rows beyond 93 bytes, wide tiles): one sub-tile per wave
  const unsigned grid = (unsigned)((nsub + 3) / 4);
  Buf colsum = dev_alloc(sizeof(int32_t) * nsub * kMaxCols, s);
  CS_HIP(hipMemsetAsync(colsum->p, 0, sizeof(int32_t) * nsub * kMaxCols, s));  // columns a sub-tile never reaches
  CS_HIP(hipMemsetAsync(mx->p, 0, 4 * sizeof(int), s));
  MeasureArgs ma{view_of(col), dpat, d64, dlen, tokens, cap_in, nsub, ptr<int32_t>(colsum), ptr<int>(mx)};
  {
    ProfScope ps("k_split_measure", s);
    hipLaunchKernelGGL(k_split_measure<0>, dim3(grid), dim3(256), (size_t)(cap_in + 32) * 4, s, ma);
  }
  CS_HIP(hipGetLastError());
  CS_HIP(hipMemcpyAsync(hmx, mx->p, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  const int ncols = hmx[0];
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
