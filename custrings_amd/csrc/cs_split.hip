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
//   pass 2  k_split_emit4: walks the tokens again; each wave owns a contiguous run
//           of sub-tiles and carries its position in every column along; column
//           k's tokens of the 64 rows are contiguous in column k's chars buffer,
//           so they are assembled in LDS and flushed with 16-byte stores; offsets
//           (one wave scan per column; int32 when every column stays below 2 GiB,
//           which halves the bytes written per output row) and validity words
//           (one ballot per column) are written coalesced.  All columns come out
//           of this single pass.
//   (k_split_measure / k_split_emit: the first tile generation, one sub-tile per
//   wave with per-sub-tile column sums; kept for rows beyond 188 bytes, 64-row spans
//   beyond 8 KB and sub-tiles of fewer rows.  The second and third emit generations
//   -- k_split_emit2 / _emit3, reachable only through measurement switches since the
//   fourth -- left the library in round 6.)
// Rows with more than kMaxCols tokens, multi-byte delimiters and whitespace
// splitting use the generic kernels in cs_ops.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "cs_internal.h"
#include "device_utils.h"
#include "tile_utils.h"
#include "split_parts.h"

using namespace cs;
using namespace csdev;

namespace cs {
bool split_fast(const cs_column* col, const unsigned char* delim, int dlen, int tokens, hipStream_t s,
                std::vector<std::unique_ptr<cs_column>>& cols, bool reverse = false);
int split_single(const cs_column* col, const unsigned char* delim, int dlen, int tokens, hipStream_t s,
                 std::vector<std::unique_ptr<cs_column>>& cols, bool reverse, int64_t span, bool plain_walk);
}

namespace {

struct MeasureArgs {
  ColView in;
  uint32_t dpat;
  unsigned long long d64;  // the delimiter's bytes, first byte lowest (multi-byte delimiters)
  int dlen;
  int tokens, cap;
  long long nsub;
  int rows_per_sub;  // 64, 32, 16 or 8
  int32_t* colsum;  // [kMaxColsWide][nsub]
  int* max_count;   // [0] most tokens in a row, [1] most bytes one column receives from one sub-tile, [2] longest row,
                    // [3] set when a sub-tile needs the generic kernels (whitespace mode: a row beyond the 96-bit masks)
};
// MODE 0: one-byte delimiter, 1: whitespace, 2: delimiter of 2..8 ASCII bytes
template <int MODE>
__global__ void __launch_bounds__(256) k_split_measure(MeasureArgs a) {
  constexpr bool WS = MODE == 1, MULTI = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap + 32);
  const long long sub = (long long)blockIdx.x * 4 + wv;
  if (sub >= a.nsub) return;
  SubTile t = load_subtile(a.in, sub, lds_in, lane, a.rows_per_sub, a.cap);
  // (an oversize sub-tile: the token walk reads the rows from memory through the same aligned words)
  const uint8_t* rows_at = t.oversize ? a.in.chars + (t.g0 - t.lead) : lds_in;
  TokensT<false, WS, MULTI> tk(rows_at, t.lead + t.rbeg, t.n, t.live, a.dpat, a.tokens, a.d64, a.dlen);
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
    if (k < kMaxColsWide) {
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
// PLAIN (a one-byte delimiter, no split limit): the sentinel walk of k_split_emit4 on W mask words (split_parts.h: RowBits --
// W = 3: rows of at most 92 bytes, a register per column for 32 columns; W = 6: rows up to 188 bytes, 64 columns).
template <int MODE, bool PLAIN = false, int W = 3>
__global__ void __launch_bounds__(256, (PLAIN && W > 3) ? 4 : 8) k_split_measure2(Measure2Args a) {  // (8 waves per SIMD: 2.05 -> 1.94 ms)
  constexpr bool WS = MODE == 1, MULTI = MODE == 2;
  constexpr int MAXC = (PLAIN && W > 3) ? kMaxColsLong : kMaxCols;
  static_assert(!PLAIN || MODE == 0, "the sentinel walk is the one-byte delimiter's");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap + 32);
  const long long sid = (long long)blockIdx.x * 4 + wv;
  if (sid >= a.nseg) return;
  const long long run = sid / a.segs_per_run;
  const long long t0 = run * a.per + (sid - run * a.segs_per_run) * a.seg;
  const long long t1 = min(min(t0 + a.seg, (run + 1) * a.per), a.nsub);
  int acc[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) acc[k] = 0;
  int most = 0, widest = 0, longest = 0;
  bool generic = false;
  for (long long sub = t0; sub < t1; ++sub) {
    SubTile t = load_subtile(a.in, sub, lds_in, lane);
    int count = 0, rowmax = 0;
    bool any_more = true;
    if constexpr (PLAIN) {
      RowBits<W> rb(lds_in, t.lead + t.rbeg, t.n, t.live, a.dpat);
      count = rb.count();
      int prev = rb.sa - 1;  // mask position of the delimiter in front of the current token
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if (any_more) {
          const bool has = rb.any();
          any_more = __any(has);
          const int q = rb.take_lowest();
          const int len = has ? q - prev - 1 : 0;
          prev = q;
          acc[k] += len;
          rowmax = max(rowmax, len);
        }
      }
      // (rows with more than MAXC tokens: `count` says so, the host takes the generic path)
    } else {
      TokensT<false, WS, MULTI> tk(lds_in, t.lead + t.rbeg, t.n, t.live, a.dpat, a.tokens, a.d64, a.dlen, a.reverse != 0);
      if ((WS || MULTI || a.reverse) && !tk.masked) {  // (wave-uniform)
        generic = true;
        break;
      }
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
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
  for (int k = 0; k < MAXC; ++k) {
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
  void* offsets;  // int64, or int32 when every column stays below 2 GiB (OFF32)
  uint8_t* validity;
  const int64_t* base;  // base[sub] = bytes of this column before sub-tile `sub` (nsub + 1 entries)
};
struct EmitArgs {
  ColView in;
  uint32_t dpat;
  int rows_per_sub;
  int tokens, cap_in, cap_out, ncols;
  long long nsub;
  const ColOut* cols;
};
template <bool OFF32>
__global__ void __launch_bounds__(256) k_split_emit(EmitArgs a) {
  using Off = typename std::conditional<OFF32, int32_t, int64_t>::type;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap_in + a.cap_out + 64);
  uint8_t* lds_out = lds_in + a.cap_in + 32;
  const long long sub = (long long)blockIdx.x * 4 + wv;
  if (sub >= a.nsub) return;
  SubTile t = load_subtile(a.in, sub, lds_in, lane, a.rows_per_sub, a.cap_in);
  // lane k holds column k's destination for this sub-tile
  uint8_t* my_chars = nullptr;
  Off* my_off = nullptr;
  uint8_t* my_valid = nullptr;
  long long my_base = 0;
  int my_sum = 0, my_lead = 0;
  if (lane < a.ncols) {
    const ColOut c = a.cols[lane];
    my_chars = c.chars;
    my_off = static_cast<Off*>(c.offsets);
    my_valid = c.validity;
    my_base = c.base[sub];
    my_sum = (int)(c.base[sub + 1] - my_base);
    my_lead = (int)((uintptr_t)(my_chars + my_base) & 15);
  }
  // LDS regions: column k's bytes start at region_k + lead_k so that 16-byte chunks
  // of the region line up with 16-byte chunks of the destination
  const int padded = lane < a.ncols ? ((my_lead + my_sum + 15) & ~15) : 0;
  const int region = wave_inclusive_scan(padded) - padded;

  const uint8_t* rows_at = t.oversize ? a.in.chars + (t.g0 - t.lead) : lds_in;
  Tokens tk(rows_at, t.lead + t.rbeg, t.n, t.live, a.dpat, a.tokens);
  const bool last_tile = t.r0 + t.nrows == a.in.rows;
  for (int k = 0; k < a.ncols; ++k) {
    int lo = 0, hi = 0;
    const bool has = tk.next(lo, hi);
    const int len = has ? hi - lo : 0;
    const int incl = wave_inclusive_scan(len);
    const int pre = incl - len;
    const long long cbase = rl64(my_base, k);
    Off* coff = reinterpret_cast<Off*>(rl64((long long)(uintptr_t)my_off, k));
    uint8_t* cvalid = reinterpret_cast<uint8_t*>(rl64((long long)(uintptr_t)my_valid, k));
    const int cstart = rl(region, k) + rl(my_lead, k);
    if (lane < t.nrows) coff[t.r0 + lane] = (Off)(cbase + pre);
    if (last_tile && lane == t.nrows - 1) coff[a.in.rows] = (Off)(cbase + incl);
    const unsigned long long vmask = __ballot(has);
    if (a.rows_per_sub == kSub) {
      if (lane == 0) *reinterpret_cast<unsigned long long*>(cvalid + sub * 8) = vmask;
    } else if (lane < a.rows_per_sub / 8) {  // (32 / 16 / 8 rows: four / two / one validity bytes)
      cvalid[sub * (a.rows_per_sub / 8) + lane] = (uint8_t)(vmask >> (8 * lane));
    }
    if (t.oversize) {
      // straight to the column's chars: a short token by its lane, a long one (the rest of a long row) by the whole wave
      uint8_t* cchars = reinterpret_cast<uint8_t*>(rl64((long long)(uintptr_t)my_chars, k)) + cbase;
      const uint8_t* src = a.in.chars + (t.g0 + t.rbeg + lo);
      if (has && len <= 256)
        for (int i = 0; i < len; ++i) cchars[pre + i] = src[i];
      for (unsigned long long m = __ballot(has && len > 256); m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        const long long sp = rl64((long long)(t.g0 + t.rbeg + lo), l);
        const int dp = rl(pre, l), L = rl(len, l);
        for (int i = lane; i < L; i += 64) cchars[dp + i] = a.in.chars[sp + i];
      }
    } else if (has) {
      cstile::lds_copy_short(lds_out, cstart + pre, lds_in, t.lead + t.rbeg + lo, len);
    }
  }
  if (t.oversize) return;  // (nothing assembled in LDS)
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
// ---- emit, third generation: all columns assembled side by side, no fence in the column loop ------------
// Same inputs and outputs as k_split_emit2 (runs of consecutive sub-tiles per wave, column positions from the
// measure pass).  What changed is the shape of the column loop, whose twenty rounds of "walk, scan, assemble,
// fence, flush head bytes / chunks / tail bytes, fence" were the kernel's latency chain, and the way bytes reach LDS:
//   * every column gets its own REGION of one LDS out tile (regions back to back, each starting at a 16-byte
//     boundary that corresponds to a 16-byte boundary of the column's chars); a round of the column loop is one token
//     walk step, one wave scan, the offsets store and the tokens' OR into the (zeroed) region with ALIGNED dword
//     accesses -- lds_or16u: the second generation's exact-size stores at any alignment were replayed a lane at a time;
//   * the bytes of a column that do not fill a 16-byte chunk are CARRIED to the wave's next sub-tile (a register
//     quad in the column's lane; the wave's sub-tiles are consecutive, so they continue where these left off): the
//     chars leave as whole, aligned 16-byte chunks only -- a run's first and last chunk, shared with the
//     neighbouring waves, are the exception and go out bytewise once per run and column;
//   * a region LEAVES DURING THE NEXT COLUMN'S ROUND: its chunks are read at the top of that round (one aligned
//     16-byte read per lane), stored to the column's chars at the bottom and zeroed again, so the LDS round trip
//     hides behind the round's own work and no pass over the out tile follows the column loop;
//   * 16 waves per CU (128 registers, 10 KB of LDS per wave): the kernel is bound by its vector instruction count
//     (a wave64 integer instruction holds its SIMD for four cycles) and by the dependent chain of a round, not by
//     HBM or LDS bandwidth, so resident waves are what fills the SIMDs.
struct Emit3Args {
  Emit2Args e;
  int cap_in;     // bytes of the in tile (the largest 64-row span, the lead of its first 16-byte chunk, read-ahead slack)
  int cap_out;    // bytes of the out tile (regions of all columns)
};
#ifndef CS_EMIT3_WAVES
#define CS_EMIT3_WAVES 4
#endif
#ifndef CS_EMIT_THREADS
#define CS_EMIT_THREADS 128  // (192 -- three waves a workgroup, 15 waves a CU instead of 14 -- measured the same: 5.47 / 5.65 against 5.42 / 5.64 ms, A / B on one box)
#endif
constexpr int kEmit3Threads = CS_EMIT_THREADS;  // two waves per workgroup: the LDS of a CU is shared out in finer grains
constexpr int kEmit4TableBytes = 1024 + 128 + 128;  // the flush tables of k_split_emit4 (in the in tile once the column loop is over)
constexpr int kEmit4MaxOut = 16 * 1024;             // bytes of its out tile (32 words of chunk bits)
// PLAIN: a one-byte delimiter without a split limit on rows of at most 92 bytes -- the token walk then runs on three
// mask words that also hold a sentinel bit behind the row's last byte (every token ends at a set bit, the walk is over
// when the mask is empty) instead of the general TokensT::next.
// (measurement builds, -DCS_EMIT3_EXP: CS_SPLIT_DEBUG bits switch phases of the kernel off -- 1 no offsets stores, 2 no chars
// stores, 4 no token assembly, 8 no column loop, 16 the token read at an aligned address (wrong bytes: what the unaligned
// read costs), 32 no pending-region traffic, 64 no wave scan)
#if defined(CS_EMIT3_EXP)
#define EXP(bit) (a.debug & (bit))
#else
#define EXP(bit) false
#endif

// ---- emit, fourth generation: two columns a round, no per-round flush ------------------------------------
// Same inputs, outputs and LDS layout as k_split_emit3 (runs of consecutive sub-tiles per wave, column positions from
// the measure pass, regions side by side in one out tile, partial 16-byte chunks carried in the column's lane).
// What changed (round 4; measured with the CS_EMIT3_EXP switches: the "pending region" traffic of a round -- chunk
// reads, address arithmetic through readlanes, carry, zeroing, the head and long-region branches -- cost 1.5 of the
// kernel's 6.5 ms, the token bytes themselves 0.5): a wave's in-order instruction stream is what bounds the kernel,
// so the column loop does only what needs the rows' lanes, and everything per COLUMN runs once per sub-tile with the
// columns in the lanes:
//   * a round takes TWO columns: two token-walk steps, ONE wave scan over both lengths packed 16 + 16 bits (a
//     sub-tile adds at most 64 x 96 bytes to a column) with the columns' state read out of their lanes in the scan's
//     wait states, offsets stores (scalar base + lane offset), both tokens read before either is OR-ed into its region
//     (no predication: a lane without a token ORs nothing at a valid address).  It leaves the column's byte count, region
//     start and validity word in the column's lane (v_writelane) and nothing else;
//   * after the loop the column lanes -- all at once -- OR their carried bytes in front of their regions, read the
//     new carry (the partial chunk behind the last whole one), advance their position and write a 16-byte flush entry
//     per non-empty region into the by then dead in tile, next to a bitmap of region starts;
//   * one flush pass over the out tile: lane i takes chunks i, i + 64, ...; the region a chunk belongs to is the
//     number of start bits at or below it (word prefix counts + one popcount), the entry gives the global address;
//     whole chunks leave with 16-byte stores and are zeroed again, a region's partial last chunk is only zeroed.
//   * rows beyond the sub-tile's last (the column's last sub-tile only) write the FINAL offset entry: their prefix
//     is the sub-tile's total, so no store in the loop is predicated on the row count.
// A column's state is its position alone: the chars buffers are 16-byte aligned (checked by the host), so the chunk the
// column ends in and the bytes in front of the end follow from the position's low four bits.

// the packed scan of a pair round; the six v_readlane that fetch the two columns' offsets pointers and positions stand
// in the wait states a DPP read of a freshly written register needs (they replace s_nop)
__device__ __forceinline__ int scan_pair(int v, int k, uint32_t off_lo, uint32_t off_hi, uint32_t pos, uint32_t& a_lo, uint32_t& a_hi, uint32_t& a_pos,
                                         uint32_t& b_lo, uint32_t& b_hi, uint32_t& b_pos) {
  const int k1 = k + 1;
  // (the lane selects go through m0, written by s_mov: a v_readlane whose lane-select SGPR was written by a VALU instruction
  // -- should a compiler ever materialise k through v_readfirstlane -- needs four wait states that the hazard recogniser,
  // which does not look into inline assembly, would not insert; m0 <- SALU has no such hazard.  The second s_mov takes the
  // slot of an s_nop.)
  asm volatile(
      "s_mov_b32 m0, %10\n\t"
      "v_readlane_b32 %1, %7, m0\n\t"
      "v_readlane_b32 %2, %8, m0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_readlane_b32 %3, %9, m0\n\t"
      "s_mov_b32 m0, %11\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_readlane_b32 %4, %7, m0\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_readlane_b32 %5, %8, m0\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_readlane_b32 %6, %9, m0\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v), "=&s"(a_lo), "=&s"(a_hi), "=&s"(a_pos), "=&s"(b_lo), "=&s"(b_hi), "=&s"(b_pos)
      : "v"(off_lo), "v"(off_hi), "v"(pos), "s"(k), "s"(k1)
      : "m0");
  return v;
}
// what a round leaves in column k's lane: validity word, byte count | region start << 16
__device__ __forceinline__ void leave_in_lane(int k, unsigned long long valid, int packed, uint32_t& vm_lo, uint32_t& vm_hi, uint32_t& t_pack) {
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "v_writelane_b32 %0, %4, m0\n\t"
      "v_writelane_b32 %1, %5, m0\n\t"
      "v_writelane_b32 %2, %6, m0"
      : "+v"(vm_lo), "+v"(vm_hi), "+v"(t_pack)
      : "s"(__builtin_amdgcn_readfirstlane(k)), "s"(__builtin_amdgcn_readfirstlane((int)(uint32_t)valid)),
        "s"(__builtin_amdgcn_readfirstlane((int)(uint32_t)(valid >> 32))), "s"(__builtin_amdgcn_readfirstlane(packed))
      : "m0");
}
// offsets store: wave-uniform base in scalar registers, the lane's byte offset in a vector register.
// (s_nop 4: a VMEM instruction that reads a scalar register written by a VALU instruction -- the readfirstlane below, when
// the compiler formed the pointer in vector registers -- needs five wait states, and the compiler's hazard recognizer does
// not look into inline assembly: the measurement build stored to garbage addresses without it)
#if defined(CS_NT_STORES)
#define CS_NT_ASM " nt"
#else
#define CS_NT_ASM ""
#endif
template <class T>
__device__ __forceinline__ T* scalar_ptr(T* p) {  // (a no-op when the compiler already knows the pointer to be wave-uniform)
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)p >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void store_off(int32_t* base, int voff, int32_t v) {
  asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" CS_NT_ASM ::"v"(voff), "v"(v), "s"(scalar_ptr(base)) : "memory");
}
__device__ __forceinline__ void store_off(int64_t* base, int voff, int64_t v) {
  asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" CS_NT_ASM ::"v"(voff), "v"(v), "s"(scalar_ptr(base)) : "memory");
}

// W: mask words of the sentinel walk (PLAIN only: 3 = rows up to 92 bytes, 6 = up to 188); PF: 16-byte prefetch chunks a lane
// (6 = sub-tiles up to 6 KB, 8 = up to 8 KB: 64 rows of 95 bytes on average).  Up to 64 columns (a lane a column).
// (the long-row form's tiles leave room for two or three waves a SIMD anyway: it may keep its registers)
template <int MODE, bool OFF32, bool PLAIN, int W = 3, int PF = cstile::kPfChunks>
__global__ void __launch_bounds__(kEmit3Threads, W > 3 ? 3 : CS_EMIT3_WAVES) k_split_emit4(Emit3Args args) {
  constexpr bool WS = MODE == 1, MULTI = MODE == 2;
  static_assert(!PLAIN || MODE == 0, "the sentinel walk is the one-byte delimiter's");
  typedef typename std::conditional<OFF32, int32_t, int64_t>::type off_t;
  const Emit2Args& a = args.e;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + 288 + (size_t)wv * (16 + args.cap_in + 32 + args.cap_out) + 16;
  uint8_t* lds_out = lds_in + args.cap_in + 32;
  cstile::u32x4* tail = reinterpret_cast<cstile::u32x4*>(smem);
  if (lane <= 16) {
    auto first = [](int k) -> uint32_t { return k >= 4 ? 0xFFFFFFFFu : (k <= 0 ? 0u : (1u << (8 * k)) - 1u); };
    tail[lane] = cstile::u32x4{first(lane), first(lane - 4), first(lane - 8), first(lane - 12)};
  }
  const cstile::u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int i = lane * 16; i < args.cap_out; i += 64 * 16) *reinterpret_cast<cstile::u32x4*>(lds_out + i) = zero4;
  // the flush tables live in the in tile once the column loop is over: 64 entries of 16 bytes, 32 words of region-start
  // bits (a bit per 16-byte chunk of the out tile: up to 16 KB), 32 words of prefix counts -- kEmit4TableBytes in all
  cstile::u32x4* f_entry = reinterpret_cast<cstile::u32x4*>(lds_in);
  uint32_t* f_bits = reinterpret_cast<uint32_t*>(lds_in + 1024);
  uint32_t* f_pfx = f_bits + 32;
  const long long per = a.per;
  const long long run = (long long)blockIdx.x * (kEmit3Threads / 64) + wv;
  long long tile = run * per;
  const long long tile_end = min(a.nsub, tile + per);
  if (tile >= tile_end) return;
  const ColView& in = a.in;
  const int ncols = a.ncols;
  // lane k keeps column k: the running position (the offsets' value; its low four bits = the bytes carried in front of it
  // in its 16-byte chunk -- the previous sub-tile's, or the neighbouring run's while c_head != 0), the carried bytes
  long long c_pos = 0;
  int c_head = 0;
  uint8_t* c_chars = nullptr;
  off_t* c_off = nullptr;
  uint8_t* c_valid = nullptr;
  cstile::u32x4 carry = zero4;
  if (lane < ncols) {
    const ColOut2 c = a.cols[lane];
    c_chars = c.chars;
    c_off = reinterpret_cast<off_t*>(c.offsets);
    c_valid = c.validity;
    c_pos = c.seg_base[run * a.segs_per_run];
    c_head = (int)c_pos & 15;
  }
  cstile::TileOffs cur = cstile::load_tile_offsets(in.offsets, in.rows, tile, lane);
  cstile::TileOffs nxt = cur;
  if (tile + 1 < tile_end) nxt = cstile::load_tile_offsets(in.offsets, in.rows, tile + 1, lane);
  cstile::TileCharsT<PF> pf;
#pragma unroll
  for (int j = 0; j < PF; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
#if defined(CS_PHASE_PROF)
  unsigned long long phase_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long phase_t = __builtin_readcyclecounter();
#endif
  for (;;) {
#if defined(CS_PHASE_PROF)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (measurement: what the staging below would wait for -- the flush stores)
    CS_PHASE_MARK(5);
#endif
    const long long r0 = tile * 64;
    const int nrows = (int)min(64ll, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool live = lane < nrows && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n = live ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const int want = (int)(g1 - g0) + lead;
    cstile::stage_chars(lds_in, want, lane, pf);
    const bool has_next = tile + 1 < tile_end;
    const cstile::TileOffs nn = cstile::load_tile_offsets(in.offsets, in.rows, tile + 2 < tile_end ? tile + 2 : tile_end - 1, lane);
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
    }
    cstile::wave_lds_fence();
    CS_PHASE_MARK(0);

    // the walker: PLAIN -- delimiter bits and a sentinel behind the row on W words (RowBits); else the general token walker
    auto make_walker = [&]() {
      if constexpr (PLAIN) return RowBits<W>(lds_in, lead + rbeg, n, live, a.dpat);
      else return TokensT<true, WS, MULTI>(lds_in, lead + rbeg, n, live, a.dpat, a.tokens, a.d64, a.dlen, a.reverse != 0);
    };
    auto tk = make_walker();
    int tcur = 0;
    auto step = [&](int& lo, int& hi) -> bool {  // the row's next token
      if constexpr (PLAIN) {
        const bool has = tk.any();
        const int q = tk.take_lowest();
        lo = tcur;
        hi = q - tk.sa;
        tcur = hi + 1;
        return has;
      } else {
        return tk.next(lo, hi);
      }
    };
    // a row beyond the sub-tile's last writes the final offset entry (its prefix is the sub-tile's total)
    const int rowoff = min(lane, nrows) * (int)sizeof(off_t);
    const uint8_t* tok_src = lds_in + lead + rbeg;
    // what a round leaves in the column's lane: the validity word; bytes added by this sub-tile (0: the column is not
    // touched below) | where its region begins in the out tile << 16
    uint32_t t_pack = 0, vm_lo = 0, vm_hi = 0;
    int rg = 0;  // where the next region begins (wave-uniform)
    int k = 0;
    const uint32_t coff_lo = (uint32_t)(uintptr_t)c_off, coff_hi = (uint32_t)((uintptr_t)c_off >> 32);
    auto or_token = [&](cstile::u32x4 v, cstile::u32x4 m, int di) {
      const uint32_t a0 = v.x & m.x, a1 = v.y & m.y, a2 = v.z & m.z, a3 = v.w & m.w;
      const unsigned up = (0u - (unsigned)di) & 3u;
      uint32_t* o = reinterpret_cast<uint32_t*>(lds_out + (((di + 3) & ~3) - 4));
      lds_or(o + 0, __builtin_amdgcn_alignbyte(a0, 0u, up));
      lds_or(o + 1, __builtin_amdgcn_alignbyte(a1, a0, up));
      lds_or(o + 2, __builtin_amdgcn_alignbyte(a2, a1, up));
      lds_or(o + 3, __builtin_amdgcn_alignbyte(a3, a2, up));
      lds_or(o + 4, __builtin_amdgcn_alignbyte(0u, a3, up));
    };
    auto long_tokens = [&](int lo, int len, int at) {  // (tokens beyond 16 bytes: the rest, 16 bytes at a time)
      for (int done = 16; __any(done < len); done += 16)
        if (done < len) lds_or16u(lds_out, at + done, tok_src, lo + done, min(len - done, 16), tail);
    };
    auto off_base = [&](uint32_t lo, uint32_t hi) -> off_t* {
      return reinterpret_cast<off_t*>((((unsigned long long)hi << 32) | lo) + (unsigned long long)r0 * sizeof(off_t));
    };
    CS_PHASE_MARK(1);
    // ---- pairs of columns
    for (; k + 1 < (EXP(8) ? 0 : ncols); k += 2) {
      int loA = 0, hiA = 0, loB = 0, hiB = 0;
      const bool hasA = step(loA, hiA);
      const unsigned long long vA = __ballot(hasA);
      if (vA == 0) break;  // no row of the sub-tile reaches column k (nor any behind it)
      const bool hasB = step(loB, hiB);
      const unsigned long long vB = __ballot(hasB);
      const int lenA = hasA ? hiA - loA : 0, lenB = hasB ? hiB - loB : 0;
      // both tokens' first sixteen bytes and tail masks (a lane without a token reads its row's start and masks it all)
#if defined(CS_EMIT_ALIGNED_TOKENS)
      // (measurement, round 5 and again round 6 with non-temporal stores: the token as five aligned dwords and four funnel
      // shifts instead of one 16-byte read at its own address, which the LDS replays a lane at a time)
      auto read_token = [&](int lo) -> cstile::u32x4 {
        const int at = lead + rbeg + lo;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(lds_in + (at & ~3));
        const unsigned sh = (unsigned)at & 3u;
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
        return cstile::u32x4{__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh), __builtin_amdgcn_alignbyte(w3, w2, sh),
                             __builtin_amdgcn_alignbyte(w4, w3, sh)};
      };
      const cstile::u32x4 rA = read_token(hasA ? loA : 0), rB = read_token(hasB ? loB : 0);
#else
      const cstile::lds_u32x4u rA = *reinterpret_cast<const cstile::lds_u32x4u*>(tok_src + (hasA ? loA : 0));
      const cstile::lds_u32x4u rB = *reinterpret_cast<const cstile::lds_u32x4u*>(tok_src + (hasB ? loB : 0));
#endif
      const cstile::u32x4 mA = tail[min(lenA, 16)], mB = tail[min(lenB, 16)];
      uint32_t a_lo, a_hi, a_pos, b_lo, b_hi, b_pos;
      const int incl = scan_pair(lenA | (lenB << 16), k, coff_lo, coff_hi, (uint32_t)c_pos, a_lo, a_hi, a_pos, b_lo, b_hi, b_pos);
      const int tot2 = rl(incl, 63);
      const int preA = (incl & 0xffff) - lenA, preB = (int)((unsigned)incl >> 16) - lenB;
      const int csumA = tot2 & 0xffff, csumB = (int)((unsigned)tot2 >> 16);
      const off_t baseA = OFF32 ? (off_t)(int)a_pos : (off_t)cstile::rl64(c_pos, k);
      const off_t baseB = OFF32 ? (off_t)(int)b_pos : (off_t)cstile::rl64(c_pos, k + 1);
      if (!EXP(1)) store_off(off_base(a_lo, a_hi), rowoff, baseA + (off_t)preA);
      if (!EXP(1)) store_off(off_base(b_lo, b_hi), rowoff, baseB + (off_t)preB);
      const int cphA = (int)a_pos & 15, cphB = (int)b_pos & 15;
      const int rgB = rg + (csumA ? (cphA + csumA + 15) & ~15 : 0);
      leave_in_lane(k, vA, csumA | (rg << 16), vm_lo, vm_hi, t_pack);
      leave_in_lane(k + 1, vB, csumB | (rgB << 16), vm_lo, vm_hi, t_pack);
      // (a lane without a token ORs zeros: at an address of its own -- the same dword from many lanes would serialise)
      const int atA = rg + cphA + preA, atB = rgB + cphB + preB;
      if (!EXP(4)) {
#if defined(CS_EMIT4_PRED)
      if (hasA) or_token(cstile::u32x4{rA.x, rA.y, rA.z, rA.w}, mA, atA);
      if (hasB) or_token(cstile::u32x4{rB.x, rB.y, rB.z, rB.w}, mB, atB);
#else
      or_token(cstile::u32x4{rA.x, rA.y, rA.z, rA.w}, mA, hasA ? atA : lane * 20);
      or_token(cstile::u32x4{rB.x, rB.y, rB.z, rB.w}, mB, hasB ? atB : lane * 20);
#endif
      }
      if (__any(max(lenA, lenB) > 16)) {
        long_tokens(loA, lenA, atA);
        long_tokens(loB, lenB, atB);
      }
      rg = rgB + (csumB ? (cphB + csumB + 15) & ~15 : 0);
    }
    // ---- an odd last column
    if (k + 1 == ncols) {
      int lo = 0, hi = 0;
      const bool has = step(lo, hi);
      const unsigned long long vA = __ballot(has);
      if (vA != 0) {
        const int len = has ? hi - lo : 0;
        const int incl = wave_inclusive_scan_fused(len);
        const int csum = rl(incl, 63), pre = incl - len;
        const off_t base = OFF32 ? (off_t)rl((int)c_pos, k) : (off_t)cstile::rl64(c_pos, k);
        store_off(off_base((uint32_t)rl((int)coff_lo, k), (uint32_t)rl((int)coff_hi, k)), rowoff, base + (off_t)pre);
        const int cph = (int)base & 15;
        leave_in_lane(k, vA, csum | (rg << 16), vm_lo, vm_hi, t_pack);
        const int at = rg + cph + pre;
        if (has) lds_or16u(lds_out, at, tok_src, lo, min(len, 16), tail);
        if (__any(len > 16)) long_tokens(lo, len, at);
        if (csum) rg += (cph + csum + 15) & ~15;
        ++k;
      }
    }
    // ---- columns no row of the sub-tile reaches: null rows at the column's running position
    for (; k < (EXP(1) ? 0 : ncols); ++k) {
      const off_t base = OFF32 ? (off_t)rl((int)c_pos, k) : (off_t)cstile::rl64(c_pos, k);
      store_off(off_base((uint32_t)rl((int)coff_lo, k), (uint32_t)rl((int)coff_hi, k)), rowoff, base);
    }
    CS_PHASE_MARK(2);
    // ---- per column, the columns in the lanes
    const int t_sum = (int)(t_pack & 0xffffu), t_rg = (int)(t_pack >> 16);
    const bool act = lane < ncols && t_sum > 0 && !EXP(64);
    const int c_cph = (int)c_pos & 15;
    const int tot = c_cph + t_sum, nwhole = tot >> 4;
    const unsigned long long actm = __ballot(act);
    if (lane < 32) f_bits[lane] = 0u;  // (the in tile is dead: every token has been copied)
    if (act) {
      // the carried bytes open the region (the tokens were OR-ed in behind them; a run's first chunk keeps zeros in
      // front: those bytes are the neighbouring run's)
      uint32_t* o = reinterpret_cast<uint32_t*>(lds_out + t_rg);
      lds_or(o + 0, carry.x);
      lds_or(o + 1, carry.y);
      lds_or(o + 2, carry.z);
      lds_or(o + 3, carry.w);
      const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(actm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)actm, 0u));
      const unsigned long long ga = (unsigned long long)(uintptr_t)c_chars + (unsigned long long)(c_pos & ~15ll);
      f_entry[rank] = cstile::u32x4{(uint32_t)ga, (uint32_t)(ga >> 32), (uint32_t)(t_rg >> 4) | ((uint32_t)nwhole << 16), (uint32_t)c_head};
      lds_or(f_bits + (t_rg >> 9), 1u << ((t_rg >> 4) & 31));
    }
    cstile::wave_lds_fence();
    if (act) {
      // the bytes behind the last whole chunk are the new carry
      carry = (tot & 15) ? *reinterpret_cast<const cstile::u32x4*>(lds_out + t_rg + 16 * nwhole) : zero4;
      c_pos += t_sum;
    }
    {
      const int cnt = lane < 32 ? __builtin_popcount(f_bits[lane]) : 0;
      const int inc = wave_inclusive_scan_fused(cnt);
      if (lane < 32) f_pfx[lane] = (uint32_t)(inc - cnt);
    }
    cstile::wave_lds_fence();
    CS_PHASE_MARK(3);
    // ---- the flush pass: every 16-byte chunk of the out tile below rg
    // (moving this pass behind the NEXT sub-tile's staging -- tables out of the in tile, entries in registers -- so that the
    // staging does not wait for these stores was built and measured: the 2 k cycles a wave waits there moved into the pass
    // itself, 5.3-5.5 ms either way; profiles/r04)
    const int nchunks = EXP(32) ? 0 : rg >> 4;
    for (int c = lane; c < nchunks; c += 64) {
      const uint32_t word = f_bits[c >> 5];
      const int rank = (int)f_pfx[c >> 5] + __builtin_popcount(word & (0xFFFFFFFFu >> (31 - (c & 31)))) - 1;
      const cstile::u32x4 e = f_entry[rank];
      const cstile::u32x4 v = *reinterpret_cast<const cstile::u32x4*>(lds_out + 16 * c);
      *reinterpret_cast<cstile::u32x4*>(lds_out + 16 * c) = zero4;
      const int rel = c - (int)(e.z & 0xffffu);
      const bool whole = rel < (int)(e.z >> 16);
      uint8_t* ga = reinterpret_cast<uint8_t*>((((unsigned long long)e.y << 32) | e.x) + (unsigned long long)(16 * rel));
      const bool head = whole && rel == 0 && e.w != 0;
      if (whole && !head && !EXP(2)) cstile::gstore16((cstile::gptr<cstile::u32x4>)cstile::as_global(ga), v);
      if (__any(head)) {
        // a run's first whole chunk of a column: its leading bytes belong to the wave in front, bytes head .. 15 go out
        // one by one (once per run and column)
        if (head) {
          const uint32_t d[4] = {v.x, v.y, v.z, v.w};
          for (int j = (int)e.w; j < 16; ++j) cstile::as_global(ga)[j] = (uint8_t)(d[j >> 2] >> (8 * (j & 3)));
        }
      }
    }
    CS_PHASE_MARK(4);
    if (act && nwhole > 0) c_head = 0;  // (the column's first whole chunk of the run has left)
    if (lane < ncols) {
      unsigned long long vm = ((unsigned long long)vm_hi << 32) | vm_lo;
      *cstile::as_global(reinterpret_cast<unsigned long long*>(c_valid + tile * 8)) = vm;
      if (r0 + 64 == in.rows) cstile::as_global(c_off)[in.rows] = (off_t)c_pos;  // (a last sub-tile of exactly 64 rows)
    }
    if (!has_next) break;
    tile += 1;
    nxt = nn;
    cstile::wave_lds_fence();  // (the tables in the in tile are read; the next sub-tile may be staged over them)
  }
#if defined(CS_PHASE_PROF)
  if (lane == 0 && a.prof)
    for (int i = 0; i < 6; ++i) atomicAdd(a.prof + i, phase_acc[i]);
#endif
  // ---- the run's last bytes of every column: what is still carried goes out bytewise
  if (lane < ncols && ((int)c_pos & 15) > c_head) {
    *reinterpret_cast<cstile::u32x4*>(lds_out + 16 * lane) = carry;
    cstile::gptr<uint8_t> d = cstile::as_global(c_chars + (c_pos & ~15ll));
    for (int j = c_head; j < ((int)c_pos & 15); ++j) d[j] = lds_out[16 * lane + j];
  }
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
  if (rows == 0 || cs::cfg("CS_SPLIT_GENERIC")) return false;
  int64_t span = max_span64(col, s);
  int cap_in = (int)((span + 15 + 32 + 127) & ~(int64_t)127);
  int cap_out = cap_in + 32 * kMaxColsWide;
  // (rows of hundreds of bytes: the first-generation kernels on sub-tiles of 32 / 16 / 8 rows -- a one-byte delimiter only)
  int rows_per_sub = kSub;
  bool outliers = false;
  if ((size_t)(cap_in + cap_out + 64) * 4 > 150 * 1024 && mode == 0 && !reverse && !cs::cfg("CS_NO_SMALL_TILES")) {
    for (int r : {32, 16, 8}) {
      const int64_t sp = max_span_rows(col, r, s);
      const int ci = (int)((sp + 15 + 32 + 127) & ~(int64_t)127);
      if ((size_t)(ci + ci + 32 * kMaxColsWide + 64) * 4 <= 150 * 1024) {
        rows_per_sub = r;
        span = sp;
        cap_in = ci;
        cap_out = ci + 32 * kMaxColsWide;
        break;
      }
    }
  }
  // (no sub-tile size fits the largest sub-tile, all but a few 64-row ones fit: the first-generation kernels with buffers
  // for those -- they read the rows of an oversize sub-tile from memory and write its tokens straight to the columns)
  // (also when the largest sub-tile would fit, at one workgroup per CU: a 10 KB row among short ones ran 39 ms that way)
  if (rows_per_sub == kSub && cap_in > 8 * 1024 && mode == 0 && !reverse && !cs::cfg("CS_NO_OUTLIER_TILES") && max_span64(col, s) < ((int64_t)1 << 30) &&
      few_spans64_over(col, 8 * 1024 - 64, s)) {
    rows_per_sub = kSub;
    span = 8 * 1024 - 64;
    cap_in = (int)((span + 15 + 32 + 127) & ~(int64_t)127);
    cap_out = cap_in + 32 * kMaxColsWide;
    outliers = true;
  }
  if ((size_t)(cap_in + cap_out + 64) * 4 > 150 * 1024) return false;
  const int64_t nsub = (rows + kSub - 1) / kSub;
  const uint32_t dpat = 0x01010101u * (ws ? 0u : (uint32_t)delim[0]);
  Buf mx = dev_alloc(4 * sizeof(int), s);
  int* hmx = (int*)pinned_scratch(4 * sizeof(int));

  // ---- the tile kernels proper (k_split_measure2 + k_split_emit4): runs of sub-tiles per wave.  Rows up to 92-93 bytes on
  // 96-bit masks and 64-row spans up to 6 KB; a plain split (one-byte delimiter, no limit) also rows up to 188 bytes on six
  // mask words, spans up to 8 KB and 64 columns -- BASELINE's C5 column, which left these kernels until round 6
  const bool plain_mode = mode == 0 && tokens <= 0 && !reverse && !cs::cfg("CS_SPLIT_GENERIC_WALK");
  const int64_t longest_known = plain_mode ? max_row_bytes(col, s) : 0;  // (column metadata, kept on the immutable column like its largest 64-row span)
  // mask words of the sentinel walk (both passes): every row's sentinel bit inside the mask; 0 = the general token walker
  const int walk_words = !plain_mode ? 0 : (longest_known + 3 <= 95 ? 3 : (longest_known + 3 <= 191 && !cs::cfg("CS_SPLIT_NO_LONG_WALK") ? 6 : 0));
  const bool tiles_ok = cap_in <= cstile::kPfBytes || (walk_words == 6 && cap_in <= 8 * 1024);
  if (rows_per_sub == kSub && !outliers && tiles_ok && !cs::cfg("CS_SPLIT_OLD_EMIT")) {
    // The run decomposition is a function of the row count alone (not of the emit kernel's
    // residency, which depends on what the measure pass finds): emit needs no co-residency.
    int dev = 0, cus = 0;
    CS_HIP(hipGetDevice(&dev));
    CS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int runs_per_cu = 256;  // (short runs share the tail out evenly: 16 / 64 / 128 / 256 runs per CU -> emit 7.06 / 7.05 / 6.77 / 6.61 ms)
    runs_per_cu = std::max(1, cs::cfg_int("CS_EMIT_RUNS_PER_CU", runs_per_cu));  // (measurement)
    int64_t runs = std::min<int64_t>(nsub, (int64_t)cus * runs_per_cu);
    const int64_t per = (nsub + runs - 1) / runs;
    runs = (nsub + per - 1) / per;
    const int segs_per_run = (int)std::min<int64_t>(2, per);
    const int64_t seg = (per + segs_per_run - 1) / segs_per_run;
    const int64_t nseg = runs * segs_per_run;
    const int maxc = walk_words == 6 ? kMaxColsLong : kMaxCols;
    Buf colsum = dev_alloc(sizeof(int32_t) * nseg * maxc, s);
    CS_HIP(hipMemsetAsync(colsum->p, 0, sizeof(int32_t) * nseg * maxc, s));
    CS_HIP(hipMemsetAsync(mx->p, 0, 4 * sizeof(int), s));
    const bool plain_walk = walk_words != 0;
    // the single pass (cs_split1.hip), on request: measured slower than the two passes below (NOTES.md, round 4:
    // 8.0 against 6.7 ms on the 100M-row column); they also take over when it gives up
#if defined(CS_EXPERIMENTS)  // (cs_split1.hip is in the experiments build only: `make exp`)
    if (walk_words != 6 && cs::cfg("CS_SPLIT_SINGLE") && !cs::cfg("CS_SPLIT_OFF64") && split_single(col, delim, dlen, tokens, s, cols, reverse, span, walk_words == 3) == 1) return true;
#endif
    cols.clear();
    Measure2Args ma{view_of(col), dpat, d64, dlen, tokens, cap_in, nsub, per, seg, nseg, segs_per_run, reverse ? 1 : 0, ptr<int32_t>(colsum), ptr<int>(mx)};
    {
      ProfScope ps("k_split_measure", s);
      const unsigned g = (unsigned)((nseg + 3) / 4);
      const size_t lds = (size_t)(cap_in + 32) * 4;
      if (mode == 1) hipLaunchKernelGGL(k_split_measure2<1>, dim3(g), dim3(256), lds, s, ma);
      else if (mode == 2) hipLaunchKernelGGL(k_split_measure2<2>, dim3(g), dim3(256), lds, s, ma);
      else if (walk_words == 6) hipLaunchKernelGGL((k_split_measure2<0, true, 6>), dim3(g), dim3(256), lds, s, ma);
      else if (walk_words == 3) hipLaunchKernelGGL((k_split_measure2<0, true, 3>), dim3(g), dim3(256), lds, s, ma);
      else hipLaunchKernelGGL(k_split_measure2<0>, dim3(g), dim3(256), lds, s, ma);
    }
    CS_HIP(hipGetLastError());
    CS_HIP(hipMemcpyAsync(hmx, mx->p, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    const int ncols = hmx[0], bound = hmx[1], longest_row = hmx[2];
    if (ncols == 0 || ncols > kMaxColsWide) return false;  // all-null column / too many columns: generic path
    // regions of the out tile: every token byte once, up to 15 carried bytes and up to 15 bytes of padding per column, 20 bytes of OR slack
    const int cap_out3 = (int)((span + 31 * ncols + 48 + 15) & ~(int64_t)15);
    // (the in tile also holds the flush tables once the column loop is over)
    const int cap_in3 = std::max((int)((span + 15 + 32 + 15) & ~(int64_t)15), kEmit4TableBytes);
    const size_t lds3 = 288 + (size_t)(16 + cap_in3 + 32 + cap_out3) * (kEmit3Threads / 64);
    const bool emit4_ok = !hmx[3] && ncols <= maxc && longest_row + 3 <= (plain_walk ? 32 * walk_words - 1 : 96) && cap_out3 <= kEmit4MaxOut && lds3 <= 64 * 1024;
    if (emit4_ok) {
      // per column: position of every segment in the column's chars buffer
      Buf base = dev_alloc(sizeof(int64_t) * (nseg + 1) * ncols, s);
      std::vector<int64_t> totals(ncols);
      offsets_from_lengths_segmented(ptr<int32_t>(colsum), nseg, ncols, ptr<int64_t>(base), totals.data(), s);
      bool off32 = !cs::cfg("CS_SPLIT_OFF64");
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
        // column metadata for free (upper bounds: they size staging buffers and pick routes): a column gets at most `bound`
        // bytes from one sub-tile (the measure pass: every row's longest token, summed), no token is longer than its row;
        // tokens cut at ASCII delimiters out of a plain column are plain
        c->max_span64 = bound;
        c->max_row = std::min<int64_t>(longest_row, bound);
        if (col->plain_bytes == 1) c->plain_bytes = 1;
        if (col->high_sample == 0) c->high_sample = 0;
        if (((uintptr_t)ptr<uint8_t>(c->chars) & 15) != 0) fail(CS_ERR_INTERNAL, "split: a column's chars buffer is not 16-byte aligned");  // (a column's state is its position)
        outs[k] = ColOut2{ptr<uint8_t>(c->chars), off32 ? c->offsets32->p : c->offsets->p, ptr<uint8_t>(c->validity),
                          ptr<const int64_t>(base) + (int64_t)k * (nseg + 1)};
        cols.push_back(std::move(c));
      }
      Buf d_outs = dev_alloc(sizeof(ColOut2) * ncols, s);
      CS_HIP(hipMemcpyAsync(d_outs->p, outs.data(), sizeof(ColOut2) * ncols, hipMemcpyHostToDevice, s));
      Emit2Args e2{view_of(col), dpat, d64, dlen, tokens, cap_in, 0, ncols, nsub, per, segs_per_run, ptr<const ColOut2>(d_outs), nullptr,
                   reverse ? 1 : 0, cs::cfg_int("CS_SPLIT_DEBUG", 0)};
#if defined(CS_PHASE_PROF)
      Buf profbuf = dev_alloc(64, s);
      CS_HIP(hipMemsetAsync(profbuf->p, 0, 64, s));
      e2.prof = ptr<unsigned long long>(profbuf);
#endif
      typedef void (*Emit3Kernel)(Emit3Args);
      static const Emit3Kernel kerns4[2][3] = {{k_split_emit4<0, false, false>, k_split_emit4<1, false, false>, k_split_emit4<2, false, false>},
                                               {k_split_emit4<0, true, false>, k_split_emit4<1, true, false>, k_split_emit4<2, true, false>}};
      const Emit3Kernel kern3 = walk_words == 6   ? (off32 ? k_split_emit4<0, true, true, 6, 8> : k_split_emit4<0, false, true, 6, 8>)
                                : walk_words == 3 ? (off32 ? k_split_emit4<0, true, true> : k_split_emit4<0, false, true>)
                                                  : kerns4[off32 ? 1 : 0][mode];
      if (lds3 > 48 * 1024) CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      Emit3Args e3{e2, cap_in3, cap_out3};
      constexpr int wpg = kEmit3Threads / 64;
      const unsigned g2 = (unsigned)((runs + wpg - 1) / wpg);
      {
        ProfScope ps("k_split_emit", s);
        hipLaunchKernelGGL(kern3, dim3(g2), dim3(kEmit3Threads), lds3, s, e3);
      }
      note_route(walk_words == 6 ? "split-tiles-188" : (walk_words == 3 ? "split-tiles-92" : "split-tiles"));
      CS_HIP(hipGetLastError());
      CS_HIP(hipStreamSynchronize(s));  // `outs` / `base` lifetime
#if defined(CS_PHASE_PROF)
      {
        unsigned long long ph[6];
        CS_HIP(hipMemcpy(ph, e2.prof, sizeof(ph), hipMemcpyDeviceToHost));
        const double it = (double)nsub;
        fprintf(stderr, "emit cycles/wave-iteration (emit4: stage, masks, column loop, column lanes, flush, store drain): %.0f %.0f %.0f %.0f %.0f %.0f | grid %u lds %zu\n",
                ph[0] / it, ph[1] / it, ph[2] / it, ph[3] / it, ph[4] / it, ph[5] / it, g2, lds3);
      }
#endif
      return true;
    }
    if (mode != 0 || reverse) return false;  // whitespace, multi-byte delimiters and rsplit exist in the masked kernels only
  } else if (mode != 0 || reverse) {
    return false;
  }

  // ---- first generation (one-byte delimiter; rows beyond 188 bytes, wide tiles, more than 32 / 64 columns): one sub-tile per wave
  const int64_t nsub1 = (rows + rows_per_sub - 1) / rows_per_sub;  // (sub-tiles of rows_per_sub rows)
  const unsigned grid = (unsigned)((nsub1 + 3) / 4);
  Buf colsum = dev_alloc(sizeof(int32_t) * nsub1 * kMaxColsWide, s);
  CS_HIP(hipMemsetAsync(colsum->p, 0, sizeof(int32_t) * nsub1 * kMaxColsWide, s));  // columns a sub-tile never reaches
  CS_HIP(hipMemsetAsync(mx->p, 0, 4 * sizeof(int), s));
  MeasureArgs ma{view_of(col), dpat, d64, dlen, tokens, cap_in, nsub1, rows_per_sub, ptr<int32_t>(colsum), ptr<int>(mx)};
  {
    ProfScope ps("k_split_measure", s);
    hipLaunchKernelGGL(k_split_measure<0>, dim3(grid), dim3(256), (size_t)(cap_in + 32) * 4, s, ma);
  }
  CS_HIP(hipGetLastError());
  CS_HIP(hipMemcpyAsync(hmx, mx->p, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  const int ncols = hmx[0], widest1 = hmx[1], longest1 = hmx[2];  // (copied: the scratch is re-used by the scan below)
  if (ncols == 0 || ncols > kMaxColsWide) return false;  // all-null column / too many columns: generic path

  // per column: position of every sub-tile in the column's chars buffer
  Buf base = dev_alloc(sizeof(int64_t) * (nsub1 + 1) * ncols, s);
  std::vector<int64_t> totals(ncols);
  offsets_from_lengths_segmented(ptr<int32_t>(colsum), nsub1, ncols, ptr<int64_t>(base), totals.data(), s);

  // int32 offsets (as the later generations write them) when every column stays below 2 GiB: half the offset bytes
  bool off32 = !cs::cfg("CS_SPLIT_OFF64");
  for (int k = 0; k < ncols; ++k) off32 = off32 && totals[k] < ((int64_t)1 << 31);
  std::vector<ColOut> outs(ncols);
  for (int k = 0; k < ncols; ++k) {
    auto c = std::make_unique<cs_column>();
    c->rows = rows;
    c->nbytes = totals[k];
    c->chars = dev_alloc((size_t)totals[k], s);
    if (off32) c->offsets32 = dev_alloc(sizeof(int32_t) * (rows + 1), s);
    else c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    c->validity = dev_alloc(validity_bytes(rows), s);
    // (upper bounds: hmx[1] is the most bytes any column gets from one sub-tile of rows_per_sub rows)
    c->max_span64 = (int64_t)widest1 * (kSub / rows_per_sub);
    c->max_row = std::min<int64_t>(longest1, widest1);
    if (col->plain_bytes == 1) c->plain_bytes = 1;
    if (col->high_sample == 0) c->high_sample = 0;
    outs[k] = ColOut{ptr<uint8_t>(c->chars), off32 ? c->offsets32->p : c->offsets->p, ptr<uint8_t>(c->validity),
                     ptr<const int64_t>(base) + (int64_t)k * (nsub1 + 1)};
    cols.push_back(std::move(c));
  }
  Buf d_outs = dev_alloc(sizeof(ColOut) * ncols, s);
  CS_HIP(hipMemcpyAsync(d_outs->p, outs.data(), sizeof(ColOut) * ncols, hipMemcpyHostToDevice, s));
  EmitArgs ea{view_of(col), dpat, rows_per_sub, tokens, cap_in, cap_out, ncols, nsub1, ptr<const ColOut>(d_outs)};
  const size_t lds = (size_t)(cap_in + cap_out + 64) * 4;
  auto kern = off32 ? &k_split_emit<true> : &k_split_emit<false>;
  if (lds > 48 * 1024) CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  {
    ProfScope ps("k_split_emit", s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, ea);
  }
  CS_HIP(hipGetLastError());
  CS_HIP(hipStreamSynchronize(s));  // `outs` / `base` lifetime
  note_route("split-first-generation");
  return true;
}

}  // namespace cs
