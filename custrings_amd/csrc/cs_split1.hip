// split() in ONE pass over the chars buffer (NVStrings::split / rsplit, split.cu:734-822 and 959-1148 are the multi-pass
// shape this replaces: count tokens per row -> sizes per column -> allocate -> write).
//
// The two-pass kernels of cs_split.hip read the input twice: k_split_measure2 for the column count and every run's
// position in every column, k_split_emit4 for the columns themselves.  Here the positions come out of the emit pass:
//   * k_split_sample looks at a strided sample of the sub-tiles (all of them when there are few): most tokens in a
//     row, per column the mean and variance of what a sub-tile contributes.  The host provisions `ncap` columns and
//     a chars buffer per column of (estimate + 8 sigma): the only thing the numbers are used for.  A column that
//     outgrows its buffer, a row with more tokens than `ncap` columns, a column beyond 2 GiB or a wait without progress
//     raise the error word, and the host repeats the call on the two-pass kernels (counted: cs_fallback_count).
//   * k_split_emit5: persistent waves (CU-sized workgroups) take sub-tiles (64 rows) in a static round-robin sequence.
//     A sub-tile's walk + scan is emit4's -- token walk, one packed wave scan per pair of columns -- but it needs no
//     position: the rows' offsets stay RELATIVE to the sub-tile (two 16-bit values a register, a register per pair of
//     columns), the token starts / lengths stay in registers too, and the sub-tile's bytes per column are PUBLISHED at
//     once (two columns a word).  Scanner waves (a pair of columns each, four helper waves a pair handing the running
//     sums on through an LDS ring, on CUs of their own) turn the published words into exclusive prefixes in tile order.
//     Behind the publish the PREVIOUS sub-tile is finished -- prefix + relative offsets -> the offsets stores, its
//     regions of the LDS out tile -> the columns' chars with 16-byte stores at whatever alignment the position has (a
//     region's last bytes: 8 / 4 / 2 / 1-byte stores from the column's lane) -- and then this sub-tile's tokens are
//     OR-ed into their regions, which begin at 16-byte boundaries of the out tile: a sub-tile has a whole iteration
//     between its publish and the moment its prefix is needed.
//   * the column count is a running maximum (one word, atomicMax): a sub-tile writes offsets and validity for the
//     columns known when it ran and records how many those were; k_split_fixup writes the null rows of columns that
//     appeared later (only launched when the count grew beyond what the sample saw).
// Output identical to the two-pass kernels': int32 offsets, validity words, chars (tests/test_gpu_split_single.py).
// Measured on the 100M-row C3 column: 8.0 ms against 6.7 for the two passes (NOTES.md, profiles/r04/single_pass_split.txt):
// opt-in (CS_SPLIT_SINGLE=1).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "cs_internal.h"
#include "device_utils.h"
#include "split_parts.h"
#include "tile_utils.h"

using namespace cs;
using namespace csdev;

namespace cs {
// returns 1: columns written; 0: not this path's case (nothing done); -1: the single pass gave up (fallback counted)
int split_single(const cs_column* col, const unsigned char* delim, int dlen, int tokens, hipStream_t s,
                 std::vector<std::unique_ptr<cs_column>>& cols, bool reverse, int64_t span, bool plain_walk);
}

namespace {

constexpr int kCols1 = 24;          // columns the single pass provisions at most (a register per pair of columns, three times)
constexpr int kTables1 = 640;       // flush tables of a wave: 32 entries of 16 bytes, 16 words of region-start bits, 16 prefix counts
constexpr int kThreads1 = 768;      // twelve waves per workgroup = a CU's worth at three waves per SIMD: the scanner waves get a CU
                                    // of their own (sharing SIMDs with workers they slowed those, and with the static sequence of
                                    // sub-tiles the slowest wave sets everybody's pace: 13.4 ms), and a workgroup's twelve
                                    // consecutive sub-tiles complete the cache lines they share in one L2
typedef uint32_t u32x12 __attribute__((ext_vector_type(12)));  // a register per pair of columns
typedef cstile::u64 u64;

// error word (ctl[0])
constexpr unsigned kErrWait = 1u, kErrCapacity = 2u, kErrWide = 4u, kErrColumns = 8u, kErrScanner = 16u;

struct ColOut5 {
  uint8_t* chars;
  int32_t* offsets;
  uint8_t* validity;
  long long cap;  // bytes provisioned for chars
};

// ---- the sample ----------------------------------------------------------------------------------------
struct SampleArgs {
  ColView in;
  uint32_t dpat;
  unsigned long long d64;
  int dlen, tokens, reverse, cap;
  long long nsub, nsamp;
  unsigned long long* sums;  // [kCols1] bytes per column over the sampled sub-tiles, [kCols1] their squares
  int* mx;                   // [0] most tokens in a row, [1] set when a row has more than kCols1 tokens
};
template <int MODE>
__global__ void __launch_bounds__(256) k_split_sample(SampleArgs a) {
  constexpr bool WS = MODE == 1, MULTI = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const long long waves = (long long)gridDim.x * 4;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap + 32);
  // (a wave takes several sampled sub-tiles and adds its sums once: a million atomics on 64 addresses took 0.3 ms)
  unsigned long long mine = 0, mine_sq = 0;
  int most = 0;
  bool wide = false;
  for (long long w = (long long)blockIdx.x * 4 + wv; w < a.nsamp; w += waves) {
    const long long sub = a.nsamp == a.nsub ? w : (long long)(((unsigned long long)w * (unsigned long long)a.nsub) / (unsigned long long)a.nsamp);
    const SubTile t = load_subtile(a.in, sub, lds_in, lane);
    TokensT<true, WS, MULTI> tk(lds_in, t.lead + t.rbeg, t.n, t.live, a.dpat, a.tokens, a.d64, a.dlen, a.reverse != 0);
    int count = 0;
    for (int k = 0; k < kCols1; ++k) {
      int lo = 0, hi = 0;
      const bool has = tk.next(lo, hi);
      if (!__any(has)) break;
      const unsigned long long sum = (unsigned long long)wave_reduce_sum(has ? hi - lo : 0);
      if (lane == k) {
        mine += sum;
        mine_sq += sum * sum;
      }
      count += has ? 1 : 0;
    }
    wide = wide || tk.more;
    most = max(most, count);
    cstile::wave_lds_fence();  // (the next sub-tile is staged over this one)
  }
  if (__any(wide) && lane == 0) atomicOr(a.mx + 1, 1);
  for (int d = 32; d > 0; d >>= 1) most = max(most, __shfl_xor(most, d, 64));
  if (lane == 0) atomicMax(a.mx, most);
  if (lane < kCols1 && mine) {
    atomicAdd(a.sums + lane, mine);
    atomicAdd(a.sums + kCols1 + lane, mine_sq);
  }
}

// ---- the single pass -----------------------------------------------------------------------------------
struct Emit5Args {
  ColView in;
  uint32_t dpat;
  unsigned long long d64;
  int dlen, tokens, reverse;
  int ncap, npairs;    // provisioned columns, pairs of them (= scanner waves)
  long long nsub;
  long long stride;    // words between two pairs' rows of status / excl
  int cap_in, cap_out; // bytes of a wave's in tile / out tile
  int wave_bytes;      // LDS of a wave (slack, in tile, out tile, tables)
  int scan_blocks;     // leading workgroups whose waves are the scanners
  const ColOut5* cols;
  u64* status;         // [npairs][stride]: flag | bytes of column 2p + 1 << 31 | bytes of column 2p, per sub-tile
  u64* excl;           // [npairs][stride]: flag | the same two columns' bytes in front of the sub-tile
  unsigned* ctl;       // [0] error word, [1] columns so far (atomicMax; starts at what the sample saw)
  long long* totals;   // [2 * npairs] every column's bytes (written by the scanners)
  uint8_t* tilecols;   // [nsub] columns the sub-tile wrote offsets / validity for
  unsigned long long* prof;   // instrumented builds (make prof): ten cycle counters
  unsigned long long* trace;  // statistics runs: four time stamps for every 1024th sub-tile
  int debug;           // CS_SPLIT_DEBUG (measurement only, wrong results): 1 no offsets stores, 2 no chars stores, 4 no prefix (no
                       // scanners, made-up positions), 8 chars stores at 16-byte boundaries
};

// what a round leaves in column k's lane: validity word, byte count | region start << 16
__device__ __forceinline__ void leave_in_lane5(int k, unsigned long long valid, int packed, uint32_t& vm_lo, uint32_t& vm_hi, uint32_t& t_pack) {
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

// two independent wave scans interleaved: each covers the other's DPP wait states
__device__ __forceinline__ void scan_two_fused(int& x, int& y) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_u32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_u32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(x), "+v"(y));
}

// The scanners: the published words of ONE pair of columns, in tile order, into exclusive prefixes (tile_utils.h:
// prefix_scanner, with two 31-bit values a word).  A lane takes FOUR consecutive sub-tiles, a STEP covers 256 (one pair of
// wave scans per 256 sub-tiles), and kHelpers waves share a pair's steps round-robin: fetching a step (a device-scope
// round trip), its scans and its stores run in parallel, only the running sums are handed from step to step -- through an
// LDS ring of the workgroup, a few hundred cycles a link.  (One wave per pair, 64 sub-tiles a step, passed 140 sub-tiles
// per microsecond and the kernel ran at that rate: 11.4 ms; 256 a step with 16-byte accesses: 9.5 ms.)
// Until a step is complete its wave polls it; every sub-tile up to and including the first unpublished one gets its
// prefix as soon as the step's base is known (a sub-tile's prefix needs its predecessors only).
constexpr int kHelpers = 4;
constexpr int kPairsPerGroup = 3;  // 12 waves a scanner workgroup
struct ScanRing {                  // one pair's ring: slot s & 7 holds the sums in front of step s once tag == s + 1
  unsigned long long val[8];       // bytes of column 2p + 1 << 32 | bytes of column 2p
  unsigned tag[8];
};
__device__ __forceinline__ bool pair_helper(const u64* status, u64* excl, long long ntiles, int lane, int h, ScanRing* ring, long long* totals,
                                            unsigned* error, unsigned long long* trace) {
  cstile::gptr<u64> ex = cstile::as_global(excl);
  for (long long s = h; s * 256 < ntiles; s += kHelpers) {
    const long long t0 = s * 256 + 4 * lane;  // the lane's first sub-tile
    const u64* at = status + t0;
    bool in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) in[i] = t0 + i < ntiles;
    bool have_base = false;
    uint32_t base_a = 0, base_b = 0;
    int delivered = -1;  // leading sub-tiles of the step that have their prefix
    cstile::SpinClock clock;
    int idle = 0;
    for (;;) {
      cstile::u32x4 q0, q1;
      asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(q0), "=&v"(q1)
                   : "v"(at)
                   : "memory");
      const u64 x[4] = {((u64)q0.y << 32) | q0.x, ((u64)q0.w << 32) | q0.z, ((u64)q1.y << 32) | q1.x, ((u64)q1.w << 32) | q1.z};
      bool pub[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pub[i] = !in[i] || (x[i] >> 62) != 0;  // (beyond the column: nothing to wait for, nothing to add)
      const int r = pub[0] ? (pub[1] ? (pub[2] ? (pub[3] ? 4 : 3) : 2) : 1) : 0;  // leading published sub-tiles of the lane
      const u64 missing = __ballot(r < 4);
      const int first = missing ? __builtin_ctzll(missing) : 64;  // lanes below are complete
      const int lead = missing ? 4 * first + __builtin_amdgcn_readlane(r, first) : 256;
      int va[4], vb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool take = in[i] && (lane < first || (lane == first && i < r));
        va[i] = take ? (int)(uint32_t)(x[i] & 0x7fffffffull) : 0;
        vb[i] = take ? (int)(uint32_t)((x[i] >> 31) & 0x7fffffffull) : 0;
      }
      const int a1 = va[0], a2 = a1 + va[1], a3 = a2 + va[2], ta = a3 + va[3];
      const int b1 = vb[0], b2 = b1 + vb[1], b3 = b2 + vb[2], tb = b3 + vb[3];
      int ia = ta, ib = tb;
      if (lead > delivered) scan_two_fused(ia, ib);
      // the sums in front of the step: wait for them only when the step is complete (else poll the step again meanwhile)
      while (!have_base) {
        if (__hip_atomic_load(&ring->tag[s & 7], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)(s + 1)) {
          const unsigned long long bv = __hip_atomic_load(&ring->val[s & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          base_a = (uint32_t)bv;
          base_b = (uint32_t)(bv >> 32);
          have_base = true;
          break;
        }
        if (missing) break;
        if (clock.expired()) return false;
        if ((++idle & 1023) == 0 && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      }
      if (have_base && !missing) {
        // complete: hand the sums on FIRST (the next step's wave is waiting for nothing else)
        const uint32_t na = base_a + (uint32_t)__builtin_amdgcn_readlane(ia, 63), nb = base_b + (uint32_t)__builtin_amdgcn_readlane(ib, 63);
        if ((na | nb) >> 31) {  // a column of 2 GiB: int32 offsets cannot name it
          if (lane == 0) atomicOr(error, kErrWide);
          return false;
        }
        if (lane == 0) {
          __hip_atomic_store(&ring->val[(s + 1) & 7], ((unsigned long long)nb << 32) | na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&ring->tag[(s + 1) & 7], (unsigned)(s + 2), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((s + 1) * 256 >= ntiles) {
            totals[0] = (long long)na;
            totals[1] = (long long)nb;
          }
        }
      }
      if (have_base && lead > delivered) {
        const uint32_t ea = base_a + (uint32_t)(ia - ta), eb = base_b + (uint32_t)(ib - tb);  // in front of the lane's first sub-tile
        const int pa[4] = {0, a1, a2, a3}, pb[4] = {0, b1, b2, b3};
        u64 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = cstile::kFlagInc | ((u64)(eb + (uint32_t)pb[i]) << 31) | (u64)(ea + (uint32_t)pa[i]);
        if (lane < first && in[3]) {  // a complete lane: 32 contiguous bytes
          const cstile::u32x4 lo = {(uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32)};
          const cstile::u32x4 hi = {(uint32_t)o[2], (uint32_t)(o[2] >> 32), (uint32_t)o[3], (uint32_t)(o[3] >> 32)};
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(excl + t0), "v"(lo), "v"(hi) : "memory");
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (in[i] && (lane < first || (lane == first && i <= r))) __hip_atomic_store(ex + (t0 + i), o[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (trace && (t0 & 1023) == 0 && in[0] && lane <= first) trace[(t0 >> 10) * 4 + 1] = wall_clock64();
        delivered = lead;
        clock.reset();
      }
      if (have_base && !missing) break;
      if (clock.expired()) return false;
      if ((++idle & 255) == 0 && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;  // (somebody gave up)
      __builtin_amdgcn_s_sleep(1);
    }
  }
  return true;
}

// stores to global memory at any alignment (the address-space pointer types spelled out: a template argument would drop
// the typedef's alignment attribute)
typedef uint32_t g_u32x4u __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned long long g_u64u __attribute__((aligned(1)));
typedef uint32_t g_u32u __attribute__((aligned(1)));
typedef uint16_t g_u16u __attribute__((aligned(1)));
typedef __attribute__((address_space(1))) g_u32x4u* gp_u32x4u;
typedef __attribute__((address_space(1))) g_u64u* gp_u64u;
typedef __attribute__((address_space(1))) g_u32u* gp_u32u;
typedef __attribute__((address_space(1))) g_u16u* gp_u16u;

// MODE 0: one-byte delimiter, 1: whitespace, 2: delimiter of 2..8 ASCII bytes.  PLAIN: the sentinel walk (cs_split.hip).
template <int MODE, bool PLAIN>
__global__ void __launch_bounds__(kThreads1) k_split_emit5(Emit5Args a) {
  constexpr bool WS = MODE == 1, MULTI = MODE == 2;
  static_assert(!PLAIN || MODE == 0, "the sentinel walk is the one-byte delimiter's");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  // the workgroup's mask table first (17 entries, both waves write the same values); then per wave 16 bytes of slack, the in
  // tile, 32 bytes, the out tile, the flush tables
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + 288 + (size_t)wv * a.wave_bytes + 16;
  uint8_t* lds_out = lds_in + a.cap_in + 32;
  cstile::u32x4* f_entry = reinterpret_cast<cstile::u32x4*>(lds_out + a.cap_out);
  uint32_t* f_bits = reinterpret_cast<uint32_t*>(lds_out + a.cap_out + 512);
  uint32_t* f_pfx = f_bits + 16;
  const ColView& in = a.in;
  const int ncap = a.ncap, npairs = a.npairs;

  // ---- scanner waves: the first workgroups, a CU each (sharing SIMDs with workers they slowed those, and with the static
  // sequence of sub-tiles the slowest wave sets everybody's pace); kHelpers waves a pair of columns
  if ((int)blockIdx.x < a.scan_blocks) {
    ScanRing* rings = reinterpret_cast<ScanRing*>(reinterpret_cast<uint8_t*>(smem) + 1024);
    if (threadIdx.x < kPairsPerGroup * 8) {
      ScanRing* rg = rings + threadIdx.x / 8;
      const int slot = threadIdx.x & 7;
      rg->val[slot] = 0ull;
      rg->tag[slot] = slot == 0 ? 1u : 0u;  // (nothing in front of step 0)
    }
    __syncthreads();
    const int pair = (int)blockIdx.x * kPairsPerGroup + wv / kHelpers;
    if (wv >= kPairsPerGroup * kHelpers || pair >= npairs || (a.debug & 4)) return;
    if (!pair_helper(a.status + (long long)pair * a.stride, a.excl + (long long)pair * a.stride, a.nsub, lane, wv % kHelpers, rings + wv / kHelpers,
                     a.totals + 2 * pair, a.ctl, pair == 0 ? a.trace : nullptr) &&
        lane == 0)
      atomicOr(a.ctl, kErrScanner);
    return;
  }

  cstile::u32x4* tail = reinterpret_cast<cstile::u32x4*>(smem);
  if (lane <= 16) {
    auto first = [](int k) -> uint32_t { return k >= 4 ? 0xFFFFFFFFu : (k <= 0 ? 0u : (1u << (8 * k)) - 1u); };
    tail[lane] = cstile::u32x4{first(lane), first(lane - 4), first(lane - 8), first(lane - 12)};
  }
  const cstile::u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int i = lane * 16; i < a.cap_out; i += 64 * 16) *reinterpret_cast<cstile::u32x4*>(lds_out + i) = zero4;

  // ---- sub-tiles: wave g of the W worker waves takes g, g + W, g + 2W, ... (the two waves of a workgroup neighbouring
  // sub-tiles).  NOT tickets: a sub-tile leaves only when everything in front of it has been published, so a wave that
  // draws its next ticket early -- the one at the frontier, while the others wait for their prefixes -- gets sub-tiles
  // close to each other, processes them one after the other with everybody waiting for each, draws again early ...;
  // the ticket form ran 80 ms with the share of such waves growing all the way (traced: every sub-tile the scanners had
  // to wait for belonged to a wave whose previous sub-tile lay 1 to 250 in front of it instead of W).  The static
  // sequence keeps a wave's sub-tiles W apart; it needs the grid resident (waits are bounded in time, the host falls back).
  const long long W = ((long long)gridDim.x - a.scan_blocks) * (kThreads1 / 64);
  long long tile = ((long long)blockIdx.x - a.scan_blocks) * (kThreads1 / 64) + wv;
  long long t_nxt = tile + W;
  if (tile >= a.nsub) return;

  // lane k keeps column k: chars, offsets, validity, capacity
  uint8_t* c_chars = nullptr;
  int32_t* c_off = nullptr;
  uint8_t* c_valid = nullptr;
  uint32_t c_cap = 0;
  if (lane < ncap) {
    const ColOut5 c = a.cols[lane];
    c_chars = c.chars;
    c_off = c.offsets;
    c_valid = c.validity;
    c_cap = (uint32_t)c.cap;
  }
  const uint32_t coff_lo = (uint32_t)(uintptr_t)c_off, coff_hi = (uint32_t)((uintptr_t)c_off >> 32);

  cstile::TileOffs cur = cstile::load_tile_offsets(in.offsets, in.rows, tile, lane);
  cstile::TileOffs nxt = cur;
  if (t_nxt < a.nsub) nxt = cstile::load_tile_offsets(in.offsets, in.rows, t_nxt, lane);
  cstile::TileChars pf;
#pragma unroll
  for (int j = 0; j < cstile::kPfChunks; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);

  // ---- the previous sub-tile, assembled and published but not yet in memory
  long long p_tile = -1;
  int p_nrows = 0, p_reach = 0, p_m = 0, p_rg = 0;
  u32x12 relv = {};                             // the rows' offsets relative to the sub-tile, two columns a register
  uint32_t t_pack = 0, vm_lo = 0, vm_hi = 0;    // column lanes: bytes | region start << 16, validity word
  const u64* my_excl = a.excl + (long long)(lane < npairs ? lane : 0) * a.stride;

#if defined(CS_PHASE_PROF)
  unsigned long long phase_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long phase_t = __builtin_readcyclecounter();
#endif
  unsigned total_spins = 0, iters = 0;
  unsigned long long gap_sum = 0;
  auto finish_pending = [&](u64 first) {
    // the columns' positions: the pair's word in lane k >> 1
    u64 v = lane < npairs ? first : cstile::kFlagInc;
    const bool traced = a.trace && (p_tile & 1023) == 0;
    if (traced && lane == 0) a.trace[(p_tile >> 10) * 4 + 3] = wall_clock64();
    if (a.debug & (4 | 16)) v = cstile::kFlagInc;  // (measurement: no wait; positions made up below)
    {
      int spins = 0;
      cstile::SpinClock clock;
      while (__any((v >> 62) == 0)) {
        ++spins;
        ++total_spins;
        bool give_up = clock.expired();
        if ((spins & 255) == 0 && __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) give_up = true;
        if (give_up) {
          if (lane == 0) atomicOr(a.ctl, kErrWait);
          return false;
        }
        // (back off: thousands of waves polling eleven lines each slow the scanners they wait for)
        if (spins < 4) __builtin_amdgcn_s_sleep(8);
        else if (spins < 16) __builtin_amdgcn_s_sleep(32);
        else __builtin_amdgcn_s_sleep(127);
        if (lane < npairs && (v >> 62) == 0) v = cstile::status_load(my_excl + p_tile);
      }
    }
    if (traced && lane == 0) a.trace[(p_tile >> 10) * 4 + 2] = wall_clock64();
    CS_PHASE_MARK(4);
    const uint32_t pa = (uint32_t)(v & 0x7fffffffull), pb = (uint32_t)((v >> 31) & 0x7fffffffull);
    const uint32_t qa = (uint32_t)__shfl((int)pa, lane >> 1, 64), qb = (uint32_t)__shfl((int)pb, lane >> 1, 64);
    uint32_t c_pos = (lane & 1) ? qb : qa;  // (lane k: column k's bytes in front of the sub-tile)
    if (a.debug & (4 | 16)) c_pos = c_cap > 16384u ? (uint32_t)(((unsigned long long)p_tile * 263ull) % (unsigned long long)(c_cap - 8192u)) : 0u;
    const int t_sum = (int)(t_pack & 0xffffu), t_rg = (int)(t_pack >> 16);
    const long long pr0 = p_tile * 64;
    // ---- offsets: position + relative offset; a row beyond the sub-tile's last writes the final entry (its relative
    // offset is the sub-tile's total); columns the sub-tile does not reach: null rows at the position
    {
      const int rowoff = min(lane, p_nrows) * 4;
      for (int k = 0; k < p_m; ++k) {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)c_pos, k);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)coff_lo, k), hi = (uint32_t)__builtin_amdgcn_readlane((int)coff_hi, k);
        const uint32_t pw = relv[k >> 1];
        const uint32_t rel = k < p_reach ? ((pw >> ((k & 1) * 16)) & 0xffffu) : 0u;
        int32_t* dst = reinterpret_cast<int32_t*>((((unsigned long long)hi << 32) | lo) + (unsigned long long)pr0 * 4ull);
        if (!(a.debug & 1)) *cstile::as_global(reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(dst) + rowoff)) = (int32_t)(base + rel);
      }
    }
    CS_PHASE_MARK(5);
    // ---- per column, the columns in the lanes
    const bool mine = lane < p_m;
    if (mine) {
      *cstile::as_global(reinterpret_cast<unsigned long long*>(c_valid + p_tile * 8)) = ((unsigned long long)vm_hi << 32) | vm_lo;
      if (pr0 + 64 == in.rows) cstile::as_global(c_off)[in.rows] = (int32_t)(c_pos + (uint32_t)t_sum);  // (a last sub-tile of exactly 64 rows)
      if (c_pos + (uint32_t)t_sum > c_cap) atomicOr(a.ctl, kErrCapacity);
    }
    const bool over = __any(mine && c_pos + (uint32_t)t_sum > c_cap);
    const bool act = mine && t_sum > 0 && !over && !(a.debug & 2);
    const int nwhole = t_sum >> 4;
    const unsigned long long actm = __ballot(act);
    if (lane < 16) f_bits[lane] = 0u;
    uint8_t* ga = c_chars + ((a.debug & 8) ? (c_pos & ~15u) : c_pos);
    cstile::wave_lds_fence();
    if (act) {
      const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(actm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)actm, 0u));
      f_entry[rank] = cstile::u32x4{(uint32_t)(uintptr_t)ga, (uint32_t)((uintptr_t)ga >> 32), (uint32_t)(t_rg >> 4) | ((uint32_t)nwhole << 16), 0u};
      lds_or(f_bits + (t_rg >> 9), 1u << ((t_rg >> 4) & 31));
      // the bytes behind the region's last whole chunk
      const uint8_t* src = lds_out + t_rg + 16 * nwhole;
      uint8_t* d = ga + 16 * nwhole;
      if (t_sum & 8) {
        *(gp_u64u)(void*)d = *reinterpret_cast<const unsigned long long*>(src);
        src += 8;
        d += 8;
      }
      if (t_sum & 4) {
        *(gp_u32u)(void*)d = *reinterpret_cast<const uint32_t*>(src);
        src += 4;
        d += 4;
      }
      if (t_sum & 2) {
        *(gp_u16u)(void*)d = *reinterpret_cast<const uint16_t*>(src);
        src += 2;
        d += 2;
      }
      if (t_sum & 1) *cstile::as_global(d) = *src;
    }
    cstile::wave_lds_fence();
    {
      const int cnt = lane < 16 ? __builtin_popcount(f_bits[lane]) : 0;
      const int inc = wave_inclusive_scan_fused(cnt);
      if (lane < 16) f_pfx[lane] = (uint32_t)(inc - cnt);
    }
    cstile::wave_lds_fence();
    CS_PHASE_MARK(6);
    // ---- the flush pass: every 16-byte chunk of the out tile below p_rg leaves (whole chunks) and is zeroed again
    const int nchunks = p_rg >> 4;
    for (int c = lane; c < nchunks; c += 64) {
      const uint32_t word = f_bits[c >> 5];
      const int rank = (int)f_pfx[c >> 5] + __builtin_popcount(word & (0xFFFFFFFFu >> (31 - (c & 31)))) - 1;
      const cstile::u32x4 v4 = *reinterpret_cast<const cstile::u32x4*>(lds_out + 16 * c);
      *reinterpret_cast<cstile::u32x4*>(lds_out + 16 * c) = zero4;
      if (rank >= 0) {  // (chunks in front of the first region: none -- regions begin at 0 -- but an empty table must not be read)
        const cstile::u32x4 e = f_entry[rank];
        const int rel = c - (int)(e.z & 0xffffu);
        if (rel < (int)(e.z >> 16)) {
          uint8_t* g = reinterpret_cast<uint8_t*>((((unsigned long long)e.y << 32) | e.x) + (unsigned long long)(16 * rel));
          *(gp_u32x4u)(void*)g = v4;
        }
      }
    }
    CS_PHASE_MARK(7);
    cstile::wave_lds_fence();
    return true;
  };

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
    // Everything fetched here for later -- the prefix poll, the column count, the offsets two tiles ahead -- is
    // assigned unconditionally (clamped addresses) and handed to its loop-carried variable at the bottom of the iteration
    // (cs_regex.hip: a conditional assignment of a value in flight makes the compiler wait for it at the join).
    const u64 p_first = cstile::status_load(my_excl + (p_tile >= 0 ? p_tile : 0));
    const unsigned g_seen_v = __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool has_next = t_nxt < a.nsub;
    const long long t_nn = t_nxt + W;
    const cstile::TileOffs nn = cstile::load_tile_offsets(in.offsets, in.rows, t_nn < a.nsub ? t_nn : a.nsub - 1, lane);
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
    }
    cstile::wave_lds_fence();

    CS_PHASE_MARK(0);
    TokensT<true, WS, MULTI> tk(lds_in, lead + rbeg, n, live, a.dpat, a.tokens, a.d64, a.dlen, a.reverse != 0);
    uint32_t m0 = 0, m1 = 0, m2 = 0;
    int tcur = 0;
    if (PLAIN) {
      m0 = (uint32_t)tk.m_lo;
      m1 = (uint32_t)(tk.m_lo >> 32);
      m2 = tk.m_hi;
      if (live) {
        const int q = tk.sa + n;
        const uint32_t bit = 1u << (q & 31);
        if (q < 32) m0 |= bit;
        else if (q < 64) m1 |= bit;
        else m2 |= bit;
      }
    }
    auto step = [&](int& lo, int& hi) -> bool {  // the row's next token
      if (PLAIN) {
        const bool has = (m0 | m1 | m2) != 0;
        const uint32_t q = lowest96(m0, m1, m2);
        lo = tcur;
        hi = (int)q - tk.sa;
        tcur = hi + 1;
        const unsigned long long l64 = ((unsigned long long)m1 << 32) | m0, d64 = l64 - 1;
        const uint32_t d2 = m2 - (l64 == 0 ? 1u : 0u);
        m0 &= (uint32_t)d64;
        m1 &= (uint32_t)(d64 >> 32);
        m2 &= d2;
        return has;
      }
      return tk.next(lo, hi);
    };

    CS_PHASE_MARK(1);
    // ---- phase 1: the token walk and the wave scans of every pair of columns.  What the assembly below needs stays in
    // registers (token starts and lengths a byte each, the rows' relative offsets), what the others need is PUBLISHED
    // before anything else happens: the previous sub-tile's prefix -- published an iteration ago -- then has had this
    // phase, the previous assembly and a staging to arrive in, and a wave does not feel the slowest wave of the grid in
    // every iteration (with the publish behind the assembly the kernel ran 9.5 ms, 2.5 of them waiting).
    const uint8_t* tok_src = lds_in + lead + rbeg;
    uint32_t t_pack_n = 0, vm_lo_n = 0, vm_hi_n = 0;
    u32x12 pre2v = {}, lolen = {};
    int rg = 0, reach = 0, k = 0;
    for (; k < ncap; k += 2) {
      int loA = 0, hiA = 0, loB = 0, hiB = 0;
      const bool hasA = step(loA, hiA);
      const unsigned long long vA = __ballot(hasA);
      if (vA == 0) break;  // no row of the sub-tile reaches column k (nor any behind it)
      const bool hasB = k + 1 < ncap ? step(loB, hiB) : false;
      const unsigned long long vB = __ballot(hasB);
      reach = k + (vB ? 2 : 1);
      const int lenA = hasA ? hiA - loA : 0, lenB = hasB ? hiB - loB : 0;
      const int len2 = lenA | (lenB << 16);
      const int incl = wave_inclusive_scan_fused(len2);
      const int tot2 = rl(incl, 63);
      pre2v[k >> 1] = (uint32_t)(incl - len2);  // (no borrow between the halves: a prefix is at least the lane's own length)
      lolen[k >> 1] = (uint32_t)(hasA ? loA : 0) | ((uint32_t)lenA << 8) | ((uint32_t)(hasB ? loB : 0) << 16) | ((uint32_t)lenB << 24);
      const int csumA = tot2 & 0xffff, csumB = (int)((unsigned)tot2 >> 16);
      const int rgB = rg + ((csumA + 15) & ~15);
      leave_in_lane5(k, vA, csumA | (rg << 16), vm_lo_n, vm_hi_n, t_pack_n);
      leave_in_lane5(k + 1, vB, csumB | (rgB << 16), vm_lo_n, vm_hi_n, t_pack_n);
      rg = rgB + ((csumB + 15) & ~15);
    }
    CS_PHASE_MARK(2);
    // a row with tokens left: more columns than provisioned
    {
      const bool more = PLAIN ? (m0 | m1 | m2) != 0 : tk.more;
      if (__any(more)) {
        if (lane == 0) atomicOr(a.ctl, kErrColumns);
        return;
      }
    }
    // ---- publish: the sub-tile's bytes per column, two columns a word; the column count
    {
      const int t_sum = (int)(t_pack_n & 0xffffu);
      const uint32_t sa = (uint32_t)__shfl(t_sum, (2 * lane) & 63, 64), sb = (uint32_t)__shfl(t_sum, (2 * lane + 1) & 63, 64);
      if (lane < npairs) cstile::status_store(a.status + (long long)lane * a.stride + tile, cstile::kFlagAgg | ((u64)sb << 31) | (u64)sa);
      if (a.trace && (tile & 1023) == 0 && lane == 0) a.trace[(tile >> 10) * 4 + 0] = wall_clock64();
    }
    const int g_seen = __builtin_amdgcn_readfirstlane((int)g_seen_v);
    int m = g_seen;
    if (reach > g_seen) {
      if (lane == 0) atomicMax(a.ctl + 1, (unsigned)reach);
      m = reach;
    }
    if (lane == 0) *cstile::as_global(a.tilecols + tile) = (uint8_t)m;
    CS_PHASE_MARK(3);
    // ---- the previous sub-tile leaves (its prefix was polled at the top)
    if (p_tile >= 0 && !finish_pending(p_first)) return;
    // ---- phase 2: the tokens into their columns' regions, side by side from byte 0 of the out tile
    {
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
      for (int c = 0; c < reach; c += 2) {
        const uint32_t pw = lolen[c >> 1], pre2 = pre2v[c >> 1];
        const int loA = (int)(pw & 0xffu), lenA = (int)((pw >> 8) & 0xffu), loB = (int)((pw >> 16) & 0xffu), lenB = (int)(pw >> 24);
        const int rgA = (int)((uint32_t)__builtin_amdgcn_readlane((int)t_pack_n, c) >> 16), rgB = (int)((uint32_t)__builtin_amdgcn_readlane((int)t_pack_n, c + 1) >> 16);
        // both tokens' first sixteen bytes and tail masks (a lane without a token reads its row's start and masks it all)
        const cstile::lds_u32x4u rA = *reinterpret_cast<const cstile::lds_u32x4u*>(tok_src + loA);
        const cstile::lds_u32x4u rB = *reinterpret_cast<const cstile::lds_u32x4u*>(tok_src + loB);
        const cstile::u32x4 mA = tail[min(lenA, 16)], mB = tail[min(lenB, 16)];
        const int atA = rgA + (int)(pre2 & 0xffffu), atB = rgB + (int)(pre2 >> 16);
        // (a lane without bytes ORs zeros: at an address of its own -- the same dword from many lanes would serialise)
        or_token(cstile::u32x4{rA.x, rA.y, rA.z, rA.w}, mA, lenA ? atA : lane * 20);
        or_token(cstile::u32x4{rB.x, rB.y, rB.z, rB.w}, mB, lenB ? atB : lane * 20);
        if (__any(max(lenA, lenB) > 16)) {
          long_tokens(loA, lenA, atA);
          long_tokens(loB, lenB, atB);
        }
      }
    }
    CS_PHASE_MARK(8);
    relv = pre2v;
    t_pack = t_pack_n;
    vm_lo = vm_lo_n;
    vm_hi = vm_hi_n;
    if (p_tile >= 0) {
      gap_sum += (unsigned long long)(tile - p_tile);
      ++iters;
    }
    p_tile = tile;
    p_nrows = nrows;
    p_reach = reach;
    p_m = m;
    p_rg = rg;
    if (!has_next) break;
    tile = t_nxt;
    t_nxt = t_nn;
    nxt = nn;
  }
#if defined(CS_PHASE_PROF)
  CS_PHASE_MARK(9);
  if (lane == 0 && a.prof)
    for (int i = 0; i < 10; ++i) atomicAdd(a.prof + i, phase_acc[i]);
#endif
  (void)finish_pending(cstile::status_load(my_excl + p_tile));
  if (lane == 0) {  // (statistics: CS_SPLIT1_STATS)
    atomicAdd(a.ctl + 4, total_spins);
    atomicAdd(a.ctl + 7, iters);
    atomicAdd(reinterpret_cast<unsigned long long*>(a.ctl + 8), gap_sum);
  }
}

// ---- null rows of columns that appeared after a sub-tile had been written ----------------------------------
struct FixupArgs {
  long long rows, nsub, stride;
  int ncols;
  const uint8_t* tilecols;
  const u64* excl;
  const ColOut5* cols;
};
__global__ void __launch_bounds__(256) k_split_fixup(FixupArgs a) {
  const int lane = threadIdx.x & 63;
  const long long waves = (long long)gridDim.x * 4;
  // a wave looks at 64 sub-tiles at once (a lane each) and visits those that wrote fewer columns than there are
  for (long long t0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; t0 < a.nsub; t0 += waves * 64) {
    const int mine = t0 + lane < a.nsub ? (int)a.tilecols[t0 + lane] : a.ncols;
    unsigned long long todo = __ballot(mine < a.ncols);
    while (todo) {
      const int j = __builtin_ctzll(todo);
      todo &= todo - 1;
      const long long t = t0 + j;
      const int m = __builtin_amdgcn_readlane(mine, j);
      const long long r0 = t * 64;
      const int nrows = (int)min(64ll, a.rows - r0);
      for (int k = m; k < a.ncols; ++k) {
        const u64 w = a.excl[(long long)(k >> 1) * a.stride + t];
        const int32_t pos = (int32_t)((k & 1) ? ((w >> 31) & 0x7fffffffull) : (w & 0x7fffffffull));
        const ColOut5 c = a.cols[k];
        c.offsets[r0 + min(lane, nrows)] = pos;
        if (lane == 0) {
          if (r0 + 64 == a.rows) c.offsets[a.rows] = pos;
          reinterpret_cast<unsigned long long*>(c.validity)[t] = 0ull;
        }
      }
    }
  }
}

}  // namespace

namespace cs {

int split_single(const cs_column* col, const unsigned char* delim, int dlen, int tokens, hipStream_t s,
                 std::vector<std::unique_ptr<cs_column>>& cols, bool reverse, int64_t span, bool plain_walk) {
  const bool ws = delim == nullptr;
  const int mode = ws ? 1 : (dlen > 1 ? 2 : 0);
  const int64_t rows = col->rows;
  if (rows + 1 >= ((int64_t)1 << 31)) return 0;
  if (max_row_bytes(col, s) + 3 > 96) return 0;  // (every row inside the 96-bit masks)
  unsigned long long d64 = 0;
  for (int i = 0; !ws && i < dlen; ++i) d64 |= (unsigned long long)delim[i] << (8 * i);
  const uint32_t dpat = 0x01010101u * (ws ? 0u : (uint32_t)delim[0]);
  const int64_t nsub = (rows + kSub - 1) / kSub;
  const int cap_in = (int)((span + 15 + 32 + 15) & ~(int64_t)15);

  // ---- the sample: every sub-tile of a small column, one in 64 of a large one (at least 4096)
  int64_t nsamp = std::min<int64_t>(nsub, std::max<int64_t>(4096, nsub / 64));
  if (const char* e = cs::cfg("CS_SPLIT1_SAMPLE")) nsamp = std::max<int64_t>(1, std::min<int64_t>(nsub, atoll(e)));  // (tests: the estimate path on small columns)
  Buf sbuf = dev_alloc(sizeof(unsigned long long) * 2 * kCols1 + 4 * sizeof(int), s);
  CS_HIP(hipMemsetAsync(sbuf->p, 0, sizeof(unsigned long long) * 2 * kCols1 + 4 * sizeof(int), s));
  SampleArgs sa{view_of(col), dpat, d64, dlen, tokens, reverse ? 1 : 0, cap_in, nsub, nsamp, ptr<unsigned long long>(sbuf),
                reinterpret_cast<int*>(ptr<unsigned long long>(sbuf) + 2 * kCols1)};
  {
    ProfScope ps("k_split_sample", s);
    const unsigned g = (unsigned)((std::min<int64_t>(nsamp, 4096) + 3) / 4);
    const size_t lds = (size_t)(cap_in + 32) * 4;
    if (mode == 1) hipLaunchKernelGGL(k_split_sample<1>, dim3(g), dim3(256), lds, s, sa);
    else if (mode == 2) hipLaunchKernelGGL(k_split_sample<2>, dim3(g), dim3(256), lds, s, sa);
    else hipLaunchKernelGGL(k_split_sample<0>, dim3(g), dim3(256), lds, s, sa);
  }
  CS_HIP(hipGetLastError());
  struct HostSample {
    unsigned long long sums[2 * kCols1];
    int mx[4];
  };
  HostSample* hs = (HostSample*)pinned_scratch(sizeof(HostSample));
  CS_HIP(hipMemcpyAsync(hs, sbuf->p, sizeof(HostSample), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  const int seen = hs->mx[0];
  if (seen <= 0 || hs->mx[1]) return 0;  // an all-null column / more than 32 columns: the other paths
  const bool exact = nsamp == nsub;
  int ncap = exact ? seen : std::min(kCols1, (seen + 4 + 1) & ~1);
  if (tokens > 0) ncap = std::min(ncap, tokens);
  ncap = std::max(ncap, seen);
  const int npairs = (ncap + 1) / 2;
  std::vector<long long> capk(ncap);
  for (int k = 0; k < ncap; ++k) {
    const double n = (double)nsamp, N = (double)nsub;
    const double mean = (double)hs->sums[k] / n;
    const double var = std::max(0.0, (double)hs->sums[kCols1 + k] / n - mean * mean);
    const double est = N * mean, sigma = exact ? 0.0 : N * std::sqrt(var / n);
    const double want = est + 8.0 * sigma + (exact ? 64.0 : 65536.0);
    if (want >= 2147483000.0) return 0;  // int32 offsets could not name the column: the two-pass kernels choose the width
    capk[k] = ((long long)want + 511) & ~511ll;
    if (const char* e = cs::cfg("CS_SPLIT1_SHRINK")) capk[k] = std::max<long long>(512, (capk[k] / std::max(1, atoi(e))) & ~511ll);  // (tests: the give-up route)
  }

  // ---- tiles, grid
  // (regions: every token byte once, up to 15 bytes of padding per column, 20 bytes of OR slack; lanes without a token OR
  // zeros at 20 x lane)
  const int cap_out = (int)((std::max<int64_t>(span + 16 * ncap + 48, 64 * 20 + 32) + 15) & ~(int64_t)15);
  if (cap_out > 8192 || span + 15 + 16 > cstile::kPfBytes) return 0;  // (16 words of region-start bits; what one prefetch holds)
  int wave_bytes = 16 + cap_in + 32 + cap_out + kTables1;
  wave_bytes = std::max(wave_bytes, 2048);  // (the scanner workgroups keep their rings in the first wave's share)
  wave_bytes = (wave_bytes + 15) & ~15;
  const size_t lds = 288 + (size_t)wave_bytes * (kThreads1 / 64);
  if (lds > 160 * 1024) return 0;
  typedef void (*Kernel)(Emit5Args);
  static const Kernel kerns[3] = {k_split_emit5<0, false>, k_split_emit5<1, false>, k_split_emit5<2, false>};
  const Kernel kern = plain_walk ? k_split_emit5<0, true> : kerns[mode];
  if (lds > 48 * 1024) CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int dev = 0, cus = 0, per_cu = 0;
  CS_HIP(hipGetDevice(&dev));
  CS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  CS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kThreads1, lds));
  if (per_cu < 1) return 0;
  constexpr int kWavesPerGroup = kThreads1 / 64;
  const int scan_blocks = (npairs + kPairsPerGroup - 1) / kPairsPerGroup;
  const int64_t resident = (int64_t)cus * per_cu;
  const int64_t want_workers = (nsub + kWavesPerGroup - 1) / kWavesPerGroup;  // (at least a sub-tile per wave)
  const int64_t workers = std::max<int64_t>(1, std::min<int64_t>(resident - scan_blocks, want_workers));
  if (resident - scan_blocks < 1) return 0;

  // ---- buffers
  const long long stride = (nsub + 256 + 255) & ~255ll;  // (a step is read whole)
  const size_t status_bytes = sizeof(u64) * npairs * stride, excl_bytes = status_bytes;
  Buf stat = dev_alloc(status_bytes + excl_bytes + 1024 + sizeof(long long) * 2 * npairs, s);
  CS_HIP(hipMemsetAsync(stat->p, 0, status_bytes + excl_bytes + 1024, s));
  u64* status = ptr<u64>(stat);
  u64* excl = status + (long long)npairs * stride;
  unsigned* ctl = reinterpret_cast<unsigned*>(excl + (long long)npairs * stride);  // 1 KiB: error word, column count, statistics
  long long* totals = reinterpret_cast<long long*>(reinterpret_cast<uint8_t*>(ctl) + 1024);
  Buf tilecols = dev_alloc((size_t)nsub + 64, s);
  unsigned init[2] = {0u, (unsigned)seen};
  CS_HIP(hipMemcpyAsync(ctl, init, sizeof(init), hipMemcpyHostToDevice, s));
  std::vector<std::unique_ptr<cs_column>> out;
  std::vector<ColOut5> outs(ncap);
  for (int k = 0; k < ncap; ++k) {
    auto c = std::make_unique<cs_column>();
    c->rows = rows;
    c->chars = dev_alloc((size_t)capk[k], s);
    c->offsets32 = dev_alloc(sizeof(int32_t) * (rows + 1), s);
    c->validity = dev_alloc(validity_bytes(rows), s);
    outs[k] = ColOut5{ptr<uint8_t>(c->chars), ptr<int32_t>(c->offsets32), ptr<uint8_t>(c->validity), capk[k]};
    out.push_back(std::move(c));
  }
  Buf d_outs = dev_alloc(sizeof(ColOut5) * ncap, s);
  CS_HIP(hipMemcpyAsync(d_outs->p, outs.data(), sizeof(ColOut5) * ncap, hipMemcpyHostToDevice, s));

  Emit5Args ea{view_of(col), dpat, d64, dlen, tokens, reverse ? 1 : 0, ncap, npairs, nsub, stride, cap_in, cap_out, wave_bytes, scan_blocks,
               ptr<const ColOut5>(d_outs), status, excl, ctl, totals, ptr<uint8_t>(tilecols), nullptr, nullptr,
               cs::cfg_int("CS_SPLIT_DEBUG", 0)};
#if defined(CS_PHASE_PROF)
  Buf profbuf = dev_alloc(128, s);
  CS_HIP(hipMemsetAsync(profbuf->p, 0, 128, s));
  ea.prof = ptr<unsigned long long>(profbuf);
#endif
  Buf tracebuf;
  const long long ntrace = (nsub >> 10) + 1;
  if (cs::cfg("CS_SPLIT1_TRACE")) {
    tracebuf = dev_alloc(sizeof(unsigned long long) * 4 * ntrace, s);
    CS_HIP(hipMemsetAsync(tracebuf->p, 0, sizeof(unsigned long long) * 4 * ntrace, s));
    ea.trace = ptr<unsigned long long>(tracebuf);
  }
  {
    ProfScope ps("k_split_emit", s);
    hipLaunchKernelGGL(kern, dim3((unsigned)(scan_blocks + workers)), dim3(kThreads1), lds, s, ea);
  }
  CS_HIP(hipGetLastError());
  struct HostCtl {
    unsigned ctl[12];
    long long totals[kCols1];
  };
  HostCtl* hc = (HostCtl*)pinned_scratch(sizeof(HostCtl));
  CS_HIP(hipMemcpyAsync(hc->ctl, ctl, sizeof(hc->ctl), hipMemcpyDeviceToHost, s));
  CS_HIP(hipMemcpyAsync(hc->totals, totals, sizeof(long long) * 2 * npairs, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (cs::cfg("CS_SPLIT1_STATS"))
    fprintf(stderr, "split single pass: error %u columns %u (sample %d, provisioned %d) | worker spins %u | scanner fetches %u (empty %u) of %d scanners, %lld windows | mean gap between a wave's sub-tiles %.0f | grid %d + %lld x %d lds %zu\n",
            hc->ctl[0], hc->ctl[1], seen, ncap, hc->ctl[4], hc->ctl[5], hc->ctl[6], npairs, (long long)((nsub + 63) / 64),
            (double)*reinterpret_cast<unsigned long long*>(hc->ctl + 8) / (double)std::max(1u, hc->ctl[7]), scan_blocks, (long long)workers, kThreads1, lds);
#if defined(CS_PHASE_PROF)
  {
    unsigned long long ph[10];
    CS_HIP(hipMemcpy(ph, ea.prof, sizeof(ph), hipMemcpyDeviceToHost));
    const double it = (double)nsub;
    fprintf(stderr, "emit5 cycles/sub-tile: stage %.0f masks %.0f walk+scan %.0f publish %.0f | wait %.0f offsets %.0f column-lanes %.0f flush %.0f | assemble %.0f rest %.0f\n",
            ph[0] / it, ph[1] / it, ph[2] / it, ph[3] / it, ph[4] / it, ph[5] / it, ph[6] / it, ph[7] / it, ph[8] / it, ph[9] / it);
  }
#endif
  if (tracebuf) {
    std::vector<unsigned long long> tr(4 * ntrace);
    CS_HIP(hipMemcpy(tr.data(), tracebuf->p, sizeof(unsigned long long) * 4 * ntrace, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (long long i = 0; i < ntrace; ++i)
      if (tr[4 * i]) t0 = std::min(t0, tr[4 * i]);
    fprintf(stderr, "trace (us since the first publish; every 1024th sub-tile): tile published scanned start-of-finish prefix-seen\n");
    for (long long i = 0; i < ntrace; i += std::max<long long>(1, ntrace / 48))
      fprintf(stderr, "trace %8lld %9.1f %9.1f %9.1f %9.1f\n", i << 10, (double)(tr[4 * i] - t0) / 100.0, (double)(tr[4 * i + 1] - t0) / 100.0,
              (double)(tr[4 * i + 3] - t0) / 100.0, (double)(tr[4 * i + 2] - t0) / 100.0);
  }
  if (hc->ctl[0] != 0) {
    static const char* const what[] = {"split (single pass: wait)", "split (single pass: a column outgrew its estimate)", "split (single pass: a column of 2 GiB)",
                                       "split (single pass: more columns than provisioned)", "split (single pass: scanner)"};
    int bit = 0;
    while (bit < 4 && !(hc->ctl[0] & (1u << bit))) ++bit;
    note_fallback(what[bit]);
    return -1;
  }
  const int ncols = (int)hc->ctl[1];
  if (ncols > seen) {  // columns that appeared after sub-tiles had been written without them
    FixupArgs fa{rows, nsub, stride, ncols, ptr<const uint8_t>(tilecols), excl, ptr<const ColOut5>(d_outs)};
    ProfScope ps("k_split_fixup", s);
    hipLaunchKernelGGL(k_split_fixup, dim3((unsigned)std::min<int64_t>((nsub + 255) / 256, 4096)), dim3(256), 0, s, fa);
    CS_HIP(hipGetLastError());
    CS_HIP(hipStreamSynchronize(s));  // `d_outs` / `stat` lifetime
  }
  for (int k = 0; k < ncols; ++k) {
    out[k]->nbytes = hc->totals[k];
    cols.push_back(std::move(out[k]));
  }
  return 1;
}

}  // namespace cs
