// Row-tile machinery shared by the single-pass kernels (device only).
//
// A workgroup of 256 threads owns a TILE of 256 consecutive rows at a time:
//   1. the tile's contiguous span of the chars buffer is staged into LDS with
//      coalesced 16-byte loads (every input byte crosses HBM once);
//   2. each thread runs the per-row logic on its row out of LDS;
//   3. output sizes become offsets by a workgroup scan + a decoupled look-back
//      across tiles (one pass over the data, no separate size kernel);
//   4. the rows' outputs -- contiguous in the output buffer, because the rows
//      are consecutive -- are assembled in LDS and flushed with coalesced
//      16-byte stores.
// Tiles are handed out by an atomic ticket, so a tile only ever waits for tiles
// that have already started (forward progress without assuming dispatch order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_utils.h"

// per-phase cycle accounting of the persistent kernels (instrumented builds only: make prof)
#if defined(CS_PHASE_PROF)
#define CS_PHASE_MARK(k)                                  \
  do {                                                    \
    const unsigned long long t_ = __builtin_readcyclecounter(); \
    phase_acc[k] += t_ - phase_t;                         \
    phase_t = t_;                                         \
  } while (0)
#else
#define CS_PHASE_MARK(k) \
  do {                   \
  } while (0)
#endif

namespace cstile {

// Pointers that went through readlane / integer arithmetic lose their address space and
// would be accessed with flat instructions; these casts put them back in global memory.
template <class T>
using gptr = __attribute__((address_space(1))) T*;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // plain vector: assignable through any address space
template <class T>
__device__ __forceinline__ gptr<T> as_global(T* p) {
  return (gptr<T>)p;
}
// 16-byte global accesses of the streaming kernels: NON-TEMPORAL -- every byte of these streams is touched once (the
// headline step 11.38-11.41 -> 11.14 ms with both, 11.28-11.29 with the stores alone, A / B on one box: profiles/r06/nt_ab.txt;
// the box's own streaming rates: read-only 6.5 -> 7.1 TB/s with nt loads, emit's store shape 5.0 -> 5.3 with nt stores:
// profiles/r06/stream_rate.txt).  -DCS_PLAIN_STREAM restores the default cache policy (measurement).
#if !defined(CS_PLAIN_STREAM)
#define CS_NT_STORES 1
#define CS_NT_LOADS 1
#endif
__device__ __forceinline__ void gstore16(gptr<u32x4> p, u32x4 v) {
#if defined(CS_NT_STORES)
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ uint4 gload16_stream(const uint4* p) {
#if defined(CS_NT_LOADS)
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}

constexpr int kTileRows = 256;
constexpr unsigned long long kFlagAgg = 1ull << 62, kFlagInc = 2ull << 62, kValMask = (1ull << 62) - 1;

typedef unsigned long long u64;

__device__ __forceinline__ u64 status_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void status_store(u64* p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Decoupled look-back, run by ONE wave (all 64 lanes).  `status` has one word
// per tile (stride `stride` words between consecutive tiles), zeroed before the
// launch.  Publishes this tile's aggregate, then its inclusive prefix; returns
// the exclusive prefix (sum of the aggregates of all earlier tiles).
// Spins are bounded: after kSpinTicks without progress the function gives
// up and returns -1 (the caller raises an error flag; the host then recomputes
// the column with the two-pass kernels).
// The bound is TIME, not a poll count (a poll's cost varies with what shares the device: a co-tenant or a CU mask made a
// count of polls anything from a blink to minutes): kSpinTicks of the constant-rate 100 MHz counter (wall_clock64), looked
// at every 64 polls.
constexpr unsigned long long kSpinTicks = 25ull * 1000 * 1000;  // a quarter of a second WITHOUT PROGRESS (the clocks are reset whenever
                                                             // something arrives): far beyond any honest wait
struct SpinClock {
  unsigned long long t0 = 0;
  int polls = 0;
  __device__ __forceinline__ bool expired() {
    if ((++polls & 63) != 0) return false;
    const unsigned long long now = wall_clock64();
    if (t0 == 0) {
      t0 = now;
      return false;
    }
    return now - t0 > kSpinTicks;
  }
  __device__ __forceinline__ void reset() {
    t0 = 0;
    polls = 0;
  }
};
__device__ __forceinline__ long long lookback(u64* status, long long stride, long long tile, long long aggregate) {
  const int lane = threadIdx.x & 63;
  u64* mine = status + tile * stride;
  if (tile == 0) {
    if (lane == 0) status_store(mine, kFlagInc | ((u64)aggregate & kValMask));
    return 0;
  }
  if (lane == 0) status_store(mine, kFlagAgg | ((u64)aggregate & kValMask));
  long long excl = 0;
  long long t = tile - 1;
  SpinClock clock;
  for (;;) {
    long long idx = t - lane;
    u64 v = idx >= 0 ? status_load(status + idx * stride) : kFlagInc;
    unsigned flag = (unsigned)(v >> 62);
    u64 not_ready = __ballot(flag == 0);
    u64 inc = __ballot(flag == 2);
    int first_inc = inc ? __builtin_ctzll(inc) : 64;
    u64 needed = first_inc < 63 ? ((2ull << first_inc) - 1) : ~0ull;
    if (not_ready & needed) {
      if (clock.expired()) return -1;
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    long long part = (lane <= first_inc) ? (long long)(v & kValMask) : 0;
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    excl += part;
    if (first_inc < 64) break;
    t -= 64;
    clock.reset();  // (progress: the bound is time WITHOUT progress)
  }
  if (lane == 0) status_store(mine, kFlagInc | ((u64)(excl + aggregate) & kValMask));
  return excl;
}

// Stages bytes [g0, g1) of `chars` into `lds` (16-byte aligned LDS buffer) so
// that LDS index i holds the byte at absolute address (A0 + i), A0 = address of
// chars[g0] rounded down to 16.  Returns the LDS index of chars[g0].
__device__ __forceinline__ int stage_in(const uint8_t* chars, long long g0, long long g1, uint8_t* lds) {
  const uintptr_t first = (uintptr_t)(chars + g0);
  const uintptr_t a0 = first & ~(uintptr_t)15;
  const int span = (int)((uintptr_t)(chars + g1) - a0);
  for (int i = threadIdx.x * 16; i < span; i += blockDim.x * 16) {
    uint4 v = *reinterpret_cast<const uint4*>(a0 + i);
    *reinterpret_cast<uint4*>(lds + i) = v;
  }
  return (int)(first - a0);
}

// Flushes `total` bytes assembled in `lds` to global memory at `dst`; the
// caller placed byte k of the output at LDS index ((uintptr_t)dst & 15) + k, so
// whole 16-byte chunks line up in both address spaces.
__device__ __forceinline__ void flush_out(uint8_t* dst, int total, const uint8_t* lds) {
  const int lead = (int)((uintptr_t)dst & 15);
  uint8_t* a0 = dst - lead;  // 16-aligned
  const int end = lead + total;
  for (int i = threadIdx.x * 16; i < end; i += blockDim.x * 16) {
    if (i >= lead && i + 16 <= end) {
      *reinterpret_cast<uint4*>(a0 + i) = *reinterpret_cast<const uint4*>(lds + i);
    } else {
      for (int k = 0; k < 16; ++k) {
        int j = i + k;
        if (j >= lead && j < end) a0[j] = lds[j];
      }
    }
  }
}

// Wave-cooperative flush of `total` bytes from an LDS region to global memory.
// `lds` points at the region's 16-byte aligned start; the caller placed output
// byte k at lds[lead + k] with lead = (address of dst) & 15, so whole 16-byte
// chunks line up in both address spaces.  Head and tail bytes (up to 15 each) are
// stored one byte per LANE (two store instructions), never in a per-lane loop.
__device__ __forceinline__ void wave_flush(uint8_t* dst, int total, const uint8_t* lds, int lead, int lane) {
  gptr<uint8_t> a0 = as_global(dst - lead);  // 16-byte aligned
  const int end = lead + total;
  const int first_full = (lead + 15) & ~15;       // first chunk boundary at or after the start
  const int last_full = end & ~15;                // end of the last whole chunk
  if (first_full >= last_full) {                  // no whole chunk: bytes only
    for (int j = lead + lane; j < end; j += 64) a0[j] = lds[j];
    return;
  }
  for (int i = first_full + lane * 16; i < last_full; i += 64 * 16)
    gstore16((gptr<u32x4>)(a0 + i), *reinterpret_cast<const u32x4*>(lds + i));
  // head bytes on lanes 0..15, tail bytes on lanes 16..31: one predicated byte store for both
  const int j = lane < 16 ? lead + lane : last_full + lane - 16;
  const bool ok = lane < 16 ? j < first_full : (lane < 32 && j < end);
  if (ok) a0[j] = lds[j];
}

// Per-thread copy inside LDS (ascending: a move down inside one buffer is safe).  gfx950 takes DS accesses at
// any alignment, so the copy is 16 bytes per trip through two 8-byte accesses and a 8 / 4 / 2 / 1 tail, with no
// alignment prologue.  Both buffers are given as (base, byte index); the pointers are never turned into integers
// (which would demote the LDS accesses to flat ones).
typedef uint32_t lds_u32x4u __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned long long lds_u64u __attribute__((aligned(1)));
typedef uint32_t lds_u32u __attribute__((aligned(1)));
typedef uint16_t lds_u16u __attribute__((aligned(1)));
__device__ __forceinline__ void lds_copy(uint8_t* dbase, int di, const uint8_t* sbase, int si, int n) {
  uint8_t* d = dbase + di;
  const uint8_t* s = sbase + si;
  int i = 0;
  for (; i + 16 <= n; i += 16) *reinterpret_cast<lds_u32x4u*>(d + i) = *reinterpret_cast<const lds_u32x4u*>(s + i);
  if (n & 8) {
    *reinterpret_cast<lds_u64u*>(d + i) = *reinterpret_cast<const lds_u64u*>(s + i);
    i += 8;
  }
  if (n & 4) {
    *reinterpret_cast<lds_u32u*>(d + i) = *reinterpret_cast<const lds_u32u*>(s + i);
    i += 4;
  }
  if (n & 2) {
    *reinterpret_cast<lds_u16u*>(d + i) = *reinterpret_cast<const lds_u16u*>(s + i);
    i += 2;
  }
  if (n & 1) d[i] = s[i];
}

// The first `len` (<= 16) bytes of `v` to an LDS position of any alignment: at most five stores of 16 / 8 / 4 / 2 / 1
// bytes, exactly the bytes asked for (neighbouring lanes write the bytes next to them).
__device__ __forceinline__ void lds_put16(uint8_t* dp, lds_u32x4u v, int len) {
  if (len >= 16) {
    *reinterpret_cast<lds_u32x4u*>(dp) = v;
    return;
  }
  uint32_t t0 = v.x, t1 = v.y;
  if (len & 8) {
    *reinterpret_cast<lds_u64u*>(dp) = ((unsigned long long)v.y << 32) | v.x;
    dp += 8;
    t0 = v.z;
    t1 = v.w;
  }
  if (len & 4) {
    *reinterpret_cast<lds_u32u*>(dp) = t0;
    dp += 4;
    t0 = t1;
  }
  if (len & 2) {
    *reinterpret_cast<lds_u16u*>(dp) = (uint16_t)t0;
    dp += 2;
    t0 >>= 16;
  }
  if (len & 1) *dp = (uint8_t)t0;
}

// lds_copy for the assembly of rows out of short pieces: ONE round trip through LDS for a piece of up to 16 bytes (a
// 16-byte read at any alignment -- it may run up to 15 bytes past the piece: the caller's source has that slack --, then
// lds_put16's stores, none waiting for another), longer pieces in 16-byte steps with the next read in flight while
// the previous one is stored, the last step overlapping the one before instead of a 8 / 4 / 2 / 1 ladder of dependent
// read-store pairs.  Writes exactly n bytes.
__device__ __forceinline__ void lds_copy_ov(uint8_t* dbase, int di, const uint8_t* sbase, int si, int n) {
  if (n <= 0) return;
  uint8_t* d = dbase + di;
  const uint8_t* s = sbase + si;
  lds_u32x4u v = *reinterpret_cast<const lds_u32x4u*>(s);
  if (n < 16) {
    lds_put16(d, v, n);
    return;
  }
  int i = 0;
  while (i + 16 < n) {
    const int j = min(i + 16, n - 16);
    const lds_u32x4u w = *reinterpret_cast<const lds_u32x4u*>(s + j);
    *reinterpret_cast<lds_u32x4u*>(d + i) = v;
    v = w;
    i = j;
  }
  *reinterpret_cast<lds_u32x4u*>(d + i) = v;
}

// Copy of a SHORT run (tokens, replacement text): the first 8 bytes go through two
// funnel-shifted source dwords and straight-line predicated byte stores, longer
// runs fall back to lds_copy for the rest.
__device__ __forceinline__ void lds_copy_short(uint8_t* dbase, int di, const uint8_t* sbase, int si, int n) {
  if (n <= 0) return;
  const uint32_t* sp = reinterpret_cast<const uint32_t*>(sbase) + (si >> 2);
  const unsigned sh = (unsigned)(si & 3);
  const uint32_t w0 = sp[0], w1 = sp[1], w2 = sp[2];
  const uint32_t a = sh ? __builtin_amdgcn_alignbyte(w1, w0, sh) : w0;  // source bytes 0..3
  const uint32_t b = sh ? __builtin_amdgcn_alignbyte(w2, w1, sh) : w1;  // source bytes 4..7
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < n) dbase[di + j] = (uint8_t)(a >> (8 * j));
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 + j < n) dbase[di + 4 + j] = (uint8_t)(b >> (8 * j));
  if (n > 8) lds_copy(dbase, di + 8, sbase, si + 8, n - 8);
}


// ---- persistent sub-tile stream with register prefetch ------------------------------
// A persistent wave walks sub-tiles wid, wid + W, wid + 2W, ... (W = waves in the
// grid, all co-resident).  While it works on one sub-tile out of LDS, the chars of
// its next sub-tile are already in flight into registers (kPfChunks x 16 bytes per
// lane) and the offsets of the one after that are being fetched, so every wave
// keeps several KB of HBM reads outstanding at all times instead of starting each
// sub-tile with two dependent round trips.
constexpr int kPfChunks = 6;
constexpr int kPfBytes = kPfChunks * 1024;  // largest (lead + span) one prefetch can hold

struct TileOffs {
  long long o0, o1;  // offsets[r0 + lane], offsets[r0 + lane + 1] (clamped to the sub-tile)
};
template <int N>
struct TileCharsT {
  uint4 v[N];
};
typedef TileCharsT<kPfChunks> TileChars;
__device__ __forceinline__ long long rl64(long long v, int k) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), k);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), k);
  return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ TileOffs load_tile_offsets(const int64_t* offsets, long long rows, long long sub, int lane) {
  const long long r0 = sub * 64;
  const int nrows = (int)min(64ll, rows - r0);
  TileOffs t;
  t.o0 = offsets[r0 + min(lane, nrows)];
  t.o1 = offsets[r0 + min(lane + 1, nrows)];
  return t;
}
// the same for tiles of R <= 64 rows (lanes beyond the tile's rows repeat its end offset)
__device__ __forceinline__ TileOffs load_tile_offsets_r(const int64_t* offsets, long long rows, long long tile, int R, int lane) {
  const long long r0 = tile * R;
  const int nrows = (int)min((long long)R, rows - r0);
  TileOffs t;
  t.o0 = offsets[r0 + min(lane, nrows)];
  t.o1 = offsets[r0 + min(lane + 1, nrows)];
  return t;
}
// issues the loads of bytes [g0 - lead, g1) of `chars` (lead = distance to the previous
// 16-byte boundary); nothing waits on them here
template <int N>
__device__ __forceinline__ void issue_chars(const uint8_t* chars, long long g0, long long g1, int lane, TileCharsT<N>& c) {
  const int lead = (int)((uintptr_t)(chars + g0) & 15);
  const uint8_t* src = chars + (g0 - lead);
  const long long want = g1 - g0 + lead;
  const int span = (int)(want < (long long)(N * 1024) ? want : (long long)(N * 1024));
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int i = j * 1024 + lane * 16;
    if (i < span) c.v[j] = gload16_stream(reinterpret_cast<const uint4*>(src + i));
  }
}
template <int N>
__device__ __forceinline__ void stage_chars(uint8_t* lds_in, int span, int lane, const TileCharsT<N>& c) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int i = j * 1024 + lane * 16;
    if (i < span) *reinterpret_cast<uint4*>(lds_in + i) = c.v[j];
  }
}
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Decoupled look-back in two halves so that independent work (assembling the output
// rows in LDS) runs between publishing this sub-tile's aggregate and needing the
// predecessors': lookback_begin publishes and issues the first poll, lookback_end
// consumes it (re-polling only if some predecessor had not published yet).
__device__ __forceinline__ void lookback_publish(u64* status, long long tile, long long aggregate, int lane) {
  if (lane == 0) status_store(status + tile, (tile == 0 ? kFlagInc : kFlagAgg) | ((u64)aggregate & kValMask));
}
__device__ __forceinline__ u64 lookback_poll(const u64* status, long long tile, int lane) {
  const long long idx = tile - 1 - lane;
  return idx >= 0 ? status_load(status + idx) : kFlagInc;
}
__device__ __forceinline__ u64 lookback_begin(u64* status, long long tile, long long aggregate, int lane) {
  lookback_publish(status, tile, aggregate, lane);
  return lookback_poll(status, tile, lane);
}
#if defined(CS_PHASE_PROF)
static __device__ unsigned long long g_lb_stats[4];  // calls, windows walked, re-polls of a window, -
// time stamps (100 MHz) of every 1024th tile: published, prefix stored by the scanners, start of its finish, prefix seen
static __device__ unsigned long long* g_tile_trace;
#define CS_TILE_TRACE(tile, slot)                                                                                       \
  do {                                                                                                                  \
    if (cstile::g_tile_trace && ((tile) & 1023) == 0) cstile::g_tile_trace[((tile) >> 10) * 4 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define CS_TILE_TRACE(tile, slot) \
  do {                            \
  } while (0)
#endif
__device__ __forceinline__ long long lookback_end(u64* status, long long tile, long long aggregate, u64 first, int lane,
                                                  unsigned long long* acc = nullptr) {
  if (tile == 0) return 0;
  long long excl = 0;
  long long t = tile - 1;
  int spins = 0;
  SpinClock clock;
  unsigned part = 0;
  u64 v = first;
#if defined(CS_PHASE_PROF)
  int windows = 1;
#endif
  for (;;) {
    const unsigned flag = (unsigned)(v >> 62);
    const u64 not_ready = __ballot(flag == 0);
    const u64 inc = __ballot(flag == 2);
    const int first_inc = inc ? __builtin_ctzll(inc) : 64;
    const u64 needed = first_inc < 63 ? ((2ull << first_inc) - 1) : ~0ull;
    if (not_ready & needed) {
      ++spins;
      if (clock.expired()) return -1;
      __builtin_amdgcn_s_sleep(2);
      const long long idx = t - lane;
      v = idx >= 0 ? status_load(status + idx) : kFlagInc;
      continue;
    }
    // aggregates of 64-row sub-tiles fit 32 bits: every lane keeps the sum of its column of the
    // windows walked and ONE reduction follows the walk (the kernel is bound by its instruction
    // count, and the walk covers four windows on average); the one inclusive prefix that ends
    // the walk is 64 bits wide and is read from its lane directly
    part += (lane < first_inc) ? (unsigned)(v & 0xffffffffull) : 0u;
    if (first_inc < 64) {
      excl = rl64((long long)(v & kValMask), first_inc);
      break;
    }
    t -= 64;
    clock.reset();  // (progress: the bound is time WITHOUT progress)
#if defined(CS_PHASE_PROF)
    ++windows;
#endif
    const long long idx = t - lane;
    v = idx >= 0 ? status_load(status + idx) : kFlagInc;
  }
  excl += (unsigned)csdev::wave_reduce_sum((int)part);
  if (lane == 0) status_store(status + tile, kFlagInc | ((u64)(excl + aggregate) & kValMask));
  (void)spins;
#if defined(CS_PHASE_PROF)
  if (acc) {
    acc[0] += 1;
    acc[1] += (unsigned long long)windows;
    acc[2] += (unsigned long long)spins;
  }
#endif
  return excl;
}


// ---- prefix resolution by a dedicated wave ---------------------------------------------------
// The look-back above costs a tile several DEPENDENT device-scope round trips (it walks back window by
// window until it meets an inclusive prefix: 4.3 windows per tile on the replace kernel).  Here ONE
// wave of the persistent grid does nothing but turn the published aggregates into exclusive prefixes,
// in tile order, several windows of 64 tiles per round trip; a tile then needs a single load of its own
// prefix.  `status[t]` = flag | aggregate (any non-zero flag counts as published), `excl[t]` = kFlagInc |
// sum of the aggregates of tiles 0 .. t-1; both zeroed before the launch.  Returns false on a timeout
// (the caller raises the error word; tiles waiting for their prefix watch that word and give up).
constexpr int kScanBatch = 16;  // windows in flight (1024 tiles per round trip: the replace kernel takes ~200 tiles per microsecond)
// `buf`: wave-private LDS, kScanBatch x 512 bytes.  `status` must be readable for kScanBatch windows past its last
// tile (whatever lies there is ignored).  The loop that consumes the windows is kept ROLLED (the fetched words
// pass through `buf`) and the fetch is sixteen loads off one address: this wave runs code nobody else on its CU
// runs, and a body of several KB was evicted from the instruction cache between trips (measured: one window per
// trip, 15 ms instead of 8).
__device__ __forceinline__ bool prefix_scanner(const u64* status, u64* excl, long long ntiles, int lane, u64* buf) {
  __builtin_amdgcn_s_setprio(3);  // every tile of the launch waits for this wave: it goes first on its SIMD
  gptr<u64> ex = as_global(excl);
  long long base = 0;
  long long w = 0;  // next window
  SpinClock clock;
  while (w * 64 < ntiles) {
    // one round trip fetches the next kScanBatch windows; the leading ones that are complete are consumed, the
    // rest is fetched again together with what follows (at the frontier of the publishing waves this keeps
    // pace with them instead of paying a round trip per window)
    gptr<const u64> at = as_global(status) + (w * 64 + lane);
    u64 v[kScanBatch];
#pragma unroll
    for (int j = 0; j < kScanBatch; ++j) v[j] = __hip_atomic_load(at + j * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < kScanBatch; ++j) buf[j * 64 + lane] = v[j];
    int done = 0;
#pragma unroll 1
    for (int j = 0; j < kScanBatch; ++j) {
      const long long idx = (w + j) * 64 + lane;
      if ((w + j) * 64 >= ntiles) break;
      const u64 x = buf[j * 64 + lane];
      const bool in = idx < ntiles;
      // A tile's prefix needs its predecessors only: the leading published tiles of an incomplete window get
      // theirs at once (and the first unpublished one too).  Waiting for whole windows deadlocks small grids, where
      // one wave holds several tiles of the same window and finishes a tile only after the scan of its next.
      const u64 missing = __ballot(in && (x >> 62) == 0);
      const int ready = missing ? __builtin_ctzll(missing) : 64;  // lanes below are published
      const int val = in && lane < ready ? (int)(unsigned)(x & 0xffffffffull) : 0;  // aggregates of 64-row sub-tiles fit 31 bits
      const int inc = csdev::wave_inclusive_scan(val);
      if (in && lane <= ready) __hip_atomic_store(ex + idx, kFlagInc | ((u64)(base + inc - val) & kValMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (missing) break;  // (the window is visited again; what was stored is stored again, unchanged)
      base += __builtin_amdgcn_readlane(inc, 63);
      ++done;
    }
    w += done;
    if (done == 0) {  // (an incomplete first window may still have made progress: the limit is far beyond any honest wait)
      if (clock.expired()) return false;
      __builtin_amdgcn_s_sleep(1);
    } else {
      clock.reset();
    }
  }
  return true;
}
// ---- the same by a TEAM of waves of one workgroup ---------------------------------------------------
// prefix_scanner above is one wave walking 64 tiles a step: fetch (a device-scope round trip), scan, store, next -- about
// 140-200 tiles per microsecond, which is what the replace kernel takes tiles at: its waves spent 30 % of their cycles
// waiting for prefixes (profiles/r04: lookback 7.6 k of 25 k cycles per sub-tile).  Here a lane takes FOUR consecutive
// tiles (two 16-byte loads, 32 contiguous bytes a lane), a STEP covers 256 tiles with one wave scan, and `team` waves of
// the workgroup share the steps round-robin: fetching, scanning and storing run in parallel, only the running sum is
// handed from step to step through an LDS ring, a few hundred cycles a link.  Until a step is complete its wave polls
// it; every tile up to and including the first unpublished one gets its prefix as soon as the step's base is known
// (a tile's prefix needs its predecessors only -- small grids hold several tiles of a step in one wave).
struct TeamRing {            // slot s & 7 holds the sum in front of step s once tag == s + 1
  unsigned long long val[8];
  unsigned tag[8];
};
__device__ __forceinline__ void team_ring_init(TeamRing* ring, int slot) {  // by eight threads, a barrier behind
  ring->val[slot] = 0ull;
  ring->tag[slot] = slot == 0 ? 1u : 0u;  // (nothing in front of step 0)
}
__device__ __forceinline__ int wave_scan_fused32(int v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
// wave `h` of `team`: steps h, h + team, ...  Returns 0 when done, 1 on a timeout, 2 when it left because somebody else
// had raised `error` (the launch is lost anyway; the word is not touched).
__device__ __forceinline__ int prefix_scanner_team(const u64* status, u64* excl, long long ntiles, int lane, int h, int team, TeamRing* ring,
                                                    const unsigned* error) {
  gptr<u64> ex = as_global(excl);
  for (long long s = h; s * 256 < ntiles; s += team) {
    const long long t0 = s * 256 + 4 * lane;  // the lane's first tile
    const u64* at = status + t0;
    bool in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) in[i] = t0 + i < ntiles;
    bool have_base = false;
    u64 base = 0;
    int delivered = -1;  // leading tiles of the step that have their prefix
    SpinClock clock;
    int idle = 0;
    for (;;) {
      u32x4 q0, q1;
      asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(q0), "=&v"(q1)
                   : "v"(at)
                   : "memory");
      const u64 x[4] = {((u64)q0.y << 32) | q0.x, ((u64)q0.w << 32) | q0.z, ((u64)q1.y << 32) | q1.x, ((u64)q1.w << 32) | q1.z};
      bool pub[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pub[i] = !in[i] || (x[i] >> 62) != 0;  // (beyond the last tile: nothing to wait for, nothing to add)
      const int r = pub[0] ? (pub[1] ? (pub[2] ? (pub[3] ? 4 : 3) : 2) : 1) : 0;  // leading published tiles of the lane
      const u64 missing = __ballot(r < 4);
      const int first = missing ? __builtin_ctzll(missing) : 64;  // lanes below are complete
      const int lead = missing ? 4 * first + __builtin_amdgcn_readlane(r, first) : 256;
      u64 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = in[i] && (lane < first || (lane == first && i < r)) ? (x[i] & kValMask) : 0ull;
      const u64 p1 = v[0], p2 = p1 + v[1], p3 = p2 + v[2], tot = p3 + v[3];
      u64 incl = tot;
      if (lead > delivered) {
        // (aggregates of 64-row sub-tiles are small: one 32-bit scan; anything bigger takes the 64-bit form)
        if (__any(tot >> 22)) incl = (u64)csdev::wave_inclusive_scan((long long)tot);
        else incl = (u64)(unsigned)wave_scan_fused32((int)(unsigned)tot);
      }
      // the sum in front of the step: wait for it only when the step is complete (else poll the step again meanwhile)
      while (!have_base) {
        if (__hip_atomic_load(&ring->tag[s & 7], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)(s + 1)) {
          base = __hip_atomic_load(&ring->val[s & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          have_base = true;
          break;
        }
        if (missing) break;
        if (clock.expired()) return 1;
        if ((++idle & 1023) == 0 && (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) != 0) return 2;
      }
      if (have_base && !missing && lane == 0) {  // complete: hand the sum on FIRST (the next step's wave waits for nothing else)
        const u64 next = base + rl64((long long)incl, 63);
        __hip_atomic_store(&ring->val[(s + 1) & 7], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ring->tag[(s + 1) & 7], (unsigned)(s + 2), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (have_base && lead > delivered) {
        const u64 e0 = base + incl - tot;  // in front of the lane's first tile
        const u64 o[4] = {kFlagInc | (e0 & kValMask), kFlagInc | ((e0 + p1) & kValMask), kFlagInc | ((e0 + p2) & kValMask), kFlagInc | ((e0 + p3) & kValMask)};
        if (lane < first && in[3]) {  // a complete lane: 32 contiguous bytes
          const u32x4 lo = {(uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32)};
          const u32x4 hi = {(uint32_t)o[2], (uint32_t)(o[2] >> 32), (uint32_t)o[3], (uint32_t)(o[3] >> 32)};
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(excl + t0), "v"(lo), "v"(hi) : "memory");
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (in[i] && (lane < first || (lane == first && i <= r))) __hip_atomic_store(ex + (t0 + i), o[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (in[0] && lane <= first) CS_TILE_TRACE(t0, 1);
        delivered = lead;
        clock.reset();
      }
      if (have_base && !missing) break;
      if (clock.expired()) return 1;
      // (somebody gave up WAITING -- bit 0: the launch is lost.  Other bits, "out of room" for one, are raised by waves that
      // go on publishing or have published all they hold: the prefixes keep coming, or the waiters would turn a clean
      // "once more, roomier" into a fallback)
      if ((++idle & 255) == 0 && (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) != 0) return 2;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  return 0;
}

// a tile's side: `first` is an early load of excl[tile]; polls until the prefix is there.  -1 = timeout / launch failed.
__device__ __forceinline__ long long prefix_wait(const u64* excl, long long tile, u64 first, const unsigned* error, int lane) {
  u64 v = first;
  int spins = 0;
  SpinClock clock;
  while ((v >> 62) == 0) {
    ++spins;
    if (clock.expired()) return -1;
    // (bit 0 = somebody gave up waiting or the scanners stopped: no more prefixes.  Other bits -- "out of room" -- leave the
    // chain intact, and a waiter that gave up on them would turn the host's clean "once more, roomier" into a fallback)
    if ((spins & 255) == 0 && (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) != 0) return -1;
    __builtin_amdgcn_s_sleep(2);
    v = status_load(excl + tile);
  }
  return (long long)(v & kValMask);
}



// ---- per-byte bitmaps shared between piece lanes and row lanes -------------------------------
// While a tile's 16-byte pieces are still in the lanes' prefetch registers, each lane can
// classify its own bytes (SWAR, 4 words) and leave one bit per byte in an LDS bitmap; a row
// lane then picks up the up to 96 bits of its row with four dword reads and a funnel shift,
// instead of re-reading and re-classifying 24 words of LDS itself.
// gathers bit 7 of the four byte lanes of `c` (which must have no other bits set) into bits 0..3
__device__ __forceinline__ uint32_t gather_bit7(uint32_t c) { return (((c >> 7) * 0x01020408u) >> 24) & 15u; }
// the same for the four words of a 16-byte piece at once: bit 4j + i of the result = bit 7 of byte i of word j.
// One byte-wise dot product per word (0x80 x weight, summed: a full-rate instruction, where the multiply above
// is quarter rate), two words per accumulator.
__device__ __forceinline__ uint32_t gather16_bit7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  const uint32_t lo = __builtin_amdgcn_udot4(c1, 0x80402010u, __builtin_amdgcn_udot4(c0, 0x08040201u, 0u, false), false);  // 128 x bits 0..7
  const uint32_t hi = __builtin_amdgcn_udot4(c3, 0x80402010u, __builtin_amdgcn_udot4(c2, 0x08040201u, 0u, false), false);  // 128 x bits 8..15
  return (lo >> 7) | (hi << 1);
}
__device__ __forceinline__ void put_bits16(uint32_t* bitmap, int byte_index, uint32_t bits16) {
  reinterpret_cast<uint16_t*>(bitmap)[byte_index >> 4] = (uint16_t)bits16;
}
// bits [p0, p0 + n) of the bitmap, n <= 96, as three words (bit i of m0 = bit p0 + i)
__device__ __forceinline__ void row_bits96(const uint32_t* bitmap, int p0, int n, uint32_t& m0, uint32_t& m1, uint32_t& m2) {
  const uint32_t* w = bitmap + (p0 >> 5);
  const unsigned sh = (unsigned)(p0 & 31);
  const uint32_t a = w[0], b = w[1], c = w[2], d = w[3];
  uint32_t q0 = sh ? (a >> sh) | (b << (32 - sh)) : a;
  uint32_t q1 = sh ? (b >> sh) | (c << (32 - sh)) : b;
  uint32_t q2 = sh ? (c >> sh) | (d << (32 - sh)) : c;
  q0 &= n >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (n & 31));
  q1 &= n >= 64 ? 0xFFFFFFFFu : (n <= 32 ? 0u : ~(0xFFFFFFFFu << (n & 31)));
  q2 &= n >= 96 ? 0xFFFFFFFFu : (n <= 64 ? 0u : ~(0xFFFFFFFFu << (n & 31)));
  m0 = q0;
  m1 = q1;
  m2 = q2;
}

// Wave-cooperative flush of `total` bytes assembled at lds[0 ..) (lds 4-byte aligned) to
// the arbitrarily aligned global address `dst`: whole 16-byte destination chunks are
// composed from five aligned LDS dwords and a wave-uniform funnel shift; the up to 15
// head and tail bytes go out one byte per lane.
__device__ __forceinline__ void wave_flush_shift(uint8_t* dst, int total, const uint8_t* lds, int lane) {
  const int olead = (int)((uintptr_t)dst & 15);
  gptr<uint8_t> a0 = as_global(dst - olead);         // 16-byte aligned
  gptr<uint8_t> gdst = as_global(dst);
  const int end = olead + total;                     // positions relative to a0
  const int first_full = (olead + 15) & ~15;         // 0 or 16
  const int last_full = end & ~15;
  if (first_full >= last_full) {
    for (int j = lane; j < total; j += 64) gdst[j] = lds[j];
    return;
  }
  const int head = first_full - olead;               // bytes before the first whole chunk
  if (lane < head) gdst[lane] = lds[lane];
  // (gfx950 reads LDS at any alignment: one 16-byte read per chunk, where five aligned dwords and four funnel
  // shifts used to compose it)
  for (int i = first_full + lane * 16; i < last_full; i += 64 * 16) {
    const lds_u32x4u v = *reinterpret_cast<const lds_u32x4u*>(lds + head + (i - first_full));
    u32x4 o = {v.x, v.y, v.z, v.w};
    gstore16((gptr<u32x4>)(a0 + i), o);
  }
  const int tail0 = last_full - olead;               // output index of the first tail byte
  if (tail0 + lane < total) gdst[tail0 + lane] = lds[tail0 + lane];
}

}  // namespace cstile
