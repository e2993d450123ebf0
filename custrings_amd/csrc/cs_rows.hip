// Tile form of the write pass of strip (strip.cu:30-199): the output row is a sub-range of the
// input row, so with the output offsets known (size pass + scan) a wave stages a tile of R
// consecutive rows in LDS, every row lane finds its kept range again on the LDS copy (it only
// touches the row's ends), copies it to its place in the output tile (4 dwords per trip) and
// the tile leaves with 16-byte stores.  Replaces the per-byte global-to-global copy of the
// row-wise write kernel.
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "device_utils.h"
#include "row_ops.h"
#include "tile_utils.h"

using namespace cs;
using namespace csdev;
using namespace csrow;

namespace cs {
bool find_tiles(const cs_column* in, const unsigned char* needle, int nb, int mode, int start, int end, int32_t* out32,
                uint8_t* out8, unsigned long long* found, hipStream_t s);
bool strip_write_tiles(const cs_column* in, const CharSet& set, int side, const int64_t* out_off, uint8_t* out_chars,
                       hipStream_t s);
bool strip_single(const cs_column* in, const CharSet& set, int side, hipStream_t s, cs_column* o);
}

namespace {

struct StripTileArgs {
  ColView in;
  CharSet set;
  int side, rows_per_tile, cap;
  long long ntiles;
  const int64_t* out_off;
  uint8_t* out_chars;
};

__global__ void __launch_bounds__(256) k_strip_tile(StripTileArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * 2 * a.cap;
  uint8_t* lds_out = lds_in + a.cap;
  const ColView& in = a.in;
  const int R = a.rows_per_tile;
  const long long waves = (long long)gridDim.x * 4;
  const long long per = (a.ntiles + waves - 1) / waves;
  long long tile = ((long long)blockIdx.x * 4 + wv) * per;
  const long long tile_end = min(a.ntiles, tile + per);
  if (tile >= tile_end) return;
  auto load_offs = [&](long long t) {
    const long long r0 = t * R;
    const int nrows = (int)min((long long)R, in.rows - r0);
    cstile::TileOffs o;
    o.o0 = in.offsets[r0 + min(lane, nrows)];
    o.o1 = in.offsets[r0 + min(lane + 1, nrows)];
    return o;
  };
  cstile::TileOffs cur = load_offs(tile);
  cstile::TileOffs nxt = cur;
  if (tile + 1 < tile_end) nxt = load_offs(tile + 1);
  cstile::TileChars pf;
#pragma unroll
  for (int j = 0; j < cstile::kPfChunks; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
  for (;;) {
    const long long r0 = tile * R;
    const int nrows = (int)min((long long)R, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool live = lane < nrows && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n = live ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const long long want64 = g1 - g0 + lead;
    const bool oversize = want64 + 48 > a.cap;  // (the host sized the buffers for all but a few tiles: a long row among short ones)
    const int want = oversize ? 0 : (int)want64;
    cstile::stage_chars(lds_in, want, lane, pf);
    // output extents of the tile's rows (the size pass and the scan already ran)
    const long long oo0 = a.out_off[r0 + min(lane, nrows)];
    const long long oo1 = a.out_off[r0 + min(lane + 1, nrows)];
    const bool has_next = tile + 1 < tile_end;
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
      if (tile + 2 < tile_end) nxt = load_offs(tile + 2);
    }
    cstile::wave_lds_fence();
    if (oversize) {
      // straight from memory: a short row by its lane, a long one by the whole wave (a byte a lane)
      int lo = 0, hi = 0;
      const uint8_t* p = in.chars + (g0 + rbeg);
      if (live && n > 0) row_strip(p, n, a.set, a.side, lo, hi);
      const int len = hi - lo;
      const unsigned long long big = __ballot(len > 256);
      if (len > 0 && len <= 256)
        for (int i = 0; i < len; ++i) a.out_chars[oo0 + i] = p[lo + i];
      for (unsigned long long m = big; m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        const long long src = cstile::rl64(g0 + rbeg + lo, l), dst = cstile::rl64(oo0, l);
        const int L = __builtin_amdgcn_readlane(len, l);
        for (int i = lane; i < L; i += 64) a.out_chars[dst + i] = in.chars[src + i];
      }
      if (!has_next) break;
      ++tile;
      continue;
    }
    const long long ob = cstile::rl64(oo0, 0), oe = cstile::rl64(oo1, 63);
    if (live && n > 0) {
      int lo, hi;
      row_strip(lds_in + lead + rbeg, n, a.set, a.side, lo, hi);
      cstile::lds_copy(lds_out, (int)(oo0 - ob), lds_in, lead + rbeg + lo, hi - lo);
    }
    cstile::wave_lds_fence();
    cstile::wave_flush_shift(a.out_chars + ob, (int)(oe - ob), lds_out, lane);
    cstile::wave_lds_fence();
    if (!has_next) break;
    ++tile;
  }
}


// find (MODE 0: char position or -1, -2 for null rows) / contains (MODE 1) over row tiles
// (find.cu:75-120,237-272): same staging.  Whole-row searches are byte-parallel: while the
// tile's 16-byte pieces are still in the prefetch registers their lanes compare every byte with
// the needle's first byte (SWAR) and leave one candidate bit per byte in LDS -- for find also
// one "continuation byte" bit per byte -- so a row lane only takes its row's bits (rows up to
// 96 bytes), verifies the few candidates against the rest of the needle, and turns the byte
// position into a character position with a population count.  Tiles with longer rows and
// searches over a character window walk the row in LDS instead.
struct FindTileArgs {
  ColView in;
  uint8_t needle[64];
  int nb, start, end, rows_per_tile, cap;
  int whole;  // the search covers the whole row (find's default window; contains always)
  long long ntiles;
  int32_t* out32;
  uint8_t* out8;
  unsigned long long* found;
};
template <int MODE>
__global__ void __launch_bounds__(256) k_find_tile(FindTileArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  __shared__ uint8_t s_needle[64];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  if (threadIdx.x < 64) s_needle[threadIdx.x] = a.needle[threadIdx.x];
  __syncthreads();
  constexpr int kBitmapBytes = cstile::kPfBytes / 8 + 32;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (a.cap + 2 * kBitmapBytes);
  uint32_t* bm_first = reinterpret_cast<uint32_t*>(lds_in + a.cap);                 // byte == needle[0]
  uint32_t* bm_cont = reinterpret_cast<uint32_t*>(lds_in + a.cap + kBitmapBytes);   // byte is 10xxxxxx
  const ColView& in = a.in;
  const int R = a.rows_per_tile;
  const long long waves = (long long)gridDim.x * 4;
  const long long per = (a.ntiles + waves - 1) / waves;
  long long tile = ((long long)blockIdx.x * 4 + wv) * per;
  const long long tile_end = min(a.ntiles, tile + per);
  if (tile >= tile_end) return;
  auto load_offs = [&](long long t) {
    const long long r0 = t * R;
    const int nrows = (int)min((long long)R, in.rows - r0);
    cstile::TileOffs o;
    o.o0 = in.offsets[r0 + min(lane, nrows)];
    o.o1 = in.offsets[r0 + min(lane + 1, nrows)];
    return o;
  };
  cstile::TileOffs cur = load_offs(tile);
  cstile::TileOffs nxt = cur;
  if (tile + 1 < tile_end) nxt = load_offs(tile + 1);
  cstile::TileChars pf;
#pragma unroll
  for (int j = 0; j < cstile::kPfChunks; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
  const uint32_t first4 = (uint32_t)a.needle[0] * 0x01010101u;
  // bits 0..3: which bytes of w are zero (exact, no borrow between bytes)
  auto zero4 = [](uint32_t w) { return cstile::gather_bit7(~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u); };
  auto cont4 = [](uint32_t w) { return cstile::gather_bit7(w & ~(w << 1) & 0x80808080u); };
  int hits = 0;
  for (;;) {
    const long long r0 = tile * R;
    const int nrows = (int)min((long long)R, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool in_tile = lane < nrows;
    const bool live = in_tile && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n = live ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const int want = (int)(g1 - g0) + lead;
    cstile::stage_chars(lds_in, want, lane, pf);
    // (a search over a character window is a search over a byte window when the tile is ASCII)
    bool by_bits = a.nb > 0 && !__any(n > 96);
    if (by_bits && !a.whole) {
      uint32_t high = 0;
#pragma unroll
      for (int j = 0; j < cstile::kPfChunks; ++j)
        if (j * 1024 + lane * 16 < want) high |= pf.v[j].x | pf.v[j].y | pf.v[j].z | pf.v[j].w;
      by_bits = !__any((high & 0x80808080u) != 0);
    }
    if (by_bits) {
#pragma unroll
      for (int j = 0; j < cstile::kPfChunks; ++j)
        if (j * 1024 < want) {  // wave-uniform
          const int i = j * 1024 + lane * 16;
          const uint4 q = pf.v[j];
          if (i < want)
            cstile::put_bits16(bm_first, i, zero4(q.x ^ first4) | (zero4(q.y ^ first4) << 4) | (zero4(q.z ^ first4) << 8) |
                                                (zero4(q.w ^ first4) << 12));
          if (MODE == 0) {
            const bool high = i < want && ((q.x | q.y | q.z | q.w) & 0x80808080u) != 0;
            uint32_t cb = 0;
            if (__any(high)) cb = cont4(q.x) | (cont4(q.y) << 4) | (cont4(q.z) << 8) | (cont4(q.w) << 12);
            if (i < want) cstile::put_bits16(bm_cont, i, cb);
          }
        }
    }
    const bool has_next = tile + 1 < tile_end;
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
      if (tile + 2 < tile_end) nxt = load_offs(tile + 2);
    }
    cstile::wave_lds_fence();
    const uint8_t* p = lds_in + lead + rbeg;
    int at = -1;  // byte offset of the first occurrence (by_bits)
    if (by_bits && live && n >= a.nb) {
      uint32_t m0, m1, m2;  // candidate starts that leave room for the needle
      int from = 0, to = n;  // the byte window (find.cu:75-120: start, then a count of end - start characters)
      if (!a.whole) {
        const int st = a.start < 0 ? 0 : a.start;
        from = st < n ? st : n;
        if (a.end - st >= 0) to = min(n, from + (a.end - st));
      }
      cstile::row_bits96(bm_first, lead + rbeg, max(0, to - a.nb + 1), m0, m1, m2);
      if (from > 0) {  // drop the candidates before the window
        if (from >= 64) m0 = 0, m1 = 0, m2 &= from >= 96 ? 0u : 0xFFFFFFFFu << (from - 64);
        else if (from >= 32) m0 = 0, m1 &= 0xFFFFFFFFu << (from - 32);
        else m0 &= 0xFFFFFFFFu << from;
      }
      while ((m0 | m1 | m2) != 0) {
        const int pos = m0 ? __builtin_ctz(m0) : (m1 ? 32 + __builtin_ctz(m1) : 64 + __builtin_ctz(m2));
        if (m0) m0 &= m0 - 1;
        else if (m1) m1 &= m1 - 1;
        else m2 &= m2 - 1;
        int j = 1;
        while (j < a.nb && p[pos + j] == s_needle[j]) ++j;
        if (j >= a.nb) {
          at = pos;
          break;
        }
      }
    }
    if (MODE == 0) {
      int v = -2;  // null row (find.cu:108)
      if (live) {
        if (by_bits) {
          v = at;
          if (at > 0) {  // character position = bytes before the hit that are not continuation bytes
            uint32_t c0, c1, c2;
            cstile::row_bits96(bm_cont, lead + rbeg, at, c0, c1, c2);
            v = at - (__builtin_popcount(c0) + __builtin_popcount(c1) + __builtin_popcount(c2));
          }
        } else {
          v = row_find(p, n, s_needle, a.nb, a.start, a.end);
        }
      }
      if (in_tile) {
        a.out32[r0 + lane] = v;
        hits += v != -1;  // null rows are counted too (find.cu:112)
      }
    } else {
      int hit = 0;
      if (by_bits) hit = at >= 0;
      else if (live && a.nb > 0) hit = find_bytes(p, 0, n, s_needle, a.nb) >= 0;
      if (in_tile) a.out8[r0 + lane] = (uint8_t)hit;
      hits += hit;
    }
    cstile::wave_lds_fence();
    if (!has_next) break;
    ++tile;
  }
  const int t = wave_reduce_sum(hits);
  if (lane == 0 && t) atomicAdd(a.found, (unsigned long long)t);
}

#if defined(CS_EXPERIMENTS)  // (`make exp`: built, bit-exact, measured slower than the two passes -- not in the product library)
// strip in ONE pass (strip.cu:97-141 sizes the rows, scans, then writes): every wave takes 64-row tiles by ticket, stages a
// tile's chars (the next tile's already in flight), finds each row's stripped range in LDS, publishes the tile's byte total
// (tile_utils.h: the decoupled look-back; a ticket's predecessors have all been started), assembles the rows in the out
// tile while the predecessors' totals arrive, then writes the tile's offsets and flushes its bytes at the prefix.  The
// output's chars are provisioned with the input's size (strip only takes bytes away).  The host falls back to the two
// passes when the launch gives up (a wait without progress: tile_utils.h) or the column does not fit 64-row tiles.
struct StripStreamArgs {
  ColView in;
  CharSet set;
  int side, cap;
  long long ntiles;
  cstile::u64* status;         // one word per tile (+ slack: the scanners read whole steps), zeroed
  cstile::u64* excl;           // the tiles' exclusive prefixes, written by the scanner team (+ slack), zeroed
  unsigned long long* ticket;  // zeroed
  unsigned* error;             // zeroed; bit 0: the launch is lost
  int64_t* out_off;
  uint8_t* out_chars;
  int debug;
};
__global__ void __launch_bounds__(256) k_strip_stream(StripStreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * 2 * a.cap;
  uint8_t* lds_out = lds_in + a.cap;
  const ColView& in = a.in;
  // Workgroup 0 takes no tiles: its four waves turn the totals the others publish into exclusive prefixes, 256 tiles a
  // step (tile_utils.h: prefix_scanner_team).  A chain of look-backs passes 64 tiles per trip through the L2 -- 3 ms for
  // this kernel's 156 k tiles of the 10M-row column, five times the work itself.  (A grid of one workgroup: the look-back.)
  const bool team = gridDim.x > 1;
  if (team && blockIdx.x == 0) {
    cstile::TeamRing* ring = reinterpret_cast<cstile::TeamRing*>(smem);
    if (threadIdx.x < 8) cstile::team_ring_init(ring, threadIdx.x);
    __syncthreads();
    if (cstile::prefix_scanner_team(a.status, a.excl, a.ntiles, lane, wv, 4, ring, a.error) == 1 && lane == 0) atomicOr(a.error, 1u);
    return;
  }
  // Tickets: one counter per class of workgroups (blockIdx mod K), counter k hands out tiles k, k + K, ... -- a single word
  // hands out some 65 tickets a microsecond, this kernel takes 500 (every class has a workgroup that takes tiles: workgroup 0
  // may be the scanners')
  const long long K = gridDim.x >= 34 ? 16 : (gridDim.x >= 18 ? 8 : 1);
  const long long key = (long long)blockIdx.x % K;
  unsigned long long* my_ticket = a.ticket + key * 8;
  auto take = [&]() -> unsigned long long {
    unsigned long long t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(my_ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  auto tile_of = [&](unsigned long long t) -> long long { return cstile::rl64((long long)t, 0) * K + key; };
  auto load_offs = [&](long long t) {
    const long long r0 = t * 64;
    const int nrows = (int)min(64ll, in.rows - r0);
    cstile::TileOffs o;
    o.o0 = in.offsets[r0 + min(lane, nrows)];
    o.o1 = in.offsets[r0 + min(lane + 1, nrows)];
    return o;
  };
  // (the first tickets a round trip apart: drawn back to back they are consecutive, and a wave's second and third tile
  // would lie in front of its neighbour's first -- cs_regex.hip, the replace stream kernel)
  long long tile = tile_of(take());
  if (tile >= a.ntiles) return;
  cstile::TileOffs cur = load_offs(tile);
  cstile::TileChars pf;
#pragma unroll
  for (int j = 0; j < cstile::kPfChunks; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
  long long t_nxt = tile_of(take());
  cstile::TileOffs nxt = load_offs(t_nxt < a.ntiles ? t_nxt : a.ntiles - 1);
  unsigned long long pending = take();
  // the tile before: assembled in the out tile, finished (offsets, flush) one iteration later -- when its prefix has had
  // this tile's staging and sizing to arrive in
  long long p_tile = -1, p_total = 0, p_pre = 0;
  bool lost = false;
  auto prefix_of = [&](long long t, long long total) -> long long {
    if (a.debug & 1) return t * 2048;
    const long long gb = team ? cstile::prefix_wait(a.excl, t, cstile::status_load(a.excl + t), a.error, lane)
                              : cstile::lookback_end(a.status, t, total, cstile::lookback_poll(a.status, t, lane), lane);
    if (gb < 0) {
      // the launch is lost; publish something, so that nobody waits for this tile as well
      if (lane == 0) {
        atomicOr(a.error, 1u);
        if (!team) cstile::status_store(a.status + t, cstile::kFlagInc);
      }
      lost = true;
    }
    return gb;
  };
  auto write_offsets = [&](long long t, long long gb, long long pre, long long total) {
    const long long r0 = t * 64;
    const int nrows = (int)min(64ll, in.rows - r0);
    if (lane < nrows) a.out_off[r0 + lane] = gb + pre;
    if (lane == nrows - 1 && r0 + nrows == in.rows) a.out_off[in.rows] = gb + total;
  };
  auto finish_pending = [&]() {
    const long long gb = prefix_of(p_tile, p_total);
    if (lost) return;
    write_offsets(p_tile, gb, p_pre, p_total);
    cstile::wave_flush_shift(a.out_chars + gb, (int)p_total, lds_out, lane);
    cstile::wave_lds_fence();
    p_tile = -1;
  };
  for (;;) {
    const long long r0 = tile * 64;
    const int nrows = (int)min(64ll, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool live = lane < nrows && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n = live ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const long long want64 = g1 - g0 + lead;
    const bool oversize = want64 + 48 > a.cap;  // (a long row among short ones: its tile goes straight from memory)
    const int want = oversize ? 0 : (int)want64;
    cstile::stage_chars(lds_in, want, lane, pf);
    // the ticket after next, the offsets of the tile after next, the next tile's chars: all in flight over this tile's work
    // (assigned unconditionally, handed to the loop-carried variables at the bottom: cs_regex.hip on why)
    const long long t_nn = tile_of(pending);
    const unsigned long long pending_new = take();
    const cstile::TileOffs nn = load_offs(t_nn < a.ntiles ? t_nn : a.ntiles - 1);
    const bool has_next = t_nxt < a.ntiles;
    const uint8_t* gp = in.chars + (g0 + rbeg);
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
    }
    cstile::wave_lds_fence();
    int lo = 0, hi = 0;
    if (live && n > 0) {
      if (oversize) row_strip(gp, n, a.set, a.side, lo, hi);
      else row_strip(lds_in + lead + rbeg, n, a.set, a.side, lo, hi);
    }
    const int len = hi - lo;
    // (a tile's bytes fit 32 bits only when no row is huge: the sum in 64 bits)
    long long incl = len;
    for (int d = 1; d < 64; d <<= 1) {
      const long long up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    const long long total = cstile::rl64(incl, 63);
    const long long pre = incl - len;
    cstile::lookback_publish(a.status, tile, total, lane);
    if (p_tile >= 0) {
      finish_pending();
      if (lost) return;
    }
    if (oversize) {
      // finished at once, straight from memory: a short row by its lane, a long one by the whole wave (a byte a lane)
      const long long gb = prefix_of(tile, total);
      if (lost) return;
      write_offsets(tile, gb, pre, total);
      const unsigned long long big = __ballot(len > 256);
      if (len > 0 && len <= 256)
        for (int i = 0; i < len; ++i) a.out_chars[gb + pre + i] = gp[lo + i];
      for (unsigned long long m = big; m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        const long long src = cstile::rl64(g0 + rbeg + lo, l), dst = cstile::rl64(gb + pre, l);
        const int L = __builtin_amdgcn_readlane(len, l);
        for (int i = lane; i < L; i += 64) a.out_chars[dst + i] = in.chars[src + i];
      }
    } else {
      if (len > 0) cstile::lds_copy(lds_out, (int)pre, lds_in, lead + rbeg + lo, len);
      cstile::wave_lds_fence();
      p_tile = tile;
      p_total = total;
      p_pre = pre;
    }
    if (!has_next) break;
    tile = t_nxt;
    t_nxt = t_nn;
    nxt = nn;
    pending = pending_new;
  }
  if (p_tile >= 0) finish_pending();
}
#endif

}  // namespace

namespace cs {


bool find_tiles(const cs_column* in, const unsigned char* needle, int nb, int mode, int start, int end, int32_t* out32,
                uint8_t* out8, unsigned long long* found, hipStream_t s) {
  if (in->rows == 0 || nb > 64 || cs::cfg("CS_FIND_ROWWISE")) return false;
  int R = 0;
  for (int r : {64, 32, 16}) {
    if (max_span_rows(in, r, s) + 32 <= cstile::kPfBytes) {
      R = r;
      break;
    }
  }
  if (!R) return false;
  FindTileArgs a{};
  a.in = view_of(in);
  for (int i = 0; i < nb; ++i) a.needle[i] = needle[i];
  a.nb = nb;
  a.start = start;
  a.end = end;
  a.rows_per_tile = R;
  a.cap = (int)((max_span_rows(in, R, s) + 48 + 15) & ~(int64_t)15);
  a.ntiles = (in->rows + R - 1) / R;
  a.out32 = out32;
  a.out8 = out8;
  a.found = found;
  a.whole = mode != 0 || (start <= 0 && end - (start < 0 ? 0 : start) < 0);
  const size_t lds = ((size_t)a.cap + 2 * (cstile::kPfBytes / 8 + 32)) * 4;
  if (lds > 150 * 1024) return false;
  auto launch = [&](auto kern) {
    if (lds > 48 * 1024)
      CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned g = resident_grid(reinterpret_cast<const void*>(kern), lds, (a.ntiles + 3) / 4);
    hipLaunchKernelGGL(kern, dim3(g), dim3(256), lds, s, a);
  };
  if (mode == 0) launch(&k_find_tile<0>);
  else launch(&k_find_tile<1>);
  CS_HIP(hipGetLastError());
  return true;
}

bool strip_write_tiles(const cs_column* in, const CharSet& set, int side, const int64_t* out_off, uint8_t* out_chars,
                       hipStream_t s) {
  if (in->rows == 0 || cs::cfg("CS_STRIP_ROWWISE")) return false;
  int R = 0;
  for (int r : {64, 32, 16}) {
    if (max_span_rows(in, r, s) + 32 <= cstile::kPfBytes) {
      R = r;
      break;
    }
  }
  int64_t span = R ? max_span_rows(in, R, s) : 0;
  if (!R && !cs::cfg("CS_NO_OUTLIER_TILES")) {
    R = 64;  // no tile size fits every tile: the kernel copies the rows of a tile beyond the staging size straight from memory, long rows by the whole wave
    span = cstile::kPfBytes - 64;
  }
  if (!R) return false;
  StripTileArgs a{};
  a.in = view_of(in);
  a.set = set;
  a.side = side;
  a.rows_per_tile = R;
  a.cap = (int)((span + 48 + 15) & ~(int64_t)15);
  a.ntiles = (in->rows + R - 1) / R;
  a.out_off = out_off;
  a.out_chars = out_chars;
  const size_t lds = (size_t)a.cap * 2 * 4;
  if (lds > 150 * 1024) return false;
  if (lds > 48 * 1024)
    CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_strip_tile), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  const unsigned g = resident_grid(reinterpret_cast<const void*>(&k_strip_tile), lds, (a.ntiles + 3) / 4);
  ProfScope ps("k_strip_write", s);
  hipLaunchKernelGGL(k_strip_tile, dim3(g), dim3(256), lds, s, a);
  CS_HIP(hipGetLastError());
  return true;
}

// strip in one pass (k_strip_stream): false when the column does not take the route or the launch gave up (the caller runs
// the two passes).  On success `o` has offsets / chars / nbytes.
// OPT-IN (CS_STRIP_SINGLE=1): measured SLOWER than the two passes on the 10M-row C2 column -- 0.98 ms against 0.63.  The
// worker alone (prefix made up: CS_STRIP_DEBUG=1) takes 0.42; the prefix protocol costs the rest although every tile's
// finish is deferred by an iteration: a tile here is 8 us of work, about what a total needs to come back as a prefix
// (publish -> scanners' poll -> ring -> store -> the worker's load, behind the slowest of the predecessors in flight).
// On the way: a chain of look-backs passes 64 tiles per L2 round trip (2.97 ms), ONE ticket counter hands out 65 tickets a
// microsecond (2.48 ms with the scanner team), sixteen counters 500.
bool strip_single(const cs_column* in, const CharSet& set, int side, hipStream_t s, cs_column* o) {
#if !defined(CS_EXPERIMENTS)
  return false;  // (the one-pass kernel is in the experiments build only: `make exp`)
#else
  if (in->rows == 0 || !cs::cfg("CS_STRIP_SINGLE") || cs::cfg("CS_STRIP_ROWWISE")) return false;
  int64_t span = max_span_rows(in, 64, s);
  if (span + 32 > cstile::kPfBytes) {
    if (cs::cfg("CS_NO_OUTLIER_TILES")) return false;
    span = cstile::kPfBytes - 64;  // (tiles beyond the staging size go straight from memory)
  }
  StripStreamArgs a{};
  a.in = view_of(in);
  a.set = set;
  a.side = side;
  a.cap = (int)((span + 48 + 15) & ~(int64_t)15);
  a.ntiles = (in->rows + 63) / 64;
  const size_t lds = (size_t)a.cap * 2 * 4;
  if (lds > 150 * 1024) return false;
  // [status ntiles + 512][exclusive prefixes ntiles + 512][tickets 16 x 8 words][error 8 words]  (slack: the scanners read whole steps)
  const size_t nst = (size_t)a.ntiles + 512;
  Buf ctl = dev_alloc(sizeof(uint64_t) * (2 * nst + 136), s);
  CS_HIP(hipMemsetAsync(ctl->p, 0, sizeof(uint64_t) * (2 * nst + 136), s));
  a.status = ptr<cstile::u64>(ctl);
  a.excl = ptr<cstile::u64>(ctl) + nst;
  a.ticket = ptr<unsigned long long>(ctl) + 2 * nst;
  a.error = reinterpret_cast<unsigned*>(ptr<unsigned long long>(ctl) + 2 * nst + 128);
  Buf off = dev_alloc(sizeof(int64_t) * (in->rows + 1), s);
  Buf chars = dev_alloc((size_t)in->nbytes + 64, s);
  a.out_off = ptr<int64_t>(off);
  a.out_chars = ptr<uint8_t>(chars);
  a.debug = cs::cfg_int("CS_STRIP_DEBUG", 0);
  if (lds > 48 * 1024)
    CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_strip_stream), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // (+ 1: workgroup 0 is the scanners')
  const unsigned g = resident_grid(reinterpret_cast<const void*>(&k_strip_stream), lds, (a.ntiles + 3) / 4 + 1);
  {
    ProfScope ps("k_strip", s);
    hipLaunchKernelGGL(k_strip_stream, dim3(g), dim3(256), lds, s, a);
  }
  CS_HIP(hipGetLastError());
  int64_t* host = (int64_t*)pinned_scratch(16);
  CS_HIP(hipMemcpyAsync(host, ptr<int64_t>(off) + in->rows, 8, hipMemcpyDeviceToHost, s));
  CS_HIP(hipMemcpyAsync(host + 1, a.error, 4, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if ((uint32_t)host[1] != 0) {
    note_fallback("strip (single pass)");
    return false;
  }
  o->offsets = off;
  o->chars = chars;
  o->nbytes = host[0];
  return true;
#endif
}

}  // namespace cs
