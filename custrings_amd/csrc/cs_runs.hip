// replace_re for patterns that are ONE character class, taken once or in a greedy `+` loop -- `[aeiou]+`, `\s+`-like sets
// of ASCII bytes, `[^ ]+`, `.`, a single literal -- by byte-parallel stream compaction (NVStrings::replace_re,
// replace.cu:39-189; the executor's semantics, regexec.inl:204-442, restated: the matches of such a pattern are the maximal
// runs of member bytes inside a row -- or every member byte, without the `+` --, leftmost first, never overlapping).
//
// Why a kernel of its own.  A row-lane scan costs the longest row of a wave times the matches in it; the bit-parallel form
// (regex_bits.h) holds a row in 96-bit masks.  BASELINE.json's C5 rows are 40-150 bytes with a vowel run every four bytes:
// replace_re([aeiou]+) took 82.8 ms on the 62.5M-row shard on the long-row forms of the automaton kernel.  Here the work is
// per BYTE, as in tokenize (cs_tokenize.hip): a wave takes a tile of R consecutive rows (64 / 32 / 16: the tile's chars span
// fits the prefetch registers), every lane classifies the 16-byte pieces it loaded through the pattern's class table
// (regex_bits.h: classify16), the row lanes leave a row-start bitmap in LDS, a run starts at a member byte whose predecessor
// IN THE SAME ROW is not one, and the output position of every kept byte and every replacement follows from popcounts and
// one wave scan per piece.  Rows of any length, any bytes (a byte >= 0x80 is a member exactly when every non-ASCII
// character is: the class's flag), nulls and empty rows.  A NUL byte ends the reference's scan of its row (regexec.inl: the
// loop runs `while (c && ...)`): nothing at or behind it matches -- a tile that holds one is taken row by row by the row
// lanes, straight from memory (rare; the same answers).  Two passes over the chars:
//   pass 0  every row's output length (the row lanes read the positions of their row's first and one-past-last byte out of
//           the pieces' prefix table), the largest tile output (sizes pass 1's LDS)
//   pass 1  compaction into an LDS tile, the replacement text at every run start, one coalesced flush at offsets[first row]
// with the generic scan over the lengths in between (which also leaves the output's metadata: cs_core.hip, LenMeta).
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "device_utils.h"
#include "regex_bits.h"
#include "tile_utils.h"

using namespace cs;
using namespace csdev;

namespace cs {
bool replace_class_runs(const cs_column* col, const int32_t* d_bits, const std::vector<int32_t>& bits, const char* repl, int rb, hipStream_t s, cs_column** out);
bool count_class_runs(const cs_column* col, const int32_t* d_bits, const std::vector<int32_t>& bits, hipStream_t s, int32_t* results, int64_t* hits);
}

namespace {

constexpr int kMaxRunRepl = 16;  // replacement bytes kept in four registers

struct RunsArgs {
  ColView in;
  int rows_per_tile;
  long long ntiles;
  const int32_t* bits;  // the pattern's bit image (class table: regex_bits.h)
  int high_member;      // bytes >= 0x80 belong to the class (a negated class, `.`)
  int flag_class;       // 0, or 1 | builtins << 8 | negated << 16: a non-ASCII character belongs through the unicode flags (regex_bits.h: F_FLAG_CLASS)
  const uint8_t* flags; // the unicode flags table (64 K entries)
  int plus;             // runs (class+) or single members
  int nul_blind;        // the pattern is a literal character (`a`, `a+`): a NUL byte ends nothing (regex_bits.h: F_CHAR_FIRST)
  int rb;
  int count_only;       // pass 0 as count_re: a row's matches instead of its output bytes (rb = 1, kept bytes count nothing, a null row 0)
  uint32_t rep[4];
  // pass 0
  int32_t* lens;  // [rows]: output bytes, -1 for a null row
  int* maxima;    // [0] most output bytes of a tile
  // pass 1
  const int64_t* out_off;  // [rows + 1]
  uint8_t* out_chars;
  int cap_out;
};

__device__ __forceinline__ uint32_t high16_of(const uint4& q) {  // bit b = byte b >= 0x80
  auto g = [](uint32_t w) { return (((w & 0x80808080u) >> 7) * 0x01020408u >> 24) & 15u; };
  return g(q.x) | (g(q.y) << 4) | (g(q.z) << 8) | (g(q.w) << 12);
}
__device__ __forceinline__ uint32_t byte_at(const uint4& q, int b) {
  const uint32_t w = b < 8 ? (b < 4 ? q.x : q.y) : (b < 12 ? q.z : q.w);
  return (w >> (8 * (b & 3))) & 255u;
}

template <int PASS>
__global__ void __launch_bounds__(256) k_runs_tile(RunsArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  constexpr int kBitmapBytes = cstile::kPfBytes / 8 + 32;
  constexpr int kPieces = cstile::kPfChunks * 64;  // pieces of 16 bytes a tile holds at most
  // block: the spread class table (128 words); per wave: row-start bitmap, per piece its output prefix and its masks, out tile
  uint32_t* spread = smem;
  const int per_wave = kBitmapBytes + (kPieces + 1) * 8 + (PASS ? a.cap_out : 0);
  uint8_t* base = reinterpret_cast<uint8_t*>(smem + 128) + (size_t)wv * per_wave;
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(base);
  uint32_t* pexcl = reinterpret_cast<uint32_t*>(base + kBitmapBytes);  // output bytes in front of the piece
  uint32_t* pmask = pexcl + kPieces + 1;                                // kept bytes | run starts << 16
  uint8_t* lds_out = reinterpret_cast<uint8_t*>(pmask + kPieces + 1);
  if (threadIdx.x < 128) {
    const uint32_t set = ((uint32_t)a.bits[csbits::kHeaderWords + (threadIdx.x >> 2)] >> (8 * (threadIdx.x & 3))) & 255u;
    spread[threadIdx.x] = csbits::spread_entry(set);
  }
  __syncthreads();
  const ColView& in = a.in;
  const int R = a.rows_per_tile, rb = a.rb;
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
  int most = 0, hit_rows = 0;
  for (;;) {
    const long long r0 = tile * R;
    const int nrows = (int)min((long long)R, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool live = lane < nrows && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n = live ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const int want = (int)(g1 - g0) + lead;  // (the host sized R so that every tile fits the prefetch registers)
    const cstile::TileChars q = pf;
    const bool has_next = tile + 1 < tile_end;
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
      if (tile + 2 < tile_end) nxt = load_offs(tile + 2);
    }
    // row-start bitmap: one bit per byte of the staged span, set by the row lanes
    for (int i = lane * 16; i < kBitmapBytes; i += 64 * 16) *reinterpret_cast<uint4*>(base + i) = make_uint4(0, 0, 0, 0);
    cstile::wave_lds_fence();
    if (n > 0) {
      const int p = lead + rbeg;
      __hip_atomic_fetch_or(bitmap + (p >> 5), 1u << (p & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    cstile::wave_lds_fence();
    int carry = 0;            // output bytes of the pieces before this chunk row (wave-uniform)
    uint32_t carry_mem = 0;   // was the last byte of the previous chunk row a member?
    uint32_t carry_ann = 0;   // continuation bytes the previous chunk row's last leads announce for this one
    uint32_t nul_seen = 0;
    uint32_t keep_j[cstile::kPfChunks], start_j[cstile::kPfChunks];
    int base_j[cstile::kPfChunks];
#pragma unroll
    for (int j = 0; j < cstile::kPfChunks; ++j) {
      keep_j[j] = start_j[j] = 0;
      base_j[j] = 0;
      if (j * 1024 < want) {  // wave-uniform
        const int i = j * 1024 + lane * 16;
        const int lo = min(16, max(0, lead - i)), hi = min(16, max(0, want - i));
        const uint32_t valid = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        uint32_t pair[4];
        csbits::classify16(spread, q.v[j].x, q.v[j].y, q.v[j].z, q.v[j].w, pair);
        const uint32_t high = high16_of(q.v[j]);
        {
          auto z = [](uint32_t w) { return (w - 0x01010101u) & ~w & 0x80808080u; };  // (exact for the lowest zero byte: enough for "any")
          const uint32_t zero = high16_of(make_uint4(z(q.v[j].x), z(q.v[j].y), z(q.v[j].z), z(q.v[j].w)));
          nul_seen |= zero & valid;
          if (a.flag_class) nul_seen |= high & valid;  // (such a tile goes row by row too: its non-ASCII characters are decoded there)
        }
        const uint32_t mem = (a.high_member ? ((pair[0] & 0xFFFFu) | high) : ((pair[0] & 0xFFFFu) & ~high)) & valid;
        const uint32_t rs = (bitmap[i >> 5] >> (i & 31)) & 0xFFFFu;
        uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(mem >> 15), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        if (lane == 0) prev = carry_mem;
        const uint32_t before = ((mem << 1) | prev) & 0xFFFFu;  // bit b = byte b - 1 is a member
        // (without the `+` every member CHARACTER is a match: a multi-byte character -- a member only when bytes >= 0x80
        // are -- counts at its lead byte, its continuation bytes 10xxxxxx just go)
        const uint32_t cont = high & ~high16_of(make_uint4(q.v[j].x << 1, q.v[j].y << 1, q.v[j].z << 1, q.v[j].w << 1));
        {
          // Byte by byte is the executor's answer only where the bytes >= 0x80 form whole characters: it takes a character's width
          // from its lead byte and swallows what follows, whatever that is (regex_vm.h: char_at) -- an ASCII member behind a
          // lead without its continuation bytes is no match, a stray continuation byte is a character of its own.  The positions
          // the leads ANNOUNCE as continuation bytes (one / two / three behind a lead >= 0xC0 / 0xE0 / 0xF0, carried into the
          // next piece) must be exactly the continuation bytes, and none may open a row: anything else sends the tile row by row.
          // (a chunk row without a byte >= 0x80 and nothing announced into it has nothing to check: most of plain text)
          if (__any((high & valid) != 0) || carry_ann != 0) {
          const uint32_t b5 = high16_of(make_uint4(q.v[j].x << 2, q.v[j].y << 2, q.v[j].z << 2, q.v[j].w << 2));
          const uint32_t b4 = high16_of(make_uint4(q.v[j].x << 3, q.v[j].y << 3, q.v[j].z << 3, q.v[j].w << 3));
          const uint32_t l2 = high & ~cont & valid, l3 = l2 & b5, l4 = l3 & b4;
          const uint32_t ann = (l2 << 1) | (l3 << 2) | (l4 << 3);  // bits 1..18
          uint32_t from_prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(ann >> 16), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
          if (lane == 0) from_prev = carry_ann;
          const uint32_t expect = (ann & 0xFFFFu) | from_prev;
          nul_seen |= ((expect ^ cont) | (cont & rs)) & valid;
          carry_ann = (uint32_t)__builtin_amdgcn_readlane((int)(ann >> 16), 63);
          }
        }
        const uint32_t starts = a.plus ? (mem & (rs | ~before) & 0xFFFFu) : (mem & ~cont);
        const uint32_t keep = ~mem & valid;
        const int nout = (a.count_only ? 0 : __builtin_popcount(keep)) + rb * __builtin_popcount(starts);
        const int incl = wave_inclusive_scan(nout);
        const int excl = carry + incl - nout;
        keep_j[j] = keep;
        start_j[j] = starts;
        base_j[j] = excl;
        pexcl[i >> 4] = (uint32_t)excl;
        pmask[i >> 4] = (a.count_only ? 0u : keep) | (starts << 16);
        carry += __builtin_amdgcn_readlane(incl, 63);
        carry_mem = (uint32_t)__builtin_amdgcn_readlane((int)(mem >> 15), 63);
      }
    }
    const int total = carry;
    if (__builtin_expect(__any(nul_seen != 0), 0)) {
      // a NUL byte somewhere in the tile: every row by its lane, from memory -- members up to the row's first NUL only
      int len = live ? 0 : -1;
      if (live) {
        const uint8_t* p = in.chars + (g0 + rbeg);
        uint8_t* o = PASS ? a.out_chars + a.out_off[r0 + lane] : nullptr;
        bool in_run = false, dead = false;
        for (int i = 0; i < n; ++i) {
          const uint32_t b = p[i];
          dead = dead || (b == 0 && !a.nul_blind);
          if (b >= 128u && !dead) {
            // a non-ASCII character, decoded as the executor decodes it (regex_vm.h: char_at -- the width from the lead byte, a stray
            // continuation byte a character of its own, a sequence cut at the row's end), all its bytes together: a member by the
            // class's flag (a byte class: every such character or none) or through the unicode flags (class_match)
            csrow::Char ch;
            unsigned w = csrow::decode_at(p, i, n, ch);
            if (w == 0) w = 1;
            if (i + (int)w > n) w = (unsigned)(n - i);
            bool m = a.high_member != 0;
            if (a.flag_class) {
              const unsigned cp = csrow::packed_to_cp(ch);
              bool cm = false;
              if (cp <= 0xFFFFu) {
                const unsigned f = a.flags[cp], bi = (unsigned)(a.flag_class >> 8) & 63u;
                const bool alnum = (f & 15u) != 0;
                cm = ((bi & 1u) && alnum) || ((bi & 2u) && (f & 16u)) || ((bi & 4u) && (f & 4u)) || ((bi & 8u) && !alnum) || ((bi & 16u) && !(f & 16u)) || ((bi & 32u) && !(f & 4u));
              }
              m = ((a.flag_class >> 16) & 1) ? !cm : cm;
            }
            if (m) {
              if (!a.plus || !in_run) {
                if (PASS)
                  for (int k = 0; k < rb; ++k) o[len + k] = (uint8_t)(a.rep[k >> 2] >> (8 * (k & 3)));
                len += rb;
              }
            } else {
              if (PASS)
                for (unsigned k = 0; k < w; ++k) o[len + (int)k] = p[i + (int)k];
              if (!a.count_only) len += (int)w;
            }
            in_run = m;
            i += (int)w - 1;
            continue;
          }
          const bool m = !dead && (b < 128u ? (spread[b] & 1u) != 0 : a.high_member != 0);
          if (m) {
            if (a.plus ? !in_run : (b & 0xC0u) != 0x80u) {
              if (PASS)
                for (int k = 0; k < rb; ++k) o[len + k] = (uint8_t)(a.rep[k >> 2] >> (8 * (k & 3)));
              len += rb;
            }
          } else {
            if (PASS) o[len] = (uint8_t)b;
            if (!a.count_only) ++len;
          }
          in_run = m;
        }
      }
      if (PASS == 0 && a.count_only) {
        if (len < 0) len = 0;
        const unsigned long long hits = __ballot(len > 0);
        hit_rows += __builtin_popcountll(hits);  // (added to the column's count once, when the wave is through: a same-address atomic a tile cost more than the tile)
      }
      if (PASS == 0 && lane < nrows) a.lens[r0 + lane] = len;
      cstile::wave_lds_fence();
      if (!has_next) break;
      ++tile;
      continue;
    }
    if (lane == 0) {  // the position one past the staged span (a row that ends there reads it)
      pexcl[(want + 15) >> 4] = (uint32_t)total;
      pmask[(want + 15) >> 4] = 0;
    }
    cstile::wave_lds_fence();
    // output position of the staged byte x (0 <= x <= want): what stands in front of it
    auto pos_of = [&](int x) -> int {
      const uint32_t m = pmask[x >> 4];
      const uint32_t below = (1u << (x & 15)) - 1u;
      return (int)pexcl[x >> 4] + __builtin_popcount(m & below) + rb * __builtin_popcount((m >> 16) & below);
    };
    if (PASS == 0) {
      int len = a.count_only ? 0 : -1;
      if (live) len = n > 0 ? pos_of(lead + rbeg + n) - pos_of(lead + rbeg) : 0;
      if (a.count_only) {
        const unsigned long long hits = __ballot(len > 0);
        hit_rows += __builtin_popcountll(hits);  // (added to the column's count once, when the wave is through: a same-address atomic a tile cost more than the tile)
      }
      if (lane < nrows) a.lens[r0 + lane] = len;
      most = max(most, total);
    } else {
#pragma unroll
      for (int j = 0; j < cstile::kPfChunks; ++j) {
        if (j * 1024 < want) {
          const uint32_t keep = keep_j[j], starts = start_j[j];
          const int kb = base_j[j];
#pragma unroll
          for (int b = 0; b < 16; ++b) {
            const uint32_t below = (1u << b) - 1u;
            const int at = kb + __builtin_popcount(keep & below) + rb * __builtin_popcount(starts & below);
            if ((keep >> b) & 1u) lds_out[at] = (uint8_t)byte_at(q.v[j], b);
            if ((starts >> b) & 1u) {
              for (int k = 0; k < rb; ++k) lds_out[at + k] = (uint8_t)(a.rep[k >> 2] >> (8 * (k & 3)));
            }
          }
        }
      }
      cstile::wave_lds_fence();
      cstile::wave_flush_shift(a.out_chars + a.out_off[r0], total, lds_out, lane);
    }
    cstile::wave_lds_fence();  // the next tile reuses the tables
    if (!has_next) break;
    ++tile;
  }
  if (PASS == 0 && lane == 0 && most > __hip_atomic_load(a.maxima, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.maxima, most);
  if (PASS == 0 && lane == 0 && hit_rows) atomicAdd(a.maxima + 1, hit_rows);
}

}  // namespace

namespace cs {

// what both entry points share: the tile size the column's spans allow, the class description out of the bit program
static bool runs_setup(const cs_column* col, const int32_t* d_bits, const std::vector<int32_t>& bits, hipStream_t s, RunsArgs& a) {
  const int64_t rows = col->rows;
  int R = 0;
  for (int r : {64, 32, 16}) {
    if (max_span_rows(col, r, s) + 16 <= cstile::kPfBytes) {
      R = r;
      break;
    }
  }
  if (!R) return false;
  a.in = view_of(col);
  a.rows_per_tile = R;
  a.ntiles = (rows + R - 1) / R;
  a.bits = d_bits;
  a.high_member = (bits[2] & csbits::F_HIGH_MEMBER) ? 1 : 0;
  a.flag_class = (bits[2] & csbits::F_FLAG_CLASS) ? (1 | (((bits[2] >> 16) & 63) << 8) | (((bits[2] >> 22) & 1) << 16)) : 0;
  a.flags = d_unicode_flags();
  a.plus = (bits[2] & csbits::F_PLUS) ? 1 : 0;
  a.nul_blind = (bits[2] & csbits::F_CHAR_FIRST) ? 1 : 0;
  return true;
}
constexpr size_t kRunsBitmapBytes = cstile::kPfBytes / 8 + 32, kRunsPieces = cstile::kPfChunks * 64;
constexpr size_t kRunsWave0 = kRunsBitmapBytes + (kRunsPieces + 1) * 8;  // LDS of a wave in the size pass

// count_re of such a pattern: the size pass counting matches (results: device, int32 per row; *hits = rows with a match)
bool count_class_runs(const cs_column* col, const int32_t* d_bits, const std::vector<int32_t>& bits, hipStream_t s, int32_t* results, int64_t* hits) {
  const int64_t rows = col->rows;
  if (rows == 0 || cs::cfg("CS_NO_CLASS_RUNS")) return false;
  RunsArgs a{};
  if (!runs_setup(col, d_bits, bits, s, a)) return false;
  a.rb = 1;
  a.count_only = 1;
  Buf maxima = dev_alloc(2 * sizeof(int), s);
  CS_HIP(hipMemsetAsync(maxima->p, 0, 2 * sizeof(int), s));
  a.lens = results;
  a.maxima = ptr<int>(maxima);
  const size_t lds0 = 512 + kRunsWave0 * 4;
  {
    const unsigned g0 = resident_grid(reinterpret_cast<const void*>(&k_runs_tile<0>), lds0, (a.ntiles + 3) / 4);
    ProfScope ps("k_runs_count", s);
    hipLaunchKernelGGL(k_runs_tile<0>, dim3(g0), dim3(256), lds0, s, a);
  }
  CS_HIP(hipGetLastError());
  int* host = (int*)pinned_scratch(2 * sizeof(int));
  CS_HIP(hipMemcpyAsync(host, maxima->p, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (hits) *hits = host[1];
  return true;
}

// false: the column does not take the route (the caller goes on with the automaton kernels)
bool replace_class_runs(const cs_column* col, const int32_t* d_bits, const std::vector<int32_t>& bits, const char* repl, int rb, hipStream_t s, cs_column** out) {
  const int64_t rows = col->rows;
  if (rows == 0 || rb > kMaxRunRepl || cs::cfg("CS_NO_CLASS_RUNS")) return false;
  RunsArgs a{};
  if (!runs_setup(col, d_bits, bits, s, a)) return false;
  a.rb = rb;
  for (int k = 0; k < rb; ++k) a.rep[k >> 2] |= (uint32_t)(unsigned char)repl[k] << (8 * (k & 3));
  Buf lens = dev_alloc(sizeof(int32_t) * (size_t)rows, s);
  Buf maxima = dev_alloc(2 * sizeof(int), s);
  CS_HIP(hipMemsetAsync(maxima->p, 0, 2 * sizeof(int), s));
  a.lens = ptr<int32_t>(lens);
  a.maxima = ptr<int>(maxima);
  {
    const size_t lds0 = 512 + kRunsWave0 * 4;
    const unsigned g0 = resident_grid(reinterpret_cast<const void*>(&k_runs_tile<0>), lds0, (a.ntiles + 3) / 4);
    ProfScope ps("k_runs_size", s);
    hipLaunchKernelGGL(k_runs_tile<0>, dim3(g0), dim3(256), lds0, s, a);
  }
  CS_HIP(hipGetLastError());
  // the largest output tile decides whether the write pass fits the LDS -- known right after the size pass: a column that
  // does not fit (a 16-byte replacement for `.` on dense rows) leaves HERE, before the offsets scan and before its
  // output is allocated (it used to find out after both)
  int most = 0;
  {
    int* host = (int*)pinned_scratch(sizeof(int));
    CS_HIP(hipMemcpyAsync(host, maxima->p, sizeof(int), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    most = host[0];
  }
  a.cap_out = (most + 32 + 15) & ~15;
  const size_t lds1 = 512 + (kRunsWave0 + (size_t)a.cap_out) * 4;
  if (lds1 > 150 * 1024) return false;
  auto o = std::make_unique<cs_column>();
  o->rows = rows;
  o->validity = col->validity;  // null rows stay null; columns are immutable, so share
  o->null_count = col->null_count;
  o->offsets = dev_alloc(sizeof(int64_t) * (size_t)(rows + 1), s);
  LenMeta meta;
  o->nbytes = offsets_from_lengths(ptr<int32_t>(lens), rows, ptr<int64_t>(o->offsets), s, nullptr, &meta);
  meta.give(o.get());
  o->chars = dev_alloc((size_t)o->nbytes, s);
  a.out_off = o->d_offsets();
  a.out_chars = ptr<uint8_t>(o->chars);
  if (lds1 > 48 * 1024) CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_runs_tile<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  {
    const unsigned g1 = resident_grid(reinterpret_cast<const void*>(&k_runs_tile<1>), lds1, (a.ntiles + 3) / 4);
    ProfScope ps("k_runs_write", s);
    hipLaunchKernelGGL(k_runs_tile<1>, dim3(g1), dim3(256), lds1, s, a);
  }
  CS_HIP(hipGetLastError());
  CS_HIP(hipStreamSynchronize(s));
  *out = o.release();
  return true;
}

}  // namespace cs
