// NVStrings::lower / upper over row tiles (case.cu:31-170).
//
// Case mapping almost never changes a character's UTF-8 width, and ASCII text needs no
// table at all, so the common case is a streaming pass: a wave takes a tile of R consecutive
// rows (their chars are one contiguous span), every lane flips the ASCII letters of the
// 16-byte pieces it loaded with SWAR arithmetic and writes them to the output tile in LDS,
// and only rows that contain non-ASCII bytes (found through a per-byte bitmap the piece
// lanes leave in LDS) are redone by their row lane with the same per-character routine the
// row-wise kernels use (row_ops.h: decode, flag/case tables, re-encode) -- LDS to LDS.  The
// tile is then flushed with 16-byte stores to the same positions it came from: the output
// shares the input's offsets and validity buffers.  A row whose mapped size differs from its
// input size raises a flag and the host recomputes the column with the two-pass row kernels.
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "device_utils.h"
#include "row_ops.h"
#include "tile_utils.h"

using namespace cs;
using namespace csdev;
using namespace csrow;

namespace cs {
bool change_case_fast(const cs_column* col, unsigned bit, bool ascii_rule_ok, hipStream_t s, cs_column** out);
}

namespace {

struct CaseTileArgs {
  ColView in;
  int rows_per_tile;
  long long ntiles;
  unsigned bit;  // 32: to lower, 64: to upper (flag bit of the characters to change)
  const uint8_t* flags;
  const uint16_t* cases;
  uint8_t* out_chars;
  unsigned* changed;  // set when a row's size would change
  int cap;            // LDS bytes per tile buffer
};

// ASCII letters of the other case, flipped (bit 5 toggled); other bytes untouched
__device__ __forceinline__ uint32_t flip_ascii(uint32_t w, unsigned bit) {
  const uint32_t x = w & 0x7F7F7F7Fu;
  // to lower: bytes 0x41..0x5A; to upper: 0x61..0x7A
  const uint32_t lo = bit == 32 ? 0x3F3F3F3Fu : 0x1F1F1F1Fu;  // 0x80 - first letter
  const uint32_t hi = bit == 32 ? 0x25252525u : 0x05050505u;  // 0x7F - last letter
  const uint32_t m = (x + lo) & ~(x + hi) & ~w & 0x80808080u;
  return w ^ (m >> 2);
}

__global__ void __launch_bounds__(256) k_case_tile(CaseTileArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  constexpr int kBitmapBytes = cstile::kPfBytes / 8 + 32;
  uint8_t* base = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (2 * a.cap + kBitmapBytes);
  uint8_t* lds_in = base;
  uint8_t* lds_out = base + a.cap;
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(base + 2 * a.cap);  // bit i: byte i of the tile is >= 0x80
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
    if (want64 + 16 > a.cap) {
      // A tile beyond the staging buffer (the host sized it for all but a few tiles: one long row among millions of
      // short ones): its rows are mapped a thread each, straight from memory, as the row-wise kernels do.
      // (first the whole span with the wave, sixteen bytes a lane: an ASCII tile -- the usual case, and a long row would
      // otherwise keep ONE lane busy for milliseconds -- is done after that; only a tile with other bytes goes row by row)
      bool high = false;
      {
        const uint8_t* src = in.chars + (g0 - lead);
        uint8_t* dst = a.out_chars + (g0 - lead);
        for (long long i = (long long)lane * 16; i < want64; i += 64 * 16) {
          const uint4 q = *reinterpret_cast<const uint4*>(src + i);
          high |= ((q.x | q.y | q.z | q.w) & 0x80808080u) != 0;
          uint4 o;
          o.x = flip_ascii(q.x, a.bit);
          o.y = flip_ascii(q.y, a.bit);
          o.z = flip_ascii(q.z, a.bit);
          o.w = flip_ascii(q.w, a.bit);
          const long long lo = lead - i, hi = want64 - i;  // the span's bytes inside this piece: [lo, hi)
          if (lo <= 0 && hi >= 16) {
            *reinterpret_cast<uint4*>(dst + i) = o;
          } else {
            const uint32_t w[4] = {o.x, o.y, o.z, o.w};
            for (int k = (int)(lo > 0 ? lo : 0); k < (int)(hi < 16 ? hi : 16); ++k) dst[i + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
          }
        }
      }
      if (__any(high)) {
        // the row lanes rewrite bytes the piece lanes have just stored: the first stores must have left the wave before
        // the second ones are issued (two stores to one address from different lanes are not ordered otherwise)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (__any(high) && n > 0) {
        const uint8_t* p = in.chars + (g0 + rbeg);
        if (row_case_size(p, n, a.flags, a.cases, a.bit) != n) atomicOr(a.changed, 1u);
        else row_case_write(p, n, a.flags, a.cases, a.bit, a.out_chars + (g0 + rbeg));
      }
      const bool more = tile + 1 < tile_end;
      if (more) {
        cur = nxt;
        cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
        if (tile + 2 < tile_end) nxt = load_offs(tile + 2);
      }
      if (!more) break;
      ++tile;
      continue;
    }
    const int want = (int)want64;
    // pieces: keep the input in LDS for the row lanes, ASCII-flipped copy in the output tile,
    // one "byte >= 0x80" bit per byte in the bitmap
    uint32_t any_high = 0;
#pragma unroll
    for (int j = 0; j < cstile::kPfChunks; ++j) {
      const int i = j * 1024 + lane * 16;
      if (i < want) {
        const uint4 q = pf.v[j];
        *reinterpret_cast<uint4*>(lds_in + i) = q;
        uint4 o;
        o.x = flip_ascii(q.x, a.bit);
        o.y = flip_ascii(q.y, a.bit);
        o.z = flip_ascii(q.z, a.bit);
        o.w = flip_ascii(q.w, a.bit);
        *reinterpret_cast<uint4*>(lds_out + i) = o;
        const uint32_t hx = q.x & 0x80808080u, hy = q.y & 0x80808080u, hz = q.z & 0x80808080u, hw = q.w & 0x80808080u;
        any_high |= hx | hy | hz | hw;
        const uint32_t bits = ((((hx >> 7) * 0x01020408u) >> 24) & 15u) | (((((hy >> 7) * 0x01020408u) >> 24) & 15u) << 4) |
                              (((((hz >> 7) * 0x01020408u) >> 24) & 15u) << 8) | (((((hw >> 7) * 0x01020408u) >> 24) & 15u) << 12);
        reinterpret_cast<uint16_t*>(bitmap)[i >> 4] = (uint16_t)bits;
      }
    }
    const bool has_next = tile + 1 < tile_end;
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
      if (tile + 2 < tile_end) nxt = load_offs(tile + 2);
    }
    cstile::wave_lds_fence();
    if (__any(any_high != 0)) {  // some row of the tile holds non-ASCII characters
      // Row lanes visit only the non-ASCII bytes of their row (bitmap bits), in order: a lead byte
      // followed by exactly its continuation bytes is mapped through the tables and patched into
      // the output tile; anything else (stray or missing continuation bytes) sends the row through
      // the sequential routines of row_ops.h, which define the behaviour for malformed input.
      if (n > 0) {
        const int p0 = lead + rbeg, p1 = p0 + n;  // this row's bits [p0, p1)
        const uint8_t* p = lds_in + p0;
        uint8_t* o = lds_out + p0;
        bool any = false, malformed = false, resized = false;
        int expect = 0, last = -2;
        for (int w = p0 >> 5; w <= (p1 - 1) >> 5; ++w) {
          uint32_t m = bitmap[w];
          if (w == (p0 >> 5)) m &= 0xFFFFFFFFu << (p0 & 31);
          if (w == ((p1 - 1) >> 5) && (p1 & 31)) m &= ~(0xFFFFFFFFu << (p1 & 31));
          while (m) {
            const int i = (w << 5) + __builtin_ctz(m) - p0;  // row offset of this non-ASCII byte
            m &= m - 1;
            any = true;
            const uint8_t b = p[i];
            if (expect > 0) {
              malformed |= !is_cont(b) || i != last + 1;
              --expect;
            } else {
              const unsigned w8 = lead_width(b);
              malformed |= w8 < 2 || i + (int)w8 > n;
              expect = (int)w8 - 1;
              if (!malformed) {
                Char ch;
                decode_at(p, i, n, ch);
                const unsigned u = packed_to_cp(ch);
                const unsigned f = u <= 0xFFFF ? a.flags[u] : 0;
                if (f & a.bit) {
                  const Char nc = cp_to_packed(a.cases[u]);
                  if (packed_width(nc) != w8) resized = true;
                  else
                    for (unsigned k = 0; k < w8; ++k) o[i + (int)k] = (uint8_t)(nc >> (8 * (w8 - 1 - k)));
                }
              }
            }
            last = i;
          }
        }
        malformed |= expect != 0;
        if (any && malformed) {
          if (row_case_size(p, n, a.flags, a.cases, a.bit) != n) resized = true;
          else row_case_write(p, n, a.flags, a.cases, a.bit, o);
        }
        if (resized) atomicOr(a.changed, 1u);
      }
      cstile::wave_lds_fence();
    }
    cstile::wave_flush(a.out_chars + g0, (int)(g1 - g0), lds_out, lead, lane);
    cstile::wave_lds_fence();
    if (!has_next) break;
    ++tile;
  }
}

}  // namespace

namespace cs {

bool change_case_fast(const cs_column* col, unsigned bit, bool ascii_rule_ok, hipStream_t s, cs_column** out) {
  const int64_t rows = col->rows;
  if (rows == 0 || !ascii_rule_ok || col->nbytes == 0 || cs::cfg("CS_CASE_ROWWISE")) return false;
  int R = 0;
  for (int r : {64, 32, 16}) {
    if (max_span_rows(col, r, s) + 32 <= cstile::kPfBytes) {
      R = r;
      break;
    }
  }
  int64_t span = R ? max_span_rows(col, R, s) : 0;
  if (!R && !cs::cfg("CS_NO_OUTLIER_TILES")) {
    // no tile size fits every tile (one long row among short ones, or rows of hundreds of bytes throughout): 64-row tiles,
    // the kernel maps a tile beyond the staging size with the whole wave, sixteen bytes a lane, straight from memory --
    // row by row only when the tile holds non-ASCII bytes
    R = 64;
    span = cstile::kPfBytes - 64;
  }
  if (!R) return false;
  CaseTileArgs a{};
  a.in = view_of(col);
  a.rows_per_tile = R;
  a.ntiles = (rows + R - 1) / R;
  a.bit = bit;
  a.flags = d_unicode_flags();
  a.cases = d_charcases();
  a.cap = (int)((span + 32 + 15) & ~(int64_t)15);
  Buf chars = dev_alloc((size_t)col->nbytes, s);
  Buf flag = dev_alloc(sizeof(unsigned), s);
  CS_HIP(hipMemsetAsync(flag->p, 0, sizeof(unsigned), s));
  a.out_chars = ptr<uint8_t>(chars);
  a.changed = ptr<unsigned>(flag);
  constexpr size_t kBitmapBytes = cstile::kPfBytes / 8 + 32;
  const size_t lds = (2 * (size_t)a.cap + kBitmapBytes) * 4;
  if (lds > 150 * 1024) return false;
  if (lds > 48 * 1024)
    CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_case_tile), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  {
    const unsigned g = resident_grid(reinterpret_cast<const void*>(&k_case_tile), lds, (a.ntiles + 3) / 4);
    ProfScope ps(bit == 32 ? "k_lower_write" : "k_upper_write", s);
    hipLaunchKernelGGL(k_case_tile, dim3(g), dim3(256), lds, s, a);
  }
  CS_HIP(hipGetLastError());
  unsigned* h = (unsigned*)pinned_scratch(sizeof(unsigned));
  CS_HIP(hipMemcpyAsync(h, flag->p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (*h) return false;  // some row changes size: the two-pass row kernels recompute the column
  auto* o = new cs_column;
  o->rows = rows;
  o->nbytes = col->nbytes;
  o->null_count = col->null_count;
  o->max_span64 = col->max_span64;
  o->max_row = col->max_row;
  col->share_extents_with(o);    // same row extents: share the immutable buffers
  o->validity = col->validity;
  o->chars = chars;
  *out = o;
  return true;
}

}  // namespace cs
