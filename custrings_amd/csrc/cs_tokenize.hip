// NVText::tokenize, byte-parallel formulation (tokens.cu:41-155).
//
// The flat token column of tokenize() is the input chars stream with the delimiter bytes
// removed, plus one offset per token start (a kept byte whose predecessor in the SAME row is
// a delimiter, or that opens a row).  Both are prefix sums over bytes, so the work is done
// per byte, not per row: a wave takes a tile of R consecutive rows (R = 64, 32 or 16, chosen
// so that the tile's chars span fits the prefetch registers), every lane classifies the
// 16-byte pieces it loaded (SWAR), row starts arrive as a bitmap the row lanes set in LDS,
// and the kept bytes / token starts are ranked with popcounts and one packed wave scan per
// piece.  Two passes over the chars:
//   pass 0  per-tile kept-byte and token totals (+ their maxima, which size pass 1's LDS)
//   pass 1  compaction into an LDS tile and a coalesced flush; token start positions go
//           through LDS as 16-bit tile-relative positions and leave as 8-byte offsets.
// Waves are persistent with contiguous tile runs and the next tile's chars in flight
// (tile_utils.h).  Delimiters: whitespace (<= ' ') or up to four ASCII bytes; other delimiter
// sets (multi-byte characters) keep the per-row kernels in cs_ops.hip.
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "device_utils.h"
#include "tile_utils.h"

using namespace cs;
using namespace csdev;

namespace cs {
bool tokenize_fast(const cs_column* col, const unsigned char* delims, int ndel, hipStream_t s, cs_column** out);
}

namespace {

struct TokTileArgs {
  ColView in;
  int rows_per_tile;  // R
  long long ntiles;
  int ndel;           // 0 = whitespace
  uint32_t dpat[4];   // delimiter byte replicated into the four byte lanes
  // pass 0
  int32_t* tile_bytes;   // [ntiles]
  int32_t* tile_tokens;  // [ntiles]
  int* maxima;           // [0] most kept bytes in a tile, [1] most tokens in a tile, [2] malformed UTF-8 seen (set mode)
  // pass 1
  const int64_t* byte_base;  // [ntiles + 1]
  const int64_t* tok_base;   // [ntiles + 1]
  uint8_t* out_chars;
  int64_t* out_off;
  int cap_out, cap_tok;  // LDS bytes for the compacted tile / token slots (u16 each)
};

// bit 7 of each byte lane set when that byte is NOT a delimiter
__device__ __forceinline__ uint32_t keep_bits(uint32_t w, const TokTileArgs& a) {
  if (a.ndel == 0) {
    // > ' ': (b & 0x7f) + 0x5f carries into bit 7 for 0x21..0x7f; bytes >= 0x80 keep their own bit 7
    return (((w & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | w) & 0x80808080u;
  }
  uint32_t hit = 0;
  for (int k = 0; k < a.ndel; ++k) {
    const uint32_t x = w ^ a.dpat[k];
    hit |= ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x);  // bit 7 set where the byte equals the delimiter
  }
  return ~hit & 0x80808080u;
}
__device__ __forceinline__ uint32_t nibble(uint32_t c) {  // bits 7,15,23,31 -> bits 0..3
  return (((c >> 7) * 0x01020408u) >> 24) & 15u;
}
__device__ __forceinline__ uint32_t keep16_of(const uint4& q, const TokTileArgs& a) {
  return nibble(keep_bits(q.x, a)) | (nibble(keep_bits(q.y, a)) << 4) | (nibble(keep_bits(q.z, a)) << 8) |
         (nibble(keep_bits(q.w, a)) << 12);
}
// bit `bit` of each of the 16 bytes of the piece, as a 16-bit mask
__device__ __forceinline__ uint32_t bit16_of(const uint4& q, int bit) {
  auto g = [&](uint32_t w) { return nibble((w << (7 - bit)) & 0x80808080u); };
  return g(q.x) | (g(q.y) << 4) | (g(q.z) << 8) | (g(q.w) << 12);
}
__device__ __forceinline__ uint32_t byte_of(const uint4& q, int b) {
  const uint32_t w = b < 8 ? (b < 4 ? q.x : q.y) : (b < 12 ? q.z : q.w);
  return (w >> (8 * (b & 3))) & 255u;
}

// SEGS: some tiles span more than the staging size (the host found no tile size that fits every tile)
template <int PASS, bool SEGS>
__global__ void __launch_bounds__(256) k_tok_tile(TokTileArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  constexpr int kBitmapBytes = cstile::kPfBytes / 8 + 32;
  const int per_wave = kBitmapBytes + (PASS ? a.cap_out + 2 * a.cap_tok : 0);
  uint8_t* base = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * per_wave;
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(base);
  uint8_t* lds_out = base + kBitmapBytes;
  uint16_t* lds_tok = reinterpret_cast<uint16_t*>(lds_out + (PASS ? a.cap_out : 0));
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
  int most_bytes = 0, most_tokens = 0;
  bool malformed = false;
  for (;;) {
    const long long r0 = tile * R;
    const int nrows = (int)min((long long)R, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool live = lane < nrows && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n = live ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const long long want64 = g1 - g0 + lead;
    // A tile is taken in SEGMENTS of at most kPfBytes staged bytes: one for the tiles the host sized the kernel for (out of
    // the prefetch registers); a tile beyond that -- a long row among short ones -- is walked segment by segment straight
    // from memory, the keep / row-start state and the running totals carried across (all of it is wave-uniform), so a
    // single long row is still tokenized by the whole wave, sixteen bytes a lane.
    constexpr int kSeg = cstile::kPfChunks * 1024;
    const int nseg = SEGS ? (int)((want64 + kSeg - 1) / kSeg) : 1;
    const cstile::TileChars q0 = pf;
    const bool has_next = tile + 1 < tile_end;
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
      if (tile + 2 < tile_end) nxt = load_offs(tile + 2);
    }
    long long tile_b = 0, tile_t = 0;  // kept bytes / tokens of the segments before this one
    uint32_t carry_keep = 0;           // was the last byte of the previous chunk row kept?
    uint32_t carry_expect = 0;         // continuation bytes announced into the next chunk row (set mode check)
    for (int seg = 0; seg < (nseg > 0 ? nseg : 1); ++seg) {
      const long long seg_lo = (long long)seg * kSeg;
      const int want = (int)(want64 - seg_lo < kSeg ? want64 - seg_lo : kSeg);  // staged bytes of this segment
      const int lead_s = seg == 0 ? lead : 0;
      cstile::TileChars q = q0;
      if (SEGS && nseg > 1) {  // (wave-uniform) an oversize tile: this segment's pieces from memory
        const uint8_t* src = in.chars + (g0 - lead) + seg_lo;
#pragma unroll
        for (int j = 0; j < cstile::kPfChunks; ++j) {
          const int i = j * 1024 + lane * 16;
          q.v[j] = i < want ? *reinterpret_cast<const uint4*>(src + i) : make_uint4(0, 0, 0, 0);
        }
      }
      // row-start bitmap: one bit per byte of the staged segment, set by the row lanes
      for (int i = lane * 16; i < kBitmapBytes; i += 64 * 16) *reinterpret_cast<uint4*>(base + i) = make_uint4(0, 0, 0, 0);
      cstile::wave_lds_fence();
      if (n > 0) {
        const long long p = (long long)lead + rbeg - seg_lo;
        if (p >= 0 && p < want) __hip_atomic_fetch_or(bitmap + (p >> 5), 1u << (p & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      cstile::wave_lds_fence();

      int carry_bytes = 0, carry_tokens = 0;  // totals of the pieces before this chunk row (wave-uniform)
#pragma unroll
      for (int j = 0; j < cstile::kPfChunks; ++j) {
        if (j * 1024 < want) {  // wave-uniform
          const int i = j * 1024 + lane * 16;
          const int lo = min(16, max(0, lead_s - i)), hi = min(16, max(0, want - i));
          const uint32_t valid = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
          const uint32_t keep = keep16_of(q.v[j], a) & valid;
          const uint32_t rs = (bitmap[i >> 5] >> (i & 31)) & 0xFFFFu;
          // keep bit of the byte before this piece: lane - 1's bit 15 (lane 0: previous chunk row)
          uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(keep >> 15), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
          if (lane == 0) prev = carry_keep;
          const uint32_t before = ((keep << 1) | prev) & 0xFFFFu;  // bit b = byte b - 1 was kept
          const uint32_t starts = keep & (rs | ~before) & 0xFFFFu;
          if (PASS == 0 && a.ndel > 0) {
            // Delimiter SETS are matched per character by the row-wise routine (row_ops.h), which
            // swallows the bytes a lead byte announces; that equals this per-byte classification only
            // on well-formed UTF-8.  Check it: the continuation bytes must sit exactly where the lead
            // bytes announce them, inside the lead's own row and inside the tile.
            const uint32_t high = bit16_of(q.v[j], 7) & valid;
            uint32_t cont = 0, expect = 0;
            if (__any(high != 0)) {
              const uint32_t b6 = bit16_of(q.v[j], 6), b5 = bit16_of(q.v[j], 5), b4 = bit16_of(q.v[j], 4);
              const uint32_t lead2 = high & b6, lead3 = lead2 & b5, lead4 = lead3 & b4;
              cont = high & ~b6;
              expect = (lead2 << 1) | (lead3 << 2) | (lead4 << 3);  // bits 16..18 fall into the next piece
            }
            uint32_t carry = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(expect >> 16), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            if (lane == 0) carry = carry_expect;
            const uint32_t announced = (expect & 0xFFFFu) | carry;
            malformed |= (announced & valid) != cont || (announced & rs) != 0 || (announced & ~valid & 0xFFFFu) != 0;
            carry_expect = (uint32_t)__builtin_amdgcn_readlane((int)(expect >> 16), 63);
          }
          const int nk = __builtin_popcount(keep), nt = __builtin_popcount(starts);
          const int packed = nk | (nt << 16);
          const int incl = wave_inclusive_scan(packed);
          const int tot = __builtin_amdgcn_readlane(incl, 63);
          if (PASS == 1) {
            const int excl = incl - packed;
            const int kbase = carry_bytes + (excl & 0xFFFF);
            const int tbase = carry_tokens + (excl >> 16);
            // compaction: kept byte b of the piece lands at kbase + (kept bytes below b)
#pragma unroll
            for (int b = 0; b < 16; ++b)
              if ((keep >> b) & 1u) lds_out[kbase + __builtin_popcount(keep & ((1u << b) - 1u))] = (uint8_t)byte_of(q.v[j], b);
            // token starts: segment-relative position of each, in token order
            uint32_t st = starts;
            int t = tbase;
            while (__any(st != 0)) {
              if (st != 0) {
                const int b = __builtin_ctz(st);
                st &= st - 1;
                lds_tok[t++] = (uint16_t)(kbase + __builtin_popcount(keep & ((1u << b) - 1u)));
              }
            }
          }
          carry_bytes += tot & 0xFFFF;
          carry_tokens += tot >> 16;
          carry_keep = (uint32_t)__builtin_amdgcn_readlane((int)(keep >> 15), 63);
        }
      }
      if (PASS == 0) {
        most_bytes = max(most_bytes, carry_bytes);  // (per SEGMENT: what pass 1's LDS regions must hold)
        most_tokens = max(most_tokens, carry_tokens);
      } else {
        cstile::wave_lds_fence();
        const long long cb = a.byte_base[tile] + tile_b, tb = a.tok_base[tile] + tile_t;
        cstile::wave_flush_shift(a.out_chars + cb, carry_bytes, lds_out, lane);
        cstile::gptr<int64_t> oo = cstile::as_global(a.out_off + tb);
        for (int t = lane; t < carry_tokens; t += 64) oo[t] = cb + lds_tok[t];
        cstile::wave_lds_fence();  // the next segment / tile reuses the regions
      }
      tile_b += carry_bytes;
      tile_t += carry_tokens;
    }
    if (PASS == 0) {
      malformed |= carry_expect != 0;  // a sequence cut off by the end of the tile's last row
      if (lane == 0) {
        a.tile_bytes[tile] = (int32_t)tile_b;
        a.tile_tokens[tile] = (int32_t)tile_t;
      }
    }
    if (!has_next) break;
    ++tile;
  }
  if (PASS == 0 && __any(malformed) && lane == 0) atomicOr(reinterpret_cast<unsigned*>(a.maxima + 2), 1u);
  if (PASS == 0 && lane == 0) {
    if (most_bytes > __hip_atomic_load(a.maxima, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.maxima, most_bytes);
    if (most_tokens > __hip_atomic_load(a.maxima + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.maxima + 1, most_tokens);
  }
}

}  // namespace

namespace cs {

bool tokenize_fast(const cs_column* col, const unsigned char* delims, int ndel, hipStream_t s, cs_column** out) {
  const int64_t rows = col->rows;
  if (rows == 0 || ndel > 4 || cs::cfg("CS_TOKENIZE_ROWWISE")) return false;
  for (int k = 0; k < ndel; ++k)
    if (delims[k] == 0 || delims[k] >= 128) return false;
  // rows per tile: the largest of 64 / 32 / 16 whose widest tile fits the prefetch registers
  int R = 0;
  for (int r : {64, 32, 16}) {
    if (max_span_rows(col, r, s) + 16 <= cstile::kPfBytes) {
      R = r;
      break;
    }
  }
  // (no tile size fits every tile: 64-row tiles, the oversize ones taken in segments by the kernel -- byte-parallel still)
  bool segs = false;
  if (!R && max_span64(col, s) < ((int64_t)1 << 30) && !cs::cfg("CS_NO_OUTLIER_TILES")) {
    R = 64;
    segs = true;
  }
  if (!R) return false;
  TokTileArgs a{};
  a.in = view_of(col);
  a.rows_per_tile = R;
  a.ntiles = (rows + R - 1) / R;
  a.ndel = ndel;
  for (int k = 0; k < ndel; ++k) a.dpat[k] = 0x01010101u * delims[k];
  Buf counts = dev_alloc(sizeof(int32_t) * 2 * a.ntiles, s);
  Buf maxima = dev_alloc(4 * sizeof(int), s);
  CS_HIP(hipMemsetAsync(maxima->p, 0, 4 * sizeof(int), s));
  a.tile_bytes = ptr<int32_t>(counts);
  a.tile_tokens = ptr<int32_t>(counts) + a.ntiles;
  a.maxima = ptr<int>(maxima);
  constexpr size_t kBitmapBytes = cstile::kPfBytes / 8 + 32;
  {
    const size_t lds0 = kBitmapBytes * 4;
    auto k0 = segs ? &k_tok_tile<0, true> : &k_tok_tile<0, false>;
    const unsigned g0 = resident_grid(reinterpret_cast<const void*>(k0), lds0, (a.ntiles + 3) / 4);
    ProfScope ps("k_tok_count", s);
    hipLaunchKernelGGL(k0, dim3(g0), dim3(256), lds0, s, a);
  }
  CS_HIP(hipGetLastError());
  // per-tile positions in the output chars and in the token sequence
  Buf bases = dev_alloc(sizeof(int64_t) * 2 * (a.ntiles + 1), s);
  int64_t totals[2];
  offsets_from_lengths_segmented(ptr<int32_t>(counts), a.ntiles, 2, ptr<int64_t>(bases), totals, s);
  int* hmax = (int*)pinned_scratch(4 * sizeof(int));
  CS_HIP(hipMemcpyAsync(hmax, maxima->p, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (hmax[2]) return false;  // malformed UTF-8 with a delimiter set: the row-wise routine defines the result
  const int64_t nbytes = totals[0], ntok = totals[1];
  auto c = std::make_unique<cs_column>();
  c->rows = ntok;
  c->null_count = 0;
  c->drops = 0;  // a token is never empty and never null
  c->nbytes = nbytes;
  c->offsets = dev_alloc(sizeof(int64_t) * (ntok + 1), s);
  c->chars = dev_alloc((size_t)nbytes, s);
  if (ntok == 0) {
    CS_HIP(hipMemsetAsync(c->offsets->p, 0, sizeof(int64_t), s));
    *out = c.release();
    return true;
  }
  a.byte_base = ptr<const int64_t>(bases);
  a.tok_base = ptr<const int64_t>(bases) + (a.ntiles + 1);
  a.out_chars = ptr<uint8_t>(c->chars);
  a.out_off = ptr<int64_t>(c->offsets);
  a.cap_out = (hmax[0] + 32 + 15) & ~15;
  a.cap_tok = (hmax[1] + 8 + 7) & ~7;
  const size_t lds1 = (kBitmapBytes + (size_t)a.cap_out + 2 * (size_t)a.cap_tok) * 4;
  if (lds1 > 150 * 1024) return false;
  auto k1 = segs ? &k_tok_tile<1, true> : &k_tok_tile<1, false>;
  if (lds1 > 48 * 1024) CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  {
    const unsigned g1 = resident_grid(reinterpret_cast<const void*>(k1), lds1, (a.ntiles + 3) / 4);
    ProfScope ps("k_tok_write", s);
    hipLaunchKernelGGL(k1, dim3(g1), dim3(256), lds1, s, a);
  }
  CS_HIP(hipGetLastError());
  // closing offset
  CS_HIP(hipMemcpyAsync(ptr<int64_t>(c->offsets) + ntok, &totals[0], sizeof(int64_t), hipMemcpyHostToDevice, s));
  CS_HIP(hipStreamSynchronize(s));
  *out = c.release();
  return true;
}

}  // namespace cs
