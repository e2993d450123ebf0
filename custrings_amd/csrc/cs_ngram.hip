// NVText::create_ngrams over tiles (ngram.cu:32-110), for the case where no token is dropped
// (no null and no empty row: every tokenize() output) and the separator is at most 8 bytes.
//
// With every token kept, the output offsets have a closed form in the input offsets:
//   out_off[g] = sum_{k<n} (off[g + k] - off[k]) + g * (n - 1) * |sep|
// so there is no size pass and no scan.  A wave takes NG = 64 * M consecutive n-grams; their
// tokens' chars are one contiguous span (prefetched into registers, staged in LDS), the
// input offsets of the tile sit in LDS relative to its first token, every lane assembles M
// n-grams in the LDS output tile -- tokens as funnel-shifted dwords OR-ed into a zeroed
// buffer (up to 16 bytes at once), separators the same way -- and the tile is flushed with
// 16-byte stores.  Output offsets leave as coalesced 8-byte stores.
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "device_utils.h"
#include "tile_utils.h"

using namespace cs;
using namespace csdev;

namespace cs {
bool ngrams_fast(const cs_column* tokens, int n, const unsigned char* sep, int sepn, hipStream_t s, cs_column** out);
}

namespace {

struct NgramArgs {
  ColView in;
  int n, sepn, M;  // n-gram order, separator bytes, n-grams per lane per tile
  uint32_t sep0, sep1;
  long long ng, ntiles;  // number of n-grams, tiles
  int64_t head;          // sum_{k<n} off[k]
  int64_t* out_off;
  uint8_t* out_chars;
  int cap_in, cap_out;   // LDS bytes
  unsigned* error;
};

__device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v) {
  __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// ORs `len` bytes (the low bytes of t0..t3, garbage above `len` allowed) into the zeroed
// buffer at byte index di; len <= 16
// (`tail`: seventeen masks of four dwords in LDS, entry n = the first n bytes -- one aligned 16-byte read and four ANDs where the
// mask was built with 64-bit shifts, the slowest integer instructions of the machine)
__device__ __forceinline__ void or_bytes(uint8_t* region, int di, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, int len, const cstile::u32x4* tail) {
  const cstile::u32x4 m = tail[len];
  t0 &= m.x, t1 &= m.y, t2 &= m.z, t3 &= m.w;
  const unsigned sd = (unsigned)(di & 3);
  uint32_t* dp = reinterpret_cast<uint32_t*>(region) + (di >> 2);
  uint32_t d0 = t0, d1 = t1, d2 = t2, d3 = t3, d4 = 0;
  if (sd) {
    const unsigned up = 4 - sd;
    d0 = t0 << (8 * sd);
    d1 = __builtin_amdgcn_alignbyte(t1, t0, up);
    d2 = __builtin_amdgcn_alignbyte(t2, t1, up);
    d3 = __builtin_amdgcn_alignbyte(t3, t2, up);
    d4 = t3 >> (8 * up);
  }
  lds_or(dp, d0);
  lds_or(dp + 1, d1);
  if (__any(len + (int)sd > 8)) {
    lds_or(dp + 2, d2);
    lds_or(dp + 3, d3);
    lds_or(dp + 4, d4);
  }
}

__global__ void __launch_bounds__(256) k_ngram_tile(NgramArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  const int NG = 64 * a.M;
  const int nrel = NG + a.n + 1;                       // relative offsets kept per tile
  const int rel_bytes = (nrel * 4 + 15) & ~15;
  cstile::u32x4* tail = reinterpret_cast<cstile::u32x4*>(smem);  // tail[n], n = 0..16 (or_bytes); 288 bytes in front of the waves' regions
  if (threadIdx.x <= 16) {
    auto first = [](int k) -> uint32_t { return k >= 4 ? 0xFFFFFFFFu : (k <= 0 ? 0u : (1u << (8 * k)) - 1u); };
    const int t = (int)threadIdx.x;
    tail[t] = cstile::u32x4{first(t), first(t - 4), first(t - 8), first(t - 12)};
  }
  __syncthreads();
  uint8_t* base = reinterpret_cast<uint8_t*>(smem) + 288 + (size_t)wv * (rel_bytes + a.cap_in + a.cap_out);
  int32_t* relo = reinterpret_cast<int32_t*>(base);     // relo[t] = off[G0 + t] - off[G0]
  uint8_t* lds_in = base + rel_bytes;
  uint8_t* lds_out = lds_in + a.cap_in;
  const ColView& in = a.in;
  const long long waves = (long long)gridDim.x * 4;
  const long long per = (a.ntiles + waves - 1) / waves;
  long long tile = ((long long)blockIdx.x * 4 + wv) * per;
  const long long tile_end = min(a.ntiles, tile + per);
  if (tile >= tile_end) return;
  const long long step = (long long)(a.n - 1) * a.sepn;  // separator bytes per n-gram

  // chars span of a tile: tokens [G0, G0 + ngt + n - 1)
  auto span_of = [&](long long t, long long& c0, long long& c1) {
    const long long G0 = t * NG;
    const long long ngt = min((long long)NG, a.ng - G0);
    c0 = in.offsets[G0];
    c1 = in.offsets[G0 + ngt + a.n - 1];
  };
  long long c0, c1;
  span_of(tile, c0, c1);
  cstile::TileChars pf;
#pragma unroll
  for (int j = 0; j < cstile::kPfChunks; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, c0, c1, lane, pf);
  for (;;) {
    const long long G0 = tile * NG;
    const int ngt = (int)min((long long)NG, a.ng - G0);
    const int lead = (int)((uintptr_t)(in.chars + c0) & 15);
    const int want = (int)(c1 - c0) + lead;
    const bool bad = want + 16 > a.cap_in;
    if (!bad) cstile::stage_chars(lds_in, want, lane, pf);
    // relative input offsets of the tile's tokens (coalesced loads), and sum_{k<n} off[G0 + k]
    for (int t = lane; t < ngt + a.n; t += 64) relo[t] = (int32_t)(in.offsets[G0 + t] - c0);
    long long nc0 = c0, nc1 = c1;
    const bool has_next = tile + 1 < tile_end;
    if (has_next) {
      span_of(tile + 1, nc0, nc1);
      cstile::issue_chars(in.chars, nc0, nc1, lane, pf);
    }
    cstile::wave_lds_fence();
    int lead_sum = 0;  // sum_{k<n} relo[k]
    for (int k = 0; k < a.n; ++k) lead_sum += relo[k];
    const long long out_base = (long long)a.n * c0 + lead_sum - a.head + G0 * step;  // out_off[G0]
    // tile-relative output position of local n-gram gl
    auto rel_out = [&](int gl) {
      int sum = 0;
      for (int k = 0; k < a.n; ++k) sum += relo[gl + k];
      return sum - lead_sum + gl * (int)step;
    };
    const int out_span = rel_out(ngt);  // (relo has ngt + n entries: gl = ngt is the end position)
    const bool bad2 = bad || out_span + 32 > a.cap_out;
    if (bad2) {
      if (lane == 0) atomicOr(a.error, 1u);
    } else {
      for (int i = lane * 16; i < out_span + 20; i += 64 * 16) *reinterpret_cast<uint4*>(lds_out + i) = make_uint4(0, 0, 0, 0);
      cstile::wave_lds_fence();
      for (int m = 0; m < a.M; ++m) {
        const int gl = m * 64 + lane;
        const bool on = gl < ngt;
        int ro = 0;
        if (on) {
          ro = rel_out(gl);
          cstile::as_global(a.out_off + G0)[gl] = out_base + ro;
        }
        int di = ro;
        for (int k = 0; k < a.n; ++k) {
          int ts = 0, len = 0;
          if (on) {
            ts = relo[gl + k];
            len = relo[gl + k + 1] - ts;
          }
          // first 16 bytes of the token (funnel shift from aligned source words)
          const int si = lead + ts;
          const uint32_t* sp = reinterpret_cast<const uint32_t*>(lds_in) + (si >> 2);
          const unsigned sh = (unsigned)(si & 3);
          uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0;
          if (on) {
            w0 = sp[0], w1 = sp[1], w2 = sp[2], w3 = sp[3], w4 = sp[4];
          }
          const int first = len < 16 ? len : 16;
          if (on)
            or_bytes(lds_out, di, __builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                     __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh), first, tail);
          if (on && len > 16) cstile::lds_copy(lds_out, di + 16, lds_in, si + 16, len - 16);
          di += len;
          if (k + 1 < a.n && a.sepn) {
            if (on) or_bytes(lds_out, di, a.sep0, a.sep1, 0, 0, a.sepn, tail);
            di += a.sepn;
          }
        }
      }
      if (G0 + ngt == a.ng && lane == 0) a.out_off[a.ng] = out_base + out_span;
      cstile::wave_lds_fence();
      cstile::wave_flush_shift(a.out_chars + out_base, out_span, lds_out, lane);
      cstile::wave_lds_fence();
    }
    if (!has_next) break;
    ++tile;
    c0 = nc0;
    c1 = nc1;
  }
}

__global__ void k_ngram_spans(const int64_t* __restrict__ off, long long ng, int n, int NG, long long step,
                              unsigned long long* __restrict__ maxima) {
  // maxima[0] = widest input span of a tile, maxima[1] = widest output span
  const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long G0 = t * NG;
  int vin = 0, vout = 0;
  if (G0 < ng) {
    const long long ngt = min((long long)NG, ng - G0);
    const long long sin = off[G0 + ngt + n - 1] - off[G0];
    long long sout = ngt * step;
    for (int k = 0; k < n; ++k) sout += off[G0 + ngt + k] - off[G0 + k];
    vin = (int)min(sin, 0x7fffffffll);
    vout = (int)min(sout, 0x7fffffffll);
  }
  const int mi = block_reduce_max(vin), mo = block_reduce_max(vout);
  if (threadIdx.x == 0) {
    // (only a workgroup that would raise a maximum issues the same-address atomic)
    if ((unsigned long long)mi > __hip_atomic_load(maxima, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxima, (unsigned long long)mi);
    if ((unsigned long long)mo > __hip_atomic_load(maxima + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxima + 1, (unsigned long long)mo);
  }
}

// sets *flag when some row is null or empty (such rows are dropped by create_ngrams)
__global__ void k_ngram_check(ColView in, unsigned* __restrict__ flag) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool drop = false;
  if (r < in.rows) drop = !row_is_valid(in.validity, r) || in.offsets[r + 1] == in.offsets[r];
  if (__any(drop) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

}  // namespace

namespace cs {

bool ngrams_fast(const cs_column* tokens, int n, const unsigned char* sep, int sepn, hipStream_t s, cs_column** out) {
  const int64_t rows = tokens->rows;
  if (n < 2 || n > 8 || sepn > 8 || rows <= n || cs::cfg("CS_NGRAM_ROWWISE")) return false;
  if (tokens->drops < 0) {  // remembered on the immutable column (the tokenizer's output is born with the answer)
    Buf drop = dev_alloc(sizeof(unsigned), s);
    CS_HIP(hipMemsetAsync(drop->p, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_ngram_check, dim3(blocks_for(rows)), dim3(kBlock), 0, s, view_of(tokens), ptr<unsigned>(drop));
    unsigned* h = (unsigned*)pinned_scratch(sizeof(unsigned));
    CS_HIP(hipMemcpyAsync(h, drop->p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    tokens->drops = *h ? 1 : 0;
  }
  if (tokens->drops) return false;  // some rows are dropped: the closed-form offsets do not apply
  const int64_t ng = rows - n + 1;
  const long long step = (long long)(n - 1) * sepn;
  // n-grams per lane per tile: the largest M whose widest tile fits the staging buffers
  int M = 0;
  int64_t span_in = 0, span_out = 0;
  Buf maxima = dev_alloc(16, s);
  const int m_first = cs::cfg_int("CS_NGRAM_M", 8);  // (measurement: n-grams per lane and tile)
  for (int m : {8, 4, 2, 1}) {
    if (m > m_first) continue;
    const int NG = 64 * m;
    const int64_t nt = (ng + NG - 1) / NG;
    CS_HIP(hipMemsetAsync(maxima->p, 0, 16, s));
    hipLaunchKernelGGL(k_ngram_spans, dim3(blocks_for(nt)), dim3(kBlock), 0, s, tokens->d_offsets(), (long long)ng, n, NG,
                       step, ptr<unsigned long long>(maxima));
    int64_t* h = (int64_t*)pinned_scratch(16);
    CS_HIP(hipMemcpyAsync(h, maxima->p, 16, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (h[0] + 32 <= cstile::kPfBytes && h[1] + 64 <= 12 * 1024) {
      // (... preferring, down to four n-grams a lane, one whose LDS leaves room for four workgroups a CU -- the kernel's registers
      // allow four waves a SIMD: C5 bigrams at M = 8 / 4 / 2 / 1: 10.4 / 9.4 / 11.7 / 16.2 ms, the first at three workgroups of 47 KB)
      const size_t wave_lds = (((size_t)(NG + n + 1) * 4 + 15) & ~(size_t)15) + (size_t)((h[0] + 32 + 15) & ~(int64_t)15) + (size_t)((h[1] + 64 + 15) & ~(int64_t)15);
      const bool roomy = wave_lds * 4 <= 40 * 1024;
      if (!M || roomy) {
        M = m;
        span_in = h[0];
        span_out = h[1];
      }
      if (roomy || m <= 4) break;
    }
  }
  if (!M) return false;
  NgramArgs a{};
  a.in = view_of(tokens);
  a.n = n;
  a.sepn = sepn;
  a.M = M;
  for (int i = 0; i < sepn; ++i) {
    if (i < 4) a.sep0 |= (uint32_t)sep[i] << (8 * i);
    else a.sep1 |= (uint32_t)sep[i] << (8 * (i - 4));
  }
  a.ng = ng;
  a.ntiles = (ng + 64 * M - 1) / (64 * M);
  // sum_{k<n} off[k] and the total size need off[0..n-1] and off[rows-n+1..rows]
  std::vector<int64_t> headv(n), tailv(n);
  CS_HIP(hipMemcpyAsync(headv.data(), tokens->d_offsets(), sizeof(int64_t) * n, hipMemcpyDeviceToHost, s));
  CS_HIP(hipMemcpyAsync(tailv.data(), tokens->d_offsets() + ng, sizeof(int64_t) * n, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  int64_t head = 0, total = ng * step;
  for (int k = 0; k < n; ++k) {
    head += headv[k];
    total += tailv[k] - headv[k];
  }
  a.head = head;
  auto c = std::make_unique<cs_column>();
  c->rows = ng;
  c->null_count = 0;
  c->nbytes = total;
  c->offsets = dev_alloc(sizeof(int64_t) * (ng + 1), s);
  c->chars = dev_alloc((size_t)total, s);
  a.out_off = ptr<int64_t>(c->offsets);
  a.out_chars = ptr<uint8_t>(c->chars);
  a.cap_in = (int)((span_in + 32 + 15) & ~(int64_t)15);
  a.cap_out = (int)((span_out + 64 + 15) & ~(int64_t)15);
  Buf err = dev_alloc(sizeof(unsigned), s);
  CS_HIP(hipMemsetAsync(err->p, 0, sizeof(unsigned), s));
  a.error = ptr<unsigned>(err);
  const size_t rel_bytes = ((size_t)(64 * M + n + 1) * 4 + 15) & ~(size_t)15;
  const size_t lds = 288 + (rel_bytes + (size_t)a.cap_in + (size_t)a.cap_out) * 4;
  if (lds > 150 * 1024) return false;
  if (lds > 48 * 1024)
    CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ngram_tile), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  {
    const unsigned g = resident_grid(reinterpret_cast<const void*>(&k_ngram_tile), lds, (a.ntiles + 3) / 4);
    ProfScope ps("k_ngram_write", s);
    hipLaunchKernelGGL(k_ngram_tile, dim3(g), dim3(256), lds, s, a);
  }
  CS_HIP(hipGetLastError());
  unsigned* h = (unsigned*)pinned_scratch(sizeof(unsigned));
  CS_HIP(hipMemcpyAsync(h, err->p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (*h) return false;
  *out = c.release();
  return true;
}

}  // namespace cs
