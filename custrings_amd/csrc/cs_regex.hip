// Regex ops: contains_re / match / count_re / replace_re.
//
// One thread per row.  The compiled program image (instructions, classes,
// ASCII bitmaps, first-character prefilter) is staged into LDS once per
// workgroup; each thread's two ordered thread lists + closure stack also live
// in LDS, interleaved across lanes (slot-major) so accesses are bank-conflict
// free.  Programs whose lists do not fit in LDS run with the same code against
// a global scratch arena (the reference switches to a global `relists_mem`
// above 1000 instructions, regexec.cpp:81-95; count.cu:74-85).
// Workgroups walk the rows grid-stride so the arena is bounded by the grid.
#include <hip/hip_runtime.h>

#include <atomic>
#include <list>

#include <cstring>

#include "cs_internal.h"
#include "device_utils.h"
#include "regex_program.h"
#include "regex_tdfa.h"
#include "regex_bits.h"
#include "regex_vm.h"
#include <mutex>

#include "tile_utils.h"

using namespace cs;
using namespace csdev;

struct cs_regex {
  csrx::Program prog;
  std::vector<int32_t> blob;   // program only (ABI: cs_regex_blob)
  std::vector<int32_t> image;  // blob + executor extras
  std::vector<int32_t> tdfa;   // tagged DFA image (empty = not convertible)
  std::vector<int32_t> gtags;  // capture-group tag image of the tagged DFA (empty = no groups / not convertible)
  std::vector<int32_t> bits;   // bit-parallel form (regex_bits.h; empty = the program does not convert)
  Buf d_image;                 // uploaded lazily
  Buf d_tdfa;
  Buf d_gtags;
  Buf d_bits;
  bool empty_pattern = false;
  std::atomic<int> refs{1};    // handles given out by cs_regex_compile + one for the list of kept patterns
};

namespace cs {
bool replace_class_runs(const cs_column* col, const int32_t* d_bits, const std::vector<int32_t>& bits, const char* repl, int rb, hipStream_t s, cs_column** out);
bool count_class_runs(const cs_column* col, const int32_t* d_bits, const std::vector<int32_t>& bits, hipStream_t s, int32_t* results, int64_t* hits);
thread_local int g_replace_plain_only = 0;  // set by cs_replace around its call of cs_replace_re: single-pass kernel or nothing
// cs_replace also leaves the needle itself here when it has at most eight bytes and no border (no proper prefix that is
// also a suffix: occurrences cannot overlap): the stream kernel then finds the matches by byte comparison, all bytes of
// the sub-tile at once, instead of walking the DFA
thread_local unsigned long long g_replace_literal = 0;
thread_local int g_replace_literal_len = 0;
// cs_replace_with_backrefs hands its template to the same single-pass kernel (the device copy of the template struct);
// nullptr: plain replace_re
thread_local const void* g_backrefs_dev = nullptr;
thread_local int g_backrefs_text_bytes = 0;
thread_local const void* g_backrefs_host = nullptr;  // the same template, host copy (sizing: how much a match can grow)
}

namespace {

constexpr size_t kLdsBudget = 64 * 1024;

struct Launch {  // kernel-visible part of the launch plan (POD)
  const int32_t* image;
  int image_words;
  int slots;        // 32-bit scratch slots per thread
  uint32_t* arena;  // global scratch (nullptr = lists in LDS)
  int image_in_lds;
};
struct Plan {
  Launch d;
  int threads;       // workgroup size
  size_t lds_bytes;  // dynamic LDS
  unsigned grid;
  bool small;
  Buf arena_buf;
};

struct RowSrc {
  ColView in;
  const uint8_t* flags;
  int64_t safe_end;  // chars bytes that may be read (logical size + allocation slack)
};

// common prologue: stage the image, carve per-thread scratch
struct Ctx {
  csvm::ProgView P;
  uint32_t* mem;
  int stride;
};
__device__ __forceinline__ Ctx setup(const Launch& L, const uint8_t* flags, uint32_t* smem) {
  Ctx c;
  const int32_t* img = L.image;
  int used = 0;
  if (L.image_in_lds) {
    for (int i = threadIdx.x; i < L.image_words; i += blockDim.x) smem[i] = (uint32_t)L.image[i];
    __syncthreads();
    img = (const int32_t*)smem;
    used = (L.image_words + 3) & ~3;
  }
  c.P = csvm::make_view(img, flags);
  if (L.arena) {
    c.mem = L.arena + (size_t)blockIdx.x * blockDim.x * L.slots + threadIdx.x;
  } else {
    c.mem = smem + used + threadIdx.x;
  }
  c.stride = blockDim.x;
  return c;
}

// MODE 0 contains_re, 1 match, 2 count_re
template <bool SMALL, int MODE>
__global__ void k_regex_scan(RowSrc src, Launch L, uint8_t* __restrict__ out8, int32_t* __restrict__ out32,
                             unsigned long long* __restrict__ found) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Ctx c = setup(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  int hits = 0;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    int v = 0;
    if (row_is_valid(in.validity, r)) {
      int64_t b = in.offsets[r];
      csvm::Vm<SMALL> vm(c.P, c.mem, c.stride, in.chars + b, (int)(in.offsets[r + 1] - b));
      if (MODE == 2) v = csvm::row_count_re(vm);
      else v = csvm::row_contains_re(vm, MODE == 1);
    }
    if (MODE == 2) out32[r] = v;
    else out8[r] = (uint8_t)v;
    hits += v > 0;
  }
  long long t = block_reduce_sum(hits);
  if (threadIdx.x == 0 && t) atomicAdd(found, (unsigned long long)t);
}

// replace_re size pass: output bytes per row (-1 for null rows) + block sums
template <bool SMALL>
__global__ void k_replace_re_size(RowSrc src, Launch L, int rb, int maxrepl, int32_t* __restrict__ lens) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Ctx c = setup(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    int len = -1;
    if (row_is_valid(in.validity, r)) {
      int64_t b = in.offsets[r];
      int n = (int)(in.offsets[r + 1] - b);
      csvm::Vm<SMALL> vm(c.P, c.mem, c.stride, in.chars + b, n);
      len = n;
      csvm::row_replace_matches(vm, maxrepl, [&](int mb, int me, int reps) { len += reps * rb - (me - mb); });
    }
    lens[r] = len;
  }
}
template <bool SMALL>
__global__ void k_replace_re_write(RowSrc src, Launch L, const uint8_t* __restrict__ repl, int rb,
                                   int maxrepl, const int64_t* __restrict__ out_off,
                                   uint8_t* __restrict__ out_chars) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Ctx c = setup(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows || !row_is_valid(in.validity, r)) continue;
    int64_t b = in.offsets[r];
    int n = (int)(in.offsets[r + 1] - b);
    const uint8_t* p = in.chars + b;
    uint8_t* o = out_chars + out_off[r];
    int copied = 0;
    csvm::Vm<SMALL> vm(c.P, c.mem, c.stride, p, n);
    csvm::row_replace_matches(vm, maxrepl, [&](int mb, int me, int reps) {
      for (int i = copied; i < mb; ++i) *o++ = p[i];
      for (int k = 0; k < reps; ++k)
        for (int i = 0; i < rb; ++i) *o++ = repl[i];
      copied = me;
    });
    for (int i = copied; i < n; ++i) *o++ = p[i];
  }
}

// ---- extract (extract.cu:36-151): capture-group spans ------------------------------------
// One thread per row: find() the leftmost match (list simulator), then one anchored run per
// capture group with that group's ranges tracked (csvm::GroupVm).  The spans go to
// begins / lens [group * rows + row] (len -1 = null row); a second kernel copies the bytes.
// The thread lists live in a global arena (12 slots per instruction per thread), interleaved
// across the lanes of a workgroup so the list walks coalesce.
constexpr int kMaxGroups = 32;
template <bool SMALL>
__global__ void __launch_bounds__(256) k_extract_spans(RowSrc src, Launch L, int groups, int32_t* __restrict__ begins,
                                                       int32_t* __restrict__ lens) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Ctx c = setup(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    int mb = 0, me = 0;
    bool hit = false;
    const int64_t b = in.offsets[r];
    const int n = (int)(in.offsets[r + 1] - b);
    if (row_is_valid(in.validity, r)) {
      csvm::Vm<SMALL> vm(c.P, c.mem, c.stride, in.chars + b, n);
      hit = vm.find(0, n, mb, me) > 0;
    }
    for (int g = 0; g < groups; ++g) {
      int x = 0, y = -1;
      if (hit) {
        csvm::GroupVm<SMALL> gv(c.P, c.mem, c.stride, in.chars + b, n);
        if (!csvm::row_group_span(gv, mb, g + 1, x, y)) y = -1;
      }
      begins[(int64_t)g * in.rows + r] = x;
      lens[(int64_t)g * in.rows + r] = y < 0 ? -1 : y - x;
    }
  }
}
struct ExtractOut {
  const int64_t* off[kMaxGroups];
  uint8_t* chars[kMaxGroups];
};
// thread per row: every group's span is copied by the row's thread (spans are short)
__global__ void __launch_bounds__(256) k_extract_write(ColView in, int groups, const int32_t* __restrict__ begins,
                                                       const int32_t* __restrict__ lens, ExtractOut out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= in.rows) return;
  const uint8_t* p = in.chars + in.offsets[r];
  for (int g = 0; g < groups; ++g) {
    const int len = lens[(int64_t)g * in.rows + r];
    if (len <= 0) continue;
    const uint8_t* q = p + begins[(int64_t)g * in.rows + r];
    uint8_t* o = out.chars[g] + out.off[g][r];
    for (int i = 0; i < len; ++i) o[i] = q[i];
  }
}

// The same on 64-row sub-tiles, a wave each (columns whose sub-tiles fit the staging buffer): the sub-tile's bytes
// arrive in LDS with whole 16-byte loads; per output column the row lanes put their span into the column's region
// with exact-size stores at any alignment (16 / 8 / 4 / 2 / 1 bytes from unaligned 16-byte reads of the staged row)
// and the wave flushes the region -- the 64 rows' spans are consecutive in the column's chars -- with aligned
// 16-byte stores.  The thread-per-row kernel above reads every row byte by byte from 64 different cache lines.
struct SpanWriteArgs {
  ColView in;
  int ncols;
  const int32_t* begins;
  const int32_t* lens;
  ExtractOut out;
  long long nsub;
  int cap;
};
__global__ void __launch_bounds__(256) k_spans_write_tile(SpanWriteArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (2 * a.cap + 64);
  uint8_t* region = lds_in + a.cap + 32;
  const ColView& in = a.in;
  for (long long tile = (long long)blockIdx.x * 4 + wv; tile < a.nsub; tile += (long long)gridDim.x * 4) {
    const long long r0 = tile * 64;
    const int nrows = (int)min(64ll, in.rows - r0);
    const cstile::TileOffs t = cstile::load_tile_offsets_r(in.offsets, in.rows, tile, 64, lane);
    const long long g0 = cstile::rl64(t.o0, 0), g1 = cstile::rl64(t.o1, 63);
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const int want = (int)(g1 - g0) + lead;
    for (int i = lane * 16; i < want; i += 64 * 16) *reinterpret_cast<uint4*>(lds_in + i) = *reinterpret_cast<const uint4*>(in.chars + (g0 - lead) + i);
    const int rpos = lead + (int)(t.o0 - g0);
    cstile::wave_lds_fence();
    for (int g = 0; g < a.ncols; ++g) {
      const long long at = (long long)g * in.rows + r0 + lane;
      int len = lane < nrows ? a.lens[at] : 0;
      const int beg = len > 0 ? a.begins[at] : 0;
      if (len < 0) len = 0;
      const long long d0 = a.out.off[g][r0 + min(lane, nrows)];
      const long long base = cstile::rl64(d0, 0);
      const int total = (int)(cstile::rl64(d0, 63) - base) + __builtin_amdgcn_readlane(len, 63);
      if (total == 0) continue;
      const int di = (int)(d0 - base);
      for (int k = 0; k < len; k += 16)
        cstile::lds_put16(region + di + k, *reinterpret_cast<const cstile::lds_u32x4u*>(lds_in + rpos + beg + k), len - k);
      cstile::wave_lds_fence();
      cstile::wave_flush_shift(a.out.chars[g] + base, total, region, lane);
      cstile::wave_lds_fence();  // the next column re-uses the region
    }
  }
}

// The same from PACKED spans (begin << 16 | length, 0xFFFFFFFF = null; written by the scan stream kernels together with
// every tile's bytes per column), sizing the columns on the way: `tile_base[c][t]` = bytes of column c before tile t
// (the exclusive scan of the scan kernel's tile totals), so a tile's wave turns its rows' lengths into offsets itself --
// one wave scan per column -- and writes offsets, validity bits and chars.  The lengths -> offsets passes over every
// column (k_chunk_sums / k_chunk_offsets: 0.35 ms per column of 100M rows) and the write kernel's reads of the offsets
// they produced are gone, and the span arrays are half as large.
struct SpanOut2 {
  void* off[kMaxGroups];  // int32 offsets when every column stays below 2 GiB (half the offset bytes: as the split kernels write them), else int64
  uint8_t* chars[kMaxGroups];
  uint8_t* valid[kMaxGroups];
};
struct SpanWrite2Args {
  ColView in;
  int ncols;
  const uint32_t* spans;     // [column][row]
  const int64_t* tile_base;  // [column][nsub + 1]
  SpanOut2 out;
  long long nsub;
  int cap;
  int off32;
};
__global__ void __launch_bounds__(256) k_spans_write_tile2(SpanWrite2Args a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  uint8_t* lds_in = reinterpret_cast<uint8_t*>(smem) + (size_t)wv * (2 * a.cap + 64);
  uint8_t* region = lds_in + a.cap + 32;
  const ColView& in = a.in;
  // a wave walks a run of consecutive tiles with the next tile's chars in flight (register prefetch, tile_utils.h)
  const long long waves = (long long)gridDim.x * 4;
  const long long per = (a.nsub + waves - 1) / waves;
  long long tile = ((long long)blockIdx.x * 4 + wv) * per;
  const long long tile_end = min(a.nsub, tile + per);
  if (tile >= tile_end) return;
  cstile::TileOffs cur = cstile::load_tile_offsets_r(in.offsets, in.rows, tile, 64, lane);
  cstile::TileOffs nxt = cur;
  if (tile + 1 < tile_end) nxt = cstile::load_tile_offsets_r(in.offsets, in.rows, tile + 1, 64, lane);
  cstile::TileChars pf;
#pragma unroll
  for (int j = 0; j < cstile::kPfChunks; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
  for (;;) {
    const long long r0 = tile * 64;
    const int nrows = (int)min(64ll, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const int want = (int)(g1 - g0) + lead;
    const int rpos = lead + (int)(cur.o0 - g0);
    cstile::stage_chars(lds_in, want, lane, pf);
    // this tile's spans and column positions, the next tile's chars, the offsets of the one after (handed over at the bottom)
    uint32_t sp[4];
    long long cb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      sp[g] = g < a.ncols && lane < nrows ? a.spans[(long long)g * in.rows + r0 + lane] : 0xFFFFFFFFu;
      cb[g] = g < a.ncols ? a.tile_base[(long long)g * (a.nsub + 1) + tile] : 0;
    }
    const bool has_next = tile + 1 < tile_end;
    const cstile::TileOffs nn = cstile::load_tile_offsets_r(in.offsets, in.rows, tile + 2 < tile_end ? tile + 2 : tile_end - 1, 64, lane);
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
    }
    cstile::wave_lds_fence();
    for (int g = 0; g < a.ncols; ++g) {
      uint32_t w;
      long long base;
      if (g < 4) {  // (requested above)
        w = sp[0];
        base = cb[0];
#pragma unroll
        for (int c = 1; c < 4; ++c)
          if (c == g) {
            w = sp[c];
            base = cb[c];
          }
      } else {
        w = lane < nrows ? a.spans[(long long)g * in.rows + r0 + lane] : 0xFFFFFFFFu;
        base = a.tile_base[(long long)g * (a.nsub + 1) + tile];
      }
      const bool valid = w != 0xFFFFFFFFu;
      const int len = valid ? (int)(w & 0xFFFFu) : 0;
      const int beg = valid ? (int)(w >> 16) : 0;
      const int inc = csdev::wave_inclusive_scan(len);
      const int di = inc - len;
      const int total = __builtin_amdgcn_readlane(inc, 63);
      if (a.off32) {
        int32_t* o32 = static_cast<int32_t*>(a.out.off[g]);
        if (lane < nrows) o32[r0 + lane] = (int32_t)(base + di);
        if (tile + 1 == a.nsub && lane == nrows - 1) o32[in.rows] = (int32_t)(base + total);
      } else {
        int64_t* o64 = static_cast<int64_t*>(a.out.off[g]);
        if (lane < nrows) o64[r0 + lane] = base + di;
        if (tile + 1 == a.nsub && lane == nrows - 1) o64[in.rows] = base + total;
      }
      const unsigned long long vbits = __ballot(valid);
      if (lane == 0) *reinterpret_cast<unsigned long long*>(a.out.valid[g] + tile * 8) = vbits;
      if (total == 0) continue;
      for (int k = 0; k < len; k += 16)
        cstile::lds_put16(region + di + k, *reinterpret_cast<const cstile::lds_u32x4u*>(lds_in + rpos + beg + k), len - k);
      cstile::wave_lds_fence();
      cstile::wave_flush_shift(a.out.chars[g] + base, total, region, lane);
      cstile::wave_lds_fence();  // the next column re-uses the region
    }
    cstile::wave_lds_fence();  // the next tile overwrites lds_in
    if (!has_next) break;
    ++tile;
    nxt = nn;
  }
}

// ---- tagged-DFA kernels (regex_tdfa.h): tables staged in LDS, no per-thread lists ----
struct TLaunch {
  const int32_t* tdfa;   // device image
  int tdfa_words;
  int in_lds;
  const int32_t* image;  // list-simulator image (global; consulted for non-ASCII chars only)
};
struct TCtx {
  cstd::View D;
  csvm::ProgView P;
};
template <bool IN_LDS>
__device__ __forceinline__ TCtx tsetup(const TLaunch& L, const uint8_t* flags, uint32_t* smem) {
  TCtx c;
  if (IN_LDS) {
    for (int i = threadIdx.x; i < L.tdfa_words; i += blockDim.x) smem[i] = (uint32_t)L.tdfa[i];
    __syncthreads();
    c.D = cstd::make_view((const int32_t*)smem);
  } else if (L.in_lds == 2) {
    // the tables stay in memory, the header and the image's last words (suffix, repetition counts, group map) come into LDS:
    // what the hot loops read of the image -- as scalar loads from memory each read also waited for every LDS access in flight
    // (replace_re of the dotted quad on this form: 5.1 against 4.7 ms)
    if (threadIdx.x < cstd::kHeadTailWords)
      smem[threadIdx.x] = threadIdx.x == 15 ? (uint32_t)cstd::kHeadTailWords
                                            : (uint32_t)L.tdfa[threadIdx.x < 32 ? (int)threadIdx.x : L.tdfa_words - cstd::kHeadTailWords + (int)threadIdx.x];
    __syncthreads();
    c.D = cstd::make_view((const int32_t*)smem, L.tdfa);
  } else {
    c.D = cstd::make_view(L.tdfa);
  }
  c.P = csvm::make_view(L.image, flags);
  return c;
}
template <int MODE, bool IN_LDS, bool WIDE = false>
__global__ void __launch_bounds__(256) k_tdfa_scan(RowSrc src, TLaunch L, uint8_t* __restrict__ out8,
                                                   int32_t* __restrict__ out32,
                                                   unsigned long long* __restrict__ found) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  TCtx c = tsetup<IN_LDS>(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  int hits = 0;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    int v = 0;
    if (row_is_valid(in.validity, r)) {
      int64_t b = in.offsets[r];
      typename std::conditional<WIDE, cstd::TdfaWide, cstd::Tdfa>::type vm(c.D, c.P, in.chars + b, (int)(in.offsets[r + 1] - b));
      vm.wide_ok = (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
      if (MODE == 2) v = csvm::row_count_re(vm);
      else v = csvm::row_contains_re(vm, MODE == 1);
    }
    if (MODE == 2) out32[r] = v;
    else out8[r] = (uint8_t)v;
    hits += v > 0;
  }
  long long t = block_reduce_sum(hits);
  if (threadIdx.x == 0 && t) atomicAdd(found, (unsigned long long)t);
}
// A row of the list kernels in LDS: the aligned 16-byte pieces that cover it (rows within the masks: seven at most), copied by
// the row's own thread -- the generic executor then walks LDS bytes instead of a chain of dependent loads from memory
// (k_tdfa_replace_list on the C5 pieces: 0.38 ms a launch from memory).  Returns the row's first byte; a longer row stays in memory.
constexpr int kListRowBytes = 112;
// (the pieces must lie inside the chars buffer -- [lo, hi): a caller's memory wrapped at an odd address has no slack around it)
__device__ __forceinline__ const uint8_t* list_row_to_lds(const uint8_t* src, int n, uint8_t* buf, const uint8_t* lo, const uint8_t* hi) {
  const int sh = (int)((uintptr_t)src & 15);
  if (sh + n > kListRowBytes || src - sh < lo || src - sh + ((sh + n + 15) & ~15) > hi) return src;
  const uint4* a = reinterpret_cast<const uint4*>(src - sh);
  for (int k = 0; k * 16 < sh + n; ++k) *reinterpret_cast<uint4*>(buf + 16 * k) = a[k];
  return buf + sh;
}
// the rows a stream launch put off (ScanStreamArgs::deferred): a thread a row, the generic executor on the row's bytes (staged in LDS)
template <int MODE>
__global__ void __launch_bounds__(256) k_tdfa_scan_list(RowSrc src, TLaunch L, const int32_t* __restrict__ list, const unsigned* __restrict__ nlist, unsigned cap,
                                                        uint8_t* __restrict__ out8, int32_t* __restrict__ out32, unsigned long long* __restrict__ found) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const unsigned total = min(*nlist, cap);
  if ((unsigned)blockIdx.x * 256u >= total) return;  // (nothing for this workgroup: not even the tables)
  TCtx c = tsetup<true>(L, src.flags, smem);
  const ColView& in = src.in;
  int hits = 0;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int64_t r = list[i];
    const int64_t b = in.offsets[r];
    const int n = (int)(in.offsets[r + 1] - b);
    uint8_t* buf = reinterpret_cast<uint8_t*>(smem) + (((size_t)L.tdfa_words * 4 + 15) & ~(size_t)15) + (size_t)threadIdx.x * kListRowBytes;
    const uint8_t* p = list_row_to_lds(in.chars + b, n, buf, in.chars, in.chars + src.safe_end);
    cstd::Tdfa vm(c.D, c.P, p, n);
    vm.wide_ok = p != in.chars + b || (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
    const int v = MODE == 2 ? csvm::row_count_re(vm) : csvm::row_contains_re(vm, false);
    if (MODE == 2) out32[r] = v;
    else out8[r] = (uint8_t)v;
    hits += v > 0;
  }
  const long long t = block_reduce_sum(hits);
  if (threadIdx.x == 0 && t) atomicAdd(found, (unsigned long long)t);
}
template <bool IN_LDS, bool WIDE = false>
__global__ void __launch_bounds__(256) k_tdfa_replace_size(RowSrc src, TLaunch L, int rb, int maxrepl,
                                                           int32_t* __restrict__ lens) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  TCtx c = tsetup<IN_LDS>(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    int len = -1;
    if (row_is_valid(in.validity, r)) {
      int64_t b = in.offsets[r];
      int n = (int)(in.offsets[r + 1] - b);
      typename std::conditional<WIDE, cstd::TdfaWide, cstd::Tdfa>::type vm(c.D, c.P, in.chars + b, n);
      vm.wide_ok = (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
      len = n;
      csvm::row_replace_matches(vm, maxrepl, [&](int mb, int me, int reps) { len += reps * rb - (me - mb); });
    }
    lens[r] = len;
  }
}
template <bool IN_LDS, bool WIDE = false>
__global__ void __launch_bounds__(256) k_tdfa_replace_write(RowSrc src, TLaunch L, const uint8_t* __restrict__ repl,
                                                            int rb, int maxrepl, const int64_t* __restrict__ out_off,
                                                            uint8_t* __restrict__ out_chars) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  TCtx c = tsetup<IN_LDS>(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows || !row_is_valid(in.validity, r)) continue;
    int64_t b = in.offsets[r];
    int n = (int)(in.offsets[r + 1] - b);
    const uint8_t* p = in.chars + b;
    uint8_t* o = out_chars + out_off[r];
    int copied = 0;
    typename std::conditional<WIDE, cstd::TdfaWide, cstd::Tdfa>::type vm(c.D, c.P, p, n);
    vm.wide_ok = (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
    csvm::row_replace_matches(vm, maxrepl, [&](int mb, int me, int reps) {
      for (int i = copied; i < mb; ++i) *o++ = p[i];
      for (int k = 0; k < reps; ++k)
        for (int i = 0; i < rb; ++i) *o++ = repl[i];
      copied = me;
    });
    for (int i = copied; i < n; ++i) *o++ = p[i];
  }
}

// replace_re of the rows in `list` (StreamArgs::hole_*): their output sizes (lens[i] for list[i]), or -- WRITE -- their bytes at
// the offsets the stream launch left them
template <bool WRITE>
__global__ void __launch_bounds__(256) k_tdfa_replace_list(RowSrc src, TLaunch L, const int32_t* __restrict__ list, int64_t count, const uint8_t* __restrict__ repl, int rb,
                                                           int32_t* __restrict__ lens, const int64_t* __restrict__ out_off, uint8_t* __restrict__ out_chars) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  TCtx c = tsetup<true>(L, src.flags, smem);
  const ColView& in = src.in;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const int64_t r = list[i];
    if (!row_is_valid(in.validity, r)) continue;  // (a null row over bytes of the chars: no hole was left for it)
    const int64_t b = in.offsets[r];
    const int n = (int)(in.offsets[r + 1] - b);
    uint8_t* buf = reinterpret_cast<uint8_t*>(smem) + (((size_t)L.tdfa_words * 4 + 15) & ~(size_t)15) + (size_t)threadIdx.x * kListRowBytes;
    const uint8_t* p = list_row_to_lds(in.chars + b, n, buf, in.chars, in.chars + src.safe_end);
    cstd::Tdfa vm(c.D, c.P, p, n);
    vm.wide_ok = p != in.chars + b || (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
    if (!WRITE) {
      int len = n;
      csvm::row_replace_matches(vm, -1, [&](int mb, int me, int reps) { len += reps * rb - (me - mb); });
      lens[i] = len;
    } else {
      uint8_t* o = out_chars + out_off[r];
      int copied = 0;
      csvm::row_replace_matches(vm, -1, [&](int mb, int me, int reps) {
        for (int k = copied; k < mb; ++k) *o++ = p[k];
        for (int k = 0; k < reps; ++k)
          for (int q = 0; q < rb; ++q) *o++ = repl[q];
        copied = me;
      });
      for (int k = copied; k < n; ++k) *o++ = p[k];
    }
  }
}

// extract with the leftmost match found by the tagged DFA (table in LDS); only rows that hold a
// match run the list simulation, and only from the match start
template <bool IN_LDS, bool SMALL>
__global__ void __launch_bounds__(256) k_extract_spans_tdfa(RowSrc src, TLaunch TL, Launch L, int groups, int use_fast,
                                                            int32_t* __restrict__ begins, int32_t* __restrict__ lens) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  TCtx c = tsetup<IN_LDS>(TL, src.flags, smem);
  uint32_t* mem = L.arena + (size_t)blockIdx.x * blockDim.x * L.slots + threadIdx.x;
  // the head of every thread's lists and closure stack in LDS, behind the DFA table
  uint32_t* region = smem + (IN_LDS ? ((TL.tdfa_words + 3) & ~3) : 0);
  uint32_t* fast = use_fast ? region + threadIdx.x : nullptr;
  if (L.image_in_lds) {  // the program image behind the fast region: the list simulation fetches an instruction per step
    uint32_t* img = region + (use_fast ? 256 * csvm::GroupVm<true>::kFastSlots : 0);
    for (int i = threadIdx.x; i < L.image_words; i += blockDim.x) img[i] = (uint32_t)L.image[i];
    __syncthreads();
    c.P = csvm::make_view((const int32_t*)img, src.flags);
  }
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    int mb = 0, me = 0;
    bool hit = false;
    const int64_t b = in.offsets[r];
    const int n = (int)(in.offsets[r + 1] - b);
    if (row_is_valid(in.validity, r)) {
      cstd::Tdfa vm(c.D, c.P, in.chars + b, n);
      vm.wide_ok = (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
      hit = vm.find(0, n, mb, me) > 0;
    }
    for (int g = 0; g < groups; ++g) {
      int x = 0, y = -1;
      if (hit) {
        csvm::GroupVm<SMALL> gv(c.P, mem, blockDim.x, in.chars + b, n, fast, blockDim.x);
        if (!csvm::row_group_span(gv, mb, g + 1, x, y)) y = -1;
      }
      begins[(int64_t)g * in.rows + r] = x;
      lens[(int64_t)g * in.rows + r] = y < 0 ? -1 : y - x;
    }
  }
}

// extract with the group ranges carried by the tagged DFA (Tdfa::group_find): the leftmost match, then
// one anchored DFA run per capture group -- no thread lists, no scratch memory
template <bool IN_LDS>
__global__ void __launch_bounds__(256) k_extract_spans_dfa(RowSrc src, TLaunch TL, const int32_t* __restrict__ gtags, int groups,
                                                           int32_t* __restrict__ begins, int32_t* __restrict__ lens) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  TCtx c = tsetup<IN_LDS>(TL, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    int mb = 0, me = 0;
    bool hit = false;
    const int64_t b = in.offsets[r];
    const int n = (int)(in.offsets[r + 1] - b);
    cstd::Tdfa vm(c.D, c.P, in.chars + b, n);
    vm.wide_ok = (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
    if (row_is_valid(in.validity, r)) hit = vm.find(0, n, mb, me) > 0;
    for (int g = 0; g < groups; ++g) {
      int x = -1, y = -1;
      const bool ok = hit && vm.group_find(mb, gtags, g + 1, x, y) && x >= 0 && y > x;
      begins[(int64_t)g * in.rows + r] = ok ? x : 0;
      lens[(int64_t)g * in.rows + r] = ok ? y - x : -1;
    }
  }
}

// ---- replace_with_backrefs (replace_backref.cu:36-207) ---------------------------------------
// Two passes, one thread per row (csvm::row_backrefs): sizes, then the bytes.  DFA: matches and group
// ranges by the tagged DFA (Tdfa::find / group_find); otherwise the list simulators over a global arena.
template <bool DFA, bool IN_LDS, bool SMALL, class Out>
__device__ __forceinline__ void backrefs_for_row(const TCtx& c, const csvm::ProgView& P, uint32_t* mem, int stride, const RowSrc& src,
                                                 const int32_t* gtags, const csvm::BackrefTemplate& t, int64_t r, Out&& out) {
  const ColView& in = src.in;
  const int64_t b = in.offsets[r];
  const int n = (int)(in.offsets[r + 1] - b);
  const uint8_t* p = in.chars + b;
  if constexpr (DFA) {
    cstd::Tdfa vm(c.D, c.P, p, n);
    vm.wide_ok = (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
    // the matches by the flat scan loop of replace_re (word-wise idle skipping), not by one find() per match
    csvm::row_backrefs(
        p, n, t, [&](auto&& f) { csvm::walk_matches(vm, f); },
        [&](int mb, int g, int& x, int& y) { return g == 0 ? vm.find(mb, mb + 1, x, y) > 0 : vm.group_find(mb, gtags, g, x, y) > 0; }, out);
  } else {
    csvm::row_backrefs(
        p, n, t,
        [&](auto&& f) {
          csvm::walk_matches_by_find(
              [&](int from, int& mb, int& me) {
                csvm::Vm<SMALL> vm(P, mem, stride, p, n);
                return vm.find(from, n, mb, me) > 0;
              },
              f);
        },
        [&](int mb, int g, int& x, int& y) {
          if (g == 0) {
            csvm::Vm<SMALL> vm(P, mem, stride, p, n);
            return vm.find(mb, mb + 1, x, y) > 0;
          }
          csvm::GroupVm<SMALL> gv(P, mem, stride, p, n);
          return gv.run(mb, g, x, y) > 0;
        },
        out);
  }
}
struct BackrefArgs {
  RowSrc src;
  TLaunch TL;
  Launch L;
  const int32_t* gtags;
  csvm::BackrefTemplate t;
};
template <bool DFA, bool IN_LDS, bool SMALL, bool WRITE>
__global__ void __launch_bounds__(256) k_backrefs(BackrefArgs a, int32_t* __restrict__ lens, const int64_t* __restrict__ out_off,
                                                  uint8_t* __restrict__ out_chars) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  TCtx c{};
  csvm::ProgView P{};
  uint32_t* mem = nullptr;
  if (DFA) {
    c = tsetup<IN_LDS>(a.TL, a.src.flags, smem);
  } else {
    P = csvm::make_view(a.L.image, a.src.flags);
    mem = a.L.arena + (size_t)blockIdx.x * blockDim.x * a.L.slots + threadIdx.x;
  }
  const ColView& in = a.src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    const bool valid = row_is_valid(in.validity, r);
    if (!WRITE) {
      int len = -1;
      if (valid) {
        len = 0;
        backrefs_for_row<DFA, IN_LDS, SMALL>(c, P, mem, blockDim.x, a.src, a.gtags, a.t, r, [&](const uint8_t*, int k) { len += k; });
      }
      lens[r] = len;
    } else if (valid) {
      uint8_t* o = out_chars + out_off[r];
      backrefs_for_row<DFA, IN_LDS, SMALL>(c, P, mem, blockDim.x, a.src, a.gtags, a.t, r, [&](const uint8_t* q, int k) {
        for (int i = 0; i < k; ++i) *o++ = q[i];
      });
    }
  }
}

// ---- replace_re with several patterns (replace_multi.cu:39-189) ----------------------------
// At every character position the patterns are tried in order, each anchored at that position
// (regexec.inl: start window [pos, pos + 1)); the first that matches is replaced and the walk
// continues behind the match.  Thread per row, size pass + write pass.  `first` holds, per
// pattern, the ASCII bytes an anchored match can begin with (a superset): most (position,
// pattern) pairs are turned away by one bit test.
struct MultiProg {
  const int32_t* tdfa;   // tagged DFA image (DFA variant)
  const int32_t* image;  // list-simulator image
  uint32_t first[4];     // ASCII bytes that can start a match; all ones when unknown
};
struct MultiArgs {
  RowSrc src;
  const MultiProg* progs;
  int nprogs;
  ColView repls;
  uint32_t* arena;  // list simulator scratch (VM variant)
  int slots;
};
template <bool DFA, bool WRITE>
__global__ void __launch_bounds__(256) k_multi_replace(MultiArgs a, int32_t* __restrict__ lens, const int64_t* __restrict__ out_off,
                                                       uint8_t* __restrict__ out_chars, unsigned* __restrict__ bad) {
  const ColView& in = a.src.in;
  uint32_t* mem = DFA ? nullptr : a.arena + (size_t)blockIdx.x * blockDim.x * a.slots + threadIdx.x;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    if (!row_is_valid(in.validity, r)) {
      if (!WRITE) lens[r] = -1;
      continue;
    }
    const int64_t b = in.offsets[r];
    const int n = (int)(in.offsets[r + 1] - b);
    const uint8_t* p = in.chars + b;
    uint8_t* o = WRITE ? out_chars + out_off[r] : nullptr;
    int total = n, copied = 0, pos = 0;
    while (pos < n) {
      const uint8_t c0 = p[pos];
      int mb = 0, me = 0;
      bool hit = false;
      int t = 0;
      for (; t < a.nprogs && !hit; ++t) {
        const MultiProg& mp = a.progs[t];
        if (c0 < 128 && !((mp.first[c0 >> 5] >> (c0 & 31)) & 1u)) continue;
        if (DFA) {
          const cstd::View D = cstd::make_view(mp.tdfa);
          const csvm::ProgView P = csvm::make_view(mp.image, a.src.flags);
          cstd::Tdfa vm(D, P, p, n);
          vm.wide_ok = false;
          hit = vm.find(pos, pos + 1, mb, me) > 0;
        } else {
          const csvm::ProgView P = csvm::make_view(mp.image, a.src.flags);
          csvm::Vm<false> vm(P, mem, blockDim.x, p, n);
          hit = vm.find(pos, pos + 1, mb, me) > 0;
        }
      }
      if (!hit) {
        unsigned w = csrow::lead_width(c0);
        pos += w ? (int)w : 1;
        continue;
      }
      --t;  // the pattern that matched
      if (me <= mb) {  // an empty match: the reference never leaves this position
        atomicOr(bad, 1u);
        break;
      }
      const int64_t rr = a.repls.rows == 1 ? 0 : t;
      const bool has = row_is_valid(a.repls.validity, rr);
      const int rn = has ? (int)(a.repls.offsets[rr + 1] - a.repls.offsets[rr]) : 0;
      total += rn - (me - mb);
      if (WRITE) {
        copy_bytes(o, p + copied, mb - copied);
        o += mb - copied;
        if (rn) copy_bytes(o, a.repls.chars + a.repls.offsets[rr], rn);
        o += rn;
        copied = me;
      }
      pos = me;
    }
    if (WRITE) copy_bytes(o, p + copied, n - copied);
    else lens[r] = total;
  }
}

// ---- findall (findall.cu:39-179): column k = every row's k-th match ------------------------
// The per-row match counts come from the count_re kernels; this pass walks the matches again
// (tagged DFA, or the list simulator) and leaves the first `ncols` spans in begins / lens
// [k * rows + row]  (len -1: the row has no k-th match).  k_extract_write copies the bytes.
__global__ void __launch_bounds__(256) k_max_i32(const int32_t* __restrict__ v, int64_t n, int* __restrict__ out) {
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = max(m, v[i]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}
template <class VM>
__device__ __forceinline__ void findall_row(VM& vm, int64_t r, int64_t rows, int ncols, int32_t* __restrict__ begins,
                                            int32_t* __restrict__ lens) {
  const int k = csvm::row_findall(vm, [&](int j, int mb, int me) {
    if (j < ncols) {
      begins[(int64_t)j * rows + r] = mb;
      lens[(int64_t)j * rows + r] = me - mb;
    }
    return j + 1 < ncols;
  });
  for (int j = k; j < ncols; ++j) lens[(int64_t)j * rows + r] = -1;
}
template <bool IN_LDS>
__global__ void __launch_bounds__(256) k_findall_spans_tdfa(RowSrc src, TLaunch L, int ncols, int32_t* __restrict__ begins,
                                                            int32_t* __restrict__ lens) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  TCtx c = tsetup<IN_LDS>(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    if (!row_is_valid(in.validity, r)) {
      for (int j = 0; j < ncols; ++j) lens[(int64_t)j * in.rows + r] = -1;
      continue;
    }
    const int64_t b = in.offsets[r];
    cstd::Tdfa vm(c.D, c.P, in.chars + b, (int)(in.offsets[r + 1] - b));
    vm.wide_ok = (b & ~(int64_t)3) + cstd::Tdfa::kMaskBytes <= src.safe_end;
    findall_row(vm, r, in.rows, ncols, begins, lens);
  }
}
template <bool SMALL>
__global__ void k_findall_spans(RowSrc src, Launch L, int ncols, int32_t* __restrict__ begins, int32_t* __restrict__ lens) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Ctx c = setup(L, src.flags, smem);
  const ColView& in = src.in;
  const int64_t nblk = (in.rows + blockDim.x - 1) / blockDim.x;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t r = blk * blockDim.x + threadIdx.x;
    if (r >= in.rows) continue;
    if (!row_is_valid(in.validity, r)) {
      for (int j = 0; j < ncols; ++j) lens[(int64_t)j * in.rows + r] = -1;
      continue;
    }
    const int64_t b = in.offsets[r];
    csvm::Vm<SMALL> vm(c.P, c.mem, c.stride, in.chars + b, (int)(in.offsets[r + 1] - b));
    findall_row(vm, r, in.rows, ncols, begins, lens);
  }
}

// ---- single-pass replace_re over row tiles (tile_utils.h) -----------------------------
// Used when the output is known not to outgrow the input (replacement no longer
// than the shortest possible match): the output buffer is allocated at the input
// size, every input byte is read once, the automaton runs once per row, and
// offsets come from a look-back scan inside the same kernel.
// CS_TILE_DEBUG (measurement switches of the stream kernels: phases skipped, other tile sequences, the decoupled look-back
// instead of the scanner team) exists in profiling builds only (make prof): in the product kernels the switches are
// compile-time zeros -- their cold code moved the hot loops out of the instruction cache (replace_re 5.4 -> 4.9 ms).
#if defined(CS_PHASE_PROF)
#define CS_DBG(a) ((a).debug)
#else
#define CS_DBG(a) 0
#endif
constexpr int kMaxRec = 4;  // matches per row kept in registers between the size and write phases

template <class VM, class Rec>
__device__ __forceinline__ int tile_row_size(VM& vm, int n, int rb, int maxrepl, Rec&& rec) {
  int len = n;
  csvm::row_replace_matches(vm, maxrepl, [&](int mb, int me, int reps) {
    len += reps * rb - (me - mb);
    rec(mb, me, reps);
  });
  return len;
}


// ---- persistent single-pass replace_re (tile_utils.h: sub-tile stream) ----------------
// (its non-persistent predecessor, k_tdfa_replace_tile -- one sub-tile a wave, plain look-back, reached only for replacements
// beyond 16 bytes that do not grow a row -- left the library in round 6: no test reached it, the two-pass kernels answer.)
// The grid is sized to
// the device's residency, wave `wid` of W walks sub-tiles wid, wid + W, ...; the chars of
// the next sub-tile are prefetched into registers while the current one is scanned, the
// look-back poll is issued before the output rows are assembled and consumed after, and
// the flush takes the destination alignment as it comes (no dependence of the assembly
// on the global position).
struct StreamArgs {
  ColView in;
  const uint8_t* flags;
  TLaunch L;
  const uint8_t* repl;
  int rb, maxrepl;
  int64_t* out_off;
  uint8_t* out_chars;
  cstile::u64* status;
  cstile::u64* excl;  // exclusive prefix per tile, written by the scanner wave (tile_utils.h: prefix_scanner)
  unsigned* error;
  long long nsub;
  int cap_in, cap_out, tbl_bytes;
  int debug;
  int outliers;  // the buffers are sized for all but a few sub-tiles: an oversize one is handled a thread per row in the kernel
  long long out_cap;  // bytes provisioned at out_chars (growing replacements)
  int rows_per_tile;  // LONG variants: 64, 32 or 16
  unsigned long long* tickets;  // 16 tile counters, 64 bytes apart, zeroed before the launch
  unsigned long long lit;       // UNITS: a literal needle (first byte lowest), litn bytes; litn == 0: none
  int litn;
  // BREFS (replace_with_backrefs on the unit scan): the template, the capture-group tag image and where both are
  // staged in LDS (behind the DFA table, inside tbl_bytes)
  const csvm::BackrefTemplate* tmpl;
  const int32_t* gtags;
  int gt_off, gt_words;
  // BITS: the bit-parallel form of the pattern (regex_bits.h), staged at byte offset bits_off of the LDS (inside tbl_bytes)
  const int32_t* bits;
  int bits_off, bits_words, bits_k;
  // HOLES (every form of 64-row tiles without a template; optional): the column's rows that hold a byte >= 0x80 or a NUL
  // (cs_virtual.hip: OddRows) are not scanned here.  Their output sizes were taken beforehand, a thread a row
  // (k_tdfa_replace_list<false>); a row lane of such a row leaves a hole of that size in the tile's output, which
  // k_tdfa_replace_list<true> fills after the launch -- and the other rows of its sub-tile keep the bit form / the chain
  // arithmetic / the unit scan, where one such row used to send all 64 to the row-by-row scan.
  const unsigned long long* hole_mask;  // [tile]
  const int64_t* hole_first;            // [tile]: index of the tile's first such row in hole_len
  const int32_t* hole_len;
};
#ifndef CS_STREAM_WAVES
#define CS_STREAM_WAVES 3
#endif
// REP16: replacement of 9..16 bytes (four registers) -- and, since round 3, of up to kMaxStreamRepl bytes, whose text the
// assembly reads from memory (a 21-byte replacement used to fall to the two-pass row kernels: 48.8 ms on the 100M-row column);
// the common short replacement keeps two registers.
constexpr int kMaxStreamRepl = 64;
// INPLACE (the output cannot outgrow the input): every row is compacted inside its own extent of
// the input tile while it is scanned -- the bytes before a match move down to the write cursor,
// the replacement follows -- so any number of matches per row costs no registers and the
// assembly is one contiguous copy per row.  Otherwise (growing replacement) up to kMaxRec matches
// per row are kept in registers and the rows are assembled piecewise; with RESCAN a row with more
// matches is scanned a second time during assembly (its size is known from the first scan),
// without it such a row fails the launch and the host repeats it with the RESCAN variant.
// LONG: rows of up to 255 bytes keep the lean scan (a sliding 96-byte window of candidate bits).
// the replacement text (rb <= 8 bytes in two registers) to an LDS position of any alignment: one to three stores
__device__ __forceinline__ void lds_put_short(uint8_t* at, uint32_t r0, uint32_t r1, int rb) {
  if (rb >= 8) {
    *reinterpret_cast<cstile::lds_u64u*>(at) = ((unsigned long long)r1 << 32) | r0;
    return;
  }
  if (rb & 4) {
    *reinterpret_cast<cstile::lds_u32u*>(at) = r0;
    at += 4;
    r0 = r1;
  }
  if (rb & 2) {
    *reinterpret_cast<cstile::lds_u16u*>(at) = (uint16_t)r0;
    at += 2;
    r0 >>= 16;
  }
  if (rb & 1) *at = (uint8_t)r0;
}

// ---- unit scan (regex_tdfa.cpp, header word 31): helpers shared by the stream kernels ------------------
constexpr int kUnitQueue = 128;  // units one round of the queue holds (a busier sub-tile scans its rows whole)
// Row lanes: the row's candidate bits (returned in m0..m2) and x bits give its units; all units of the sub-tile are
// queued as (row | first byte << 8 | end << 16).  `between` runs once every lane holds its masks (the bitmaps may
// be re-used from there on).  Returns the number of units, or -1 when the queue cannot hold them (nothing is
// written then and `between` has not run).
template <class Between>
__device__ __forceinline__ int unit_discover(const cstd::View& D, const uint32_t* bitmap, const uint32_t* xbitmap, int p0, int n, int lane,
                                             uint32_t* uqueue, uint32_t& m0, uint32_t& m1, uint32_t& m2, Between&& between) {
  using namespace cstd;
  uint32_t x0 = 0, x1 = 0, x2 = 0;
  cstile::row_bits96(bitmap, p0, n, m0, m1, m2);
  if ((D.units >> 8) & 127u) cstile::row_bits96(xbitmap, p0, n, x0, x1, x2);
  const U128 C = u128(m0 | ((unsigned long long)m1 << 32), m2), X = u128(x0 | ((unsigned long long)x1 << 32), x2);
  U128 N;
  U128 W = unit_ends(C, X, ((D.units >> 16) & 1u) != 0, N);
  const int cnt = u128_popc(W);
  const int uincl = csdev::wave_inclusive_scan(cnt);
  const int total = __builtin_amdgcn_readlane(uincl, 63);
  if (total > kUnitQueue) return -1;
  cstile::wave_lds_fence();
  between();
  int slot = uincl - cnt;
  while (__any(u128_any(W))) {
    if (u128_any(W)) {
      const int q = u128_ctz(W);
      W = u128_clear_lowest(W);
      uqueue[slot++] = (uint32_t)lane | ((uint32_t)unit_start(N, q) << 8) | ((uint32_t)q << 16);
    }
  }
  cstile::wave_lds_fence();
  return total;
}
// Unit lanes: queue entry -> the unit's row (its position and length come from the row's lane) and the row's
// candidate bits cut to the unit.  Every lane of the wave must call this (shuffles).
// `clip`: the row handed to the unit's scan ends where the unit ends (see reclassify_high: the byte behind it may be >= 0x80).
__device__ __forceinline__ void unit_take(uint32_t ent, int rbeg, int n, uint32_t m0, uint32_t m1, uint32_t m2, int& r, int& rbeg_r, int& n_r,
                                          uint32_t& c0, uint32_t& c1, uint32_t& c2, bool clip = false) {
  using namespace cstd;
  r = (int)(ent & 63u);
  const int us = (int)((ent >> 8) & 255u), uq = (int)((ent >> 16) & 255u);
  rbeg_r = __shfl(rbeg, r, 64);
  n_r = __shfl(n, r, 64);
  if (clip && uq < n_r) n_r = uq;
  const U128 keep = u128_andn(u128_below(uq), u128_below(us));
  c0 = (uint32_t)__shfl((int)m0, r, 64) & (uint32_t)keep.lo;
  c1 = (uint32_t)__shfl((int)m1, r, 64) & (uint32_t)(keep.lo >> 32);
  c2 = (uint32_t)__shfl((int)m2, r, 64) & (uint32_t)keep.hi;
}
// "byte == x" bits of one 16-byte piece (staging, from the prefetch registers)
__device__ __forceinline__ uint32_t unit_xbits16(const uint4& q, uint32_t xpat) {
  auto eq = [&](uint32_t w) {  // bit 7 of every byte lane that equals x (exact: no borrow between lanes)
    const uint32_t t = w ^ xpat;
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
  };
  return cstile::gather16_bit7(eq(q.x), eq(q.y), eq(q.z), eq(q.w));
}

// Sub-tiles that hold bytes >= 0x80 (non-ASCII text).  The lean scan indexes its table with ASCII bytes only, so such a
// sub-tile used to take the generic per-character scan, three to four times slower.  For a pattern none of whose atoms can
// match a non-ASCII character and that has no anchors and no \b (header word 31, bit 17) such a character kills every
// thread, as the end of the row does: the UNIT route can take the sub-tile -- units are runs of candidate / x bytes, ASCII
// all of them -- provided a unit's scan ends at the unit's end (unit_take: clip) instead of reading the killer behind it.
// This pass re-derives the candidate bits from the staged bytes with the bytes >= 0x80 taken out (the classification out
// of the prefetch registers looks at their low seven bits) and says whether the sub-tile holds a NUL byte (then: generic).
// Header word 31 bit 18 (mask in bits 19-23): the pattern's builtin classes (\w, \s, \d) do match some non-ASCII characters
// -- those whose unicode flags meet the mask; the row lanes then walk their rows' non-ASCII characters: the sub-tile
// qualifies when its UTF-8 is well formed inside every row and no character's flags meet the mask.
__device__ __forceinline__ bool reclassify_high(const cstd::View& D, bool has_r2, const uint8_t* lds_in, int want, int lane, uint32_t* bitmap,
                                                const uint8_t* flags, const uint8_t* row, int n) {
  const unsigned fmask = (D.units >> 19) & 31u;
  bool row_bad = false;
  {
    // aligned words of the row, only the bytes >= 0x80 looked at one by one (a lead byte checks its continuation bytes and
    // covers them; a byte >= 0x80 that no lead covers is a stray one).  Also for the patterns such characters can only kill
    // (bit 17): a lead byte WITHOUT its continuation bytes swallows the ASCII bytes behind it (regex_vm.h: char_at), which the
    // unit route would scan -- count_re(b|ab) counted the b of `4\xc3b` (found by the soak, round 5)
    const int al = (int)((uintptr_t)row & 3);
    int covered = 0;  // row offsets below this one belong to a character already checked
    for (int i = -al; i < n && !row_bad; i += 4) {
      uint32_t h = *reinterpret_cast<const uint32_t*>(row + i) & 0x80808080u;
      if (i < 0) h &= 0xFFFFFFFFu << (8 * -i);
      if (i + 4 > n) h &= ~(0xFFFFFFFFu << (8 * (n - i)));
      while (h && !row_bad) {
        const int p = i + (__builtin_ctz(h) >> 3);
        h &= h - 1;
        if (p < covered) continue;
        const uint8_t b = row[p];
        const unsigned w = csrow::lead_width(b);
        row_bad = w < 2 || p + (int)w > n;
        for (unsigned k = 1; k < w && !row_bad; ++k) row_bad = !csrow::is_cont(row[p + (int)k]);
        if (!row_bad) {
          csrow::Char ch;
          csrow::decode_at(row, p, n, ch);
          const unsigned u = csrow::packed_to_cp(ch);
          row_bad = ((D.units >> 18) & 1u) != 0 && u <= 0xFFFFu && (flags[u] & fmask) != 0;
        }
        covered = p + (int)w;
      }
    }
  }
  uint32_t zr = row_bad ? 0x80u : 0u;
  for (int i = lane * 16; i < want; i += 64 * 16) {
    const uint4 q = *reinterpret_cast<const uint4*>(lds_in + i);
    zr |= ((q.x - 0x01010101u) & ~q.x) | ((q.y - 0x01010101u) & ~q.y) | ((q.z - 0x01010101u) & ~q.z) | ((q.w - 0x01010101u) & ~q.w);
    uint32_t bits;
    if (has_r2)
      bits = cstile::gather16_bit7(cstd::Tdfa::cand_bits_ascii<true>(D, q.x) & ~q.x, cstd::Tdfa::cand_bits_ascii<true>(D, q.y) & ~q.y,
                                   cstd::Tdfa::cand_bits_ascii<true>(D, q.z) & ~q.z, cstd::Tdfa::cand_bits_ascii<true>(D, q.w) & ~q.w);
    else
      bits = cstile::gather16_bit7(cstd::Tdfa::cand_bits_ascii<false>(D, q.x) & ~q.x, cstd::Tdfa::cand_bits_ascii<false>(D, q.y) & ~q.y,
                                   cstd::Tdfa::cand_bits_ascii<false>(D, q.z) & ~q.z, cstd::Tdfa::cand_bits_ascii<false>(D, q.w) & ~q.w);
    cstile::put_bits16(bitmap, i, bits);
  }
  cstile::wave_lds_fence();
  // (a zero byte below a byte >= 0x80 may go unseen by the borrow trick's neighbour term; a byte >= 0x80 never looks like one)
  return __any((zr & 0x80808080u) != 0);
}


// The backrefs template in registers (the replace kernel's BREFS form).  The struct lives in device memory; read per match
// and per reference in the kernel's inner loops -- `T.nrefs`, `T.idx[j]`, `T.pos[j]` -- every read was a scalar load and a
// wait that also drains the LDS counter.  Up to sixteen references, group numbers up to 15 and positions up to 255 pack into
// three 64-bit words that stay in scalar registers; the host offers the single pass only to templates that fit.
struct TemplateRegs {
  int bytes, nrefs, groups;
  unsigned long long idx4, pos8lo, pos8hi;
  __device__ __forceinline__ int idx(int j) const { return (int)((idx4 >> (4 * j)) & 15ull); }
  __device__ __forceinline__ int pos(int j) const { return (int)(((j < 8 ? pos8lo >> (8 * j) : pos8hi >> (8 * (j - 8)))) & 255ull); }
};
__device__ __forceinline__ TemplateRegs template_regs(const csvm::BackrefTemplate* t) {
  TemplateRegs r;
  r.bytes = __builtin_amdgcn_readfirstlane(t->bytes);
  r.nrefs = __builtin_amdgcn_readfirstlane(t->nrefs);
  r.groups = __builtin_amdgcn_readfirstlane(t->groups);
  unsigned long long i4 = 0, lo = 0, hi = 0;
  for (int j = 0; j < csvm::BackrefTemplate::kMaxRefs; ++j) {
    const unsigned long long g = (unsigned long long)(t->idx[j] & 15), p = (unsigned long long)(t->pos[j] & 255);
    if (j < r.nrefs) {
      i4 |= g << (4 * j);
      if (j < 8) lo |= p << (8 * j);
      else hi |= p << (8 * (j - 8));
    }
  }
  r.idx4 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(i4 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)i4);
  r.pos8lo = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(lo >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
  r.pos8hi = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(hi >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)hi);
  return r;
}

// ---- the bit-parallel form (regex_bits.h): staging helpers shared by the stream kernels ------------------
// LDS image of the form: the program words as built by the host, then 128 words of the SPREAD class table --
// entry b holds byte b's class set with class k at bit 4k, so that the entries of four bytes, shifted by 0..3
// and OR-ed, leave a nibble per class: the four bytes' membership bits.
__device__ __forceinline__ const uint32_t* bits_stage(const int32_t* g_bits, int words, int32_t* lds) {
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = g_bits[i];
  uint32_t* sp = reinterpret_cast<uint32_t*>(lds + ((words + 3) & ~3));
  if (threadIdx.x < 128) {
    const uint32_t set = ((uint32_t)g_bits[csbits::kHeaderWords + (threadIdx.x >> 2)] >> (8 * (threadIdx.x & 3))) & 255u;
    sp[threadIdx.x] = csbits::spread_entry(set);
  }
  __syncthreads();
  return sp;
}
__host__ __device__ inline int bits_lds_bytes(int words) { return ((words + 3) & ~3) * 4 + 128 * 4; }
// sixteen staged bytes -> for every class the sixteen membership bits (regex_bits.h: classify16)
__device__ __forceinline__ void bits_classify16(const uint32_t* spread, const uint4& q, uint32_t pair[4]) {
  csbits::classify16(spread, q.x, q.y, q.z, q.w, pair);
}
// a row lane's mask of class k, shifted right by `off`: cut out of the class's bitmap on demand (regex_bits.h: cls(k, off))
#define CS_BITS_CLS(bitmap, bm_words, p0, n)                                                                       \
  [&](int k_, int off_) -> csbits::M96 {                                                                           \
    uint32_t m0_, m1_, m2_;                                                                                        \
    cstile::row_bits96((bitmap) + __builtin_amdgcn_readfirstlane(k_) * (bm_words), (p0) + off_, max((n) - off_, 0), m0_, m1_, m2_); \
    return csbits::m96(m0_, m1_, m2_);                                                                             \
  }
// ... without the cut at the row's end (regex_bits.h: raw(k, off)): three funnel shifts over four words of the bitmap
#define CS_BITS_RAW(bitmap, bm_words, p0)                                                                          \
  [&](int k_, int off_) -> csbits::M96 {                                                                           \
    const int q_ = (p0) + off_;                                                                                    \
    const uint32_t* w_ = (bitmap) + __builtin_amdgcn_readfirstlane(k_) * (bm_words) + (q_ >> 5);                   \
    const uint32_t a_ = w_[0], b_ = w_[1], c_ = w_[2], d_ = w_[3];                                                 \
    const unsigned sh_ = (unsigned)q_ & 31u;                                                                       \
    const uint32_t m0_ = __builtin_amdgcn_alignbit(b_, a_, sh_), m1_ = __builtin_amdgcn_alignbit(c_, b_, sh_),     \
                   m2_ = __builtin_amdgcn_alignbit(d_, c_, sh_);                                                   \
    return csbits::m96(m0_, m1_, m2_);                                                                             \
  }

// UNITS (with !INPLACE, RESCAN, !LONG): the scan runs per UNIT instead of per row (regex_tdfa.cpp, header word 31).
// Row lanes cut their rows into units with bit arithmetic on two per-byte bitmaps (candidate bytes, bytes equal
// to x) and queue them in LDS; then every lane takes a unit -- whichever row it belongs to -- and runs the lean scan
// over it, leaving a start bit and a last-byte bit per match in the (re-used) bitmaps, from which the row lanes
// read their rows' matches.  A wave's lock-step scan then costs the longest UNIT (a dotted quad) instead of the
// busiest ROW (two dotted quads and a status code), and runs that cannot hold a match are never scanned.
// PF: 16-byte chunks per lane of the register prefetch (1 KB of the sub-tile each): 6 covers every span the kernel
// takes; the unit variant also exists with 5 (spans up to 5 KB, the usual case), four registers less where it spills.
// BREFS (a UNITS variant): replace_with_backrefs in the same single pass.  The matches come from the unit scan; every
// row lane then runs ONE anchored group run per match of its row (Tdfa::group_find_all) to size the expansion of the
// template, keeps the group ranges of its first two matches, and assembles text pieces and group substrings straight
// from the staged row -- where the two-pass form ran the groups twice at two waves per SIMD and wrote its output a
// byte per lane.
// WIDE: programs of five to eight live threads (counted repetitions): no lean scan, the rows' generic scan with eight start
// offsets (regex_tdfa.h: TdfaWide) -- on the staged rows, where the two-pass kernels read them from memory a thread per row.
// OUTL: the buffers are sized for all but a few sub-tiles (one long row among millions of short ones); an oversize
// sub-tile's rows are sized and written a thread each, straight from memory, inside the prefix chain (separate forms:
// the code costs the others registers).
// HOLES: the hole words of StreamArgs are compiled in.  On by default -- the chain, bit and plain forms measured the same with
// and without them -- but the unit form proper (UNITS, no CHAIN), which literal needles ride too, lost 15 % to the eight registers
// (replace('a','xx') on C2: 0.73 -> 0.83 ms): it exists in both variants, and the host takes the one with holes only when it has some.
template <bool IN_LDS, bool REP16, bool INPLACE, bool RESCAN = false, bool LONG = false, bool UNITS = false, int PF = cstile::kPfChunks, bool BREFS = false,
          bool WIDE = false, bool OUTL = false, bool CHAIN = false, bool BITS = false, bool HOLES = true>
__global__ void __launch_bounds__(256, ((BREFS && !CHAIN) || (BITS && IN_LDS)) ? 2 : CS_STREAM_WAVES) k_tdfa_replace_stream(StreamArgs a) {
  // BITS (a CHAIN form): the pattern has a bit-parallel form (regex_bits.h) -- one bitmap per character class, staged by
  // table lookup; the row lanes derive their rows' matches from the class masks (alternations of word-bounded literals,
  // small sets in a `+` loop: patterns whose candidates are everywhere); a sub-tile with a byte >= 0x80 / NUL or a row
  // beyond the masks goes to the generic scan row by row.  A bitmap per class: two workgroups a CU.
  static_assert(!BITS || CHAIN, "the bit form is a chain-form variant (no unit / lean scans compiled in)");
  // CHAIN (a UNITS form): the pattern is a chain and a sample of the column holds no byte >= 0x80 -- the unit scan, the
  // literal scan and the lean scans are compiled out (their registers with them); a sub-tile the chain arithmetic does
  // not take (a non-ASCII byte after all, a row beyond the masks) goes to the generic scan row by row.
  static_assert(!CHAIN || UNITS, "the chain form is a unit-scan variant");
  // BREFS + CHAIN: replace_with_backrefs on a chain pattern whose groups are runs of items, a sample of the column plain ASCII
  // -- matches and group ranges by marker arithmetic alone, so no table, no group tags, no unit queue in LDS and none of the
  // unit / lean / group-run code in the kernel: three workgroups a CU where the backrefs form proper holds two.  A sub-tile
  // the arithmetic does not take gives the launch up exactly as it does there (error 32: the two-pass form).
  static_assert(!(BREFS && CHAIN) || (!IN_LDS && !BITS), "the backrefs chain form reads no table");
  static_assert(!OUTL || (IN_LDS && !LONG && !UNITS && !BREFS && !WIDE), "oversize sub-tiles: the plain forms only");
  static_assert(!UNITS || (!INPLACE && RESCAN && !LONG), "the unit scan builds on the register-record assembly");
  static_assert(!WIDE || (!UNITS && IN_LDS && !BREFS), "the wide form: generic scan only");
  static_assert(!BREFS || (UNITS && (IN_LDS || CHAIN) && !REP16), "the backrefs form is a unit-scan variant");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint8_t* base = reinterpret_cast<uint8_t*>(smem);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  // one candidate bit per staged byte (BITS: sixteen bytes of slack instead of thirty-two -- a class bitmap is written up to
  // cap / 8 and read a few words beyond that into the next one, whose bits do not count: the seven bitmaps of the gtest
  // pattern then leave room for a third workgroup)
  const int bm_bytes = (a.cap_in >> 3) + (BITS ? 16 : 32);
  // second bitmap, unit queue, bail word (+ BREFS: a record of three words per match, a growth counter per row)
  const int nbm = BITS ? max(a.bits_k, 2) : 2;  // bitmaps per wave (BITS: one per character class)
  // (BREFS + CHAIN: the second bitmap, the bail word's slot and the match records -- no unit queue, no growth counters)
  // (CHAIN, BITS: no unit queue either)
  const int unit_bytes = (BREFS && CHAIN) ? bm_bytes + 16 + kUnitQueue * 12
                         : CHAIN          ? (nbm - 1) * bm_bytes + 16
                         : UNITS          ? (nbm - 1) * bm_bytes + kUnitQueue * 4 + 16 + (BREFS ? kUnitQueue * 12 + 64 * 4 : 0)
                                          : 0;
  uint8_t* lds_in = base + a.tbl_bytes + (size_t)wv * (a.cap_in + a.cap_out + 64 + bm_bytes + unit_bytes);
  uint8_t* lds_out = lds_in + a.cap_in + 32;
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(lds_out + a.cap_out + 32);
  uint32_t* xbitmap = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(bitmap) + bm_bytes);  // UNITS: "byte == x", later the matches' last bytes
  uint32_t* uqueue = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(bitmap) + (size_t)nbm * bm_bytes);
  uint32_t* bailw = uqueue + (CHAIN ? 0 : kUnitQueue);  // one bit per row: the lean scan handed a unit of the row over
  uint32_t* mrec = bailw + 4;              // BREFS: per match (group ranges of groups 1-2, of groups 3-4, match end | ok << 8)
  uint32_t* rowgrow = mrec + kUnitQueue * 3;  // BREFS: bytes the row's expansions add
  const TCtx c = tsetup<IN_LDS>(a.L, a.flags, smem);  // (block barrier inside when staging)
  const cstd::View& D = c.D;
  const csvm::ProgView& P = c.P;
  const ColView& in = a.in;
  const int rb = a.rb;
  const int32_t* gt = nullptr;    // BREFS: the group tags and the template text, in LDS
  const uint8_t* ttext = nullptr;
  if (BREFS) {
    int32_t* g = reinterpret_cast<int32_t*>(base + a.gt_off);
    for (int i = threadIdx.x; i < a.gt_words; i += blockDim.x) g[i] = a.gtags[i];
    uint8_t* tt = base + a.gt_off + ((a.gt_words * 4 + 15) & ~15);
    const int tb = a.tmpl->bytes;
    for (int i = threadIdx.x; i < tb; i += blockDim.x) tt[i] = a.tmpl->text[i];
    __syncthreads();
    gt = g;
    ttext = tt;
  }
  TemplateRegs T{};
  if (BREFS) T = template_regs(a.tmpl);
  const uint32_t* spread = nullptr;
  csbits::View BV{};
  if (BITS) {
    int32_t* bl = reinterpret_cast<int32_t*>(base + a.bits_off);
    spread = bits_stage(a.bits, a.bits_words, bl);
    BV = csbits::make_view(bl);
  }
  const bool has_r2 = (((uint32_t)D.img[30] & 255u) <= (((uint32_t)D.img[30] >> 8) & 255u));
  const uint32_t unit_x = (D.units >> 8) & 127u, unit_xpat = unit_x * 0x01010101u;
  // rows per tile: 64, or fewer for the long-row variants (so that the tile fits the prefetch registers)
  const int R = LONG ? a.rows_per_tile : 64;
  // Tiles are handed out by tickets, so every tile's predecessors were started before it and a
  // wave that meets a slow tile does not hold back the tiles it would have taken next (with a static
  // round-robin the look-back of everybody else waits for them).  One counter per class of
  // workgroups (blockIdx mod K, K = 16): counter k hands out tiles k, k + K, ...; a single word hands out some 65 tickets a
  // microsecond (measured with the strip kernel, cs_rows.hip), this kernel takes 320 -- eight counters ran at 40 each and
  // still cost 0.1 ms against sixteen (4.87 -> 4.75 ms; thirty-two: the same).  The
  // ticket of the tile after next is in flight while the current tile is processed.
  // (sixteen for the launches that hold three workgroups a CU; the slower forms -- two a CU -- measured a little better with eight)
  const long long K = gridDim.x >= 640 ? 16 : (gridDim.x >= 16 ? 8 : 1);  // (every class needs a workgroup that takes tiles: workgroup 0 may be the scanners')
  const long long key = (long long)blockIdx.x % K;
  const bool fixed = (CS_DBG(a) & (256 | 16384)) != 0;  // measurement: the static round-robin
  // Workgroup 0 takes no tiles: its four waves turn the aggregates the other waves publish into each tile's exclusive
  // prefix, in order (a tile then needs ONE load instead of a walk over its predecessors' words): tile_utils.h,
  // prefix_scanner_team.  (debug 512: the decoupled look-back, for comparison; debug 1024: the single scanner wave.)
  const bool fixed_team = (CS_DBG(a) & 16384) != 0 && gridDim.x > 1;  // measurement: the static round-robin with the scanner team
  const bool team = (fixed_team || !fixed) && !(CS_DBG(a) & (512 | 1024)) && gridDim.x > 1;
  const bool scanner = team || (!fixed && !(CS_DBG(a) & 512) && a.cap_in + a.cap_out + 32 >= cstile::kScanBatch * 512);
  if (team && blockIdx.x == 0) {
    cstile::TeamRing* ring = reinterpret_cast<cstile::TeamRing*>(lds_in - (size_t)wv * (a.cap_in + a.cap_out + 64 + bm_bytes + unit_bytes));
    if (threadIdx.x < 8) cstile::team_ring_init(ring, threadIdx.x);
    __syncthreads();
    if (cstile::prefix_scanner_team(a.status, a.excl, a.nsub, lane, wv, 4, ring, a.error) == 1 && lane == 0) atomicOr(a.error, 1u | 16u);  // (16: the scanners timed out)
    return;
  }
  if (scanner && !team && blockIdx.x == 0 && wv == 0) {
    // (its tile buffers are free: they hold the fetched status words)
    if (!cstile::prefix_scanner(a.status, a.excl, a.nsub, lane, reinterpret_cast<cstile::u64*>(lds_in)) && lane == 0)
      atomicOr(a.error, 1u | 16u);  // (16: the scanner wave timed out)
    return;
  }
  const long long W = (long long)gridDim.x * 4;
  unsigned long long* my_ticket = a.tickets + key * 8;
  auto take = [&]() -> unsigned long long {
    unsigned long long t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(my_ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  auto tile_of = [&](unsigned long long t) -> long long { return (long long)cstile::rl64((long long)t, 0) * K + key; };
  long long tile, t_nxt, t_nn = 0;
  unsigned long long pending = 0;
  if (fixed_team) {  // (workgroup 0 is the scanners')
    tile = ((long long)blockIdx.x - 1) * 4 + wv;
    t_nxt = tile + W - 4;
  } else if (fixed) {
    tile = (long long)blockIdx.x * 4 + wv;
    t_nxt = tile + W;
  } else {
    // A wave's first three tickets are drawn ONE AT A TIME, each after the one before has arrived (the tile number is
    // read out of it) and the first tile's loads have been issued: drawn back to back they are consecutive numbers, a wave's
    // second and third tile then lie in front of its neighbour's first, whose prefix has to wait for them -- a dependency
    // chain through every wave of the class.  A round trip apart they land among the other waves' tickets of the same round, as in the steady state (one
    // ticket a wave and iteration).  Nothing here counts waves: workgroups that are not resident draw nothing.
    const unsigned long long q0 = take();
    tile = tile_of(q0);
    t_nxt = a.nsub;  // (known below)
  }
  if (tile >= a.nsub) return;
  // replacement text in registers (this kernel is only taken for rb <= 8, or <= 16 with REP16)
  constexpr int kRepRegs = REP16 ? 4 : 2;
  uint32_t rep[kRepRegs];
#pragma unroll
  for (int i = 0; i < kRepRegs; ++i) rep[i] = 0;
#pragma unroll
  for (int i = 0; i < 4 * kRepRegs; ++i)
    if (i < rb) rep[i >> 2] |= (uint32_t)a.repl[i] << (8 * (i & 3));
  cstile::TileOffs cur = cstile::load_tile_offsets_r(in.offsets, in.rows, tile, R, lane);
  cstile::TileCharsT<PF> pf;
#pragma unroll
  for (int j = 0; j < PF; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
  if (!fixed) {
    const unsigned long long q1 = take();
    t_nxt = tile_of(q1);
  }
  cstile::TileOffs nxt = cur;
  if (t_nxt < a.nsub) nxt = cstile::load_tile_offsets_r(in.offsets, in.rows, t_nxt, R, lane);
  if (!fixed) pending = take();
  // The previous sub-tile's output stays assembled in lds_out while this one is scanned;
  // its look-back completes afterwards, when every predecessor's aggregate has long been
  // published, so waves do not wait on each other's scans.
  long long p_tile = -1;
  int p_total = 0, p_lo = 0, p_len = 0;
  int m_span = 0;  // the output column's largest 64-row span as a by-product (wave-uniform: a scalar maximum per sub-tile)
  // HOLES: the tile's mask of rows that are sized and written elsewhere, and where their sizes begin -- fetched a tile ahead,
  // like the offsets (unconditionally, from a harmless address when the launch has none: see the note at the first poll)
#if defined(CS_NO_HOLE_FORMS)  // (measurement: the kernels without the hole words)
  constexpr bool kHoles = false;
#else
  constexpr bool kHoles = HOLES && !BREFS && !LONG && !OUTL && !WIDE;
#endif
  const bool holes = kHoles && a.hole_mask != nullptr;
  // (the column's first offsets: read-only, sixteen bytes at least -- not the ticket words, which every wave's atomics hammer)
  const unsigned long long* hole_mask_p = holes ? a.hole_mask : reinterpret_cast<const unsigned long long*>(in.offsets);
  const int64_t* hole_first_p = holes ? a.hole_first : in.offsets;
  unsigned long long c_hm = 0;
  long long c_hf = 0;
  if (kHoles) {
    c_hm = hole_mask_p[holes ? tile : 0];
    c_hf = hole_first_p[holes ? tile : 0];
  }
#if defined(CS_PHASE_PROF)
  unsigned long long phase_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long lb_acc[3] = {0, 0, 0};
  unsigned long long phase_t = __builtin_readcyclecounter();
#endif
  auto finish_pending = [&](cstile::u64 first) {
    long long gb;
    if (lane == 0) CS_TILE_TRACE(p_tile, 2);
    if (CS_DBG(a) & 128) {  // measurement only: one late poll, no chase for an inclusive prefix
      const cstile::u64 v = cstile::lookback_poll(a.status, p_tile, lane);
      const int part = (int)(unsigned)(v & 0xffffffffull);
      gb = p_tile * 4096 + (csdev::wave_reduce_sum(part) & 0) + (long long)(first & 0);
      if (lane == 0) cstile::status_store(a.status + p_tile, cstile::kFlagInc | 1);
    } else if (__builtin_expect(scanner, 1)) {
      gb = (CS_DBG(a) & 8) ? p_tile * 4096 : cstile::prefix_wait(a.excl, p_tile, first, a.error, lane);
    } else {
#if defined(CS_PHASE_PROF)
      gb = (CS_DBG(a) & 8) ? p_tile * 4096 : cstile::lookback_end(a.status, p_tile, p_total, first, lane, lb_acc);
#else
      gb = (CS_DBG(a) & 8) ? p_tile * 4096 : cstile::lookback_end(a.status, p_tile, p_total, first, lane);
#endif
    }
    if (gb < 0) {
#if defined(CS_PHASE_PROF)  // (a device printf costs every variant of this kernel thousands of instructions: profiling builds only)
      if (lane == 0 && (CS_DBG(a) & 2048)) printf("no prefix: tile %lld (wave %d of block %d, current tile %lld) error %u\n", p_tile, wv, (int)blockIdx.x, tile, *a.error);
#endif
      if (lane == 0) atomicOr(a.error, 1u | 8u);  // (8: no prefix for the tile)
      gb = 0;
    }
    CS_PHASE_MARK(6);
    if (lane == 0) CS_TILE_TRACE(p_tile, 3);
    const long long pr0 = p_tile * R;
    const int pn = (int)min((long long)R, in.rows - pr0);
#if defined(CS_NT_STORES)
    if (lane < pn) __builtin_nontemporal_store(gb + p_lo, a.out_off + pr0 + lane);
#else
    if (lane < pn) a.out_off[pr0 + lane] = gb + p_lo;
#endif
    if (lane == pn - 1 && pr0 + pn == in.rows) a.out_off[in.rows] = gb + p_lo + p_len;
    if (!INPLACE && gb + p_total > a.out_cap) {  // more growth than the host provisioned for
      if (lane == 0) atomicOr(a.error, 2u);
      return;
    }
    if (!(CS_DBG(a) & 4)) cstile::wave_flush_shift(a.out_chars + gb, p_total, lds_out, lane);
  };
  for (;;) {
    const long long r0 = tile * R;
    const int nrows = (int)min((long long)R, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool live0 = lane < nrows && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n0 = live0 ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const long long want = g1 - g0 + lead;
    bool bad = want + 16 > a.cap_in || want > PF * 1024;
    // A sub-tile beyond the staging buffer.  The host sizes the buffers for all but a few sub-tiles when the column's
    // largest does not fit (one long row among millions of short ones): such a sub-tile's rows are sized and written a
    // thread each, straight from memory -- inside the prefix chain -- instead of the whole column leaving the single pass.
    const bool oversize = OUTL && bad && a.outliers;
    if (!bad) cstile::stage_chars(lds_in, (int)want, lane, pf);
    CS_PHASE_MARK(9);   // (wait for the prefetched chars + the LDS writes)
    // bytes that the lean scan does not take (non-ASCII, NUL) anywhere in the staged span, and one
    // candidate bit per byte for the row lanes (classified here, out of the prefetch registers)
    uint32_t odd = 0;
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (j * 1024 + lane * 16 < (int)want) {
        uint4 q = pf.v[j];
        if (__builtin_expect(tile + 1 == a.nsub && j * 1024 + lane * 16 + 16 > (int)want, 0)) {
          // the column's last piece: what lies behind its last byte is allocation slack, not data (zeros there made the
          // last sub-tile look as if it held NUL bytes and sent it to the generic scan)
          const int keep = (int)want - (j * 1024 + lane * 16);  // 1..15 bytes of data
          auto cut = [&](uint32_t w, int k) {
            const int left = keep - 4 * k;
            const uint32_t m = left >= 4 ? 0xFFFFFFFFu : (left <= 0 ? 0u : (1u << (8 * left)) - 1u);
            return (w & m) | (0x20202020u & ~m);
          };
          q = make_uint4(cut(q.x, 0), cut(q.y, 1), cut(q.z, 2), cut(q.w, 3));
        }
        odd |= q.x | ((q.x - 0x01010101u) & ~q.x);
        odd |= q.y | ((q.y - 0x01010101u) & ~q.y);
        odd |= q.z | ((q.z - 0x01010101u) & ~q.z);
        odd |= q.w | ((q.w - 0x01010101u) & ~q.w);
        if (BITS) {
          uint32_t pair[4];
          bits_classify16(spread, q, pair);
#pragma unroll
          for (int k = 0; k < csbits::kMaxClasses; ++k)
            if (k < BV.K && !bad) cstile::put_bits16(bitmap + k * (bm_bytes >> 2), j * 1024 + lane * 16, (pair[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
          continue;
        }
        uint32_t bits;
        if (has_r2)
          bits = cstile::gather16_bit7(cstd::Tdfa::cand_bits_ascii<true>(D, q.x), cstd::Tdfa::cand_bits_ascii<true>(D, q.y), cstd::Tdfa::cand_bits_ascii<true>(D, q.z),
                                       cstd::Tdfa::cand_bits_ascii<true>(D, q.w));
        else
          bits = cstile::gather16_bit7(cstd::Tdfa::cand_bits_ascii<false>(D, q.x), cstd::Tdfa::cand_bits_ascii<false>(D, q.y), cstd::Tdfa::cand_bits_ascii<false>(D, q.z),
                                       cstd::Tdfa::cand_bits_ascii<false>(D, q.w));
        if (!bad) cstile::put_bits16(bitmap, j * 1024 + lane * 16, bits);
        if (UNITS && unit_x != 0 && !bad) cstile::put_bits16(xbitmap, j * 1024 + lane * 16, unit_xbits16(q, unit_xpat));
      }
    // first look-back poll for the previous sub-tile: issued only now, after the staging above has
    // waited for its own data (vmcnt is in order: a poll issued earlier would sit in front of it)
    CS_PHASE_MARK(10);  // (classification into the bitmaps)
    // Everything fetched here for later -- the poll, the ticket, the offsets two tiles ahead -- is assigned
    // UNCONDITIONALLY (clamped addresses) and handed to its loop-carried variable only at the bottom of the iteration:
    // a conditional assignment of a value that is still in flight makes the compiler copy it at the join of the branch,
    // i.e. wait for the load right behind its issue -- s_waitcnt vmcnt(0), which drained the whole prefetch every
    // iteration (found in the ISA of every stream kernel; the "prefetch" had never overlapped anything).
    cstile::u64 p_first, p_second;
    bool has_second = false;
    {
      const long long pt = p_tile >= 0 ? p_tile : 0;
      p_first = __builtin_expect(scanner, 1) ? cstile::status_load(a.excl + pt) : cstile::lookback_poll(a.status, pt, lane);
    }
    // keep the memory pipe busy: next sub-tile's chars, and the offsets of the one after
    const bool has_next = t_nxt < a.nsub;
    t_nn = fixed ? t_nxt + (fixed_team ? W - 4 : W) : tile_of(pending);
    const unsigned long long pending_new = fixed ? 0ull : take();  // (tickets drawn past the end are harmless)
    const cstile::TileOffs nn = cstile::load_tile_offsets_r(in.offsets, in.rows, t_nn < a.nsub ? t_nn : a.nsub - 1, R, lane);
    unsigned long long n_hm = 0;
    long long n_hf = 0;
    if (kHoles) {
      const long long tq = holes ? (has_next ? t_nxt : tile) : 0;
      n_hm = hole_mask_p[tq];
      n_hf = hole_first_p[tq];
    }
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
    }
    cstile::wave_lds_fence();
    CS_PHASE_MARK(0);
    // HOLES: the rows of this sub-tile that hold a byte >= 0x80 or a NUL take no part below -- and the others are plain
    bool hole = false, tile_plain = false;
    int hole_bytes = 0;
    if (kHoles && holes && !bad) {
      const unsigned long long hm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(c_hm >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)c_hm);
      // (a row beyond the masks: the sub-tile goes row by row as before -- its rows of the list are then written twice, the same bytes)
      tile_plain = !__any(live0 && n0 + ((lead + rbeg) & 3) > cstd::Tdfa::kMaskBytes);
      if (tile_plain && hm) {
        hole = live0 && ((hm >> lane) & 1ull) != 0;
        if (hole) hole_bytes = a.hole_len[cstile::rl64(c_hf, 0) + __builtin_popcountll(hm & ((1ull << lane) - 1ull))];
      }
    }
    const bool live = live0 && !hole;
    const int n = live ? n0 : 0;

    int rec_mb[kMaxRec], rec_me[kMaxRec], rec_reps[kMaxRec];
    int nm = 0;
    int out_len = 0;
    cstd::U128 uS = cstd::u128(0, 0), uE = cstd::u128(0, 0);  // UNITS: the row's match starts / last bytes
    bool from_masks = false;
    int mslot = 0;  // BREFS: where the row's matches stand in the match queue / record table
    bool dense_tile = false;  // BREFS + CHAIN: more matches than the record table holds -- the assembly derives the groups again
    if (live && !bad) out_len = n;
    if (oversize && live) {  // the row's output size by the generic scan on the row in memory
      cstd::Tdfa vg(D, P, in.chars + (g0 + rbeg), n);
      int len = n;
      auto add = [&](int mb, int me, int reps) { len += reps * rb - (me - mb); };
      if (WIDE) vg.template scan<cstd::Tdfa::K_REPLACE, decltype(add)&, cstd::kMaxSlotsWide>(a.maxrepl, add, 0, 0);
      else vg.template scan<cstd::Tdfa::K_REPLACE>(a.maxrepl, add, 0, 0);
      out_len = len;
    }
    if (!bad && !(CS_DBG(a) & 1)) {
      cstd::Tdfa vm(D, P, lds_in + lead + rbeg, n, (lead + rbeg) & 3);
      const int pi = lead + rbeg;          // byte index of the row in lds_in
      int wr = 0, copied = 0, pend = -1;   // INPLACE: write cursor, input consumed, replacement not yet written
      auto put_repl_at = [&](int at) {
        if (REP16) {
          if (rb > 16) {  // (a long replacement: its bytes from memory, the same for every lane)
            for (int i = 0; i < rb; ++i) lds_in[pi + at + i] = a.repl[i];
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              if (i < rb) lds_in[pi + at + i] = (uint8_t)(rep[i >> 2] >> (8 * (i & 3)));
          }
        } else {
          lds_put_short(lds_in + pi + at, rep[0], rep[1], rb);
        }
      };
      auto rec = [&](int mb, int me, int reps) {
        if (INPLACE) {
          // the previous match's replacement is written only now: until the next round has started the
          // scan may still look at the byte in front of that match's end (word / line context)
          if (pend >= 0) put_repl_at(pend);
          if (wr != copied) cstile::lds_copy(lds_in, pi + wr, lds_in, pi + copied, mb - copied);  // moves down: ascending copy is safe
          wr += mb - copied;
          pend = wr;
          wr += reps * rb;  // (reps > 1 only for patterns that match the empty string, where rb == 0)
          copied = me;
        } else {
          out_len += reps * rb - (me - mb);
          // (UNITS: the rows that come through here -- non-ASCII sub-tiles, rows a unit handed over -- are only
          // measured; the assembly scans them again.  Twelve registers less in a kernel at its VGPR limit, whose
          // spill reloads -- VMEM operations -- otherwise wait behind the prefix poll in every iteration.)
          if (!UNITS) {
#pragma unroll
            for (int j = 0; j < kMaxRec; ++j)
              if (nm == j) {
                rec_mb[j] = mb;
                rec_me[j] = me;
                rec_reps[j] = reps;
              }
          }
        }
        ++nm;
      };
      // wave-uniform choice: the lean scan when every row of the sub-tile qualifies
      const bool has_odd = !tile_plain && __any((odd & 0x80808080u) != 0);
      // (a sub-tile with bytes >= 0x80, a pattern they can only kill: the UNIT route alone -- reclassify_high)
      bool hi_units = false;
      if (!CHAIN && UNITS && !BREFS && has_odd && a.litn == 0 && ((D.units >> 17) & 3u) && (D.units & 1u) && a.maxrepl < 0 && !(CS_DBG(a) & 4096))
        hi_units = !reclassify_high(D, has_r2, lds_in, (int)want, lane, bitmap, a.flags, lds_in + lead + rbeg, n);
      const bool lean = D.nskip > 0 && D.img[12] <= 4 && !(CS_DBG(a) & 32) && (!has_odd || hi_units) &&
                        !__any(live && (LONG ? n > cstd::Tdfa::kLongBytes : !vm.masks_fit()));
      bool redo = live && !lean && a.maxrepl != 0;
      int resume = 0;
      bool units_done = false;
      from_masks = false;
      if (!CHAIN && UNITS && a.litn > 0) {
        // A literal needle without a border: every match of the sub-tile by byte comparison, sixteen positions per
        // lane and step -- no automaton.  Match starts land in `bitmap`; `xbitmap` holds the row starts, so that a
        // match never spans two rows (or the end of the staged span).  Rows of any bytes qualify (an ASCII needle
        // never matches inside a multi-byte character), rows beyond the 96-bit masks do not.
        if (a.maxrepl < 0 && !__any(live && !vm.masks_fit())) {  // (wave-uniform)
          using namespace cstd;
          const int m = a.litn;
          for (int i = lane * 16; i < bm_bytes; i += 64 * 16) *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(xbitmap) + i) = make_uint4(0, 0, 0, 0);
          cstile::wave_lds_fence();
          {
            const int p = lane < nrows ? lead + rbeg : (int)want;  // (every row, null or not)
            __hip_atomic_fetch_or(xbitmap + (p >> 5), 1u << (p & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            // and the end of the staged span: the last row's match must not run on into the next sub-tile's bytes
            if (lane == 0) __hip_atomic_fetch_or(xbitmap + ((int)want >> 5), 1u << ((int)want & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
          cstile::wave_lds_fence();
#pragma unroll
          for (int j = 0; j < PF; ++j) {
            const int i = j * 1024 + lane * 16;
            if (i < (int)want) {
              const uint32_t* wp = reinterpret_cast<const uint32_t*>(lds_in + i);
              const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3], w4 = wp[4], w5 = wp[5];
              uint32_t M = 0xFFFFFFu;
              for (int k = 0; k < m; ++k) {
                const uint32_t cpat = (uint32_t)((a.lit >> (8 * k)) & 255u) * 0x01010101u;
                auto eq = [&](uint32_t w) {
                  const uint32_t t = w ^ cpat;
                  return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
                };
                const uint32_t hi8 = __builtin_amdgcn_udot4(eq(w5), 0x80402010u, __builtin_amdgcn_udot4(eq(w4), 0x08040201u, 0u, false), false) >> 7;
                M &= (cstile::gather16_bit7(eq(w0), eq(w1), eq(w2), eq(w3)) | (hi8 << 16)) >> k;
              }
              // row starts at i .. i + 23 (a row start at p + k, 0 < k < m, forbids a match at p)
              const uint32_t* xb = xbitmap + (i >> 5);
              const unsigned sh = (unsigned)(i & 31);
              const uint32_t rs = sh ? (xb[0] >> sh) | (xb[1] << (32 - sh)) : xb[0];
              for (int k = 1; k < m; ++k) M &= ~(rs >> k);
              const int room = (int)want - i;  // positions of this piece inside the staged span
              if (room < 16) M &= (1u << room) - 1u;
              cstile::put_bits16(bitmap, i, M & 0xFFFFu);
            }
          }
          cstile::wave_lds_fence();
          uint32_t s0, s1, s2;
          cstile::row_bits96(bitmap, lead + rbeg, n, s0, s1, s2);
          if (live) {
            uS = u128(s0 | ((unsigned long long)s1 << 32), s2);
            const int up = m - 1;  // the match's last byte
            uE = up ? u128(uS.lo << up, (uS.hi << up) | (uS.lo >> (64 - up))) : uS;
            nm = u128_popc(uS);
            out_len = n + nm * (rb - m);
            from_masks = true;
          }
          redo = false;  // (no row needs the automaton)
          units_done = true;
        }
      } else if (UNITS) {
        if (BITS && !has_odd && a.maxrepl < 0 && !__any(live && n > csbits::kMaxRowBytes)) {  // (wave-uniform) plain ASCII, rows within the masks
          using namespace cstd;
          p_second = cstile::status_load(a.excl + (p_tile >= 0 ? p_tile : 0));
          has_second = scanner;
          if (live) {
            auto cls = CS_BITS_CLS(bitmap, bm_bytes >> 2, lead + rbeg, n);
            auto raw = CS_BITS_RAW(bitmap, bm_bytes >> 2, lead + rbeg);
            csbits::match(BV, cls, raw, n, uS, uE);
            nm = u128_popc(uS);
            out_len = n - u128_popc(u128_sub(u128_shl1(uE), uS)) + nm * rb;
            from_masks = true;
          }
          redo = false;
          units_done = true;
        } else if (!BITS && !BREFS && lean && !has_odd && a.maxrepl < 0 && (D.chain >> 16) && !(CS_DBG(a) & 8192)) {  // (wave-uniform)
          // A chain pattern (regex_tdfa.h: chain_match) on a sub-tile of plain ASCII: every row lane derives its row's
          // matches from the row's candidate and x bits by integer arithmetic -- no unit queue, no table walk, no LDS
          // traffic beyond the two mask reads.
          using namespace cstd;
          uint32_t r0, r1, r2, x0 = 0, x1 = 0, x2 = 0;
          cstile::row_bits96(bitmap, lead + rbeg, n, r0, r1, r2);
          if (unit_x != 0) cstile::row_bits96(xbitmap, lead + rbeg, n, x0, x1, x2);
          p_second = cstile::status_load(a.excl + (p_tile >= 0 ? p_tile : 0));
          has_second = scanner;
          if (live) {
            chain_match96(r0, r1, r2, x0, x1, x2, D.chain, D.img, uS, uE, D.sfx, n, [&](int i) { return lds_in[lead + rbeg + i]; });
            nm = u128_popc(uS);
            out_len = n - u128_popc(u128_sub(u128_shl1(uE), uS)) + nm * rb;
            from_masks = true;
          }
          redo = false;
          units_done = true;
        } else if (BREFS && lean && !has_odd && a.maxrepl < 0 && (D.chain >> 16) && (((uint32_t)D.img[30] >> 20) & 1u) && !(CS_DBG(a) & 8192)) {  // (wave-uniform)
          // replace_with_backrefs on a chain pattern: the matches by chain_match, and every capture group is a run of
          // items, so its range follows from the item boundaries of the match -- a walk over the row's two masks per
          // match where the automaton needed an anchored group run (half the kernel's time).
          using namespace cstd;
          const uint32_t gmap = (uint32_t)D.img[D.img[15] - 1];
          uint32_t r0, r1, r2, x0 = 0, x1 = 0, x2 = 0;
          cstile::row_bits96(bitmap, lead + rbeg, n, r0, r1, r2);
          if (unit_x != 0) cstile::row_bits96(xbitmap, lead + rbeg, n, x0, x1, x2);
          const U128 R = u128(r0 | ((unsigned long long)r1 << 32), r2), X = u128(x0 | ((unsigned long long)x1 << 32), x2);
          p_second = cstile::status_load(a.excl + (p_tile >= 0 ? p_tile : 0));
          has_second = scanner;
          if (live) {
            chain_match96(r0, r1, r2, x0, x1, x2, D.chain, D.img, uS, uE, D.sfx, n, [&](int i) { return lds_in[lead + rbeg + i]; });
            nm = u128_popc(uS);
            from_masks = true;
          }
          const int cnt = from_masks ? nm : 0;
          const int mincl = csdev::wave_inclusive_scan(cnt);
          const int total_m = __builtin_amdgcn_readlane(mincl, 63);
          mslot = mincl - cnt;
          // (the chain form proper keeps going when the sub-tile holds more matches than the record table: the sizes now, the
          // group ranges once more at the assembly -- replace_with_backrefs((\d+), <\1>) on the C3 column, 5 numbers a row, gave
          // every launch up and took 174 ms on the two-pass form)
          dense_tile = CHAIN && total_m > kUnitQueue;
          if (!CHAIN && total_m > kUnitQueue) {
            if (lane == 0) atomicOr(a.error, 1u | 32u);
          } else {
            int grow_row = 0, mi = 0;
            U128 S = uS, E = uE;
            while (u128_any(S)) {
              const int mb = u128_ctz(S), me = u128_ctz(E) + 1;
              S = u128_clear_lowest(S);
              E = u128_clear_lowest(E);
              int gb[4] = {-1, -1, -1, -1}, ge[4] = {-1, -1, -1, -1};
              if (T.nrefs > 0) chain_group_bounds(R, X, D.chain, D.img, gmap, mb, gb, ge);  // (a template without references needs no groups)
              int grow = T.bytes - (me - mb);
              for (int j = 0; j < T.nrefs; ++j) {
                const int g = T.idx(j);
                int x = -1, y = -1;
                if (g == 0) {
                  x = mb;
                  y = me;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  if (g == q + 1 && g <= T.groups) {
                    x = gb[q];
                    y = ge[q];
                  }
                if (x >= 0 && y > x) grow += y - x;
              }
              grow_row += grow;
              auto by = [&](int v) { return (uint32_t)(v >= 0 ? v : 255) & 255u; };
              if (!dense_tile) {
                uint32_t* rec = mrec + (mslot + mi) * 3;
                rec[0] = by(gb[0]) | (by(ge[0]) << 8) | (by(gb[1]) << 16) | (by(ge[1]) << 24);
                rec[1] = by(gb[2]) | (by(ge[2]) << 8) | (by(gb[3]) << 16) | (by(ge[3]) << 24);
                rec[2] = by(me) | (1u << 8);
              }
              ++mi;
            }
            if (from_masks) out_len = n + grow_row;
          }
          redo = false;
          units_done = true;
        } else if (!CHAIN && lean && a.maxrepl < 0 && (D.units & 1u) && !(CS_DBG(a) & 1024)) {  // (wave-uniform)
          using namespace cstd;
          // -- row lanes: the row's units from its candidate and x bits
          uint32_t m0, m1, m2;
          const int total_units = unit_discover(D, bitmap, xbitmap, lead + rbeg, n, lane, uqueue, m0, m1, m2, [&] {
            // every lane holds its masks: both bitmaps are re-used for the matches
            for (int i = lane * 16; i < bm_bytes; i += 64 * 16) {
              *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(bitmap) + i) = make_uint4(0, 0, 0, 0);
              *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(xbitmap) + i) = make_uint4(0, 0, 0, 0);
            }
            if (lane < 2) bailw[lane] = 0;
          });
          if (total_units >= 0) {
            CS_PHASE_MARK(7);
            // -- unit lanes: one unit each, whatever row it lies in
            for (int u0 = 0; u0 < total_units; u0 += 64) {
              const bool act = u0 + lane < total_units;
              int r, rbeg_r, n_r;
              uint32_t c0, c1, c2;
              unit_take(act ? uqueue[u0 + lane] : 0u, rbeg, n, m0, m1, m2, r, rbeg_r, n_r, c0, c1, c2, hi_units);
              if (act) {
                const int pu = lead + rbeg_r;
                cstd::Tdfa vu(D, P, lds_in + pu, n_r, pu & 3);
                bool ubail = false;
                auto recu = [&](int mb, int me, int) {
                  const int ps = pu + mb, pe = pu + me - 1;
                  __hip_atomic_fetch_or(bitmap + (ps >> 5), 1u << (ps & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  __hip_atomic_fetch_or(xbitmap + (pe >> 5), 1u << (pe & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                };
                vu.scan_lean_dispatch(-1, c0, c1, c2, recu, ubail);
                if (ubail) __hip_atomic_fetch_or(bailw + (r >> 5), 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              }
            }
            cstile::wave_lds_fence();
            CS_PHASE_MARK(8);
            // the previous tile's prefix: the poll issued before the scan came too early more often than not (the
            // scanner wave needs every earlier aggregate first); this one has the extraction below to arrive in
            // (into its own variable, unconditionally: see the note at the first poll)
            p_second = cstile::status_load(a.excl + (p_tile >= 0 ? p_tile : 0));
            has_second = scanner;
            // -- row lanes again: the row's matches from the start / last-byte bits
            uint32_t s0, s1, s2, e0, e1, e2;
            cstile::row_bits96(bitmap, lead + rbeg, n, s0, s1, s2);
            cstile::row_bits96(xbitmap, lead + rbeg, n, e0, e1, e2);
            const bool rbail = ((bailw[lane >> 5] >> (lane & 31)) & 1u) != 0;
            if (live && !rbail) {
              // the matches stay in the two masks until the rows are assembled (any number of them per row)
              uS = u128(s0 | ((unsigned long long)s1 << 32), s2);
              uE = u128(e0 | ((unsigned long long)e1 << 32), e2);
              nm = u128_popc(uS);
              out_len = n - u128_popc(u128_sub(u128_shl1(uE), uS)) + nm * rb;
              from_masks = true;
            }
            redo = live && rbail;  // (such a row is scanned whole; nothing of it was recorded above)
            if (BREFS) {
              if (__any(redo)) {  // (the template needs the unit route's masks: the host repeats with the two-pass form)
                if (lane == 0) atomicOr(a.error, 1u | 32u);
                redo = false;
              }
              {
                // the matches of the sub-tile go through the queue once more, one per lane whatever its row (as the units
                // did): ONE anchored group run per match sizes its expansion and leaves its group ranges in LDS
                const int cnt = from_masks ? nm : 0;
                const int mincl = csdev::wave_inclusive_scan(cnt);
                const int total_m = __builtin_amdgcn_readlane(mincl, 63);
                mslot = mincl - cnt;
                if (total_m > kUnitQueue) {
                  if (lane == 0) atomicOr(a.error, 1u | 32u);
                } else {
                  cstile::wave_lds_fence();  // (the unit lanes are done with the queue)
                  {
                    U128 S = uS, E = uE;
                    int at = mslot;
                    while (__any(u128_any(S))) {
                      if (u128_any(S)) {
                        const int mb = u128_ctz(S), me = u128_ctz(E) + 1;
                        S = u128_clear_lowest(S);
                        E = u128_clear_lowest(E);
                        uqueue[at++] = (uint32_t)lane | ((uint32_t)mb << 8) | ((uint32_t)me << 16);
                      }
                    }
                  }
                  rowgrow[lane] = 0;
                  cstile::wave_lds_fence();
                  for (int u0 = 0; u0 < total_m; u0 += 64) {
                    const bool act = u0 + lane < total_m;
                    const uint32_t ent = act ? uqueue[u0 + lane] : 0u;
                    const int r = (int)(ent & 63u), mb = (int)((ent >> 8) & 255u), me = (int)((ent >> 16) & 255u);
                    const int rbeg_r = __shfl(rbeg, r, 64), n_r = __shfl(n, r, 64);
                    if (act) {
                      const int pu = lead + rbeg_r;
                      cstd::Tdfa vg(D, P, lds_in + pu, n_r, pu & 3);
                      int gb[cstd::Tdfa::kGroupBatch], ge[cstd::Tdfa::kGroupBatch], mend = me;
                      // backwards from the match where that applies (regex_tdfa.h: group_find_back); the history bytes
                      // lie in the two bitmaps, whose bits every row lane has taken into its masks by now
                      const int hsteps = min(cstd::Tdfa::kBackSteps, (2 * bm_bytes) >> 6);
                      int got = T.nrefs > 0 ? vg.group_find_back(mb, gt, 1, T.groups, gb, ge, mend, cstd::Tdfa::HistBytes{reinterpret_cast<uint8_t*>(bitmap) + lane, 64}, hsteps) : 0;
                      if (got < 0) got = vg.group_find_all(mb, gt, 1, T.groups, gb, ge, mend);
                      const bool ok = got > 0;
                      int grow = T.bytes - (me - mb);
                      for (int j = 0; j < T.nrefs; ++j) {
                        const int g = T.idx(j);
                        int x = -1, y = -1;
                        if (g == 0) {
                          x = mb;
                          y = mend;
                        }
#pragma unroll
                        for (int q = 0; q < cstd::Tdfa::kGroupBatch; ++q)
                          if (g == q + 1 && g <= T.groups) {
                            x = gb[q];
                            y = ge[q];
                          }
                        if (ok && x >= 0 && y > x) grow += y - x;
                      }
                      __hip_atomic_fetch_add(rowgrow + r, (uint32_t)grow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                      auto by = [&](int v) { return (uint32_t)(ok && v >= 0 ? v : 255) & 255u; };
                      uint32_t* rec = mrec + (u0 + lane) * 3;
                      rec[0] = by(gb[0]) | (by(ge[0]) << 8) | (by(gb[1]) << 16) | (by(ge[1]) << 24);
                      rec[1] = by(gb[2]) | (by(ge[2]) << 8) | (by(gb[3]) << 16) | (by(ge[3]) << 24);
                      rec[2] = by(mend) | ((ok ? 1u : 0u) << 8);
                    }
                  }
                  cstile::wave_lds_fence();
                  if (from_masks) out_len = n + (int)rowgrow[lane];
                }
              }
            }
            units_done = true;
          }
        }
      }
      if (BREFS && !units_done) {  // (only the unit route carries the template: the host repeats with the two-pass form)
        if (lane == 0) atomicOr(a.error, 1u | 32u);
        redo = false;
      }
      if (hi_units && !units_done) redo = live;  // (more units than the queue holds: the generic scan, never the lean scan on such bytes)
      if (CHAIN && !BREFS && !units_done) redo = live && a.maxrepl != 0;  // (the generic scan)
      if (!CHAIN && UNITS && !BREFS && !units_done && lean && !hi_units && a.maxrepl < 0 && D.img[13] >= 1 && !(CS_DBG(a) & 2048)) {  // (wave-uniform)
        // A sub-tile the unit route did not take (more units than the queue holds: patterns whose candidate bytes are
        // everywhere, such as alternations of word-bounded literals; or no decomposition at all): every row lane scans its
        // own row, but the matches still go into the two bitmaps -- a start bit and a last-byte bit each, as the unit lanes
        // leave them -- so that the rows are assembled from the masks.  Before, such rows were only MEASURED here and
        // scanned a second time at assembly by the generic byte-at-a-time scanner: one matching row in a wave made the
        // other 63 lanes wait for that whole scan (52 ms on the 100M-row column for (\bin\b)|(\ba\b)|(\bthe\b), which
        // matches in 1.3 % of the rows).
        using namespace cstd;
        uint32_t m0, m1, m2;
        cstile::row_bits96(bitmap, lead + rbeg, n, m0, m1, m2);
        cstile::wave_lds_fence();
        for (int i = lane * 16; i < bm_bytes; i += 64 * 16) {
          *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(bitmap) + i) = make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(xbitmap) + i) = make_uint4(0, 0, 0, 0);
        }
        cstile::wave_lds_fence();
        bool bail = false;
        if (live) {
          const int pu = lead + rbeg;
          auto recm = [&](int mb, int me, int) {
            const int ps = pu + mb, pe = pu + me - 1;
            __hip_atomic_fetch_or(bitmap + (ps >> 5), 1u << (ps & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_or(xbitmap + (pe >> 5), 1u << (pe & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          };
          vm.scan_lean_dispatch(-1, m0, m1, m2, recm, bail);
        }
        cstile::wave_lds_fence();
        uint32_t s0, s1, s2, e0, e1, e2;
        cstile::row_bits96(bitmap, lead + rbeg, n, s0, s1, s2);
        cstile::row_bits96(xbitmap, lead + rbeg, n, e0, e1, e2);
        if (live && !bail) {
          uS = u128(s0 | ((unsigned long long)s1 << 32), s2);
          uE = u128(e0 | ((unsigned long long)e1 << 32), e2);
          nm = u128_popc(uS);
          out_len = n - u128_popc(u128_sub(u128_shl1(uE), uS)) + nm * rb;
          from_masks = true;
        }
        redo = live && bail;  // (such a row is scanned whole by the generic scan below; nothing of it counts from above)
        units_done = true;
      }
      if (!CHAIN && !BREFS && !units_done && lean && !hi_units && live && a.maxrepl != 0) {
        bool bail = false;
        uint32_t m0, m1, m2;  // candidate bits, bit i = byte i of the row
        if (LONG) {
          const int p0 = lead + rbeg;
          cstile::row_bits96(bitmap, p0, min(n, 96), m0, m1, m2);
          auto refill = [&](int wb, uint32_t& x0, uint32_t& x1, uint32_t& x2) {
            cstile::row_bits96(bitmap, p0 + wb, min(n - wb, 96), x0, x1, x2);
          };
          vm.scan_lean_dispatch_long(a.maxrepl, m0, m1, m2, rec, bail, refill);
        } else {
          cstile::row_bits96(bitmap, lead + rbeg, n, m0, m1, m2);
          vm.scan_lean_dispatch(a.maxrepl, m0, m1, m2, rec, bail);
        }
        redo = bail;
        resume = vm.lean_resume_from;
      }
      if (!(BREFS && CHAIN) && __builtin_expect(__any(redo), 0)) {
        // rows the lean scan handed over continue with the generic scan in the round they stopped in
        // (the matches reported so far are final)
        if (redo) {
          if (WIDE) vm.template scan<cstd::Tdfa::K_REPLACE, decltype(rec)&, cstd::kMaxSlotsWide>(a.maxrepl, rec, resume, nm);
          else vm.template scan<cstd::Tdfa::K_REPLACE>(a.maxrepl, rec, resume, nm);
        }
      }
      if (INPLACE && live) {
        if (pend >= 0) put_repl_at(pend);
        if (nm > 0) {
          if (wr != copied) cstile::lds_copy(lds_in, pi + wr, lds_in, pi + copied, n - copied);
          out_len = wr + (n - copied);
        }
      }
    }
    CS_PHASE_MARK(1);
    if (kHoles && hole) out_len = hole_bytes;
    const int incl = csdev::wave_inclusive_scan(out_len);
    const int lo = incl - out_len;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    m_span = max(m_span, total);
    // (error bit 1: a roomier launch, which also takes rows with many matches, can succeed)
    bool grew = false;
    if (!INPLACE) grew = !bad && (total + 32 > a.cap_out || (!RESCAN && __any(nm > kMaxRec)));
    bad |= total + 32 > a.cap_out;
    if (!INPLACE) bad |= grew;
    if (oversize) {
      cstile::lookback_publish(a.status, tile, total, lane);
      if (p_tile >= 0) finish_pending(p_first);
      p_tile = -1;
      // this sub-tile is finished at once: its prefix, the offsets, every row written by its lane
      long long gb = __builtin_expect(scanner, 1) ? cstile::prefix_wait(a.excl, tile, cstile::status_load(a.excl + tile), a.error, lane)
                             : cstile::lookback_end(a.status, tile, total, cstile::lookback_poll(a.status, tile, lane), lane);
      if (gb < 0) {
        if (lane == 0) atomicOr(a.error, 1u | 8u);
        gb = 0;
      }
      if (lane < nrows) a.out_off[r0 + lane] = gb + lo;
      if (lane == nrows - 1 && r0 + nrows == in.rows) a.out_off[in.rows] = gb + lo + out_len;
      if (gb + total > a.out_cap) {
        if (lane == 0) atomicOr(a.error, 2u);
        return;
      }
      if (live) {
        const uint8_t* p = in.chars + (g0 + rbeg);
        uint8_t* o = a.out_chars + gb + lo;
        int copied = 0;
        auto piece = [&](int mb, int me, int reps) {
          for (int i = copied; i < mb; ++i) *o++ = p[i];
          for (int k = 0; k < reps; ++k)
            for (int i = 0; i < rb; ++i) *o++ = a.repl[i];
          copied = me;
        };
        cstd::Tdfa vg(D, P, p, n);
        if (WIDE) vg.template scan<cstd::Tdfa::K_REPLACE, decltype(piece)&, cstd::kMaxSlotsWide>(a.maxrepl, piece, 0, 0);
        else vg.template scan<cstd::Tdfa::K_REPLACE>(a.maxrepl, piece, 0, 0);
        for (int i = copied; i < n; ++i) *o++ = p[i];
      }
    } else if (__builtin_expect(bad, 0)) {
      // the host discards this launch's output; publish something so successors do not spin
      if (lane == 0) {
        atomicOr(a.error, !INPLACE && grew ? 2u : (1u | 4u));  // (4: a sub-tile beyond the staging capacity)
        cstile::status_store(a.status + tile, cstile::kFlagInc);
      }
      if (p_tile >= 0) finish_pending(p_first);
      p_tile = -1;
    } else {
      if (!(CS_DBG(a) & 8)) cstile::lookback_publish(a.status, tile, total, lane);
      if (lane == 0) CS_TILE_TRACE(tile, 0);
      CS_PHASE_MARK(2);
      if (p_tile >= 0)
        finish_pending((CS_DBG(a) & 64) ? (__builtin_expect(scanner, 1) ? cstile::status_load(a.excl + p_tile) : cstile::lookback_poll(a.status, p_tile, lane))
                                      : (has_second && (p_first >> 62) == 0 ? p_second : p_first));
      CS_PHASE_MARK(3);
      if (INPLACE) {
        if (live && !(CS_DBG(a) & 2)) cstile::lds_copy_ov(lds_out, lo, lds_in, lead + rbeg, out_len);
      } else if (live && !(CS_DBG(a) & 2)) {
        int oi = lo;                 // byte index into lds_out
        const int pi = lead + rbeg;  // byte index of the row in lds_in
        int copied = 0;
        if (BREFS && from_masks) {
          int mi = 0;
          while (cstd::u128_any(uS)) {
            const int mb = cstd::u128_ctz(uS), me = cstd::u128_ctz(uE) + 1;
            uS = cstd::u128_clear_lowest(uS);
            uE = cstd::u128_clear_lowest(uE);
            cstile::lds_copy_ov(lds_out, oi, lds_in, pi + copied, mb - copied);
            oi += mb - copied;
            int gb[cstd::Tdfa::kGroupBatch], ge[cstd::Tdfa::kGroupBatch], mend = me;
            bool ok;
            if (CHAIN && dense_tile) {  // (wave-uniform) no record of this match: its groups off the row's masks again
              uint32_t r0, r1, r2, x0 = 0, x1 = 0, x2 = 0;
              cstile::row_bits96(bitmap, pi, n, r0, r1, r2);
              if (unit_x != 0) cstile::row_bits96(xbitmap, pi, n, x0, x1, x2);
              int g4b[4], g4e[4];
              cstd::chain_group_bounds(cstd::u128(r0 | ((unsigned long long)r1 << 32), r2), cstd::u128(x0 | ((unsigned long long)x1 << 32), x2), D.chain, D.img,
                                       (uint32_t)D.img[D.img[15] - 1], mb, g4b, g4e);
#pragma unroll
              for (int q = 0; q < cstd::Tdfa::kGroupBatch; ++q) gb[q] = q < 4 ? g4b[q] : -1, ge[q] = q < 4 ? g4e[q] : -1;
              ok = true;
            } else {
              const uint32_t* rec = mrec + (mslot + mi) * 3;
              const uint32_t lo4 = rec[0], hi4 = rec[1], e4 = rec[2];
              auto un = [](uint32_t v) { return v == 255u ? -1 : (int)v; };
              gb[0] = un(lo4 & 255u), ge[0] = un((lo4 >> 8) & 255u), gb[1] = un((lo4 >> 16) & 255u), ge[1] = un(lo4 >> 24);
              gb[2] = un(hi4 & 255u), ge[2] = un((hi4 >> 8) & 255u), gb[3] = un((hi4 >> 16) & 255u), ge[3] = un(hi4 >> 24);
              mend = (int)(e4 & 255u);
              ok = ((e4 >> 8) & 1u) != 0;
            }
            int il = 0;
            for (int j = 0; j < T.nrefs; ++j) {
              cstile::lds_copy_ov(lds_out, oi, ttext, il, T.pos(j) - il);
              oi += T.pos(j) - il;
              il = T.pos(j);
              const int g = T.idx(j);
              int x = -1, y = -1;
              if (g == 0) {
                x = mb;
                y = mend;
              }
#pragma unroll
              for (int q = 0; q < cstd::Tdfa::kGroupBatch; ++q)
                if (g == q + 1 && g <= T.groups) {
                  x = gb[q];
                  y = ge[q];
                }
              if (ok && x >= 0 && y > x) {
                cstile::lds_copy_ov(lds_out, oi, lds_in, pi + x, y - x);
                oi += y - x;
              }
            }
            cstile::lds_copy_ov(lds_out, oi, ttext, il, T.bytes - il);
            oi += T.bytes - il;
            copied = me;
            ++mi;
          }
        } else if (UNITS && from_masks) {
          while (cstd::u128_any(uS)) {
            const int mb = cstd::u128_ctz(uS), me = cstd::u128_ctz(uE) + 1;
            uS = cstd::u128_clear_lowest(uS);
            uE = cstd::u128_clear_lowest(uE);
            cstile::lds_copy_ov(lds_out, oi, lds_in, pi + copied, mb - copied);
            oi += mb - copied;
            if (REP16) {
              if (rb > 16) {
                for (int i = 0; i < rb; ++i) lds_out[oi + i] = a.repl[i];
              } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                  if (i < rb) lds_out[oi + i] = (uint8_t)(rep[i >> 2] >> (8 * (i & 3)));
              }
              oi += rb;
            } else {
              for (int i = 0; i < rb; ++i) lds_out[oi++] = (uint8_t)((i < 4 ? rep[0] >> (8 * i) : rep[1] >> (8 * (i - 4))));
            }
            copied = me;
          }
        } else if (!UNITS && (!RESCAN || nm <= kMaxRec)) {
#pragma unroll
          for (int j = 0; j < kMaxRec; ++j)
            if (j < nm) {
              cstile::lds_copy_ov(lds_out, oi, lds_in, pi + copied, rec_mb[j] - copied);
              oi += rec_mb[j] - copied;
              for (int k = 0; k < rec_reps[j]; ++k) {
                if (REP16) {
                  if (rb > 16) {
                    for (int i = 0; i < rb; ++i) lds_out[oi + i] = a.repl[i];
                  } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                      if (i < rb) lds_out[oi + i] = (uint8_t)(rep[i >> 2] >> (8 * (i & 3)));
                  }
                  oi += rb;
                } else {
                  for (int i = 0; i < rb; ++i) lds_out[oi++] = (uint8_t)((i < 4 ? rep[0] >> (8 * i) : rep[1] >> (8 * (i - 4))));
                }
              }
              copied = rec_me[j];
            }
        } else if (!(BREFS && CHAIN) && __builtin_expect(nm > 0, 0)) {
          // more matches than the registers keep (UNITS: any row measured by the generic scan): the row's size is
          // known from the first scan, so scan it again and assemble as the matches are reported
          auto piece2 = [&](int mb, int me, int reps) {
            cstile::lds_copy_ov(lds_out, oi, lds_in, pi + copied, mb - copied);
            oi += mb - copied;
            for (int k = 0; k < reps; ++k)
              for (int i = 0; i < rb; ++i) lds_out[oi++] = a.repl[i];
            copied = me;
          };
          cstd::Tdfa vm2(D, P, lds_in + pi, n, pi & 3);
          if (WIDE) vm2.template scan<cstd::Tdfa::K_REPLACE, decltype(piece2)&, cstd::kMaxSlotsWide>(a.maxrepl, piece2, 0, 0);
          else vm2.template scan<cstd::Tdfa::K_REPLACE>(a.maxrepl, piece2, 0, 0);
        }
        cstile::lds_copy_ov(lds_out, oi, lds_in, pi + copied, n - copied);
      }
      cstile::wave_lds_fence();
      CS_PHASE_MARK(4);
      p_tile = tile;
      p_total = total;
      p_lo = lo;
      p_len = out_len;
    }
    if (!has_next) break;
    tile = t_nxt;
    t_nxt = t_nn;
    nxt = nn;
    pending = pending_new;
    c_hm = n_hm;
    c_hf = n_hf;
  }
  if (p_tile >= 0) finish_pending((CS_DBG(a) & 8) ? 0 : (__builtin_expect(scanner, 1) ? cstile::status_load(a.excl + p_tile) : cstile::lookback_poll(a.status, p_tile, lane)));
  {
    // (only a wave that would raise the maximum issues the same-address atomic; slot 14 behind the error word)
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(a.error) + 14;
    if (lane == 0 && (unsigned long long)m_span > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, (unsigned long long)m_span);
  }
#if defined(CS_PHASE_PROF)
  CS_PHASE_MARK(5);
  if (lane == 0)
    for (int k = 0; k < 12; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(a.error) + 1 + k, phase_acc[k]);
  if (lane == 0)
    for (int k = 0; k < 3; ++k) atomicAdd(&cstile::g_lb_stats[k], lb_acc[k]);
#endif
}




// ---- persistent contains_re / count_re over row tiles -----------------------------------
// Same staging as the replace kernel (contiguous tile runs per wave, next tile's chars in
// flight), no output assembly: LDS holds the DFA table and one input tile per wave, so seven
// workgroups fit a CU.  MODE 0 contains_re, 2 count_re, 3 findall spans (begins / lens [k * rows + row]),
// 4 extract spans (leftmost match, then Tdfa::group_find per capture group, all on the staged row),
// 5 / 6 replace_with_backrefs sizes / bytes (csvm::row_backrefs on the staged row; the bytes go out per row lane).
struct ScanStreamArgs {
  ColView in;
  const uint8_t* flags;
  TLaunch L;
  uint8_t* out8;
  int32_t* out32;
  unsigned long long* found;
  long long nsub;
  int cap_in, tbl_bytes;
  int rows_per_tile;  // LONG variant: 64, 32 or 16
  int32_t* begins;    // MODE 3, 4
  int32_t* lens;
  int ncols;
  int* maxp;             // MODE 3 (optional): receives the largest match count of a row
  // MODE 3, 4 (optional): the spans as ONE word per (column, row) -- begin << 16 | length, 0xFFFFFFFF = null -- instead of
  // begins / lens (half the span traffic; rows of a tile are far shorter than 64 KiB), and the bytes every 64-row tile
  // gives to each column, [column][tile] (rows_per_tile == 64 only): k_spans_write_tile2 sizes the columns from those
  uint32_t* spans;
  int32_t* tile_tot;
  const int32_t* gtags;  // MODE 4, 5, 6: capture-group tag image
  int gt_off, gt_words;  // when gt_words > 0 the image is staged into LDS at byte offset gt_off (inside tbl_bytes)
  const csvm::BackrefTemplate* tmpl;  // MODE 5, 6 (device memory: indexed per reference, must not live in the kernel arguments)
  const int64_t* out_off;      // MODE 6
  uint8_t* out_chars;
  // BITS: the bit-parallel form of the pattern (regex_bits.h), staged at byte offset bits_off of the LDS (inside tbl_bytes)
  const int32_t* bits;
  int bits_off, bits_words, bits_k;
  // MODE 0 / 2, 64-row tiles (optional): rows that hold a byte >= 0x80 or a NUL are not scanned here -- their indices go to
  // this list (k_tdfa_scan_list scans them afterwards, a thread a row), and the tile's other rows keep the form the launch was
  // chosen for (bit form, chain arithmetic, unit scan, lean scan) instead of the whole sub-tile going to the row-by-row scan
  int32_t* deferred;
  unsigned* ndeferred;
  unsigned deferred_cap;
};
// UNITS (MODE 0 and 2, !LONG): the unit scan of the replace kernel -- units queued by the row lanes, one unit per lane
// whatever its row, the per-row result summed (count_re) / OR-ed (contains_re) in LDS.
// CHAIN (a UNITS form, count_re / findall): the pattern is a chain and the column's sample holds no byte >= 0x80 -- the unit
// and lean scans are compiled out, a sub-tile the chain arithmetic does not take is scanned row by row by the generic executor.
// BITS (a CHAIN form, contains_re / count_re): the pattern has a bit-parallel form (regex_bits.h) and the column's sample holds
// no byte >= 0x80 -- every staged byte is classified into one bitmap per character class by table lookup, the row lanes
// read their rows' class masks and derive the matches by mask arithmetic; no automaton on a plain-ASCII sub-tile.
template <int MODE, bool IN_LDS, bool LONG = false, bool UNITS = false, bool CHAIN = false, bool BITS = false>
__global__ void __launch_bounds__(256, MODE == 4 ? ((CHAIN && !IN_LDS) ? 4 : 3) : CHAIN ? 4 : (UNITS && MODE == 3) ? 3 : (UNITS || MODE == 3) ? 4 : (MODE <= 1 && !LONG) ? 5 : 1) k_tdfa_scan_stream(ScanStreamArgs a) {
  static_assert(!UNITS || ((MODE == 0 || MODE == 2 || MODE == 3 || (MODE == 4 && CHAIN)) && !LONG), "unit scan: contains_re / count_re / findall on rows within the 96-byte masks");
  static_assert(!CHAIN || (UNITS && (MODE == 2 || MODE == 3 || MODE == 4 || (BITS && MODE == 0))), "the chain form: count_re / findall / extract");
  static_assert(!BITS || (CHAIN && (MODE == 0 || MODE == 2)), "the bit form: contains_re / count_re");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint8_t* base = reinterpret_cast<uint8_t*>(smem);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (scalar: what derives from it stays in SGPRs)
  const int bm_bytes = (a.cap_in >> 3) + 32;
  const int nbm = BITS ? max(a.bits_k, 2) : 2;  // bitmaps per wave (BITS: one per character class)
  // (MODE 4 keeps a lane-private byte per step of a group run behind the bitmap: regex_tdfa.h, group_find_back)
  const int unit_bytes = UNITS ? (nbm - 1) * bm_bytes + kUnitQueue * 4 + 64 * 4 + 16 : (MODE == 4 ? cstd::Tdfa::kBackSteps * 64 + kMaxGroups * 4 : 0);  // x bitmap, unit queue, per-row results, bail word
  uint8_t* lds_in = base + a.tbl_bytes + (size_t)wv * (a.cap_in + 32 + bm_bytes + unit_bytes + (((MODE == 0 || MODE == 2) && !LONG && a.deferred) ? bm_bytes + 64 * 4 : 0));
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(lds_in + a.cap_in + 32);
  uint32_t* xbitmap = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(bitmap) + bm_bytes);
  uint32_t* uqueue = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(bitmap) + (size_t)nbm * bm_bytes);
  uint32_t* rowres = uqueue + kUnitQueue;
  uint32_t* bailw = rowres + 64;
  constexpr bool kDefer = (MODE == 0 || MODE == 2) && !LONG;
  const bool deferring = kDefer && a.deferred != nullptr;  // (wave-uniform)
  uint32_t* oddmap = reinterpret_cast<uint32_t*>(lds_in + a.cap_in + 32 + bm_bytes + unit_bytes);  // (behind everything else of the wave)
  // the wave's rows put off and not yet in the list: one atomic on the list's counter per 64 of them at most (one per sub-tile
  // with such a row was 600K on one address for the C5 column's pieces -- 2 ms of a 4 ms kernel)
  int32_t* pend = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(oddmap) + bm_bytes);
  int npend = 0;  // (wave-uniform)
  auto flush_pend = [&] {
    cstile::wave_lds_fence();
    unsigned at = 0;
    if (lane == 0) at = atomicAdd(a.ndeferred, (unsigned)npend);
    at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
    if (lane < npend && at + (unsigned)lane < a.deferred_cap) a.deferred[at + lane] = pend[lane];
    cstile::wave_lds_fence();
    npend = 0;
  };
  // MODE 4: per group, the tile's bytes (the chain form keeps the "equals x" bitmap where the plain form has its history bytes)
  uint32_t* gtot = UNITS ? rowres : reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(xbitmap) + cstd::Tdfa::kBackSteps * 64);
  if (MODE == 4 && lane < kMaxGroups) gtot[lane] = 0;
  const TCtx c = tsetup<IN_LDS>(a.L, a.flags, smem);
  const cstd::View& D = c.D;
  const csvm::ProgView& P = c.P;
  const ColView& in = a.in;
  if (MODE >= 4 && a.gt_words > 0) {  // the group-tag tables are read once per DFA step: keep them next to the DFA table
    int32_t* g = reinterpret_cast<int32_t*>(base + a.gt_off);
    for (int i = threadIdx.x; i < a.gt_words; i += blockDim.x) g[i] = a.gtags[i];
    __syncthreads();
    a.gtags = g;
  }
  const uint32_t* spread = nullptr;
  csbits::View BV{};
  if (BITS) {
    int32_t* bl = reinterpret_cast<int32_t*>(base + a.bits_off);
    spread = bits_stage(a.bits, a.bits_words, bl);
    BV = csbits::make_view(bl);
  }
  const bool has_r2 = (((uint32_t)D.img[30] & 255u) <= (((uint32_t)D.img[30] >> 8) & 255u));
  const uint32_t unit_x = (D.units >> 8) & 127u, unit_xpat = unit_x * 0x01010101u;
  const int R = LONG ? a.rows_per_tile : 64;
  const long long waves = (long long)gridDim.x * 4;
  const long long per = (a.nsub + waves - 1) / waves;
  long long tile = ((long long)blockIdx.x * 4 + wv) * per;
  const long long tile_end = min(a.nsub, tile + per);
  if (tile >= tile_end) return;
  cstile::TileOffs cur = cstile::load_tile_offsets_r(in.offsets, in.rows, tile, R, lane);
  cstile::TileOffs nxt = cur;
  if (tile + 1 < tile_end) nxt = cstile::load_tile_offsets_r(in.offsets, in.rows, tile + 1, R, lane);
  cstile::TileChars pf;
#pragma unroll
  for (int j = 0; j < cstile::kPfChunks; ++j) pf.v[j] = make_uint4(0, 0, 0, 0);
  cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
  int hits = 0;
  for (;;) {
    const long long r0 = tile * R;
    const int nrows = (int)min((long long)R, in.rows - r0);
    const long long g0 = cstile::rl64(cur.o0, 0), g1 = cstile::rl64(cur.o1, 63);
    const bool live0 = lane < nrows && row_is_valid(in.validity, r0 + lane);
    const int rbeg = (int)(cur.o0 - g0);
    const int n0 = live0 ? (int)(cur.o1 - cur.o0) : 0;
    const int lead = (int)((uintptr_t)(in.chars + g0) & 15);
    const int want = (int)(g1 - g0) + lead;
    cstile::stage_chars(lds_in, want, lane, pf);
    uint32_t odd = 0;
#pragma unroll
    for (int j = 0; j < cstile::kPfChunks; ++j)
      if (j * 1024 + lane * 16 < want) {
        const uint4 q = pf.v[j];
        const uint32_t ox = q.x | ((q.x - 0x01010101u) & ~q.x), oy = q.y | ((q.y - 0x01010101u) & ~q.y);
        const uint32_t oz = q.z | ((q.z - 0x01010101u) & ~q.z), ow = q.w | ((q.w - 0x01010101u) & ~q.w);
        odd |= ox | oy | oz | ow;
        if (deferring) cstile::put_bits16(oddmap, j * 1024 + lane * 16, cstile::gather16_bit7(ox & 0x80808080u, oy & 0x80808080u, oz & 0x80808080u, ow & 0x80808080u));
        if (BITS) {
          uint32_t pair[4];
          bits_classify16(spread, q, pair);
#pragma unroll
          for (int k = 0; k < csbits::kMaxClasses; ++k)
            if (k < BV.K) cstile::put_bits16(bitmap + k * (bm_bytes >> 2), j * 1024 + lane * 16, (pair[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
          continue;
        }
        uint32_t bits;
        if (has_r2)
          bits = cstile::gather16_bit7(cstd::Tdfa::cand_bits_ascii<true>(D, q.x), cstd::Tdfa::cand_bits_ascii<true>(D, q.y), cstd::Tdfa::cand_bits_ascii<true>(D, q.z),
                                       cstd::Tdfa::cand_bits_ascii<true>(D, q.w));
        else
          bits = cstile::gather16_bit7(cstd::Tdfa::cand_bits_ascii<false>(D, q.x), cstd::Tdfa::cand_bits_ascii<false>(D, q.y), cstd::Tdfa::cand_bits_ascii<false>(D, q.z),
                                       cstd::Tdfa::cand_bits_ascii<false>(D, q.w));
        cstile::put_bits16(bitmap, j * 1024 + lane * 16, bits);
        if (UNITS && unit_x != 0) cstile::put_bits16(xbitmap, j * 1024 + lane * 16, unit_xbits16(q, unit_xpat));
      }
    const bool has_next = tile + 1 < tile_end;
    // (the offsets two tiles ahead are fetched unconditionally, at a clamped index, and become `nxt` at the bottom of
    // the iteration: see the note in k_tdfa_replace_stream)
    const cstile::TileOffs nn = cstile::load_tile_offsets_r(in.offsets, in.rows, tile + 2 < tile_end ? tile + 2 : tile_end - 1, R, lane);
    if (has_next) {
      cur = nxt;
      cstile::issue_chars(in.chars, cstile::rl64(cur.o0, 0), cstile::rl64(cur.o1, 63), lane, pf);
    }
    cstile::wave_lds_fence();
    bool has_odd = __any((odd & 0x80808080u) != 0);
    bool live = live0;
    if (deferring && has_odd) {  // (wave-uniform)
      uint32_t d0, d1, d2;
      cstile::row_bits96(oddmap, lead + rbeg, n0 < 96 ? n0 : 96, d0, d1, d2);
      // (a row beyond the masks is not looked at: the tile is scanned row by row as before)
      const bool fits = !__any(live0 && n0 + ((lead + rbeg) & 3) > cstd::Tdfa::kMaskBytes);
      const bool put_off = fits && live0 && (d0 | d1 | d2) != 0;
      const unsigned long long who = __ballot(put_off);
      if (who) {
        const int k = __builtin_popcountll(who);
        if (npend + k > 64) flush_pend();
        if (put_off) pend[npend + __builtin_popcountll(who & ((1ull << lane) - 1ull))] = (int32_t)(r0 + lane);
        npend += k;
      }
      if (fits) {
        live = live0 && !put_off;
        has_odd = false;  // (what is left of the tile is plain)
        odd = 0;
      }
    }
    const int n = live ? n0 : 0;
    int v = 0;
    if (MODE == 5 || MODE == 6) {
      const uint8_t* p = lds_in + lead + rbeg;
      cstd::Tdfa vm(D, P, p, n, (lead + rbeg) & 3);
      auto find = [&](auto&& f) { csvm::walk_matches(vm, f); };
      // the groups of a match come four at a time from ONE anchored run (regex_tdfa.h: group_find_all) and are kept
      // until the walk moves to the next match: a template with four references costs one run, not four
      int c_mb = -1, c_batch = -1, c_ok = 0, c_end = 0;
      int c_gb[cstd::Tdfa::kGroupBatch], c_ge[cstd::Tdfa::kGroupBatch];
      const int tgroups = a.tmpl->groups;
      auto group = [&](int mb, int g, int& x, int& y) -> bool {
        if (n >= 255 || !a.gtags) return g == 0 ? vm.find(mb, mb + 1, x, y) > 0 : vm.group_find(mb, a.gtags, g, x, y) > 0;
        const int batch = g == 0 ? (c_mb == mb ? c_batch : 0) : (g - 1) / cstd::Tdfa::kGroupBatch;
        if (c_mb != mb || c_batch != batch) {
          const int first = batch * cstd::Tdfa::kGroupBatch + 1;
          int cnt = tgroups - first + 1;
          cnt = cnt < 0 ? 0 : (cnt > cstd::Tdfa::kGroupBatch ? cstd::Tdfa::kGroupBatch : cnt);
          c_ok = vm.group_find_all(mb, a.gtags, first, cnt, c_gb, c_ge, c_end);
          c_mb = mb;
          c_batch = batch;
        }
        if (g == 0) {
          x = mb;
          y = c_end;
        } else {
          const int k = (g - 1) % cstd::Tdfa::kGroupBatch;
#pragma unroll
          for (int i = 0; i < cstd::Tdfa::kGroupBatch; ++i)
            if (i == k) {
              x = c_gb[i];
              y = c_ge[i];
            }
        }
        return c_ok > 0;
      };
      if (MODE == 5) {
        int len = -1;
        if (live) {
          len = 0;
          csvm::row_backrefs(p, n, *a.tmpl, find, group, [&](const uint8_t*, int k) { len += k; });
        }
        if (lane < nrows) a.out32[r0 + lane] = len;
      } else if (live) {
        uint8_t* o = a.out_chars + a.out_off[r0 + lane];
        csvm::row_backrefs(p, n, *a.tmpl, find, group, [&](const uint8_t* q, int k) {
          for (int i = 0; i < k; ++i) *o++ = q[i];
        });
      }
    } else if (MODE == 1) {
      // match (count.cu:113-165): the anchored run on the staged row -- the rows arrive through the same coalesced
      // tiles as contains_re's instead of a thread per row reading its bytes from HBM
      cstd::Tdfa vm(D, P, lds_in + lead + rbeg, n, (lead + rbeg) & 3);
      v = live ? csvm::row_contains_re(vm, true) : 0;
    } else if (MODE >= 7) {
      // programs of five to eight live threads (MODE 7 contains_re, 8 match, 9 count_re): the generic executor with
      // eight start offsets on the staged row (regex_tdfa.h: TdfaWide)
      cstd::TdfaWide vm(D, P, lds_in + lead + rbeg, n, (lead + rbeg) & 3);
      v = !live ? 0 : (MODE == 9 ? csvm::row_count_re(vm) : csvm::row_contains_re(vm, MODE == 8));
    } else if (MODE == 4) {
      cstd::Tdfa vm(D, P, lds_in + lead + rbeg, n, (lead + rbeg) & 3);
      int mb = 0, me = 0;
      bool chain_done = false;
      if (CHAIN) {
        // extract on a chain pattern whose groups are runs of items (regex_tdfa.h: chain_match, chain_group_bounds) on a
        // sub-tile of plain ASCII: the row's first match and its group ranges from the row's two masks, no table walk
        const bool plain = D.nskip > 0 && D.img[12] <= 4 && !__any((odd & 0x80808080u) != 0) && !__any(live && !vm.masks_fit());
        if (plain && (D.chain >> 16) && (((uint32_t)D.img[30] >> 20) & 1u) && a.ncols <= 4 && a.spans) {  // (wave-uniform)
          using namespace cstd;
          uint32_t r0w, r1w, r2w, x0w = 0, x1w = 0, x2w = 0;
          cstile::row_bits96(bitmap, lead + rbeg, n, r0w, r1w, r2w);
          if (unit_x != 0) cstile::row_bits96(xbitmap, lead + rbeg, n, x0w, x1w, x2w);
          const U128 Rm = u128(r0w | ((unsigned long long)r1w << 32), r2w), Xm = u128(x0w | ((unsigned long long)x1w << 32), x2w);
          U128 S = u128(0, 0), E = u128(0, 0);
          if (live) chain_match96(r0w, r1w, r2w, x0w, x1w, x2w, D.chain, D.img, S, E, D.sfx, n, [&](int i) { return lds_in[lead + rbeg + i]; });
          const bool hit = live && u128_any(S);
          v = hit;
          int gb[4], ge[4];
          chain_group_bounds(Rm, Xm, D.chain, D.img, (uint32_t)D.img[D.img[15] - 1], hit ? u128_ctz(S) : 0, gb, ge);
          if (lane < nrows) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (k < a.ncols) {
                const bool ok = hit && gb[k] >= 0 && ge[k] > gb[k];
                a.spans[(long long)k * in.rows + r0 + lane] = ok ? (((uint32_t)gb[k] << 16) | (uint32_t)(ge[k] - gb[k])) : 0xFFFFFFFFu;
                if (a.tile_tot && ok) __hip_atomic_fetch_add(gtot + k, (uint32_t)(ge[k] - gb[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              }
          }
          chain_done = true;
        }
      }
      if (__builtin_expect(!chain_done, !CHAIN)) {  // (the chain form: a cold path)
      // leftmost match: the lean scan on ASCII tiles of short rows (as contains_re), else the generic find
      const bool lean = !LONG && D.nskip > 0 && D.img[12] <= 4 && !__any((odd & 0x80808080u) != 0) && !__any(live && !vm.masks_fit());
      int f = -1;
      if (lean && live) {
        uint32_t m0, m1, m2;
        cstile::row_bits96(bitmap, lead + rbeg, n, m0, m1, m2);
        f = vm.scan_lean_first(m0, m1, m2, [&](int b0, int e0, int) {
          mb = b0;
          me = e0;
        });
      }
      if (live && f < 0) f = vm.find(0, n, mb, me);
      const bool hit = live && f > 0;
      v = hit;
      if (lane < nrows) {
        if (n < 255) {  // every group of the match from one anchored run, four at a time (regex_tdfa.h: group_find_all)
          for (int g0 = 0; g0 < a.ncols; g0 += cstd::Tdfa::kGroupBatch) {
            int gb[cstd::Tdfa::kGroupBatch], ge[cstd::Tdfa::kGroupBatch], mend = 0;
            const int cnt = min(cstd::Tdfa::kGroupBatch, a.ncols - g0);
            // (backwards from the match where that applies -- short ASCII matches: regex_tdfa.h -- else the forward run)
            int got = -1;
            if (!UNITS && hit) got = vm.group_find_back(mb, a.gtags, g0 + 1, cnt, gb, ge, mend, cstd::Tdfa::HistBytes{reinterpret_cast<uint8_t*>(xbitmap) + lane, 64});
            if (hit && got < 0) got = vm.group_find_all(mb, a.gtags, g0 + 1, cnt, gb, ge, mend);
            const bool found = hit && got > 0;
#pragma unroll
            for (int k = 0; k < cstd::Tdfa::kGroupBatch; ++k)
              if (k < cnt) {
                const bool ok = found && gb[k] >= 0 && ge[k] > gb[k];
                if (a.spans) {
                  a.spans[(long long)(g0 + k) * in.rows + r0 + lane] = ok ? (((uint32_t)gb[k] << 16) | (uint32_t)(ge[k] - gb[k])) : 0xFFFFFFFFu;
                } else {
                  a.begins[(long long)(g0 + k) * in.rows + r0 + lane] = ok ? gb[k] : 0;
                  a.lens[(long long)(g0 + k) * in.rows + r0 + lane] = ok ? ge[k] - gb[k] : -1;
                }
                if (a.tile_tot && ok) __hip_atomic_fetch_add(gtot + g0 + k, (uint32_t)(ge[k] - gb[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              }
          }
        } else {
          for (int g = 0; g < a.ncols; ++g) {
            int x = -1, y = -1;
            const bool ok = hit && vm.group_find(mb, a.gtags, g + 1, x, y) && x >= 0 && y > x;
            if (a.tile_tot && ok) __hip_atomic_fetch_add(gtot + g, (uint32_t)(y - x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (a.spans) {
              a.spans[(long long)g * in.rows + r0 + lane] = ok ? (((uint32_t)x << 16) | (uint32_t)(y - x)) : 0xFFFFFFFFu;
            } else {
              a.begins[(long long)g * in.rows + r0 + lane] = ok ? x : 0;
              a.lens[(long long)g * in.rows + r0 + lane] = ok ? y - x : -1;
            }
          }
        }
      }
      }  // (!chain_done)
      if (a.tile_tot) {  // the bytes this tile gives to each group's column (summed in LDS by the row lanes above)
        cstile::wave_lds_fence();
        if (lane < a.ncols) {
          a.tile_tot[(long long)lane * a.nsub + tile] = (int32_t)gtot[lane];
          gtot[lane] = 0;
        }
        cstile::wave_lds_fence();
      }
    } else {
      cstd::Tdfa vm(D, P, lds_in + lead + rbeg, n, (lead + rbeg) & 3);
      bool hi_units = false;  // (bytes >= 0x80 that can only kill: the unit route alone -- reclassify_high)
      if (!CHAIN && UNITS && (MODE == 0 || MODE == 2 || MODE == 3) && has_odd && ((D.units >> 17) & 3u) && (D.units & 1u))
        hi_units = !reclassify_high(D, has_r2, lds_in, want, lane, bitmap, a.flags, lds_in + lead + rbeg, n);
      const bool lean = D.nskip > 0 && D.img[12] <= 4 && (!has_odd || hi_units) &&
                        !__any(live && (LONG ? n > cstd::Tdfa::kLongBytes : !vm.masks_fit()));
      bool redo = live && !lean;
      int k = 0;  // MODE 3: matches reported so far
      int cl[4] = {0, 0, 0, 0};  // (packed spans: the lengths of the row's first four matches, for the tile's column totals)
      auto span = [&](int mb, int me, int) {
        if (k < a.ncols) {
          if (a.spans) {
            a.spans[(long long)k * in.rows + r0 + lane] = ((uint32_t)mb << 16) | (uint32_t)(me - mb);
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c == k) cl[c] = me - mb;
          } else {
            a.begins[(long long)k * in.rows + r0 + lane] = mb;
            a.lens[(long long)k * in.rows + r0 + lane] = me - mb;
          }
        }
        ++k;
      };
      bool units_done = false;
      if (UNITS && MODE == 3) {
        // findall: the units leave each match's first and last byte in the two bitmaps (as in the replace kernel);
        // the row lanes read their matches back in order
        if (lean && !has_odd && (D.chain >> 16)) {  // (wave-uniform) a chain pattern on plain ASCII: regex_tdfa.h, chain_match
          using namespace cstd;
          uint32_t r0, r1, r2, x0 = 0, x1 = 0, x2 = 0;
          cstile::row_bits96(bitmap, lead + rbeg, n, r0, r1, r2);
          if (unit_x != 0) cstile::row_bits96(xbitmap, lead + rbeg, n, x0, x1, x2);
          if (live) {
            U128 S, E;
            chain_match96(r0, r1, r2, x0, x1, x2, D.chain, D.img, S, E, D.sfx, n, [&](int i) { return lds_in[lead + rbeg + i]; });
            while (u128_any(S)) {
              const int mb = u128_ctz(S), me = u128_ctz(E) + 1;
              S = u128_clear_lowest(S);
              E = u128_clear_lowest(E);
              span(mb, me, 0);
            }
            v = k;
          }
          redo = false;
          units_done = true;
        } else if (!CHAIN && lean && (D.units & 1u)) {  // (wave-uniform)
          using namespace cstd;
          uint32_t m0, m1, m2;
          const int total_units = unit_discover(D, bitmap, xbitmap, lead + rbeg, n, lane, uqueue, m0, m1, m2, [&] {
            for (int i = lane * 16; i < bm_bytes; i += 64 * 16) {
              *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(bitmap) + i) = make_uint4(0, 0, 0, 0);
              *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(xbitmap) + i) = make_uint4(0, 0, 0, 0);
            }
            if (lane < 2) bailw[lane] = 0;
          });
          if (total_units >= 0) {
            for (int u0 = 0; u0 < total_units; u0 += 64) {
              const bool act = u0 + lane < total_units;
              int r, rbeg_r, n_r;
              uint32_t c0, c1, c2;
              unit_take(act ? uqueue[u0 + lane] : 0u, rbeg, n, m0, m1, m2, r, rbeg_r, n_r, c0, c1, c2, hi_units);
              if (act) {
                const int pu = lead + rbeg_r;
                cstd::Tdfa vu(D, P, lds_in + pu, n_r, pu & 3);
                bool ubail = false;
                auto recu = [&](int mb, int me, int) {
                  const int ps = pu + mb, pe = pu + me - 1;
                  __hip_atomic_fetch_or(bitmap + (ps >> 5), 1u << (ps & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  __hip_atomic_fetch_or(xbitmap + (pe >> 5), 1u << (pe & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                };
                vu.scan_lean_dispatch(-1, c0, c1, c2, recu, ubail);
                if (ubail) __hip_atomic_fetch_or(bailw + (r >> 5), 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              }
            }
            cstile::wave_lds_fence();
            uint32_t s0, s1, s2, e0, e1, e2;
            cstile::row_bits96(bitmap, lead + rbeg, n, s0, s1, s2);
            cstile::row_bits96(xbitmap, lead + rbeg, n, e0, e1, e2);
            const bool rbail = ((bailw[lane >> 5] >> (lane & 31)) & 1u) != 0;
            if (live && !rbail) {
              U128 S = u128(s0 | ((unsigned long long)s1 << 32), s2), E = u128(e0 | ((unsigned long long)e1 << 32), e2);
              while (u128_any(S)) {
                const int mb = u128_ctz(S), me = u128_ctz(E) + 1;
                S = u128_clear_lowest(S);
                E = u128_clear_lowest(E);
                span(mb, me, 0);
              }
              v = k;
            }
            redo = live && rbail;  // (such a row is scanned whole)
            units_done = true;
          }
        }
      } else if (UNITS) {
        // (contains_re stops at a row's first match: its ASCII tiles keep the row lanes' scan -- scanning every unit cost more
        // than the balance won --, the unit route serves its tiles with bytes >= 0x80, which the row lanes' scan cannot take)
        if (BITS && !has_odd && !__any(live && n > csbits::kMaxRowBytes)) {  // (wave-uniform) plain ASCII, rows within the masks
          using namespace cstd;
          auto cls = CS_BITS_CLS(bitmap, bm_bytes >> 2, lead + rbeg, n);
          auto raw = CS_BITS_RAW(bitmap, bm_bytes >> 2, lead + rbeg);
          if (MODE == 0) {
            v = live && csbits::contains(BV, cls, raw, n) ? 1 : 0;
          } else {
            U128 S = u128(0, 0), E;
            if (live) csbits::match(BV, cls, raw, n, S, E);
            v = u128_popc(S);
          }
          redo = false;
          units_done = true;
        } else if (!BITS && lean && !has_odd && (D.chain >> 16)) {  // (wave-uniform) a chain pattern on plain ASCII: regex_tdfa.h, chain_match
          using namespace cstd;
          uint32_t r0, r1, r2, x0 = 0, x1 = 0, x2 = 0;
          cstile::row_bits96(bitmap, lead + rbeg, n, r0, r1, r2);
          if (unit_x != 0) cstile::row_bits96(xbitmap, lead + rbeg, n, x0, x1, x2);
          const U128 R = u128(r0 | ((unsigned long long)r1 << 32), r2), X = u128(x0 | ((unsigned long long)x1 << 32), x2);
          if (MODE == 0 && !(D.chain & kChainLeadB)) {  // (a `\b` in front of the chain is checked per match: chain_match)
            v = live && u128_any(chain_suffix_filter(chain_ends(R, X, D.chain, D.img), D.chain, D.sfx, n, [&](int i) { return lds_in[lead + rbeg + i]; })) ? 1 : 0;
          } else if (MODE == 0) {
            U128 S = u128(0, 0), E;
            if (live) chain_match96(r0, r1, r2, x0, x1, x2, D.chain, D.img, S, E, D.sfx, n, [&](int i) { return lds_in[lead + rbeg + i]; });
            v = u128_any(S) ? 1 : 0;
          } else {
            U128 S = u128(0, 0), E;
            if (live) chain_match96(r0, r1, r2, x0, x1, x2, D.chain, D.img, S, E, D.sfx, n, [&](int i) { return lds_in[lead + rbeg + i]; });
            v = u128_popc(S);
          }
          redo = false;
          units_done = true;
        } else if (!CHAIN && lean && (D.units & 1u) && (MODE != 0 || hi_units)) {  // (wave-uniform)
          constexpr int KIND = MODE == 0 ? cstd::Tdfa::K_CONTAINS : cstd::Tdfa::K_COUNT;
          uint32_t m0, m1, m2;
          const int total_units = unit_discover(D, bitmap, xbitmap, lead + rbeg, n, lane, uqueue, m0, m1, m2, [&] {
            rowres[lane] = 0;
            if (lane < 2) bailw[lane] = 0;
          });
          if (total_units >= 0) {
            for (int u0 = 0; u0 < total_units; u0 += 64) {
              const bool act = u0 + lane < total_units;
              int r, rbeg_r, n_r;
              uint32_t c0, c1, c2;
              unit_take(act ? uqueue[u0 + lane] : 0u, rbeg, n, m0, m1, m2, r, rbeg_r, n_r, c0, c1, c2, hi_units);
              if (act) {
                const int pu = lead + rbeg_r;
                cstd::Tdfa vu(D, P, lds_in + pu, n_r, pu & 3);
                const int got = vu.template scan_lean_count<KIND>(c0, c1, c2);
                if (got < 0) __hip_atomic_fetch_or(bailw + (r >> 5), 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                else if (got > 0 && MODE == 0) __hip_atomic_fetch_or(rowres + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                else if (got > 0) __hip_atomic_fetch_add(rowres + r, (uint32_t)got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              }
            }
            cstile::wave_lds_fence();
            v = live ? (int)rowres[lane] : 0;
            redo = live && ((bailw[lane >> 5] >> (lane & 31)) & 1u) != 0;  // (such a row is scanned whole)
            units_done = true;
          }
        }
      }
      if (hi_units && !units_done) redo = live;  // (never the lean scan over whole rows of such a tile)
      if (CHAIN && !units_done) redo = live;       // (the generic scan)
      if (!CHAIN && !units_done && lean && !hi_units && live) {
        uint32_t m0, m1, m2;
        constexpr int KIND = MODE == 0 ? cstd::Tdfa::K_CONTAINS : cstd::Tdfa::K_COUNT;
        if (LONG) {
          const int p0 = lead + rbeg;
          cstile::row_bits96(bitmap, p0, min(n, 96), m0, m1, m2);
          auto refill = [&](int wb, uint32_t& x0, uint32_t& x1, uint32_t& x2) {
            cstile::row_bits96(bitmap, p0 + wb, min(n - wb, 96), x0, x1, x2);
          };
          if (MODE == 3) v = vm.scan_lean_spans_long(m0, m1, m2, span, refill);
          else v = vm.template scan_lean_count_long<KIND>(m0, m1, m2, refill);
        } else {
          cstile::row_bits96(bitmap, lead + rbeg, n, m0, m1, m2);
          if (MODE == 3) v = vm.scan_lean_spans(m0, m1, m2, span);
          else v = vm.template scan_lean_count<KIND>(m0, m1, m2);
        }
        redo = v < 0;
      }
      // (count_re / findall: a cold path, and saying so straightens the hot one -- findall 6.3 -> 5.7 ms; contains_re measured
      // slower with the hint, 2.09 -> 2.22, and keeps the plain branch)
      if ((MODE == 2 || MODE == 3) ? __builtin_expect(__any(redo), 0) : __any(redo)) {
        if (redo) {
          if (MODE == 3) {
            k = 0;  // (the spans reported so far are written again, identically)
#pragma unroll
            for (int c = 0; c < 4; ++c) cl[c] = 0;
            v = csvm::row_findall(vm, [&](int, int mb, int me) {
              span(mb, me, 1);
              return true;
            });
          } else {
            v = MODE == 2 ? csvm::row_count_re(vm) : csvm::row_contains_re(vm, false);
          }
        }
      }
      if (MODE == 3 && lane < nrows) {
        if (a.spans)
          for (int j = live ? k : 0; j < a.ncols; ++j) a.spans[(long long)j * in.rows + r0 + lane] = 0xFFFFFFFFu;
        else
          for (int j = live ? k : 0; j < a.ncols; ++j) a.lens[(long long)j * in.rows + r0 + lane] = -1;
      }
      if (MODE == 3 && a.tile_tot) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < a.ncols) {
            // (most tiles give the later columns nothing: no reduction then)
            const int t = __any(live && k > c) ? wave_reduce_sum(live ? cl[c] : 0) : 0;
            if (lane == 0) a.tile_tot[(long long)c * a.nsub + tile] = t;
          }
        // (columns beyond the fourth -- rows with many matches, `[a-z]+` on log lines: eighteen --: no registers kept their lengths;
        // every lane reads its own spans back, just written, and the wave adds them up.  The exact-width pass then leaves packed
        // spans and tile totals like the provisional one, where it used to write begins / lens for a scan over the lengths of
        // every column: findall([a-z]+) on the C3 column 31.8 ms)
        for (int c = 4; c < a.ncols; ++c) {
          int t = 0;
          if (__any(live && k > c)) {  // (wave-uniform)
            const uint32_t w = (live && k > c) ? a.spans[(long long)c * in.rows + r0 + lane] : 0u;
            t = wave_reduce_sum((int)(w & 0xFFFFu));
          }
          if (lane == 0) a.tile_tot[(long long)c * a.nsub + tile] = t;
        }
      }
      if (MODE == 3 && a.maxp) {
        int m = live ? k : 0;
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        // (only a wave that would raise the maximum issues the same-address atomic)
        if (lane == 0 && m > __hip_atomic_load(a.maxp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.maxp, m);
      }
    }
    if (lane < nrows) {
      if (MODE == 2 || MODE == 9) a.out32[r0 + lane] = v;
      else if (MODE == 0 || MODE == 1 || MODE == 7 || MODE == 8) a.out8[r0 + lane] = (uint8_t)v;
    }
    hits += v > 0;
    cstile::wave_lds_fence();  // the next tile overwrites lds_in
    if (!has_next) break;
    ++tile;
    nxt = nn;
  }
  if (kDefer && npend > 0) flush_pend();
  const int t = wave_reduce_sum(hits);
  if (lane == 0 && t) atomicAdd(a.found, (unsigned long long)t);
}

struct TPlan {
  TLaunch d;
  size_t lds_bytes;
  unsigned grid;
};
// the tagged DFA with at most four live threads: every DFA kernel (lean scans, unit decomposition, capture groups)
bool use_tdfa(const cs_regex* re) { return !re->tdfa.empty() && re->tdfa[12] <= cstd::kMaxSlots && !cs::cfg("CS_REGEX_NO_TDFA"); }
// ... with five to eight (counted repetitions): contains_re / match / count_re and replace_re run it on the generic executor
// with eight start offsets (regex_tdfa.h: TdfaWide); the other ops keep the list simulator for such programs
bool use_tdfa_wide(const cs_regex* re) { return !re->tdfa.empty() && re->tdfa[12] > cstd::kMaxSlots && !cs::cfg("CS_REGEX_NO_TDFA"); }
// The bit-parallel form (regex_bits.h) is taken when the pattern has one, is no chain (chains run on two SWAR-classified
// bitmaps, cheaper than the class-table lookup), the column's sample is plain ASCII, and the automaton's candidate bytes --
// the bytes at which its idle skip has to stop -- make up a good part of the column's sample: with few candidates the lean
// scans skip most of every row and win; with candidates everywhere (alternations of word-bounded literals, small sets in
// a `+` loop) they walk every byte in a dependent chain of table reads (VERDICT r4 missing #3: 8.9 / 21.4 ms on C3).
// Thresholds from tools/probe_bits.py on the C3 column (profiles/r05/probe_bits.txt): share of candidate bytes -> bits / automaton ms
//   contains_re: 86.6 % ([^ ]+) 2.15 / 1.82 (a match at once: the first-match scan wins), 11.9 % ([aeiou]+) 2.2 / 4.4, 7.2 % (the
//   gtest pattern) 3.9 / 8.9, 2.4 % with three live threads (cat|cot|cut) 3.3 / 5.3, 2.4-2.6 % with one 2.3-2.7 / 2.5-2.7, below 1 % 2.3-3.3 / 1.4-2.9
//   count_re:    from 2 % up the bit form wins or ties (1.9 / 41.6 at 11.9 %, 2.5 / 4.1 at 2.4 %), below 1 % it loses (2.2 / 1.6)
//   replace_re:  (two workgroups a CU against three) 7.3 / 21.4 at 7.2 %, 8.6 / 51.7 at 11.9 %, 6.4 / 9.3 at 2.4 % with three threads;
//   5.4-5.8 / 4.6-4.9 at 2.4 % with one thread
enum { BITS_CONTAINS = 0, BITS_COUNT = 2, BITS_REPLACE = 3 };
// the share of the column's sampled bytes at which the automaton's idle skip has to stop
double candidate_share(const cs_regex* re, const cs_column* col, hipStream_t s) {
  if (re->tdfa.empty()) return 1.0;
  const uint32_t* hist = sample_byte_hist(col, s);
  uint64_t all = 0, cand = 0;
  for (unsigned c = 0; c < 256; ++c) {
    all += hist[c];
    if (c < 128 && (((uint32_t)re->tdfa[21 + (c >> 5)] >> (c & 31)) & 1u)) cand += hist[c];
  }
  return all ? (double)cand / (double)all : 0.0;
}
// Rows with a byte >= 0x80 or a NUL, by the column's sample: are they few enough (at most about one row in thirty -- an upper
// bound: every lead byte and every NUL counted as a row of its own) for a scan to put them off (ScanStreamArgs::deferred) and run
// as on plain ASCII?  (Measured on the C2 column, one row in twenty: the bit form's ops gain -- gtest contains_re 1.08 -> 0.57 ms,
// replace_re 2.31 -> 1.56 --, but a pattern whose generic scan is slow pays more for its rows on the list, a thread each, than the
// sub-tiles cost row by row: `[a-z]+ing\b` replace_re 3.9 -> 5.6, `\d+` 0.72 -> 1.35.  At one row in 170 -- C5 -- everything gains.)
bool odd_rows_few(const cs_column* col, hipStream_t s) {
  if (cs::cfg("CS_NO_DEFERRED_ROWS") || col->rows == 0) return false;
  const uint32_t* hist = sample_byte_hist(col, s);
  uint64_t all = 0, oddb = hist[0];
  for (unsigned c = 0; c < 256; ++c) {
    all += hist[c];
    if (c >= 0xC0) oddb += hist[c];  // (one lead byte a character; a stray continuation byte has no lead in front of it -- rare enough not to matter for a hint)
  }
  if (all == 0) return false;
  return (double)oddb / (double)all * ((double)col->nbytes / (double)col->rows) <= cs::cfg_int("CS_ODD_ROWS_PERMILLE", 30) / 1000.0;
}
bool bits_route(const cs_regex* re, const cs_column* col, hipStream_t s, int op, bool high_ok = false) {
  if (re->bits.empty() || cs::cfg("CS_NO_BITS_FORM")) return false;
  if (!re->tdfa.empty() && ((re->tdfa[30] >> 16) & 15) != 0) return false;  // a chain
  if (sample_has_high_bytes(col, s) && !high_ok) return false;
  if (cs::cfg("CS_BITS_ALWAYS")) return true;
  if (re->tdfa.empty()) return true;  // (no automaton: the list simulator is the alternative)
  const uint32_t* hist = sample_byte_hist(col, s);
  uint64_t all = 0, cand = 0;
  for (unsigned c = 0; c < 128; ++c) {
    all += hist[c];
    if (((uint32_t)re->tdfa[21 + (c >> 5)] >> (c & 31)) & 1u) cand += hist[c];
  }
  if (cs::cfg("CS_STREAM_INFO"))
    fprintf(stderr, "bits route: candidates %.2f %% of the sample (%d classes, %d alternatives, %d states, %d threads)\n", all ? 100.0 * (double)cand / (double)all : 0.0,
            re->bits[1], re->bits[3], re->tdfa[1], re->tdfa[12]);
  if (all == 0) return false;
  const double f = (double)cand / (double)all;
  const bool threads3 = re->tdfa[12] >= 3;
  if (op == BITS_COUNT) return f >= 0.02;
  // (the C5 column's pieces, the gtest pattern, 4.8 %: 4.0 / 7.1 -- the line between the 2.4 % tie and 7.2 % moved from 5 % to 3.5 %)
  // (candidates everywhere and assertions behind the `+` loop -- `[^ ]+$` --: the first-match scan does not "find a match at once",
  // it fails at every start: 8.4 ms on the C3 column against the bit form's 2.3)
  const bool tail = (re->bits[2] & csbits::F_TAIL) != 0;
  if (op == BITS_CONTAINS) return (f >= 0.035 && (f <= 0.5 || tail)) || (threads3 && f >= 0.02 && f <= 0.5);
  // (a pattern whose shortest match is one byte matches at most of its candidates: the automaton's routes then pay per match --
  // replace_re('e') on the C3 column, 4 % candidates: units 7.65, the bit form 6.45 ms)
  return f >= 0.05 || ((threads3 || re->tdfa[13] == 1) && f >= 0.02);
}
void upload(cs_regex* re, hipStream_t s) {
  // a compiled pattern may be shared between host threads (and is kept in the process-wide pattern cache): the device
  // images are made once.  All three are built into locals and committed together only after the copies completed --
  // an allocation or copy that throws half way must not leave d_image set with the DFA images missing (the guard
  // below would then skip the upload for every later caller of the cached pattern).
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (re->d_image) return;
  Buf image = dev_alloc(re->image.size() * 4, s), tdfa, gtags, bits;
  CS_HIP(hipMemcpyAsync(image->p, re->image.data(), re->image.size() * 4, hipMemcpyHostToDevice, s));
  if (!re->bits.empty()) {
    bits = dev_alloc(re->bits.size() * 4, s);
    CS_HIP(hipMemcpyAsync(bits->p, re->bits.data(), re->bits.size() * 4, hipMemcpyHostToDevice, s));
  }
  if (!re->tdfa.empty()) {
    tdfa = dev_alloc(re->tdfa.size() * 4, s);
    CS_HIP(hipMemcpyAsync(tdfa->p, re->tdfa.data(), re->tdfa.size() * 4, hipMemcpyHostToDevice, s));
    if (!re->gtags.empty()) {
      gtags = dev_alloc(re->gtags.size() * 4, s);
      CS_HIP(hipMemcpyAsync(gtags->p, re->gtags.data(), re->gtags.size() * 4, hipMemcpyHostToDevice, s));
    }
  }
  CS_HIP(hipStreamSynchronize(s));
  re->d_tdfa = std::move(tdfa);
  re->d_gtags = std::move(gtags);
  re->d_bits = std::move(bits);
  re->d_image = std::move(image);  // (last: it is what the guard looks at)
}
TPlan tplan(cs_regex* re, int64_t rows, hipStream_t s) {
  require_device();
  upload(re, s);
  if (!re->d_tdfa) fail(CS_ERR_INTERNAL, "regex: the tagged-DFA image is not on the device");
  TPlan pl{};
  pl.d.tdfa = ptr<const int32_t>(re->d_tdfa);
  pl.d.tdfa_words = (int)re->tdfa.size();
  pl.d.image = ptr<const int32_t>(re->d_image);
  pl.d.in_lds = re->tdfa.size() * 4 <= kLdsBudget && !cs::cfg("CS_TDFA_GLOBAL_TABLE");
  pl.lds_bytes = pl.d.in_lds ? ((re->tdfa.size() * 4 + 15) & ~size_t(15)) : 0;
  int64_t nblk = (rows + 255) / 256;
  pl.grid = (unsigned)std::min<int64_t>(std::max<int64_t>(nblk, 1), 256 * 8);
  return pl;
}

Plan plan(cs_regex* re, int64_t rows, hipStream_t s) {
  require_device();
  upload(re, s);
  Plan pl{};
  Launch& L = pl.d;
  L.image = ptr<const int32_t>(re->d_image);
  L.image_words = (int)re->image.size();
  const int ninst = (int)re->prog.insts.size();
  pl.small = ninst <= 64;
  L.slots = csvm::vm_slots(ninst);
  const size_t img_bytes = (((size_t)L.image_words + 3) & ~size_t(3)) * 4;
  L.image_in_lds = img_bytes <= kLdsBudget / 2;
  L.arena = nullptr;
  pl.threads = 0;
  for (int t : {256, 128, 64}) {
    size_t need = (L.image_in_lds ? img_bytes : 0) + (size_t)t * L.slots * 4;
    if (need <= kLdsBudget) {
      pl.threads = t;
      pl.lds_bytes = need;
      break;
    }
  }
  if (pl.threads) {
    int64_t nblk = (rows + pl.threads - 1) / pl.threads;
    pl.grid = (unsigned)std::min<int64_t>(nblk, 256 * 16);
  } else {
    // lists in a global arena, one region per resident workgroup
    pl.threads = 256;
    pl.lds_bytes = L.image_in_lds ? img_bytes : 0;
    int64_t nblk = (rows + pl.threads - 1) / pl.threads;
    pl.grid = (unsigned)std::min<int64_t>(nblk, 256 * 4);
    if (pl.grid == 0) pl.grid = 1;
    size_t bytes = (size_t)pl.grid * pl.threads * L.slots * 4;
    pl.arena_buf = dev_alloc(bytes, s);
    L.arena = ptr<uint32_t>(pl.arena_buf);
  }
  if (pl.grid == 0) pl.grid = 1;
  return pl;
}

// Rows per tile and staging capacity for the stream kernels: 64 rows when their widest span fits
// the prefetch registers and no row outgrows the 96-byte candidate masks; otherwise the long-row
// variants (rows up to 255 bytes) with 64, 32 or 16 rows per tile.  R == 0: no stream kernel.
// matches per tile of R rows out of count_re's per-row counts: the column's total and the busiest tile's (replace_re with a
// growing replacement sizes its out tile and its output from them instead of provisioning for the worst case)
__global__ void __launch_bounds__(256) k_match_stats(const int32_t* __restrict__ counts, int64_t rows, int R, unsigned long long* __restrict__ total, unsigned* __restrict__ tilemax) {
  const int lane = threadIdx.x & 63;
  const int64_t ntiles = (rows + R - 1) / R;
  unsigned long long sum = 0;
  unsigned most = 0;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t r = t * R + lane;
    const int c = lane < R && r < rows ? max(counts[r], 0) : 0;
    const int ts = wave_reduce_sum(c);
    sum += (unsigned)ts;
    most = max(most, (unsigned)ts);
  }
  if (lane == 0) {
    if (sum) atomicAdd(total, sum);
    if (most > __hip_atomic_load(tilemax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(tilemax, most);
  }
}

__global__ void k_counts_to_flags(const int32_t* __restrict__ counts, int64_t rows, uint8_t* __restrict__ flags) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r < rows) flags[r] = counts[r] > 0 ? 1 : 0;
}

struct TileChoice {
  int R, cap;
  bool lng;
  int64_t span;  // the largest span of R consecutive rows (cap = that plus slack, rounded up to 128)
};
// `small`: also tiles of eight and four rows, also for rows beyond the sliding window (replace_re: its alternative is the
// two-pass thread-per-row kernels; the scans' row-wise kernels beat such tiles: contains_re 1.9 against 4.6 ms, findall 27
// against 64 ms on 520-byte rows)
// (replace_re on 64-row tiles of up to 8 KB for long rows -- eight prefetch chunks a lane -- was built and measured in round 6:
// 14.5 ms on the C5 column either way.  What such a column waits for is its tiles' PREFIXES, and twice the work per tile does
// not shorten that; nor does drawing the tickets an iteration later (4.33 ms on C3, 14.7 on C5 either way): profiles/r06/prefix_latency.txt.)
TileChoice choose_tile(const cs_column* col, hipStream_t s, bool small = false) {
  const int64_t longest = max_row_bytes(col, s);
  auto cap_of = [](int64_t span) { return (int64_t)((span + 15 + 32 + 127) & ~(int64_t)127); };
  const int64_t span64 = max_span64(col, s);
  const int64_t cap64 = cap_of(span64);
  const bool fits64 = cap64 <= cstile::kPfBytes;
  if (fits64 && longest + 3 <= cstd::Tdfa::kMaskBytes) return {64, (int)cap64, false, span64};
  if (longest > cstd::Tdfa::kLongBytes && (fits64 || !small)) return {fits64 ? 64 : 0, (int)cap64, false, span64};  // (such tiles scan generically)
  if (fits64) return {64, (int)cap64, true, span64};
  // (eight and four rows a tile: rows of hundreds of bytes -- few lanes of a wave hold a row then, but the rows still arrive
  // through coalesced tiles and are scanned in LDS; the thread-per-row kernels read them byte by byte from memory)
  for (int r : {32, 16, 8, 4}) {
    if (r < 16 && !small) break;
    const int64_t sp = max_span_rows(col, r, s);
    const int64_t c = cap_of(sp);
    if (c <= cstile::kPfBytes) return {r, (int)c, true, sp};
  }
  return {0, (int)cap64, false, span64};
}

// A column with rows beyond the 96-bit masks, a program for which white space is a safe cut (regex_tdfa.cpp, header word 31
// bit 25): the op runs on the column's PIECES (cs_virtual.hip) -- chain arithmetic, bit form, unit scan as on short rows --
// and the rows' results follow from the pieces'.
// Not for a single class, once or in a `+` loop (cs_runs.hip takes rows of any length byte-parallel: count_re([aeiou]+) on the
// C5 column 3.0 ms there, 5.6 on the pieces), and not -- `replacing` -- for a program that would take the unit scan on a column
// whose sample holds bytes >= 0x80: the unit route hands every sub-tile with such a byte to the row-by-row scan, and more,
// shorter rows make that worse (replace_re of the gtest pattern on C5: 40 ms on the pieces, 30 on the long-row form).
const VirtualRows* pieces_for(const cs_column* col, const cs_regex* re, hipStream_t s, bool replacing, bool containing = false) {
  if (cs::cfg("CS_NO_VIRTUAL_ROWS") || col->virt_state < 0 || !use_tdfa(re) || !((re->tdfa[31] >> 25) & 1)) return nullptr;
  if (max_row_bytes(col, s) + 3 <= cstd::Tdfa::kMaskBytes) return nullptr;  // (the masks hold the rows as they are)
  const bool chain = ((re->tdfa[30] >> 16) & 15) != 0;
  // One class, once or in a `+` loop: cs_runs.hip counts and replaces such a pattern byte-parallel on rows of any length -- and
  // keeps the patterns that would take the chain arithmetic here (`[a-z]+` -> '_' on the C5 column: 11.2 ms there, 13.1 on the
  // pieces) and the columns where bytes >= 0x80 are common (it takes them as they come; the mask forms do not).  The others go to
  // the pieces' bit form once such rows are put off / left holes: `\w+` 27.7 -> 7.9 ms (cs_runs.hip sends every tile with such a
  // byte row by row for a class with builtins), `[aeiou]+` 11.1 -> 8.0, count_re 2.9 -> 2.5; contains_re has no form there at all.
  if (!containing && !re->bits.empty() && (re->bits[2] & (csbits::F_BYTE_CLASS | csbits::F_FLAG_CLASS)) && !cs::cfg("CS_NO_CLASS_RUNS") &&
      (chain || (sample_has_high_bytes(col, s) && !odd_rows_few(col, s)) || cs::cfg("CS_CLASS_RUNS_ALWAYS")))
    return nullptr;
  // (... unless such bytes are rare: the single pass then leaves the rows that hold them holes -- StreamArgs::hole_mask)
  if (replacing && (re->tdfa[31] & 1) && !chain && sample_has_high_bytes(col, s) && !odd_rows_few(col, s)) return nullptr;
  // The executor ends a row's scan at a NUL byte and takes a character's width from its lead byte (an ASCII byte behind a
  // lead without its continuation bytes is swallowed -- a cut byte too): a piece behind such a byte would be scanned where
  // the row is not.  Columns of well-formed text only (`plain_bytes`: column metadata, one pass when nobody has looked yet).
  if (!bytes_plain(col, s)) return nullptr;
  return virtual_rows(col, s);
}

template <int MODE>
void scan(const cs_column* col, cs_regex* re, uint8_t* out8, int32_t* out32, int on_device, hipStream_t s,
          int64_t* found, const char* name) {
  if (found) *found = 0;
  note_route("");
  if (col->rows == 0) return;
  if (MODE == 0 || MODE == 2) {
    if (const VirtualRows* vr = pieces_for(col, re, s, false, MODE == 0)) {
      const cs_column* pc = vr->col.get();
      const size_t esz1 = MODE == 2 ? 4 : 1;
      Buf piece_res = dev_alloc(esz1 * (size_t)pc->rows, s);
      scan<MODE>(pc, re, MODE == 0 ? ptr<uint8_t>(piece_res) : nullptr, MODE == 2 ? ptr<int32_t>(piece_res) : nullptr, 1, s, nullptr, name);
      note_route_pieces();
      Buf row_res;
      void* dst = MODE == 2 ? (void*)out32 : (void*)out8;
      if (!on_device) {
        row_res = dev_alloc(esz1 * (size_t)col->rows, s);
        dst = row_res->p;
      }
      const int64_t hits = MODE == 2 ? virtual_reduce_i32(vr, ptr<const int32_t>(piece_res), col->rows, static_cast<int32_t*>(dst), s)
                                     : virtual_reduce_u8(vr, ptr<const uint8_t>(piece_res), col->rows, static_cast<uint8_t*>(dst), s);
      if (!on_device) {
        CS_HIP(hipMemcpyAsync(MODE == 2 ? (void*)out32 : (void*)out8, row_res->p, esz1 * (size_t)col->rows, hipMemcpyDeviceToHost, s));
        CS_HIP(hipStreamSynchronize(s));
      }
      if (found) *found = hits;
      return;
    }
  }
  // contains_re of a program with a unit decomposition whose candidate bytes are most of the column (`\w+@\w+` on log lines):
  // the row lanes' scan restarts at every candidate -- 9.0 ms on the C3 column where count_re, on the unit scan, takes 1.9.
  // "Holds a match" is "counts at least one": the unit scan's counts, turned into flags.
  if (MODE == 0 && use_tdfa(re) && (re->tdfa[31] & 1) && ((re->tdfa[30] >> 16) & 15) == 0 && !cs::cfg("CS_NO_CONTAINS_BY_COUNT") && !cs::cfg("CS_REGEX_ROWWISE") &&
      !cs::cfg("CS_NO_UNITS") && (((re->tdfa[31] >> 17) & 3) != 0 || !sample_has_high_bytes(col, s) || odd_rows_few(col, s)) && max_row_bytes(col, s) + 3 <= cstd::Tdfa::kMaskBytes &&
      !bits_route(re, col, s, BITS_CONTAINS, odd_rows_few(col, s)) && !bits_route(re, col, s, BITS_COUNT, odd_rows_few(col, s)) &&
      candidate_share(re, col, s) >= 0.5) {
    Buf counts = dev_alloc(sizeof(int32_t) * (size_t)col->rows, s);
    int64_t hits = 0;
    scan<2>(col, re, nullptr, ptr<int32_t>(counts), 1, s, &hits, name);
    Buf flags;
    uint8_t* dst = out8;
    if (!on_device) {
      flags = dev_alloc((size_t)col->rows, s);
      dst = ptr<uint8_t>(flags);
    }
    hipLaunchKernelGGL(k_counts_to_flags, dim3(blocks_for(col->rows)), dim3(256), 0, s, ptr<const int32_t>(counts), col->rows, dst);
    CS_HIP(hipGetLastError());
    if (!on_device) CS_HIP(hipMemcpyAsync(out8, flags->p, (size_t)col->rows, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (found) *found = hits;
    return;
  }
  const bool wide = use_tdfa_wide(re);  // (five to eight live threads: TdfaWide on the same kernels' generic row path)
  const bool tdfa = use_tdfa(re) || wide;
  Plan pl{};
  TPlan tp{};
  if (tdfa) tp = tplan(re, col->rows, s);
  else pl = plan(re, col->rows, s);
  const size_t esz = MODE == 2 ? 4 : 1;
  void* host_out = MODE == 2 ? (void*)out32 : (void*)out8;
  Buf tmp;
  if (!on_device) {
    tmp = dev_alloc(esz * col->rows, s);
    if (MODE == 2) out32 = ptr<int32_t>(tmp);
    else out8 = ptr<uint8_t>(tmp);
  }
  // count_re of ONE class, once or in a `+` loop, on a column the 96-bit-mask forms do not take (long rows, non-ASCII text) and
  // whose candidates are everywhere: the byte-parallel size pass of cs_runs.hip counting matches -- count_re(\w+), the words of
  // a row, on the C5 column: 15.8 ms on the long-row automaton form
  if (MODE == 2 && !re->bits.empty() && (re->bits[2] & (csbits::F_BYTE_CLASS | csbits::F_FLAG_CLASS))) {
    upload(re, s);
    const TileChoice tc0 = choose_tile(col, s, true);
    const bool masks_form = tc0.R == 64 && !tc0.lng && (!sample_has_high_bytes(col, s) || odd_rows_few(col, s));
    const bool chain_there = !re->tdfa.empty() && ((re->tdfa[30] >> 16) & 15) != 0 && tc0.R == 64 && !tc0.lng;
    const bool flag_class = (re->bits[2] & csbits::F_BYTE_CLASS) == 0;
    int64_t hits = 0;
    if (re->d_bits && (cs::cfg("CS_CLASS_RUNS_ALWAYS") || (!masks_form && !(flag_class && chain_there) && candidate_share(re, col, s) >= 0.05)) &&
        count_class_runs(col, ptr<const int32_t>(re->d_bits), re->bits, s, out32, &hits)) {
      note_route("runs");
      if (!on_device) {
        CS_HIP(hipMemcpyAsync(host_out, tmp->p, esz * col->rows, hipMemcpyDeviceToHost, s));
        CS_HIP(hipStreamSynchronize(s));
      }
      if (found) *found = hits;
      return;
    }
  }
  Buf cnt = dev_alloc(8, s);
  CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
  RowSrc src{view_of(col), d_unicode_flags(), col->nbytes + (col->chars && col->chars->capacity ? 64 : 0)};
  bool streamed = false;
  if (tdfa && tp.d.in_lds && !cs::cfg("CS_REGEX_ROWWISE")) {
    const TileChoice tc = choose_tile(col, s);
    const int cap = tc.cap;
    // the unit scan (k_tdfa_scan_stream<.., UNITS>): patterns whose tagged DFA offers the decomposition, rows within the masks
    // (count_re only: contains_re stops at a row's first match, and scanning every unit of the row cost more than the
    // balance won -- 3.87 against 2.66 ms on the 100M-row C3 column)
    // (... but the unit form of the kernel is taken for contains_re too when the pattern lets the unit route serve tiles with
    // bytes >= 0x80 -- header word 31 bits 17 / 18 -- and a sample of the chars holds such bytes: ASCII tiles keep the row
    // lanes' scan inside it, at a few per cent more than the plain form)
    // (chain patterns -- header words 29 / 30 -- were tried on the unit form for contains_re as well: 2.18 against 2.10 ms; the
    // plain form's first-match scan stays)
    // (a chain pattern without a unit decomposition -- one with a suffix, regex_tdfa.cpp -- takes the same kernels: their unit
    // routes test header word 31 bit 0 themselves)
    // (a column whose sample holds a few bytes >= 0x80: the rows that hold them are put off -- ScanStreamArgs::deferred,
    // k_tdfa_scan_list -- and the forms are chosen as on plain ASCII; the C5 column, one row in 170: contains_re of the gtest
    // pattern 9.9 ms when every sub-tile with such a row went to the row-by-row scan)
    const bool put_off = (MODE == 0 || MODE == 2) && !wide && !tc.lng && tc.R == 64 && sample_has_high_bytes(col, s) && odd_rows_few(col, s);
    const bool high_sample = sample_has_high_bytes(col, s) && !put_off;
    const bool units = !wide && (MODE == 2 || (MODE == 0 && ((re->tdfa[31] >> 17) & 3) != 0 && high_sample)) &&
                       ((re->tdfa[31] & 1) != 0 || (MODE == 2 && ((re->tdfa[30] >> 16) & 15) != 0)) &&
                       !tc.lng && tc.R == 64 && !cs::cfg("CS_NO_UNITS");
    // the bit-parallel form (regex_bits.h): contains_re / count_re of patterns whose candidates are everywhere
    const bool bits_form = !wide && (MODE == 0 || MODE == 2) && !tc.lng && tc.R == 64 && bits_route(re, col, s, MODE == 2 ? BITS_COUNT : BITS_CONTAINS, put_off);
    const int bits_k = bits_form ? std::max(re->bits[1], 2) : 0;
    const size_t bits_lds = bits_form ? (size_t)bits_lds_bytes((int)re->bits.size()) : 0;
    size_t lds = bits_form ? tp.lds_bytes + bits_lds + (size_t)(cap + 32 + bits_k * ((cap >> 3) + 32) + kUnitQueue * 4 + 64 * 4 + 16) * 4
                                 : tp.lds_bytes + (size_t)(cap + 32 + (cap >> 3) + 32 + (units ? (cap >> 3) + 32 + kUnitQueue * 4 + 64 * 4 + 16 : 0)) * 4;
    if (put_off) lds += (size_t)((cap >> 3) + 32 + 64 * 4) * 4;  // (the bitmap of such bytes and the rows waiting for the list, a wave)
    // (a chain's arithmetic reads no table: where the tables cost the launch a workgroup per CU -- four fit in 40 KB each --
    // they stay in memory, as in cs_replace_re)
    const bool chain_scan = units && MODE == 2 && !bits_form && ((re->tdfa[30] >> 16) & 15) != 0 && !high_sample && !cs::cfg("CS_NO_CHAIN_FORM");
    // (the bit form is such a form too: its sub-tiles of plain ASCII never touch the automaton)
    const bool chain_global = (chain_scan || bits_form) && lds > 40 * 1024 && lds - tp.lds_bytes + cstd::kHeadTailWords * 4 <= 40 * 1024 && !cs::cfg("CS_CHAIN_TABLES_IN_LDS");
    const size_t scan_tbl = chain_global ? (size_t)cstd::kHeadTailWords * 4 : tp.lds_bytes;  // (header + tail words: tsetup)
    if (chain_global) lds -= tp.lds_bytes - scan_tbl;
    if (tc.R && lds <= 150 * 1024) {
      ScanStreamArgs sa{};
      sa.in = view_of(col);
      sa.flags = d_unicode_flags();
      sa.L = tp.d;
      if (chain_global) sa.L.in_lds = 2;
      sa.out8 = out8;
      sa.out32 = out32;
      sa.found = ptr<unsigned long long>(cnt);
      sa.nsub = (col->rows + tc.R - 1) / tc.R;
      sa.rows_per_tile = tc.R;
      sa.cap_in = cap;
      sa.tbl_bytes = (int)(scan_tbl + bits_lds);
      sa.bits = bits_form ? ptr<const int32_t>(re->d_bits) : nullptr;
      sa.bits_off = (int)scan_tbl;
      sa.bits_words = bits_form ? (int)re->bits.size() : 0;
      sa.bits_k = bits_form ? re->bits[1] : 0;
      auto kern = tc.lng ? &k_tdfa_scan_stream<MODE, true, true> : &k_tdfa_scan_stream<MODE, true, false>;
      if (units) kern = &k_tdfa_scan_stream<MODE == 2 ? 2 : 0, true, false, true>;
      if (chain_scan) kern = &k_tdfa_scan_stream<2, true, false, true, true>;  // (a chain pattern on a column whose sample is plain ASCII)
      if (chain_global) kern = &k_tdfa_scan_stream<2, false, false, true, true>;
      if (bits_form) kern = chain_global ? &k_tdfa_scan_stream<MODE == 2 ? 2 : 0, false, false, true, true, true> : &k_tdfa_scan_stream<MODE == 2 ? 2 : 0, true, false, true, true, true>;
      note_route(bits_form ? "bits" : wide ? "wide" : chain_scan ? "chain" : units ? "units" : "plain");
      if (wide) kern = tc.lng ? &k_tdfa_scan_stream<MODE + 7, true, true> : &k_tdfa_scan_stream<MODE + 7, true, false>;
      if (lds > 48 * 1024)
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const unsigned grid = resident_grid(reinterpret_cast<const void*>(kern), lds, (sa.nsub + 3) / 4);
      Buf later, nlater;
      if (put_off) {
        sa.deferred_cap = (unsigned)std::min<int64_t>(col->rows, col->rows / 8 + 4096);
        later = dev_alloc(sizeof(int32_t) * (size_t)sa.deferred_cap, s);
        nlater = dev_alloc(sizeof(unsigned), s);
        CS_HIP(hipMemsetAsync(nlater->p, 0, sizeof(unsigned), s));
        sa.deferred = ptr<int32_t>(later);
        sa.ndeferred = ptr<unsigned>(nlater);
      }
      {
        ProfScope ps(name, s);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, sa);
      }
      streamed = true;
      if (put_off) {
        note_route_put_off();
        const unsigned lgrid = (unsigned)std::min<int64_t>(((int64_t)sa.deferred_cap + 255) / 256, 2048);
        const size_t list_lds = tp.lds_bytes + (size_t)256 * kListRowBytes;  // (the tables and a row a thread)
        if (list_lds > 48 * 1024)
          CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tdfa_scan_list<MODE == 2 ? 2 : 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)list_lds));
        ProfScope ps("k_tdfa_scan_list", s);
        hipLaunchKernelGGL((k_tdfa_scan_list<MODE == 2 ? 2 : 0>), dim3(lgrid), dim3(256), list_lds, s, src, tp.d, ptr<const int32_t>(later), ptr<const unsigned>(nlater),
                           sa.deferred_cap, out8, out32, ptr<unsigned long long>(cnt));
        // (more such rows than the sample promised and the list holds: the column is scanned again with nothing put off)
        unsigned* hn = (unsigned*)pinned_scratch(sizeof(unsigned));
        CS_HIP(hipMemcpyAsync(hn, nlater->p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        CS_HIP(hipStreamSynchronize(s));
        if (*hn > sa.deferred_cap) {
          note_fallback("regex-deferred-rows-overflow");
          streamed = false;
          CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
        }
      }
    }
  }
  if (!streamed) {
    ProfScope ps(name, s);
    if (wide)
      if (tp.d.in_lds)
        hipLaunchKernelGGL((k_tdfa_scan<MODE, true, true>), dim3(tp.grid), dim3(256), tp.lds_bytes, s, src, tp.d, out8, out32,
                           ptr<unsigned long long>(cnt));
      else
        hipLaunchKernelGGL((k_tdfa_scan<MODE, false, true>), dim3(tp.grid), dim3(256), 0, s, src, tp.d, out8, out32,
                           ptr<unsigned long long>(cnt));
    else if (tdfa)
      if (tp.d.in_lds)
        hipLaunchKernelGGL((k_tdfa_scan<MODE, true>), dim3(tp.grid), dim3(256), tp.lds_bytes, s, src, tp.d, out8, out32,
                           ptr<unsigned long long>(cnt));
      else
        hipLaunchKernelGGL((k_tdfa_scan<MODE, false>), dim3(tp.grid), dim3(256), 0, s, src, tp.d, out8, out32,
                           ptr<unsigned long long>(cnt));
    else if (pl.small)
      hipLaunchKernelGGL((k_regex_scan<true, MODE>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d, out8,
                         out32, ptr<unsigned long long>(cnt));
    else
      hipLaunchKernelGGL((k_regex_scan<false, MODE>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d, out8,
                         out32, ptr<unsigned long long>(cnt));
  }
  CS_HIP(hipGetLastError());
  if (!on_device)
    CS_HIP(hipMemcpyAsync(host_out, tmp->p, esz * col->rows, hipMemcpyDeviceToHost, s));
  int64_t* host = (int64_t*)pinned_scratch(8);
  CS_HIP(hipMemcpyAsync(host, cnt->p, 8, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (found) *found = host[0];
}

}  // namespace

extern "C" {

// Reprog::create_from + dreprog::create_from -- regcomp.cpp:954-960, regexec.cpp:12-73
int cs_regex_compile(const char* pattern, cs_regex** out) {
  return guard([&] {
    if (!pattern || !out) fail(CS_ERR_INVALID_ARG, "regex pattern cannot be null");
    // A compiled pattern is immutable (the device images are uploaded once, under a lock) and costs 0.6 - 1.6 ms of host
    // time to build -- a tenth of the replace_re kernel on 100M rows, more than the kernel on a few million -- while
    // the callers above the C ABI compile per call, as the reference does.  The last 32 patterns are kept; a handle
    // is a counted reference to the shared object.
    static std::mutex& mu = *new std::mutex;  // (never destroyed: see the note on process exit in cs_core.hip's buffer cache)
    static auto& kept = *new std::list<std::pair<std::string, cs_regex*>>;
    const bool keep = !cs::cfg("CS_REGEX_NO_CACHE");
    if (keep) {
      std::lock_guard<std::mutex> lk(mu);
      for (auto it = kept.begin(); it != kept.end(); ++it)
        if (it->first == pattern) {
          kept.splice(kept.begin(), kept, it);
          it->second->refs.fetch_add(1);
          *out = it->second;
          return;
        }
    }
    std::unique_ptr<cs_regex> re(new cs_regex);
    re->empty_pattern = *pattern == 0;
    re->prog = csrx::compile(pattern);
    re->blob = re->prog.to_blob();
    re->image = re->prog.to_device_image(h_unicode_flags());
    re->tdfa = csrx::build_tdfa(re->prog, re->image, h_unicode_flags(), &re->gtags);
    re->bits = csrx::build_bits(re->prog, re->image, h_unicode_flags());
    cs_regex* dropped = nullptr;
    if (keep) {
      std::lock_guard<std::mutex> lk(mu);
      re->refs.store(2);
      kept.emplace_front(pattern, re.get());
      if (kept.size() > 32) {
        dropped = kept.back().second;
        kept.pop_back();
      }
    }
    *out = re.release();
    if (dropped && dropped->refs.fetch_sub(1) == 1) delete dropped;
  });
}
int cs_regex_destroy(cs_regex* re) {
  return guard([&] {
    if (re && re->refs.fetch_sub(1) == 1) delete re;
  });
}
int cs_regex_inst_count(const cs_regex* re) { return re ? (int)re->prog.insts.size() : 0; }
int cs_regex_engine(const cs_regex* re) {
  if (!re) return 0;
  int e = 0;
  if (!re->tdfa.empty()) {
    e |= 1;                               // the tagged DFA (regex_tdfa.h) runs the pattern
    if (re->tdfa[31] & 1) e |= 2;         // ... and offers the unit decomposition (regex_tdfa.cpp)
    if (!re->gtags.empty()) e |= 4;       // ... and carries the capture-group tags
    e |= (re->tdfa[12] & 15) << 8;        // live threads the automaton keeps at most
    e |= (re->tdfa[1] & 0xFFFF) << 16;    // states
  }
  return e;
}
int cs_regex_blob(const cs_regex* re, const int32_t** words, int* nwords) {
  return guard([&] {
    if (!re || !words || !nwords) fail(CS_ERR_INVALID_ARG, "null argument");
    *words = re->blob.data();
    *nwords = (int)re->blob.size();
  });
}

// NVStrings::contains_re -- count.cu:59-110
int cs_contains_re(const cs_column* col, const cs_regex* re, uint8_t* results, int on_device, cs_stream stream,
                   int64_t* found) {
  return guard([&] {
    if (found) *found = -1;
    if (!col || !re || !results) fail(CS_ERR_INVALID_ARG, "contains_re: null argument");
    scan<0>(col, const_cast<cs_regex*>(re), results, nullptr, on_device, S(stream), found, "k_contains_re");
  });
}
// NVStrings::match -- count.cu:113-165
int cs_match_re(const cs_column* col, const cs_regex* re, uint8_t* results, int on_device, cs_stream stream,
                int64_t* found) {
  return guard([&] {
    if (found) *found = -1;
    if (!col || !re || !results) fail(CS_ERR_INVALID_ARG, "match: null argument");
    scan<1>(col, const_cast<cs_regex*>(re), results, nullptr, on_device, S(stream), found, "k_match_re");
  });
}
// NVStrings::count_re -- count.cu:168-250
int cs_count_re(const cs_column* col, const cs_regex* re, int32_t* results, int on_device, cs_stream stream,
                int64_t* found) {
  return guard([&] {
    if (found) *found = -1;
    if (!col || !re || !results) fail(CS_ERR_INVALID_ARG, "count_re: null argument");
    scan<2>(col, const_cast<cs_regex*>(re), nullptr, results, on_device, S(stream), found, "k_count_re");
  });
}

// NVStrings::replace_re -- replace.cu:110-189
int cs_replace_re(const cs_column* col, const cs_regex* cre, const char* repl, int maxrepl, cs_stream stream,
                  cs_column** out) {
  return guard([&] {
    if (!col || !cre || !out) fail(CS_ERR_INVALID_ARG, "replace_re: null argument");
    if (cre->empty_pattern)
      fail(CS_ERR_INVALID_ARG, "nvstrings::replace_re parameter cannot be null or empty");  // replace.cu:112-113
    cs_regex* re = const_cast<cs_regex*>(cre);
    hipStream_t s = S(stream);
    require_device();
    if (col->rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    if (!repl) repl = "";
    const int rb = (int)strlen(repl);
    note_route("");
    Buf d_repl = dev_alloc((size_t)rb + 1, s);
    CS_HIP(hipMemcpyAsync(d_repl->p, repl, (size_t)rb + 1, hipMemcpyHostToDevice, s));
    const bool tdfa = use_tdfa(re);
    const bool wide = use_tdfa_wide(re);  // (five to eight live threads: the two-pass kernels on TdfaWide)
    Plan pl{};
    TPlan tp{};
    if (tdfa || wide) tp = tplan(re, col->rows, s);
    else pl = plan(re, col->rows, s);
    // rows beyond the 96-bit masks, white space a safe cut for this program: the pieces' replace_re is the rows' (cs_virtual.hip)
    if (!cs::g_backrefs_dev && !cs::g_replace_plain_only && maxrepl < 0 && tdfa && !cs::cfg("CS_CLASS_RUNS_ALWAYS")) {
      if (const VirtualRows* vr = pieces_for(col, re, s, true)) {
        cs_column* po = nullptr;
        const int rc = cs_replace_re(vr->col.get(), cre, repl, maxrepl, stream, &po);
        if (rc != 0) fail(rc, cs_last_error());
        note_route_pieces();
        *out = virtual_rows_to_rows(col, vr, std::unique_ptr<cs_column>(po), s);
        return;
      }
    }
    // One character class, once or in a `+` loop, on a column the 96-bit-mask forms do not take -- rows beyond 93 bytes,
    // tiles of fewer than 64 rows, non-ASCII text -- and whose candidates are everywhere: byte-parallel stream compaction
    // (cs_runs.hip; BASELINE.json C5: replace_re([aeiou]+) on 40-150-byte rows, 82.8 ms on the long-row automaton forms).
    if (!cs::g_backrefs_dev && !cs::g_replace_plain_only && maxrepl < 0 && col->rows > 0 && !re->bits.empty() && (re->bits[2] & (csbits::F_BYTE_CLASS | csbits::F_FLAG_CLASS)) &&
        re->d_bits) {
      const TileChoice tc0 = choose_tile(col, s, true);
      const bool masks_form = tc0.R == 64 && !tc0.lng && (!sample_has_high_bytes(col, s) || odd_rows_few(col, s));  // (few rows with such bytes: holes)
      // (a class with builtins -- `\w+` -- sends every tile with a byte >= 0x80 row by row: worth it against the automaton's
      // long-row and dense forms -- `\w+` on the C5 column 97.7 -> 11.2 ms, on C2 16.2 -> 5.1 --, not against the chain
      // arithmetic where that runs: `\s+` on C2 4.4 there, 6.7 here)
      const bool chain_there = !re->tdfa.empty() && ((re->tdfa[30] >> 16) & 15) != 0 && tc0.R == 64 && !tc0.lng;
      const bool flag_class = (re->bits[2] & csbits::F_BYTE_CLASS) == 0;
      const bool wanted = cs::cfg("CS_CLASS_RUNS_ALWAYS") || (!masks_form && !(flag_class && chain_there) && candidate_share(re, col, s) >= 0.05);
      cs_column* r = nullptr;
      if (wanted && replace_class_runs(col, ptr<const int32_t>(re->d_bits), re->bits, repl, rb, s, &r)) {
        note_route("runs");
        *out = r;
        return;
      }
    }
    RowSrc src{view_of(col), d_unicode_flags(), col->nbytes + (col->chars && col->chars->capacity ? 64 : 0)};
    auto* o = new cs_column;
    std::unique_ptr<cs_column> holder(o);
    o->rows = col->rows;
    o->validity = col->validity;
    o->null_count = col->null_count;
    const int minlen = (tdfa || wide) ? re->tdfa[13] : 0;
    // Single pass when a match cannot be empty: a match is at least `minlen` bytes, so a row grows
    // by at most (rb - minlen) bytes per match; rb <= minlen means "never grows" and the rows are
    // rewritten in place.  A growing replacement is provisioned for kMaxRec matches per row first
    // (or, for one- and two-byte patterns, which tend to match often, for up to the worst case);
    // a launch that runs out of room says so in its error word and is repeated once with the
    // roomier sizing.  A pattern that matches the empty string with a non-empty replacement is
    // unbounded (the zero-length repeat rule) and takes the two-pass kernels.
    const int growth = rb > minlen ? rb - minlen : 0;
    // (a pattern that matches the empty string inserts at most one replacement per character and one at the row's end:
    // short replacements are provisioned for exactly that -- `x*` -> '-' doubles the column -- and stay on the single pass)
    const bool empties = minlen == 0 && growth > 0;
    const int minlen_p = std::max(minlen, 1);
    const bool bounded = growth == 0 || minlen >= 1 || rb <= 8;  // (longer ones: the out tile of 1 + rb times the input does not fit the LDS)
    // (programs of five to eight threads: the stream kernel's WIDE forms -- tables in LDS, replacements of up to 16 bytes)
    const bool wide_stream = wide && tp.d.in_lds && rb <= 16 && !cs::g_backrefs_dev && !cs::cfg("CS_WIDE_TWO_PASS");
    if ((tdfa || wide_stream) && bounded && !cs::cfg("CS_REGEX_TWO_PASS")) {
      const int64_t rows = col->rows;
      const int64_t ntiles = (rows + cstile::kTileRows - 1) / cstile::kTileRows;
      const int64_t nsub = ntiles * 4;
      TileChoice tc = choose_tile(col, s, !cs::cfg("CS_NO_SMALL_TILES"));
      // (the column's largest 64-row span does not fit, all but a few do: buffers for those, the kernel handles the rest)
      bool outliers = false;
      constexpr int64_t kOutlierSpan = cstile::kPfBytes - 176;  // (its capacity is kPfBytes)
      if (!tc.R && tdfa && tp.d.in_lds && rb <= 16 && !cs::g_backrefs_dev && !cs::cfg("CS_NO_OUTLIER_TILES") && max_span64(col, s) <= (16 << 20) &&
          few_spans64_over(col, kOutlierSpan, s)) {
        tc = TileChoice{64, (int)((kOutlierSpan + 15 + 32 + 127) & ~(int64_t)127), false, kOutlierSpan};
        outliers = true;
      }
      const int cap = tc.cap;
      const size_t tbl = tp.d.in_lds ? tp.lds_bytes : 0;
      size_t lds = tbl + (size_t)(cap + cap + 64 + (cap >> 3) + 32) * 4 + 16;
      // persistent stream kernel (grid = what is resident at once); returns its error word, or -1
      // when the sizing does not fit
      int64_t sized_total = -1, sized_tilemax = 0;  // matches in the column / in its busiest tile, when they were counted first
      // a column whose sample holds a few bytes >= 0x80: the unit forms run as on plain ASCII, the rows that hold such bytes are
      // sized beforehand and written afterwards, a thread a row (StreamArgs::hole_mask) -- replace_re of the gtest pattern on the C5
      // column's pieces: 34 ms when a third of the sub-tiles went to the row-by-row scan for the sake of one row each
      const bool holes_ok = tdfa && tp.d.in_lds && maxrepl < 0 && !cs::g_backrefs_dev && !cs::g_replace_plain_only && !wide_stream && !outliers && tc.R == 64 && !tc.lng &&
                            sample_has_high_bytes(col, s) && odd_rows_few(col, s);
      Buf hole_lens;  // (taken once: a second, roomier attempt uses them again)
      auto stream_attempt = [&](bool roomy) -> int {
        // (a DFA whose tables do not fit the LDS budget: the two-pass kernels read them from memory)
        if (!tp.d.in_lds) return -1;
        const int64_t few = (int64_t)rows * kMaxRec * growth;  // extra bytes if no row has more than kMaxRec matches
        int cap_out = cap + ((64 * kMaxRec * growth + 127) & ~127);
        // (no growth: the out tile holds at most the in tile's span -- rounded up to 16, not to the in tile's 128: a form of
        // 54 496 bytes of LDS ran two workgroups per CU where 54 240 run three)
        if (growth == 0 && !outliers) cap_out = (int)std::min<int64_t>(cap, (tc.span + 15 + 32 + 15) & ~(int64_t)15);
        int64_t extra = std::min<int64_t>(few, col->nbytes + (1ll << 30));
        if (roomy && sized_total >= 0) {
          // (the matches were counted first: the busiest tile's growth and the column's, exactly)
          const int64_t span_room = std::min<int64_t>(cap, (tc.span + 15 + 32 + 15) & ~(int64_t)15);
          cap_out = (int)((span_room + sized_tilemax * growth + 127) & ~(int64_t)127);
          extra = sized_total * growth + 64;
        } else if (roomy) {
          const int64_t worst = ((int64_t)cap * rb + minlen_p - 1) / minlen_p;
          cap_out = std::max(cap_out, (int)((std::min<int64_t>(worst, 3ll * cap) + 127) & ~(int64_t)127));
          const int64_t worst_extra = (col->nbytes * growth + minlen_p - 1) / minlen_p;
          extra = std::min(worst_extra, std::max<int64_t>(extra, col->nbytes));
          if (empties) {  // every character and every row end may take a replacement
            cap_out = (int)(((int64_t)cap * (1 + rb) + 64 * rb + 127) & ~(int64_t)127);
            extra = col->nbytes * rb + rows * rb;
          }
        }
        // the unit scan (k_tdfa_replace_stream<.., UNITS>): patterns whose tagged DFA offers the decomposition, no limit
        // on the number of replacements, rows within the 96-byte masks
        // (not for the literal needles of cs_replace: short needles match densely, and then the per-row kernel with its
        // in-place compaction is the faster one -- 'ab' -> 'x' on the C2 column: 1.25 against 1.77 ms)
        // replace_with_backrefs on this kernel (cs_replace_with_backrefs left its template in g_backrefs_dev): the unit
        // scan, groups carried by the DFA (four at most), tables in LDS
        const bool brefs = cs::g_backrefs_dev != nullptr;
        // (a chain without a unit decomposition has no route for sub-tiles with bytes >= 0x80 in this form: only on columns whose sample is plain ASCII)
        if (brefs && !(((re->tdfa[31] & 1) != 0 || (((re->tdfa[30] >> 16) & 15) != 0 && ((re->tdfa[30] >> 20) & 1) != 0 && !sample_has_high_bytes(col, s))) && !tc.lng && tc.R == 64 && tp.d.in_lds && !re->gtags.empty() && re->prog.num_groups <= cstd::Tdfa::kGroupBatch &&
                       re->gtags.size() * 4 <= 8 * 1024))
          return -1;
        const bool literal = !brefs && !outliers && cs::g_replace_plain_only && cs::g_replace_literal_len > 0 && maxrepl < 0 && !tc.lng && tc.R == 64 && !cs::cfg("CS_NO_LITERAL_SCAN");
        const bool offers = (re->tdfa[31] & 1) != 0 || ((re->tdfa[30] >> 16) & 15) != 0;  // (units, or a chain pattern without them: one with a suffix)
        // the bit-parallel form (regex_bits.h): patterns whose candidates are everywhere -- a bitmap per character class (it
        // builds on the unit forms' layout and assembly, whether or not the pattern offers a unit decomposition)
        const bool bits_form = !literal && !brefs && !wide_stream && tdfa && maxrepl < 0 && !tc.lng && tc.R == 64 && !outliers && !cs::g_replace_plain_only && tp.d.in_lds &&
                               bits_route(re, col, s, BITS_REPLACE, holes_ok);
        // replace_with_backrefs on a chain whose groups are runs of items, the sample plain ASCII: the form that keeps neither
        // tables nor group tags in LDS (k_tdfa_replace_stream<.., BREFS, .., CHAIN>).  Its out tile is sized from the
        // template: an expansion is the template's literal bytes plus the groups it names, so a match grows by at most
        //   G = literal bytes + sum over items of (times the item is named - 1) * its length
        // -- with every item named at most once, G = literal bytes - the least length of the items NOT named, often <= 0
        // (`\4.\3.\2.\1` on four dotted groups: 3 - 3).  An item named twice and unbounded leaves the old "twice the input".
        bool bchain = false;
        int64_t brefs_grow = -1;  // per match; -1: unknown
        if (brefs && ((re->tdfa[30] >> 16) & 15) != 0 && ((re->tdfa[30] >> 20) & 1) != 0 && !tc.lng && tc.R == 64 && maxrepl < 0 &&
            cs::g_backrefs_host && !cs::cfg("CS_NO_BREFS_CHAIN") && !sample_has_high_bytes(col, s)) {
          const auto* ht = static_cast<const csvm::BackrefTemplate*>(cs::g_backrefs_host);
          const cstd::View hv = cstd::make_view(re->tdfa.data());
          const int ni = (int)((hv.chain >> 16) & 15u);
          const unsigned long long hcrep = cstd::chain_crep(re->tdfa.data());
          const uint32_t gmap = (uint32_t)re->tdfa[re->tdfa[15] - 1];
          int named[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int j = 0; j < ht->nrefs; ++j) {
            const int g = ht->idx[j];
            int lo = 0, hi = 0;
            if (g == 0) hi = ni;
            else if (g <= 4 && g <= ht->groups) lo = (int)((gmap >> (8 * (g - 1))) & 15u), hi = (int)((gmap >> (8 * (g - 1) + 4)) & 15u);
            for (int k = lo; k < hi && k < 8; ++k) ++named[k];
          }
          int64_t G = ht->bytes;
          bool bounded_g = true;
          for (int k = 0; k < ni && k < 8; ++k) {
            const int least = (int)(cstd::chain_rep(hcrep, k) & 15u), most = (int)(cstd::chain_rep(hcrep, k) >> 4);
            if (named[k] == 0) G -= least;
            else if (named[k] > 1) {
              if (most == 0) bounded_g = false;
              else G += (int64_t)(named[k] - 1) * most;
            }
          }
          bchain = true;
          if (bounded_g) brefs_grow = std::max<int64_t>(G, 0);
        }
        const bool units = literal || brefs || bits_form || (offers && maxrepl < 0 && !tc.lng && tc.R == 64 && !outliers && !cs::g_replace_plain_only && !cs::cfg("CS_NO_UNITS"));
        const int bits_k = bits_form ? std::max(re->bits[1], 2) : 2;
        const size_t bits_lds = bits_form ? (size_t)bits_lds_bytes((int)re->bits.size()) : 0;
        // (a chain pattern on a column whose sample is plain ASCII: the form without the unit / lean scans)
        const bool chain_form = !bits_form && !brefs && !wide_stream && !outliers && units && cap <= 5 * 1024 && tp.d.in_lds && !literal && maxrepl < 0 &&
                                ((re->tdfa[30] >> 16) & 15) != 0 && (!sample_has_high_bytes(col, s) || holes_ok) && !cs::cfg("CS_NO_CHAIN_FORM");
        // (the kernel's layout, k_tdfa_replace_stream: a bitmap's bytes, what stands behind the first one)
        const size_t bm = (size_t)(cap >> 3) + (bits_form ? 16 : 32);
        const size_t unit_bytes = bchain                      ? bm + 16 + kUnitQueue * 12
                                  : (chain_form || bits_form) ? (size_t)(bits_k - 1) * bm + 16
                                  : units                     ? (size_t)(bits_k - 1) * bm + kUnitQueue * 4 + 16 + (cs::g_backrefs_dev ? kUnitQueue * 12 + 64 * 4 : 0)
                                                              : 0;
        // (backrefs: the group tags and the template text sit behind the DFA table; the template may grow a row by any
        // amount: room for twice the input, a launch that needs more says so and the two-pass form takes over)
        const size_t gt_bytes = brefs ? (bchain ? 0 : ((re->gtags.size() * 4 + 15) & ~size_t(15))) + (((size_t)cs::g_backrefs_text_bytes + 31) & ~size_t(15)) : 0;
        if (brefs && brefs_grow >= 0) {
          // (at most span / minlen matches in a sub-tile, each growing by at most brefs_grow bytes -- the roomy sizing; the first
          // attempt provisions a quarter of the input instead when that is less: `(\d+)` -> `<\1>` may triple a row of digits
          // and spaces, and adds a sixth to a log line -- at the worst case's 84 KB of LDS one workgroup fits a CU, 48.8 ms on the
          // C3 column; a launch that runs out of room says so and the roomy one follows)
          const int64_t span_room = std::min<int64_t>(cap, (tc.span + 15 + 32 + 15) & ~(int64_t)15);
          int64_t tile_grow = (tc.span / minlen_p + 1) * brefs_grow, col_grow = (col->nbytes / minlen_p + 1) * brefs_grow;
          if (!roomy) {
            tile_grow = std::min<int64_t>(tile_grow, std::max<int64_t>(tc.span / 4, 256));
            col_grow = std::min<int64_t>(col_grow, col->nbytes / 4 + (1 << 20));
          }
          cap_out = (int)((span_room + tile_grow + (tile_grow ? 127 : 0)) & ~(int64_t)(tile_grow ? 127 : 15));
          extra = col_grow;
        } else if (brefs) {
          cap_out = std::max(cap_out, 2 * cap);
          extra = std::max<int64_t>(extra, col->nbytes);
        }
        // The chain arithmetic reads no table: when the tables are what keeps a third workgroup off the CU (the 26-instruction
        // dotted quad with `\b` and {1,3}: 36 states, 19 KB), they stay in memory -- only a sub-tile the arithmetic does not take
        // (bytes >= 0x80, a row beyond the masks) walks them there, and the sample says those are rare.
        const size_t tile_lds = gt_bytes + bits_lds + (size_t)(cap + cap_out + 64 + bm + unit_bytes) * 4 + 16;
        constexpr size_t kThird = 160 * 1024 / 3;
        // (the bit form likewise: its kernel with the tables in memory is built for three workgroups a CU -- 168 registers)
        // (... whenever its tile leaves room for three, whatever the tables' size: the form with the tables in LDS is the 189-register
        // one -- for patterns of up to five classes: the gtest pattern's seven fit a third workgroup too once the layout was trimmed,
        // but its evaluation then spills where it hurts: 8.4 against 7.1 ms)
        const bool chain_global = ((chain_form && tbl + tile_lds > kThird && tile_lds + cstd::kHeadTailWords * 4 <= kThird) ||
                                   (bits_form && re->bits[1] <= 5 && tile_lds + cstd::kHeadTailWords * 4 <= kThird) || ((chain_form || bits_form) && cs::cfg("CS_CHAIN_TABLES_IN_MEMORY"))) &&
                                  !cs::cfg("CS_CHAIN_TABLES_IN_LDS");
        const size_t tbl_lds = (chain_global || bchain) ? (size_t)cstd::kHeadTailWords * 4 : tbl;  // (header + tail words: tsetup)
        const size_t lds1 = tbl_lds + tile_lds;
        if (lds1 > 150 * 1024) return -1;
        StreamArgs sa{};
        sa.in = view_of(col);
        sa.flags = d_unicode_flags();
        sa.L = tp.d;
        if (chain_global || bchain) sa.L.in_lds = 2;
        sa.repl = ptr<const uint8_t>(d_repl);
        sa.rb = rb;
        sa.maxrepl = maxrepl;
        Buf out_off = dev_alloc(sizeof(int64_t) * (rows + 1), s);
        Buf out_chars = dev_alloc((size_t)col->nbytes + (size_t)extra + 64, s);
        sa.out_off = ptr<int64_t>(out_off);
        sa.out_chars = ptr<uint8_t>(out_chars);
        sa.out_cap = col->nbytes + extra;
        const int64_t nsub1 = (rows + tc.R - 1) / tc.R;
        sa.rows_per_tile = tc.R;
        // [aggregates nsub1][error word + phase counters 16][tickets 128: sixteen counters][exclusive prefixes nsub1]
        // (+ kScanBatch windows of slack: the scanner wave fetches whole batches)
        const size_t status_bytes = sizeof(cstile::u64) * (2 * (size_t)nsub1 + 16 + 128 + 64 * cstile::kScanBatch);
        Buf status = dev_alloc(status_bytes, s);
        CS_HIP(hipMemsetAsync(status->p, 0, status_bytes, s));
        sa.status = ptr<cstile::u64>(status);
        sa.error = reinterpret_cast<unsigned*>(ptr<cstile::u64>(status) + nsub1);
        sa.tickets = ptr<cstile::u64>(status) + nsub1 + 16;
        sa.excl = ptr<cstile::u64>(status) + nsub1 + 16 + 128;
        sa.nsub = nsub1;
        sa.cap_in = cap;
        sa.cap_out = cap_out;
        sa.tbl_bytes = (int)(tbl_lds + gt_bytes + bits_lds);
        sa.bits = bits_form ? ptr<const int32_t>(re->d_bits) : nullptr;
        sa.bits_off = (int)(tbl_lds + gt_bytes);
        sa.bits_words = bits_form ? (int)re->bits.size() : 0;
        sa.bits_k = bits_form ? re->bits[1] : 0;
        sa.tmpl = static_cast<const csvm::BackrefTemplate*>(cs::g_backrefs_dev);
        sa.gtags = brefs ? ptr<const int32_t>(re->d_gtags) : nullptr;
        sa.gt_off = (int)tbl_lds;
        sa.gt_words = brefs && !bchain ? (int)re->gtags.size() : 0;
        sa.debug = cs::cfg_int("CS_TILE_DEBUG", 0);
        sa.outliers = outliers ? 1 : 0;
        sa.lit = literal ? cs::g_replace_literal : 0;
        sa.litn = literal ? cs::g_replace_literal_len : 0;
        const OddRows* od = holes_ok && !literal && !brefs ? odd_rows(col, s) : nullptr;
        const bool with_holes = od && od->count > 0;
        const unsigned hole_grid = with_holes ? (unsigned)std::min<int64_t>((od->count + 255) / 256, 2048) : 0;
        const size_t list_lds = tp.lds_bytes + (size_t)256 * kListRowBytes;  // (the tables and a row a thread)
        if (with_holes) {
          if (list_lds > 48 * 1024) {
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tdfa_replace_list<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)list_lds));
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tdfa_replace_list<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)list_lds));
          }
          if (!hole_lens) {
            hole_lens = dev_alloc(sizeof(int32_t) * (size_t)od->count, s);
            ProfScope ps("k_tdfa_replace_list", s);
            hipLaunchKernelGGL(k_tdfa_replace_list<false>, dim3(hole_grid), dim3(256), list_lds, s, src, tp.d, ptr<const int32_t>(od->list), od->count, ptr<const uint8_t>(d_repl), rb,
                               ptr<int32_t>(hole_lens), (const int64_t*)nullptr, (uint8_t*)nullptr);
            CS_HIP(hipGetLastError());
          }
          sa.hole_mask = ptr<const unsigned long long>(od->mask);
          sa.hole_first = ptr<const int64_t>(od->first);
          sa.hole_len = ptr<const int32_t>(hole_lens);
        }
        // (tables in LDS: a DFA beyond the LDS budget left this function above -- the forms with the tables in memory that are
        // not chain forms were never reached by a test and spilled 20-60 registers: out since round 6, kernel_coverage.txt)
        auto pick2 = [&](auto inplace, auto rescan, auto lng) {
          constexpr bool IP = decltype(inplace)::value, RS = decltype(rescan)::value, LG = decltype(lng)::value;
          return rb > 8 ? &k_tdfa_replace_stream<true, true, IP, RS, LG> : &k_tdfa_replace_stream<true, false, IP, RS, LG>;
        };
        // rows beyond the 96-byte candidate masks (and up to 255 bytes) take the sliding-window form
        const bool lng = tc.lng;
        auto pick = [&](auto inplace, auto rescan) {
          return lng ? pick2(inplace, rescan, std::true_type{}) : pick2(inplace, rescan, std::false_type{});
        };
        auto kern = growth == 0 ? pick(std::true_type{}, std::false_type{})
                                : roomy ? pick(std::false_type{}, std::true_type{}) : pick(std::false_type{}, std::false_type{});
        if (outliers) {  // (the forms that handle oversize sub-tiles: in place, or with the rescan assembly)
          constexpr int P6 = cstile::kPfChunks;
          if (growth == 0) kern = rb > 8 ? &k_tdfa_replace_stream<true, true, true, false, false, false, P6, false, false, true>
                                         : &k_tdfa_replace_stream<true, false, true, false, false, false, P6, false, false, true>;
          else kern = rb > 8 ? &k_tdfa_replace_stream<true, true, false, true, false, false, P6, false, false, true>
                             : &k_tdfa_replace_stream<true, false, false, true, false, false, P6, false, false, true>;
        } else if (wide_stream) {
          constexpr int P6 = cstile::kPfChunks;
          if (rb <= 8) {
            if (growth == 0) kern = lng ? &k_tdfa_replace_stream<true, false, true, false, true, false, P6, false, true> : &k_tdfa_replace_stream<true, false, true, false, false, false, P6, false, true>;
            else kern = lng ? &k_tdfa_replace_stream<true, false, false, true, true, false, P6, false, true> : &k_tdfa_replace_stream<true, false, false, true, false, false, P6, false, true>;
          } else {
            if (growth == 0) kern = lng ? &k_tdfa_replace_stream<true, true, true, false, true, false, P6, false, true> : &k_tdfa_replace_stream<true, true, true, false, false, false, P6, false, true>;
            else kern = lng ? &k_tdfa_replace_stream<true, true, false, true, true, false, P6, false, true> : &k_tdfa_replace_stream<true, true, false, true, false, false, P6, false, true>;
          }
        } else if (bchain)
          kern = cap <= 5 * 1024 ? &k_tdfa_replace_stream<false, false, false, true, false, true, 5, true, false, false, true>
                                 : &k_tdfa_replace_stream<false, false, false, true, false, true, cstile::kPfChunks, true, false, false, true>;
        else if (brefs)
          kern = cap <= 5 * 1024 ? &k_tdfa_replace_stream<true, false, false, true, false, true, 5, true> : &k_tdfa_replace_stream<true, false, false, true, false, true, cstile::kPfChunks, true>;
        else if (bits_form && chain_global)
          kern = rb > 8 ? &k_tdfa_replace_stream<false, true, false, true, false, true, cstile::kPfChunks, false, false, false, true, true>
                        : &k_tdfa_replace_stream<false, false, false, true, false, true, cstile::kPfChunks, false, false, false, true, true>;
        else if (bits_form)
          kern = rb > 8 ? &k_tdfa_replace_stream<true, true, false, true, false, true, cstile::kPfChunks, false, false, false, true, true>
                        : &k_tdfa_replace_stream<true, false, false, true, false, true, cstile::kPfChunks, false, false, false, true, true>;
        else if (chain_form && chain_global)
          kern = rb > 8 ? &k_tdfa_replace_stream<false, true, false, true, false, true, 5, false, false, false, true>
                        : &k_tdfa_replace_stream<false, false, false, true, false, true, 5, false, false, false, true>;
        else if (chain_form)
          kern = rb > 8 ? &k_tdfa_replace_stream<true, true, false, true, false, true, 5, false, false, false, true>
                        : &k_tdfa_replace_stream<true, false, false, true, false, true, 5, false, false, false, true>;
        else if (units && cap <= 5 * 1024 && with_holes)
          kern = rb > 8 ? &k_tdfa_replace_stream<true, true, false, true, false, true, 5> : &k_tdfa_replace_stream<true, false, false, true, false, true, 5>;
        else if (units && with_holes)
          kern = rb > 8 ? &k_tdfa_replace_stream<true, true, false, true, false, true> : &k_tdfa_replace_stream<true, false, false, true, false, true>;
        else if (units && cap <= 5 * 1024)
          kern = rb > 8 ? &k_tdfa_replace_stream<true, true, false, true, false, true, 5, false, false, false, false, false, false>
                        : &k_tdfa_replace_stream<true, false, false, true, false, true, 5, false, false, false, false, false, false>;
        else if (units)
          kern = rb > 8 ? &k_tdfa_replace_stream<true, true, false, true, false, true, cstile::kPfChunks, false, false, false, false, false, false>
                        : &k_tdfa_replace_stream<true, false, false, true, false, true, cstile::kPfChunks, false, false, false, false, false, false>;
        note_route(bits_form ? "bits" : bchain ? "brefs-chain" : brefs ? "brefs" : literal ? "literal" : wide_stream ? "wide" : units ? (((re->tdfa[30] >> 16) & 15) != 0 ? "chain" : "units") : "plain");
        if (with_holes) note_route_put_off();
        if (lds1 > 48 * 1024)
          CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds1));
        const unsigned grid = resident_grid(reinterpret_cast<const void*>(kern), lds1, (nsub1 + 3) / 4);
        if (cs::cfg("CS_STREAM_INFO"))
          fprintf(stderr, "replace stream: grid %u lds %zu (tables %zu, tile %d + %d) rows/tile %d units %d chain %d wide %d brefs %d roomy %d growth %d rb %d\n", grid, lds1, tbl_lds + gt_bytes, cap, cap_out,
                  tc.R, (int)units, bchain ? 3 : chain_form ? (chain_global ? 2 : 1) : 0, (int)wide_stream, (int)brefs, (int)roomy, (int)growth, rb);
#if defined(CS_PHASE_PROF)
        Buf tracebuf;
        const long long ntrace = (nsub1 >> 10) + 1;
        {
          unsigned long long* tp_ = nullptr;
          if (cs::cfg("CS_REPLACE_TRACE")) {
            tracebuf = dev_alloc(sizeof(unsigned long long) * 4 * ntrace, s);
            CS_HIP(hipMemsetAsync(tracebuf->p, 0, sizeof(unsigned long long) * 4 * ntrace, s));
            tp_ = ptr<unsigned long long>(tracebuf);
          }
          CS_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(cstile::g_tile_trace), &tp_, sizeof(tp_), 0, hipMemcpyHostToDevice, s));
        }
#endif
        {
          ProfScope ps("k_replace_re", s);
          hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds1, s, sa);
        }
        CS_HIP(hipGetLastError());
        int64_t* host = (int64_t*)pinned_scratch(24);
        CS_HIP(hipMemcpyAsync(host, ptr<int64_t>(out_off) + rows, 8, hipMemcpyDeviceToHost, s));
        CS_HIP(hipMemcpyAsync(host + 1, sa.error, 4, hipMemcpyDeviceToHost, s));
        CS_HIP(hipMemcpyAsync(host + 2, reinterpret_cast<unsigned long long*>(sa.error) + 14, 8, hipMemcpyDeviceToHost, s));
        CS_HIP(hipStreamSynchronize(s));
#if defined(CS_PHASE_PROF)
        {
          unsigned long long ph[12];
          CS_HIP(hipMemcpy(ph, reinterpret_cast<unsigned long long*>(sa.error) + 1, sizeof(ph), hipMemcpyDeviceToHost));
          if (tracebuf) {
            std::vector<unsigned long long> tr(4 * ntrace);
            CS_HIP(hipMemcpy(tr.data(), tracebuf->p, sizeof(unsigned long long) * 4 * ntrace, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (long long i = 0; i < ntrace; ++i)
              if (tr[4 * i]) t0 = std::min(t0, tr[4 * i]);
            fprintf(stderr, "trace (us since the first publish; every 1024th sub-tile): tile published scanned start-of-finish prefix-seen\n");
            for (long long i = 0; i < ntrace; i += std::max<long long>(1, ntrace / 40))
              fprintf(stderr, "trace %8lld %9.1f %9.1f %9.1f %9.1f\n", i << 10, (double)(tr[4 * i] - t0) / 100.0, (double)(tr[4 * i + 1] - t0) / 100.0,
                      (double)(tr[4 * i + 2] - t0) / 100.0, (double)(tr[4 * i + 3] - t0) / 100.0);
          }
          unsigned long long lb[4] = {0, 0, 0, 0};
          CS_HIP(hipMemcpyFromSymbol(lb, HIP_SYMBOL(cstile::g_lb_stats), sizeof(lb)));
          fprintf(stderr, "look-back (cumulative): %llu calls, %.2f windows per call, %.2f re-polls per call\n", lb[0], (double)lb[1] / (double)(lb[0] ? lb[0] : 1),
                  (double)lb[2] / (double)(lb[0] ? lb[0] : 1));
          const double waves = (double)grid * 4;
          fprintf(stderr, "stage split: wait+lds %.0f classify %.0f rest %.0f | ", ph[9] / waves / (nsub1 / waves), ph[10] / waves / (nsub1 / waves), ph[0] / waves / (nsub1 / waves));
          fprintf(stderr, "phase cycles/wave-iteration: stage %.0f scan %.0f (units: discovery %.0f rounds %.0f) wscan+publish %.0f finish_prev(offsets+flush) %.0f assemble %.0f tail %.0f lookback %.0f (iters/wave %.1f)\n",
                  ph[0] / waves / (nsub1 / waves), ph[1] / waves / (nsub1 / waves), ph[7] / waves / (nsub1 / waves), ph[8] / waves / (nsub1 / waves), ph[2] / waves / (nsub1 / waves),
                  ph[3] / waves / (nsub1 / waves), ph[4] / waves / (nsub1 / waves), ph[5] / waves / (nsub1 / waves), ph[6] / waves / (nsub1 / waves), nsub1 / waves);
        }
#endif
        const int err = (int)(uint32_t)host[1];
        if (err != 0 && cs::cfg("CS_DUMP_STATUS")) {  // development aid: where did the prefix chain stop?
          std::vector<cstile::u64> st((size_t)nsub1), ex((size_t)nsub1);
          CS_HIP(hipMemcpy(st.data(), sa.status, sizeof(cstile::u64) * nsub1, hipMemcpyDeviceToHost));
          CS_HIP(hipMemcpy(ex.data(), sa.excl, sizeof(cstile::u64) * nsub1, hipMemcpyDeviceToHost));
          unsigned long long tk[8];
          CS_HIP(hipMemcpy(tk, sa.tickets, sizeof(tk), hipMemcpyDeviceToHost));
          fprintf(stderr, "status dump: err %d nsub %lld grid %u ticket0 %llu\n", err, (long long)nsub1, grid, tk[0]);
          for (int64_t t = 0; t < nsub1 && t < 64; ++t)
            fprintf(stderr, "  tile %lld: agg flag %u val %llu | excl flag %u val %llu\n", (long long)t, (unsigned)(st[t] >> 62), (unsigned long long)(st[t] & cstile::kValMask),
                    (unsigned)(ex[t] >> 62), (unsigned long long)(ex[t] & cstile::kValMask));
        }
        if (err == 0 && with_holes) {  // the holes' bytes
          ProfScope ps("k_tdfa_replace_list", s);
          hipLaunchKernelGGL(k_tdfa_replace_list<true>, dim3(hole_grid), dim3(256), list_lds, s, src, tp.d, ptr<const int32_t>(od->list), od->count, ptr<const uint8_t>(d_repl), rb,
                             (int32_t*)nullptr, ptr<const int64_t>(out_off), ptr<uint8_t>(out_chars));
          CS_HIP(hipGetLastError());
          CS_HIP(hipStreamSynchronize(s));
        }
        if (err == 0) {  // (a launch whose error word is set never yields a column, measurement switches or not)
          o->offsets = out_off;
          o->chars = out_chars;
          o->nbytes = host[0];
          // column metadata as a by-product: the largest 64-row span is the largest sub-tile total the kernel saw; a
          // replacement no longer than the shortest match cannot lengthen a row (an upper bound serves: the numbers size
          // staging buffers and pick routes); plain input + plain replacement stays plain
          if (tc.R == 64) o->max_span64 = host[2];
          if (growth == 0 && col->max_row >= 0) o->max_row = col->max_row;
          if (col->high_sample == 0 && !brefs) {
            bool ascii = true;
            for (int i = 0; i < rb; ++i) ascii = ascii && (unsigned char)repl[i] < 0x80 && repl[i] != 0;
            if (ascii) o->high_sample = 0;
          }
          *out = holder.release();
          return 0;
        }
        return err;
      };
      // (replacements of 17 .. kMaxStreamRepl bytes ride the four-register variants, their text read from memory at assembly)
      // (... where a match is long enough for the row not to outgrow the out tile: a 19-byte replacement of one-digit matches
      // goes to the two-pass kernels at once instead of failing the single pass twice first)
      // (... or where the sample says matches are few: at most about three candidate bytes a row -- the first attempt is
      // provisioned for kMaxRec matches a row; `-` -> 35 bytes on the log lines, two or three dashes each: 58 ms on the two-pass
      // kernels.  A column that has more after all loses one launch before them)
      const bool few_matches = rb <= kMaxStreamRepl && minlen >= 1 && col->rows > 0 &&
                               candidate_share(re, col, s) * ((double)col->nbytes / (double)col->rows) <= (double)(kMaxRec - 1);
      if (lds <= 150 * 1024 && (rb <= 16 || (rb <= kMaxStreamRepl && rb <= 4 * minlen) || few_matches) && tc.R) {
        const bool roomy_first = growth > 0 && (minlen <= 2 || wide_stream || outliers || cs::cfg("CS_REPLACE_ROOMY"));
        // A growing replacement for a pattern of one or two bytes (`\d+` -> '<number>', `\d` -> '##'): provisioned for the worst
        // case the out tile is three times the in tile -- two workgroups a CU -- and the output twice the column (18 ms on the
        // C3 column where `\d+` -> '#' takes 5.9).  count_re first (2.4 ms) says how many matches the column and its busiest
        // tile hold: the out tile and the output are sized exactly: 17.9 -> 12.1 ms.
        // (where the worst case is within three times the in tile anyway -- `\d` -> '##' -- the count only costs: 15.6 -> 17.9 ms)
        if (roomy_first && rb >= 3 * minlen_p && tdfa && !wide_stream && !outliers && !cs::g_backrefs_dev && !cs::g_replace_plain_only && maxrepl < 0 && !empties && tp.d.in_lds &&
            !cs::cfg("CS_NO_COUNT_SIZING")) {
          Buf counts = dev_alloc(sizeof(int32_t) * (size_t)rows, s), stats = dev_alloc(16, s);
          scan<2>(col, re, nullptr, ptr<int32_t>(counts), 1, s, nullptr, "k_count_re");
          CS_HIP(hipMemsetAsync(stats->p, 0, 16, s));
          hipLaunchKernelGGL(k_match_stats, dim3(1024), dim3(256), 0, s, ptr<const int32_t>(counts), rows, tc.R, ptr<unsigned long long>(stats), reinterpret_cast<unsigned*>(ptr<unsigned long long>(stats) + 1));
          CS_HIP(hipGetLastError());
          unsigned long long* h = (unsigned long long*)pinned_scratch(16);
          CS_HIP(hipMemcpyAsync(h, stats->p, 16, hipMemcpyDeviceToHost, s));
          CS_HIP(hipStreamSynchronize(s));
          sized_total = (int64_t)h[0];
          sized_tilemax = (int64_t)(unsigned)h[1];
          note_route("");
        }
        int err = stream_attempt(roomy_first);
        if (err == 2 && !roomy_first) err = stream_attempt(true);  // only ran out of room: once more, roomier
        if (err == 0) return;
        if (err > 0) {  // (counted: cs_fallback_count)
          // what gave up, in words: the error word's bits and the column's shape (profiles/r04/soak.txt showed "error word 33"
          // without saying that it is the backrefs form meeting a sub-tile with more than 128 matches)
          char what[256];
          snprintf(what, sizeof(what), "%s (error word %d:%s%s%s%s%s%s; %lld rows, longest %lld bytes, largest 64-row span %lld)",
                   cs::g_backrefs_dev ? "replace_with_backrefs" : "replace_re", err, (err & 1) ? " launch abandoned" : "", (err & 2) ? " out of output room" : "",
                   (err & 4) ? " a sub-tile beyond the staging buffer" : "", (err & 8) ? " no prefix for a tile" : "", (err & 16) ? " the scanners timed out" : "",
                   (err & 32) ? " a sub-tile with more than 128 matches or a row handed over by the unit route (the template needs the unit route's masks)" : "",
                   (long long)col->rows, (long long)col->max_row, (long long)col->max_span64);
          note_fallback(what);
        }
      }
      // an oversize sub-tile or a look-back timeout: fall through to the two-pass kernels
    }
    if (cs::g_replace_plain_only) fail(CS_ERR_INTERNAL, "literal replace: single-pass kernel not applicable");  // cs_replace falls back to its own kernels
    Buf lens = dev_alloc(sizeof(int32_t) * col->rows, s);
    {
      ProfScope ps("k_replace_re_size", s);
      if (wide && tp.d.in_lds)
        hipLaunchKernelGGL((k_tdfa_replace_size<true, true>), dim3(tp.grid), dim3(256), tp.lds_bytes, s, src, tp.d, rb, maxrepl,
                           ptr<int32_t>(lens));
      else if (wide)
        hipLaunchKernelGGL((k_tdfa_replace_size<false, true>), dim3(tp.grid), dim3(256), 0, s, src, tp.d, rb, maxrepl,
                           ptr<int32_t>(lens));
      else if (tdfa && tp.d.in_lds)
        hipLaunchKernelGGL(k_tdfa_replace_size<true>, dim3(tp.grid), dim3(256), tp.lds_bytes, s, src, tp.d, rb, maxrepl,
                           ptr<int32_t>(lens));
      else if (tdfa)
        hipLaunchKernelGGL(k_tdfa_replace_size<false>, dim3(tp.grid), dim3(256), 0, s, src, tp.d, rb, maxrepl,
                           ptr<int32_t>(lens));
      else if (pl.small)
        hipLaunchKernelGGL((k_replace_re_size<true>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d, rb,
                           maxrepl, ptr<int32_t>(lens));
      else
        hipLaunchKernelGGL((k_replace_re_size<false>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d, rb,
                           maxrepl, ptr<int32_t>(lens));
    }
    CS_HIP(hipGetLastError());
    o->offsets = dev_alloc(sizeof(int64_t) * (col->rows + 1), s);
    LenMeta meta;
    o->nbytes = offsets_from_lengths(ptr<int32_t>(lens), col->rows, ptr<int64_t>(o->offsets), s, nullptr, &meta);
    meta.give(o);
    o->chars = dev_alloc((size_t)o->nbytes, s);
    {
      ProfScope ps("k_replace_re_write", s);
      if (wide && tp.d.in_lds)
        hipLaunchKernelGGL((k_tdfa_replace_write<true, true>), dim3(tp.grid), dim3(256), tp.lds_bytes, s, src, tp.d,
                           ptr<const uint8_t>(d_repl), rb, maxrepl, o->d_offsets(), ptr<uint8_t>(o->chars));
      else if (wide)
        hipLaunchKernelGGL((k_tdfa_replace_write<false, true>), dim3(tp.grid), dim3(256), 0, s, src, tp.d,
                           ptr<const uint8_t>(d_repl), rb, maxrepl, o->d_offsets(), ptr<uint8_t>(o->chars));
      else if (tdfa && tp.d.in_lds)
        hipLaunchKernelGGL(k_tdfa_replace_write<true>, dim3(tp.grid), dim3(256), tp.lds_bytes, s, src, tp.d,
                           ptr<const uint8_t>(d_repl), rb, maxrepl, o->d_offsets(), ptr<uint8_t>(o->chars));
      else if (tdfa)
        hipLaunchKernelGGL(k_tdfa_replace_write<false>, dim3(tp.grid), dim3(256), 0, s, src, tp.d,
                           ptr<const uint8_t>(d_repl), rb, maxrepl, o->d_offsets(), ptr<uint8_t>(o->chars));
      else if (pl.small)
        hipLaunchKernelGGL((k_replace_re_write<true>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d,
                           ptr<const uint8_t>(d_repl), rb, maxrepl, o->d_offsets(), ptr<uint8_t>(o->chars));
      else
        hipLaunchKernelGGL((k_replace_re_write<false>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d,
                           ptr<const uint8_t>(d_repl), rb, maxrepl, o->d_offsets(), ptr<uint8_t>(o->chars));
    }
    CS_HIP(hipGetLastError());
    CS_HIP(hipStreamSynchronize(s));  // d_repl / arena lifetime
    *out = holder.release();
  });
}

// The spans' bytes into the output columns (extract / findall): 64-row sub-tiles through LDS when the column's
// sub-tiles fit the staging buffer, else the thread-per-row kernel.
void write_spans(const cs_column* col, int ncols, const int32_t* begins, const int32_t* lens, const ExtractOut& eo, hipStream_t s) {
  const int64_t rows = col->rows;
  const TileChoice tc = choose_tile(col, s);
  ProfScope ps("k_extract_write", s);
  if (tc.R == 64 && !cs::cfg("CS_SPANS_ROWWISE")) {
    SpanWriteArgs a{};
    a.in = view_of(col);
    a.ncols = ncols;
    a.begins = begins;
    a.lens = lens;
    a.out = eo;
    a.nsub = (rows + 63) / 64;
    a.cap = tc.cap;
    const size_t lds = (size_t)4 * (2 * tc.cap + 64);
    if (lds > 48 * 1024)
      CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spans_write_tile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)std::min<long long>((a.nsub + 3) / 4, 256 * 16);
    hipLaunchKernelGGL(k_spans_write_tile, dim3(grid), dim3(256), lds, s, a);
  } else {
    hipLaunchKernelGGL(k_extract_write, dim3(blocks_for(rows)), dim3(256), 0, s, view_of(col), ncols, begins, lens, eo);
  }
}

// The output columns from PACKED spans and the scan kernel's tile totals (k_spans_write_tile2): no pass over the lengths.
void columns_from_packed_spans(const cs_column* col, int ncols, const uint32_t* spans, const int32_t* tile_tot, int cap, hipStream_t s,
                               std::vector<std::unique_ptr<cs_column>>& cols) {
  const int64_t rows = col->rows, nsub = (rows + 63) / 64;
  Buf base = dev_alloc(sizeof(int64_t) * (nsub + 1) * ncols, s);
  std::vector<int64_t> totals(ncols), largest(ncols);
  offsets_from_lengths_segmented(tile_tot, nsub, ncols, ptr<int64_t>(base), totals.data(), s, largest.data());
  SpanWrite2Args a{};
  a.in = view_of(col);
  a.ncols = ncols;
  a.spans = spans;
  a.tile_base = ptr<const int64_t>(base);
  a.nsub = nsub;
  a.cap = cap;
  bool off32 = !cs::cfg("CS_SPANS_OFF64");
  for (int k = 0; k < ncols; ++k) off32 = off32 && totals[k] < ((int64_t)1 << 31);
  a.off32 = off32 ? 1 : 0;
  for (int k = 0; k < ncols; ++k) {
    auto o = std::make_unique<cs_column>();
    o->rows = rows;
    o->nbytes = totals[k];
    if (off32) o->offsets32 = dev_alloc(sizeof(int32_t) * (rows + 1), s);
    else o->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    o->validity = dev_alloc(validity_bytes(rows), s);
    o->chars = dev_alloc((size_t)o->nbytes, s);
    // (metadata for free: the largest tile total is the column's largest 64-row span; a span is a piece of its row)
    o->max_span64 = largest[k];
    o->max_row = col->max_row >= 0 ? std::min<int64_t>(col->max_row, largest[k]) : largest[k];
    if (col->plain_bytes == 1) o->plain_bytes = 1;
    if (col->high_sample == 0) o->high_sample = 0;
    a.out.off[k] = off32 ? o->offsets32->p : o->offsets->p;
    a.out.valid[k] = ptr<uint8_t>(o->validity);
    a.out.chars[k] = ptr<uint8_t>(o->chars);
    cols.push_back(std::move(o));
  }
  const size_t lds = (size_t)4 * (2 * cap + 64);
  if (lds > 48 * 1024)
    CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spans_write_tile2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)std::min<long long>((nsub + 3) / 4, 256 * 64);  // (runs of consecutive tiles per wave; short ones share the tail out)
  {
    ProfScope ps("k_extract_write", s);
    hipLaunchKernelGGL(k_spans_write_tile2, dim3(grid), dim3(256), lds, s, a);
  }
  CS_HIP(hipGetLastError());
  CS_HIP(hipStreamSynchronize(s));  // `base` / span buffers
}

// NVStrings::extract(pattern, results) (NVStrings.h:682; extract.cu:69-151): one column per
// capture group; a pattern without groups, or an empty column, yields no columns.
int cs_extract(const cs_column* col, const cs_regex* cre, cs_stream stream, cs_column*** out_cols, int* ncols_out) {
  return guard([&] {
    if (!col || !cre || !out_cols || !ncols_out) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    hipStream_t s = S(stream);
    cs_regex* re = const_cast<cs_regex*>(cre);
    *out_cols = nullptr;
    *ncols_out = 0;
    const int groups = re->prog.num_groups;
    const int64_t rows = col->rows;
    if (groups <= 0 || rows == 0) return;
    if (groups > kMaxGroups) fail(CS_ERR_RANGE, "extract: more than 32 capture groups");
    upload(re, s);
    const int ninst = (int)re->prog.insts.size();
    Launch L{};
    L.image = ptr<const int32_t>(re->d_image);
    L.image_words = (int)re->image.size();
    L.slots = csvm::gvm_slots(ninst);
    const size_t img_bytes = (((size_t)L.image_words + 3) & ~size_t(3)) * 4;
    L.image_in_lds = img_bytes <= kLdsBudget / 2;
    const unsigned grid = (unsigned)std::min<int64_t>((rows + 255) / 256, 256 * 4);
    const bool dfa_groups = use_tdfa(re) && re->d_gtags && !cs::cfg("CS_EXTRACT_LISTS");
    Buf arena;  // thread lists of the list simulation (not needed when the DFA carries the group ranges)
    if (!dfa_groups) {
      arena = dev_alloc((size_t)grid * 256 * L.slots * 4, s);
      L.arena = ptr<uint32_t>(arena);
    }
    RowSrc src{view_of(col), d_unicode_flags(), col->nbytes + (col->chars && col->chars->capacity ? 64 : 0)};
    Buf begins, lens, spans, tile_tot;  // (begins / lens, or packed spans + tile totals from the scan stream kernel on 64-row tiles)
    const bool tdfa = use_tdfa(re);
    bool streamed = false, packed = false;
    if (dfa_groups && !cs::cfg("CS_REGEX_ROWWISE")) {  // rows staged through LDS tiles by the scan stream kernel
      TPlan tp = tplan(re, rows, s);
      const TileChoice tc = choose_tile(col, s);
      const int cap = tc.cap;
      const size_t gt_bytes = re->gtags.size() * 4 <= 16 * 1024 ? ((re->gtags.size() * 4 + 15) & ~size_t(15)) : 0;
      // (a chain pattern whose groups are runs of items, on a column whose sample is plain ASCII: the chain form -- the match
      // and the group ranges by mask arithmetic; it keeps the "equals x" bitmap and the unit form's layout)
      const bool chain_form = tp.d.in_lds && tc.R == 64 && !tc.lng && !cs::cfg("CS_SPANS_UNPACKED") && ((re->tdfa[30] >> 16) & 15) != 0 && ((re->tdfa[30] >> 20) & 1) != 0 &&
                              groups <= 4 && !cs::cfg("CS_NO_CHAIN_FORM") && !sample_has_high_bytes(col, s);
      // (the chain form with the tables in memory: a fourth workgroup per CU -- the kernel is built for 128 registers then)
      const bool chain_mem = chain_form && !cs::cfg("CS_CHAIN_TABLES_IN_LDS");
      const size_t tbl_x = chain_mem ? (size_t)cstd::kHeadTailWords * 4 : tp.lds_bytes;
      const size_t lds = chain_form ? tbl_x + gt_bytes + (size_t)(cap + 32 + 2 * ((cap >> 3) + 32) + kUnitQueue * 4 + 64 * 4 + 16) * 4
                                    : tp.lds_bytes + gt_bytes + (size_t)(cap + 32 + (cap >> 3) + 32 + cstd::Tdfa::kBackSteps * 64 + kMaxGroups * 4) * 4;
      if (tp.d.in_lds && tc.R && lds <= 150 * 1024) {
        packed = tc.R == 64 && !cs::cfg("CS_SPANS_UNPACKED");
        if (packed) {
          spans = dev_alloc(sizeof(uint32_t) * rows * groups, s);
          tile_tot = dev_alloc(sizeof(int32_t) * ((rows + 63) / 64) * groups, s);
        } else {
          begins = dev_alloc(sizeof(int32_t) * rows * groups, s);
          lens = dev_alloc(sizeof(int32_t) * rows * groups, s);
        }
        Buf cnt = dev_alloc(8, s);
        CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
        ScanStreamArgs sa{};
        sa.in = view_of(col);
        sa.flags = d_unicode_flags();
        sa.L = tp.d;
        sa.found = ptr<unsigned long long>(cnt);
        sa.nsub = (rows + tc.R - 1) / tc.R;
        sa.rows_per_tile = tc.R;
        sa.cap_in = cap;
        sa.tbl_bytes = (int)(tbl_x + gt_bytes);
        sa.gt_off = (int)tbl_x;
        if (chain_mem) sa.L.in_lds = 2;
        sa.gt_words = gt_bytes ? (int)re->gtags.size() : 0;
        sa.begins = ptr<int32_t>(begins);
        sa.lens = ptr<int32_t>(lens);
        sa.spans = ptr<uint32_t>(spans);
        sa.tile_tot = ptr<int32_t>(tile_tot);
        sa.ncols = groups;
        sa.gtags = ptr<const int32_t>(re->d_gtags);
        auto kern = tc.lng ? &k_tdfa_scan_stream<4, true, true> : &k_tdfa_scan_stream<4, true, false>;
        if (chain_form) kern = chain_mem ? &k_tdfa_scan_stream<4, false, false, true, true> : &k_tdfa_scan_stream<4, true, false, true, true>;
        if (lds > 48 * 1024)
          CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const unsigned sgrid = resident_grid(reinterpret_cast<const void*>(kern), lds, (sa.nsub + 3) / 4);
        ProfScope ps("k_extract_spans", s);
        hipLaunchKernelGGL(kern, dim3(sgrid), dim3(256), lds, s, sa);
        streamed = true;
      }
    }
    if (!streamed) {
      begins = dev_alloc(sizeof(int32_t) * rows * groups, s);
      lens = dev_alloc(sizeof(int32_t) * rows * groups, s);
    }
    if (streamed) {
    } else if (dfa_groups) {
      TPlan tp = tplan(re, rows, s);
      ProfScope ps("k_extract_spans", s);
      if (tp.d.in_lds)
        hipLaunchKernelGGL(k_extract_spans_dfa<true>, dim3(tp.grid), dim3(256), tp.lds_bytes, s, src, tp.d, ptr<const int32_t>(re->d_gtags),
                           groups, ptr<int32_t>(begins), ptr<int32_t>(lens));
      else
        hipLaunchKernelGGL(k_extract_spans_dfa<false>, dim3(tp.grid), dim3(256), 0, s, src, tp.d, ptr<const int32_t>(re->d_gtags), groups,
                           ptr<int32_t>(begins), ptr<int32_t>(lens));
    } else if (tdfa) {
      TPlan tp = tplan(re, rows, s);
      size_t fast_bytes = (size_t)256 * csvm::GroupVm<true>::kFastSlots * 4;
      const int use_fast = tp.lds_bytes + fast_bytes <= kLdsBudget;
      if (!use_fast) fast_bytes = 0;
      L.image_in_lds = tp.lds_bytes + fast_bytes + img_bytes <= kLdsBudget;
      const size_t lds = tp.lds_bytes + fast_bytes + (L.image_in_lds ? img_bytes : 0);
      ProfScope ps("k_extract_spans", s);
      if (tp.d.in_lds && ninst <= 64)
        hipLaunchKernelGGL((k_extract_spans_tdfa<true, true>), dim3(grid), dim3(256), lds, s, src, tp.d, L, groups, use_fast,
                           ptr<int32_t>(begins), ptr<int32_t>(lens));
      else if (tp.d.in_lds)
        hipLaunchKernelGGL((k_extract_spans_tdfa<true, false>), dim3(grid), dim3(256), lds, s, src, tp.d, L, groups, use_fast,
                           ptr<int32_t>(begins), ptr<int32_t>(lens));
      else if (ninst <= 64)
        hipLaunchKernelGGL((k_extract_spans_tdfa<false, true>), dim3(grid), dim3(256), lds, s, src, tp.d, L, groups, use_fast,
                           ptr<int32_t>(begins), ptr<int32_t>(lens));
      else
        hipLaunchKernelGGL((k_extract_spans_tdfa<false, false>), dim3(grid), dim3(256), lds, s, src, tp.d, L, groups, use_fast,
                           ptr<int32_t>(begins), ptr<int32_t>(lens));
    } else {
      ProfScope ps("k_extract_spans", s);
      if (ninst <= 64)
        hipLaunchKernelGGL((k_extract_spans<true>), dim3(grid), dim3(256), L.image_in_lds ? img_bytes : 0, s, src, L, groups,
                           ptr<int32_t>(begins), ptr<int32_t>(lens));
      else
        hipLaunchKernelGGL((k_extract_spans<false>), dim3(grid), dim3(256), L.image_in_lds ? img_bytes : 0, s, src, L, groups,
                           ptr<int32_t>(begins), ptr<int32_t>(lens));
    }
    CS_HIP(hipGetLastError());
    std::vector<std::unique_ptr<cs_column>> cols;
    if (packed) {
      columns_from_packed_spans(col, groups, ptr<const uint32_t>(spans), ptr<const int32_t>(tile_tot), choose_tile(col, s).cap, s, cols);
      cs_column** arr = (cs_column**)malloc(sizeof(cs_column*) * groups);
      if (!arr) fail(CS_ERR_ALLOC, "host allocation failed");
      for (int g = 0; g < groups; ++g) arr[g] = cols[g].release();
      *out_cols = arr;
      *ncols_out = groups;
      return;
    }
    ExtractOut eo{};
    for (int g = 0; g < groups; ++g) {
      auto o = std::make_unique<cs_column>();
      o->rows = rows;
      const int32_t* gl = ptr<int32_t>(lens) + (size_t)g * rows;
      o->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
      LenMeta meta;
      o->nbytes = offsets_and_validity_from_lengths(gl, rows, ptr<int64_t>(o->offsets), &o->validity, s, &meta);
      meta.give(o.get());
      o->chars = dev_alloc((size_t)o->nbytes, s);
      eo.off[g] = o->d_offsets();
      eo.chars[g] = ptr<uint8_t>(o->chars);
      cols.push_back(std::move(o));
    }
    write_spans(col, groups, ptr<int32_t>(begins), ptr<int32_t>(lens), eo, s);
    CS_HIP(hipGetLastError());
    CS_HIP(hipStreamSynchronize(s));  // arena / span buffers
    cs_column** arr = (cs_column**)malloc(sizeof(cs_column*) * groups);
    if (!arr) fail(CS_ERR_ALLOC, "host allocation failed");
    for (int g = 0; g < groups; ++g) arr[g] = cols[g].release();
    *out_cols = arr;
    *ncols_out = groups;
  });
}

// NVStrings::findall(pattern, results) (NVStrings.h:943; findall.cu:99-179): column k holds each
// row's k-th match (null where the row has fewer); no match in any row -> one all-null column.
int cs_findall(const cs_column* col, const cs_regex* cre, cs_stream stream, cs_column*** out_cols, int* ncols_out) {
  return guard([&] {
    if (!col || !cre || !out_cols || !ncols_out) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    hipStream_t s = S(stream);
    cs_regex* re = const_cast<cs_regex*>(cre);
    *out_cols = nullptr;
    *ncols_out = 0;
    const int64_t rows = col->rows;
    if (rows == 0) return;
    auto finish = [&](std::vector<std::unique_ptr<cs_column>>& cols) {
      cs_column** arr = (cs_column**)malloc(sizeof(cs_column*) * cols.size());
      if (!arr) fail(CS_ERR_ALLOC, "host allocation failed");
      for (size_t k = 0; k < cols.size(); ++k) arr[k] = cols[k].release();
      *out_cols = arr;
      *ncols_out = (int)cols.size();
    };
    std::vector<std::unique_ptr<cs_column>> cols;
    Buf dmax = dev_alloc(8, s);
    int* hmax = (int*)pinned_scratch(8);
    auto read_max = [&]() {
      CS_HIP(hipMemcpyAsync(hmax, dmax->p, 4, hipMemcpyDeviceToHost, s));
      CS_HIP(hipStreamSynchronize(s));
      return hmax[0];
    };
    RowSrc src{view_of(col), d_unicode_flags(), col->nbytes + (col->chars && col->chars->capacity ? 64 : 0)};
    Buf begins, lens, spans, tile_tot;
    Plan pl{};
    int ncols = -1;  // unknown
    bool streamed = false, packed = false;
    int packed_cap = 0;
    // The scan stream kernel reports spans and the largest match count in ONE pass when the rows hold at most
    // kProvisional matches (the span arrays are laid out [k * rows + row], so unused columns cost only memory);
    // a column with busier rows is scanned a second time with the exact column count.
    constexpr int kProvisional = 4;
    if (use_tdfa(re) && !cs::cfg("CS_REGEX_ROWWISE")) {
      TPlan tp = tplan(re, rows, s);
      const TileChoice tc = choose_tile(col, s);
      const int cap = tc.cap;
      // the unit scan where the tagged DFA offers the decomposition (as count_re)
      const bool units = ((re->tdfa[31] & 1) != 0 || ((re->tdfa[30] >> 16) & 15) != 0) && !tc.lng && tc.R == 64 && !cs::cfg("CS_NO_UNITS");
      size_t lds = tp.lds_bytes + (size_t)(cap + 32 + (cap >> 3) + 32 + (units ? (cap >> 3) + 32 + kUnitQueue * 4 + 64 * 4 + 16 : 0)) * 4;
      const bool chain_scan = units && ((re->tdfa[30] >> 16) & 15) != 0 && !sample_has_high_bytes(col, s) && !cs::cfg("CS_NO_CHAIN_FORM");
      const bool chain_global = chain_scan && tp.d.in_lds && lds > 40 * 1024 && lds - tp.lds_bytes + cstd::kHeadTailWords * 4 <= 40 * 1024 && !cs::cfg("CS_CHAIN_TABLES_IN_LDS");  // (as in count_re)
      const size_t scan_tbl = chain_global ? (size_t)cstd::kHeadTailWords * 4 : tp.lds_bytes;
      if (chain_global) lds -= tp.lds_bytes - scan_tbl;
      if (tp.d.in_lds && tc.R && lds <= 150 * 1024) {
        Buf hits = dev_alloc(8, s);
        CS_HIP(hipMemsetAsync(hits->p, 0, 8, s));
        ScanStreamArgs sa{};
        sa.in = view_of(col);
        sa.flags = d_unicode_flags();
        sa.L = tp.d;
        sa.found = ptr<unsigned long long>(hits);  // (hit counter: not used here)
        sa.nsub = (rows + tc.R - 1) / tc.R;
        sa.rows_per_tile = tc.R;
        sa.cap_in = cap;
        sa.tbl_bytes = (int)scan_tbl;
        if (chain_global) sa.L.in_lds = 2;
        sa.maxp = ptr<int>(dmax);
        auto kern = tc.lng ? &k_tdfa_scan_stream<3, true, true> : &k_tdfa_scan_stream<3, true, false>;
        if (units) kern = &k_tdfa_scan_stream<3, true, false, true>;
        if (chain_scan) kern = &k_tdfa_scan_stream<3, true, false, true, true>;  // (a chain pattern on a column whose sample is plain ASCII)
        if (chain_global) kern = &k_tdfa_scan_stream<3, false, false, true, true>;
        if (lds > 48 * 1024)
          CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const unsigned grid = resident_grid(reinterpret_cast<const void*>(kern), lds, (sa.nsub + 3) / 4);
        int width = kProvisional;
        for (int pass = 0; pass < 2; ++pass) {
          // (the provisional pass on 64-row tiles leaves packed spans and tile totals: k_spans_write_tile2 needs no pass
          // over the lengths; a second pass -- rows with more than kProvisional matches -- writes begins / lens)
          packed = tc.R == 64 && width <= kMaxGroups && (pass == 0 || !cs::cfg("CS_SPANS_EXACT_UNPACKED")) && !cs::cfg("CS_SPANS_UNPACKED");
          if (packed) {
            spans = dev_alloc(sizeof(uint32_t) * rows * width, s);
            tile_tot = dev_alloc(sizeof(int32_t) * ((rows + 63) / 64) * width, s);
          } else {
            begins = dev_alloc(sizeof(int32_t) * rows * width, s);
            lens = dev_alloc(sizeof(int32_t) * rows * width, s);
          }
          CS_HIP(hipMemsetAsync(dmax->p, 0, 8, s));
          sa.begins = packed ? nullptr : ptr<int32_t>(begins);
          sa.lens = packed ? nullptr : ptr<int32_t>(lens);
          sa.spans = packed ? ptr<uint32_t>(spans) : nullptr;
          sa.tile_tot = packed ? ptr<int32_t>(tile_tot) : nullptr;
          sa.ncols = width;
          {
            ProfScope ps("k_findall_spans", s);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, sa);
          }
          CS_HIP(hipGetLastError());
          ncols = read_max();
          if (ncols <= width) break;
          width = ncols;  // busier rows than provisioned for: once more, exactly
        }
        streamed = true;
        packed_cap = cap;
      }
    }
    if (!streamed) {
      // matches per row (the count_re kernels), their maximum = number of columns
      Buf counts = dev_alloc(sizeof(int32_t) * rows, s);
      scan<2>(col, re, nullptr, ptr<int32_t>(counts), 1, s, nullptr, "k_count_re");
      CS_HIP(hipMemsetAsync(dmax->p, 0, 8, s));
      hipLaunchKernelGGL(k_max_i32, dim3((unsigned)std::min<int64_t>((rows + 255) / 256, 2048)), dim3(256), 0, s, ptr<int32_t>(counts), rows,
                         ptr<int>(dmax));
      ncols = read_max();
    }
    if (ncols == 0) {  // findall.cu:149-151
      cols.emplace_back(make_all_null(rows, s));
      finish(cols);
      return;
    }
    if (streamed && packed) {
      columns_from_packed_spans(col, ncols, ptr<const uint32_t>(spans), ptr<const int32_t>(tile_tot), packed_cap, s, cols);
      finish(cols);
      return;
    }
    if (!streamed) {
      begins = dev_alloc(sizeof(int32_t) * rows * ncols, s);
      lens = dev_alloc(sizeof(int32_t) * rows * ncols, s);
    }
    if (!streamed) {
      ProfScope ps("k_findall_spans", s);
      if (use_tdfa(re)) {
        TPlan tp = tplan(re, rows, s);
        if (tp.d.in_lds)
          hipLaunchKernelGGL(k_findall_spans_tdfa<true>, dim3(tp.grid), dim3(256), tp.lds_bytes, s, src, tp.d, ncols, ptr<int32_t>(begins),
                             ptr<int32_t>(lens));
        else
          hipLaunchKernelGGL(k_findall_spans_tdfa<false>, dim3(tp.grid), dim3(256), 0, s, src, tp.d, ncols, ptr<int32_t>(begins),
                             ptr<int32_t>(lens));
      } else {
        pl = plan(re, rows, s);
        if (pl.small)
          hipLaunchKernelGGL((k_findall_spans<true>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d, ncols,
                             ptr<int32_t>(begins), ptr<int32_t>(lens));
        else
          hipLaunchKernelGGL((k_findall_spans<false>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d, ncols,
                             ptr<int32_t>(begins), ptr<int32_t>(lens));
      }
    }
    CS_HIP(hipGetLastError());
    // columns, kMaxGroups at a time through the shared write kernel
    for (int k0 = 0; k0 < ncols; k0 += kMaxGroups) {
      const int nk = std::min(kMaxGroups, ncols - k0);
      ExtractOut eo{};
      for (int k = 0; k < nk; ++k) {
        auto o = std::make_unique<cs_column>();
        o->rows = rows;
        const int32_t* gl = ptr<int32_t>(lens) + (size_t)(k0 + k) * rows;
        o->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
        LenMeta meta;
        o->nbytes = offsets_and_validity_from_lengths(gl, rows, ptr<int64_t>(o->offsets), &o->validity, s, &meta);
        meta.give(o.get());
        o->chars = dev_alloc((size_t)o->nbytes, s);
        eo.off[k] = o->d_offsets();
        eo.chars[k] = ptr<uint8_t>(o->chars);
        cols.push_back(std::move(o));
      }
      write_spans(col, nk, ptr<int32_t>(begins) + (size_t)k0 * rows, ptr<int32_t>(lens) + (size_t)k0 * rows, eo, s);
    }
    CS_HIP(hipGetLastError());
    CS_HIP(hipStreamSynchronize(s));
    finish(cols);
  });
}

// NVStrings::replace_with_backrefs(pattern, repl) (NVStrings.h:788; replace_backref.cu:128-207).
// repl NULL -> a column of all nulls, as the reference returns.  A pattern that can match the empty
// string is refused (CS_ERR_INVALID_ARG): the reference does not terminate on it (replace_backref.cu:112).
int cs_replace_with_backrefs(const cs_column* col, const cs_regex* cre, const char* repl, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !cre || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    require_device();
    hipStream_t s = S(stream);
    cs_regex* re = const_cast<cs_regex*>(cre);
    if (re->empty_pattern) fail(CS_ERR_INVALID_ARG, "nvstrings::replace_with_backrefs parameter cannot be null or empty");
    const int64_t rows = col->rows;
    if (rows == 0 || !repl) {
      *out = make_all_null(rows, s);
      return;
    }
    // backref.h:31-57: a backslash followed by digits is a reference
    std::string text;
    csvm::BackrefTemplate t{};
    for (const char* p = repl; *p;) {
      if (*p == '\\' && p[1] >= '0' && p[1] <= '9') {
        const char* q = p + 1;
        while (*q >= '0' && *q <= '9') ++q;
        if (t.nrefs == csvm::BackrefTemplate::kMaxRefs) fail(CS_ERR_RANGE, "replace_with_backrefs: more than 16 references in the template");
        t.idx[t.nrefs] = atoi(p + 1);
        t.pos[t.nrefs] = (int)text.size();
        ++t.nrefs;
        p = q;
      } else {
        text.push_back(*p++);
      }
    }
    t.bytes = (int)text.size();
    t.groups = re->prog.num_groups;
    upload(re, s);
    const bool dfa = use_tdfa(re) && (re->d_gtags || re->prog.num_groups == 0);
    if (csrx::min_match_chars(re->prog) == 0) fail(CS_ERR_INVALID_ARG, "replace_with_backrefs: the pattern matches the empty string");
    Buf d_text = dev_alloc(text.size() + 1, s);
    CS_HIP(hipMemcpyAsync(d_text->p, text.c_str(), text.size() + 1, hipMemcpyHostToDevice, s));
    t.text = ptr<const uint8_t>(d_text);
    // First choice: the single-pass replace kernel in its backrefs form (unit scan, one group run per match, coalesced
    // output).  It declines patterns without the unit decomposition, more than four groups, long rows; a launch that
    // runs out of output room or meets a row the unit route hands over reports it, and the two-pass form below runs.
    bool packs = t.bytes <= 255;  // (the kernel keeps the references' group numbers and positions packed in registers)
    for (int j = 0; j < t.nrefs; ++j) packs = packs && t.idx[j] <= 15 && t.pos[j] <= 255;
    if (dfa && packs && re->prog.num_groups >= 1 && !cs::cfg("CS_BACKREFS_TWO_PASS")) {
      Buf d_t = dev_alloc(sizeof(csvm::BackrefTemplate), s);
      CS_HIP(hipMemcpyAsync(d_t->p, &t, sizeof(t), hipMemcpyHostToDevice, s));
      CS_HIP(hipStreamSynchronize(s));  // (`t` lives on this stack frame)
      cs::g_backrefs_dev = d_t->p;
      cs::g_backrefs_text_bytes = t.bytes;
      cs::g_backrefs_host = &t;
      cs::g_replace_plain_only = 1;  // (single-pass kernel or nothing)
      cs_column* fast = nullptr;
      const int rc = cs_replace_re(col, re, "", -1, stream, &fast);
      cs::g_backrefs_dev = nullptr;
      cs::g_backrefs_text_bytes = 0;
      cs::g_backrefs_host = nullptr;
      cs::g_replace_plain_only = 0;
      if (rc == CS_OK) {
        *out = fast;
        return;
      }
    }
    BackrefArgs a{};
    a.src = RowSrc{view_of(col), d_unicode_flags(), col->nbytes + (col->chars && col->chars->capacity ? 64 : 0)};
    a.gtags = ptr<const int32_t>(re->d_gtags);
    a.t = t;
    const int ninst = (int)re->prog.insts.size();
    unsigned grid = (unsigned)std::min<int64_t>((rows + 255) / 256, 256 * 8);
    size_t lds = 0;
    Buf arena;
    if (dfa) {
      TPlan tp = tplan(re, rows, s);
      a.TL = tp.d;
      lds = tp.lds_bytes;
    } else {
      a.L.image = ptr<const int32_t>(re->d_image);
      a.L.image_words = (int)re->image.size();
      a.L.slots = csvm::gvm_slots(ninst);
      grid = std::min(grid, 1024u);
      arena = dev_alloc((size_t)grid * 256 * a.L.slots * 4, s);
      a.L.arena = ptr<uint32_t>(arena);
    }
    auto launch = [&](bool write, int32_t* lens, const int64_t* off, uint8_t* chars) {
      const bool small = ninst <= 64;
#define CS_BR(D_, I_, S_)                                                                                        \
  do {                                                                                                           \
    if (write) hipLaunchKernelGGL((k_backrefs<D_, I_, S_, true>), dim3(grid), dim3(256), lds, s, a, lens, off, chars);  \
    else hipLaunchKernelGGL((k_backrefs<D_, I_, S_, false>), dim3(grid), dim3(256), lds, s, a, lens, off, chars);       \
  } while (0)
      if (dfa && a.TL.in_lds) CS_BR(true, true, true);
      else if (dfa) CS_BR(true, false, true);
      else if (small) CS_BR(false, false, true);
      else CS_BR(false, false, false);
#undef CS_BR
      CS_HIP(hipGetLastError());
    };
    // rows staged through LDS tiles by the scan stream kernel when the DFA carries the groups and the tiles fit
    ScanStreamArgs sa{};
    Buf d_tmpl;
    size_t slds = 0;
    unsigned sgrid = 0;
    bool lng = false, stream = false;
    if (dfa && a.TL.in_lds && !cs::cfg("CS_REGEX_ROWWISE")) {
      const TileChoice tc = choose_tile(col, s);
      const size_t gt_bytes = (!re->gtags.empty() && re->gtags.size() * 4 <= 16 * 1024) ? ((re->gtags.size() * 4 + 15) & ~size_t(15)) : 0;
      slds = lds + gt_bytes + (size_t)(tc.cap + 32 + (tc.cap >> 3) + 32) * 4;
      if (tc.R && slds <= 150 * 1024) {
        sa.gt_off = (int)lds;
        sa.gt_words = gt_bytes ? (int)re->gtags.size() : 0;
        stream = true;
        lng = tc.lng;
        sa.in = view_of(col);
        sa.flags = d_unicode_flags();
        sa.L = a.TL;
        sa.nsub = (rows + tc.R - 1) / tc.R;
        sa.rows_per_tile = tc.R;
        sa.cap_in = tc.cap;
        sa.tbl_bytes = (int)(lds + gt_bytes);
        sa.gtags = a.gtags;
        d_tmpl = dev_alloc(sizeof(csvm::BackrefTemplate), s);
        CS_HIP(hipMemcpyAsync(d_tmpl->p, &t, sizeof(t), hipMemcpyHostToDevice, s));
        sa.tmpl = ptr<const csvm::BackrefTemplate>(d_tmpl);
      }
    }
    Buf scnt;
    auto launch_stream = [&](bool write) {
      if (!scnt) {
        scnt = dev_alloc(8, s);
        CS_HIP(hipMemsetAsync(scnt->p, 0, 8, s));
        sa.found = ptr<unsigned long long>(scnt);
      }
      const void* kern = write ? (lng ? (const void*)&k_tdfa_scan_stream<6, true, true> : (const void*)&k_tdfa_scan_stream<6, true, false>)
                               : (lng ? (const void*)&k_tdfa_scan_stream<5, true, true> : (const void*)&k_tdfa_scan_stream<5, true, false>);
      if (slds > 48 * 1024) CS_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
      sgrid = resident_grid(kern, slds, (sa.nsub + 3) / 4);
      void* args[] = {&sa};
      CS_HIP(hipLaunchKernel(kern, dim3(sgrid), dim3(256), args, slds, s));
    };
    auto o = std::make_unique<cs_column>();
    o->rows = rows;
    o->validity = col->validity;
    o->null_count = col->null_count;
    Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
    {
      ProfScope ps("k_backrefs_size", s);
      if (stream) {
        sa.out32 = ptr<int32_t>(lens);
        launch_stream(false);
      } else {
        launch(false, ptr<int32_t>(lens), nullptr, nullptr);
      }
    }
    o->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    LenMeta meta;
    o->nbytes = offsets_from_lengths(ptr<int32_t>(lens), rows, ptr<int64_t>(o->offsets), s, nullptr, &meta);
    meta.give(o.get());
    o->chars = dev_alloc((size_t)o->nbytes, s);
    {
      ProfScope ps("k_backrefs_write", s);
      if (stream) {
        sa.out_off = o->d_offsets();
        sa.out_chars = ptr<uint8_t>(o->chars);
        launch_stream(true);
      } else {
        launch(true, nullptr, o->d_offsets(), ptr<uint8_t>(o->chars));
      }
    }
    CS_HIP(hipStreamSynchronize(s));
    *out = o.release();
  });
}


// NVStrings::replace_re(patterns, repls) (NVStrings.h:777; replace_multi.cu:110-189)
int cs_replace_re_multi(const cs_column* col, const cs_regex* const* res, int npatterns, const cs_column* repls, cs_stream stream,
                        cs_column** out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    if (npatterns <= 0 || !res || !repls || repls->rows == 0)
      fail(CS_ERR_INVALID_ARG, "replace_re patterns and repls parameters cannot be empty");
    if (repls->rows > 1 && repls->rows != npatterns)
      fail(CS_ERR_INVALID_ARG, "replace_re patterns and repls must have the same number of strings");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = col->rows;
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    std::vector<MultiProg> progs;
    bool all_dfa = !cs::cfg("CS_REGEX_NO_TDFA");
    int max_inst = 0;
    for (int i = 0; i < npatterns; ++i) {
      cs_regex* re = const_cast<cs_regex*>(res[i]);
      if (!re || re->empty_pattern) continue;  // (a null pattern is skipped, replace_multi.cu:125-127)
      upload(re, s);
      if (csrx::min_match_chars(re->prog) == 0) fail(CS_ERR_INVALID_ARG, "replace_re: a pattern matches the empty string");
      MultiProg mp{};
      mp.tdfa = ptr<const int32_t>(re->d_tdfa);
      mp.image = ptr<const int32_t>(re->d_image);
      for (int k = 0; k < 4; ++k) mp.first[k] = 0xFFFFFFFFu;
      if (!re->tdfa.empty() && re->tdfa[16] > 0)  // idle states exist: the candidate bitmap says which bytes leave them
        for (int k = 0; k < 4; ++k) mp.first[k] = (uint32_t)re->tdfa[21 + k];
      all_dfa = all_dfa && use_tdfa(re);  // (a program of more than four threads takes the list simulator here)
      max_inst = std::max(max_inst, (int)re->prog.insts.size());
      progs.push_back(mp);
      if (repls->rows > 1 && (int)progs.size() - 1 != i) fail(CS_ERR_INVALID_ARG, "replace_re: a null pattern among several replacements");
    }
    if (progs.empty()) fail(CS_ERR_INVALID_ARG, "replace_re invalid patterns");
    Buf d_progs = dev_alloc(sizeof(MultiProg) * progs.size(), s);
    CS_HIP(hipMemcpyAsync(d_progs->p, progs.data(), sizeof(MultiProg) * progs.size(), hipMemcpyHostToDevice, s));
    MultiArgs a{};
    a.src = RowSrc{view_of(col), d_unicode_flags(), col->nbytes};
    a.progs = ptr<const MultiProg>(d_progs);
    a.nprogs = (int)progs.size();
    a.repls = view_of(repls);
    unsigned grid = (unsigned)std::min<int64_t>((rows + 255) / 256, 256 * 8);
    Buf arena;
    if (!all_dfa) {
      a.slots = 6 * max_inst + (max_inst + 31) / 32 + 2;
      grid = std::min(grid, 1024u);
      arena = dev_alloc((size_t)grid * 256 * a.slots * 4, s);
      a.arena = ptr<uint32_t>(arena);
    }
    Buf bad = dev_alloc(sizeof(unsigned), s);
    CS_HIP(hipMemsetAsync(bad->p, 0, sizeof(unsigned), s));
    Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
    if (all_dfa) hipLaunchKernelGGL((k_multi_replace<true, false>), dim3(grid), dim3(256), 0, s, a, ptr<int32_t>(lens), (const int64_t*)nullptr, (uint8_t*)nullptr, ptr<unsigned>(bad));
    else hipLaunchKernelGGL((k_multi_replace<false, false>), dim3(grid), dim3(256), 0, s, a, ptr<int32_t>(lens), (const int64_t*)nullptr, (uint8_t*)nullptr, ptr<unsigned>(bad));
    CS_HIP(hipGetLastError());
    auto o = std::make_unique<cs_column>();
    o->rows = rows;
    o->validity = col->validity;
    o->null_count = col->null_count;
    o->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    LenMeta meta;
    o->nbytes = offsets_from_lengths(ptr<int32_t>(lens), rows, ptr<int64_t>(o->offsets), s, nullptr, &meta);
    meta.give(o.get());
    unsigned* hb = (unsigned*)pinned_scratch(sizeof(unsigned));
    CS_HIP(hipMemcpyAsync(hb, bad->p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (*hb) fail(CS_ERR_INVALID_ARG, "replace_re: a pattern matched the empty string");
    o->chars = dev_alloc((size_t)o->nbytes, s);
    if (all_dfa) hipLaunchKernelGGL((k_multi_replace<true, true>), dim3(grid), dim3(256), 0, s, a, (int32_t*)nullptr, o->d_offsets(), ptr<uint8_t>(o->chars), ptr<unsigned>(bad));
    else hipLaunchKernelGGL((k_multi_replace<false, true>), dim3(grid), dim3(256), 0, s, a, (int32_t*)nullptr, o->d_offsets(), ptr<uint8_t>(o->chars), ptr<unsigned>(bad));
    CS_HIP(hipGetLastError());
    CS_HIP(hipStreamSynchronize(s));
    *out = o.release();
  });
}

}  // extern "C"
