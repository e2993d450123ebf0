// VIRTUAL ROWS: a column whose rows exceed the 96-bit masks, seen as a column of pieces that fit them.
//
// The fast regex forms -- chain arithmetic, the bit-parallel form, the unit scan (cs_regex.hip) -- hold a row in three
// 32-bit mask words: rows of at most 92-93 bytes.  BASELINE.json's C5 column (tweet-like rows of 40-150 bytes) left them
// for the long-row automaton form: 3-20 times the time per byte (profiles/r06/c5regex.jsonl).  The reference's functors take
// any row length (regexec.inl:204-442, custring_view.cuh:36-42).  Instead of widening every mask, the COLUMN is narrowed:
// each row longer than 92 bytes is cut BEHIND a space, tab, line feed or carriage return into pieces of at most 92 bytes,
// and the pieces are the rows of a second column over the SAME chars buffer (new offsets, nothing copied).  For a program
// whose tagged DFA says those four bytes are safe cuts (regex_tdfa.cpp, header word 31 bit 25: whatever state consumes one
// keeps nothing, and a scan begun behind it behaves as one begun at a row's first byte) no match spans a cut and no piece
// can tell itself from a row (on well-formed text without NUL bytes: the executor ends a row's scan at a NUL and lets a lead
// byte swallow what follows it -- cs_regex.hip: pieces_for asks the column's `plain_bytes`), so
//   contains_re(row) = OR of contains_re(piece),  count_re(row) = sum of count_re(piece),
//   replace_re(row)  = the concatenation of replace_re(piece)  -- i.e. the pieces' output chars ARE the rows' output chars,
//                      and a row's output offset is the output offset of its first piece.
// The view depends on the column alone (not on the program), is built once -- two thread-per-row passes (the second reads the
// first's cut records, not the chars) and a scan -- and is
// kept on the (immutable) column like its other metadata; a column with a row that has no cut byte within 92 bytes has no
// view, and says so from then on.
#include <hip/hip_runtime.h>

#include <mutex>

#include "cs_internal.h"
#include "device_utils.h"

using namespace cs;
using namespace csdev;

namespace {
constexpr int kPiece = 92;  // bytes: with the row's start inside its first aligned word (<= 3) the masks hold 96

__device__ __forceinline__ bool cut_byte(uint8_t b) { return b == 32 || b == 9 || b == 10 || b == 13; }
// index of the last cut byte in chars[lo, hi), or -1: eight bytes a step from the back (one unaligned load; the cut bytes by
// SWAR), single bytes where the eight would reach beyond the buffer
__device__ __forceinline__ long long last_cut(const uint8_t* __restrict__ chars, long long lo, long long hi, long long nbytes) {
  long long p = hi;
  while (p > lo) {
    const long long q = p - 8 > lo ? p - 8 : lo;
    if (q + 8 > nbytes) {
      for (long long i = p - 1; i >= q; --i)
        if (cut_byte(chars[i])) return i;
    } else {
      unsigned long long w;
      __builtin_memcpy(&w, chars + q, 8);
      auto zero_bytes = [](unsigned long long x) { return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull; };
      unsigned long long m = zero_bytes(w ^ 0x2020202020202020ull) | zero_bytes(w ^ 0x0909090909090909ull) | zero_bytes(w ^ 0x0A0A0A0A0A0A0A0Aull) |
                             zero_bytes(w ^ 0x0D0D0D0D0D0D0D0Dull);
      const int take = (int)(p - q);  // bytes of the word inside [q, p)
      if (take < 8) m &= (1ull << (8 * take)) - 1ull;
      if (m) return q + ((63 - __builtin_clzll(m)) >> 3);
    }
    p = q;
  }
  return -1;
}
// pieces per row (1 for a null row, an empty row and every row within kPiece bytes) and the starts of a row's second to fourth
// piece relative to the row's (the second pass then does not read the chars again); *impossible is raised by a row with a
// stretch of kPiece bytes without a cut byte
__global__ void k_virt_count(ColView in, long long nbytes, int32_t* __restrict__ pieces, uint4* __restrict__ cuts, int* __restrict__ impossible) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  int n = 1;
  uint32_t c[3] = {0, 0, 0};
  const long long b = in.offsets[r], e = in.offsets[r + 1];
  if (e - b > kPiece && row_is_valid(in.validity, r)) {
    long long s = b;
    while (e - s > kPiece) {
      const long long p = last_cut(in.chars, s, s + kPiece, nbytes);
      if (p < 0) {
        *impossible = 1;
        break;
      }
      s = p + 1;
      if (n <= 3) c[n - 1] = (uint32_t)(s - b);
      ++n;
    }
  }
  pieces[r] = n;
  cuts[r] = make_uint4((uint32_t)n, c[0], c[1], c[2]);
}
// the pieces' offsets (absolute positions in the shared chars buffer) and, for a column with null rows, -1 / 0 per piece
__global__ void k_virt_write(ColView in, long long nbytes, const int64_t* __restrict__ first, const uint4* __restrict__ cuts, int64_t* __restrict__ voff,
                             int32_t* __restrict__ vnull) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  int64_t v = first[r];
  const uint4 c = cuts[r];
  const long long b = in.offsets[r], e = in.offsets[r + 1];
  const int n = (int)c.x;
  voff[v] = b;
  if (vnull) vnull[v] = (n > 1 || row_is_valid(in.validity, r)) ? 0 : -1;
  if (n > 1) {
    voff[v + 1] = b + c.y;
    if (n > 2) voff[v + 2] = b + c.z;
    if (n > 3) voff[v + 3] = b + c.w;
    if (vnull)
      for (int k = 1; k < n; ++k) vnull[v + k] = 0;
    if (n > 4) {  // (a row of more than four pieces: the cuts behind the fourth's start are found again)
      long long s = b + c.w;
      v += 3;
      while (e - s > kPiece) {
        const long long p = last_cut(in.chars, s, s + kPiece, nbytes);
        if (p < 0) break;  // (k_virt_count raised the flag: the view is discarded)
        s = p + 1;
        ++v;
        voff[v] = s;
      }
    }
  }
  if (r == in.rows - 1) voff[first[in.rows]] = e;
}
// a row's result out of its pieces': OR (bytes) or sum (int32); rows with a result > 0 are counted
template <class T>
__global__ void k_virt_reduce(const int64_t* __restrict__ first, const T* __restrict__ piece_res, int64_t rows, T* __restrict__ out, unsigned long long* __restrict__ hits) {
  // (a bounded grid, the threads taking rows a grid apart: with a workgroup per 256 rows the hit counter took 244K additions on
  // one address when most rows hold a match -- 2.95 ms for 62.5M rows where the kernel's traffic is worth 0.3)
  int hit = 0;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (int64_t)gridDim.x * kBlock) {
    const int64_t a = first[r], b = first[r + 1];
    T v = piece_res[a];  // (a null row has one piece: its value passes through)
    for (int64_t k = a + 1; k < b; ++k) v = sizeof(T) == 1 ? (T)(v | piece_res[k]) : (T)(v + piece_res[k]);
    out[r] = v;
    hit += v > 0;
  }
  const long long t = csdev::block_reduce_sum(hit);
  if (threadIdx.x == 0 && t) atomicAdd(hits, (unsigned long long)t);
}
// a row's output offset = its first piece's
__global__ void k_virt_offsets(const int64_t* __restrict__ first, const int64_t* __restrict__ piece_off, int64_t rows, int64_t* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r <= rows) out[r] = piece_off[first[r]];
}

// ---- the rows with a byte >= 0x80 or a NUL (OddRows) ----
// One streaming pass over the chars, 16 bytes a lane: a piece that holds such a byte (rare -- the list is only built for columns
// whose sample says so) finds its row by bisection of the offsets and sets the row's bit.  (A thread a row reading its own
// bytes took 4.7 ms for the C5 column's 125M pieces; this pass runs at the read rate.)  The zero test's borrow can flag a 0x01
// behind a NUL byte: a row listed without need, which only sends it the way of the list.
__global__ void __launch_bounds__(256) k_odd_masks(const uint8_t* __restrict__ chars, int64_t nbytes, const int64_t* __restrict__ offsets, int64_t rows,
                                                   unsigned long long* __restrict__ mask) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int64_t n16 = (nbytes + 15) >> 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(chars) + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) any |= (w[k] | ((w[k] - 0x01010101u) & ~w[k])) & 0x80808080u;
    if (!any) continue;
    int64_t done = -1;  // bytes below this one belong to rows already marked
    for (int b = 0; b < 16; ++b) {
      const int64_t p = i * 16 + b;
      if (p >= nbytes) break;
      const uint32_t x = w[b >> 2];
      const uint32_t f = ((x | ((x - 0x01010101u) & ~x)) >> (8 * (b & 3) + 7)) & 1u;
      if (!f || p < done) continue;
      // the row r with offsets[r] <= p < offsets[r + 1]
      int64_t lo = 0, hi = rows;  // offsets[lo] <= p < offsets[hi]
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= p) lo = mid;
        else hi = mid;
      }
      atomicOr(mask + (lo >> 6), 1ull << (lo & 63));
      done = offsets[lo + 1];
    }
  }
}
__global__ void k_odd_counts(const unsigned long long* __restrict__ mask, int64_t tiles, int32_t* __restrict__ cnt) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < tiles) cnt[t] = __builtin_popcountll(mask[t]);
}
__global__ void __launch_bounds__(256) k_odd_list(const unsigned long long* __restrict__ mask, const int64_t* __restrict__ first, int64_t tiles, int32_t* __restrict__ list) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // a lane a row again
  const int64_t t = i >> 6;
  if (t >= tiles) return;
  const unsigned long long m = mask[t];
  const int j = (int)(i & 63);
  if ((m >> j) & 1ull) list[first[t] + __builtin_popcountll(m & ((1ull << j) - 1ull))] = (int32_t)i;
}

std::mutex g_virt_mu;
}  // namespace

namespace cs {

int virtual_piece_bytes() { return kPiece; }

const OddRows* odd_rows(const cs_column* col, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_virt_mu);
  if (col->odd) return col->odd.get();
  auto od = std::make_shared<OddRows>();
  const int64_t tiles = (col->rows + 63) / 64;
  // (the chars are read as aligned 16-byte pieces: a buffer's first and last piece lie inside its allocation -- dev_alloc
  // rounds up -- unless a caller's buffer was wrapped at an odd address, which has no list: count -1)
  if (tiles == 0 || col->nbytes == 0) {
    col->odd = od;
    return col->odd.get();
  }
  if (((uintptr_t)col->d_chars() & 15) || !col->chars || col->chars->capacity < (((size_t)col->nbytes + 15) & ~(size_t)15)) {
    od->count = -1;
    col->odd = od;
    return col->odd.get();
  }
  od->mask = dev_alloc(sizeof(unsigned long long) * (size_t)tiles, s);
  od->first = dev_alloc(sizeof(int64_t) * (size_t)(tiles + 1), s);
  Buf cnt = dev_alloc(sizeof(int32_t) * (size_t)tiles, s);
  CS_HIP(hipMemsetAsync(od->mask->p, 0, sizeof(unsigned long long) * (size_t)tiles, s));
  {
    ProfScope ps("k_odd_masks", s);
    const int64_t n16 = (col->nbytes + 15) >> 4;
    hipLaunchKernelGGL(k_odd_masks, dim3((unsigned)std::min<int64_t>((n16 + 255) / 256, 256 * 32)), dim3(256), 0, s, col->d_chars(), col->nbytes, col->d_offsets(), col->rows,
                       ptr<unsigned long long>(od->mask));
  }
  hipLaunchKernelGGL(k_odd_counts, dim3(blocks_for(tiles)), dim3(kBlock), 0, s, ptr<const unsigned long long>(od->mask), tiles, ptr<int32_t>(cnt));
  CS_HIP(hipGetLastError());
  od->count = offsets_from_lengths(ptr<int32_t>(cnt), tiles, ptr<int64_t>(od->first), s);
  od->list = dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(od->count, 1), s);
  if (od->count > 0) {
    hipLaunchKernelGGL(k_odd_list, dim3(blocks_for(tiles * 64)), dim3(kBlock), 0, s, ptr<const unsigned long long>(od->mask), ptr<const int64_t>(od->first), tiles, ptr<int32_t>(od->list));
    CS_HIP(hipGetLastError());
  }
  CS_HIP(hipStreamSynchronize(s));
  col->odd = od;
  return col->odd.get();
}

// the column's view, or nullptr when it has none (built on first use, kept on the column)
const VirtualRows* virtual_rows(const cs_column* col, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_virt_mu);
  if (col->virt_state > 0) return col->virt.get();
  if (col->virt_state < 0) return nullptr;
  const int64_t rows = col->rows;
  col->virt_state = -1;
  if (rows == 0 || rows >= ((int64_t)1 << 31)) return nullptr;
  ColView in = view_of(col);
  Buf pieces = dev_alloc(sizeof(int32_t) * (size_t)rows, s), flag = dev_alloc(sizeof(int), s), cuts = dev_alloc(sizeof(uint4) * (size_t)rows, s);
  CS_HIP(hipMemsetAsync(flag->p, 0, sizeof(int), s));
  {
    ProfScope ps("k_virt_count", s);
    hipLaunchKernelGGL(k_virt_count, dim3(blocks_for(rows)), dim3(kBlock), 0, s, in, (long long)col->nbytes, ptr<int32_t>(pieces), ptr<uint4>(cuts), ptr<int>(flag));
  }
  CS_HIP(hipGetLastError());
  auto vr = std::make_shared<VirtualRows>();
  vr->first = dev_alloc(sizeof(int64_t) * (size_t)(rows + 1), s);
  const int64_t vrows = offsets_from_lengths(ptr<int32_t>(pieces), rows, ptr<int64_t>(vr->first), s);  // (synchronises)
  int* h = (int*)pinned_scratch(sizeof(int));
  CS_HIP(hipMemcpyAsync(h, flag->p, sizeof(int), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (*h || vrows >= ((int64_t)1 << 31)) return nullptr;
  auto vc = std::make_unique<cs_column>();
  vc->rows = vrows;
  vc->nbytes = col->nbytes;
  vc->chars = col->chars;  // the SAME bytes
  vc->offsets = dev_alloc(sizeof(int64_t) * (size_t)(vrows + 1), s);
  Buf vnull;
  if (col->validity) vnull = dev_alloc(sizeof(int32_t) * (size_t)vrows, s);
  {
    ProfScope ps("k_virt_write", s);
    hipLaunchKernelGGL(k_virt_write, dim3(blocks_for(rows)), dim3(kBlock), 0, s, in, (long long)col->nbytes, ptr<const int64_t>(vr->first), ptr<const uint4>(cuts), ptr<int64_t>(vc->offsets),
                       ptr<int32_t>(vnull));
  }
  CS_HIP(hipGetLastError());
  if (col->validity) vc->validity = validity_from_lengths(ptr<int32_t>(vnull), vrows, s);
  vc->null_count = col->null_count;  // (a null row is one null piece)
  // metadata, as upper bounds (they size staging buffers and pick routes): no piece beyond kPiece bytes
  vc->max_row = col->max_row >= 0 ? std::min<int64_t>(col->max_row, kPiece) : kPiece;
  // (the largest 64-piece span is measured on first use: the bound 64 x 92 would keep the chain form's 5 KB tiles away)
  vc->plain_bytes = col->plain_bytes;
  vc->high_sample = col->high_sample;
  vc->byte_hist = col->byte_hist;
  vc->virt_state = -1;  // (a view has no view)
  CS_HIP(hipStreamSynchronize(s));
  vr->col = std::move(vc);
  col->virt = vr;
  col->virt_state = 1;
  return vr.get();
}

// piece results -> row results (device arrays); returns the rows with a result > 0
int64_t virtual_reduce_u8(const VirtualRows* vr, const uint8_t* piece_res, int64_t rows, uint8_t* out, hipStream_t s) {
  Buf hits = dev_alloc(sizeof(unsigned long long), s);
  CS_HIP(hipMemsetAsync(hits->p, 0, sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_virt_reduce<uint8_t>, dim3(std::min(blocks_for(rows), 4096u)), dim3(kBlock), 0, s, ptr<const int64_t>(vr->first), piece_res, rows, out, ptr<unsigned long long>(hits));
  CS_HIP(hipGetLastError());
  unsigned long long* h = (unsigned long long*)pinned_scratch(sizeof(unsigned long long));
  CS_HIP(hipMemcpyAsync(h, hits->p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return (int64_t)*h;
}
int64_t virtual_reduce_i32(const VirtualRows* vr, const int32_t* piece_res, int64_t rows, int32_t* out, hipStream_t s) {
  Buf hits = dev_alloc(sizeof(unsigned long long), s);
  CS_HIP(hipMemsetAsync(hits->p, 0, sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_virt_reduce<int32_t>, dim3(std::min(blocks_for(rows), 4096u)), dim3(kBlock), 0, s, ptr<const int64_t>(vr->first), piece_res, rows, out, ptr<unsigned long long>(hits));
  CS_HIP(hipGetLastError());
  unsigned long long* h = (unsigned long long*)pinned_scratch(sizeof(unsigned long long));
  CS_HIP(hipMemcpyAsync(h, hits->p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return (int64_t)*h;
}
// the column over `pieces_out`'s chars whose rows are `col`'s: a replace_re of the pieces turned into one of the rows
cs_column* virtual_rows_to_rows(const cs_column* col, const VirtualRows* vr, std::unique_ptr<cs_column> pieces_out, hipStream_t s) {
  auto o = std::make_unique<cs_column>();
  o->rows = col->rows;
  o->nbytes = pieces_out->nbytes;
  o->chars = pieces_out->chars;
  o->validity = col->validity;
  o->null_count = col->null_count;
  o->offsets = dev_alloc(sizeof(int64_t) * (size_t)(col->rows + 1), s);
  hipLaunchKernelGGL(k_virt_offsets, dim3(blocks_for(col->rows + 1)), dim3(kBlock), 0, s, ptr<const int64_t>(vr->first), pieces_out->d_offsets(), col->rows, ptr<int64_t>(o->offsets));
  CS_HIP(hipGetLastError());
  if (col->plain_bytes == 1 && pieces_out->plain_bytes == 1) o->plain_bytes = 1;
  CS_HIP(hipStreamSynchronize(s));  // (`pieces_out`'s offsets leave scope)
  return o.release();
}

}  // namespace cs
