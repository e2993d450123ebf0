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

#include <cstring>

#include "cs_internal.h"
#include "device_utils.h"
#include "regex_program.h"
#include "regex_vm.h"

using namespace cs;
using namespace csdev;

struct cs_regex {
  csrx::Program prog;
  std::vector<int32_t> blob;   // program only (ABI: cs_regex_blob)
  std::vector<int32_t> image;  // blob + executor extras
  Buf d_image;                 // uploaded lazily
  bool empty_pattern = false;
};

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

Plan plan(cs_regex* re, int64_t rows, hipStream_t s) {
  require_device();
  if (!re->d_image) {
    re->d_image = dev_alloc(re->image.size() * 4, s);
    CS_HIP(hipMemcpyAsync(re->d_image->p, re->image.data(), re->image.size() * 4, hipMemcpyHostToDevice, s));
    CS_HIP(hipStreamSynchronize(s));
  }
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

template <int MODE>
void scan(const cs_column* col, cs_regex* re, uint8_t* out8, int32_t* out32, int on_device, hipStream_t s,
          int64_t* found, const char* name) {
  if (found) *found = 0;
  if (col->rows == 0) return;
  Plan pl = plan(re, col->rows, s);
  const size_t esz = MODE == 2 ? 4 : 1;
  void* host_out = MODE == 2 ? (void*)out32 : (void*)out8;
  Buf tmp;
  if (!on_device) {
    tmp = dev_alloc(esz * col->rows, s);
    if (MODE == 2) out32 = ptr<int32_t>(tmp);
    else out8 = ptr<uint8_t>(tmp);
  }
  Buf cnt = dev_alloc(8, s);
  CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
  RowSrc src{view_of(col), d_unicode_flags()};
  {
    ProfScope ps(name, s);
    if (pl.small)
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
    auto* re = new cs_regex;
    re->empty_pattern = *pattern == 0;
    re->prog = csrx::compile(pattern);
    re->blob = re->prog.to_blob();
    re->image = re->prog.to_device_image(h_unicode_flags());
    *out = re;
  });
}
int cs_regex_destroy(cs_regex* re) {
  return guard([&] { delete re; });
}
int cs_regex_inst_count(const cs_regex* re) { return re ? (int)re->prog.insts.size() : 0; }
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
    Buf d_repl = dev_alloc((size_t)rb + 1, s);
    CS_HIP(hipMemcpyAsync(d_repl->p, repl, (size_t)rb + 1, hipMemcpyHostToDevice, s));
    Plan pl = plan(re, col->rows, s);
    RowSrc src{view_of(col), d_unicode_flags()};
    auto* o = new cs_column;
    std::unique_ptr<cs_column> holder(o);
    o->rows = col->rows;
    o->validity = col->validity;
    o->null_count = col->null_count;
    Buf lens = dev_alloc(sizeof(int32_t) * col->rows, s);
    {
      ProfScope ps("k_replace_re_size", s);
      if (pl.small)
        hipLaunchKernelGGL((k_replace_re_size<true>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d, rb,
                           maxrepl, ptr<int32_t>(lens));
      else
        hipLaunchKernelGGL((k_replace_re_size<false>), dim3(pl.grid), dim3(pl.threads), pl.lds_bytes, s, src, pl.d, rb,
                           maxrepl, ptr<int32_t>(lens));
    }
    CS_HIP(hipGetLastError());
    o->offsets = dev_alloc(sizeof(int64_t) * (col->rows + 1), s);
    o->nbytes = offsets_from_lengths(ptr<int32_t>(lens), col->rows, ptr<int64_t>(o->offsets), s);
    o->chars = dev_alloc((size_t)o->nbytes, s);
    {
      ProfScope ps("k_replace_re_write", s);
      if (pl.small)
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

}  // extern "C"
