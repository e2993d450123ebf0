// Stable LSD radix sort of (64-bit key, 32-bit item) pairs -- what NVCategory's key build sorts its distinct keys
// with (the key is the first 8 key bytes, big-endian; category/NVCategory.cu:246-304 sorts custring_view pointers with
// thrust) and what NVStrings::order sorts rows with (array.cu:262-330).  Hand-written for gfx950, eight bits a pass:
//   k_radix_hist8    one read of the keys: the histogram of every one of the eight digits (a pass whose digit takes a
//                    single value everywhere -- short keys, small row counts -- is skipped);
//   per pass         k_radix_count (digit histogram per tile of 4096 pairs, laid out digit-major), an exclusive scan
//                    over digits x tiles, k_radix_scatter: every wave ranks its 1024 pairs in memory order -- 16 steps
//                    of 64, a step's equal-digit lanes found with eight ballots (one per digit bit), the running count
//                    per digit in LDS -- and the pairs go to base[digit][tile] + rank.
// 32 bytes of traffic per pair and pass: 100M pairs of 8-byte keys sort in 8 passes = 25.6 GB.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "cs_internal.h"
#include "device_utils.h"

using namespace cs;

namespace {

constexpr int kRadixThreads = 256;
constexpr int kRadixSteps = 16;                              // pairs per thread
constexpr int kRadixTile = kRadixThreads * kRadixSteps;      // 4096 pairs per workgroup
constexpr int kWaveChunk = 64 * kRadixSteps;                 // 1024 consecutive pairs per wave

__global__ void __launch_bounds__(256) k_radix_hist8(const uint64_t* __restrict__ keys, int64_t n, unsigned long long* __restrict__ hist) {
  __shared__ uint32_t h[8 * 256];
  for (int i = threadIdx.x; i < 8 * 256; i += 256) h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint64_t k = keys[i];
#pragma unroll
    for (int d = 0; d < 8; ++d) atomicAdd(&h[d * 256 + (int)((k >> (8 * d)) & 255u)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 256; i += 256)
    if (h[i]) atomicAdd(hist + i, (unsigned long long)h[i]);
}

__global__ void __launch_bounds__(kRadixThreads) k_radix_count(const uint64_t* __restrict__ keys, int64_t n, int shift, int64_t ntiles,
                                                               int32_t* __restrict__ counts) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRadixTile;
#pragma unroll
  for (int j = 0; j < kRadixSteps; ++j) {
    const int64_t i = base + j * kRadixThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255u)], 1u);
  }
  __syncthreads();
  counts[(int64_t)threadIdx.x * ntiles + blockIdx.x] = (int32_t)h[threadIdx.x];
}

// lanes of the wave whose `digit` equals this lane's (active lanes only): one ballot per digit bit
__device__ __forceinline__ unsigned long long same_digit(unsigned digit, bool active) {
  unsigned long long m = __ballot(active);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned long long has = __ballot(active && ((digit >> b) & 1u));
    m &= ((digit >> b) & 1u) ? has : ~has;
  }
  return m;
}

__global__ void __launch_bounds__(kRadixThreads) k_radix_scatter(const uint64_t* __restrict__ keys, const int32_t* __restrict__ items, int64_t n,
                                                                 int shift, int64_t ntiles, const int64_t* __restrict__ bases,
                                                                 uint64_t* __restrict__ keys_out, int32_t* __restrict__ items_out) {
  __shared__ uint32_t cnt[4][256];   // pairs of each digit seen so far by each wave (after the steps: the wave's totals)
  __shared__ long long gbase[256];   // where this tile's pairs of each digit begin in the output
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4 * 256; i += kRadixThreads) (&cnt[0][0])[i] = 0;
  gbase[threadIdx.x] = bases[(int64_t)threadIdx.x * ntiles + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRadixTile + (int64_t)wv * kWaveChunk;
  uint64_t key[kRadixSteps];
  int32_t item[kRadixSteps];
  uint32_t rank[kRadixSteps];  // among the wave's pairs of the same digit, in memory order
#pragma unroll
  for (int j = 0; j < kRadixSteps; ++j) {
    const int64_t i = base + j * 64 + lane;
    const bool active = i < n;
    key[j] = active ? keys[i] : 0;
    item[j] = active ? items[i] : 0;
  }
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < kRadixSteps; ++j) {
    const bool active = base + j * 64 + lane < n;
    const unsigned digit = (unsigned)((key[j] >> shift) & 255u);
    const unsigned long long grp = same_digit(digit, active);
    const uint32_t before = active ? cnt[wv][digit] : 0;  // (every lane of the group reads the same count ...)
    rank[j] = before + (uint32_t)__builtin_popcountll(grp & below);
    // ... and its lowest lane adds the group: the LDS takes a wave's operations in order
    if (active && (grp & below) == 0) cnt[wv][digit] = before + (uint32_t)__builtin_popcountll(grp);
  }
  __syncthreads();
  // the waves' totals become each wave's start inside the tile's block of the digit
  {
    const int d = threadIdx.x;
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = cnt[w][d];
      cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kRadixSteps; ++j) {
    const int64_t i = base + j * 64 + lane;
    if (i < n) {
      const unsigned digit = (unsigned)((key[j] >> shift) & 255u);
      const long long at = gbase[digit] + cnt[wv][digit] + rank[j];
      keys_out[at] = key[j];
      items_out[at] = item[j];
    }
  }
}

}  // namespace

namespace cs {

// Sorts the n pairs (keys[i], items[i]) by key, ascending and stable; the result is left in keys / items.
void radix_sort_pairs64(uint64_t* keys, int32_t* items, int64_t n, hipStream_t s) {
  if (n <= 1) return;
  Buf hist = dev_alloc(sizeof(unsigned long long) * 8 * 256, s);
  CS_HIP(hipMemsetAsync(hist->p, 0, sizeof(unsigned long long) * 8 * 256, s));
  const unsigned hgrid = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_radix_hist8, dim3(hgrid), dim3(256), 0, s, keys, n, ptr<unsigned long long>(hist));
  std::vector<unsigned long long> h(8 * 256);
  CS_HIP(hipMemcpyAsync(h.data(), hist->p, sizeof(unsigned long long) * 8 * 256, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  const int64_t ntiles = (n + kRadixTile - 1) / kRadixTile;
  Buf keys2 = dev_alloc(sizeof(uint64_t) * n, s), items2 = dev_alloc(sizeof(int32_t) * n, s);
  Buf counts = dev_alloc(sizeof(int32_t) * 256 * ntiles, s), bases = dev_alloc(sizeof(int64_t) * (256 * ntiles + 1), s);
  uint64_t* kin = keys;
  int32_t* iin = items;
  uint64_t* kout = ptr<uint64_t>(keys2);
  int32_t* iout = ptr<int32_t>(items2);
  for (int pass = 0; pass < 8; ++pass) {
    bool trivial = false;
    for (int b = 0; b < 256; ++b)
      if (h[pass * 256 + b] == (unsigned long long)n) trivial = true;
    if (trivial) continue;  // every key holds the same digit here: the pass would be the identity
    const int shift = 8 * pass;
    hipLaunchKernelGGL(k_radix_count, dim3((unsigned)ntiles), dim3(kRadixThreads), 0, s, kin, n, shift, ntiles, ptr<int32_t>(counts));
    offsets_from_lengths_async(ptr<int32_t>(counts), 256 * ntiles, ptr<int64_t>(bases), s);  // (no round trip to the host inside the passes)
    hipLaunchKernelGGL(k_radix_scatter, dim3((unsigned)ntiles), dim3(kRadixThreads), 0, s, kin, iin, n, shift, ntiles, ptr<const int64_t>(bases), kout,
                       iout);
    std::swap(kin, kout);
    std::swap(iin, iout);
  }
  CS_HIP(hipGetLastError());
  if (kin != keys) {  // an odd number of passes ran: the result sits in the scratch buffers
    CS_HIP(hipMemcpyAsync(keys, kin, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, s));
    CS_HIP(hipMemcpyAsync(items, iin, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, s));
  }
  CS_HIP(hipStreamSynchronize(s));  // (the scratch buffers go back to the cache)
}

}  // namespace cs
