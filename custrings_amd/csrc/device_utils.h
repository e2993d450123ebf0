// Wavefront / workgroup primitives for gfx950 (wave64): DPP prefix scans and
// reductions, ballot-packed validity words.  Device-only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csdev {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // rows per workgroup in the row-parallel kernels
constexpr int kMaxWaves = 16;  // workgroups of up to 1024 threads

// ---- DPP building blocks --------------------------------------------------
// v_mov_b32 with a DPP control; lanes whose source is out of range (or masked
// off by row_mask/bank_mask) receive `old`.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND_CTRL = false>
__device__ __forceinline__ int dpp_mov(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114,
              DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// Inclusive prefix sum across the 64 lanes of a wave: 4 row_shr steps scan each
// 16-lane row, row_bcast:15 / row_bcast:31 carry the row totals across rows.
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  // (old = 0 with bound_ctrl lets the compiler fold each move into a v_add_u32_dpp: 6 VALU in all)
  v += dpp_mov<DPP_ROW_SHR1, 0xF, 0xF, true>(0, v);
  v += dpp_mov<DPP_ROW_SHR2, 0xF, 0xF, true>(0, v);
  v += dpp_mov<DPP_ROW_SHR4, 0xF, 0xF, true>(0, v);
  v += dpp_mov<DPP_ROW_SHR8, 0xF, 0xF, true>(0, v);
  v += dpp_mov<DPP_ROW_BCAST15, 0xA>(0, v);  // rows 1 and 3 += last lane of rows 0 and 2
  v += dpp_mov<DPP_ROW_BCAST31, 0xC>(0, v);  // rows 2,3 += lane 31
  return v;
}
__device__ __forceinline__ long long wave_inclusive_scan(long long v) {
  // 64-bit: scan low and high halves with carry propagation done by splitting
  // into two 32-bit lanes would lose carries, so use the shuffle form here
  // (only used for per-block totals, off the per-row path).
  for (int d = 1; d < kWave; d <<= 1) {
    long long t = __shfl_up(v, d, kWave);
    if ((int)(threadIdx.x & (kWave - 1)) >= d) v += t;
  }
  return v;
}

__device__ __forceinline__ int wave_reduce_sum(int v) {
  v = wave_inclusive_scan(v);
  return __builtin_amdgcn_readlane(v, kWave - 1);
}
__device__ __forceinline__ int wave_reduce_max(int v) {
  for (int d = kWave / 2; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
  return v;
}

// ---- workgroup (any multiple of 64 threads) ---------------------------------------
// Exclusive prefix sum of one int per thread; *total receives the block sum.
// Per-row lengths are < 2^31 and a block holds 256 rows, so the in-block prefix
// is carried in 64 bits only at the wave-total level.
__device__ __forceinline__ long long block_exclusive_scan(int v, long long* total) {
  __shared__ long long wave_tot[kMaxWaves];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  const int nwaves = (blockDim.x + kWave - 1) / kWave;
  // in-wave scan in 64 bits when lengths can be large; rows are almost always
  // short, so do the 32-bit DPP scan and detect overflow via the 64-bit sum of
  // the wave computed separately
  long long wide = v;
  for (int d = kWave / 2; d > 0; d >>= 1) wide += __shfl_xor(wide, d, kWave);
  long long incl;
  if (wide < 0x7fffffffLL) {
    incl = wave_inclusive_scan(v);
  } else {
    incl = wave_inclusive_scan((long long)v);
  }
  __syncthreads();  // protect wave_tot reuse across consecutive calls
  if (lane == kWave - 1) wave_tot[wv] = incl;
  __syncthreads();
  long long base = 0, all = 0;
  for (int k = 0; k < nwaves; ++k) {
    long long t = wave_tot[k];
    if (k < wv) base += t;
    all += t;
  }
  if (total) *total = all;
  return base + incl - v;
}
__device__ __forceinline__ long long block_reduce_sum(int v) {
  long long t;
  block_exclusive_scan(v, &t);
  return t;
}
// (64-bit addends: what a thread gathered over the rows of a grid-stride loop)
__device__ __forceinline__ long long block_reduce_sum_ll(long long v) {
  __shared__ long long wave_sum[kMaxWaves];
  const int nwaves = (blockDim.x + kWave - 1) / kWave;
  for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) wave_sum[threadIdx.x / kWave] = v;
  __syncthreads();
  long long t = 0;
  for (int k = 0; k < nwaves; ++k) t += wave_sum[k];
  return t;
}
__device__ __forceinline__ int block_reduce_max(int v) {
  __shared__ int wave_max[kMaxWaves];
  const int nwaves = (blockDim.x + kWave - 1) / kWave;
  v = wave_reduce_max(v);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) wave_max[threadIdx.x / kWave] = v;
  __syncthreads();
  int m = wave_max[0];
  for (int k = 1; k < nwaves; ++k) m = max(m, wave_max[k]);
  return m;
}

// Arrow validity: bit r of the mask (LSB first) = row r valid.  A wave covers
// 64 consecutive rows = one aligned 8-byte word, written by its first lane.
__device__ __forceinline__ void store_validity_word(uint8_t* validity, long long first_row_of_wave,
                                                    bool valid, long long rows) {
  unsigned long long m = __ballot(valid);
  if ((threadIdx.x & (kWave - 1)) == 0 && first_row_of_wave < rows)
    *reinterpret_cast<unsigned long long*>(validity + (first_row_of_wave >> 3)) = m;
}
__device__ __forceinline__ bool row_is_valid(const uint8_t* validity, long long r) {
  return validity == nullptr || ((validity[r >> 3] >> (r & 7)) & 1);
}

// copies n bytes between arbitrarily aligned global addresses, 8 bytes at a time
__device__ __forceinline__ void copy_bytes(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, int n) {
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t v;
    __builtin_memcpy(&v, s + i, 8);
    __builtin_memcpy(d + i, &v, 8);
  }
  if (i + 4 <= n) {
    uint32_t v;
    __builtin_memcpy(&v, s + i, 4);
    __builtin_memcpy(d + i, &v, 4);
    i += 4;
  }
  for (; i < n; ++i) d[i] = s[i];
}

}  // namespace csdev
