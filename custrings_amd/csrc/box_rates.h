// Streaming-rate kernels: what THIS box's memory delivers to hand-written streaming kernels of the shapes the headline
// kernels have.  Used twice: `cs_box_rates` (cs_boxrate.hip; bench.py prints the numbers as the `box` block of its line, so
// that a slow box and a regression can be told apart) and tools/ubench/stream_rate.hip (every variant, for DESIGN.md).
//   copy     16 bytes a lane, one read stream -> one write stream
//   read     16 bytes a lane, summed
//   fill     16 bytes a lane, written
//   scatter  emit's shape: a wave owns a run of consecutive "sub-tiles"; per sub-tile it reads `in_bytes` contiguous bytes and
//            writes, for each of `ncols` columns, 256 bytes to an offsets stream (a dword a lane), `piece` bytes (16 a lane) to a
//            chars stream and 8 bytes to a validity stream -- 3 x ncols write streams, each contiguous per wave run
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csbox {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st16(u32x4* p, u32x4 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// grid-stride in units of (workgroup x UNROLL x 16 bytes): every workgroup streams whole 4 KB x UNROLL pieces
template <int UNROLL, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_copy16(u32x4* __restrict__ dst, const u32x4* __restrict__ src, long long n16) {
  const long long stride = (long long)gridDim.x * 256 * UNROLL;
  for (long long i = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n16; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      if (i + j * 256 < n16) v[j] = ld16<NTL>(src + i + j * 256);
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      if (i + j * 256 < n16) st16<NTS>(dst + i + j * 256, v[j]);
  }
}
template <int UNROLL, bool NTL>
__global__ void __launch_bounds__(256) k_read16(const u32x4* __restrict__ src, long long n16, uint32_t* __restrict__ sink) {
  const long long stride = (long long)gridDim.x * 256 * UNROLL;
  uint32_t acc = 0;
  for (long long i = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n16; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      v[j] = u32x4{0, 0, 0, 0};
      if (i + j * 256 < n16) v[j] = ld16<NTL>(src + i + j * 256);
    }
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;  // (never: keeps the loads)
}
template <int UNROLL, bool NTS>
__global__ void __launch_bounds__(256) k_fill16(u32x4* __restrict__ dst, long long n16, uint32_t val) {
  const long long stride = (long long)gridDim.x * 256 * UNROLL;
  const u32x4 v = {val, val + 1, val + 2, val + 3};
  for (long long i = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n16; i += stride) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      if (i + j * 256 < n16) st16<NTS>(dst + i + j * 256, v);
  }
}

struct ScatterArgs {
  const uint8_t* in;     // nsub * in_bytes
  uint8_t* offs;         // ncols streams of nsub * 256 bytes, `off_stride` apart
  uint8_t* chars;        // ncols streams of nsub * piece bytes, `chars_stride` apart
  uint8_t* valid;        // ncols streams of nsub * 8 bytes, `valid_stride` apart
  long long off_stride, chars_stride, valid_stride;
  long long nsub, per;   // sub-tiles, sub-tiles per wave run
  int in_bytes, piece, ncols;
};
// one wave per run; WAVES waves a workgroup.  TWO_PHASE: the wave first reads its whole run (a measure pass of its own:
// the bytes summed), then reads it AGAIN for the scatter -- the second read finds the run in the L2 / the Infinity Cache
// when what the resident waves hold between their two reads fits there (resident waves x per x in_bytes).
template <int WAVES, bool NTS, bool TWO_PHASE>
__global__ void __launch_bounds__(WAVES * 64) k_scatter(ScatterArgs a) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const long long run = (long long)blockIdx.x * WAVES + wv;
  long long t = run * a.per;
  const long long t1 = t + a.per < a.nsub ? t + a.per : a.nsub;
  if (t >= t1) return;
  const int nld = (a.in_bytes + 1023) >> 10;  // 16-byte loads a lane (<= 6)
  u32x4 pf[6];
  auto issue = [&](long long sub) {
    const u32x4* s = reinterpret_cast<const u32x4*>(a.in + sub * a.in_bytes);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < nld && (j * 64 + lane) * 16 < a.in_bytes) pf[j] = s[j * 64 + lane];
  };
  uint32_t acc = 0;
  if (TWO_PHASE) {
    issue(t);
    for (long long u = t; u < t1; ++u) {
      u32x4 cur[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) cur[j] = pf[j];
      if (u + 1 < t1) issue(u + 1);
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j < nld) acc += cur[j].y ^ cur[j].z;
    }
  }
  issue(t);
  for (; t < t1; ++t) {
    u32x4 cur[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) cur[j] = pf[j];
    if (t + 1 < t1) issue(t + 1);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < nld) acc += cur[j].x + cur[j].w;
    const u32x4 v = {acc, acc ^ 1u, acc ^ 2u, acc ^ 3u};
    for (int k = 0; k < a.ncols; ++k) {
      uint32_t* o = reinterpret_cast<uint32_t*>(a.offs + k * a.off_stride + t * 256);
      if (NTS) __builtin_nontemporal_store(acc + (uint32_t)k, o + lane);
      else o[lane] = acc + (uint32_t)k;
      if (lane * 16 < a.piece) st16<NTS>(reinterpret_cast<u32x4*>(a.chars + k * a.chars_stride + t * a.piece) + lane, v);
    }
    if (lane < a.ncols) *reinterpret_cast<unsigned long long*>(a.valid + lane * a.valid_stride + t * 8) = acc;
  }
}

// The shader clock under load: every CU runs dependent integer adds and LDS round trips for a fixed number of iterations;
// wave 0 of workgroup 0 reads the shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around
// its loop.  out[0] = shader cycles, out[1] = 100 MHz ticks.  (The row-lane kernels -- emit above all -- follow the clock the
// box sustains, the streaming rates above do not: a box can be "slow" in one and not in the other.)
__global__ void __launch_bounds__(256) k_clock_probe(int iters, unsigned long long* __restrict__ out, uint32_t* __restrict__ sink) {
  __shared__ uint32_t buf[256];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  uint32_t a = threadIdx.x, b = blockIdx.x;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      a = a * 3u + b;
      b = b ^ (a >> 3);
    }
    a += buf[(a + threadIdx.x) & 255];
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
  }
  if (a == 0x12345u && b == 0x54321u) sink[0] = a;  // (never: keeps the loop)
}

}  // namespace csbox
