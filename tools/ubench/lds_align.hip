// Microbenchmark: cost of DS accesses at arbitrary alignment on gfx950 (cycles per wave-instruction).
//   hipcc --offload-arch=gfx950 -O3 -o lds_align lds_align.hip && ./lds_align
// Every lane accesses its own 64-byte slot (+ a per-lane byte phase); `mode` picks the instruction and the phases.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(16)));
typedef unsigned long long u64u __attribute__((aligned(1)));
typedef uint32_t u32u __attribute__((aligned(1)));
typedef uint16_t u16u __attribute__((aligned(1)));

template <int OP>
__global__ void __launch_bounds__(256) k(int phase_mode, int iters, unsigned long long* cycles, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint8_t* base = lds + wv * (64 * 64 + 64);
  // phase_mode 0: aligned; 1: every lane phase 1; 2: lane-dependent phase (lane * 5 + 1) & 15; 3: phase 4 (dword aligned, not 8/16)
  int ph = 0;
  if (phase_mode == 1) ph = 1;
  if (phase_mode == 2) ph = (lane * 5 + 1) & 15;
  if (phase_mode == 3) ph = 4;
  if (phase_mode == 4) ph = (lane & 1) ? 3 : 0;  // half the lanes aligned
  int lane_eff = lane;
  if (phase_mode == 5) lane_eff = lane >> 1;                  // pairs of lanes on one address
  if (phase_mode == 6) lane_eff = (lane * 37) & 63;            // a permutation: no conflicts, scattered
  if (phase_mode == 7) lane_eff = (lane & 15) * 4;             // 4-way: lanes l, l+16, l+32, l+48 on one address... (same address -> broadcast for reads)
  if (phase_mode == 8) lane_eff = lane * 32;                   // everybody on one bank, different addresses (stride 128 bytes)
  // slot stride 64 bytes would put all lanes on few banks; use a conflict-free stride of 16 * 1 + ... -> lane * 16 for b128, lane * 4 for b32
  const int stride = OP <= 1 ? 16 : (OP <= 3 ? 8 : 4);
  uint8_t* p = base + (OP >= 9 ? lane_eff * 4 % 1024 + (phase_mode == 8 ? 0 : 0) : lane * stride) + ph;
  for (int i = threadIdx.x; i < (64 * 64 + 64) * 4 / 4 && i < 4200; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc = 0;
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  typedef uint32_t v2 __attribute__((ext_vector_type(2)));
  v4 d4 = {1, 2, 3, 4};
  v2 d2 = {5, 6};
  uint32_t d1 = 7;
  const uint32_t a0 = (uint32_t)(uintptr_t)p;  // LDS byte address (the low 32 bits of a local pointer)
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const uint32_t q = a0 + ((u & 3) << 10);
      if (OP == 0) { v4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(q)); d4 = r; }
      if (OP == 1) { asm volatile("ds_write_b128 %0, %1" :: "v"(q), "v"(d4)); }
      if (OP == 2) { v2 r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(q)); d2 = r; }
      if (OP == 3) { asm volatile("ds_write_b64 %0, %1" :: "v"(q), "v"(d2)); }
      if (OP == 4) { uint32_t r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(q)); d1 = r; }
      if (OP == 5) { asm volatile("ds_write_b32 %0, %1" :: "v"(q), "v"(d1)); }
      if (OP == 6) { asm volatile("ds_write_b16 %0, %1" :: "v"(q), "v"(d1)); }
      if (OP == 7) { asm volatile("ds_write_b8 %0, %1" :: "v"(q), "v"(d1)); }
      if (OP == 8) { uint32_t r; asm volatile("ds_read_u16 %0, %1" : "=v"(r) : "v"(q)); d1 = r; }
      if (OP == 9) { asm volatile("ds_or_b32 %0, %1" :: "v"(q), "v"(d1)); }
      if (OP == 10) { uint32_t r; asm volatile("ds_or_rtn_b32 %0, %1, %2" : "=v"(r) : "v"(q), "v"(d1)); d1 = r; }
      if (OP == 11) { v2 r; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(r) : "v"(q)); d2 = r; }
      if (OP == 12) { asm volatile("ds_write_b8 %0, %1" :: "v"(q), "v"(d1)); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  acc = d4.x ^ d2.x ^ d1;
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) atomicAdd(cycles, t1 - t0);
  if (acc == 0x12345) sink[0] = acc;
}

template <int OP>
double run(int phase_mode, int blocks_per_cu, const char* name) {
  unsigned long long* d;
  uint32_t* sink;
  hipMalloc(&d, 8);
  hipMalloc(&sink, 4);
  hipMemset(d, 0, 8);
  const int iters = 2000;
  const int grid = 256 * blocks_per_cu;
  const size_t lds = (64 * 64 + 64) * 4;
  k<OP><<<grid, 256, lds>>>(phase_mode, 10, d, sink);
  hipDeviceSynchronize();
  hipMemset(d, 0, 8);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  k<OP><<<grid, 256, lds>>>(phase_mode, iters, d, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long c;
  hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
  const double waves = (double)grid * 4;
  const double per_wave_instr = (double)c / waves / (iters * 16.0);
  // CU-level cycles per wave-instruction: wall time * clock / (instructions per CU)
  const double cu_cycles = ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * 4 * iters * 16.0);
  printf("%-10s phase_mode %d  waves/CU %2d : %7.1f wave-clock cycles per instr, %6.2f CU cycles per wave-instr (2.4 GHz assumed)\n", name, phase_mode, blocks_per_cu * 4,
         per_wave_instr, cu_cycles);
  hipFree(d);
  hipFree(sink);
  return cu_cycles;
}

int main() {
  const char* names[] = {"rd_b128", "wr_b128", "rd_b64", "wr_b64", "rd_b32", "wr_b32", "wr_b16", "wr_b8", "rd_b16", "or_b32", "or_rtn_b32", "rd2_b32", "wr_b8"};
  for (int pm : {0, 5, 6, 7, 8}) {
    run<9>(pm, 2, names[9]);
    run<10>(pm, 2, names[10]);
    run<11>(pm, 2, names[11]);
    run<12>(pm, 2, names[12]);
  }
  for (int bpc : {1, 2}) {
    for (int pm = 0; pm <= 4; ++pm) {
      run<0>(pm, bpc, names[0]);
      run<1>(pm, bpc, names[1]);
      run<2>(pm, bpc, names[2]);
      run<3>(pm, bpc, names[3]);
      run<4>(pm, bpc, names[4]);
      run<5>(pm, bpc, names[5]);
      run<6>(pm, bpc, names[6]);
      run<7>(pm, bpc, names[7]);
      run<8>(pm, bpc, names[8]);
    }
  }
  return 0;
}
