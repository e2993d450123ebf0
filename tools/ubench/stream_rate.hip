// Microbenchmark: what the box's memory delivers to hand-written streaming kernels ("Next round" 1a).
//   hipcc --offload-arch=gfx950 -O3 -I custrings_amd/csrc -o tools/ubench/stream_rate tools/ubench/stream_rate.hip
//   tools/ubench/stream_rate > profiles/r06/stream_rate.txt
// Questions: (1) does an own 16-byte-a-lane copy reach the guide's 6.29 TB/s where a torch copy gets 4.6-4.9 on these boxes;
// (2) what do read-only and write-only streams get; (3) what does emit's SHAPE get -- one read stream into 3 x 20 write streams
// in 8 / 192 / 256-byte pieces per wave and sub-tile -- with plain and non-temporal stores, short and long wave runs;
// (4) does a second read of a run a wave has just read (the measure pass folded into the emit kernel, TWO_PHASE) cost HBM
// time or is it absorbed by the L2 / the Infinity Cache, as a function of the run length.
// Kernels: custrings_amd/csrc/box_rates.h (the same ones `cs_box_rates` runs inside bench.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <vector>

#include "box_rates.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace csbox;

static double time_ms(const std::function<void()>& launch, int reps) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  launch();  // warm
  CHK(hipDeviceSynchronize());
  std::vector<float> ms(reps);
  for (int r = 0; r < reps; ++r) {
    CHK(hipEventRecord(a, 0));
    launch();
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms[r], a, b));
  }
  CHK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  CHK(hipEventDestroy(a));
  CHK(hipEventDestroy(b));
  return ms[reps / 2];
}

int main(int argc, char** argv) {
  int dev = 0, cus = 0;
  CHK(hipGetDevice(&dev));
  CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  const long long GiB = 1ll << 30;
  const long long nbytes = 4 * GiB, n16 = nbytes / 16;
  uint8_t *src, *dst;
  uint32_t* sink;
  CHK(hipMalloc(&src, nbytes));
  CHK(hipMalloc(&dst, nbytes));
  CHK(hipMalloc(&sink, 64));
  CHK(hipMemset(src, 1, nbytes));
  CHK(hipMemset(dst, 2, nbytes));
  printf("# stream_rate: %d CUs, 4 GiB buffers, median of %d launches, HIP events; TB/s = bytes read + bytes written / time\n", cus, reps);
  {
    double ms = time_ms([&] { CHK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, 0)); }, reps);
    printf("hipMemcpyAsync D2D                         %8.3f ms  %6.2f TB/s\n", ms, 2.0 * nbytes / ms / 1e9);
    ms = time_ms([&] { CHK(hipMemsetAsync(dst, 3, nbytes, 0)); }, reps);
    printf("hipMemsetAsync                             %8.3f ms  %6.2f TB/s\n", ms, 1.0 * nbytes / ms / 1e9);
  }
  // ---- copy / read / fill: grid = CUs x wg_per_cu (persistent grid-stride) or one workgroup per piece (wg_per_cu = 0)
  for (int wg_per_cu : {0, 4, 8, 16, 32}) {
    auto grid = [&](int unroll) { return wg_per_cu ? (unsigned)(cus * wg_per_cu) : (unsigned)((n16 + 256 * unroll - 1) / (256 * unroll)); };
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL((k_copy16<1, false, false>), dim3(grid(1)), dim3(256), 0, 0, (u32x4*)dst, (const u32x4*)src, n16); }, reps);
    printf("copy16 unroll 1 plain      wg/CU %2d       %8.3f ms  %6.2f TB/s\n", wg_per_cu, ms, 2.0 * nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_copy16<4, false, false>), dim3(grid(4)), dim3(256), 0, 0, (u32x4*)dst, (const u32x4*)src, n16); }, reps);
    printf("copy16 unroll 4 plain      wg/CU %2d       %8.3f ms  %6.2f TB/s\n", wg_per_cu, ms, 2.0 * nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_copy16<4, true, true>), dim3(grid(4)), dim3(256), 0, 0, (u32x4*)dst, (const u32x4*)src, n16); }, reps);
    printf("copy16 unroll 4 nt ld+st   wg/CU %2d       %8.3f ms  %6.2f TB/s\n", wg_per_cu, ms, 2.0 * nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_copy16<4, false, true>), dim3(grid(4)), dim3(256), 0, 0, (u32x4*)dst, (const u32x4*)src, n16); }, reps);
    printf("copy16 unroll 4 nt st      wg/CU %2d       %8.3f ms  %6.2f TB/s\n", wg_per_cu, ms, 2.0 * nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_read16<4, false>), dim3(grid(4)), dim3(256), 0, 0, (const u32x4*)src, n16, sink); }, reps);
    printf("read16 unroll 4 plain      wg/CU %2d       %8.3f ms  %6.2f TB/s\n", wg_per_cu, ms, 1.0 * nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_read16<4, true>), dim3(grid(4)), dim3(256), 0, 0, (const u32x4*)src, n16, sink); }, reps);
    printf("read16 unroll 4 nt         wg/CU %2d       %8.3f ms  %6.2f TB/s\n", wg_per_cu, ms, 1.0 * nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_fill16<4, false>), dim3(grid(4)), dim3(256), 0, 0, (u32x4*)dst, n16, 7u); }, reps);
    printf("fill16 unroll 4 plain      wg/CU %2d       %8.3f ms  %6.2f TB/s\n", wg_per_cu, ms, 1.0 * nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_fill16<4, true>), dim3(grid(4)), dim3(256), 0, 0, (u32x4*)dst, n16, 7u); }, reps);
    printf("fill16 unroll 4 nt         wg/CU %2d       %8.3f ms  %6.2f TB/s\n", wg_per_cu, ms, 1.0 * nbytes / ms / 1e9);
  }
  CHK(hipFree(src));
  CHK(hipFree(dst));

  // ---- emit's shape: 100M rows = 1 562 500 sub-tiles of 64 rows; 72 B of chars + 8 B of offsets a row in; per column
  // 256 B of offsets, 192 B of chars, 8 B of validity out
  const long long nsub = 1562500;
  struct Shape { const char* name; int in_bytes, piece, ncols; };
  const Shape shapes[] = {{"emit (20 cols x 256+192+8 B)", 5120, 192, 20}, {"wide pieces (8 cols x 256+896+8 B)", 5120, 896, 8},
                          {"replace-like (1 col x 256+1024+8 B.. x4)", 5120, 1024, 4}};
  for (const Shape& sh : shapes) {
    ScatterArgs a;
    a.nsub = nsub;
    a.in_bytes = sh.in_bytes;
    a.piece = sh.piece;
    a.ncols = sh.ncols;
    a.off_stride = ((nsub * 256 + 4095) / 4096 + 3) * 4096;  // (streams a few pages apart, not a power of two)
    a.chars_stride = ((nsub * sh.piece + 4095) / 4096 + 5) * 4096;
    a.valid_stride = ((nsub * 8 + 4095) / 4096 + 7) * 4096;
    uint8_t *in, *offs, *chars, *valid;
    CHK(hipMalloc(&in, nsub * sh.in_bytes + 4096));
    CHK(hipMalloc(&offs, a.off_stride * sh.ncols));
    CHK(hipMalloc(&chars, a.chars_stride * sh.ncols));
    CHK(hipMalloc(&valid, a.valid_stride * sh.ncols));
    CHK(hipMemset(in, 5, nsub * sh.in_bytes));
    a.in = in;
    a.offs = offs;
    a.chars = chars;
    a.valid = valid;
    const double rd = (double)nsub * sh.in_bytes, wr = (double)nsub * sh.ncols * (256 + sh.piece + 8);
    printf("# shape %s: %.2f GB read, %.2f GB written per launch\n", sh.name, rd / 1e9, wr / 1e9);
    for (int per : {2, 4, 8, 24, 96}) {
      a.per = per;
      const long long runs = (nsub + per - 1) / per;
      double ms;
      ms = time_ms([&] { hipLaunchKernelGGL((k_scatter<2, false, false>), dim3((unsigned)((runs + 1) / 2)), dim3(128), 0, 0, a); }, reps);
      printf("scatter plain st,  one read,   run %3d, 2 waves/wg  %8.3f ms  %6.2f TB/s\n", per, ms, (rd + wr) / ms / 1e9);
      ms = time_ms([&] { hipLaunchKernelGGL((k_scatter<2, true, false>), dim3((unsigned)((runs + 1) / 2)), dim3(128), 0, 0, a); }, reps);
      printf("scatter nt st,     one read,   run %3d, 2 waves/wg  %8.3f ms  %6.2f TB/s\n", per, ms, (rd + wr) / ms / 1e9);
      ms = time_ms([&] { hipLaunchKernelGGL((k_scatter<2, false, true>), dim3((unsigned)((runs + 1) / 2)), dim3(128), 0, 0, a); }, reps);
      printf("scatter plain st,  TWO reads,  run %3d, 2 waves/wg  %8.3f ms  %6.2f TB/s (algorithmic bytes: the second read not counted)\n", per, ms, (rd + wr) / ms / 1e9);
      ms = time_ms([&] { hipLaunchKernelGGL((k_scatter<2, true, true>), dim3((unsigned)((runs + 1) / 2)), dim3(128), 0, 0, a); }, reps);
      printf("scatter nt st,     TWO reads,  run %3d, 2 waves/wg  %8.3f ms  %6.2f TB/s (algorithmic bytes: the second read not counted)\n", per, ms, (rd + wr) / ms / 1e9);
      ms = time_ms([&] { hipLaunchKernelGGL((k_scatter<4, false, false>), dim3((unsigned)((runs + 3) / 4)), dim3(256), 0, 0, a); }, reps);
      printf("scatter plain st,  one read,   run %3d, 4 waves/wg  %8.3f ms  %6.2f TB/s\n", per, ms, (rd + wr) / ms / 1e9);
    }
    {
      // the read pass alone over the same input (what the separate measure kernel costs at its best)
      const long long m16 = nsub * sh.in_bytes / 16;
      double ms = time_ms([&] { hipLaunchKernelGGL((k_read16<4, false>), dim3((unsigned)(cus * 16)), dim3(256), 0, 0, (const u32x4*)in, m16, sink); }, reps);
      printf("read16 over the scatter input (separate pass)          %8.3f ms  %6.2f TB/s\n", ms, rd / ms / 1e9);
    }
    CHK(hipFree(in));
    CHK(hipFree(offs));
    CHK(hipFree(chars));
    CHK(hipFree(valid));
  }
  return 0;
}
