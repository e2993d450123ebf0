// cs_box_rates: what this box's memory delivers to hand-written streaming kernels (box_rates.h), measured in the
// process and on the stream the benchmark runs on.  bench.py prints the numbers as the `box` block of its line: the
// headline's kernels follow the box's mixed read / write rate (emit 5.36-6.06 ms across boxes of one pool), so a line
// without them cannot tell a regression from a slow box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>

#include "box_rates.h"
#include "cs_internal.h"

using namespace cs;
using namespace csbox;

namespace {
double median_ms(const std::function<void()>& launch, hipStream_t s, int reps) {
  hipEvent_t a, b;
  CS_HIP(hipEventCreate(&a));
  CS_HIP(hipEventCreate(&b));
  launch();
  CS_HIP(hipStreamSynchronize(s));
  float ms[9];
  reps = std::min(std::max(reps, 1), 9);
  for (int r = 0; r < reps; ++r) {
    CS_HIP(hipEventRecord(a, s));
    launch();
    CS_HIP(hipEventRecord(b, s));
    CS_HIP(hipEventSynchronize(b));
    CS_HIP(hipEventElapsedTime(&ms[r], a, b));
  }
  CS_HIP(hipGetLastError());
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  std::sort(ms, ms + reps);
  return ms[reps / 2];
}
}  // namespace

// tbps[5] = the shader clock in GHz that wave 0 saw while every CU ran k_clock_probe (about a millisecond of integer and LDS work).
// tbps[0..4] = TB/s (bytes read + bytes written over the median launch time) of: a 16-byte-a-lane copy, a read-only
// stream, a write-only stream, emit's shape (one read stream -> 20 x (256 + 192 + 8)-byte pieces a sub-tile, runs of 24
// sub-tiles a wave) with plain and with non-temporal stores.  `mbytes`: size of the copy buffers (the scatter reads as much).
extern "C" int cs_box_rates(int64_t mbytes, int reps, cs_stream stream, double* tbps) {
  return guard([&] {
    if (!tbps || mbytes < 16) fail(CS_ERR_INVALID_ARG, "box_rates: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const long long nbytes = (long long)mbytes << 20, n16 = nbytes / 16;
    {
      Buf src = dev_alloc((size_t)nbytes, s), dst = dev_alloc((size_t)nbytes, s), sink = dev_alloc(64, s);
      CS_HIP(hipMemsetAsync(src->p, 1, (size_t)nbytes, s));
      CS_HIP(hipMemsetAsync(dst->p, 2, (size_t)nbytes, s));
      const unsigned g1 = (unsigned)((n16 + 255) / 256), g4 = (unsigned)((n16 + 1023) / 1024);
      double ms = median_ms([&] { hipLaunchKernelGGL((k_copy16<1, false, false>), dim3(g1), dim3(256), 0, s, (u32x4*)dst->p, (const u32x4*)src->p, n16); }, s, reps);
      tbps[0] = 2.0 * nbytes / ms / 1e9;
      ms = median_ms([&] { hipLaunchKernelGGL((k_read16<4, false>), dim3(g4), dim3(256), 0, s, (const u32x4*)src->p, n16, (uint32_t*)sink->p); }, s, reps);
      tbps[1] = 1.0 * nbytes / ms / 1e9;
      ms = median_ms([&] { hipLaunchKernelGGL((k_fill16<4, false>), dim3(g4), dim3(256), 0, s, (u32x4*)dst->p, n16, 7u); }, s, reps);
      tbps[2] = 1.0 * nbytes / ms / 1e9;
    }
    ScatterArgs a;
    a.in_bytes = 5120;
    a.piece = 192;
    a.ncols = 20;
    a.per = 24;
    a.nsub = nbytes / a.in_bytes;
    a.off_stride = ((a.nsub * 256 + 4095) / 4096 + 3) * 4096;
    a.chars_stride = ((a.nsub * a.piece + 4095) / 4096 + 5) * 4096;
    a.valid_stride = ((a.nsub * 8 + 4095) / 4096 + 7) * 4096;
    Buf in = dev_alloc((size_t)(a.nsub * a.in_bytes + 4096), s), offs = dev_alloc((size_t)(a.off_stride * a.ncols), s),
        chars = dev_alloc((size_t)(a.chars_stride * a.ncols), s), valid = dev_alloc((size_t)(a.valid_stride * a.ncols), s);
    CS_HIP(hipMemsetAsync(in->p, 5, (size_t)(a.nsub * a.in_bytes), s));
    a.in = (const uint8_t*)in->p;
    a.offs = (uint8_t*)offs->p;
    a.chars = (uint8_t*)chars->p;
    a.valid = (uint8_t*)valid->p;
    const double moved = (double)a.nsub * (a.in_bytes + a.ncols * (256 + a.piece + 8));
    const unsigned g = (unsigned)(((a.nsub + a.per - 1) / a.per + 1) / 2);
    double ms = median_ms([&] { hipLaunchKernelGGL((k_scatter<2, false, false>), dim3(g), dim3(128), 0, s, a); }, s, reps);
    tbps[3] = moved / ms / 1e9;
    ms = median_ms([&] { hipLaunchKernelGGL((k_scatter<2, true, false>), dim3(g), dim3(128), 0, s, a); }, s, reps);
    tbps[4] = moved / ms / 1e9;
    {
      int dev = 0, cus = 0;
      CS_HIP(hipGetDevice(&dev));
      CS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      Buf counters = dev_alloc(64, s);
      unsigned long long* host = (unsigned long long*)pinned_scratch(16);
      double ghz = 0;
      for (int r = 0; r < 3; ++r) {  // (the last of three launches: the clock has settled under the load)
        hipLaunchKernelGGL(k_clock_probe, dim3((unsigned)(cus * 8)), dim3(256), 0, s, 20000, (unsigned long long*)counters->p, (uint32_t*)counters->p + 8);
        CS_HIP(hipMemcpyAsync(host, counters->p, 16, hipMemcpyDeviceToHost, s));
        CS_HIP(hipStreamSynchronize(s));
        ghz = host[1] ? (double)host[0] / (double)host[1] * 0.1 : 0.0;
      }
      tbps[5] = ghz;
    }
  });
}
