// Synthetic benchmark columns generated directly into the native device column
// (BASELINE.md section 3).  Row content is defined by include/cs_synth_spec.h.
#include <hip/hip_runtime.h>

#include "cs_internal.h"
#include "cs_synth_spec.h"
#include "device_utils.h"

using namespace cs;
using namespace csdev;

namespace {
__global__ void k_synth_sizes(int kind, int64_t first_row, int64_t rows, uint64_t seed, int64_t param,
                              int32_t* __restrict__ lens, int64_t* __restrict__ block_sums) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int len = -1;
  if (i < rows && !cs_synth_is_null(kind, seed, first_row + i))
    len = cs_synth_row(kind, seed, first_row + i, param, nullptr);
  if (i < rows) lens[i] = len;
  long long t = block_reduce_sum(len < 0 ? 0 : len);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = t;
}
__global__ void k_synth_write(int kind, int64_t first_row, int64_t rows, uint64_t seed, int64_t param,
                              const int32_t* __restrict__ lens, const int64_t* __restrict__ off,
                              uint8_t* __restrict__ chars) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= rows || lens[i] < 0) return;
  cs_synth_row(kind, seed, first_row + i, param, chars + off[i]);
}
}  // namespace

extern "C" int cs_synth_column(int kind, int64_t first_row, int64_t rows, uint64_t seed, int64_t param,
                               cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!out || rows < 0 || kind < 2 || kind > 5) fail(CS_ERR_INVALID_ARG, "synth: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    auto c = std::make_unique<cs_column>();
    c->rows = rows;
    unsigned nb = blocks_for(rows);
    Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
    Buf sums = dev_alloc(sizeof(int64_t) * nb, s);
    hipLaunchKernelGGL(k_synth_sizes, dim3(nb), dim3(kBlock), 0, s, kind, first_row, rows, seed, param,
                       ptr<int32_t>(lens), ptr<int64_t>(sums));
    c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    LenMeta meta;  // (a generated column is sized at ingest like any other: NVStringsImpl.cu:399-444)
    c->nbytes = offsets_from_lengths(ptr<int32_t>(lens), rows, ptr<int64_t>(c->offsets), s, sums, &meta);
    meta.give(c.get());
    c->chars = dev_alloc((size_t)c->nbytes, s);
    if (kind != 3) c->validity = validity_from_lengths(ptr<int32_t>(lens), rows, s);
    else c->null_count = 0;
    hipLaunchKernelGGL(k_synth_write, dim3(nb), dim3(kBlock), 0, s, kind, first_row, rows, seed, param,
                       ptr<const int32_t>(lens), c->d_offsets(), ptr<uint8_t>(c->chars));
    CS_HIP(hipStreamSynchronize(s));
    *out = c.release();
  });
}
