// Library core: error reporting, device buffer cache, unicode tables, column
// construction / export (NVStrings.cu:74-153,402-544; NVStringsImpl.cu:126-444;
// attrs.cu:72-112), the lengths->offsets scan and measurement hooks.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <unordered_map>

#include "_gen/unicode_tables.inc"  // cs_unicode_flags[65536], cs_charcases[65536]
#include "cs_internal.h"
#include "cs_synth_spec.h"
#include "device_utils.h"
#include "tile_utils.h"

using namespace csdev;

namespace cs {

// ---------------------------------------------------------------- errors ----
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
void fail(int code, const std::string& msg) { throw Error{code, msg}; }

// ------------------------------------------------------------ device state --
static std::mutex g_mu;
static int g_device = -1;
static uint8_t* g_d_flags = nullptr;
static uint16_t* g_d_cases = nullptr;
static int64_t g_in_use = 0;
// A cached block remembers the stream it was last used on and, once a second stream has been seen, an EVENT recorded on
// that stream when the block was released: a different stream that takes the block waits for the event on the device
// (hipStreamWaitEvent) -- the host does not wait, where a device-wide synchronisation per released block used to turn
// every op of a multi-stream caller into dozens of full stalls.
struct CachedBlock {
  void* first;
  hipStream_t second;
  std::vector<hipEvent_t> released;  // one per stream known at the release (a column made on one stream may have been read on another)
};
static std::multimap<size_t, CachedBlock> g_cache;  // capacity -> block
static int64_t g_cached_bytes = 0;                  // sum of the capacities in g_cache
static int64_t g_cache_limit = 0;                   // 0 = not sized yet (half the device's memory, CS_POOL_MAX_MB overrides)
static std::atomic<long long> g_mallocs{0};         // hipMalloc calls made by dev_alloc (cs_debug_malloc_count)
static const hipStream_t kNoStream = reinterpret_cast<hipStream_t>(~(uintptr_t)0);  // "last used by nobody we still know"
static std::vector<hipEvent_t> g_event_pool;
// Streams that have asked for buffers.  While there is one (the usual case) stream order alone
// makes the reuse of a released block safe.  With more, a block may still be read by a kernel on
// a stream other than the one it was allocated on when its last handle goes away, so a release
// then records an event on every known stream, and whoever takes the block next waits for them on the device.
static std::set<hipStream_t> g_streams;
static std::atomic<bool> g_multi_stream{false};
static std::atomic<long long> g_fallbacks{0};

void require_device() {
  if (g_device < 0) fail(CS_ERR_NO_DEVICE, "cs_init has not succeeded: no usable gfx950 device (there is no CPU fallback)");
  // hipSetDevice is per thread: a host thread other than the one that ran cs_init would launch on device 0
  static thread_local int bound = -1;
  if (bound != g_device) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != g_device) CS_HIP(hipSetDevice(g_device));
    bound = g_device;
  }
}
int bound_device() { return g_device; }
thread_local const char* g_last_route = "";
void note_route(const char* route) { g_last_route = route; }
// the op ran on the column's pieces (cs_virtual.hip): the inner launch's route under a "pieces:" prefix
thread_local char g_route_text[64];
void note_route_pieces() {
  char inner[48];
  snprintf(inner, sizeof inner, "%s", g_last_route);
  snprintf(g_route_text, sizeof g_route_text, "pieces:%s", inner);
  g_last_route = g_route_text;
}
// the scan put the rows with bytes >= 0x80 off to a second launch (cs_regex.hip: ScanStreamArgs::deferred)
void note_route_put_off() {
  char inner[48];
  snprintf(inner, sizeof inner, "%s", g_last_route);
  snprintf(g_route_text, sizeof g_route_text, "%s+later", inner);
  g_last_route = g_route_text;
}
void note_fallback(const char* what) {
  if (g_fallbacks.fetch_add(1) == 0 || cs::cfg("CS_LOG_FALLBACKS"))
    fprintf(stderr, "custrings_amd: %s: the single-pass kernel gave up, recomputing with the two-pass kernels\n", what);
}
const uint8_t* d_unicode_flags() { return g_d_flags; }
const uint16_t* d_charcases() { return g_d_cases; }
const uint8_t* h_unicode_flags() { return cs_unicode_flags; }
const uint16_t* h_charcases() { return cs_charcases; }
int64_t dev_bytes_in_use() { return g_in_use; }

static void release_cache_locked() {
  for (auto& kv : g_cache) {
    (void)hipFree(kv.second.first);
    for (hipEvent_t e : kv.second.released) g_event_pool.push_back(e);
  }
  g_cache.clear();
  g_cached_bytes = 0;
}

// Capacity a new block gets for a request of `want` bytes (reference: the RMM pool behind every device_alloc,
// cpp/src/util.inl:90-106, python/tests/utils.py:25-34 -- a column a few KB larger than the last one must not cost a
// hipMalloc of gigabytes, 120-134 ms on this machine).  From 1 MiB on: 1/32 of headroom, then the next of eight
// geometric steps per octave, so the block also serves every later request up to 3 % larger (and, by the reuse rule in
// dev_alloc, down to 20 % smaller); at most 16 % over the request.  Below that the sizes of a pipeline's buffers vary
// more from column to column (a split column that few rows reach: +-10 %) and their bytes do not matter: 1/8 of headroom
// and four steps per octave from 4 KiB on, whole 4 KiB pages below.
static size_t size_class(size_t want) {
  if (want <= 4096) return 4096;
  const bool small = want < ((size_t)1 << 20);
  const size_t w = want + want / (small ? 8 : 32);
  const int e = 63 - __builtin_clzll((unsigned long long)w);
  const size_t step = (size_t)1 << (e - (small ? 2 : 3));
  return (w + step - 1) / step * step;
}
// the largest idle block a request may take: half as much again (one class up from the block a request 3 % larger was
// given: with 25 % a request for 15.06 MB missed an idle 18.87 MB block by 50 KB -- tools/pool_replay.py replays a trace of
// the pool's requests through this policy), or 256 KiB (the small blocks' classes are coarse)
static size_t reuse_limit(size_t want) { return want + std::max<size_t>(want / 2 + 4096, (size_t)256 << 10); }
constexpr int kPageBlocksAtFirstTouch = 64;  // page-sized blocks allocated together the first time one is needed (below)

// The cache is bounded: beyond the limit the largest idle blocks go back to the driver (hipFree waits for the device:
// rare by construction -- the limit is half the device's memory unless CS_POOL_MAX_MB says otherwise).
static void trim_cache_locked() {
  if (g_cache_limit == 0) {
    size_t free_b = 0, total_b = 0;
    if (const char* e = cs::cfg("CS_POOL_MAX_MB")) g_cache_limit = std::max<int64_t>(1, atoll(e)) << 20;
    else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) g_cache_limit = (int64_t)(total_b / 2);
    else g_cache_limit = (int64_t)64 << 30;
  }
  while (g_cached_bytes > g_cache_limit && !g_cache.empty()) {
    auto it = std::prev(g_cache.end());
    (void)hipFree(it->second.first);
    for (hipEvent_t e : it->second.released) g_event_pool.push_back(e);
    g_cached_bytes -= (int64_t)it->first;
    g_cache.erase(it);
  }
}

DevBuf::~DevBuf() {
  if (ipc_mapped && p) {
    (void)hipIpcCloseMemHandle(p);
    return;
  }
  if (!capacity || !p) return;
  std::vector<hipEvent_t> evs;
  if (g_multi_stream.load(std::memory_order_relaxed)) {
    std::vector<hipStream_t> streams;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      streams.assign(g_streams.begin(), g_streams.end());
      while (evs.size() < streams.size() && !g_event_pool.empty()) {
        evs.push_back(g_event_pool.back());
        g_event_pool.pop_back();
      }
    }
    bool ok = true;
    while (ok && evs.size() < streams.size()) {
      hipEvent_t e = nullptr;
      ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
      if (ok) evs.push_back(e);
    }
    std::vector<hipStream_t> dead;
    for (size_t i = 0; ok && i < streams.size(); ++i)
      if (hipEventRecord(evs[i], streams[i]) != hipSuccess) {  // (a stream its owner destroyed: forget it)
        (void)hipGetLastError();
        dead.push_back(streams[i]);
        ok = false;
      }
    if (!ok) {
      (void)hipDeviceSynchronize();  // (no complete set of events to wait for: the block is idle before anyone else may take it)
      std::lock_guard<std::mutex> lk(g_mu);
      for (hipEvent_t e : evs) g_event_pool.push_back(e);
      for (hipStream_t d : dead) g_streams.erase(d);
      evs.clear();
    }
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_in_use -= (int64_t)capacity;
  if (cs::cfg_int("CS_POOL_TRACE", 0) >= 2) fprintf(stderr, "pool- %zu %zu\n", bytes, capacity);
  g_cache.emplace(capacity, CachedBlock{p, stream, std::move(evs)});
  g_cached_bytes += (int64_t)capacity;
  trim_cache_locked();
}

Buf dev_alloc(size_t bytes, hipStream_t stream) {
  require_device();
  size_t want = ((bytes + 64 + 511) / 512) * 512;
  auto b = std::make_shared<DevBuf>();
  b->bytes = bytes;
  b->stream = stream;
  bool reused = false;
  hipStream_t prev = nullptr;
  std::vector<hipEvent_t> released;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_streams.insert(stream).second && g_streams.size() > 1) g_multi_stream.store(true);
    auto it = g_cache.lower_bound(want);
    if (it != g_cache.end() && it->first <= reuse_limit(want)) {
      b->p = it->second.first;
      b->capacity = it->first;
      prev = it->second.second;
      // (a block whose last stream is no longer known -- cs_stream_forget synchronised it, its owner may have destroyed
      // it since -- is idle: nothing to wait for, and the handle must not be touched)
      if (prev != stream && !g_streams.count(prev)) prev = stream;
      released = std::move(it->second.released);
      g_cache.erase(it);
      g_cached_bytes -= (int64_t)b->capacity;
      g_in_use += (int64_t)b->capacity;
      reused = true;
    }
  }
  if (cs::cfg_int("CS_POOL_TRACE", 0) >= 2) fprintf(stderr, "pool+ %zu %zu %s\n", bytes, reused ? b->capacity : (size_t)0, reused ? "hit" : "miss");
  if (reused) {
    // a block last used on another stream may still be in flight there: this stream waits for the release event on the
    // device (blocks released before a second stream existed carry none: the host waits for their stream once)
    if (!released.empty()) {
      struct BackToPool {  // (also when a wait below throws: the events are not lost)
        std::vector<hipEvent_t>& evs;
        ~BackToPool() {
          std::lock_guard<std::mutex> lk(g_mu);
          for (hipEvent_t e : evs) g_event_pool.push_back(e);
        }
      } back{released};
      for (hipEvent_t e : released) CS_HIP(hipStreamWaitEvent(stream, e, 0));
    } else if (prev != stream) {
      CS_HIP(hipStreamSynchronize(prev));
    }
    return b;
  }
  void* p = nullptr;
  const size_t asked = want;
  want = size_class(want);
  g_mallocs.fetch_add(1, std::memory_order_relaxed);
  if (cs::cfg("CS_POOL_TRACE")) {  // (tests / tools: which requests miss the pool)
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.lower_bound(asked);
    fprintf(stderr, "custrings_amd pool: hipMalloc %zu bytes for a request of %zu (%zu idle blocks, %lld bytes; the next larger idle block: %zu)\n", want, asked, g_cache.size(),
            (long long)g_cached_bytes, it == g_cache.end() ? (size_t)0 : it->first);
  }
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lk(g_mu);
      (void)hipDeviceSynchronize();
      release_cache_locked();
    }
    e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      fail(CS_ERR_ALLOC, "allocate error: " + std::to_string(want) + " bytes (" + hipGetErrorString(e) + ")");
    }
  }
  b->p = p;
  b->capacity = want;
  std::lock_guard<std::mutex> lk(g_mu);
  g_in_use += (int64_t)want;
  // Page-sized blocks (scalars, flags, the chars of a column that few rows reach): how many of them are live at once
  // varies from column to column as sizes cross the page -- the first one brings its siblings along (each its own
  // allocation: any of them may later be exported over HIP IPC), so that a later op does not stop for a 4 KB hipMalloc.
  static bool page_blocks_made = false;
  if (want == 4096 && !page_blocks_made) {
    page_blocks_made = true;
    for (int i = 0; i < kPageBlocksAtFirstTouch; ++i) {
      void* q = nullptr;
      if (hipMalloc(&q, 4096) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      g_mallocs.fetch_add(1, std::memory_order_relaxed);
      g_cache.emplace((size_t)4096, CachedBlock{q, kNoStream, {}});
      g_cached_bytes += 4096;
    }
  }
  return b;
}

Buf dev_wrap(const void* p, size_t bytes) {
  auto b = std::make_shared<DevBuf>();
  b->p = const_cast<void*>(p);
  b->bytes = bytes;
  b->capacity = 0;
  return b;
}

void* pinned_scratch(size_t bytes) {
  static thread_local void* p = nullptr;
  static thread_local size_t cap = 0;
  if (bytes > cap) {
    if (p) (void)hipHostFree(p);
    size_t want = bytes < 4096 ? 4096 : bytes;
    CS_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
  }
  return p;
}

// ------------------------------------------------------------- profiling ----
struct ProfEntry {
  double ms = 0;
  int64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
static bool g_prof_on = false;
static std::unordered_map<std::string, ProfEntry> g_prof;

ProfScope::ProfScope(const char* nm, hipStream_t st) : name(nm), s(st) {
  if (!g_prof_on) return;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a, s);
}
ProfScope::~ProfScope() {
  if (!a) return;
  (void)hipEventRecord(b, s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof[name].pending.emplace_back(a, b);
}
static void prof_collect(ProfEntry& e) {
  for (auto& pr : e.pending) {
    (void)hipEventSynchronize(pr.second);
    float ms = 0;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      e.ms += ms;
      e.launches += 1;
    }
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  e.pending.clear();
}

// ------------------------------------------------------------ scan kernels --
// block sums of 256 lengths (negative = null -> 0)
__global__ void k_block_sums(const int32_t* __restrict__ lens, int64_t n, int64_t* __restrict__ sums) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int v = (i < n) ? lens[i] : 0;
  if (v < 0) v = 0;
  long long t = block_reduce_sum(v);
  if (threadIdx.x == 0) sums[blockIdx.x] = t;
}
// One workgroup per segment scans `nb` int64 block sums in place (exclusive),
// 8 per thread per sweep, and publishes the segment total.
__global__ void __launch_bounds__(1024) k_scan_block_sums(int64_t* __restrict__ sums, int64_t nb,
                                                          int64_t* __restrict__ totals) {
  __shared__ long long wave_tot[16];
  __shared__ long long carry_s;
  int64_t* seg = sums + (int64_t)blockIdx.x * nb;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += 8192) {
    long long v[8], run = 0;
    int64_t i0 = base + (int64_t)threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = (i0 + k < nb) ? seg[i0 + k] : 0;
      run += v[k];
    }
    long long incl = wave_inclusive_scan(run);
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    long long before = carry_s, all = 0;
    for (int k = 0; k < 16; ++k) {
      long long t = wave_tot[k];
      if (k < wv) before += t;
      all += t;
    }
    long long ex = before + incl - run;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (i0 + k < nb) seg[i0 + k] = ex;
      ex += v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}
// offsets[i] = block_base[b] + in-block exclusive prefix; offsets[n] = total
// (meta: [0] longest row, [1] largest 64-row span -- a wave holds 64 consecutive rows starting at a multiple of 64; only a
// wave that would raise a maximum issues the same-address atomic)
__device__ __forceinline__ void meta_raise(unsigned long long* slot, long long v) {
  if ((unsigned long long)v > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, (unsigned long long)v);
}
__global__ void k_write_offsets(const int32_t* __restrict__ lens, int64_t n,
                                const int64_t* __restrict__ block_base, int64_t nb,
                                int64_t* __restrict__ offsets, unsigned long long* __restrict__ meta) {
  // grid.y = segment
  const int32_t* sl = lens + (int64_t)blockIdx.y * n;
  const int64_t* sb = block_base + (int64_t)blockIdx.y * nb;
  int64_t* so = offsets + (int64_t)blockIdx.y * (n + 1);
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int v = (i < n) ? sl[i] : 0;
  if (v < 0) v = 0;
  long long ex = block_exclusive_scan(v, nullptr) + sb[blockIdx.x];
  if (i < n) so[i] = ex;
  if (i == n - 1) so[n] = ex + v;
  if (meta) {  // (two words per segment)
    const int longest = wave_reduce_max(v);
    long long span = v;
    for (int d = 32; d > 0; d >>= 1) span += __shfl_xor(span, d, 64);
    if ((threadIdx.x & 63) == 0) {
      meta_raise(meta + 2 * blockIdx.y, longest);
      meta_raise(meta + 2 * blockIdx.y + 1, span);
    }
  }
}
__global__ void k_block_sums_seg(const int32_t* __restrict__ lens, int64_t n, int64_t nb,
                                 int64_t* __restrict__ sums) {
  const int32_t* sl = lens + (int64_t)blockIdx.y * n;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int v = (i < n) ? sl[i] : 0;
  if (v < 0) v = 0;
  long long t = block_reduce_sum(v);
  if (threadIdx.x == 0) sums[(int64_t)blockIdx.y * nb + blockIdx.x] = t;
}

void offsets_from_lengths_segmented(const int32_t* lens, int64_t n, int segs, int64_t* offsets,
                                    int64_t* totals_host, hipStream_t s, int64_t* largest_host) {
  if (n == 0) {
    CS_HIP(hipMemsetAsync(offsets, 0, sizeof(int64_t) * segs, s));
    for (int k = 0; k < segs; ++k) totals_host[k] = 0;
    for (int k = 0; largest_host && k < segs; ++k) largest_host[k] = 0;
    return;
  }
  int64_t nb = (n + kBlock - 1) / kBlock;
  Buf sums = dev_alloc(sizeof(int64_t) * nb * segs, s);
  // [segs] totals, then (with largest_host) two maxima per segment: [0] the largest length, [1] unused here
  Buf totals = dev_alloc(sizeof(int64_t) * segs * 3, s);
  if (largest_host) CS_HIP(hipMemsetAsync(totals->p, 0, sizeof(int64_t) * segs * 3, s));
  hipLaunchKernelGGL(k_block_sums_seg, dim3((unsigned)nb, segs), dim3(kBlock), 0, s, lens, n, nb,
                     ptr<int64_t>(sums));
  hipLaunchKernelGGL(k_scan_block_sums, dim3(segs), dim3(1024), 0, s, ptr<int64_t>(sums), nb,
                     ptr<int64_t>(totals));
  hipLaunchKernelGGL(k_write_offsets, dim3((unsigned)nb, segs), dim3(kBlock), 0, s, lens, n,
                     ptr<int64_t>(sums), nb, offsets, largest_host ? ptr<unsigned long long>(totals) + segs : nullptr);
  int64_t* host = (int64_t*)pinned_scratch(sizeof(int64_t) * segs * 3);
  CS_HIP(hipMemcpyAsync(host, ptr<int64_t>(totals), sizeof(int64_t) * segs * (largest_host ? 3 : 1), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  for (int k = 0; k < segs; ++k) totals_host[k] = host[k];
  for (int k = 0; largest_host && k < segs; ++k) largest_host[k] = host[segs + 2 * k];
}

// lengths -> offsets on chunks of 2048 lengths, a wave each, with whole 16-byte loads and stores: chunk sums, the
// single-workgroup scan of the sums (49 K of them for 100 M rows), offsets.  (The kernels above work on 256
// lengths per workgroup, one per thread, with two workgroup barriers per scan: 1.0 ms for 100 M lengths against
// the 0.3 ms the 1.6 GB of traffic needs.  A one-pass form with a decoupled look-back per 1024-length tile was
// measured slower than either -- 1.23 ms: with thousands of small tiles in flight each walks many windows back.)
constexpr int kChunkRounds = 8, kChunk = kChunkRounds * 256;
// (gfx950 takes 16-byte global accesses at dword alignment: a column's lengths may start anywhere in a span array)
typedef int int4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef long long ll2_a8 __attribute__((ext_vector_type(2), aligned(8)));
struct ChunkVals {
  int v[kChunkRounds][4];
  unsigned ok[kChunkRounds];  // bit k: length k of the round is not negative (a row, not a null) and lies below n
};
__device__ __forceinline__ void load_chunk(const int32_t* __restrict__ lens, int64_t n, int64_t base, int lane, ChunkVals& c) {
#pragma unroll
  for (int j = 0; j < kChunkRounds; ++j) {
    const int64_t i = base + j * 256 + lane * 4;
    if (i + 3 < n) {
      const int4_a4 q = *reinterpret_cast<const int4_a4*>(lens + i);
      c.v[j][0] = q.x;
      c.v[j][1] = q.y;
      c.v[j][2] = q.z;
      c.v[j][3] = q.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) c.v[j][k] = i + k < n ? lens[i + k] : 0;
    }
    c.ok[j] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < n && c.v[j][k] >= 0) c.ok[j] |= 1u << k;
      c.v[j][k] = c.v[j][k] < 0 ? 0 : c.v[j][k];
    }
  }
}
__device__ __forceinline__ long long chunk_total(const ChunkVals& c) {
  long long t = 0;
#pragma unroll
  for (int j = 0; j < kChunkRounds; ++j) t += (long long)c.v[j][0] + c.v[j][1] + c.v[j][2] + c.v[j][3];
  for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d, 64);
  return t;
}
__global__ void __launch_bounds__(256) k_chunk_sums(const int32_t* __restrict__ lens, int64_t n, int64_t nchunks, int64_t* __restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= nchunks) return;
  ChunkVals c;
  load_chunk(lens, n, chunk * kChunk, lane, c);
  const long long t = chunk_total(c);
  if (lane == 0) sums[chunk] = t;
}
__global__ void __launch_bounds__(256) k_chunk_offsets(const int32_t* __restrict__ lens, int64_t n, int64_t nchunks,
                                                       const int64_t* __restrict__ chunk_base, int64_t* __restrict__ offsets,
                                                       uint8_t* __restrict__ validity, int64_t validity_len, unsigned long long* __restrict__ meta) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= nchunks) return;
  const int64_t base = chunk * kChunk;
  long long carry = chunk_base[chunk];
  ChunkVals c;
  load_chunk(lens, n, base, lane, c);
  const bool narrow = chunk_total(c) < 0x7fffffffLL;  // (wave-uniform: the sums inside the chunk fit 32 bits)
  if (meta) {
    // the column's metadata as a by-product: a lane holds four consecutive rows, sixteen lanes the 64 rows of a tile
    // that starts at a multiple of 64 (the chunk starts at a multiple of 2048)
    long long span = 0;
    int longest = 0;
#pragma unroll
    for (int j = 0; j < kChunkRounds; ++j) {
      long long s4 = (long long)c.v[j][0] + c.v[j][1] + c.v[j][2] + c.v[j][3];
      longest = max(longest, max(max(c.v[j][0], c.v[j][1]), max(c.v[j][2], c.v[j][3])));
      for (int d = 8; d > 0; d >>= 1) s4 += __shfl_xor(s4, d, 64);
      span = max(span, s4);
    }
    longest = wave_reduce_max(longest);
    for (int d = 32; d > 0; d >>= 1) span = max(span, (long long)__shfl_xor(span, d, 64));
    if (lane == 0) {
      meta_raise(meta, longest);
      meta_raise(meta + 1, span);
    }
  }
#pragma unroll
  for (int j = 0; j < kChunkRounds; ++j) {
    const int64_t i = base + j * 256 + lane * 4;
    long long first, round_total;
    if (narrow) {
      const int s4 = c.v[j][0] + c.v[j][1] + c.v[j][2] + c.v[j][3];
      const int inc = wave_inclusive_scan(s4);
      first = carry + (inc - s4);
      round_total = __builtin_amdgcn_readlane(inc, 63);
    } else {
      const long long s4 = (long long)c.v[j][0] + c.v[j][1] + c.v[j][2] + c.v[j][3];
      const long long inc = wave_inclusive_scan(s4);
      first = carry + (inc - s4);
      round_total = cstile::rl64(inc, 63);
    }
    long long o[5];
    o[0] = first;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k + 1] = o[k] + c.v[j][k];
    if (i + 3 < n) {
      *reinterpret_cast<ll2_a8*>(offsets + i) = ll2_a8{o[0], o[1]};
      *reinterpret_cast<ll2_a8*>(offsets + i + 2) = ll2_a8{o[2], o[3]};
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k < n) offsets[i + k] = o[k];
    }
    // the end of the last row: the column's byte count
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k == n - 1) offsets[n] = o[k + 1];
    // the validity bits of the round's 256 rows: two lanes' nibbles make a byte
    if (validity) {
      const unsigned pair = c.ok[j] | ((unsigned)__shfl_down((int)c.ok[j], 1, 64) << 4);
      const int64_t at = ((base + j * 256) >> 3) + (lane >> 1);
      if (!(lane & 1) && at < validity_len) validity[at] = (uint8_t)pair;
    }
    carry += round_total;
  }
}

static int64_t offsets_by_chunks(const int32_t* lens, int64_t n, int64_t* offsets, uint8_t* validity, hipStream_t s, LenMeta* meta);
static int64_t offsets_by_workgroups(const int32_t* lens, int64_t n, int64_t* offsets, hipStream_t s, Buf block_sums, LenMeta* meta);
int64_t offsets_from_lengths(const int32_t* lens, int64_t n, int64_t* offsets, hipStream_t s,
                             Buf block_sums, LenMeta* meta) {
  if (n == 0) {
    CS_HIP(hipMemsetAsync(offsets, 0, sizeof(int64_t), s));
    if (meta) meta->max_row = meta->max_span64 = 0;
    return 0;
  }
  if (!block_sums && !cs::cfg("CS_SCAN_BY_WORKGROUPS")) return offsets_by_chunks(lens, n, offsets, nullptr, s, meta);
  return offsets_by_workgroups(lens, n, offsets, s, block_sums, meta);
}
// the same scan, nothing read back: the total is offsets[n] in device memory, the stream is not waited for (the radix sort's
// eight passes each scan their tile counts -- a round trip to the host per pass was most of its time on a million keys)
void offsets_from_lengths_async(const int32_t* lens, int64_t n, int64_t* offsets, hipStream_t s) {
  if (n == 0) {
    CS_HIP(hipMemsetAsync(offsets, 0, sizeof(int64_t), s));
    return;
  }
  const int64_t nchunks = (n + kChunk - 1) / kChunk;
  Buf sums = dev_alloc(sizeof(int64_t) * nchunks, s);
  Buf total = dev_alloc(3 * sizeof(int64_t), s);
  const unsigned grid = (unsigned)((nchunks + 3) / 4);
  hipLaunchKernelGGL(k_chunk_sums, dim3(grid), dim3(kBlock), 0, s, lens, n, nchunks, ptr<int64_t>(sums));
  hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, ptr<int64_t>(sums), nchunks, ptr<int64_t>(total));
  hipLaunchKernelGGL(k_chunk_offsets, dim3(grid), dim3(kBlock), 0, s, lens, n, nchunks, ptr<const int64_t>(sums), offsets, (uint8_t*)nullptr, (int64_t)validity_bytes(n),
                     (unsigned long long*)nullptr);
  CS_HIP(hipGetLastError());
}
// offsets and the validity mask (length >= 0) of a column in the same pass over its lengths
int64_t offsets_and_validity_from_lengths(const int32_t* lens, int64_t n, int64_t* offsets, Buf* validity, hipStream_t s, LenMeta* meta) {
  if (n == 0 || cs::cfg("CS_SCAN_BY_WORKGROUPS")) {
    *validity = validity_from_lengths(lens, n, s);
    return offsets_from_lengths(lens, n, offsets, s, nullptr, meta);
  }
  *validity = dev_alloc(validity_bytes(n), s);
  return offsets_by_chunks(lens, n, offsets, ptr<uint8_t>(*validity), s, meta);
}
// (the total and the two metadata words share one small buffer and one copy back: no extra synchronisation)
static int64_t offsets_by_chunks(const int32_t* lens, int64_t n, int64_t* offsets, uint8_t* validity, hipStream_t s, LenMeta* meta) {
  {
    const int64_t nchunks = (n + kChunk - 1) / kChunk;
    Buf sums = dev_alloc(sizeof(int64_t) * nchunks, s);
    Buf total = dev_alloc(3 * sizeof(int64_t), s);
    if (meta) CS_HIP(hipMemsetAsync(total->p, 0, 3 * sizeof(int64_t), s));
    const unsigned grid = (unsigned)((nchunks + 3) / 4);
    hipLaunchKernelGGL(k_chunk_sums, dim3(grid), dim3(kBlock), 0, s, lens, n, nchunks, ptr<int64_t>(sums));
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, ptr<int64_t>(sums), nchunks, ptr<int64_t>(total));
    {
      ProfScope ps("k_write_offsets", s);
      hipLaunchKernelGGL(k_chunk_offsets, dim3(grid), dim3(kBlock), 0, s, lens, n, nchunks, ptr<const int64_t>(sums), offsets, validity,
                         (int64_t)validity_bytes(n), meta ? ptr<unsigned long long>(total) + 1 : nullptr);
    }
    int64_t* host = (int64_t*)pinned_scratch(3 * sizeof(int64_t));
    CS_HIP(hipMemcpyAsync(host, ptr<int64_t>(total), (meta ? 3 : 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (meta) {
      meta->max_row = host[1];
      meta->max_span64 = host[2];
    }
    return host[0];
  }
}
static int64_t offsets_by_workgroups(const int32_t* lens, int64_t n, int64_t* offsets, hipStream_t s, Buf block_sums, LenMeta* meta) {
  int64_t nb = (n + kBlock - 1) / kBlock;
  Buf sums = block_sums;
  if (!sums) {
    sums = dev_alloc(sizeof(int64_t) * nb, s);
    hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nb), dim3(kBlock), 0, s, lens, n, ptr<int64_t>(sums));
  }
  Buf total = dev_alloc(3 * sizeof(int64_t), s);
  if (meta) CS_HIP(hipMemsetAsync(total->p, 0, 3 * sizeof(int64_t), s));
  hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, ptr<int64_t>(sums), nb,
                     ptr<int64_t>(total));
  {
    ProfScope ps("k_write_offsets", s);
    hipLaunchKernelGGL(k_write_offsets, dim3((unsigned)nb, 1), dim3(kBlock), 0, s, lens, n,
                       ptr<int64_t>(sums), nb, offsets, meta ? ptr<unsigned long long>(total) + 1 : nullptr);
  }
  int64_t* host = (int64_t*)pinned_scratch(3 * sizeof(int64_t));
  CS_HIP(hipMemcpyAsync(host, ptr<int64_t>(total), (meta ? 3 : 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (meta) {
    meta->max_row = host[1];
    meta->max_span64 = host[2];
  }
  return host[0];
}

__global__ void k_validity_from_lengths(const int32_t* __restrict__ lens, int64_t n,
                                        uint8_t* __restrict__ validity) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool ok = i < n && lens[i] >= 0;
  store_validity_word(validity, i - (threadIdx.x & 63), ok, n);
}
Buf validity_from_lengths(const int32_t* lens, int64_t n, hipStream_t s) {
  Buf v = dev_alloc(validity_bytes(n), s);
  if (n)
    hipLaunchKernelGGL(k_validity_from_lengths, dim3(blocks_for(n)), dim3(kBlock), 0, s, lens, n,
                       ptr<uint8_t>(v));
  return v;
}

// one thread per 64-row validity word (the bits past `rows` in the last word are zero)
__global__ void k_count_valid(const uint8_t* __restrict__ validity, int64_t rows,
                              unsigned long long* __restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int v = 0;
  if (w * 64 < rows) {
    unsigned long long bits = reinterpret_cast<const unsigned long long*>(validity)[w];
    const int64_t left = rows - w * 64;
    if (left < 64) bits &= (1ull << left) - 1ull;
    v = __builtin_popcountll(bits);
  }
  long long t = block_reduce_sum(v);
  if (threadIdx.x == 0 && t) atomicAdd(out, (unsigned long long)t);
}
int64_t count_nulls(const cs_column* c, hipStream_t s) {
  if (c->null_count >= 0) return c->null_count;
  if (!c->validity || c->rows == 0) return c->null_count = 0;
  Buf cnt = dev_alloc(8, s);
  CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
  hipLaunchKernelGGL(k_count_valid, dim3(blocks_for((c->rows + 63) / 64)), dim3(kBlock), 0, s, c->d_validity(),
                     c->rows, ptr<unsigned long long>(cnt));
  int64_t* host = (int64_t*)pinned_scratch(8);
  CS_HIP(hipMemcpyAsync(host, cnt->p, 8, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return c->null_count = c->rows - host[0];
}

__global__ void k_max_span64(const int64_t* __restrict__ offsets, int64_t rows, unsigned long long* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int64_t r0 = t * 64;
  int v = 0;
  if (r0 < rows) {
    int64_t r1 = r0 + 64 < rows ? r0 + 64 : rows;
    v = (int)(offsets[r1] - offsets[r0] > 0x7fffffff ? 0x7fffffff : offsets[r1] - offsets[r0]);
  }
  int m = block_reduce_max(v);
  if (threadIdx.x == 0 && (unsigned long long)m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, (unsigned long long)m);
}
// Number of 256-thread workgroups of `kern` that are resident at once on this device
// (capped by `wanted`): the persistent kernels' look-back needs every wave of the grid
// to be running.
unsigned resident_grid(const void* kern, size_t lds, int64_t wanted) {
  // the occupancy query costs about a millisecond: remember the answers
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, std::pair<int, int>> cache;
  int cus = 0, per = 0;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find({kern, lds});
    if (it == cache.end()) {
      int dev = 0;
      CS_HIP(hipGetDevice(&dev));
      CS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      CS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, 256, lds));
      cache[{kern, lds}] = {cus, per};
    } else {
      cus = it->second.first;
      per = it->second.second;
    }
  }
  if (per < 1) per = 1;
  if (const char* e = cs::cfg("CS_STREAM_BLOCKS_PER_CU")) per = std::max(1, std::min(per, atoi(e)));
  const int64_t cap = (int64_t)cus * per;
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cap, wanted));
}

__global__ void k_max_span_rows(const int64_t* __restrict__ offsets, int64_t rows, int per,
                                unsigned long long* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int64_t r0 = t * per;
  int v = 0;
  if (r0 < rows) {
    int64_t r1 = r0 + per < rows ? r0 + per : rows;
    v = (int)(offsets[r1] - offsets[r0] > 0x7fffffff ? 0x7fffffff : offsets[r1] - offsets[r0]);
  }
  int m = block_reduce_max(v);
  // (only a workgroup that would raise the maximum issues the same-address atomic: 390 K of them in a row cost
  // 4.4 ms on a 100M-row column, more than most kernels that ask for this number)
  if (threadIdx.x == 0 && (unsigned long long)m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, (unsigned long long)m);
}
int64_t max_span_rows(const cs_column* c, int per, hipStream_t s) {
  if (per == 64) return max_span64(c, s);
  if (c->rows == 0) return 0;
  Buf acc = dev_alloc(8, s);
  CS_HIP(hipMemsetAsync(acc->p, 0, 8, s));
  int64_t n = (c->rows + per - 1) / per;
  hipLaunchKernelGGL(k_max_span_rows, dim3(blocks_for(n)), dim3(kBlock), 0, s, c->d_offsets(), c->rows, per,
                     ptr<unsigned long long>(acc));
  int64_t* host = (int64_t*)pinned_scratch(8);
  CS_HIP(hipMemcpyAsync(host, acc->p, 8, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return host[0];
}
// How many 64-row tiles span more than `limit` bytes: a column whose LARGEST tile does not fit a tile kernel's staging
// buffer may still have all but a few that do (one long row among millions of short ones) -- the tile kernels then take
// the column and handle the oversize tiles a thread per row themselves, instead of the whole column going row-wise.
__global__ void k_count_spans_over(const int64_t* __restrict__ offsets, int64_t rows, int64_t limit, unsigned long long* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t r0 = t * 64;
  int v = 0;
  if (r0 < rows) {
    const int64_t r1 = r0 + 64 < rows ? r0 + 64 : rows;
    v = offsets[r1] - offsets[r0] > limit ? 1 : 0;
  }
  const long long n = block_reduce_sum(v);
  if (threadIdx.x == 0 && n) atomicAdd(out, (unsigned long long)n);
}
int64_t count_spans64_over(const cs_column* c, int64_t limit, hipStream_t s) {
  if (c->rows == 0) return 0;
  if (max_span64(c, s) <= limit) return 0;
  Buf acc = dev_alloc(8, s);
  CS_HIP(hipMemsetAsync(acc->p, 0, 8, s));
  const int64_t nsub = (c->rows + 63) / 64;
  hipLaunchKernelGGL(k_count_spans_over, dim3(blocks_for(nsub)), dim3(kBlock), 0, s, c->d_offsets(), c->rows, limit, ptr<unsigned long long>(acc));
  int64_t* host = (int64_t*)pinned_scratch(8);
  CS_HIP(hipMemcpyAsync(host, acc->p, 8, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return host[0];
}
// true when all but a few (at most 8, or one in a thousand) of the 64-row tiles span at most `limit` bytes
bool few_spans64_over(const cs_column* c, int64_t limit, hipStream_t s) {
  const int64_t nsub = (c->rows + 63) / 64;
  const int64_t over = count_spans64_over(c, limit, s);
  return over <= std::max<int64_t>(8, nsub / 1000);
}
// A hint for choosing between kernel forms (never for correctness): does the column look like non-ASCII text?  Three windows
// of 64 KiB (start, middle, end of the chars), cached on the immutable column.
__global__ void k_sample_high(const uint8_t* __restrict__ chars, int64_t nbytes, int* __restrict__ out) {
  const int64_t win = 64 * 1024;
  const int64_t base = blockIdx.x == 0 ? 0 : (blockIdx.x == 1 ? (nbytes / 2) & ~(int64_t)15 : (nbytes > win ? (nbytes - win) & ~(int64_t)15 : 0));
  uint32_t any = 0;
  for (int64_t i = base + (int64_t)threadIdx.x * 16; i + 16 <= nbytes && i < base + win; i += (int64_t)blockDim.x * 16) {
    const uint4 q = *reinterpret_cast<const uint4*>(chars + i);
    any |= (q.x | q.y | q.z | q.w) & 0x80808080u;
  }
  if (any) *out = 1;
}
bool sample_has_high_bytes(const cs_column* c, hipStream_t s) {
  if (c->high_sample >= 0) return c->high_sample != 0;
  if (c->nbytes < 16 || !c->chars) return (c->high_sample = 0) != 0;
  Buf flag = dev_alloc(sizeof(int), s);
  CS_HIP(hipMemsetAsync(flag->p, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_sample_high, dim3(3), dim3(256), 0, s, c->d_chars(), c->nbytes, ptr<int>(flag));
  int* host = (int*)pinned_scratch(sizeof(int));
  CS_HIP(hipMemcpyAsync(host, flag->p, sizeof(int), hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return (c->high_sample = host[0] ? 1 : 0) != 0;
}
// Byte counts over the same three windows: how common a pattern's candidate bytes are in THIS column (regex route choice).
__global__ void __launch_bounds__(256) k_sample_hist(const uint8_t* __restrict__ chars, int64_t nbytes, uint32_t* __restrict__ out) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t win = 64 * 1024;
  const int64_t base = blockIdx.x == 0 ? 0 : (blockIdx.x == 1 ? (nbytes / 2) & ~(int64_t)15 : (nbytes > win ? (nbytes - win) & ~(int64_t)15 : 0));
  // (windows that coincide on a short column are counted once)
  const bool dup = (blockIdx.x == 1 && base == 0) || (blockIdx.x == 2 && (base == 0 || base == ((nbytes / 2) & ~(int64_t)15)));
  if (!dup)
    for (int64_t i = base + threadIdx.x; i < nbytes && i < base + win; i += blockDim.x) atomicAdd(&h[chars[i]], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(out + threadIdx.x, h[threadIdx.x]);
}
const uint32_t* sample_byte_hist(const cs_column* c, hipStream_t s) {
  static std::mutex mu;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (c->byte_hist) return c->byte_hist->data();
  }
  auto hist = std::make_shared<std::array<uint32_t, 256>>();
  hist->fill(0);
  if (c->nbytes > 0 && c->chars) {
    Buf acc = dev_alloc(256 * sizeof(uint32_t), s);
    CS_HIP(hipMemsetAsync(acc->p, 0, 256 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_sample_hist, dim3(3), dim3(256), 0, s, c->d_chars(), c->nbytes, ptr<uint32_t>(acc));
    uint32_t* host = (uint32_t*)pinned_scratch(256 * sizeof(uint32_t));
    CS_HIP(hipMemcpyAsync(host, acc->p, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    std::copy(host, host + 256, hist->begin());
  }
  std::lock_guard<std::mutex> lk(mu);
  if (!c->byte_hist) c->byte_hist = hist;
  return c->byte_hist->data();
}
int64_t max_row_bytes(const cs_column* c, hipStream_t s) {
  if (c->max_row >= 0) return c->max_row;
  return c->max_row = max_span_rows(c, 1, s);
}
int64_t max_span64(const cs_column* c, hipStream_t s) {
  if (c->max_span64 >= 0) return c->max_span64;
  if (c->rows == 0) return c->max_span64 = 0;
  Buf acc = dev_alloc(8, s);
  CS_HIP(hipMemsetAsync(acc->p, 0, 8, s));
  int64_t nsub = (c->rows + 63) / 64;
  hipLaunchKernelGGL(k_max_span64, dim3(blocks_for(nsub)), dim3(kBlock), 0, s, c->d_offsets(), c->rows,
                     ptr<unsigned long long>(acc));
  int64_t* host = (int64_t*)pinned_scratch(8);
  CS_HIP(hipMemcpyAsync(host, acc->p, 8, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return c->max_span64 = host[0];
}

// One streaming pass over the chars buffer, remembered on the (immutable) column: true when
// the bytes hold no NUL and no UTF-8 lead byte whose announced continuation positions hold an
// ASCII byte.  On such a column a per-character scan and a per-byte scan see every ASCII byte
// at a character boundary, so an ASCII needle may be searched either way (cs_replace routes a
// literal needle through the regex stream kernel on the strength of this).
__global__ void __launch_bounds__(256) k_bytes_plain(const uint8_t* __restrict__ chars, int64_t nbytes, unsigned* __restrict__ flag) {
  const int64_t pieces = (nbytes + 15) / 16;
  uint32_t hit = 0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pieces; p += (int64_t)gridDim.x * 256) {
    uint32_t w[5];
    w[0] = p > 0 ? *reinterpret_cast<const uint32_t*>(chars + 16 * p - 4) : 0x20202020u;
    if (16 * p + 16 <= nbytes) {
      const uint4 q = *reinterpret_cast<const uint4*>(chars + 16 * p);
      w[1] = q.x, w[2] = q.y, w[3] = q.z, w[4] = q.w;
    } else {
      for (int k = 1; k < 5; ++k) w[k] = 0x20202020u;
      for (int64_t i = 16 * p; i < nbytes; ++i) {
        const int j = (int)(i - 16 * p);
        w[1 + (j >> 2)] = (w[1 + (j >> 2)] & ~(0xFFu << (8 * (j & 3)))) | ((uint32_t)chars[i] << (8 * (j & 3)));
      }
    }
    uint32_t l2[5], l3[5], l4[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      l2[k] = w[k] & (w[k] << 1) & 0x80808080u;  // bytes >= 0xC0
      l3[k] = l2[k] & (w[k] << 2);               // >= 0xE0
      l4[k] = l3[k] & (w[k] << 3);               // >= 0xF0
    }
#pragma unroll
    for (int k = 1; k < 5; ++k) {
      const uint32_t announced = __funnelshift_l(l2[k - 1], l2[k], 8) | __funnelshift_l(l3[k - 1], l3[k], 16) |
                                 __funnelshift_l(l4[k - 1], l4[k], 24);
      hit |= (announced | (w[k] - 0x01010101u)) & ~w[k] & 0x80808080u;
    }
  }
  if (__any(hit != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
bool bytes_plain(const cs_column* c, hipStream_t s) {
  if (c->plain_bytes >= 0) return c->plain_bytes != 0;
  if (c->nbytes == 0) return (c->plain_bytes = 1) != 0;
  // (a wrapped caller buffer at an odd address: the check reads aligned 16-byte pieces -- decline)
  if ((uintptr_t)c->d_chars() & 15) return (c->plain_bytes = 0) != 0;
  Buf acc = dev_alloc(8, s);
  CS_HIP(hipMemsetAsync(acc->p, 0, 8, s));
  const int64_t pieces = (c->nbytes + 15) / 16;
  const unsigned grid = (unsigned)std::min<int64_t>((pieces + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_bytes_plain, dim3(grid), dim3(256), 0, s, c->d_chars(), c->nbytes, ptr<unsigned>(acc));
  CS_HIP(hipGetLastError());
  unsigned* host = (unsigned*)pinned_scratch(8);
  CS_HIP(hipMemcpyAsync(host, acc->p, 4, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return (c->plain_bytes = host[0] ? 0 : 1) != 0;
}

cs_column* make_all_null(int64_t rows, hipStream_t s) {
  auto* c = new cs_column;
  c->rows = rows;
  c->nbytes = 0;
  c->null_count = rows;
  c->chars = dev_alloc(0, s);
  c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
  CS_HIP(hipMemsetAsync(c->offsets->p, 0, sizeof(int64_t) * (rows + 1), s));
  if (rows) {
    c->validity = dev_alloc(validity_bytes(rows), s);
    CS_HIP(hipMemsetAsync(c->validity->p, 0, validity_bytes(rows), s));
  }
  return c;
}

// ------------------------------------------------------- ingest / export ----
__global__ void k_widen_offsets(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void k_narrow_offsets(const int64_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}
}  // namespace cs
// int64 form of a column born with int32 offsets, built once (any thread) and kept
static std::mutex g_widen_mu;
const int64_t* cs_column::d_offsets() const {
  std::lock_guard<std::mutex> lk(g_widen_mu);
  if (!offsets && offsets32) {
    cs::Buf o64 = cs::dev_alloc(sizeof(int64_t) * (rows + 1), nullptr);
    hipLaunchKernelGGL(cs::k_widen_offsets, dim3(cs::blocks_for(rows + 1)), dim3(kBlock), 0, nullptr, d_offsets32(), rows + 1,
                       cs::ptr<int64_t>(o64));
    CS_HIP(hipGetLastError());
    CS_HIP(hipStreamSynchronize(nullptr));
    offsets = o64;
  }
  return cs::ptr<const int64_t>(offsets);
}
void cs_column::share_extents_with(cs_column* o) const {
  std::lock_guard<std::mutex> lk(g_widen_mu);
  o->offsets = offsets;
  o->offsets32 = offsets32;
}
namespace cs {
// a null row contributes no bytes: clamp its extent to zero while copying
// (NVStringsImpl.cu:408-432 skips rows whose validity bit is clear)
__global__ void k_lengths_from_offsets(cs::ColView in, int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows) return;
  lens[r] = row_is_valid(in.validity, r) ? (int32_t)(in.offsets[r + 1] - in.offsets[r]) : -1;
}
__global__ void k_gather_rows(cs::ColView in, const int64_t* __restrict__ out_off,
                              uint8_t* __restrict__ out_chars) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows || !row_is_valid(in.validity, r)) return;
  const uint8_t* src = in.chars + in.offsets[r];
  uint8_t* dst = out_chars + out_off[r];
  int n = (int)(in.offsets[r + 1] - in.offsets[r]);
  for (int i = 0; i < n; ++i) dst[i] = src[i];
}
__global__ void k_null_bitarray(cs::ColView in, int empty_is_null, uint8_t* __restrict__ bits,
                                unsigned long long* __restrict__ nulls) {
  // one thread per output byte (NVStrings.cu:512-527)
  int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int64_t nb = (in.rows + 7) / 8;
  int cleared = 0;
  if (b < nb) {
    unsigned byte = 0;
    for (int i = 0; i < 8; ++i) {
      int64_t r = b * 8 + i;
      if (r >= in.rows) break;
      bool ok = row_is_valid(in.validity, r);
      if (ok && empty_is_null) ok = in.offsets[r + 1] > in.offsets[r];
      if (ok) byte |= 1u << i;
      else ++cleared;
    }
    bits[b] = (uint8_t)byte;
  }
  long long t = block_reduce_sum(cleared);
  if (threadIdx.x == 0 && t) atomicAdd(nulls, (unsigned long long)t);
}
__global__ void k_digest(cs::ColView in, unsigned long long* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned long long h = 0;
  if (r < in.rows) {
    bool ok = row_is_valid(in.validity, r);
    int n = ok ? (int)(in.offsets[r + 1] - in.offsets[r]) : 0;
    h = cs_digest_row((uint64_t)r, ok ? in.chars + in.offsets[r] : nullptr, n, ok);
  }
  for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d, 64);
  if ((threadIdx.x & 63) == 0 && h) atomicAdd(out, h);
}

__global__ void k_slice(cs::ColView in, int64_t first, int64_t rows, int64_t* __restrict__ out_off,
                        uint8_t* __restrict__ out_valid) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t base = in.offsets[first];
  if (i <= rows) out_off[i] = in.offsets[first + i] - base;
  if (out_valid) {
    bool ok = i < rows && row_is_valid(in.validity, first + i);
    store_validity_word(out_valid, i - (threadIdx.x & 63), ok, rows);
  }
}

static void copy_in(void* dst, const void* src, size_t bytes, int on_device, hipStream_t s) {
  if (!bytes) return;
  CS_HIP(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
}
static void copy_out(void* dst, const void* src, size_t bytes, int on_device, hipStream_t s) {
  if (!bytes) return;
  CS_HIP(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
}

// sets *flag when a null row has a non-empty extent
__global__ void k_null_rows_hold_bytes(cs::ColView in, unsigned* __restrict__ flag) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool bad = false;
  if (r < in.rows) bad = !row_is_valid(in.validity, r) && in.offsets[r + 1] != in.offsets[r];
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
// Turns an ingested (chars, offsets, validity) triple into a canonical column:
// offsets[0] == 0, null rows have zero extent, chars packed.
static cs_column* canonicalise(Buf chars, Buf offs, Buf validity, int64_t rows, int64_t span_bytes,
                               bool has_validity, hipStream_t s) {
  auto* c = new cs_column;
  std::unique_ptr<cs_column> guard_c(c);
  c->rows = rows;
  if (!has_validity) {
    // offsets may start at a non-zero base: rebase by gathering only when needed
    int64_t first = 0;
    int64_t* host = (int64_t*)pinned_scratch(16);
    CS_HIP(hipMemcpyAsync(host, offs->p, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(host + 1, ptr<int64_t>(offs) + rows, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    first = host[0];
    if (first == 0) {
      c->chars = chars;
      c->offsets = offs;
      c->nbytes = host[1];
      c->null_count = 0;
      (void)span_bytes;
      return guard_c.release();
    }
  }
  ColView in{ptr<const uint8_t>(chars), ptr<const int64_t>(offs), ptr<const uint8_t>(validity), rows};
  if (has_validity) {
    // The usual Arrow column is canonical already (offsets from 0, nothing stored under a null):
    // one pass over offsets + bitmask decides, and the ingested buffers are then used as they are.
    Buf flag = dev_alloc(sizeof(unsigned), s);
    CS_HIP(hipMemsetAsync(flag->p, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_null_rows_hold_bytes, dim3(blocks_for(rows)), dim3(kBlock), 0, s, in, ptr<unsigned>(flag));
    int64_t* host = (int64_t*)pinned_scratch(24);
    CS_HIP(hipMemcpyAsync(host, offs->p, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(host + 1, ptr<int64_t>(offs) + rows, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(host + 2, flag->p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (host[0] == 0 && (unsigned)host[2] == 0) {
      c->chars = chars;
      c->offsets = offs;
      c->validity = validity;
      c->nbytes = host[1];
      return guard_c.release();
    }
  }
  Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
  hipLaunchKernelGGL(k_lengths_from_offsets, dim3(blocks_for(rows)), dim3(kBlock), 0, s, in,
                     ptr<int32_t>(lens));
  c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
  c->nbytes = offsets_from_lengths(ptr<int32_t>(lens), rows, ptr<int64_t>(c->offsets), s);
  c->chars = dev_alloc((size_t)c->nbytes, s);
  hipLaunchKernelGGL(k_gather_rows, dim3(blocks_for(rows)), dim3(kBlock), 0, s, in,
                     c->d_offsets(), ptr<uint8_t>(c->chars));
  c->validity = validity;
  return guard_c.release();
}

__global__ void k_gather_rows_at(cs::ColView in, const int64_t* __restrict__ out_off,
                                 uint8_t* __restrict__ out_chars) {
  // out_off is already advanced to this input's first output row
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= in.rows || !row_is_valid(in.validity, r)) return;
  const uint8_t* src = in.chars + in.offsets[r];
  uint8_t* dst = out_chars + out_off[r];
  int n = (int)(in.offsets[r + 1] - in.offsets[r]);
  for (int i = 0; i < n; ++i) dst[i] = src[i];
}
// ---- record (row-major) forms: columns -> one flat column + list offsets ---------------------------
// The reference's *_record methods return one NVStrings instance per row (extract_record.cu:146-152,
// findall_record.cu:144-151: a device allocation per row).  Natively a record result is ONE column whose rows are
// the records' strings in row-major order plus rows+1 list offsets: record r = flat rows [list[r], list[r+1]).
struct RecordCols {
  static constexpr int kMax = 64;
  cs::ColView col[kMax];
};
// (the columns go through the kernels kMax at a time: c.col[j] is column k0 + j)
// entries per record: all columns (fixed) or the row's leading non-null columns (ragged)
__global__ void k_record_counts(RecordCols c, int k0, int nb, int ncols, int ragged, int64_t rows, int32_t* __restrict__ counts) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  if (!ragged) {
    counts[r] = ncols;
    return;
  }
  int n = k0 ? counts[r] : 0;
  if (n == k0) {  // every earlier column was non-null for this row
    int j = 0;
    while (j < nb && row_is_valid(c.col[j].validity, r)) ++j;
    n += j;
  }
  counts[r] = n;
}
__global__ void k_record_lengths(RecordCols c, int k0, int nb, int64_t rows, const int64_t* __restrict__ list,
                                 int32_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  const int64_t e0 = list[r];
  const int n = (int)(list[r + 1] - e0);
  for (int j = 0; j < nb && k0 + j < n; ++j) {
    const cs::ColView& v = c.col[j];
    lens[e0 + k0 + j] = row_is_valid(v.validity, r) ? (int32_t)(v.offsets[r + 1] - v.offsets[r]) : -1;
  }
}
__global__ void k_record_copy(RecordCols c, int k0, int nb, int64_t rows, const int64_t* __restrict__ list,
                              const int64_t* __restrict__ out_off, uint8_t* __restrict__ out_chars) {
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  const int64_t e0 = list[r];
  const int n = (int)(list[r + 1] - e0);
  for (int j = 0; j < nb && k0 + j < n; ++j) {
    const cs::ColView& v = c.col[j];
    if (!row_is_valid(v.validity, r)) continue;
    const uint8_t* src = v.chars + v.offsets[r];
    uint8_t* dst = out_chars + out_off[e0 + k0 + j];
    const int len = (int)(v.offsets[r + 1] - v.offsets[r]);
    for (int i = 0; i < len; ++i) dst[i] = src[i];
  }
}

cs_column* concat_columns(const std::vector<const cs_column*>& cols, hipStream_t s) {
  int64_t rows = 0;
  bool any_mask = false;
  for (auto* c : cols) {
    rows += c->rows;
    any_mask |= c->validity != nullptr;
  }
  if (rows == 0) return make_all_null(0, s);
  auto out = std::make_unique<cs_column>();
  out->rows = rows;
  Buf lens = dev_alloc(sizeof(int32_t) * rows, s);
  int64_t base = 0;
  for (auto* c : cols) {
    if (c->rows)
      hipLaunchKernelGGL(k_lengths_from_offsets, dim3(blocks_for(c->rows)), dim3(kBlock), 0, s, view_of(c),
                         ptr<int32_t>(lens) + base);
    base += c->rows;
  }
  out->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
  out->nbytes = offsets_from_lengths(ptr<int32_t>(lens), rows, ptr<int64_t>(out->offsets), s);
  out->chars = dev_alloc((size_t)out->nbytes, s);
  if (any_mask) out->validity = validity_from_lengths(ptr<int32_t>(lens), rows, s);
  else out->null_count = 0;
  base = 0;
  for (auto* c : cols) {
    if (c->rows)
      hipLaunchKernelGGL(k_gather_rows_at, dim3(blocks_for(c->rows)), dim3(kBlock), 0, s, view_of(c),
                         out->d_offsets() + base, ptr<uint8_t>(out->chars));
    base += c->rows;
  }
  CS_HIP(hipStreamSynchronize(s));
  return out.release();
}

}  // namespace cs

using namespace cs;

// =============================================================== public ABI ==
extern "C" {

int cs_version(void) { return 100; }
int cs_has_experiments(void) {
#if defined(CS_EXPERIMENTS)
  return 1;
#else
  return 0;
#endif
}
const char* cs_last_error(void) { return g_last_error.c_str(); }

int cs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int cs_init(int device) {
  return guard([&] {
    (void)cs::cfg("CS_DEVICE");  // (the CS_* switches are read from the environment here, once: cs_config.h)
    int n = cs_device_count();
    if (n <= 0) fail(CS_ERR_NO_DEVICE, "no HIP device visible (there is no CPU fallback)");
    if (device < 0 || device >= n) fail(CS_ERR_INVALID_ARG, "device index out of range");
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_device == device) {
      CS_HIP(hipSetDevice(device));
      return;
    }
    if (g_device >= 0) fail(CS_ERR_INVALID_ARG, "this process is already bound to another device (one process per GPU)");
    CS_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    CS_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      fail(CS_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    CS_HIP(hipMalloc((void**)&g_d_flags, 65536));
    CS_HIP(hipMalloc((void**)&g_d_cases, 65536 * 2));
    CS_HIP(hipMemcpy(g_d_flags, cs_unicode_flags, 65536, hipMemcpyHostToDevice));
    CS_HIP(hipMemcpy(g_d_cases, cs_charcases, 65536 * 2, hipMemcpyHostToDevice));
    g_device = device;
  });
}

int cs_current_device(void) { return g_device; }
int64_t cs_fallback_count(void) { return (int64_t)g_fallbacks.load(); }
const char* cs_debug_last_route(void) { return cs::g_last_route; }
int cs_config_set(const char* name, const char* value) {
  return guard([&] {
    if (!name || std::strncmp(name, "CS_", 3) != 0) fail(CS_ERR_INVALID_ARG, "config: the switches are named CS_*");
    cs::cfg_set(name, value);
  });
}
int64_t cs_device_bytes_in_use(void) { return dev_bytes_in_use(); }
int64_t cs_debug_malloc_count(void) { return (int64_t)g_mallocs.load(); }
int64_t cs_pool_cached_bytes(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_cached_bytes;
}
int cs_pool_trim(int64_t keep_bytes) {
  return guard([&] {
    require_device();
    CS_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t limit = g_cache_limit;
    g_cache_limit = std::max<int64_t>(keep_bytes, 1);
    trim_cache_locked();
    g_cache_limit = limit;
  });
}
void cs_free(void* p) { free(p); }

int cs_column_from_host_strings(const char* const* strs, int64_t rows, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!out || (rows > 0 && !strs) || rows < 0) fail(CS_ERR_INVALID_ARG, "create_from_array: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    // one staging buffer, one H2D copy (NVStringsImpl.cu:126-206)
    std::vector<int64_t> off((size_t)rows + 1, 0);
    bool any_null = false;
    for (int64_t r = 0; r < rows; ++r) {
      size_t n = strs[r] ? strlen(strs[r]) : 0;
      any_null |= strs[r] == nullptr;
      off[r + 1] = off[r] + (int64_t)n;
    }
    std::vector<uint8_t> chars((size_t)off[rows]);
    std::vector<uint8_t> valid;
    if (any_null) valid.assign(validity_bytes(rows), 0);
    for (int64_t r = 0; r < rows; ++r) {
      if (!strs[r]) continue;
      memcpy(chars.data() + off[r], strs[r], (size_t)(off[r + 1] - off[r]));
      if (any_null) valid[r >> 3] |= (uint8_t)(1u << (r & 7));
    }
    auto* c = new cs_column;
    std::unique_ptr<cs_column> holder(c);
    c->rows = rows;
    c->nbytes = off[rows];
    c->chars = dev_alloc(chars.size(), s);
    c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    copy_in(c->chars->p, chars.data(), chars.size(), 0, s);
    copy_in(c->offsets->p, off.data(), sizeof(int64_t) * (rows + 1), 0, s);
    if (any_null) {
      c->validity = dev_alloc(valid.size(), s);
      copy_in(c->validity->p, valid.data(), valid.size(), 0, s);
    } else {
      c->null_count = 0;
    }
    CS_HIP(hipStreamSynchronize(s));  // staging vectors die here
    *out = holder.release();
  });
}

int cs_column_from_offsets32(const char* chars, int64_t rows, const int32_t* offsets,
                             const uint8_t* validity, int on_device, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!out || rows < 0 || (rows > 0 && (!offsets))) fail(CS_ERR_INVALID_ARG, "create_from_offsets: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    Buf o32 = dev_alloc(sizeof(int32_t) * (rows + 1), s);
    copy_in(o32->p, offsets, sizeof(int32_t) * (rows + 1), on_device, s);
    int32_t ends[2];
    CS_HIP(hipMemcpyAsync(&ends[0], ptr<int32_t>(o32), 4, hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(&ends[1], ptr<int32_t>(o32) + rows, 4, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (ends[1] < ends[0] || ends[0] < 0) fail(CS_ERR_INVALID_ARG, "create_from_offsets: offsets are not ascending");
    int64_t span = ends[1];
    if (span > 0 && !chars) fail(CS_ERR_INVALID_ARG, "create_from_offsets: chars is null");
    Buf o64 = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    hipLaunchKernelGGL(k_widen_offsets, dim3(blocks_for(rows + 1)), dim3(kBlock), 0, s,
                       ptr<int32_t>(o32), rows + 1, ptr<int64_t>(o64));
    Buf ch = dev_alloc((size_t)span, s);
    copy_in(ch->p, chars, (size_t)span, on_device, s);
    Buf v;
    if (validity) {
      v = dev_alloc(validity_bytes(rows), s);
      CS_HIP(hipMemsetAsync(v->p, 0, validity_bytes(rows), s));
      copy_in(v->p, validity, (size_t)((rows + 7) / 8), on_device, s);
    }
    *out = canonicalise(ch, o64, v, rows, span, validity != nullptr, s);
  });
}

int cs_column_from_offsets64(const uint8_t* chars, int64_t rows, const int64_t* offsets,
                             const uint8_t* validity, int on_device, int copy, cs_stream stream,
                             cs_column** out) {
  return guard([&] {
    if (!out || rows < 0 || (rows > 0 && !offsets)) fail(CS_ERR_INVALID_ARG, "from_offsets64: bad arguments");
    if (!copy && !on_device) fail(CS_ERR_INVALID_ARG, "from_offsets64: zero-copy needs device buffers");
    require_device();
    hipStream_t s = S(stream);
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    int64_t ends[2];
    if (on_device) {
      CS_HIP(hipMemcpyAsync(&ends[0], offsets, 8, hipMemcpyDeviceToHost, s));
      CS_HIP(hipMemcpyAsync(&ends[1], offsets + rows, 8, hipMemcpyDeviceToHost, s));
      CS_HIP(hipStreamSynchronize(s));
    } else {
      ends[0] = offsets[0];
      ends[1] = offsets[rows];
    }
    if (ends[1] < ends[0] || ends[0] < 0) fail(CS_ERR_INVALID_ARG, "from_offsets64: offsets are not ascending");
    if (!copy) {
      // borrowed buffers must already be canonical (offsets[0]==0; null rows empty)
      if (ends[0] != 0) fail(CS_ERR_INVALID_ARG, "from_offsets64: zero-copy needs offsets[0]==0");
      auto* c = new cs_column;
      c->rows = rows;
      c->nbytes = ends[1];
      c->chars = dev_wrap(chars, (size_t)ends[1]);
      c->offsets = dev_wrap(offsets, sizeof(int64_t) * (rows + 1));
      if (validity) {
        // the engine writes/reads validity in 8-byte words: copy the small mask
        c->validity = dev_alloc(validity_bytes(rows), s);
        CS_HIP(hipMemsetAsync(c->validity->p, 0, validity_bytes(rows), s));
        copy_in(c->validity->p, validity, (size_t)((rows + 7) / 8), 1, s);
      } else {
        c->null_count = 0;
      }
      *out = c;
      return;
    }
    Buf o64 = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    copy_in(o64->p, offsets, sizeof(int64_t) * (rows + 1), on_device, s);
    Buf ch = dev_alloc((size_t)ends[1], s);
    copy_in(ch->p, chars, (size_t)ends[1], on_device, s);
    Buf v;
    if (validity) {
      v = dev_alloc(validity_bytes(rows), s);
      CS_HIP(hipMemsetAsync(v->p, 0, validity_bytes(rows), s));
      copy_in(v->p, validity, (size_t)((rows + 7) / 8), on_device, s);
    }
    cs_column* c = canonicalise(ch, o64, v, rows, ends[1], validity != nullptr, s);
    if (!on_device) CS_HIP(hipStreamSynchronize(s));
    *out = c;
  });
}

int cs_column_concat(const cs_column* const* cols, int n, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!out || n < 0 || (n > 0 && !cols)) fail(CS_ERR_INVALID_ARG, "create_from_strings: bad arguments");
    require_device();
    std::vector<const cs_column*> v;
    for (int i = 0; i < n; ++i) {
      if (!cols[i]) fail(CS_ERR_INVALID_ARG, "create_from_strings: null column");
      v.push_back(cols[i]);
    }
    *out = concat_columns(v, S(stream));
  });
}

int cs_column_slice(const cs_column* col, int64_t first, int64_t rows, cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!col || !out || first < 0 || rows < 0 || first + rows > col->rows)
      fail(CS_ERR_INVALID_ARG, "sublist: row range out of bounds");
    require_device();
    hipStream_t s = S(stream);
    if (rows == 0) {
      *out = make_all_null(0, s);
      return;
    }
    auto c = std::make_unique<cs_column>();
    c->rows = rows;
    c->offsets = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    if (col->validity) c->validity = dev_alloc(validity_bytes(rows), s);
    else c->null_count = 0;
    hipLaunchKernelGGL(k_slice, dim3(blocks_for(rows + 1)), dim3(kBlock), 0, s, view_of(col), first, rows,
                       ptr<int64_t>(c->offsets), ptr<uint8_t>(c->validity));
    int64_t* host = (int64_t*)pinned_scratch(16);
    CS_HIP(hipMemcpyAsync(host, col->d_offsets() + first, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(host + 1, col->d_offsets() + first + rows, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    c->nbytes = host[1] - host[0];
    c->chars = dev_alloc((size_t)c->nbytes, s);
    if (c->nbytes)
      CS_HIP(hipMemcpyAsync(c->chars->p, col->d_chars() + host[0], (size_t)c->nbytes, hipMemcpyDeviceToDevice, s));
    *out = c.release();
  });
}

// ---- HIP IPC (NVStrings::create_ipc_transfer / create_from_ipc; ipc_transfer.h:31-107) ----
namespace {
void ipc_handle_of(const Buf& b, unsigned char* out) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "cs_ipc_column carries 64-byte handles");
  hipIpcMemHandle_t h;
  CS_HIP(hipIpcGetMemHandle(&h, b->p));
  memcpy(out, &h, sizeof(h));
}
Buf ipc_open(const unsigned char* handle, size_t bytes) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  CS_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  auto b = std::make_shared<DevBuf>();
  b->p = p;
  b->bytes = bytes;
  b->capacity = 0;
  b->ipc_mapped = true;
  return b;
}
void export_column(const cs_column* col, cs_ipc_column* out) {
  memset(out, 0, sizeof(*out));
  for (const Buf& b : {col->chars, col->offsets, col->offsets32, col->validity})
    if (b && b->p && b->capacity == 0) fail(CS_ERR_INVALID_ARG, "ipc export: the column wraps memory it does not own (zero-copy ingest): copy it first");
  out->rows = col->rows;
  out->nbytes = col->nbytes;
  out->null_count = col->null_count;
  out->device = bound_device();
  if (col->chars && col->chars->p) ipc_handle_of(col->chars, out->chars);
  else out->nbytes = 0;
  if (col->offsets32) {
    out->offset_width = 4;
    ipc_handle_of(col->offsets32, out->offsets);
  } else {
    out->offset_width = 8;
    (void)col->d_offsets();
    ipc_handle_of(col->offsets, out->offsets);
  }
  if (col->validity && col->validity->p) {
    out->has_validity = 1;
    ipc_handle_of(col->validity, out->validity);
  }
}
cs_column* import_column(const cs_ipc_column* ipc) {
  if (ipc->rows < 0 || ipc->nbytes < 0 || (ipc->offset_width != 4 && ipc->offset_width != 8)) fail(CS_ERR_INVALID_ARG, "ipc import: malformed transfer record");
  auto c = std::make_unique<cs_column>();
  c->rows = ipc->rows;
  c->nbytes = ipc->nbytes;
  c->null_count = ipc->null_count;
  if (ipc->nbytes > 0) c->chars = ipc_open(ipc->chars, (size_t)ipc->nbytes);
  if (ipc->offset_width == 4) c->offsets32 = ipc_open(ipc->offsets, sizeof(int32_t) * (size_t)(ipc->rows + 1));
  else c->offsets = ipc_open(ipc->offsets, sizeof(int64_t) * (size_t)(ipc->rows + 1));
  if (ipc->has_validity) c->validity = ipc_open(ipc->validity, validity_bytes(ipc->rows));
  return c.release();
}
}  // namespace

int cs_column_ipc_export(const cs_column* col, cs_ipc_column* out) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "ipc export: null argument");
    require_device();
    CS_HIP(hipDeviceSynchronize());  // the importer must find finished buffers
    export_column(col, out);
  });
}
int cs_column_ipc_import(const cs_ipc_column* ipc, cs_column** out) {
  return guard([&] {
    if (!ipc || !out) fail(CS_ERR_INVALID_ARG, "ipc import: null argument");
    require_device();
    *out = import_column(ipc);
  });
}
int cs_category_ipc_export(const cs_category* cat, cs_ipc_category* out) {
  return guard([&] {
    if (!cat || !out) fail(CS_ERR_INVALID_ARG, "ipc export: null argument");
    require_device();
    CS_HIP(hipDeviceSynchronize());
    memset(out, 0, sizeof(*out));
    export_column(cat->keys.get(), &out->keys);
    out->rows = cat->rows;
    if (cat->rows > 0) {
      if (cat->values->capacity == 0) fail(CS_ERR_INVALID_ARG, "ipc export: the category's values are not owned memory");
      ipc_handle_of(cat->values, out->values);
    }
  });
}
int cs_category_ipc_import(const cs_ipc_category* ipc, cs_category** out) {
  return guard([&] {
    if (!ipc || !out) fail(CS_ERR_INVALID_ARG, "ipc import: null argument");
    require_device();
    auto c = std::make_unique<cs_category>();
    c->keys.reset(import_column(&ipc->keys));
    c->rows = ipc->rows;
    if (ipc->rows > 0) c->values = ipc_open(ipc->values, sizeof(int32_t) * (size_t)ipc->rows);
    *out = c.release();
  });
}

int cs_column_destroy(cs_column* col) {
  return guard([&] { delete col; });
}
int64_t cs_column_rows(const cs_column* col) { return col ? col->rows : 0; }
int64_t cs_column_nbytes(const cs_column* col) { return col ? col->nbytes : 0; }
int cs_column_offset_width(const cs_column* col) { return !col ? 0 : (col->offsets32 ? 4 : 8); }
int64_t cs_column_null_count(const cs_column* col) {
  int64_t n = -1;
  int st = guard([&] {
    if (!col) fail(CS_ERR_INVALID_ARG, "null column");
    n = count_nulls(col, nullptr);
  });
  return st == CS_OK ? n : -1;
}
int cs_column_cached_meta(const cs_column* col, int64_t out[4]) {
  return guard([&] {
    if (!col || !out) fail(CS_ERR_INVALID_ARG, "null argument");
    out[0] = col->max_span64;
    out[1] = col->max_row;
    out[2] = col->plain_bytes;
    out[3] = col->high_sample;
  });
}
int cs_column_get_view(const cs_column* col, cs_column_view* view) {
  return guard([&] {
    if (!col || !view) fail(CS_ERR_INVALID_ARG, "null argument");
    view->chars = col->d_chars();
    view->offsets = col->d_offsets();
    view->validity = col->d_validity();
    view->rows = col->rows;
    view->nbytes = col->nbytes;
    view->null_count = col->null_count;
  });
}

int cs_column_export_offsets64(const cs_column* col, uint8_t* chars, int64_t* offsets, uint8_t* validity,
                               int on_device, cs_stream stream) {
  return guard([&] {
    if (!col) fail(CS_ERR_INVALID_ARG, "null column");
    hipStream_t s = S(stream);
    if (col->rows == 0) return;
    if (chars) copy_out(chars, col->d_chars(), (size_t)col->nbytes, on_device, s);
    if (offsets) copy_out(offsets, col->d_offsets(), sizeof(int64_t) * (col->rows + 1), on_device, s);
    if (validity) {
      int64_t nulls = 0;
      // materialise (rows+7)/8 bytes even when the column has no mask
      Buf bits = dev_alloc((size_t)((col->rows + 7) / 8) + 8, s);
      Buf cnt = dev_alloc(8, s);
      CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
      hipLaunchKernelGGL(k_null_bitarray, dim3(blocks_for((col->rows + 7) / 8)), dim3(kBlock), 0, s,
                         view_of(col), 0, ptr<uint8_t>(bits), ptr<unsigned long long>(cnt));
      copy_out(validity, bits->p, (size_t)((col->rows + 7) / 8), on_device, s);
      CS_HIP(hipStreamSynchronize(s));
      (void)nulls;
    }
    if (!on_device) CS_HIP(hipStreamSynchronize(s));
  });
}

int cs_column_export_offsets32(const cs_column* col, char* chars, int32_t* offsets, uint8_t* validity,
                               int on_device, cs_stream stream) {
  return guard([&] {
    if (!col) fail(CS_ERR_INVALID_ARG, "null column");
    if (col->rows == 0) return;
    if (!chars || !offsets) return;  // the reference returns 0 without doing anything (NVStrings.cu:406-407)
    if (col->nbytes >= (1LL << 31)) fail(CS_ERR_RANGE, "create_offsets: column holds >= 2 GiB of chars; int32 offsets cannot address it");
    hipStream_t s = S(stream);
    if (col->offsets32) {  // born with int32 offsets: they go out as they are
      copy_out(offsets, col->d_offsets32(), sizeof(int32_t) * (col->rows + 1), on_device, s);
    } else {
      Buf o32 = dev_alloc(sizeof(int32_t) * (col->rows + 1), s);
      hipLaunchKernelGGL(k_narrow_offsets, dim3(blocks_for(col->rows + 1)), dim3(kBlock), 0, s,
                         col->d_offsets(), col->rows + 1, ptr<int32_t>(o32));
      copy_out(offsets, o32->p, sizeof(int32_t) * (col->rows + 1), on_device, s);
    }
    CS_HIP(hipStreamSynchronize(s));
    int st = cs_column_export_offsets64(col, (uint8_t*)chars, nullptr, validity, on_device, stream);
    if (st != CS_OK) fail(st, g_last_error);
  });
}

int cs_column_byte_count(const cs_column* col, int32_t* lengths, int on_device, cs_stream stream,
                         int64_t* total) {
  return guard([&] {
    if (!col) fail(CS_ERR_INVALID_ARG, "null column");
    hipStream_t s = S(stream);
    if (total) *total = 0;
    if (col->rows == 0) return;
    Buf tmp;
    int32_t* d_out = lengths;
    if (!on_device || !lengths) {
      tmp = dev_alloc(sizeof(int32_t) * col->rows, s);
      d_out = ptr<int32_t>(tmp);
    }
    // (null rows have zero extent, so the total is the column's byte count: no reduction)
    if (lengths) {
      hipLaunchKernelGGL(k_lengths_from_offsets, dim3(blocks_for(col->rows)), dim3(kBlock), 0, s, view_of(col), d_out);
      if (!on_device) copy_out(lengths, d_out, sizeof(int32_t) * col->rows, 0, s);
      CS_HIP(hipStreamSynchronize(s));
    }
    if (total) *total = col->nbytes;
  });
}

int cs_column_null_bitarray(const cs_column* col, uint8_t* bitarray, int empty_is_null, int on_device,
                            cs_stream stream, int64_t* null_count) {
  return guard([&] {
    if (!col || !bitarray) fail(CS_ERR_INVALID_ARG, "null argument");
    hipStream_t s = S(stream);
    if (null_count) *null_count = 0;
    if (col->rows == 0) return;
    size_t nb = (size_t)((col->rows + 7) / 8);
    Buf tmp;
    uint8_t* d_bits = bitarray;
    if (!on_device) {
      tmp = dev_alloc(nb, s);
      d_bits = ptr<uint8_t>(tmp);
    }
    Buf cnt = dev_alloc(8, s);
    CS_HIP(hipMemsetAsync(cnt->p, 0, 8, s));
    hipLaunchKernelGGL(k_null_bitarray, dim3(blocks_for((int64_t)nb)), dim3(kBlock), 0, s, view_of(col),
                       empty_is_null, d_bits, ptr<unsigned long long>(cnt));
    if (!on_device) copy_out(bitarray, d_bits, nb, 0, s);
    int64_t* host = (int64_t*)pinned_scratch(8);
    CS_HIP(hipMemcpyAsync(host, cnt->p, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (null_count) *null_count = host[0];
  });
}

int cs_column_digest(const cs_column* col, cs_stream stream, uint64_t* digest) {
  return guard([&] {
    if (!col || !digest) fail(CS_ERR_INVALID_ARG, "null argument");
    hipStream_t s = S(stream);
    *digest = 0;
    if (col->rows == 0) return;
    Buf acc = dev_alloc(8, s);
    CS_HIP(hipMemsetAsync(acc->p, 0, 8, s));
    hipLaunchKernelGGL(k_digest, dim3(blocks_for(col->rows)), dim3(kBlock), 0, s, view_of(col),
                       ptr<unsigned long long>(acc));
    uint64_t* host = (uint64_t*)pinned_scratch(8);
    CS_HIP(hipMemcpyAsync(host, acc->p, 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    *digest = host[0];
  });
}

int cs_prof_reset(void) {
  return guard([&] {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_prof) prof_collect(kv.second);
    g_prof.clear();
  });
}
int cs_prof_enable(int on) {
  g_prof_on = on != 0;
  return CS_OK;
}
int cs_stream_forget(cs_stream stream) {
  return guard([&] {
    if (!stream) return;
    (void)hipStreamSynchronize(S(stream));  // (what it still runs on cached blocks is done before they can be handed out again)
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams.erase(S(stream));
    // cached blocks that name the stream as their last user are idle now; the handle may be destroyed by its owner, so
    // nobody may synchronise it again: dev_alloc treats a block whose stream is not in g_streams as idle (blocks that
    // are released LATER with this stream in their DevBuf are covered by the same rule).  Their release events stay:
    // waiting for an event recorded on a destroyed stream is defined (it has completed).
    for (auto& kv : g_cache)
      if (kv.second.second == S(stream)) kv.second.second = kNoStream;
  });
}
__global__ void __launch_bounds__(256) k_debug_spin(unsigned long long ticks) {
  extern __shared__ uint32_t spin_lds[];
  if (threadIdx.x == 0) spin_lds[0] = 1;  // (the LDS is what keeps other workgroups off the CU)
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int cs_debug_spin(int blocks, int lds_bytes, int milliseconds, cs_stream stream) {
  return guard([&] {
    if (blocks < 1 || lds_bytes < 4 || lds_bytes > 160 * 1024 || milliseconds < 0) fail(CS_ERR_INVALID_ARG, "debug_spin: bad arguments");
    require_device();
    if (lds_bytes > 48 * 1024)
      CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_debug_spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(k_debug_spin, dim3((unsigned)blocks), dim3(256), (size_t)lds_bytes, S(stream), (unsigned long long)milliseconds * 100000ull);  // (100 MHz)
    CS_HIP(hipGetLastError());
  });
}
int cs_prof_get(const char* kernel, double* total_ms, int64_t* launches) {
  return guard([&] {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_prof.find(kernel ? kernel : "");
    if (it == g_prof.end()) {
      if (total_ms) *total_ms = 0;
      if (launches) *launches = 0;
      return;
    }
    prof_collect(it->second);
    if (total_ms) *total_ms = it->second.ms;
    if (launches) *launches = it->second.launches;
  });
}

// Row-major ("record") view of a column-major result (cs_split / cs_rsplit / cs_extract / cs_findall).
int cs_records_from_columns(const cs_column* const* cols, int ncols, int ragged, int64_t* list_offsets, int on_device,
                            cs_stream stream, cs_column** out) {
  return guard([&] {
    if (!out || !list_offsets || ncols < 0 || (ncols > 0 && !cols)) fail(CS_ERR_INVALID_ARG, "records: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    const int64_t rows = ncols ? cols[0]->rows : 0;
    for (int k = 0; k < ncols; ++k)
      if (!cols[k] || cols[k]->rows != rows) fail(CS_ERR_INVALID_ARG, "records: columns of different row counts");
    auto batch = [&](int k0, RecordCols& rc) {
      const int nb = std::min(RecordCols::kMax, ncols - k0);
      for (int j = 0; j < nb; ++j) rc.col[j] = view_of(cols[k0 + j]);
      return nb;
    };
    Buf list = dev_alloc(sizeof(int64_t) * (rows + 1), s);
    int64_t total = 0;
    if (rows && ncols) {
      Buf counts = dev_alloc(sizeof(int32_t) * rows, s);
      for (int k0 = 0; k0 < ncols; k0 += RecordCols::kMax) {
        RecordCols rc{};
        const int nb = batch(k0, rc);
        hipLaunchKernelGGL(k_record_counts, dim3(blocks_for(rows)), dim3(kBlock), 0, s, rc, k0, nb, ncols, ragged, rows, ptr<int32_t>(counts));
      }
      total = offsets_from_lengths(ptr<int32_t>(counts), rows, ptr<int64_t>(list), s);
    } else {
      CS_HIP(hipMemsetAsync(list->p, 0, sizeof(int64_t) * (rows + 1), s));
    }
    if (total >= (int64_t)1 << 31) fail(CS_ERR_RANGE, "records: more than 2^31 strings");
    auto o = std::make_unique<cs_column>();
    o->rows = total;
    o->offsets = dev_alloc(sizeof(int64_t) * (total + 1), s);
    if (total) {
      Buf lens = dev_alloc(sizeof(int32_t) * total, s);
      for (int k0 = 0; k0 < ncols; k0 += RecordCols::kMax) {
        RecordCols rc{};
        const int nb = batch(k0, rc);
        hipLaunchKernelGGL(k_record_lengths, dim3(blocks_for(rows)), dim3(kBlock), 0, s, rc, k0, nb, rows, ptr<int64_t>(list), ptr<int32_t>(lens));
      }
      o->nbytes = offsets_from_lengths(ptr<int32_t>(lens), total, ptr<int64_t>(o->offsets), s);
      o->chars = dev_alloc((size_t)o->nbytes, s);
      o->validity = validity_from_lengths(ptr<int32_t>(lens), total, s);
      for (int k0 = 0; k0 < ncols; k0 += RecordCols::kMax) {
        RecordCols rc{};
        const int nb = batch(k0, rc);
        hipLaunchKernelGGL(k_record_copy, dim3(blocks_for(rows)), dim3(kBlock), 0, s, rc, k0, nb, rows, ptr<int64_t>(list), o->d_offsets(),
                           ptr<uint8_t>(o->chars));
      }
      CS_HIP(hipGetLastError());
    } else {
      CS_HIP(hipMemsetAsync(o->offsets->p, 0, sizeof(int64_t), s));
      o->chars = dev_alloc(0, s);
      o->nbytes = 0;
    }
    CS_HIP(hipMemcpyAsync(list_offsets, list->p, sizeof(int64_t) * (rows + 1), on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    *out = o.release();
  });
}

}  // extern "C"
