// The distributed NVCategory build behind the C ABI (BASELINE.json north_star: local build -> exchange of the ranks'
// sorted key sets over RCCL -> merge -> remap; the reference itself is single-GPU, the merge semantics are
// NVCategory::create_from_categories, NVCategory.cu:430-514).  One process per GPU; nothing else on the hot path
// exchanges data.  custrings_amd/dist.py does the same through torch.distributed; this is what a C++ consumer of
// libNVCategory.so calls with the ncclComm_t it already has.
//
// Two routes, chosen from the ranks' key counts (the first exchange):
//   * all-gather of the key sets, padded to the largest, every rank merges all of them (K far below N);
//   * from kPartitionMinKeys keys in all (K close to N: BASELINE.json C4 lists K = 100M) the merge is partitioned by key
//     RANGE -- splitters from a sample, an all-to-all of every key to its range's owner, an all-gather of the merged
//     ranges: every rank merges 1/G of the keys instead of all of them (merge_partitioned below).
//
// RCCL is looked up in the process at run time (dlsym on what is loaded, else dlopen of librccl.so): the communicator
// belongs to the CALLER's RCCL, and a process that never builds a distributed category needs no RCCL at all (the CPU
// test box loads this library without a GPU runtime behind it).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "cs_internal.h"

using namespace cs;

extern "C" int cs_remap_codes(const int32_t* codes, int64_t n, const int32_t* table, int32_t* out, cs_stream stream);

namespace {

// ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t, ncclComm_t, hipStream_t); ncclUint8 = 1
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_sendrecv_t)(void*, size_t, int, int, void*, hipStream_t);  // ncclSend (const void*) / ncclRecv: buff, count, type, peer, comm, stream
typedef int (*nccl_group_t)();
typedef const char* (*nccl_errstr_t)(int);
nccl_allgather_t g_allgather = nullptr;
nccl_sendrecv_t g_send = nullptr, g_recv = nullptr;
nccl_group_t g_group_start = nullptr, g_group_end = nullptr;
nccl_errstr_t g_errstr = nullptr;
void resolve_rccl() {
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = nullptr;
    auto find = [&](const char* name) -> void* {
      void* sym = dlsym(RTLD_DEFAULT, name);
      if (!sym && !lib) {
        for (const char* so : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
          lib = dlopen(so, RTLD_NOW | RTLD_GLOBAL);
          if (lib) break;
        }
      }
      if (!sym && lib) sym = dlsym(lib, name);
      return sym;
    };
    g_allgather = reinterpret_cast<nccl_allgather_t>(find("ncclAllGather"));
    g_send = reinterpret_cast<nccl_sendrecv_t>(find("ncclSend"));
    g_recv = reinterpret_cast<nccl_sendrecv_t>(find("ncclRecv"));
    g_group_start = reinterpret_cast<nccl_group_t>(find("ncclGroupStart"));
    g_group_end = reinterpret_cast<nccl_group_t>(find("ncclGroupEnd"));
    g_errstr = reinterpret_cast<nccl_errstr_t>(find("ncclGetErrorString"));
  });
}
void rccl_check(int rc, const char* what) {
  if (rc != 0) fail(CS_ERR_INTERNAL, std::string(what) + ": " + (g_errstr ? g_errstr(rc) : "error"));
}
int rccl_allgather(void* comm, const void* send, void* recv, size_t bytes, void* stream) {
  rccl_check(g_allgather(send, recv, bytes, /* ncclUint8 */ 1, comm, static_cast<hipStream_t>(stream)), "ncclAllGather");
  return 0;
}
// the all-to-all of variable pieces as one group of point-to-point operations (xGMI is point to point: every pair its own link)
int rccl_alltoallv(void* comm, const void* send, const size_t* send_bytes, const size_t* send_off, void* recv, const size_t* recv_bytes, const size_t* recv_off,
                   int nranks, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  rccl_check(g_group_start(), "ncclGroupStart");
  // (an error inside the group is remembered, the group is CLOSED whatever happened -- an open group would stay on the
  // caller's communicator -- and only then reported)
  int first = 0;
  const char* where = "";
  for (int d = 0; d < nranks && first == 0; ++d) {
    if (send_bytes[d]) {
      first = g_send(const_cast<uint8_t*>(static_cast<const uint8_t*>(send)) + send_off[d], send_bytes[d], 1, d, comm, s);
      where = "ncclSend";
    }
    if (first == 0 && recv_bytes[d]) {
      first = g_recv(static_cast<uint8_t*>(recv) + recv_off[d], recv_bytes[d], 1, d, comm, s);
      where = "ncclRecv";
    }
  }
  const int end = g_group_end();
  rccl_check(first, where);
  rccl_check(end, "ncclGroupEnd");
  return 0;
}

struct KeyHeader {  // what every rank tells the others about a column it contributes
  int64_t keys, bytes, null_first;
  int64_t status;  // 0, or the status of a failure on that rank: every rank still takes part in this exchange and all stop together
};

struct Exchange {
  cs_allgather_fn allgather;
  cs_alltoallv_fn alltoallv;  // may be null: the all-gather route only
  void* ctx;
  int nranks, rank;
  hipStream_t s;
};

bool null_first_of(const cs_column* c, hipStream_t s) {  // (the null key, if any, is key 0)
  if (!c || c->rows == 0 || !c->validity) return false;
  uint8_t first = 0xFF;
  CS_HIP(hipMemcpyAsync(&first, c->validity->p, 1, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  return !(first & 1);
}
// one fixed-size record per rank; a rank that failed locally says so and every rank fails with it
std::vector<KeyHeader> gather_headers(const Exchange& x, const cs_column* mine, int status, const char* what) {
  hipStream_t s = x.s;
  const int n = x.nranks;
  KeyHeader me{mine ? mine->rows : 0, mine ? mine->nbytes : 0, status == 0 && null_first_of(mine, s) ? 1 : 0, status};
  Buf hdr = dev_alloc(sizeof(KeyHeader) * (n + 1), s);
  KeyHeader* d_hdr = ptr<KeyHeader>(hdr);
  CS_HIP(hipMemcpyAsync(d_hdr + n, &me, sizeof(me), hipMemcpyHostToDevice, s));
  CS_HIP(hipStreamSynchronize(s));  // (`me` is pageable)
  if (x.allgather(x.ctx, d_hdr + n, d_hdr, sizeof(KeyHeader), s) != 0)
    fail(CS_ERR_INTERNAL, std::string("category_build_distributed: the exchange of the sizes failed (") + what + ")");
  std::vector<KeyHeader> all(n);
  CS_HIP(hipMemcpyAsync(all.data(), d_hdr, sizeof(KeyHeader) * n, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  for (int r = 0; r < n; ++r)
    if (all[r].status != 0)
      fail(all[r].status == CS_ERR_ALLOC ? CS_ERR_ALLOC : CS_ERR_INTERNAL, "category_build_distributed: rank " + std::to_string(r) + " failed before the exchange (status " +
                                                                               std::to_string(all[r].status) + ", " + what + "); every rank stops");
  if (all[x.rank].keys != me.keys || all[x.rank].bytes != me.bytes) fail(CS_ERR_INTERNAL, "category_build_distributed: the exchange returned another rank's record at this rank's place");
  return all;
}
// An agreement point of the partitioned merge: every rank contributes its status word, and when any rank has failed
// since the last exchange ALL ranks stop here together -- the one that failed with its own error, the others naming it --
// instead of one rank throwing while its peers wait in the next collective for ever.
void agree(const Exchange& x, int status, const std::string& why, const char* stage) {
  hipStream_t s = x.s;
  const int n = x.nranks;
  int64_t me = status;
  Buf w = dev_alloc(sizeof(int64_t) * (n + 1), s);
  int64_t* d = ptr<int64_t>(w);
  CS_HIP(hipMemcpyAsync(d + n, &me, sizeof(me), hipMemcpyHostToDevice, s));
  CS_HIP(hipStreamSynchronize(s));
  if (x.allgather(x.ctx, d + n, d, sizeof(int64_t), s) != 0)
    fail(CS_ERR_INTERNAL, std::string("category_build_distributed: the status exchange failed (") + stage + ")");
  std::vector<int64_t> all(n);
  CS_HIP(hipMemcpyAsync(all.data(), d, sizeof(int64_t) * n, hipMemcpyDeviceToHost, s));
  CS_HIP(hipStreamSynchronize(s));
  if (status) fail(status, why + " (" + stage + "; every rank stops)");
  for (int r = 0; r < n; ++r)
    if (all[r] != 0)
      fail(all[r] == CS_ERR_ALLOC ? CS_ERR_ALLOC : CS_ERR_INTERNAL, "category_build_distributed: rank " + std::to_string(r) + " failed (status " + std::to_string(all[r]) +
                                                                        ", " + stage + "); every rank stops");
}

// Every rank's column on every rank: offsets and chars padded to the largest (an all-gather moves equal pieces); the
// columns are views into the two gathered buffers.
struct Gathered {
  std::vector<KeyHeader> hdr;
  std::vector<std::unique_ptr<cs_column>> cols;
  Buf offs, chrs;
  int64_t total_keys = 0;
  std::vector<const cs_column*> ptrs() const {
    std::vector<const cs_column*> v;
    for (const auto& c : cols) v.push_back(c.get());
    return v;
  }
};
Gathered gather_columns(const Exchange& x, const cs_column* mine, std::vector<KeyHeader> hdr, const char* what) {
  hipStream_t s = x.s;
  const int n = x.nranks;
  Gathered g;
  g.hdr = std::move(hdr);
  int64_t max_keys = 0, max_bytes = 0;
  for (const KeyHeader& h : g.hdr) {
    max_keys = std::max(max_keys, h.keys);
    max_bytes = std::max(max_bytes, h.bytes);
    g.total_keys += h.keys;
  }
  const KeyHeader& me = g.hdr[x.rank];
  const size_t off_piece = sizeof(int64_t) * (size_t)(max_keys + 1), chr_piece = (size_t)((max_bytes + 15) & ~(int64_t)15);
  {  // (the gathered buffers hold every rank's piece: a rank that cannot have them says so before anybody enters the all-gather)
    int st = 0;
    std::string why;
    try {
      g.offs = dev_alloc(off_piece * (n + 1), s);
      g.chrs = dev_alloc(chr_piece * (n + 1) + 16, s);
    } catch (const Error& e) {
      st = e.code ? e.code : CS_ERR_INTERNAL;
      why = e.msg;
    }
    agree(x, st, why, what);
  }
  uint8_t* my_off = ptr<uint8_t>(g.offs) + off_piece * n;
  uint8_t* my_chr = ptr<uint8_t>(g.chrs) + chr_piece * n;
  CS_HIP(hipMemsetAsync(my_off, 0, off_piece, s));
  if (me.keys > 0) CS_HIP(hipMemcpyAsync(my_off, mine->d_offsets(), sizeof(int64_t) * (size_t)(me.keys + 1), hipMemcpyDeviceToDevice, s));
  if (me.bytes > 0) CS_HIP(hipMemcpyAsync(my_chr, mine->d_chars(), (size_t)me.bytes, hipMemcpyDeviceToDevice, s));
  if (x.allgather(x.ctx, my_off, g.offs->p, off_piece, s) != 0) fail(CS_ERR_INTERNAL, std::string("category_build_distributed: the exchange of the offsets failed (") + what + ")");
  if (chr_piece && x.allgather(x.ctx, my_chr, g.chrs->p, chr_piece, s) != 0)
    fail(CS_ERR_INTERNAL, std::string("category_build_distributed: the exchange of the bytes failed (") + what + ")");
  for (int r = 0; r < n; ++r) {
    auto c = std::make_unique<cs_column>();
    c->rows = g.hdr[r].keys;
    c->nbytes = g.hdr[r].bytes;
    c->chars = dev_wrap(ptr<uint8_t>(g.chrs) + chr_piece * r, (size_t)g.hdr[r].bytes);
    c->offsets = dev_wrap(ptr<uint8_t>(g.offs) + off_piece * r, sizeof(int64_t) * (size_t)(g.hdr[r].keys + 1));
    if (g.hdr[r].null_first && g.hdr[r].keys > 0) {
      c->validity = dev_alloc(validity_bytes(g.hdr[r].keys), s);
      CS_HIP(hipMemsetAsync(c->validity->p, 0xFF, validity_bytes(g.hdr[r].keys), s));
      CS_HIP(hipMemsetAsync(c->validity->p, 0xFE, 1, s));
      c->null_count = 1;
    } else {
      c->null_count = 0;
    }
    g.cols.push_back(std::move(c));
  }
  return g;
}

// ---- the merge partitioned by key RANGE --------------------------------------------------------------------------------
// Splitters from a strided sample of every rank's sorted keys cut the key space into one range per rank; every local key
// goes to its range's owner (the local keys are sorted: contiguous slices; one all-to-all for the lengths, one for the
// bytes); the owner merges what it receives -- 1/G of the keys -- and returns every sender the positions of its keys in
// the merged range; an all-gather of the merged ranges, in rank order, is the global sorted key set (what
// NVCategory::create_from_categories makes of the gathered key sets, NVCategory.cu:430-514), and a local key's global
// code is its range's base plus the position that came back.
constexpr int64_t kPartitionMinKeys = (int64_t)1 << 21;
constexpr int kSplitterSamplesPerRank = 64;  // per destination rank

// cut[j + 1] = index of the first local key (from `first` on: a null key stays in range 0) that is not below splitter j
// (custring.inl:240-261: unsigned bytewise, shorter is less)
__global__ void k_range_cuts(ColView keys, int64_t first, ColView split, int nsplit, int64_t* __restrict__ cut) {
  const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (j >= nsplit) return;
  const uint8_t* sp = split.chars + split.offsets[j];
  const int64_t sn = split.offsets[j + 1] - split.offsets[j];
  int64_t lo = first, hi = keys.rows;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    const uint8_t* kp = keys.chars + keys.offsets[mid];
    const int64_t kn = keys.offsets[mid + 1] - keys.offsets[mid];
    const int64_t m = kn < sn ? kn : sn;
    int cmp = 0;
    for (int64_t i = 0; i < m && cmp == 0; ++i) cmp = (int)kp[i] - (int)sp[i];
    if (cmp == 0) cmp = kn < sn ? -1 : (kn > sn ? 1 : 0);
    if (cmp < 0) lo = mid + 1;
    else hi = mid;
  }
  cut[j + 1] = lo;
}
__global__ void k_key_lens(const int64_t* __restrict__ off, int64_t n, int32_t* __restrict__ lens) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) lens[i] = (int32_t)(off[i + 1] - off[i]);
}
// table[i] = base of key i's range + its position in the merged range (`cut`: nranks + 1 entries, `base`: nranks)
__global__ void k_range_table(const int32_t* __restrict__ back, const int64_t* __restrict__ cut, const int64_t* __restrict__ base, int nranks, int64_t n,
                              int32_t* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int d = 0;
  while (d + 1 < nranks && i >= cut[d + 1]) ++d;
  table[i] = (int32_t)(base[d] + back[i]);
}
Buf upload_i64(const std::vector<int64_t>& v, hipStream_t s) {
  Buf b = dev_alloc(sizeof(int64_t) * std::max<size_t>(v.size(), 1), s);
  if (!v.empty()) CS_HIP(hipMemcpyAsync(b->p, v.data(), sizeof(int64_t) * v.size(), hipMemcpyHostToDevice, s));
  CS_HIP(hipStreamSynchronize(s));  // (pageable source)
  return b;
}
std::unique_ptr<cs_column> rows_at(const cs_column* col, const std::vector<int32_t>& pos, hipStream_t s) {
  Buf d = dev_alloc(sizeof(int32_t) * std::max<size_t>(pos.size(), 1), s);
  if (!pos.empty()) CS_HIP(hipMemcpyAsync(d->p, pos.data(), sizeof(int32_t) * pos.size(), hipMemcpyHostToDevice, s));
  CS_HIP(hipStreamSynchronize(s));
  return std::unique_ptr<cs_column>(gather_rows(col, ptr<const int32_t>(d), (int64_t)pos.size(), s));
}

// Failure agreement: between the collectives below sit multi-GB allocations and consistency checks that may
// throw on ONE rank (the owner of a skewed range running out of memory, say).  Every stretch of local work runs under
// `attempt`, which turns a throw into a status; the next exchange -- a header gather, the size exchange (one more slot per
// row) or an `agree` in front of an all-to-all whose sizes are already fixed -- carries the status, and all ranks stop
// there together.  CS_DIST_TEST_FAIL_AT=<rank>:<stage> simulates a failure in stage 1..4 (tests).
cs_category* merge_partitioned(const Exchange& x, const cs_category* local, const std::vector<KeyHeader>& heads) {
  hipStream_t s = x.s;
  const int n = x.nranks;
  const cs_column* keys = local->keys.get();
  const int64_t K = keys->rows, first = heads[x.rank].null_first ? 1 : 0;
  int status = 0;
  std::string why;
  int fail_stage = 0;
  if (const char* f = cs::cfg("CS_DIST_TEST_FAIL_AT")) {
    int r = -1, st = 0;
    if (sscanf(f, "%d:%d", &r, &st) == 2 && r == x.rank) fail_stage = st;
  }
  int stage = 0;
  auto attempt = [&](auto&& work) {
    ++stage;
    if (status) return;
    try {
      if (stage == fail_stage) fail(CS_ERR_ALLOC, "simulated failure in stage " + std::to_string(stage) + " of the partitioned merge (CS_DIST_TEST_FAIL_AT)");
      work();
    } catch (const Error& e) {
      status = e.code ? e.code : CS_ERR_INTERNAL;
      why = e.msg;
    } catch (const std::exception& e) {
      status = CS_ERR_INTERNAL;
      why = e.what();
    }
  };
  // ---- stage 1: splitters -- a strided sample of every rank's keys, pooled, sorted, cut into n ranges
  std::unique_ptr<cs_column> sample;
  attempt([&] {
    std::vector<int32_t> pos;
    const int64_t want = (int64_t)kSplitterSamplesPerRank * n, step = std::max<int64_t>((K - first) / want, 1);
    for (int64_t i = first; i < K; i += step) pos.push_back((int32_t)i);
    sample = rows_at(keys, pos, s);
  });
  Gathered gs;
  {
    std::vector<KeyHeader> sh;
    try {
      sh = gather_headers(x, status ? nullptr : sample.get(), status, "splitter samples");
    } catch (const Error& e) {
      if (status) fail(status, why + " (" + e.msg + ")");
      throw;
    }
    gs = gather_columns(x, sample.get(), std::move(sh), "splitter samples");
  }
  // ---- stage 2: this rank's keys by range: keys cut[d] .. cut[d + 1] go to rank d
  std::vector<int64_t> cut(n + 1, K), byte_at(n + 1, 0);
  cut[0] = 0;
  Buf d_cut;
  attempt([&] {
    std::unique_ptr<cs_column> splitters;
    int nsplit = 0;
    if (gs.total_keys > 0) {
      std::unique_ptr<cs_column> pooled(concat_columns(gs.ptrs(), s));
      std::unique_ptr<cs_category> pool(category_build(pooled.get(), s));  // sorted, unique: the same on every rank
      const int64_t M = pool->keys->rows, stride = std::max<int64_t>(M / n, 1);
      std::vector<int32_t> sp;
      for (int64_t i = stride; i < M && (int)sp.size() < n - 1; i += stride) sp.push_back((int32_t)i);
      nsplit = (int)sp.size();
      if (nsplit) splitters = rows_at(pool->keys.get(), sp, s);
    }
    d_cut = upload_i64(cut, s);
    if (nsplit && K > 0) {
      hipLaunchKernelGGL(k_range_cuts, dim3((unsigned)((nsplit + 255) / 256)), dim3(256), 0, s, view_of(keys), first, view_of(splitters.get()), nsplit, ptr<int64_t>(d_cut));
      CS_HIP(hipGetLastError());
      CS_HIP(hipMemcpyAsync(cut.data(), d_cut->p, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, s));
      CS_HIP(hipStreamSynchronize(s));
      for (int d = nsplit + 1; d <= n; ++d) cut[d] = K;
      for (int d = 1; d <= n; ++d) cut[d] = std::max(cut[d], cut[d - 1]);
      CS_HIP(hipMemcpyAsync(d_cut->p, cut.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, s));
      CS_HIP(hipStreamSynchronize(s));
    }
    if (K > 0) {
      for (int d = 0; d <= n; ++d) CS_HIP(hipMemcpyAsync(&byte_at[d], keys->d_offsets() + cut[d], sizeof(int64_t), hipMemcpyDeviceToHost, s));
      CS_HIP(hipStreamSynchronize(s));
    }
  });
  // ---- who sends how much to whom: every rank's (keys, bytes) per destination and its status, all-gathered
  const size_t slots = 2 * (size_t)n + 1, row = sizeof(int64_t) * slots;
  std::vector<int64_t> mine(slots, 0);
  for (int d = 0; d < n && !status; ++d) {
    mine[2 * d] = cut[d + 1] - cut[d];
    mine[2 * d + 1] = byte_at[d + 1] - byte_at[d];
  }
  mine[2 * n] = status;
  std::vector<int64_t> counts(slots * n);
  {
    Buf d_counts = dev_alloc(row * (n + 1), s);  // (small: a failure here is the exchange's own)
    CS_HIP(hipMemcpyAsync(ptr<uint8_t>(d_counts) + row * n, mine.data(), row, hipMemcpyHostToDevice, s));
    CS_HIP(hipStreamSynchronize(s));
    if (x.allgather(x.ctx, ptr<uint8_t>(d_counts) + row * n, d_counts->p, row, s) != 0) fail(CS_ERR_INTERNAL, "category_build_distributed: the exchange of the range sizes failed");
    CS_HIP(hipMemcpyAsync(counts.data(), d_counts->p, row * n, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
  }
  if (status) fail(status, why + " (before the exchange of the range sizes; every rank stops)");
  for (int r = 0; r < n; ++r)
    if (counts[slots * r + 2 * n] != 0)
      fail(counts[slots * r + 2 * n] == CS_ERR_ALLOC ? CS_ERR_ALLOC : CS_ERR_INTERNAL, "category_build_distributed: rank " + std::to_string(r) + " failed (status " +
                                                                                          std::to_string(counts[slots * r + 2 * n]) + ", range sizes); every rank stops");
  std::vector<int64_t> got_keys(n), got_bytes(n);
  int64_t in_keys = 0, in_bytes = 0;
  for (int r = 0; r < n; ++r) {
    got_keys[r] = counts[slots * r + 2 * x.rank];
    got_bytes[r] = counts[slots * r + 2 * x.rank + 1];
    in_keys += got_keys[r];
    in_bytes += got_bytes[r];
  }
  // ---- stage 3: room for what arrives (the owner of a skewed range may not have it: agreed on BEFORE anything is sent)
  Buf lens, in_lens, in_chars, back;
  attempt([&] {
    lens = dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(K, 1), s);
    if (K) hipLaunchKernelGGL(k_key_lens, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, keys->d_offsets(), K, ptr<int32_t>(lens));
    in_lens = dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(in_keys, 1), s);
    in_chars = dev_alloc((size_t)in_bytes + 16, s);
    back = dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(K, 1), s);
  });
  agree(x, status, why, "receive buffers of the key ranges");
  // ---- the keys travel: lengths, then bytes
  std::vector<size_t> sb(n), so(n), rb(n), lro(n), cro(n);
  size_t acc = 0;
  for (int d = 0; d < n; ++d) {
    sb[d] = sizeof(int32_t) * (size_t)(cut[d + 1] - cut[d]);
    so[d] = sizeof(int32_t) * (size_t)cut[d];
    rb[d] = sizeof(int32_t) * (size_t)got_keys[d];
    lro[d] = acc;
    acc += rb[d];
  }
  if (x.alltoallv(x.ctx, lens->p, sb.data(), so.data(), in_lens->p, rb.data(), lro.data(), n, s) != 0)
    fail(CS_ERR_INTERNAL, "category_build_distributed: the all-to-all of the key lengths failed");
  acc = 0;
  for (int d = 0; d < n; ++d) {
    sb[d] = (size_t)(byte_at[d + 1] - byte_at[d]);
    so[d] = (size_t)(byte_at[d] - byte_at[0]);
    rb[d] = (size_t)got_bytes[d];
    cro[d] = acc;
    acc += rb[d];
  }
  const uint8_t* my_chars = keys->nbytes > 0 ? keys->d_chars() + byte_at[0] : ptr<const uint8_t>(in_chars);  // (a valid address when nothing is sent)
  if (x.alltoallv(x.ctx, my_chars, sb.data(), so.data(), in_chars->p, rb.data(), cro.data(), n, s) != 0)
    fail(CS_ERR_INTERNAL, "category_build_distributed: the all-to-all of the key bytes failed");
  // ---- stage 4: this rank's range -- the parts that arrived (each sorted and unique), merged
  std::vector<std::unique_ptr<cs_column>> parts;
  std::vector<int64_t> part_at(n, 0);  // where sender r's keys stand in the concatenation
  std::unique_ptr<cs_category> range;
  std::unique_ptr<cs_column> range_keys;
  attempt([&] {
    std::vector<const cs_column*> part_ptrs;
    int64_t at = 0;
    for (int r = 0; r < n; ++r) {
      part_at[r] = at;
      if (got_keys[r] == 0) continue;
      auto c = std::make_unique<cs_column>();
      c->rows = got_keys[r];
      c->nbytes = got_bytes[r];
      c->offsets = dev_alloc(sizeof(int64_t) * (size_t)(got_keys[r] + 1), s);
      const int64_t total = offsets_from_lengths(ptr<const int32_t>(in_lens) + at, got_keys[r], ptr<int64_t>(c->offsets), s);
      if (total != got_bytes[r]) fail(CS_ERR_INTERNAL, "category_build_distributed: a received part's lengths do not add up to its bytes");
      c->chars = dev_wrap(ptr<uint8_t>(in_chars) + cro[r], (size_t)got_bytes[r]);
      // (a null key travels as a key of no bytes at the head of its sender's slice for range 0 -- the sizes exchange said
      // which ranks have one)
      if (x.rank == 0 && heads[r].null_first) {
        c->validity = dev_alloc(validity_bytes(got_keys[r]), s);
        CS_HIP(hipMemsetAsync(c->validity->p, 0xFF, validity_bytes(got_keys[r]), s));
        CS_HIP(hipMemsetAsync(c->validity->p, 0xFE, 1, s));
        c->null_count = 1;
      } else {
        c->null_count = 0;
      }
      at += got_keys[r];
      part_ptrs.push_back(c.get());
      parts.push_back(std::move(c));
    }
    if (!part_ptrs.empty()) {
      std::unique_ptr<cs_column> all(concat_columns(part_ptrs, s));
      range.reset(category_build(all.get(), s));
      range_keys = std::move(range->keys);
    } else {
      range_keys.reset(make_all_null(0, s));
    }
  });
  agree(x, status, why, "merge of the received key range");
  // ---- positions back to the senders (in the senders' key order: slice d lands at this rank's keys cut[d] ..)
  for (int d = 0; d < n; ++d) {
    sb[d] = sizeof(int32_t) * (size_t)got_keys[d];
    so[d] = sizeof(int32_t) * (size_t)part_at[d];
    rb[d] = sizeof(int32_t) * (size_t)(cut[d + 1] - cut[d]);
    lro[d] = sizeof(int32_t) * (size_t)cut[d];
  }
  const void* codes = range ? range->values->p : back->p;
  if (x.alltoallv(x.ctx, codes, sb.data(), so.data(), back->p, rb.data(), lro.data(), n, s) != 0)
    fail(CS_ERR_INTERNAL, "category_build_distributed: the all-to-all of the merged positions failed");
  // ---- the merged ranges to everybody: in rank order they are the global key set
  Gathered gr = gather_columns(x, range_keys.get(), gather_headers(x, range_keys.get(), 0, "merged ranges"), "merged ranges");
  std::vector<int64_t> base(n, 0);
  std::vector<const cs_column*> nonempty;
  int64_t run = 0;
  for (int r = 0; r < n; ++r) {
    base[r] = run;
    run += gr.hdr[r].keys;
    if (gr.hdr[r].keys > 0) nonempty.push_back(gr.cols[r].get());
  }
  // (the same sum on every rank: all of them stop here together)
  if (run >= ((int64_t)1 << 31)) fail(CS_ERR_RANGE, "category_build_distributed: more than 2^31 keys in all");
  auto merged = std::make_unique<cs_category>();
  merged->rows = local->rows;
  merged->keys.reset(nonempty.empty() ? make_all_null(0, s) : concat_columns(nonempty, s));
  merged->values = dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(local->rows, 1), s);
  if (K > 0 && local->rows > 0) {
    Buf table = dev_alloc(sizeof(int32_t) * (size_t)K, s), d_base = upload_i64(base, s);
    hipLaunchKernelGGL(k_range_table, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, ptr<const int32_t>(back), ptr<const int64_t>(d_cut), ptr<const int64_t>(d_base), n, K,
                       ptr<int32_t>(table));
    CS_HIP(hipGetLastError());
    const int rc = cs_remap_codes(ptr<const int32_t>(local->values), local->rows, ptr<const int32_t>(table), ptr<int32_t>(merged->values), x.s);
    if (rc != 0) fail(rc, cs_last_error());
  }
  CS_HIP(hipStreamSynchronize(s));  // (the gathered buffers leave scope; what follows the last collective fails locally, nobody waits)
  return merged.release();
}

cs_category* merge_gathered(const Exchange& x, const cs_category* local, std::vector<KeyHeader> heads) {
  hipStream_t s = x.s;
  const int n = x.nranks;
  Gathered g = gather_columns(x, local->keys.get(), std::move(heads), "key sets");
  // every rank merges the same key sets: the category of their concatenation has the merged keys, and its codes are the
  // ranks' old-code -> new-code tables, back to back
  std::unique_ptr<cs_column> all_keys(concat_columns(g.ptrs(), s));
  std::unique_ptr<cs_category> all(category_build(all_keys.get(), s));
  int64_t before = 0;
  for (int r = 0; r < x.rank; ++r) before += g.hdr[r].keys;
  auto merged = std::make_unique<cs_category>();
  merged->rows = local->rows;
  merged->values = dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(local->rows, 1), s);
  if (local->rows > 0) {
    const int rc = cs_remap_codes(ptr<const int32_t>(local->values), local->rows, ptr<const int32_t>(all->values) + before, ptr<int32_t>(merged->values), x.s);
    if (rc != 0) fail(rc, cs_last_error());
  }
  merged->keys = std::move(all->keys);
  CS_HIP(hipStreamSynchronize(s));  // (the gathered buffers leave scope)
  (void)n;
  return merged.release();
}

void build_distributed(const cs_column* col, const Exchange& x, bool run_alone, cs_category** out) {
  hipStream_t s = x.s;
  // the local build may fail (memory, a key too long): this rank then still takes part in the first exchange and tells
  // the others, so that nobody is left waiting in a collective
  std::unique_ptr<cs_category> local;
  int status = 0;
  std::string why;
  try {
    local.reset(category_build(col, s));
  } catch (const Error& e) {
    status = e.code ? e.code : CS_ERR_INTERNAL;
    why = e.msg;
  } catch (const std::exception& e) {
    status = CS_ERR_INTERNAL;
    why = e.what();
  }
  if (const char* f = cs::cfg("CS_DIST_TEST_FAIL"))  // tests: this rank's local build "fails"
    if (status == 0 && atoi(f) == x.rank) {
      status = CS_ERR_ALLOC;
      why = "simulated failure of the local build (CS_DIST_TEST_FAIL)";
      local.reset();
    }
  if (x.nranks == 1 && !run_alone) {  // the local codes are the global ones
    if (status) fail(status, why);
    *out = local.release();
    return;
  }
  std::vector<KeyHeader> heads;
  try {
    heads = gather_headers(x, status ? nullptr : local->keys.get(), status, "key sets");
  } catch (const Error& e) {
    if (status) fail(status, why + " (" + e.msg + ")");  // (this rank's own failure is the one to report here)
    throw;
  }
  int64_t total = 0;
  for (const KeyHeader& h : heads) total += h.keys;
  const char* force = cs::cfg("CS_DIST_PARTITIONED");  // tests: 1 = always (given an all-to-all), 0 = never
  const bool partitioned = x.alltoallv && x.nranks > 1 && (force ? atoi(force) != 0 : total >= kPartitionMinKeys);
  *out = partitioned ? merge_partitioned(x, local.get(), heads) : merge_gathered(x, local.get(), std::move(heads));
}

}  // namespace

extern "C" {

int cs_category_build_distributed_with(const cs_column* col, cs_allgather_fn allgather, void* ctx, int nranks, int rank, cs_stream stream,
                                       cs_category** out) {
  return cs_category_build_distributed_with2(col, allgather, nullptr, ctx, nranks, rank, stream, out);
}

int cs_category_build_distributed_with2(const cs_column* col, cs_allgather_fn allgather, cs_alltoallv_fn alltoallv, void* ctx, int nranks, int rank,
                                        cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!col || !out || !allgather || nranks < 1 || rank < 0 || rank >= nranks) fail(CS_ERR_INVALID_ARG, "category_build_distributed: bad arguments");
    require_device();
    const Exchange x{allgather, alltoallv, ctx, nranks, rank, S(stream)};
    build_distributed(col, x, ctx != nullptr, out);  // (with a context the exchange runs even for one rank: a rank with itself)
  });
}

int cs_category_build_distributed(const cs_column* col, void* nccl_comm, int nranks, int rank, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!nccl_comm && nranks > 1) fail(CS_ERR_INVALID_ARG, "category_build_distributed: no communicator");
    if (nranks > 1 || nccl_comm) {
      resolve_rccl();
      if (!g_allgather) fail(CS_ERR_INTERNAL, "category_build_distributed: no RCCL in this process (ncclAllGather not found, librccl.so not loadable)");
    }
    const bool p2p = g_send && g_recv && g_group_start && g_group_end;
    const int rc = cs_category_build_distributed_with2(col, &rccl_allgather, p2p ? &rccl_alltoallv : nullptr, nccl_comm, nranks, rank, stream, out);
    if (rc != 0) fail(rc, cs_last_error());
  });
}

}  // extern "C"
