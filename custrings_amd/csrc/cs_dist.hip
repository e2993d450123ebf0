// The distributed NVCategory build behind the C ABI (BASELINE.json north_star: local build -> all-gather of the ranks'
// sorted key sets over RCCL -> merge -> remap; the reference itself is single-GPU, the merge semantics are
// NVCategory::create_from_categories, NVCategory.cu:430-514).  One process per GPU; nothing else on the hot path
// exchanges data.  custrings_amd/dist.py does the same through torch.distributed; this is what a C++ consumer of
// libNVCategory.so calls with the ncclComm_t it already has.
//
// RCCL is looked up in the process at run time (dlsym on what is loaded, else dlopen of librccl.so): the communicator
// belongs to the CALLER's RCCL, and a process that never builds a distributed category needs no RCCL at all (the CPU
// test box loads this library without a GPU runtime behind it).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "cs_internal.h"

using namespace cs;

extern "C" int cs_category_merge_gathered(const cs_category* local, const cs_column* const* keysets, int nranks, int rank, cs_stream stream,
                                          cs_column** merged_keys, int32_t* values);

namespace {

// ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t, ncclComm_t, hipStream_t); ncclUint8 = 1
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_t)(int);
nccl_allgather_t g_allgather = nullptr;
nccl_errstr_t g_errstr = nullptr;
void resolve_rccl() {
  static std::once_flag once;
  std::call_once(once, [] {
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
    void* lib = nullptr;
    if (!sym) {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
      }
      if (lib) sym = dlsym(lib, "ncclAllGather");
    }
    g_allgather = reinterpret_cast<nccl_allgather_t>(sym);
    void* es = dlsym(RTLD_DEFAULT, "ncclGetErrorString");
    if (!es && lib) es = dlsym(lib, "ncclGetErrorString");
    g_errstr = reinterpret_cast<nccl_errstr_t>(es);
  });
}
int rccl_allgather(void* comm, const void* send, void* recv, size_t bytes, void* stream) {
  const int rc = g_allgather(send, recv, bytes, /* ncclUint8 */ 1, comm, static_cast<hipStream_t>(stream));
  if (rc != 0) fail(CS_ERR_INTERNAL, std::string("ncclAllGather: ") + (g_errstr ? g_errstr(rc) : "error"));
  return 0;
}

struct KeyHeader {  // what every rank tells the others about its key set
  int64_t keys, bytes, null_first;
};

}  // namespace

extern "C" {

int cs_category_build_distributed_with(const cs_column* col, cs_allgather_fn allgather, void* ctx, int nranks, int rank, cs_stream stream,
                                       cs_category** out) {
  return guard([&] {
    if (!col || !out || !allgather || nranks < 1 || rank < 0 || rank >= nranks) fail(CS_ERR_INVALID_ARG, "category_build_distributed: bad arguments");
    require_device();
    hipStream_t s = S(stream);
    std::unique_ptr<cs_category> local(category_build(col, s));
    if (nranks == 1 && !ctx) {  // the local codes are the global ones (with a context the exchange runs, a rank with itself)
      *out = local.release();
      return;
    }
    const cs_column* keys = local->keys.get();
    // ---- sizes: one fixed-size record per rank
    KeyHeader mine{keys->rows, keys->nbytes, 0};
    if (keys->rows > 0 && keys->validity) {  // (the null key, if any, is key 0)
      uint8_t first = 0xFF;
      CS_HIP(hipMemcpyAsync(&first, keys->validity->p, 1, hipMemcpyDeviceToHost, s));
      CS_HIP(hipStreamSynchronize(s));
      mine.null_first = (first & 1) ? 0 : 1;
    }
    Buf hdr = dev_alloc(sizeof(KeyHeader) * (nranks + 1), s);
    KeyHeader* d_hdr = ptr<KeyHeader>(hdr);
    CS_HIP(hipMemcpyAsync(d_hdr + nranks, &mine, sizeof(mine), hipMemcpyHostToDevice, s));
    CS_HIP(hipStreamSynchronize(s));  // (`mine` is pageable)
    if (allgather(ctx, d_hdr + nranks, d_hdr, sizeof(KeyHeader), s) != 0) fail(CS_ERR_INTERNAL, "category_build_distributed: the exchange of the key counts failed");
    std::vector<KeyHeader> all(nranks);
    CS_HIP(hipMemcpyAsync(all.data(), d_hdr, sizeof(KeyHeader) * nranks, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    if (all[rank].keys != mine.keys || all[rank].bytes != mine.bytes) fail(CS_ERR_INTERNAL, "category_build_distributed: the exchange returned another rank's record at this rank's place");
    int64_t max_keys = 0, max_bytes = 0;
    for (const KeyHeader& h : all) {
      max_keys = std::max(max_keys, h.keys);
      max_bytes = std::max(max_bytes, h.bytes);
    }
    // ---- the key sets, padded to the largest (an all-gather moves equal pieces): offsets, then chars
    const size_t off_piece = sizeof(int64_t) * (size_t)(max_keys + 1), chr_piece = (size_t)((max_bytes + 15) & ~(int64_t)15);
    Buf offs = dev_alloc(off_piece * (nranks + 1), s), chrs = dev_alloc(chr_piece * (nranks + 1) + 16, s);
    uint8_t* my_off = ptr<uint8_t>(offs) + off_piece * nranks;
    uint8_t* my_chr = ptr<uint8_t>(chrs) + chr_piece * nranks;
    CS_HIP(hipMemsetAsync(my_off, 0, off_piece, s));
    if (keys->rows > 0) CS_HIP(hipMemcpyAsync(my_off, keys->d_offsets(), sizeof(int64_t) * (size_t)(keys->rows + 1), hipMemcpyDeviceToDevice, s));
    if (keys->nbytes > 0) CS_HIP(hipMemcpyAsync(my_chr, keys->d_chars(), (size_t)keys->nbytes, hipMemcpyDeviceToDevice, s));
    if (allgather(ctx, my_off, offs->p, off_piece, s) != 0) fail(CS_ERR_INTERNAL, "category_build_distributed: the exchange of the key offsets failed");
    if (chr_piece && allgather(ctx, my_chr, chrs->p, chr_piece, s) != 0) fail(CS_ERR_INTERNAL, "category_build_distributed: the exchange of the key bytes failed");
    // ---- every rank merges the same key sets; this rank's codes go through its part of the merged table
    std::vector<std::unique_ptr<cs_column>> sets(nranks);
    std::vector<const cs_column*> set_ptrs(nranks);
    for (int r = 0; r < nranks; ++r) {
      auto c = std::make_unique<cs_column>();
      c->rows = all[r].keys;
      c->nbytes = all[r].bytes;
      c->chars = dev_wrap(ptr<uint8_t>(chrs) + chr_piece * r, (size_t)all[r].bytes);
      c->offsets = dev_wrap(ptr<uint8_t>(offs) + off_piece * r, sizeof(int64_t) * (size_t)(all[r].keys + 1));
      if (all[r].null_first && all[r].keys > 0) {
        c->validity = dev_alloc(validity_bytes(all[r].keys), s);
        CS_HIP(hipMemsetAsync(c->validity->p, 0xFF, validity_bytes(all[r].keys), s));
        CS_HIP(hipMemsetAsync(c->validity->p, 0xFE, 1, s));
        c->null_count = 1;
      } else {
        c->null_count = 0;
      }
      set_ptrs[r] = c.get();
      sets[r] = std::move(c);
    }
    auto merged = std::make_unique<cs_category>();
    merged->rows = local->rows;
    merged->values = dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(local->rows, 1), s);
    cs_column* mk = nullptr;
    const int rc = cs_category_merge_gathered(local.get(), set_ptrs.data(), nranks, rank, stream, &mk, ptr<int32_t>(merged->values));
    if (rc != 0) fail(rc, cs_last_error());
    merged->keys.reset(mk);
    CS_HIP(hipStreamSynchronize(s));  // (the gathered buffers leave scope)
    *out = merged.release();
  });
}

int cs_category_build_distributed(const cs_column* col, void* nccl_comm, int nranks, int rank, cs_stream stream, cs_category** out) {
  return guard([&] {
    if (!nccl_comm && nranks > 1) fail(CS_ERR_INVALID_ARG, "category_build_distributed: no communicator");
    if (nranks > 1 || nccl_comm) {
      resolve_rccl();
      if (!g_allgather) fail(CS_ERR_INTERNAL, "category_build_distributed: no RCCL in this process (ncclAllGather not found, librccl.so not loadable)");
    }
    const int rc = cs_category_build_distributed_with(col, &rccl_allgather, nccl_comm, nranks, rank, stream, out);
    if (rc != 0) fail(rc, cs_last_error());
  });
}

}  // extern "C"
