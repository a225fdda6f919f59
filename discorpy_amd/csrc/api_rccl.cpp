// api_rccl.cpp -- the north star's exchange step without torch: one process per GPU, every rank runs the stack kernel on
// its depth shard and ONE RCCL all-gather over xGMI reassembles the (depth, nrows, width) block on every rank
// (the loops over depth of postprocessing.py:226-228, 310-312 carry no state).
//
// librccl.so is NOT a link-time dependency: it is dlopen'ed on first use, from the directory of the HIP runtime this
// process already runs on (PyTorch-ROCm bundles its own libamdhip64.so + librccl.so; two HIP runtimes in one process
// do not work, so the RCCL next to the loaded runtime is the only one that can be used), falling back to the
// default search path.  Everything here fails with DCP_ERR_UNSUPPORTED when no RCCL can be loaded.
#include "api_common.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

using namespace dcpapi;

namespace {

struct RcclUniqueId {
  char internal[128];
};
typedef void* rcclComm;
typedef int (*fn_get_unique_id)(RcclUniqueId*);
typedef int (*fn_comm_init_rank)(rcclComm*, int, RcclUniqueId, int);
typedef int (*fn_comm_destroy)(rcclComm);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, rcclComm, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, rcclComm, hipStream_t);
typedef int (*fn_group)(void);
typedef const char* (*fn_error_string)(int);
constexpr int kRcclFloat = 7;      // ncclFloat32 (rccl.h)

struct Rccl {
  void* handle = nullptr;
  std::string path, why;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_broadcast broadcast = nullptr;
  fn_group group_start = nullptr, group_end = nullptr;
  fn_error_string error_string = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    std::string cands[4];
    int n = 0;
    if (const char* env = getenv("DCP_RCCL_PATH")) cands[n++] = env;
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {     // the HIP runtime this library is bound to
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        cands[n++] = dir + "librccl.so";
        cands[n++] = dir + "librccl.so.1";
      }
    }
    cands[n++] = "librccl.so.1";
    for (int i = 0; i < n && !r.handle; ++i) {
      r.handle = dlopen(cands[i].c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.handle) r.path = cands[i];
      else {
        const char* e = dlerror();
        r.why = e ? e : "dlopen failed";
      }
    }
    if (!r.handle) return;
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
    r.broadcast = (fn_broadcast)dlsym(r.handle, "ncclBroadcast");
    r.group_start = (fn_group)dlsym(r.handle, "ncclGroupStart");
    r.group_end = (fn_group)dlsym(r.handle, "ncclGroupEnd");
    r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.broadcast || !r.group_start || !r.group_end) {
      r.why = "librccl.so lacks an expected symbol";
      dlclose(r.handle);
      r.handle = nullptr;
    }
  });
  return r;
}

int need_rccl() {
  Rccl& r = rccl();
  if (!r.handle) return fail(DCP_ERR_UNSUPPORTED, "RCCL is not available: %s", r.why.empty() ? "librccl.so not found" : r.why.c_str());
  return DCP_OK;
}

#define DCP_RCCL(expr)                                                                                       \
  do {                                                                                                       \
    const int r_ = (expr);                                                                                   \
    if (r_ != 0)                                                                                             \
      return fail(DCP_ERR_HIP, "%s failed: %s", #expr, rccl().error_string ? rccl().error_string(r_) : "RCCL error"); \
  } while (0)

// what dcp_rccl_comm_create hands out: the communicator plus the side stream and events of the pipelined exchange
struct Comm {
  rcclComm comm = nullptr;
  int world = 0, rank = 0, device = 0;
  hipStream_t side = nullptr;
  hipEvent_t computed = nullptr, gathered = nullptr;
};

}  // namespace

extern "C" {

int dcp_rccl_available(void) { return rccl().handle ? 1 : 0; }

int dcp_rccl_unique_id(void* id, size_t bytes) {
  int rc;
  if (!id || bytes < sizeof(RcclUniqueId)) return fail(DCP_ERR_INVALID_ARG, "the id buffer must hold %zu bytes", sizeof(RcclUniqueId));
  if ((rc = need_rccl()) != DCP_OK) return rc;
  RcclUniqueId u;
  DCP_RCCL(rccl().get_unique_id(&u));
  memcpy(id, &u, sizeof(u));
  return DCP_OK;
}

int dcp_rccl_comm_create(void** comm, int world_size, int rank, const void* id, int device) {
  int rc;
  if (!comm || !id) return fail(DCP_ERR_INVALID_ARG, "null communicator / id pointer");
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(DCP_ERR_INVALID_ARG, "rank %d outside a world of %d", rank, world_size);
  if ((rc = need_rccl()) != DCP_OK) return rc;
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  int dev = 0;
  DCP_HIP(hipGetDevice(&dev));
  RcclUniqueId u;
  memcpy(&u, id, sizeof(u));
  Comm* c = new Comm();
  c->world = world_size;
  c->rank = rank;
  c->device = dev;
  const int r = rccl().comm_init_rank(&c->comm, world_size, u, rank);
  if (r != 0) {
    delete c;
    return fail(DCP_ERR_HIP, "ncclCommInitRank failed: %s", rccl().error_string ? rccl().error_string(r) : "RCCL error");
  }
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->computed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->gathered, hipEventDisableTiming) != hipSuccess) {
    (void)rccl().comm_destroy(c->comm);
    delete c;
    return fail(DCP_ERR_HIP, "cannot create the exchange stream / events");
  }
  *comm = c;
  return DCP_OK;
}

int dcp_rccl_comm_destroy(void* comm) {
  if (!comm) return DCP_OK;
  Comm* c = (Comm*)comm;
  DeviceScope scope(c->device);
  if (c->side) {
    (void)hipStreamSynchronize(c->side);
    (void)hipStreamDestroy(c->side);
  }
  if (c->computed) (void)hipEventDestroy(c->computed);
  if (c->gathered) (void)hipEventDestroy(c->gathered);
  const int r = rccl().handle ? rccl().comm_destroy(c->comm) : 0;
  delete c;
  if (r != 0) return fail(DCP_ERR_HIP, "ncclCommDestroy failed");
  return DCP_OK;
}

int dcp_unwarp_stack_rows_rccl_f32(const float* vol, float* out, int64_t depth_local, int64_t height, int64_t width, int64_t proj_stride,
                                   int64_t row_stride, double xcenter, double ycenter, const double* list_fact, int nfact, double row_start,
                                   int64_t nrows, int coord_round_f32, int blend_mode, void* comm, int pipeline, void* stream) {
  int rc;
  if (!comm) return fail(DCP_ERR_INVALID_ARG, "null communicator (dcp_rccl_comm_create)");
  if ((rc = need_rccl()) != DCP_OK) return rc;
  Comm* c = (Comm*)comm;
  if (depth_local < 0 || nrows < 0 || width <= 0 || height <= 0) return fail(DCP_ERR_INVALID_ARG, "negative depth / nrows or empty projections");
  if (depth_local > 0 && nrows > 0 && (!vol || !out)) return fail(DCP_ERR_INVALID_ARG, "null volume / result pointer");
  if (depth_local == 0 || nrows == 0) return DCP_OK;
  DeviceScope scope(c->device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", c->device, hipGetErrorString(scope.status));
  const size_t block = (size_t)nrows * (size_t)width;                 // floats per projection of the result
  float* mine = out + (size_t)c->rank * (size_t)depth_local * block;  // depth is the outer axis: this rank's block is contiguous
  hipStream_t hs = (hipStream_t)stream;
  int nsub = pipeline < 1 ? 1 : pipeline;
  if ((int64_t)nsub > depth_local) nsub = (int)depth_local;
  if (nsub <= 1 || c->world == 1) {
    // kernel, then the in-place all-gather behind it on the same stream (sendbuff = recvbuff + rank * count)
    if ((rc = dcp_unwarp_stack_rows_f32(vol, mine, depth_local, height, width, proj_stride, row_stride, xcenter, ycenter, list_fact, nfact,
                                        row_start, nrows, coord_round_f32, blend_mode, DCP_MEM_DEVICE, c->device, stream)) != DCP_OK)
      return rc;
    if (c->world > 1) DCP_RCCL(rccl().all_gather(mine, out, (size_t)depth_local * block, kRcclFloat, c->comm, hs));
    return DCP_OK;
  }
  // pipelined: the shard in `nsub` depth sub-blocks; the exchange of sub-block s (every rank's piece broadcast into its place
  // of the result, one group on the side stream) runs under the kernel of sub-block s + 1
  DCP_HIP(hipEventRecord(c->gathered, hs));                       // the side stream starts behind whatever the caller queued
  DCP_HIP(hipStreamWaitEvent(c->side, c->gathered, 0));
  const int64_t base = depth_local / nsub, extra = depth_local % nsub;
  int64_t s0 = 0;
  for (int s = 0; s < nsub; ++s) {
    const int64_t n = base + (s < extra ? 1 : 0);
    if ((rc = dcp_unwarp_stack_rows_f32(vol + (size_t)s0 * (size_t)proj_stride, mine + (size_t)s0 * block, n, height, width, proj_stride,
                                        row_stride, xcenter, ycenter, list_fact, nfact, row_start, nrows, coord_round_f32, blend_mode,
                                        DCP_MEM_DEVICE, c->device, stream)) != DCP_OK)
      return rc;
    DCP_HIP(hipEventRecord(c->computed, hs));
    DCP_HIP(hipStreamWaitEvent(c->side, c->computed, 0));
    DCP_RCCL(rccl().group_start());
    for (int r = 0; r < c->world; ++r) {
      float* piece = out + ((size_t)r * (size_t)depth_local + (size_t)s0) * block;
      const int e = rccl().broadcast(piece, piece, (size_t)n * block, kRcclFloat, r, c->comm, c->side);
      if (e != 0) {
        (void)rccl().group_end();
        return fail(DCP_ERR_HIP, "ncclBroadcast failed: %s", rccl().error_string ? rccl().error_string(e) : "RCCL error");
      }
    }
    DCP_RCCL(rccl().group_end());
    s0 += n;
  }
  DCP_HIP(hipEventRecord(c->gathered, c->side));                  // the caller's stream continues when the last exchange has landed
  DCP_HIP(hipStreamWaitEvent(hs, c->gathered, 0));
  return DCP_OK;
}

}  // extern "C"
