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
typedef int (*fn_comm_query)(const rcclComm, int*);
typedef int (*fn_get_version)(int*);
constexpr int kRcclFloat = 7;      // ncclFloat32 (rccl.h)
constexpr int kRcclInt64 = 4;      // ncclInt64

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
  fn_comm_query comm_count = nullptr, comm_user_rank = nullptr, comm_device = nullptr;      // optional: what the communicator itself says
  fn_get_version get_version = nullptr;
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
    r.comm_count = (fn_comm_query)dlsym(r.handle, "ncclCommCount");
    r.comm_user_rank = (fn_comm_query)dlsym(r.handle, "ncclCommUserRank");
    r.comm_device = (fn_comm_query)dlsym(r.handle, "ncclCommCuDevice");
    r.get_version = (fn_get_version)dlsym(r.handle, "ncclGetVersion");
    {                                  // the file that was really mapped (dlopen may have resolved a bare name through the search path)
      Dl_info where;
      if (r.get_unique_id && dladdr((void*)r.get_unique_id, &where) && where.dli_fname) r.path = where.dli_fname;
    }
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

// what dcp_rccl_comm_create hands out: the communicator, the side stream and events of the pipelined exchange, and the
// small device / pinned-host tables through which the ranks agree on their shard shapes before every exchange
constexpr int kMetaWords = 5;      // depth_local, nrows, width, pipeline, 1 if this rank's own arguments are unusable
struct Comm {
  rcclComm comm = nullptr;
  int world = 0, rank = 0, device = 0;
  hipStream_t side = nullptr;
  hipEvent_t computed = nullptr, gathered = nullptr;
  int64_t* meta_dev = nullptr;     // world x kMetaWords
  int64_t* meta_host = nullptr;    // pinned, world x kMetaWords
  bool agreed = false;             // meta_host holds the table of a completed agreement
  bool fixed = false;              // dcp_rccl_comm_fixed_shards: the caller vouches that every rank repeats the agreed shapes
  int64_t exchanges = 0, agreements = 0;
};

void release(Comm* c) {
  if (c->side) {
    (void)hipStreamSynchronize(c->side);
    (void)hipStreamDestroy(c->side);
  }
  if (c->computed) (void)hipEventDestroy(c->computed);
  if (c->gathered) (void)hipEventDestroy(c->gathered);
  if (c->meta_dev) (void)hipFree(c->meta_dev);
  if (c->meta_host) (void)hipHostFree(c->meta_host);
}

// Every rank's (depth_local, nrows, width, pipeline, arguments-ok) on every rank: one all-gather of 40 bytes per rank on the SIDE stream (the
// caller's stream is not waited for), then a wait for that stream only.  Collective: every rank of the world makes this call.
int agree_on_shards(Comm* c, int64_t depth_local, int64_t nrows, int64_t width, int pipeline, bool local_error) {
  int64_t* mine = c->meta_host + (size_t)c->rank * kMetaWords;
  if (c->fixed && c->agreed) {
    // The caller has promised (on EVERY rank) that the shapes of the last agreement repeat: no collective, no host wait -- the call is
    // then stream-ordered end to end, and back-to-back calls overlap.  A rank that breaks the promise is told so; its peers cannot be.
    if (local_error || mine[0] != depth_local || mine[1] != nrows || mine[2] != width || mine[3] != pipeline)
      return fail(DCP_ERR_INVALID_ARG, "dcp_rccl_comm_fixed_shards is set but this call's depth_local / nrows / width / pipeline differ from the "
                                       "agreed ones (%lld, %lld, %lld, %lld): clear it first", (long long)mine[0], (long long)mine[1],
                  (long long)mine[2], (long long)mine[3]);
    return DCP_OK;
  }
  c->agreed = false;
  mine[0] = depth_local;
  mine[1] = nrows;
  mine[2] = width;
  mine[3] = pipeline;
  mine[4] = local_error ? 1 : 0;
  ++c->agreements;
  if (c->world == 1) {
    c->agreed = !local_error;
    return DCP_OK;
  }
  // (the side stream may still carry the previous call's pipelined exchange, and RCCL serialises the operations of one communicator:
  // this wait is therefore also a wait for the previous exchange -- see the header; dcp_rccl_comm_fixed_shards avoids it)
  DCP_HIP(hipMemcpyAsync(c->meta_dev + (size_t)c->rank * kMetaWords, mine, kMetaWords * sizeof(int64_t), hipMemcpyHostToDevice, c->side));
  DCP_RCCL(rccl().all_gather(c->meta_dev + (size_t)c->rank * kMetaWords, c->meta_dev, kMetaWords, kRcclInt64, c->comm, c->side));
  DCP_HIP(hipMemcpyAsync(c->meta_host, c->meta_dev, (size_t)c->world * kMetaWords * sizeof(int64_t), hipMemcpyDeviceToHost, c->side));
  DCP_HIP(hipStreamSynchronize(c->side));
  c->agreed = true;
  return DCP_OK;
}

// sub-block s of `nsub` of a shard of n projections: [first, first + count)
inline void sub_block(int64_t n, int nsub, int s, int64_t* first, int64_t* count) {
  const int64_t base = n / nsub, extra = n % nsub;
  *first = (int64_t)s * base + (s < extra ? s : extra);
  *count = base + (s < extra ? 1 : 0);
}

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
  const size_t meta = (size_t)world_size * kMetaWords * sizeof(int64_t);
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->computed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->gathered, hipEventDisableTiming) != hipSuccess || hipMalloc((void**)&c->meta_dev, meta) != hipSuccess ||
      hipHostMalloc((void**)&c->meta_host, meta, hipHostMallocDefault) != hipSuccess) {
    release(c);
    (void)rccl().comm_destroy(c->comm);
    delete c;
    return fail(DCP_ERR_HIP, "cannot create the exchange stream / events / shard table");
  }
  *comm = c;
  return DCP_OK;
}

int dcp_rccl_comm_destroy(void* comm) {
  if (!comm) return DCP_OK;
  Comm* c = (Comm*)comm;
  DeviceScope scope(c->device);
  release(c);
  const int r = rccl().handle ? rccl().comm_destroy(c->comm) : 0;
  delete c;
  if (r != 0) return fail(DCP_ERR_HIP, "ncclCommDestroy failed");
  return DCP_OK;
}

int dcp_rccl_comm_fixed_shards(void* comm, int on) {
  if (!comm) return fail(DCP_ERR_INVALID_ARG, "null communicator (dcp_rccl_comm_create)");
  ((Comm*)comm)->fixed = on != 0;
  return DCP_OK;
}

int dcp_rccl_comm_info(void* comm, int64_t* info, int ninfo, int64_t* shard_depths, int nshards, char* librccl_path, size_t path_bytes) {
  if (!comm) return fail(DCP_ERR_INVALID_ARG, "null communicator (dcp_rccl_comm_create)");
  int rc;
  if ((rc = need_rccl()) != DCP_OK) return rc;
  Comm* c = (Comm*)comm;
  Rccl& r = rccl();
  int count = -1, user_rank = -1, dev = -1, version = -1;
  if (r.comm_count) DCP_RCCL(r.comm_count(c->comm, &count));
  if (r.comm_user_rank) DCP_RCCL(r.comm_user_rank(c->comm, &user_rank));
  if (r.comm_device) DCP_RCCL(r.comm_device(c->comm, &dev));
  if (r.get_version) DCP_RCCL(r.get_version(&version));
  const int64_t v[10] = {count, user_rank, dev, version, c->world, c->rank, c->device, c->agreed ? 1 : 0, c->exchanges, c->agreements};
  for (int i = 0; i < ninfo && i < 10; ++i)
    if (info) info[i] = v[i];
  for (int i = 0; shard_depths && i < nshards; ++i) shard_depths[i] = (c->agreed && i < c->world) ? c->meta_host[(size_t)i * kMetaWords] : -1;
  if (librccl_path && path_bytes) snprintf(librccl_path, path_bytes, "%s", r.path.c_str());
  return DCP_OK;
}

int dcp_unwarp_stack_rows_rccl_f32(const float* vol, float* out, int64_t depth_local, int64_t height, int64_t width, int64_t proj_stride,
                                   int64_t row_stride, double xcenter, double ycenter, const double* list_fact, int nfact, double row_start,
                                   int64_t nrows, int coord_round_f32, int blend_mode, void* comm, int pipeline, void* stream) {
  int rc;
  if (!comm) return fail(DCP_ERR_INVALID_ARG, "null communicator (dcp_rccl_comm_create)");
  if ((rc = need_rccl()) != DCP_OK) return rc;
  Comm* c = (Comm*)comm;
  // A rank whose own arguments are unusable still takes part in the agreement and says so there: every rank then returns an
  // error from THIS call and none is left waiting in a collective its peer never entered.
  const bool local_error = depth_local < 0 || nrows < 0 || width <= 0 || height <= 0 || (depth_local > 0 && nrows > 0 && !vol) ||
                           ((depth_local > 0 || c->world > 1) && nrows > 0 && !out);   // a rank without projections still receives
  DeviceScope scope(c->device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", c->device, hipGetErrorString(scope.status));
  if (pipeline < 1) pipeline = 1;
  if ((rc = agree_on_shards(c, depth_local, nrows, width, pipeline, local_error)) != DCP_OK) return rc;
  if (local_error) {
    if (depth_local < 0 || nrows < 0 || width <= 0 || height <= 0) return fail(DCP_ERR_INVALID_ARG, "negative depth / nrows or empty projections");
    return fail(DCP_ERR_INVALID_ARG, "null volume / result pointer");
  }
  for (int r = 0; r < c->world; ++r)
    if (c->meta_host[(size_t)r * kMetaWords + 4] != 0) {
      c->agreed = false;
      return fail(DCP_ERR_INVALID_ARG, "rank %d was called with unusable arguments (its own error says which); nothing was exchanged", r);
    }
  int64_t total = 0, deepest = 0, first_of_mine = 0;
  bool even = true;
  for (int r = 0; r < c->world; ++r) {
    const int64_t* m = c->meta_host + (size_t)r * kMetaWords;
    if (m[1] != nrows || m[2] != width || m[3] != pipeline || m[0] < 0) c->agreed = false;
    if (m[1] != nrows || m[2] != width || m[3] != pipeline)
      return fail(DCP_ERR_INVALID_ARG, "rank %d was called with nrows %lld, width %lld, pipeline %lld; rank %d with %lld, %lld, %d: the ranks must agree",
                  r, (long long)m[1], (long long)m[2], (long long)m[3], c->rank, (long long)nrows, (long long)width, pipeline);
    if (m[0] < 0) return fail(DCP_ERR_INVALID_ARG, "rank %d reports a negative shard depth", r);
    if (r < c->rank) first_of_mine += m[0];
    if (m[0] != c->meta_host[0]) even = false;
    if (m[0] > deepest) deepest = m[0];
    total += m[0];
  }
  if (total == 0 || nrows == 0) return DCP_OK;                       // nothing to compute anywhere: every rank returns here
  ++c->exchanges;
  const size_t block = (size_t)nrows * (size_t)width;                // floats per projection of the result
  // depth is the outer axis of the result and the shards are laid down in rank order: rank r's block starts at the sum of the
  // shards before it and is contiguous
  float* mine = out + (size_t)first_of_mine * block;
  hipStream_t hs = (hipStream_t)stream;
  int nsub = (int64_t)pipeline > deepest ? (int)deepest : pipeline;
  if (c->world == 1) nsub = pipeline > depth_local ? (int)depth_local : pipeline;
  if (nsub <= 1 && (even || c->world == 1)) {
    // kernel, then the in-place all-gather behind it on the same stream (sendbuff = recvbuff + rank * count)
    if ((rc = dcp_unwarp_stack_rows_f32(vol, mine, depth_local, height, width, proj_stride, row_stride, xcenter, ycenter, list_fact, nfact,
                                        row_start, nrows, coord_round_f32, blend_mode, DCP_MEM_DEVICE, c->device, stream)) != DCP_OK)
      return rc;
    if (c->world > 1) DCP_RCCL(rccl().all_gather(mine, out, (size_t)depth_local * block, kRcclFloat, c->comm, hs));
    return DCP_OK;
  }
  if (c->world == 1) {                                               // sub-blocks, no exchange
    for (int s = 0; s < nsub; ++s) {
      int64_t s0, n;
      sub_block(depth_local, nsub, s, &s0, &n);
      if (n > 0 && (rc = dcp_unwarp_stack_rows_f32(vol + (size_t)s0 * (size_t)proj_stride, mine + (size_t)s0 * block, n, height, width, proj_stride,
                                                   row_stride, xcenter, ycenter, list_fact, nfact, row_start, nrows, coord_round_f32, blend_mode,
                                                   DCP_MEM_DEVICE, c->device, stream)) != DCP_OK)
        return rc;
    }
    return DCP_OK;
  }
  // Sub-blocks and / or ragged shards: every rank cuts ITS shard into `nsub` depth sub-blocks (a shard shorter than nsub has
  // empty ones); the exchange of sub-block s -- every rank's piece broadcast into its place of the result, one group on the
  // side stream, empty pieces skipped by everybody -- runs under the kernel of sub-block s + 1.
  if (nsub < 1) nsub = 1;
  DCP_HIP(hipEventRecord(c->gathered, hs));                          // the side stream starts behind whatever the caller queued
  DCP_HIP(hipStreamWaitEvent(c->side, c->gathered, 0));
  int status = DCP_OK;
  for (int s = 0; s < nsub && status == DCP_OK; ++s) {
    int64_t s0, n;
    sub_block(depth_local, nsub, s, &s0, &n);
    if (n > 0)
      status = dcp_unwarp_stack_rows_f32(vol + (size_t)s0 * (size_t)proj_stride, mine + (size_t)s0 * block, n, height, width, proj_stride,
                                         row_stride, xcenter, ycenter, list_fact, nfact, row_start, nrows, coord_round_f32, blend_mode,
                                         DCP_MEM_DEVICE, c->device, stream);
    if (status != DCP_OK) break;
    if (hipEventRecord(c->computed, hs) != hipSuccess || hipStreamWaitEvent(c->side, c->computed, 0) != hipSuccess) {
      status = fail(DCP_ERR_HIP, "cannot chain the exchange stream behind the kernel");
      break;
    }
    int e = rccl().group_start();
    int64_t first = 0;
    for (int r = 0; r < c->world && e == 0; ++r) {
      const int64_t dr = c->meta_host[(size_t)r * kMetaWords];
      int64_t r0, rn;
      sub_block(dr, nsub, s, &r0, &rn);
      float* piece = out + (size_t)(first + r0) * block;
      if (rn > 0) e = rccl().broadcast(piece, piece, (size_t)rn * block, kRcclFloat, r, c->comm, c->side);
      first += dr;
    }
    const int e2 = rccl().group_end();
    if (e != 0 || e2 != 0)
      status = fail(DCP_ERR_HIP, "the exchange of depth sub-block %d failed: %s", s,
                    rccl().error_string ? rccl().error_string(e != 0 ? e : e2) : "RCCL error");
  }
  // on every path, failed ones included, the caller's stream continues behind whatever was queued on the side stream; after a
  // failure the other ranks may be blocked in their collectives and the communicator must be destroyed (header)
  if (hipEventRecord(c->gathered, c->side) != hipSuccess || hipStreamWaitEvent(hs, c->gathered, 0) != hipSuccess) {
    if (status == DCP_OK) status = fail(DCP_ERR_HIP, "cannot re-join the exchange stream");
  }
  return status;
}

}  // extern "C"
