// api_spline.cpp -- spline orders 2..5 (scipy's prefiltered B-spline interpolation, spline_kernels.hip): the
// per-device coefficient workspace, host staging, and the *_spline_f32 entry points of the C ABI.
#include "api_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <mutex>

using namespace dcpapi;

namespace {

// Per-device float64 coefficient workspaces (grow-only): kSlots of them, so that spline calls on TWO streams -- independent frames
// handed over alternately, INTEGRATION.md section 7 -- each keep their planes and the prefilter of one frame (memory-bound) runs
// under the gather of the other (LDS- and VALU-bound) instead of waiting for the device (rounds 2-5: one workspace, and a call on
// another stream than the last one synchronised the device first).  A slot remembers the stream that used it last and an event
// recorded behind that use; a call on another stream waits for the event ON THE DEVICE (hipStreamWaitEvent), never on the host.
// A slot's mutex is held by run_spline from acquire() until its last kernel is enqueued (host callers: until the result is back),
// so two host threads never interleave their passes over one slot's planes; calls on different GPUs run side by side.
struct SplineWorkspace {
  static constexpr int kSlots = 2;
  struct Slot {
    std::mutex use;
    void* buf = nullptr;
    size_t cap = 0;
    hipStream_t last = nullptr;
    bool used = false;
    hipEvent_t done = nullptr;
    unsigned long long tick = 0;
  };
  std::mutex mu;                     // the tables below (never held while waiting for a slot)
  Slot slot[64][kSlots];
  unsigned long long clock = 0;

  // The slot for a call on `stream` of the current device, LOCKED (release() unlocks), its planes grown to `bytes` and ordered behind
  // the slot's previous use.  Preference: the slot this stream used last; a slot never used; the least recently used one.
  hipError_t acquire(size_t bytes, hipStream_t stream, int dev, Slot** out) {
    int order[kSlots];
    bool own = false;                  // the preferred slot was last used by this very stream (read under `mu`, like every table entry)
    {
      std::lock_guard<std::mutex> lock(mu);
      int n = 0;
      for (int k = 0; k < kSlots; ++k)
        if (slot[dev][k].used && slot[dev][k].last == stream) order[n++] = k;
      for (int k = 0; k < kSlots; ++k)
        if (!slot[dev][k].used) order[n++] = k;
      for (int pass = 0; pass < kSlots; ++pass) {          // the rest, least recently used first
        int best = -1;
        for (int k = 0; k < kSlots; ++k) {
          bool taken = false;
          for (int i = 0; i < n; ++i) taken = taken || order[i] == k;
          if (!taken && (best < 0 || slot[dev][k].tick < slot[dev][best].tick)) best = k;
        }
        if (best >= 0) order[n++] = best;
      }
      own = slot[dev][order[0]].used && slot[dev][order[0]].last == stream;
    }
    Slot* s = nullptr;
    // a slot of this stream is taken even if another thread holds it right now (calls on one stream are ordered anyway); otherwise
    // the first free one in preference order, and if every slot is busy, wait for the preferred one
    if (own) {
      s = &slot[dev][order[0]];
      s->use.lock();
    } else {
      for (int i = 0; i < kSlots && !s; ++i)
        if (slot[dev][order[i]].use.try_lock()) s = &slot[dev][order[i]];
      if (!s) {
        s = &slot[dev][order[0]];
        s->use.lock();
      }
    }
    hipError_t e = hipSuccess;
    if (!s->done) e = hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
    if (e == hipSuccess && s->cap < bytes) {
      if (s->buf) {                                          // the previous user must be done before its planes go away
        e = hipEventSynchronize(s->done);
        if (e == hipSuccess) e = hipFree(s->buf);
        s->buf = nullptr;
        s->cap = 0;
      }
      if (e == hipSuccess) e = hipMalloc(&s->buf, bytes);
      if (e == hipSuccess) s->cap = bytes;
    }
    // another stream used these planes last: this call's kernels start behind that use (nothing to wait for on the same stream).
    // (`used` / `last` of a slot whose `use` mutex this thread holds change only under that mutex)
    if (e == hipSuccess && s->used && s->last != stream) e = hipStreamWaitEvent(stream, s->done, 0);
    if (e != hipSuccess) {
      s->use.unlock();
      return e;
    }
    {
      std::lock_guard<std::mutex> lock(mu);
      s->used = true;
      s->last = stream;
      s->tick = ++clock;
    }
    *out = s;
    return hipSuccess;
  }
  // every kernel of the call has been enqueued on `stream`: mark the end of the use and let the next caller in
  void release(Slot* s, hipStream_t stream) {
    (void)hipEventRecord(s->done, stream);
    s->use.unlock();
  }
};
SplineWorkspace g_spline_ws;
struct SlotGuard {           // releases the slot on every return path of run_spline
  SplineWorkspace::Slot* s;
  hipStream_t st;
  ~SlotGuard() { g_spline_ws.release(s, st); }
};

int spline_poles(int order, double* z) {
  switch (order) {
    case 2: z[0] = std::sqrt(8.0) - 3.0; return 1;
    case 3: z[0] = std::sqrt(3.0) - 2.0; return 1;
    case 4:
      z[0] = std::sqrt(664.0 - std::sqrt(438976.0)) + std::sqrt(304.0) - 19.0;
      z[1] = std::sqrt(664.0 + std::sqrt(438976.0)) - std::sqrt(304.0) - 19.0;
      return 2;
    case 5:
      z[0] = std::sqrt(67.5 - std::sqrt(4436.25)) + std::sqrt(26.25) - 6.5;
      z[1] = std::sqrt(67.5 + std::sqrt(4436.25)) - std::sqrt(26.25) - 6.5;
      return 2;
    default: return 0;
  }
}

}  // namespace

namespace dcpapi {

int release_spline_workspace() {
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) return DCP_OK;      // no runtime / no device: nothing was ever allocated
  for (int dev = 0; dev < 64; ++dev)
    for (int k = 0; k < SplineWorkspace::kSlots; ++k) {
      SplineWorkspace::Slot& sl = g_spline_ws.slot[dev][k];
      std::lock_guard<std::mutex> exclusive(sl.use);
      if (!sl.buf) continue;
      DCP_HIP(hipSetDevice(dev));
      DCP_HIP(hipDeviceSynchronize());
      (void)hipFree(sl.buf);
      std::lock_guard<std::mutex> lock(g_spline_ws.mu);
      sl.buf = nullptr;
      sl.cap = 0;
      sl.used = false;
      sl.last = nullptr;
    }
  (void)hipSetDevice(prev);
  return DCP_OK;
}

int run_spline(int map_kind, const void* src, void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs,
               const dcp::MapArgs& map_in, const void* ycoord, const void* xcoord, int coord_dtype, int64_t npts, int order,
               int mode, int mem_kind, int device, void* stream) {
  int rc;
  if ((rc = check_image_typed(src, dst, dtype, H, W, rs, cs)) != DCP_OK) return rc;
  dcp::MapArgs map = map_in;
  // the host's tile-deviation certificate lets the gather stage its taps in LDS (spline_wg_kernel)
  map.tile_dev_ok = (map_kind == 0 || map_kind == 1) && g_tile_cert.load() && H > 0 && W > 0
                        ? tile_deviation_certified(map_kind == 0 ? dcp::kRadial : dcp::kPersp, map, H, W)
                        : 0;
  if (order < 2 || order > 5) return fail(DCP_ERR_INVALID_ARG, "spline order %d outside [2, 5]", order);
  // bit 8 of boundary_mode (DCP_SPLINE_SCIPY_SUM): accumulate the taps in scipy's operation order instead of the factorised sum
  const int exact_sum = (mode >= 0 && (mode & DCP_SPLINE_SCIPY_SUM)) ? 1 : 0;
  if (mode >= 0) mode &= ~DCP_SPLINE_SCIPY_SUM;
  if (mode < 0 || mode > 7) return fail(DCP_ERR_INVALID_ARG, "unknown boundary mode %d", mode);
  if (map_kind == 2) {
    if (npts < 0) return fail(DCP_ERR_INVALID_ARG, "npts < 0");
    if (npts > 0 && (!ycoord || !xcoord)) return fail(DCP_ERR_INVALID_ARG, "null coordinate pointer");
    if (coord_dtype != DCP_COORD_F32 && coord_dtype != DCP_COORD_F64) return fail(DCP_ERR_INVALID_ARG, "unknown coord_dtype %d", coord_dtype);
  }
  if (H > 1000000 || W > 1000000) return fail(DCP_ERR_UNSUPPORTED, "image too large for the spline path");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipStream_t st = (hipStream_t)stream;
  dcp::SplineArgs a;
  memset(&a, 0, sizeof(a));
  a.H = (int32_t)H;
  a.W = (int32_t)W;
  a.src_dtype = a.dst_dtype = dtype;
  a.order = order;
  a.mode = mode;
  a.exact_sum = exact_sum;
  a.pad = (mode == dcp::kModeNearest || mode == dcp::kModeGridConstant) ? 12 : 0;
  a.Hp = a.H + 2 * a.pad;
  a.Wp = a.W + 2 * a.pad;
  // ('nearest': scipy prefilters the edge-padded array with the reflect boundary -- see spline_filter_kind() in the oracle)
  a.filter_kind = (mode == dcp::kModeReflect || mode == dcp::kModeGridMirror || mode == dcp::kModeNearest) ? dcp::kSplReflect
                  : mode == dcp::kModeGridWrap                                 ? dcp::kSplWrap
                                                                               : dcp::kSplMirror;
  a.npoles = spline_poles(order, a.poles);
  for (int axis = 0; axis < 2; ++axis) {
    const double n = axis == 0 ? (double)a.Hp : (double)a.Wp;
    for (int p = 0; p < a.npoles; ++p)
      a.zpow[axis][p] = std::pow(a.poles[p], a.filter_kind == dcp::kSplMirror ? n - 1.0 : n);
  }
  const size_t plane = (size_t)a.Hp * (size_t)a.Wp * sizeof(double);
  int cur_dev = 0;
  DCP_HIP(hipGetDevice(&cur_dev));                           // (DeviceScope above has selected it)
  if (cur_dev < 0 || cur_dev >= 64) return fail(DCP_ERR_UNSUPPORTED, "device index %d", cur_dev);
  SplineWorkspace::Slot* slot = nullptr;
  DCP_HIP(g_spline_ws.acquire(2 * plane, st, cur_dev, &slot));
  SlotGuard guard{slot, st};
  a.coef = (double*)slot->buf;
  a.scratch = a.coef + (size_t)a.Hp * (size_t)a.Wp;
  dcp::CoordArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.npts = npts;
  ca.is_f64 = coord_dtype == DCP_COORD_F64;
  const int64_t nout = map_kind == 2 ? npts : H * W;
  if (mem_kind == DCP_MEM_DEVICE) {
    a.src = src;
    a.src_stride = (int32_t)rs;
    a.src_cstride = (int32_t)cs;
    ca.ycoord = ycoord;
    ca.xcoord = xcoord;
    DCP_HIP(dcp::launch_spline(a, map_kind, map, ca, dst, st));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  void *dsrc, *ddst, *dy = nullptr, *dx = nullptr;
  const size_t ext = extent_bytes_typed(H, W, rs, cs, dtype), esz = (size_t)dcp::elem_size(dtype);
  DCP_HIP(g_staging.get(0, ext, &dsrc));
  DCP_HIP(g_staging.get(1, (size_t)(nout > 0 ? nout : 1) * esz, &ddst));
  DCP_HIP(hipMemcpyAsync(dsrc, src, ext, hipMemcpyHostToDevice, st));
  if (map_kind == 2 && npts > 0) {
    const size_t csz = (size_t)npts * (ca.is_f64 ? 8 : 4);
    DCP_HIP(g_staging.get(2, csz, &dy));
    DCP_HIP(g_staging.get(3, csz, &dx));
    DCP_HIP(hipMemcpyAsync(dy, ycoord, csz, hipMemcpyHostToDevice, st));
    DCP_HIP(hipMemcpyAsync(dx, xcoord, csz, hipMemcpyHostToDevice, st));
  }
  a.src = dsrc;
  a.src_stride = (int32_t)rs;
  a.src_cstride = (int32_t)cs;
  ca.ycoord = dy;
  ca.xcoord = dx;
  DCP_HIP(dcp::launch_spline(a, map_kind, map, ca, ddst, st));
  if (nout > 0) DCP_HIP(hipMemcpyAsync(dst, ddst, (size_t)nout * esz, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

}  // namespace dcpapi

extern "C" {

int dcp_unwarp_image_spline_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                                int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                                int nfact, int order, int boundary_mode, int mem_kind, int device, void* stream) {
  int rc;
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  return run_spline(0, src, dst, dcp::kF32, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0, order,
                    boundary_mode, mem_kind, device, stream);
}

int dcp_perspective_image_spline_f32(const float* src, float* dst, int64_t height, int64_t width,
                                     int64_t src_row_stride, int64_t src_col_stride, const double* list_coef, int order,
                                     int boundary_mode, int mem_kind, int device, void* stream) {
  int rc;
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  dcp::MapArgs map;
  if ((rc = fill_map(&map, 0.0, 0.0, nullptr, 0, list_coef)) != DCP_OK) return rc;
  return run_spline(1, src, dst, dcp::kF32, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0, order,
                    boundary_mode, mem_kind, device, stream);
}

int dcp_unwarp_fused_spline_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                                int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                                int nfact, const double* list_coef, int order, int boundary_mode, int mem_kind, int device,
                                void* stream) {
  int rc;
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, list_coef)) != DCP_OK) return rc;
  if (height > 0 && width > 0) map.fast_div = homography_is_tame(list_coef, height, width);
  return run_spline(3, src, dst, dcp::kF32, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0,
                    order, boundary_mode, mem_kind, device, stream);
}

int dcp_remap_coords_spline_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                                int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype,
                                int64_t npts, int order, int boundary_mode, int mem_kind, int device, void* stream) {
  dcp::MapArgs map;
  memset(&map, 0, sizeof(map));
  if (npts == 0) return (order < 2 || order > 5) ? fail(DCP_ERR_INVALID_ARG, "spline order %d outside [2, 5]", order) : DCP_OK;
  return run_spline(2, src, dst, dcp::kF32, height, width, src_row_stride, src_col_stride, map, ycoord, xcoord, coord_dtype, npts,
                    order, boundary_mode, mem_kind, device, stream);
}

}  // extern "C"
