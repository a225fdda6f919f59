// unwarp_api.cpp -- the C ABI declared in include/discorpy_hip.h: argument validation,
// error reporting, host<->device staging for DCP_MEM_HOST callers, tuning knobs, and thin
// memory/stream/event helpers.  The kernels live in unwarp_kernels.hip.
#include "../../include/discorpy_hip.h"
#include "dcp_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define DCP_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(DCP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

std::atomic<int> g_tile_rows{16}, g_xcd_remap{2}, g_coef_lds{0}, g_d_chunk{16}, g_pipe_depth{2}, g_lds_gather{1}, g_stack_chunk_kb{24576};

dcp::LaunchOpts current_opts() {
  dcp::LaunchOpts o;
  o.tile_rows = g_tile_rows.load();
  o.xcd_remap = g_xcd_remap.load();
  o.coef_lds = g_coef_lds.load();
  o.d_chunk = g_d_chunk.load();
  o.pipe_depth = g_pipe_depth.load();
  o.lds_gather = g_lds_gather.load();
  return o;
}

// Selects `device` for the calling thread for the lifetime of the object (no-op for device < 0).
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t status = hipSuccess;
  explicit DeviceScope(int device) {
    if (device < 0) return;
    status = hipGetDevice(&prev);
    if (status != hipSuccess) return;
    if (prev != device) {
      status = hipSetDevice(device);
      switched = status == hipSuccess;
    }
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// Grow-only device scratch used for DCP_MEM_HOST calls; one set per host thread.
struct Staging {
  void* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap[4] = {0, 0, 0, 0};
  int device = -1;
  ~Staging() { release(); }
  void release() {
    for (int i = 0; i < 4; ++i) {
      if (buf[i]) (void)hipFree(buf[i]);
      buf[i] = nullptr;
      cap[i] = 0;
    }
  }
  hipError_t get(int slot, size_t bytes, void** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev != device) {
      release();
      device = dev;
    }
    if (bytes == 0) bytes = 4;
    if (cap[slot] < bytes) {
      if (buf[slot]) (void)hipFree(buf[slot]);
      buf[slot] = nullptr;
      cap[slot] = 0;
      e = hipMalloc(&buf[slot], bytes);
      if (e != hipSuccess) return e;
      cap[slot] = bytes;
    }
    *out = buf[slot];
    return hipSuccess;
  }
};
thread_local Staging g_staging;

// Two non-blocking streams per host thread for the streamed DCP_MEM_HOST stack path (uploads + kernels,
// downloads).  The legacy null stream would serialise the two directions.
struct HostStreams {
  hipStream_t up = nullptr, down = nullptr;
  int device = -1;
  ~HostStreams() { release(); }
  void release() {
    if (up) (void)hipStreamDestroy(up);
    if (down) (void)hipStreamDestroy(down);
    up = down = nullptr;
  }
  hipError_t get(hipStream_t* u, hipStream_t* d) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev != device) {
      release();
      device = dev;
    }
    if (!up && (e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking)) != hipSuccess) return e;
    if (!down && (e = hipStreamCreateWithFlags(&down, hipStreamNonBlocking)) != hipSuccess) return e;
    *u = up;
    *d = down;
    return hipSuccess;
  }
};
thread_local HostStreams g_host_streams;

int check_image_typed(const void* src, const void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs);

int sampler_of(int order, int blend_mode, int* sampler) {
  if (order == 0) {
    *sampler = dcp::kNearest;
    return DCP_OK;
  }
  if (order != 1)
    return fail(DCP_ERR_UNSUPPORTED, "spline order %d is not implemented on the GPU path (only 0 and 1)", order);
  switch (blend_mode) {
    case DCP_BLEND_SCIPY: *sampler = dcp::kScipy; return DCP_OK;
    case DCP_BLEND_F64LERP: *sampler = dcp::kF64Lerp; return DCP_OK;
    case DCP_BLEND_F32LERP: *sampler = dcp::kF32Lerp; return DCP_OK;
    default: return fail(DCP_ERR_INVALID_ARG, "unknown blend_mode %d", blend_mode);
  }
}

int check_image(const void* src, const void* dst, int64_t H, int64_t W, int64_t rs, int64_t cs) {
  if (!src || !dst) return fail(DCP_ERR_INVALID_ARG, "null image pointer");
  if (H <= 0 || W <= 0) return fail(DCP_ERR_INVALID_ARG, "image must be non-empty (got %lld x %lld)", (long long)H, (long long)W);
  if (cs < 1 || rs < 1) return fail(DCP_ERR_INVALID_ARG, "strides must be positive (row %lld, col %lld)", (long long)rs, (long long)cs);
  if (rs < (W - 1) * cs + 1 && H > 1) return fail(DCP_ERR_INVALID_ARG, "row stride %lld overlaps rows of width %lld", (long long)rs, (long long)W);
  // the gather addresses the source with 32-bit byte offsets
  const double extent = ((double)(H - 1) * (double)rs + (double)(W - 1) * (double)cs + 1.0) * 4.0;
  if (extent > 4294967040.0 || H > 2147483647LL / 2 || W > 2147483647LL / 2)
    return fail(DCP_ERR_UNSUPPORTED, "source extent %.0f bytes exceeds the 4 GiB the 32-bit gather offsets address", extent);
  return DCP_OK;
}

int fill_map(dcp::MapArgs* m, double xc, double yc, const double* fact, int nfact, const double* coef) {
  memset(m, 0, sizeof(*m));
  m->xc = xc;
  m->yc = yc;
  if (nfact < 0 || nfact > dcp::kMaxFact)
    return fail(DCP_ERR_INVALID_ARG, "nfact = %d outside [0, %d]", nfact, dcp::kMaxFact);
  if (nfact > 0 && !fact) return fail(DCP_ERR_INVALID_ARG, "null coefficient pointer");
  for (int i = 0; i < nfact; ++i) m->fact[i] = fact[i];
  m->nfact = nfact;
  if (coef)
    for (int i = 0; i < 8; ++i) m->coef[i] = coef[i];
  return DCP_OK;
}

// The shared-reciprocal division of the perspective kernels (div2_rn) is exact while nothing
// leaves the normal range: every coefficient is 0 or of moderate magnitude, and the denominator
// c7*x + c8*y + 1 keeps one sign and a moderate magnitude over the whole image (checked at the four
// corners; it is affine in x and y).
int homography_is_tame(const double* c, int64_t H, int64_t W) {
  for (int i = 0; i < 8; ++i) {
    const double a = std::fabs(c[i]);
    if (!(a == 0.0 || (a > 1e-100 && a < 1e100))) return 0;
  }
  const double xs[2] = {0.0, (double)(W - 1)}, ys[2] = {0.0, (double)(H - 1)};
  double lo = 1e300, hi = -1e300;
  for (double x : xs)
    for (double y : ys) {
      const double d = (c[6] * x + c[7] * y) + 1.0;
      lo = d < lo ? d : lo;
      hi = d > hi ? d : hi;
    }
  if (!(lo > 0.0 || hi < 0.0)) return 0;
  const double m = std::fabs(lo) < std::fabs(hi) ? std::fabs(lo) : std::fabs(hi);
  const double M = std::fabs(lo) > std::fabs(hi) ? std::fabs(lo) : std::fabs(hi);
  return m > 1e-6 && M < 1e6;
}

uint32_t extent_bytes(int64_t H, int64_t W, int64_t rs, int64_t cs) {
  return (uint32_t)(((H - 1) * rs + (W - 1) * cs + 1) * 4);
}

// Rows of a projection that the radial map of output rows row_start .. row_start+nrows-1 can
// touch: [*b0, *b1).  yd = yc + yu * B(r) is bilinear in (yu, B), so its range over the rows is spanned
// by the corners of [yu_min, yu_max] x [B_min, B_max], with B's range taken over every radius the rows
// reach.  B is sampled every 1/4 pixel of radius (a few thousand evaluations, instead of one per output
// pixel) and the range is widened by twice the largest step between neighbouring samples; the hull is
// then grown by a safety row on each side.  A non-finite model gets the whole projection.
void host_row_band(const dcp::MapArgs& m, int64_t H, int64_t W, double row_start, int64_t nrows, int64_t* b0,
                   int64_t* b1) {
  *b0 = 0;
  *b1 = H;
  const int n = m.nfact, ne = (n + 1) / 2, no = n / 2;
  auto B = [&](double ru) {
    if (n <= 0) return 0.0;
    const double r2 = ru * ru;
    double E = m.fact[2 * (ne - 1)];
    for (int k = ne - 2; k >= 0; --k) E = E * r2 + m.fact[2 * k];
    if (no == 0) return E;
    double O = m.fact[2 * (no - 1) + 1];
    for (int k = no - 2; k >= 0; --k) O = O * r2 + m.fact[2 * k + 1];
    return ru * O + E;
  };
  const double yu0 = row_start - m.yc, yu1 = (row_start + (double)(nrows - 1)) - m.yc;
  const double ya = std::fmin(std::fabs(yu0), std::fabs(yu1));
  const double ay_min = (yu0 <= 0.0 && yu1 >= 0.0) ? 0.0 : ya;                    // smallest |yu| over the rows
  const double ay_max = std::fmax(std::fabs(yu0), std::fabs(yu1));
  const double xl = 0.0 - m.xc, xr = (double)(W - 1) - m.xc;
  const double ax_min = (xl <= 0.0 && xr >= 0.0) ? 0.0 : std::fmin(std::fabs(xl), std::fabs(xr));
  const double ax_max = std::fmax(std::fabs(xl), std::fabs(xr));
  const double rlo = std::sqrt(ax_min * ax_min + ay_min * ay_min), rhi = std::sqrt(ax_max * ax_max + ay_max * ay_max);
  if (!std::isfinite(rlo) || !std::isfinite(rhi) || rhi > 1e9) return;
  const int64_t ns = (int64_t)std::ceil((rhi - rlo) * 4.0) + 1;
  double bmin = 1e300, bmax = -1e300, step = 0.0, prev = 0.0;
  for (int64_t i = 0; i <= ns; ++i) {
    const double r = i == ns ? rhi : rlo + 0.25 * (double)i;
    const double v = B(r < rhi ? r : rhi);
    if (!std::isfinite(v)) return;
    if (i > 0) step = std::fmax(step, std::fabs(v - prev));
    prev = v;
    bmin = std::fmin(bmin, v);
    bmax = std::fmax(bmax, v);
  }
  bmin -= 2.0 * step;
  bmax += 2.0 * step;
  double ymin = 1e300, ymax = -1e300;
  for (double yu : {yu0, yu1})
    for (double bv : {bmin, bmax}) {
      double yd = m.yc + yu * bv;
      if (!(yd >= 0.0)) yd = 0.0;               // also catches NaN
      if (yd > (double)(H - 1)) yd = (double)(H - 1);
      ymin = std::fmin(ymin, yd);
      ymax = std::fmax(ymax, yd);
    }
  int64_t lo = (int64_t)std::floor(ymin) - 1, hi = (int64_t)std::floor(ymax) + 3;
  if (lo < 0) lo = 0;
  if (hi > H) hi = H;
  if (hi - lo < 2) {  // the gather needs two rows
    lo = lo > 0 ? lo - 1 : lo;
    hi = lo + 2 <= H ? (hi > lo + 2 ? hi : lo + 2) : H;
  }
  *b0 = lo;
  *b1 = hi;
}

// Shared driver of the three whole-image entry points.
int run_image(dcp::MapKind kind, const float* src, float* dst, int64_t H, int64_t W, int64_t rs, int64_t cs,
              const dcp::MapArgs& map, int sampler, bool round_f32, int mem_kind, int device, void* stream) {
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  dcp::ImageArgs img;
  memset(&img, 0, sizeof(img));
  img.H = (int32_t)H;
  img.W = (int32_t)W;
  const dcp::LaunchOpts opts = current_opts();
  if (mem_kind == DCP_MEM_DEVICE) {
    img.src = src;
    img.dst = dst;
    img.src_stride = (int32_t)rs;
    img.src_col_stride = (int32_t)cs;
    img.src_bytes = extent_bytes(H, W, rs, cs);
    DCP_HIP(dcp::launch_image(kind, img, map, sampler, round_f32, opts, (hipStream_t)stream));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  // host memory: pack rows densely on the way in, run on the stream, copy back, synchronise
  hipStream_t st = (hipStream_t)stream;
  void *dsrc = nullptr, *ddst = nullptr;
  const size_t frame = (size_t)H * (size_t)W * sizeof(float);
  DCP_HIP(g_staging.get(0, frame, &dsrc));
  DCP_HIP(g_staging.get(1, frame, &ddst));
  if (cs == 1 && rs == W) {
    DCP_HIP(hipMemcpyAsync(dsrc, src, frame, hipMemcpyHostToDevice, st));
    img.src_stride = (int32_t)W;
    img.src_col_stride = 1;
    img.src_bytes = (uint32_t)frame;
  } else if (cs == 1) {
    DCP_HIP(hipMemcpy2DAsync(dsrc, (size_t)W * 4, src, (size_t)rs * 4, (size_t)W * 4, (size_t)H, hipMemcpyHostToDevice, st));
    img.src_stride = (int32_t)W;
    img.src_col_stride = 1;
    img.src_bytes = (uint32_t)frame;
  } else {
    // column-strided host view (e.g. one channel of an interleaved HxWxC image): ship the
    // enclosing extent and let the kernel's strided gather pick the channel
    const size_t ext = extent_bytes(H, W, rs, cs);
    DCP_HIP(g_staging.get(0, ext, &dsrc));
    DCP_HIP(hipMemcpyAsync(dsrc, src, ext, hipMemcpyHostToDevice, st));
    img.src_stride = (int32_t)rs;
    img.src_col_stride = (int32_t)cs;
    img.src_bytes = (uint32_t)ext;
  }
  img.src = (const float*)dsrc;
  img.dst = (float*)ddst;
  DCP_HIP(dcp::launch_image(kind, img, map, sampler, round_f32, opts, st));
  DCP_HIP(hipMemcpyAsync(dst, ddst, frame, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

}  // namespace

extern "C" {

int dcp_version(void) { return 100; }

int dcp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* dcp_last_error(void) { return g_err; }

int dcp_set_option(const char* key, int value) {
  if (!key) return fail(DCP_ERR_INVALID_ARG, "null option key");
  if (!strcmp(key, "tile_rows")) {
    if (value < 1 || value > dcp::kMaxTileRows) return fail(DCP_ERR_INVALID_ARG, "tile_rows must be in [1, %d]", dcp::kMaxTileRows);
    g_tile_rows = value;
  } else if (!strcmp(key, "xcd_remap")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "xcd_remap must be 0, 1 or 2");
    g_xcd_remap = value;
  } else if (!strcmp(key, "coef_lds")) {
    g_coef_lds = value ? 1 : 0;
  } else if (!strcmp(key, "lds_gather")) {
    g_lds_gather = value ? 1 : 0;
  } else if (!strcmp(key, "pipe_depth")) {
    if (value != 1 && value != 2 && value != 4) return fail(DCP_ERR_INVALID_ARG, "pipe_depth must be 1, 2 or 4");
    g_pipe_depth = value;
  } else if (!strcmp(key, "d_chunk")) {
    if (value < 1) return fail(DCP_ERR_INVALID_ARG, "d_chunk must be >= 1");
    g_d_chunk = value;
  } else if (!strcmp(key, "stack_chunk_kb")) {
    if (value < 1) return fail(DCP_ERR_INVALID_ARG, "stack_chunk_kb must be >= 1");
    g_stack_chunk_kb = value;
  } else {
    return fail(DCP_ERR_INVALID_ARG, "unknown option '%s'", key);
  }
  return DCP_OK;
}

int dcp_get_option(const char* key, int* value) {
  if (!key || !value) return fail(DCP_ERR_INVALID_ARG, "null argument");
  if (!strcmp(key, "tile_rows")) *value = g_tile_rows;
  else if (!strcmp(key, "xcd_remap")) *value = g_xcd_remap;
  else if (!strcmp(key, "coef_lds")) *value = g_coef_lds;
  else if (!strcmp(key, "d_chunk")) *value = g_d_chunk;
  else if (!strcmp(key, "pipe_depth")) *value = g_pipe_depth;
  else if (!strcmp(key, "lds_gather")) *value = g_lds_gather;
  else if (!strcmp(key, "stack_chunk_kb")) *value = g_stack_chunk_kb;
  else return fail(DCP_ERR_INVALID_ARG, "unknown option '%s'", key);
  return DCP_OK;
}

int dcp_unwarp_image_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                         int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                         int nfact, int order, int coord_round_f32, int blend_mode, int mem_kind, int device,
                         void* stream) {
  int rc, sampler;
  if ((rc = check_image(src, dst, height, width, src_row_stride, src_col_stride)) != DCP_OK) return rc;
  if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  return run_image(dcp::kRadial, src, dst, height, width, src_row_stride, src_col_stride, map, sampler,
                   coord_round_f32 != 0, mem_kind, device, stream);
}

int dcp_perspective_image_f32(const float* src, float* dst, int64_t height, int64_t width,
                              int64_t src_row_stride, int64_t src_col_stride, const double* list_coef,
                              int order, int blend_mode, int mem_kind, int device, void* stream) {
  int rc, sampler;
  if ((rc = check_image(src, dst, height, width, src_row_stride, src_col_stride)) != DCP_OK) return rc;
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
  dcp::MapArgs map;
  if ((rc = fill_map(&map, 0.0, 0.0, nullptr, 0, list_coef)) != DCP_OK) return rc;
  map.fast_div = homography_is_tame(list_coef, height, width);
  return run_image(dcp::kPersp, src, dst, height, width, src_row_stride, src_col_stride, map, sampler, true,
                   mem_kind, device, stream);
}

int dcp_unwarp_fused_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                         int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                         int nfact, const double* list_coef, int order, int blend_mode, int mem_kind,
                         int device, void* stream) {
  int rc, sampler;
  if ((rc = check_image(src, dst, height, width, src_row_stride, src_col_stride)) != DCP_OK) return rc;
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, list_coef)) != DCP_OK) return rc;
  map.fast_div = homography_is_tame(list_coef, height, width);
  return run_image(dcp::kFused, src, dst, height, width, src_row_stride, src_col_stride, map, sampler, true,
                   mem_kind, device, stream);
}

int dcp_remap_coords_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                         int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype,
                         int64_t npts, int order, int blend_mode, int mem_kind, int device, void* stream) {
  int rc, sampler;
  if ((rc = check_image(src, dst, height, width, src_row_stride, src_col_stride)) != DCP_OK) return rc;
  if (npts < 0) return fail(DCP_ERR_INVALID_ARG, "npts < 0");
  if (npts > 0 && (!ycoord || !xcoord)) return fail(DCP_ERR_INVALID_ARG, "null coordinate pointer");
  if (coord_dtype != DCP_COORD_F32 && coord_dtype != DCP_COORD_F64)
    return fail(DCP_ERR_INVALID_ARG, "unknown coord_dtype %d", coord_dtype);
  if (npts > 2147483647LL * 256) return fail(DCP_ERR_UNSUPPORTED, "too many points");
  if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
  if (npts == 0) return DCP_OK;
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  dcp::ImageArgs img;
  memset(&img, 0, sizeof(img));
  img.H = (int32_t)height;
  img.W = (int32_t)width;
  img.src_stride = (int32_t)src_row_stride;
  img.src_col_stride = (int32_t)src_col_stride;
  img.src_bytes = extent_bytes(height, width, src_row_stride, src_col_stride);
  dcp::CoordArgs ca;
  ca.npts = npts;
  ca.is_f64 = coord_dtype == DCP_COORD_F64;
  hipStream_t st = (hipStream_t)stream;
  if (mem_kind == DCP_MEM_DEVICE) {
    img.src = src;
    img.dst = dst;
    ca.ycoord = ycoord;
    ca.xcoord = xcoord;
    DCP_HIP(dcp::launch_coords(img, ca, sampler, st));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  void *dsrc, *ddst, *dy, *dx;
  const size_t csz = (size_t)npts * (ca.is_f64 ? 8 : 4);
  DCP_HIP(g_staging.get(0, img.src_bytes, &dsrc));
  DCP_HIP(g_staging.get(1, (size_t)npts * 4, &ddst));
  DCP_HIP(g_staging.get(2, csz, &dy));
  DCP_HIP(g_staging.get(3, csz, &dx));
  DCP_HIP(hipMemcpyAsync(dsrc, src, img.src_bytes, hipMemcpyHostToDevice, st));
  DCP_HIP(hipMemcpyAsync(dy, ycoord, csz, hipMemcpyHostToDevice, st));
  DCP_HIP(hipMemcpyAsync(dx, xcoord, csz, hipMemcpyHostToDevice, st));
  img.src = (const float*)dsrc;
  img.dst = (float*)ddst;
  ca.ycoord = dy;
  ca.xcoord = dx;
  DCP_HIP(dcp::launch_coords(img, ca, sampler, st));
  DCP_HIP(hipMemcpyAsync(dst, ddst, (size_t)npts * 4, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

}  // extern "C"

namespace {

// One description of every stack call: rows row_start .. row_start+nrows-1 of the corrected stack from
// `vol`, which holds rows [band_start, band_start + band_rows) of each of `depth` projections
// (band_start = 0, band_rows = height for a whole stack).
struct StackCall {
  const void* vol;
  void* out;
  int dtype, out_f32;
  int64_t depth, height, width, band_start, band_rows, proj_stride, row_stride;
  dcp::MapArgs map;
  double row_start;
  int64_t nrows;
  int round_f32, sampler;      // sampler: float32 data only (the other element types use scipy's exact blend)
  int mem_kind, device;
  void* stream;
};

// base = address of (projection 0, row 0) -- possibly before the buffer when the band starts later; the
// kernels only touch rows inside the band
hipError_t launch_stack_any(const StackCall& c, const void* base, void* out, int64_t n, int64_t proj_stride,
                            int64_t row_stride, int64_t rows_end, const dcp::LaunchOpts& opts, hipStream_t hs) {
  if (c.dtype == dcp::kF32 && !c.out_f32) {
    dcp::StackArgs st;
    memset(&st, 0, sizeof(st));
    st.D = (int32_t)n;
    st.H = (int32_t)c.height;
    st.W = (int32_t)c.width;
    st.row_start = c.row_start;
    st.nrows = (int32_t)c.nrows;
    st.vol = (const float*)base;
    st.out = (float*)out;
    st.proj_stride = proj_stride;
    st.row_stride = (int32_t)row_stride;
    st.proj_bytes = (uint32_t)(((rows_end - 1) * row_stride + c.width) * 4);
    return dcp::launch_stack(st, c.map, c.sampler, c.round_f32 != 0, opts, hs);
  }
  dcp::TypedStackArgs st;
  memset(&st, 0, sizeof(st));
  st.D = (int32_t)n;
  st.H = (int32_t)c.height;
  st.W = (int32_t)c.width;
  st.row_start = c.row_start;
  st.nrows = (int32_t)c.nrows;
  st.d_chunk = opts.d_chunk;
  st.dtype = c.dtype;
  st.out_f32 = c.out_f32;
  st.round_f32 = c.round_f32 != 0;
  st.vol = base;
  st.out = out;
  st.proj_stride = proj_stride;
  st.row_stride = row_stride;
  return dcp::launch_typed_stack(st, c.map, hs);
}

int run_stack(const StackCall& c) {
  const int64_t depth = c.depth, height = c.height, width = c.width, nrows = c.nrows;
  if (c.dtype < 0 || c.dtype >= dcp::kNumElemTypes) return fail(DCP_ERR_INVALID_ARG, "unknown element type %d", c.dtype);
  if (depth < 0 || nrows < 0) return fail(DCP_ERR_INVALID_ARG, "negative depth / nrows");
  if (height <= 0 || width <= 0) return fail(DCP_ERR_INVALID_ARG, "projections must be non-empty");
  if (depth > 0 && nrows > 0 && (!c.vol || !c.out)) return fail(DCP_ERR_INVALID_ARG, "null volume pointer");
  if (c.band_start < 0 || c.band_rows < 1 || c.band_start + c.band_rows > height)
    return fail(DCP_ERR_INVALID_ARG, "band rows [%lld, %lld) outside the projection height %lld", (long long)c.band_start,
                (long long)(c.band_start + c.band_rows), (long long)height);
  if (c.row_stride < width || c.proj_stride < (c.band_rows - 1) * c.row_stride + width)
    return fail(DCP_ERR_INVALID_ARG, "strides overlap (row %lld, projection %lld)", (long long)c.row_stride, (long long)c.proj_stride);
  const bool fast = c.dtype == dcp::kF32 && !c.out_f32;
  if (fast) {
    if (height < 2 || width < 2) return fail(DCP_ERR_UNSUPPORTED, "stack path needs projections of at least 2 x 2");
    if ((double)height * (double)c.row_stride * 4.0 > 4294967040.0)
      return fail(DCP_ERR_UNSUPPORTED, "one projection exceeds the 4 GiB the 32-bit gather offsets address");
  }
  if (height > 1073741823LL || width > 1073741823LL || depth > 2147483647LL) return fail(DCP_ERR_UNSUPPORTED, "stack too large");
  if (nrows > 65535) return fail(DCP_ERR_UNSUPPORTED, "nrows > 65535 in one call");
  if (!std::isfinite(c.row_start)) return fail(DCP_ERR_INVALID_ARG, "row_start is not finite");
  // rows of a projection the requested rows can reach (the reference slices mat3D[i, yd_min:yd_max, :] for
  // the same reason, postprocessing.py:221-228)
  int64_t band0 = 0, band1 = height;
  host_row_band(c.map, height, width, c.row_start, nrows, &band0, &band1);
  const bool partial = c.band_start != 0 || c.band_rows != height;
  if (partial && (band0 < c.band_start || band1 > c.band_start + c.band_rows))
    return fail(DCP_ERR_INVALID_ARG, "the rows need source rows [%lld, %lld) but the band holds [%lld, %lld) (see dcp_stack_row_band)",
                (long long)band0, (long long)band1, (long long)c.band_start, (long long)(c.band_start + c.band_rows));
  if (depth == 0 || nrows == 0) return DCP_OK;
  DeviceScope scope(c.device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", c.device, hipGetErrorString(scope.status));
  dcp::LaunchOpts opts = current_opts();
  if ((depth + opts.d_chunk - 1) / opts.d_chunk > 65535) opts.d_chunk = (int)((depth + 65534) / 65535);
  const size_t esz = (size_t)dcp::elem_size(c.dtype), osz = c.out_f32 ? 4 : esz;
  hipStream_t hs = (hipStream_t)c.stream;
  if (c.mem_kind == DCP_MEM_DEVICE) {
    const char* base = (const char*)c.vol - (size_t)(c.band_start * c.row_stride) * esz;
    DCP_HIP(launch_stack_any(c, base, c.out, depth, c.proj_stride, c.row_stride, c.band_start + c.band_rows, opts, hs));
    return DCP_OK;
  }
  if (c.mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", c.mem_kind);
  // Host stack.  Only the reachable row band is shipped, and projections are independent
  // (postprocessing.py:226-228, 310-312), so the stack streams through the GPU in depth chunks: while
  // chunk k is copied back by a second host thread, chunk k+1 is uploaded and computed (PCIe is full
  // duplex; pageable copies block their calling thread, hence two threads rather than two streams --
  // tools/ubench_pcie.hip).  Device scratch: two band buffers and two output buffers of one chunk each.
  const int64_t bh = band1 - band0;
  const size_t pbytes = (size_t)bh * (size_t)width * esz;               // one projection's band
  const size_t obytes = (size_t)nrows * (size_t)width * osz;            // one projection's output rows
  int64_t dc = (int64_t)(((size_t)g_stack_chunk_kb.load() << 10) / (pbytes > obytes ? pbytes : obytes));
  dc = dc < 1 ? 1 : (dc > depth ? depth : dc);
  const int64_t nchunks = (depth + dc - 1) / dc;
  void *din[2], *dout[2];
  for (int b = 0; b < 2; ++b) {
    DCP_HIP(g_staging.get(b, pbytes * (size_t)dc, &din[b]));
    DCP_HIP(g_staging.get(2 + b, obytes * (size_t)dc, &dout[b]));
  }
  int cur_dev = 0;
  DCP_HIP(hipGetDevice(&cur_dev));
  hipStream_t s_down = nullptr;
  DCP_HIP(g_host_streams.get(&hs, &s_down));   // host memory: nothing to order against the caller's stream

  const bool trace = getenv("DISCORPY_AMD_TRACE") != nullptr;   // per-chunk timeline on stderr
  const auto t_begin = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  std::mutex mu;
  std::condition_variable cv;
  int64_t computed = 0, downloaded = 0;     // chunks whose kernel has finished / whose D2H has finished
  hipError_t down_err = hipSuccess;
  bool abort_down = false;
  std::thread downloader([&]() {
    hipError_t e = hipSetDevice(cur_dev);
    for (int64_t k = 0; k < nchunks && e == hipSuccess; ++k) {
      {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return computed > k || abort_down; });
        if (abort_down) break;
      }
      const int64_t d0 = k * dc, n = (d0 + dc > depth ? depth - d0 : dc);
      const double td0 = ms();
      e = hipMemcpyAsync((char*)c.out + (size_t)d0 * obytes, dout[k & 1], obytes * (size_t)n, hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess) e = hipStreamSynchronize(s_down);
      if (trace) fprintf(stderr, "down %lld: %.3f -> %.3f\n", (long long)k, td0, ms());
      {
        std::lock_guard<std::mutex> lock(mu);
        downloaded = k + 1;
      }
      cv.notify_all();
    }
    std::lock_guard<std::mutex> lock(mu);
    down_err = e;
    downloaded = nchunks;      // never leave the uploader waiting
    cv.notify_all();
  });
  hipError_t up_err = hipSuccess;
  for (int64_t k = 0; k < nchunks && up_err == hipSuccess; ++k) {
    const int64_t d0 = k * dc, n = (d0 + dc > depth ? depth - d0 : dc);
    if (k >= 2) {   // output buffer k & 1 is free once chunk k-2 has been copied back
      std::unique_lock<std::mutex> lock(mu);
      cv.wait(lock, [&] { return downloaded >= k - 1; });
    }
    const char* hsrc = (const char*)c.vol + (size_t)(d0 * c.proj_stride + (band0 - c.band_start) * c.row_stride) * esz;
    const double tu0 = ms();
    if (c.row_stride == width) {   // the bands of n projections: n runs of pbytes, proj_stride apart
      up_err = hipMemcpy2DAsync(din[k & 1], pbytes, hsrc, (size_t)c.proj_stride * esz, pbytes, (size_t)n, hipMemcpyHostToDevice, hs);
    } else {
      for (int64_t d = 0; d < n && up_err == hipSuccess; ++d)
        up_err = hipMemcpy2DAsync((char*)din[k & 1] + (size_t)d * pbytes, (size_t)width * esz,
                                  hsrc + (size_t)(d * c.proj_stride) * esz, (size_t)c.row_stride * esz, (size_t)width * esz,
                                  (size_t)bh, hipMemcpyHostToDevice, hs);
    }
    if (up_err != hipSuccess) break;
    const double tu1 = ms();
    // absolute row indexing: the staged band starts at row band0 (never dereferenced below it)
    const char* base = (const char*)din[k & 1] - (size_t)(band0 * width) * esz;
    up_err = launch_stack_any(c, base, dout[k & 1], n, bh * width, width, band1, opts, hs);
    if (up_err == hipSuccess) up_err = hipStreamSynchronize(hs);
    if (trace) fprintf(stderr, "up %lld: issue %.3f -> %.3f, done %.3f\n", (long long)k, tu0, tu1, ms());
    if (up_err != hipSuccess) break;
    {
      std::lock_guard<std::mutex> lock(mu);
      computed = k + 1;
    }
    cv.notify_all();
  }
  if (up_err != hipSuccess) {
    std::lock_guard<std::mutex> lock(mu);
    abort_down = true;
    cv.notify_all();
  }
  downloader.join();
  if (up_err != hipSuccess) return fail(DCP_ERR_HIP, "stack upload / kernel failed: %s", hipGetErrorString(up_err));
  if (down_err != hipSuccess) return fail(DCP_ERR_HIP, "stack download failed: %s", hipGetErrorString(down_err));
  return DCP_OK;
}

int make_stack_call(StackCall* c, const void* vol, void* out, int dtype, int out_f32, int64_t depth, int64_t height,
                    int64_t width, int64_t band_start, int64_t band_rows, int64_t proj_stride, int64_t row_stride,
                    double xcenter, double ycenter, const double* list_fact, int nfact, double row_start, int64_t nrows,
                    int coord_round_f32, int blend_mode, int mem_kind, int device, void* stream) {
  int rc;
  c->vol = vol;
  c->out = out;
  c->dtype = dtype;
  c->out_f32 = out_f32 != 0;
  c->depth = depth;
  c->height = height;
  c->width = width;
  c->band_start = band_start;
  c->band_rows = band_rows;
  c->proj_stride = proj_stride;
  c->row_stride = row_stride;
  c->row_start = row_start;
  c->nrows = nrows;
  c->round_f32 = coord_round_f32;
  c->mem_kind = mem_kind;
  c->device = device;
  c->stream = stream;
  c->sampler = dcp::kScipy;
  if (dtype == dcp::kF32 && !out_f32 && (rc = sampler_of(1, blend_mode, &c->sampler)) != DCP_OK) return rc;
  return fill_map(&c->map, xcenter, ycenter, list_fact, nfact, nullptr);
}

}  // namespace

extern "C" {

int dcp_unwarp_stack_rows_f32(const float* vol, float* out, int64_t depth, int64_t height, int64_t width,
                              int64_t proj_stride, int64_t row_stride, double xcenter, double ycenter,
                              const double* list_fact, int nfact, double row_start, int64_t nrows,
                              int coord_round_f32, int blend_mode, int mem_kind, int device, void* stream) {
  int rc;
  StackCall c;
  if ((rc = make_stack_call(&c, vol, out, dcp::kF32, 0, depth, height, width, 0, height, proj_stride, row_stride, xcenter,
                            ycenter, list_fact, nfact, row_start, nrows, coord_round_f32, blend_mode, mem_kind, device,
                            stream)) != DCP_OK)
    return rc;
  return run_stack(c);
}

int dcp_unwarp_stack_rows_typed(const void* vol, void* out, int dtype, int out_float32, int64_t depth, int64_t height,
                                int64_t width, int64_t proj_stride, int64_t row_stride, double xcenter, double ycenter,
                                const double* list_fact, int nfact, double row_start, int64_t nrows, int coord_round_f32,
                                int mem_kind, int device, void* stream) {
  int rc;
  StackCall c;
  if ((rc = make_stack_call(&c, vol, out, dtype, out_float32, depth, height, width, 0, height, proj_stride, row_stride,
                            xcenter, ycenter, list_fact, nfact, row_start, nrows, coord_round_f32, DCP_BLEND_SCIPY, mem_kind,
                            device, stream)) != DCP_OK)
    return rc;
  return run_stack(c);
}

int dcp_unwarp_image_channels(const void* src, void* dst, int dtype, int64_t height, int64_t width, int channels,
                              int64_t src_row_stride, int64_t src_pixel_stride, double xcenter, double ycenter,
                              const double* list_fact, int nfact, int order, int mem_kind, int device, void* stream) {
  int rc;
  if (channels < 1 || channels > 64) return fail(DCP_ERR_INVALID_ARG, "channels = %d outside [1, 64]", channels);
  if (order < 0 || order > 1) return fail(DCP_ERR_UNSUPPORTED, "the interleaved-channel kernel takes orders 0 and 1 (got %d)", order);
  if (src_pixel_stride < channels) return fail(DCP_ERR_INVALID_ARG, "pixel stride %lld smaller than %d channels", (long long)src_pixel_stride, channels);
  if ((rc = check_image_typed(src, dst, dtype, height, width, src_row_stride, src_pixel_stride)) != DCP_OK) return rc;
  if (src_row_stride < (width - 1) * src_pixel_stride + channels && height > 1)
    return fail(DCP_ERR_INVALID_ARG, "row stride %lld overlaps rows of %lld pixels", (long long)src_row_stride, (long long)width);
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipStream_t st = (hipStream_t)stream;
  dcp::TypedImageArgs a;
  memset(&a, 0, sizeof(a));
  a.H = (int32_t)height;
  a.W = (int32_t)width;
  a.src_stride = src_row_stride;
  a.src_cstride = src_pixel_stride;
  a.order = order;
  a.dtype = dtype;
  if (mem_kind == DCP_MEM_DEVICE) {
    a.src = src;
    a.dst = dst;
    DCP_HIP(dcp::launch_typed_channels(a, map, channels, st));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  const size_t esz = (size_t)dcp::elem_size(dtype);
  const size_t ext = (size_t)((height - 1) * src_row_stride + (width - 1) * src_pixel_stride + channels) * esz;
  const size_t obytes = (size_t)height * (size_t)width * (size_t)channels * esz;
  void *dsrc, *ddst;
  DCP_HIP(g_staging.get(0, ext, &dsrc));
  DCP_HIP(g_staging.get(1, obytes, &ddst));
  DCP_HIP(hipMemcpyAsync(dsrc, src, ext, hipMemcpyHostToDevice, st));
  a.src = dsrc;
  a.dst = ddst;
  DCP_HIP(dcp::launch_typed_channels(a, map, channels, st));
  DCP_HIP(hipMemcpyAsync(dst, ddst, obytes, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

int dcp_stack_row_band(int64_t height, int64_t width, double xcenter, double ycenter, const double* list_fact, int nfact,
                       double row_start, int64_t nrows, int64_t* band_start, int64_t* band_rows) {
  int rc;
  if (!band_start || !band_rows) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  if (height <= 0 || width <= 0 || nrows < 1) return fail(DCP_ERR_INVALID_ARG, "empty projection or no rows");
  if (!std::isfinite(row_start)) return fail(DCP_ERR_INVALID_ARG, "row_start is not finite");
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  int64_t b0 = 0, b1 = height;
  host_row_band(map, height, width, row_start, nrows, &b0, &b1);
  *band_start = b0;
  *band_rows = b1 - b0;
  return DCP_OK;
}

int dcp_unwarp_stack_band(const void* band, void* out, int dtype, int out_float32, int64_t depth, int64_t height,
                          int64_t width, int64_t band_start, int64_t band_rows, int64_t proj_stride, int64_t row_stride,
                          double xcenter, double ycenter, const double* list_fact, int nfact, double row_start,
                          int64_t nrows, int coord_round_f32, int blend_mode, int mem_kind, int device, void* stream) {
  int rc;
  StackCall c;
  if ((rc = make_stack_call(&c, band, out, dtype, out_float32, depth, height, width, band_start, band_rows, proj_stride,
                            row_stride, xcenter, ycenter, list_fact, nfact, row_start, nrows, coord_round_f32, blend_mode,
                            mem_kind, device, stream)) != DCP_OK)
    return rc;
  return run_stack(c);
}

int dcp_unwarp_stack_rows_multi_f32(const float* vol, float* out, int64_t depth, int64_t height, int64_t width,
                                    int64_t proj_stride, int64_t row_stride, double xcenter, double ycenter,
                                    const double* list_fact, int nfact, double row_start, int64_t nrows,
                                    int coord_round_f32, int blend_mode, const int* devices, int ndev) {
  if (ndev < 1 || !devices) return fail(DCP_ERR_INVALID_ARG, "need at least one device");
  if (ndev > 64) return fail(DCP_ERR_INVALID_ARG, "ndev = %d > 64", ndev);
  const int have = dcp_device_count();
  for (int i = 0; i < ndev; ++i)
    if (devices[i] < 0 || devices[i] >= have)
      return have == 0 ? fail(DCP_ERR_NO_DEVICE, "no HIP device visible")
                       : fail(DCP_ERR_INVALID_ARG, "devices[%d] = %d outside [0, %d)", i, devices[i], have);
  if (depth < 0 || nrows < 0 || width <= 0) return fail(DCP_ERR_INVALID_ARG, "negative depth / nrows or empty projections");
  // shard i = projections [d0, d1) with the sizes of numpy.array_split(range(depth), ndev); every worker
  // stages its own shard, runs K4 on its device and copies its block of `out` back -- the blocks are
  // disjoint, contiguous along depth, so the "gather" is the D2H copies themselves
  std::vector<int> rcs((size_t)ndev, DCP_OK);
  std::vector<std::string> msgs((size_t)ndev);
  std::vector<std::thread> workers;
  const int64_t base = depth / ndev, extra = depth % ndev;
  int64_t d0 = 0;
  for (int i = 0; i < ndev; ++i) {
    const int64_t n = base + (i < extra ? 1 : 0);
    const float* v = vol ? vol + d0 * proj_stride : vol;
    float* o = out ? out + d0 * nrows * width : out;
    const int dev = devices[i];
    workers.emplace_back([=, &rcs, &msgs]() {
      rcs[(size_t)i] = dcp_unwarp_stack_rows_f32(v, o, n, height, width, proj_stride, row_stride, xcenter, ycenter, list_fact,
                                                 nfact, row_start, nrows, coord_round_f32, blend_mode, DCP_MEM_HOST, dev,
                                                 nullptr);
      if (rcs[(size_t)i] != DCP_OK) msgs[(size_t)i] = dcp_last_error();
      g_staging.release();   // the worker's scratch lives on `dev`; free it before the thread ends
      g_host_streams.release();
    });
    d0 += n;
  }
  for (auto& w : workers) w.join();
  for (int i = 0; i < ndev; ++i)
    if (rcs[(size_t)i] != DCP_OK) return fail(rcs[(size_t)i], "shard %d on device %d: %s", i, devices[i], msgs[(size_t)i].c_str());
  return DCP_OK;
}

}  // extern "C"

// ---- spline orders 2..5 -------------------------------------------------------------------

namespace {

// Per-device float64 coefficient workspace (grow-only).  Reused across calls; a call on a stream
// other than the previous one first waits for the device so that the old user is done.
struct SplineWorkspace {
  std::mutex mu;
  void* buf[64] = {};
  size_t cap[64] = {};
  hipStream_t last[64] = {};
  hipError_t get(size_t bytes, hipStream_t stream, double** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (buf[dev] && last[dev] != stream) {
      e = hipDeviceSynchronize();
      if (e != hipSuccess) return e;
    }
    if (cap[dev] < bytes) {
      if (buf[dev]) {
        e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
        (void)hipFree(buf[dev]);
        buf[dev] = nullptr;
        cap[dev] = 0;
      }
      e = hipMalloc(&buf[dev], bytes);
      if (e != hipSuccess) return e;
      cap[dev] = bytes;
    }
    last[dev] = stream;
    *out = (double*)buf[dev];
    return hipSuccess;
  }
};
SplineWorkspace g_spline_ws;

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

int check_image_typed(const void* src, const void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs) {
  if (dtype < 0 || dtype >= dcp::kNumElemTypes) return fail(DCP_ERR_INVALID_ARG, "unknown element type %d", dtype);
  if (!src || !dst) return fail(DCP_ERR_INVALID_ARG, "null image pointer");
  if (H <= 0 || W <= 0) return fail(DCP_ERR_INVALID_ARG, "image must be non-empty (got %lld x %lld)", (long long)H, (long long)W);
  if (cs < 1 || rs < 1) return fail(DCP_ERR_INVALID_ARG, "strides must be positive (row %lld, col %lld)", (long long)rs, (long long)cs);
  if (rs < (W - 1) * cs + 1 && H > 1) return fail(DCP_ERR_INVALID_ARG, "row stride %lld overlaps rows of width %lld", (long long)rs, (long long)W);
  if (H > 1073741823LL || W > 1073741823LL) return fail(DCP_ERR_UNSUPPORTED, "image too large");
  return DCP_OK;
}

size_t extent_bytes_typed(int64_t H, int64_t W, int64_t rs, int64_t cs, int dtype) {
  return (size_t)((H - 1) * rs + (W - 1) * cs + 1) * (size_t)dcp::elem_size(dtype);
}

int run_spline(int map_kind, const void* src, void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs,
               const dcp::MapArgs& map, const void* ycoord, const void* xcoord, int coord_dtype, int64_t npts, int order,
               int mode, int mem_kind, int device, void* stream) {
  int rc;
  if ((rc = check_image_typed(src, dst, dtype, H, W, rs, cs)) != DCP_OK) return rc;
  if (order < 2 || order > 5) return fail(DCP_ERR_INVALID_ARG, "spline order %d outside [2, 5]", order);
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
  a.pad = (mode == dcp::kModeNearest || mode == dcp::kModeGridConstant) ? 12 : 0;
  a.Hp = a.H + 2 * a.pad;
  a.Wp = a.W + 2 * a.pad;
  a.filter_kind = (mode == dcp::kModeReflect || mode == dcp::kModeGridMirror) ? dcp::kSplReflect
                  : mode == dcp::kModeGridWrap                                 ? dcp::kSplWrap
                                                                               : dcp::kSplMirror;
  a.npoles = spline_poles(order, a.poles);
  for (int axis = 0; axis < 2; ++axis) {
    const double n = axis == 0 ? (double)a.Hp : (double)a.Wp;
    for (int p = 0; p < a.npoles; ++p)
      a.zpow[axis][p] = std::pow(a.poles[p], a.filter_kind == dcp::kSplMirror ? n - 1.0 : n);
  }
  const size_t plane = (size_t)a.Hp * (size_t)a.Wp * sizeof(double);
  DCP_HIP(g_spline_ws.get(2 * plane, st, &a.coef));
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

}  // namespace

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

namespace {

// Orders 0..5 on any element type: 0/1 through typed_kernels.hip, 2..5 through the spline path.
// map_kind 0 radial, 1 perspective, 2 fused, 3 explicit coordinates.
int run_typed(int map_kind, const void* src, void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs,
              const dcp::MapArgs& map, const void* ycoord, const void* xcoord, int coord_dtype, int64_t npts, int order,
              int mode, int mem_kind, int device, void* stream) {
  int rc;
  if (order < 0 || order > 5) return fail(DCP_ERR_INVALID_ARG, "spline order %d outside [0, 5]", order);
  if (mode < 0 || mode > 7) return fail(DCP_ERR_INVALID_ARG, "unknown boundary mode %d", mode);
  if (order >= 2) {
    if (map_kind == 2) return fail(DCP_ERR_UNSUPPORTED, "the fused map is implemented for orders 0 and 1");
    return run_spline(map_kind == 3 ? 2 : map_kind, src, dst, dtype, H, W, rs, cs, map, ycoord, xcoord, coord_dtype, npts,
                      order, mode, mem_kind, device, stream);
  }
  if ((rc = check_image_typed(src, dst, dtype, H, W, rs, cs)) != DCP_OK) return rc;
  if (map_kind == 3) {
    if (npts < 0) return fail(DCP_ERR_INVALID_ARG, "npts < 0");
    if (npts > 0 && (!ycoord || !xcoord)) return fail(DCP_ERR_INVALID_ARG, "null coordinate pointer");
    if (coord_dtype != DCP_COORD_F32 && coord_dtype != DCP_COORD_F64) return fail(DCP_ERR_INVALID_ARG, "unknown coord_dtype %d", coord_dtype);
    if (npts > 2147483647LL * 256) return fail(DCP_ERR_UNSUPPORTED, "too many points");
    if (npts == 0) return DCP_OK;
  }
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipStream_t st = (hipStream_t)stream;
  dcp::TypedImageArgs a;
  memset(&a, 0, sizeof(a));
  a.H = (int32_t)H;
  a.W = (int32_t)W;
  a.src_stride = rs;
  a.src_cstride = cs;
  a.order = order;
  a.dtype = dtype;
  dcp::CoordArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.npts = npts;
  ca.is_f64 = coord_dtype == DCP_COORD_F64;
  const int64_t nout = map_kind == 3 ? npts : H * W;
  if (mem_kind == DCP_MEM_DEVICE) {
    a.src = src;
    a.dst = dst;
    ca.ycoord = ycoord;
    ca.xcoord = xcoord;
    DCP_HIP(dcp::launch_typed_image(map_kind, a, map, ca, st));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  void *dsrc, *ddst, *dy = nullptr, *dx = nullptr;
  const size_t ext = extent_bytes_typed(H, W, rs, cs, dtype), esz = (size_t)dcp::elem_size(dtype);
  DCP_HIP(g_staging.get(0, ext, &dsrc));
  DCP_HIP(g_staging.get(1, (size_t)nout * esz, &ddst));
  DCP_HIP(hipMemcpyAsync(dsrc, src, ext, hipMemcpyHostToDevice, st));
  if (map_kind == 3) {
    const size_t csz = (size_t)npts * (ca.is_f64 ? 8 : 4);
    DCP_HIP(g_staging.get(2, csz, &dy));
    DCP_HIP(g_staging.get(3, csz, &dx));
    DCP_HIP(hipMemcpyAsync(dy, ycoord, csz, hipMemcpyHostToDevice, st));
    DCP_HIP(hipMemcpyAsync(dx, xcoord, csz, hipMemcpyHostToDevice, st));
  }
  a.src = dsrc;
  a.dst = ddst;
  ca.ycoord = dy;
  ca.xcoord = dx;
  DCP_HIP(dcp::launch_typed_image(map_kind, a, map, ca, st));
  DCP_HIP(hipMemcpyAsync(dst, ddst, (size_t)nout * esz, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

}  // namespace

extern "C" {

int dcp_unwarp_image_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width, int64_t src_row_stride,
                           int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact, int nfact,
                           int order, int boundary_mode, int mem_kind, int device, void* stream) {
  int rc;
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  return run_typed(0, src, dst, dtype, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0, order,
                   boundary_mode, mem_kind, device, stream);
}

int dcp_perspective_image_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width,
                                int64_t src_row_stride, int64_t src_col_stride, const double* list_coef, int order,
                                int boundary_mode, int mem_kind, int device, void* stream) {
  int rc;
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  dcp::MapArgs map;
  if ((rc = fill_map(&map, 0.0, 0.0, nullptr, 0, list_coef)) != DCP_OK) return rc;
  if (height > 0 && width > 0) map.fast_div = homography_is_tame(list_coef, height, width);
  return run_typed(1, src, dst, dtype, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0, order,
                   boundary_mode, mem_kind, device, stream);
}

int dcp_unwarp_fused_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width, int64_t src_row_stride,
                           int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact, int nfact,
                           const double* list_coef, int order, int boundary_mode, int mem_kind, int device, void* stream) {
  int rc;
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, list_coef)) != DCP_OK) return rc;
  if (height > 0 && width > 0) map.fast_div = homography_is_tame(list_coef, height, width);
  return run_typed(2, src, dst, dtype, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0, order,
                   boundary_mode, mem_kind, device, stream);
}

int dcp_remap_coords_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width, int64_t src_row_stride,
                           int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype, int64_t npts,
                           int order, int boundary_mode, int mem_kind, int device, void* stream) {
  dcp::MapArgs map;
  memset(&map, 0, sizeof(map));
  return run_typed(3, src, dst, dtype, height, width, src_row_stride, src_col_stride, map, ycoord, xcoord, coord_dtype,
                   npts, order, boundary_mode, mem_kind, device, stream);
}

int dcp_coordinate_map_f32(float* ymap, float* xmap, int64_t height, int64_t width, int map_kind, double xcenter,
                           double ycenter, const double* list_fact, int nfact, const double* list_coef, int mem_kind,
                           int device, void* stream) {
  int rc;
  if (!ymap || !xmap) return fail(DCP_ERR_INVALID_ARG, "null map pointer");
  if (height <= 0 || width <= 0 || height > 1073741823LL || width > 1073741823LL)
    return fail(DCP_ERR_INVALID_ARG, "map must be non-empty (got %lld x %lld)", (long long)height, (long long)width);
  if (map_kind < DCP_MAP_RADIAL || map_kind > DCP_MAP_FUSED) return fail(DCP_ERR_INVALID_ARG, "unknown map_kind %d", map_kind);
  if (map_kind != DCP_MAP_RADIAL && !list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, map_kind == DCP_MAP_PERSPECTIVE ? nullptr : list_fact,
                     map_kind == DCP_MAP_PERSPECTIVE ? 0 : nfact, map_kind == DCP_MAP_RADIAL ? nullptr : list_coef)) != DCP_OK)
    return rc;
  if (map_kind != DCP_MAP_RADIAL) map.fast_div = homography_is_tame(list_coef, height, width);
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  dcp::ImageArgs img;
  memset(&img, 0, sizeof(img));
  img.H = (int32_t)height;
  img.W = (int32_t)width;
  const dcp::MapKind kind = map_kind == DCP_MAP_RADIAL ? dcp::kRadial : map_kind == DCP_MAP_PERSPECTIVE ? dcp::kPersp : dcp::kFused;
  hipStream_t st = (hipStream_t)stream;
  if (mem_kind == DCP_MEM_DEVICE) {
    DCP_HIP(dcp::launch_coord_map(kind, img, map, ymap, xmap, st));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  void *dy, *dx;
  const size_t plane = (size_t)height * (size_t)width * 4;
  DCP_HIP(g_staging.get(2, plane, &dy));
  DCP_HIP(g_staging.get(3, plane, &dx));
  DCP_HIP(dcp::launch_coord_map(kind, img, map, (float*)dy, (float*)dx, st));
  DCP_HIP(hipMemcpyAsync(ymap, dy, plane, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipMemcpyAsync(xmap, dx, plane, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

int dcp_debug_counters(uint64_t* out, int n, int reset) {
  if (!out || n < 2) return fail(DCP_ERR_INVALID_ARG, "need room for 2 counters");
  unsigned long long v[2];
  DCP_HIP(hipDeviceSynchronize());
  DCP_HIP(dcp::read_lds_stats(v, reset != 0));
  out[0] = v[0];
  out[1] = v[1];
  return DCP_OK;
}

int dcp_malloc(void** ptr, size_t bytes, int device) {
  if (!ptr) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  DCP_HIP(hipMalloc(ptr, bytes ? bytes : 4));
  return DCP_OK;
}

int dcp_free(void* ptr, int device) {
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  DCP_HIP(hipFree(ptr));
  return DCP_OK;
}

int dcp_memcpy(void* dst, const void* src, size_t bytes, int kind, int device, void* stream) {
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipMemcpyKind k;
  switch (kind) {
    case DCP_COPY_H2D: k = hipMemcpyHostToDevice; break;
    case DCP_COPY_D2H: k = hipMemcpyDeviceToHost; break;
    case DCP_COPY_D2D: k = hipMemcpyDeviceToDevice; break;
    default: return fail(DCP_ERR_INVALID_ARG, "unknown copy kind %d", kind);
  }
  DCP_HIP(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
  if (kind != DCP_COPY_D2D) DCP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return DCP_OK;
}

int dcp_stream_synchronize(int device, void* stream) {
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  DCP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return DCP_OK;
}

int dcp_event_create(void** event, int device) {
  if (!event) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipEvent_t e;
  DCP_HIP(hipEventCreate(&e));
  *event = (void*)e;
  return DCP_OK;
}

int dcp_event_record(void* event, void* stream) {
  DCP_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return DCP_OK;
}

int dcp_event_synchronize(void* event) {
  DCP_HIP(hipEventSynchronize((hipEvent_t)event));
  return DCP_OK;
}

int dcp_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  DCP_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return DCP_OK;
}

int dcp_event_destroy(void* event) {
  DCP_HIP(hipEventDestroy((hipEvent_t)event));
  return DCP_OK;
}

}  // extern "C"
