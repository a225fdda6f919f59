// api_core.cpp -- library / device / option / memory / stream / event entry points of the C ABI
// (include/discorpy_hip.h) and the helpers every other api_*.cpp uses (api_common.h).
#include "api_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace dcpapi {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

const char* last_error() { return g_err; }

std::atomic<int> g_tile_rows{16}, g_xcd_remap{2}, g_coef_lds{0}, g_d_chunk{16}, g_pipe_depth{2}, g_lds_gather{1}, g_stack_chunk_kb{24576}, g_stack_lds{1}, g_host_duplex{1}, g_host_bands{6};

dcp::LaunchOpts current_opts() {
  dcp::LaunchOpts o;
  o.tile_rows = g_tile_rows.load();
  o.xcd_remap = g_xcd_remap.load();
  o.coef_lds = g_coef_lds.load();
  o.d_chunk = g_d_chunk.load();
  o.pipe_depth = g_pipe_depth.load();
  o.lds_gather = g_lds_gather.load();
  o.stack_lds = g_stack_lds.load();
  return o;
}

thread_local Staging g_staging;
thread_local HostStreams g_host_streams;

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
  host_row_band_rect(m, H, 0.0, (double)(W - 1), row_start, row_start + (double)(nrows - 1), b0, b1);
}

// The same for the radial map evaluated at any position of the rectangle [x_lo, x_hi] x [y_lo, y_hi] (the fused map
// evaluates it at perspective-corrected positions rather than on the pixel grid).
void host_row_band_rect(const dcp::MapArgs& m, int64_t H, double x_lo, double x_hi, double y_lo, double y_hi, int64_t* b0,
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
  const double yu0 = y_lo - m.yc, yu1 = y_hi - m.yc;
  const double ya = std::fmin(std::fabs(yu0), std::fabs(yu1));
  const double ay_min = (yu0 <= 0.0 && yu1 >= 0.0) ? 0.0 : ya;                    // smallest |yu| over the rows
  const double ay_max = std::fmax(std::fabs(yu0), std::fabs(yu1));
  const double xl = x_lo - m.xc, xr = x_hi - m.xc;
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

}  // namespace dcpapi

using namespace dcpapi;

extern "C" {

int dcp_version(void) { return 100; }

int dcp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* dcp_last_error(void) { return dcpapi::last_error(); }

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
  } else if (!strcmp(key, "stack_lds")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "stack_lds must be 0, 1 or 2");
    g_stack_lds = value;
  } else if (!strcmp(key, "host_duplex")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "host_duplex must be 0, 1 or 2");
    g_host_duplex = value;
  } else if (!strcmp(key, "host_bands")) {
    if (value < 1 || value > 256) return fail(DCP_ERR_INVALID_ARG, "host_bands must be in [1, 256]");
    g_host_bands = value;
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
  else if (!strcmp(key, "stack_lds")) *value = g_stack_lds;
  else if (!strcmp(key, "host_duplex")) *value = g_host_duplex;
  else if (!strcmp(key, "host_bands")) *value = g_host_bands;
  else return fail(DCP_ERR_INVALID_ARG, "unknown option '%s'", key);
  return DCP_OK;
}

int dcp_release_scratch(void) {
  // the calling thread's staging buffers and streams (DCP_MEM_HOST calls), and the spline planes of every device
  g_staging.release();
  g_host_streams.release();
  return release_spline_workspace();
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

int dcp_stream_create(void** stream, int device) {
  if (!stream) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipStream_t s;
  DCP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return DCP_OK;
}

int dcp_stream_destroy(void* stream) {
  if (stream) DCP_HIP(hipStreamDestroy((hipStream_t)stream));
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
