// api_stack.cpp -- stack entry points of the C ABI (unwarp_slice_backward / unwarp_chunk_slices_backward over a
// (depth, height, width) stack): device-resident stacks, host stacks streamed through the GPU in depth chunks,
// buffers that hold only the reachable row band (out-of-core callers), host stacks sharded over several GPUs.
#include "api_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace dcpapi;

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
  int32_t rb0, rbh;            // the reference's row band of a chunk call when a coordinate might leave it (rbh = 0: cannot)
};

// base = address of (projection 0, row 0) -- possibly before the buffer when the band starts later; the
// kernels only touch rows inside the band
hipError_t launch_stack_any(const StackCall& c, bool fast, const void* base, void* out, int64_t n, int64_t proj_stride,
                            int64_t row_stride, int64_t rows_end, const dcp::LaunchOpts& opts, hipStream_t hs) {
  if (fast) {
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
    st.rb0 = c.rb0;
    st.rbh = c.rbh;
    if (c.rbh > 0) {                 // a model that may fold: the direct kernel, which checks every row coordinate against the band
      dcp::LaunchOpts plain = opts;
      plain.stack_wg = 0;
      plain.stack_lds = 0;
      return dcp::launch_stack(st, c.map, c.sampler, c.round_f32 != 0, plain, hs);
    }
    return dcp::launch_stack(st, c.map, c.sampler, c.round_f32 != 0, opts, hs);
  }
  if (c.round_f32 && !c.out_f32 && opts.stack_wg && c.rbh == 0) {
    // 8- / 16- / 32-bit integer stacks under a certified map: the workgroup-box kernel (uint16 cfg4 shard 0.42 of the 8 TB/s peak
    // against 0.30 for the generic kernel; its launcher declines small launches and ineligible layouts)
    const int64_t esz = dcp::elem_size(c.dtype);
    const double ext = (double)((rows_end - 1) * row_stride + c.width) * (double)esz;
    if (ext < 4294900000.0) {
      dcp::StackArgs sw;
      memset(&sw, 0, sizeof(sw));
      sw.D = (int32_t)n;
      sw.H = (int32_t)c.height;
      sw.W = (int32_t)c.width;
      sw.row_start = c.row_start;
      sw.nrows = (int32_t)c.nrows;
      sw.vol = (const float*)base;
      sw.out = (float*)out;
      sw.proj_stride = proj_stride;
      sw.row_stride = (int32_t)row_stride;
      sw.proj_bytes = (uint32_t)ext;
      bool taken = false;
      hipError_t e = dcp::launch_stack_wg_typed(sw, c.map, c.dtype, opts, hs, &taken);
      if (e != hipSuccess || taken) return e;
    }
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
  st.rb0 = c.rb0;
  st.rbh = c.rbh;
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
  // the tuned float32 kernel gathers 8-byte pairs with 32-bit byte offsets inside a projection; tiny or
  // huge (> 4 GiB) projections take the generic kernel (64-bit addressing, scipy's exact blend)
  const bool fast = c.dtype == dcp::kF32 && !c.out_f32 && height >= 2 && width >= 2 &&
                    (double)height * (double)c.row_stride * 4.0 <= 4294967040.0;
  if (height > 1073741823LL || width > 1073741823LL || depth > 2147483647LL) return fail(DCP_ERR_UNSUPPORTED, "stack too large");
  if (nrows > 65535) return fail(DCP_ERR_UNSUPPORTED, "nrows > 65535 in one call");
  if (!std::isfinite(c.row_start)) return fail(DCP_ERR_INVALID_ARG, "row_start is not finite");
  // rows of a projection the requested rows can reach (the reference slices mat3D[i, yd_min:yd_max, :] for
  // the same reason, postprocessing.py:221-228)
  int64_t band0 = 0, band1 = height;
  const bool partial = c.band_start != 0 || c.band_rows != height;
  if (partial || c.mem_kind == DCP_MEM_HOST)    // (a whole device-resident stack needs no band: skip the host arithmetic)
    host_row_band(c.map, height, width, c.row_start, nrows, &band0, &band1);
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
    DCP_HIP(launch_stack_any(c, fast, base, c.out, depth, c.proj_stride, c.row_stride, c.band_start + c.band_rows, opts, hs));
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
    up_err = launch_stack_any(c, fast, base, dout[k & 1], n, bh * width, width, band1, opts, hs);
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
  // coord_round_f32 = 2: float32 coordinates under unwarp_image_backward's semantics -- the `depth` projections are FRAMES of a 3-D
  // array, every coordinate is clipped to the whole image (postprocessing.py:144-145) and there is no row band to crop or reflect
  // in, whatever the model does (a folding model just reads other rows of the frame)
  const bool frames = coord_round_f32 == 2;
  c->round_f32 = coord_round_f32 != 0;
  c->mem_kind = mem_kind;
  c->device = device;
  c->stream = stream;
  c->sampler = dcp::kScipy;
  if (dtype == dcp::kF32 && !out_f32 && (rc = sampler_of(1, blend_mode, &c->sampler)) != DCP_OK) return rc;
  if ((rc = fill_map(&c->map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  // the certificate (and the monotonicity test below) bound the model over the radii of the FRAME [0, W-1] x [0, H-1]: rows
  // requested outside it are evaluated beyond the proven radius and take the per-pixel-checked kernels
  const bool rows_inside = std::isfinite(row_start) && row_start >= 0.0 && nrows >= 1 && row_start + (double)(nrows - 1) <= (double)(height - 1);
  if (g_tile_cert.load() && coord_round_f32 && height > 0 && width > 0 && rows_inside)
    c->map.tile_dev_ok = tile_deviation_certified(dcp::kRadial, c->map, height, width);
  // unwarp_chunk_slices_backward crops the rows [yd_min, yd_max) spanned by its first and last rows and lets scipy reflect
  // inside that band (postprocessing.py:289-312).  Under a model whose row coordinate increases with the row nothing can
  // leave the band; otherwise (a folding model) the band goes to the kernel, which then checks every pixel.
  c->rb0 = c->rbh = 0;
  if (frames && !rows_inside) return fail(DCP_ERR_INVALID_ARG, "coord_round_f32 = 2 (whole frames) needs the requested rows inside the frame");
  if (!frames && coord_round_f32 && nrows > 0 && height > 0 && width > 0 && std::isfinite(row_start) &&
      (!rows_inside || !radial_monotone_in_y(c->map, height, width))) {
    int64_t b0 = 0, b1 = height;
    reference_chunk_band(c->map, height, width, row_start, row_start + (double)(nrows - 1), &b0, &b1);
    if (b1 <= b0) return fail(DCP_ERR_INVALID_ARG, "the model folds the requested rows onto an empty band of source rows [%lld, %lld)",
                              (long long)b0, (long long)b1);
    c->rb0 = (int32_t)b0;
    c->rbh = (int32_t)(b1 - b0);
    c->map.tile_dev_ok = 0;
  }
  return DCP_OK;
}

}  // namespace

namespace dcpapi {

// `nframes` device-resident float32 frames of ONE calibration, `pitch` elements apart, results dense (nframes, height, width): the
// frames are the projections of a stack whose every row is wanted -- dcp_unwarp_stack_rows_f32 with row_start = 0, nrows = height --
// and stack_wg_kernel evaluates the coordinates of a tile once for all frames (0.72 of the HBM peak against 0.62 for the
// frame-per-blockIdx.z kernel).  *taken = false (and nothing launched) unless that kernel is the one that would run.
int frames_as_stack(const float* src0, float* dst0, int nframes, int64_t height, int64_t width, int64_t pitch, int64_t row_stride,
                    double xcenter, double ycenter, const double* list_fact, int nfact, int blend_mode, int device, void* stream,
                    bool* taken) {
  *taken = false;
  int rc;
  StackCall c;
  if (height > 65535 || height < 2 || width < 2 || (double)height * (double)row_stride * 4.0 > 4294967040.0) return DCP_OK;
  if ((rc = make_stack_call(&c, src0, dst0, dcp::kF32, 0, nframes, height, width, 0, height, pitch, row_stride, xcenter, ycenter,
                            list_fact, nfact, 0.0, height, 2, blend_mode, DCP_MEM_DEVICE, device, stream)) != DCP_OK)
    return rc;
  dcp::StackArgs st;
  memset(&st, 0, sizeof(st));
  st.D = nframes;
  st.H = (int32_t)height;
  st.W = (int32_t)width;
  st.nrows = (int32_t)height;
  st.vol = src0;
  st.out = dst0;
  st.proj_stride = pitch;
  st.row_stride = (int32_t)row_stride;
  if (!dcp::stack_wg_would_take(st, c.map, current_opts())) return DCP_OK;
  *taken = true;
  return run_stack(c);
}

}  // namespace dcpapi

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

int dcp_unwarp_stack_rows_centres_f32(const float* vol, float* out, int64_t depth, int64_t height, int64_t width, int64_t proj_stride,
                                      int64_t row_stride, const double* xcenters, const double* ycenters, int ncentres,
                                      const double* list_fact, int nfact, double row_start, int64_t nrows, int coord_round_f32,
                                      int blend_mode, int mem_kind, int device, void* stream) {
  int rc, sampler;
  if (ncentres < 0) return fail(DCP_ERR_INVALID_ARG, "ncentres < 0");
  if (ncentres == 0) return DCP_OK;
  if (!xcenters || !ycenters) return fail(DCP_ERR_INVALID_ARG, "null centre array");
  if (depth < 0 || nrows < 0) return fail(DCP_ERR_INVALID_ARG, "negative depth / nrows");
  if (height <= 0 || width <= 0) return fail(DCP_ERR_INVALID_ARG, "projections must be non-empty");
  if ((rc = sampler_of(1, blend_mode, &sampler)) != DCP_OK) return rc;
  const size_t per_centre = (size_t)depth * (size_t)nrows * (size_t)width;
  auto one_by_one = [&]() {
    for (int k = 0; k < ncentres; ++k) {
      const int r = dcp_unwarp_stack_rows_f32(vol, out + (size_t)k * per_centre, depth, height, width, proj_stride, row_stride, xcenters[k],
                                              ycenters[k], list_fact, nfact, row_start, nrows, coord_round_f32, blend_mode, mem_kind, device, stream);
      if (r != DCP_OK) return r;
    }
    return DCP_OK;
  };
  // what the batched kernel covers: the tuned float32 gather (32-bit offsets inside a projection), no row-band check
  bool batched = ncentres > 1 && height >= 2 && width >= 2 && (double)height * (double)row_stride * 4.0 <= 4294967040.0 && nrows <= 65535 &&
                 depth <= 2147483647LL && std::isfinite(row_start) && row_stride >= width && proj_stride >= (height - 1) * row_stride + width &&
                 (mem_kind == DCP_MEM_DEVICE || mem_kind == DCP_MEM_HOST) && nfact >= 0 && nfact <= dcp::kMaxFact && (nfact == 0 || list_fact);
  dcp::MapArgs map;
  int64_t band0 = height, band1 = 0;
  if (batched) {
    const bool rows_inside = row_start >= 0.0 && nrows >= 1 && row_start + (double)(nrows - 1) <= (double)(height - 1);
    for (int k = 0; k < ncentres && batched; ++k) {
      if ((rc = fill_map(&map, xcenters[k], ycenters[k], list_fact, nfact, nullptr)) != DCP_OK) return rc;
      // unwarp_chunk_slices_backward semantics: a model that may fold its rows out of the reference's band goes call by call
      // (the direct kernel then checks every pixel against that centre's band, postprocessing.py:289-312)
      if (coord_round_f32 && nrows > 0 && (!rows_inside || !radial_monotone_in_y(map, height, width))) batched = false;
      if (batched && mem_kind == DCP_MEM_HOST && nrows > 0) {
        int64_t b0 = 0, b1 = height;
        host_row_band(map, height, width, row_start, nrows, &b0, &b1);
        band0 = std::min(band0, b0);
        band1 = std::max(band1, b1);
      }
    }
  }
  if (!batched) return one_by_one();
  if (depth == 0 || nrows == 0) return DCP_OK;
  if (!vol || !out) return fail(DCP_ERR_INVALID_ARG, "null volume pointer");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  const dcp::LaunchOpts opts = current_opts();
  dcp::StackArgs st;
  memset(&st, 0, sizeof(st));
  st.H = (int32_t)height;
  st.W = (int32_t)width;
  st.row_start = row_start;
  st.nrows = (int32_t)nrows;
  hipStream_t hs = (hipStream_t)stream;
  if (mem_kind == DCP_MEM_DEVICE) {
    st.D = (int32_t)depth;
    st.vol = vol;
    st.out = out;
    st.proj_stride = proj_stride;
    st.row_stride = (int32_t)row_stride;
    st.proj_bytes = (uint32_t)(((height - 1) * row_stride + width) * 4);
    DCP_HIP(dcp::launch_stack_centres(st, map, xcenters, ycenters, ncentres, sampler, coord_round_f32 != 0, opts, hs));
    return DCP_OK;
  }
  // host stack: the union of the centres' row bands of a chunk of projections goes up, every centre's rows of that chunk come
  // back (one 2-D copy: `ncentres` runs, one per centre block of the (ncentres, depth, nrows, width) result)
  const int64_t bh = band1 - band0;
  const size_t pbytes = (size_t)bh * (size_t)width * 4, obytes = (size_t)nrows * (size_t)width * 4;
  int64_t dc = (int64_t)(((size_t)g_stack_chunk_kb.load() << 10) / std::max(pbytes, obytes * (size_t)ncentres));
  dc = dc < 1 ? 1 : (dc > depth ? depth : dc);
  void *din = nullptr, *dout = nullptr;
  DCP_HIP(g_staging.get(0, pbytes * (size_t)dc, &din));
  DCP_HIP(g_staging.get(2, obytes * (size_t)ncentres * (size_t)dc, &dout));
  for (int64_t d0 = 0; d0 < depth; d0 += dc) {
    const int64_t n = d0 + dc > depth ? depth - d0 : dc;
    const char* hsrc = (const char*)(vol + (size_t)d0 * (size_t)proj_stride + (size_t)band0 * (size_t)row_stride);
    if (row_stride == width) {
      DCP_HIP(hipMemcpy2DAsync(din, pbytes, hsrc, (size_t)proj_stride * 4, pbytes, (size_t)n, hipMemcpyHostToDevice, hs));
    } else {
      for (int64_t d = 0; d < n; ++d)
        DCP_HIP(hipMemcpy2DAsync((char*)din + (size_t)d * pbytes, (size_t)width * 4, hsrc + (size_t)d * (size_t)proj_stride * 4, (size_t)row_stride * 4,
                                 (size_t)width * 4, (size_t)bh, hipMemcpyHostToDevice, hs));
    }
    st.D = (int32_t)n;
    st.vol = (const float*)((const char*)din - (size_t)band0 * (size_t)width * 4);     // absolute row indexing; rows below band0 are never touched
    st.out = (float*)dout;
    st.proj_stride = bh * width;
    st.row_stride = (int32_t)width;
    st.proj_bytes = (uint32_t)((size_t)band1 * (size_t)width * 4);
    DCP_HIP(dcp::launch_stack_centres(st, map, xcenters, ycenters, ncentres, sampler, coord_round_f32 != 0, opts, hs));
    DCP_HIP(hipMemcpy2DAsync(out + (size_t)d0 * (size_t)nrows * (size_t)width, per_centre * 4, dout, obytes * (size_t)n, obytes * (size_t)n,
                             (size_t)ncentres, hipMemcpyDeviceToHost, hs));
    DCP_HIP(hipStreamSynchronize(hs));
  }
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

// Device-resident shards, one per GPU of this process, and the exchange without torch / RCCL: after its kernel every
// device PUSHES its (d1 - d0, nrows, width) block into the depth-outer result buffer of every other device with
// hipMemcpyPeerAsync -- on an 8-GPU node seven copies per device, each over its own xGMI link (SURVEY.md section 2 C1 /
// section 8(e): "7 concurrent hipMemcpyPeerAsync pushes").  Depth is the outer axis, so each block is contiguous at the
// same offset in every buffer.  The same device may be listed more than once (shards of one GPU; the pushes are then
// plain device-to-device copies) -- which is how the one-GPU test boxes exercise this path.
int dcp_unwarp_stack_rows_peer_f32(const float* const* vol, float* const* out, int64_t depth, int64_t height, int64_t width,
                                   int64_t proj_stride, int64_t row_stride, double xcenter, double ycenter,
                                   const double* list_fact, int nfact, double row_start, int64_t nrows, int coord_round_f32,
                                   int blend_mode, const int* devices, int ndev, int gather) {
  if (ndev < 1 || !devices || !vol || !out) return fail(DCP_ERR_INVALID_ARG, "need at least one device and the pointer arrays");
  if (ndev > 64) return fail(DCP_ERR_INVALID_ARG, "ndev = %d > 64", ndev);
  const int have = dcp_device_count();
  for (int i = 0; i < ndev; ++i)
    if (devices[i] < 0 || devices[i] >= have)
      return have == 0 ? fail(DCP_ERR_NO_DEVICE, "no HIP device visible")
                       : fail(DCP_ERR_INVALID_ARG, "devices[%d] = %d outside [0, %d)", i, devices[i], have);
  if (depth < 0 || nrows < 0 || width <= 0 || height <= 0) return fail(DCP_ERR_INVALID_ARG, "negative depth / nrows or empty projections");
  const int64_t base = depth / ndev, extra = depth % ndev;
  std::vector<int64_t> d0((size_t)ndev + 1, 0);
  for (int i = 0; i < ndev; ++i) d0[(size_t)i + 1] = d0[(size_t)i] + base + (i < extra ? 1 : 0);
  for (int i = 0; i < ndev; ++i)
    if ((d0[(size_t)i + 1] > d0[(size_t)i] && !vol[i]) || !out[i]) return fail(DCP_ERR_INVALID_ARG, "null shard / result pointer for device slot %d", i);
  int prev = 0;
  DCP_HIP(hipGetDevice(&prev));
  std::vector<hipStream_t> streams((size_t)ndev, nullptr);
  int rc = DCP_OK;
  // drains and destroys every stream; an asynchronous fault of a kernel or a peer copy surfaces here and must not be
  // reported as success (the caller would read a partly written gather)
  auto finish = [&](int code) {
    hipError_t first = hipSuccess;
    int first_dev = -1;
    for (int i = 0; i < ndev; ++i)
      if (streams[(size_t)i]) {
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamSynchronize(streams[(size_t)i]);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess && first == hipSuccess) {
          first = e;
          first_dev = devices[i];
        }
        (void)hipStreamDestroy(streams[(size_t)i]);
        (void)hipGetLastError();
      }
    (void)hipSetDevice(prev);
    if (code == DCP_OK && first != hipSuccess)
      return fail(DCP_ERR_HIP, "stack kernel / peer copy on device %d failed: %s", first_dev, hipGetErrorString(first));
    return code;
  };
  const size_t block = (size_t)nrows * (size_t)width;           // floats per projection of the result
  // 1. every device: its kernel on its own stream, writing its block in place (offset d0 of its result when gathering)
  for (int i = 0; i < ndev && rc == DCP_OK; ++i) {
    if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&streams[(size_t)i], hipStreamNonBlocking) != hipSuccess) {
      rc = fail(DCP_ERR_HIP, "cannot set up device %d", devices[i]);
      break;
    }
    if (gather)
      for (int h = 0; h < ndev && rc == DCP_OK; ++h)
        if (devices[h] != devices[i]) {
          const hipError_t e = hipDeviceEnablePeerAccess(devices[h], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            rc = fail(DCP_ERR_HIP, "device %d cannot access device %d: %s", devices[i], devices[h], hipGetErrorString(e));
          (void)hipGetLastError();
        }
    if (rc != DCP_OK) break;
    const int64_t n = d0[(size_t)i + 1] - d0[(size_t)i];
    if (n == 0) continue;
    float* mine = out[i] + (gather ? (size_t)d0[(size_t)i] * block : 0);
    rc = dcp_unwarp_stack_rows_f32(vol[i], mine, n, height, width, proj_stride, row_stride, xcenter, ycenter, list_fact, nfact, row_start,
                                   nrows, coord_round_f32, blend_mode, DCP_MEM_DEVICE, devices[i], streams[(size_t)i]);
  }
  if (rc != DCP_OK) return finish(rc);
  // 2. the exchange: device i pushes its block to every other slot's result, behind its kernel on the same stream
  if (gather) {
    for (int i = 0; i < ndev; ++i) {
      const int64_t n = d0[(size_t)i + 1] - d0[(size_t)i];
      if (n == 0) continue;
      if (hipSetDevice(devices[i]) != hipSuccess) return finish(fail(DCP_ERR_HIP, "cannot select device %d", devices[i]));
      const size_t off = (size_t)d0[(size_t)i] * block, bytes = (size_t)n * block * sizeof(float);
      for (int k = 1; k < ndev; ++k) {
        const int h = (i + k) % ndev;                           // staggered: at any moment the ndev pushes in flight go to distinct targets
        if (out[h] == out[i]) continue;                         // slots sharing one result buffer (same device listed twice)
        hipError_t e = hipMemcpyPeerAsync(out[h] + off, devices[h], out[i] + off, devices[i], bytes, streams[(size_t)i]);
        if (e != hipSuccess) return finish(fail(DCP_ERR_HIP, "peer copy %d -> %d failed: %s", devices[i], devices[h], hipGetErrorString(e)));
      }
    }
  }
  return finish(DCP_OK);
}

}  // extern "C"
