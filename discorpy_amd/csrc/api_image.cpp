// api_image.cpp -- whole-image entry points of the C ABI: radial / perspective / fused remaps, explicit
// coordinates and coordinate maps for float32 (tuned kernels, unwarp_kernels.hip), the same for the other
// element types and interleaved channels (typed_kernels.hip; orders >= 2 go to api_spline.cpp).
#include "api_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>
#include <thread>
#include <vector>

using namespace dcpapi;

namespace {

// The float32 kernels address their source with 32-bit byte offsets (buffer instructions).  A source that
// needs more -- a 40 000 x 40 000 frame fits this GPU's memory many times over -- goes to the generic
// kernels of typed_kernels.hip (64-bit addressing, scipy's exact blend) instead of being refused.
bool beyond_32bit_offsets(int64_t H, int64_t W, int64_t rs, int64_t cs) {
  if (H <= 0 || W <= 0 || rs < 1 || cs < 1) return false;
  const double extent = ((double)(H - 1) * (double)rs + (double)(W - 1) * (double)cs + 1.0) * 4.0;
  return extent > 4294967040.0 && H <= 1073741823LL && W <= 1073741823LL;
}

int run_typed(int map_kind, const void* src, void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs,
              const dcp::MapArgs& map, const void* ycoord, const void* xcoord, int coord_dtype, int64_t npts, int order,
              int mode, int mem_kind, int device, void* stream);

// Does this HIP runtime move pageable data in both PCIe directions at once when two host threads copy on two
// streams?  ROCm 7.2's does (45 GB/s each way); the runtime bundled with PyTorch-ROCm 2.10 serialises the two
// directions, and the banded path then only adds overhead.  Measured once per process on 32 MiB buffers.
// (dcp_get_option("host_direct_applies"): would a registered float32 host destination be written directly?  The Python pool
// pins its blocks only then)
bool runtime_overlaps_directions();
bool host_direct_applies_here() { return g_host_direct.load() && (g_host_direct.load() == 2 || !g_host_duplex.load() || !runtime_overlaps_directions()); }

bool runtime_overlaps_directions() {
  static std::once_flag once;
  static bool overlaps = false;
  std::call_once(once, []() {
    const size_t n = 32u << 20;
    void *d0 = nullptr, *d1 = nullptr;
    hipStream_t s_up = nullptr, s_down = nullptr;
    if (g_host_streams.get(&s_up, &s_down) != hipSuccess) return;
    if (g_staging.get(2, n, &d0) != hipSuccess || g_staging.get(3, n, &d1) != hipSuccess) return;
    std::vector<char> up(n, 1), down(n, 2);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto copy_up = [&]() { (void)hipMemcpyAsync(d0, up.data(), n, hipMemcpyHostToDevice, s_up); (void)hipStreamSynchronize(s_up); };
    auto copy_down = [&]() { (void)hipMemcpyAsync(down.data(), d1, n, hipMemcpyDeviceToHost, s_down); (void)hipStreamSynchronize(s_down); };
    auto now = []() { return std::chrono::steady_clock::now(); };
    copy_up();
    copy_down();                                   // first use of the host pages
    double serial = 1e30, parallel = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      auto t0 = now();
      copy_up();
      copy_down();
      serial = std::min(serial, std::chrono::duration<double>(now() - t0).count());
      t0 = now();
      std::thread th([&]() { (void)hipSetDevice(dev); copy_down(); });
      copy_up();
      th.join();
      parallel = std::min(parallel, std::chrono::duration<double>(now() - t0).count());
    }
    overlaps = parallel < 0.85 * serial;
  });
  return overlaps;
}

// Row boundaries of the bands a host frame of H rows travels in (run_host_banded, run_host_direct): start[k] .. start[k + 1].
// Nominal height = H / nbands rounded up to 64 rows; the first bands grow 128, 256, 512 .. up to it and the last ones shrink the same
// way, so that the stretch during which only ONE direction of the link is busy -- the first upload, the last download -- is short.
// Frames too small for that (fewer than four nominal bands' worth of rows) are cut evenly.
static std::vector<int64_t> band_plan(int64_t H, int64_t nbands) {
  if (nbands < 1) nbands = 1;
  int64_t big = ((H + nbands - 1) / nbands + 63) / 64 * 64;
  if (big > 65535) big = 65535 / 64 * 64;
  std::vector<int64_t> head;
  for (int64_t r = 128; r < big; r *= 2) head.push_back(r);
  int64_t ramp = 0;
  for (int64_t r : head) ramp += r;
  std::vector<int64_t> start{0};
  if (nbands < 3 || 2 * ramp + 2 * big > H) {
    for (int64_t r = big; r < H; r += big) start.push_back(r);
    start.push_back(H);
    return start;
  }
  int64_t r = 0;
  for (int64_t h : head) start.push_back(r += h);
  const int64_t tail0 = (H - ramp) / 64 * 64;          // where the closing ramp begins (every boundary a multiple of 64 rows)
  while (r + big <= tail0) start.push_back(r += big);
  if (tail0 - r >= 64) start.push_back(r = tail0);     // (a short band in front of the ramp; under 64 rows the ramp's first band takes them)
  for (size_t i = head.size(); i-- > 0;) {
    r = H;
    for (size_t j = 0; j < i; ++j) r -= head[j];
    if (i > 0) r = r / 64 * 64;
    if (r > start.back()) start.push_back(r);
  }
  if (start.back() != H) start.push_back(H);
  return start;
}

// Host frame, radial map, order 1: the frame goes through the GPU in bands of output rows so that PCIe carries data
// in both directions at once.  Output rows [r, r + n) only need the source rows host_row_band() reports, so while
// this thread uploads the source top to bottom and launches one stack-kernel band (depth 1: a band of image rows is
// a chunk of rows of a one-projection stack, bit-identical to the image kernels) as soon as its source rows have
// arrived, a second thread copies finished bands back.  With a runtime that overlaps the two directions (ROCm 7.2's)
// a 4096 x 4096 frame takes ~1.5 ms instead of 2.45 ms; with one that serialises them it costs the same as before.
// `pix` = bytes per pixel (all channels), `rs_bytes` = host row stride in bytes, source_rows(r0, n, &b0, &b1) = source rows
// [b0, b1) that output rows [r0, r0 + n) can reach, launch_band(dsrc, dband, r0, n, stream) enqueues the kernel for
// those output rows reading the whole-frame device copy `dsrc`.
template <typename Hull, typename LaunchBand>
int run_host_banded(const void* src, void* dst, int64_t H, int64_t W, size_t pix, size_t rs_bytes, Hull&& source_rows,
                    LaunchBand&& launch_band) {
  const size_t row_bytes = (size_t)W * pix;
  const size_t frame = (size_t)H * row_bytes;
  void *dsrc = nullptr, *ddst = nullptr;
  DCP_HIP(g_staging.get(0, frame, &dsrc));
  DCP_HIP(g_staging.get(1, frame, &ddst));
  hipStream_t s_up = nullptr, s_down = nullptr;
  DCP_HIP(g_host_streams.get(&s_up, &s_down));
  int cur_dev = 0;
  DCP_HIP(hipGetDevice(&cur_dev));
  // bands of output rows: `host_bands` equal ones would leave the first band's upload and the last band's download uncovered by the
  // other direction (2 / host_bands of the transfer time); the plan below opens with 128, 256, 512 .. rows, runs at the nominal band
  // height and closes .., 512, 256, 128 -- head and tail shrink to ~3 % of a 4096-row frame each
  // (twice `host_bands` nominal bands here: 1.77-1.85 ms against 1.85-1.87 per 4096^2 frame on /opt/rocm's runtime, profiles/r06g_host_bands.txt;
  // the direct-write path, whose bands are kernels storing over PCIe, is best at `host_bands` itself)
  const std::vector<int64_t> start = band_plan(H, 2 * (int64_t)g_host_bands.load());
  const int64_t nb = (int64_t)start.size() - 1;
  // one event per band, recorded behind its kernel: the downloader's stream waits for it on the device, this thread goes straight
  // on to the next upload (round 6; it used to synchronise the stream after every band: ~25 us x bands on the upload's critical path)
  thread_local std::vector<hipEvent_t> ev;
  thread_local int ev_dev = -1;
  if (ev_dev != cur_dev) {
    for (auto e : ev) (void)hipEventDestroy(e);
    ev.clear();
    ev_dev = cur_dev;
  }
  while ((int64_t)ev.size() < nb) {
    hipEvent_t e = nullptr;
    DCP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ev.push_back(e);
  }
  hipEvent_t* const band_done = ev.data();      // (the downloader thread must not name `ev`: a thread_local is ITS OWN, empty, vector there)

  std::mutex mu;
  std::condition_variable cv;
  int64_t computed = 0;
  bool abort_down = false;
  hipError_t down_err = hipSuccess;
  std::thread downloader([&]() {
    hipError_t e = hipSetDevice(cur_dev);
    for (int64_t k = 0; k < nb && e == hipSuccess; ++k) {
      {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return computed > k || abort_down; });
        if (abort_down) break;
      }
      const int64_t r0 = start[(size_t)k], n = start[(size_t)k + 1] - r0;
      e = hipStreamWaitEvent(s_down, band_done[k], 0);             // band k's kernel has finished (recorded before `computed` moved)
      if (e == hipSuccess)
        e = hipMemcpyAsync((char*)dst + (size_t)r0 * row_bytes, (const char*)ddst + (size_t)r0 * row_bytes, (size_t)n * row_bytes,
                           hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess) e = hipStreamSynchronize(s_down);
    }
    std::lock_guard<std::mutex> lock(mu);
    down_err = e;
  });
  hipError_t up_err = hipSuccess;
  int64_t uploaded = 0;      // source rows [0, uploaded) are on the device
  for (int64_t k = 0; k < nb && up_err == hipSuccess; ++k) {
    const int64_t r0 = start[(size_t)k], n = start[(size_t)k + 1] - r0;
    int64_t b0 = 0, b1 = H;
    source_rows(r0, n, &b0, &b1);
    // the source arrives top to bottom; a band whose rows reach further down simply waits for more of it
    // (for the last band everything is uploaded whatever the hull says)
    const int64_t need = (k == nb - 1) ? H : b1;
    if (need > uploaded) {
      up_err = hipMemcpy2DAsync((char*)dsrc + (size_t)uploaded * row_bytes, row_bytes, (const char*)src + (size_t)uploaded * rs_bytes,
                                rs_bytes, row_bytes, (size_t)(need - uploaded), hipMemcpyHostToDevice, s_up);
      uploaded = need;
      if (up_err != hipSuccess) break;
    }
    if (b0 < 0 || b1 > uploaded) { up_err = hipErrorInvalidValue; break; }   // cannot happen: need >= b1
    up_err = launch_band(dsrc, (char*)ddst + (size_t)r0 * row_bytes, r0, n, s_up);
    if (up_err == hipSuccess) up_err = hipEventRecord(band_done[k], s_up);
    if (up_err == hipSuccess && g_host_band_sync.load()) up_err = hipStreamSynchronize(s_up);      // (A/B: rounds 1-5 waited here for every band)
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
  // (whatever happened, nothing of this call is left running: the kernels read the staging buffer the next call reuses)
  const hipError_t drain = hipStreamSynchronize(s_up);
  if (up_err != hipSuccess) return fail(DCP_ERR_HIP, "banded frame upload / kernel failed: %s", hipGetErrorString(up_err));
  if (down_err != hipSuccess) return fail(DCP_ERR_HIP, "banded frame download failed: %s", hipGetErrorString(down_err));
  if (drain != hipSuccess) return fail(DCP_ERR_HIP, "banded frame: %s", hipGetErrorString(drain));
  return DCP_OK;
}

// Is [p, p + bytes) host memory the GPU can address (hipHostRegister / hipHostMalloc)?  Returns its device-visible address, or
// nullptr.  BOTH ends are asked about: a view that starts inside a registered range shorter than the frame (a C caller's partly
// registered buffer, a NumPy view past a pinned block) must take the staged path, or the kernels would store to unmapped memory.
void* registered_host_alias(const void* p, size_t bytes) {
  if (!bytes) return nullptr;
  hipPointerAttribute_t first, last;
  memset(&first, 0, sizeof(first));
  memset(&last, 0, sizeof(last));
  if (hipPointerGetAttributes(&first, p) != hipSuccess || hipPointerGetAttributes(&last, (const char*)p + (bytes - 1)) != hipSuccess) {
    (void)hipGetLastError();            // plain pageable memory: not an error
    return nullptr;
  }
  if (first.type != hipMemoryTypeHost || last.type != hipMemoryTypeHost || !first.devicePointer || !last.devicePointer) return nullptr;
  // one mapping from end to end: the device alias advances exactly as the host pointer does
  if ((const char*)last.devicePointer - (const char*)first.devicePointer != (ptrdiff_t)(bytes - 1)) return nullptr;
  return first.devicePointer;
}

// Host frame whose DESTINATION is registered host memory (the recycled outputs of the Python front end are; so is anything a C
// caller got from hipHostMalloc / hipHostRegister): the kernels write their rows straight into it over PCIe -- no device result,
// no download.  The source goes up in bands on one stream; the kernel of band k, on a second stream behind an event, writes band k
// of the result while band k + 1 is uploaded: the upload is the copy engine's, the result the shader's stores, and the two
// directions overlap whatever the runtime does with two copies (the runtime bundled with PyTorch-ROCm serialises those: 2.45 ms
// per 4096^2 frame; this path: see DESIGN.md section 6, round 3).
template <typename Hull, typename LaunchBand>
int run_host_direct(const void* src, void* dst_alias, int64_t H, int64_t W, size_t pix, size_t rs_bytes, Hull&& source_rows,
                    LaunchBand&& launch_band) {
  const size_t row_bytes = (size_t)W * pix;
  void* dsrc = nullptr;
  DCP_HIP(g_staging.get(0, (size_t)H * row_bytes, &dsrc));
  hipStream_t s_up = nullptr, s_k = nullptr;
  DCP_HIP(g_host_streams.get(&s_up, &s_k));
  thread_local hipEvent_t ev[2] = {nullptr, nullptr};
  thread_local int ev_dev = -1;
  int cur = 0;
  DCP_HIP(hipGetDevice(&cur));
  if (ev_dev != cur) {
    for (auto& e : ev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
      DCP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    ev_dev = cur;
  }
  const int64_t nbands = g_host_bands.load();
  int64_t rows_per = ((H + nbands - 1) / nbands + 63) / 64 * 64;
  if (rows_per > 65535) rows_per = 65535 / 64 * 64;
  const int64_t nb = (H + rows_per - 1) / rows_per;
  // (an error in the middle leaves work queued on both streams that reads the staging buffer and writes the caller's memory:
  // whatever happens, both streams are drained before this function returns)
  const char* what = "";
  auto enqueue = [&]() -> hipError_t {
    int64_t uploaded = 0;
    for (int64_t k = 0; k < nb; ++k) {
      const int64_t r0 = k * rows_per, n = (r0 + rows_per > H ? H - r0 : rows_per);
      int64_t b0 = 0, b1 = H;
      source_rows(r0, n, &b0, &b1);
      const int64_t need = (k == nb - 1) ? H : b1;
      hipError_t e;
      if (need > uploaded) {
        what = "upload of a band";
        e = hipMemcpy2DAsync((char*)dsrc + (size_t)uploaded * row_bytes, row_bytes, (const char*)src + (size_t)uploaded * rs_bytes, rs_bytes, row_bytes,
                             (size_t)(need - uploaded), hipMemcpyHostToDevice, s_up);
        if (e != hipSuccess) return e;
        uploaded = need;
      }
      what = "band hull outside the uploaded rows";
      if (b0 < 0 || b1 > uploaded) return hipErrorInvalidValue;      // cannot happen: need >= b1
      what = "event between the upload and the kernel of a band";
      if ((e = hipEventRecord(ev[k & 1], s_up)) != hipSuccess) return e;
      if ((e = hipStreamWaitEvent(s_k, ev[k & 1], 0)) != hipSuccess) return e;
      what = "kernel of a band";
      if ((e = launch_band(dsrc, (char*)dst_alias + (size_t)r0 * row_bytes, r0, n, s_k)) != hipSuccess) return e;
    }
    return hipSuccess;
  };
  const hipError_t e_enq = enqueue();
  const hipError_t e_up = hipStreamSynchronize(s_up), e_k = hipStreamSynchronize(s_k);
  if (e_enq != hipSuccess) return fail(DCP_ERR_HIP, "host frame written in place: %s failed: %s", what, hipGetErrorString(e_enq));
  if (e_up != hipSuccess || e_k != hipSuccess)
    return fail(DCP_ERR_HIP, "host frame written in place: %s", hipGetErrorString(e_up != hipSuccess ? e_up : e_k));
  return DCP_OK;
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
  dcp::LaunchOpts opts = current_opts();
  if (mem_kind == DCP_MEM_DEVICE_UNORDERED) {      // device pointers, and the launch need not wait for earlier work of `stream`
    opts.any_order = 1;
    mem_kind = DCP_MEM_DEVICE;
  }
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
  // a destination the GPU can address: the kernels write into it directly (run_host_direct)
  // (taken where the runtime cannot run an upload and a download at once -- the one bundled with PyTorch-ROCm: 2.16 ms against
  // 2.44 per 4096^2 frame; where it can -- /opt/rocm's -- the two-copy banded path below is the faster one, 1.78 ms against 2.15.
  // "host_direct" = 2 forces it, 0 forbids it)
  void* dst_alias = nullptr;
  if (g_host_direct.load() && cs == 1 && H >= 512 && W >= 2 && (double)H * (double)W * 4.0 >= 16.0 * 1048576.0 &&
      (double)H * (double)W * 4.0 <= 4294967040.0 && (kind == dcp::kRadial || map.fast_div) &&
      (g_host_direct.load() == 2 || !g_host_duplex.load() || !runtime_overlaps_directions()))
    dst_alias = registered_host_alias(dst, (size_t)H * (size_t)W * sizeof(float));
  if (dst_alias) {
    auto band_rows = [&](int64_t r0, int64_t n, int64_t* b0, int64_t* b1) {
      if (kind == dcp::kRadial) {
        host_row_band(map, H, W, (double)r0, n, b0, b1);
        return;
      }
      // homography with a denominator of one sign over the frame: the extremes of a band of rows are at its four corners
      double ylo = 1e300, yhi = -1e300, xlo = 1e300, xhi = -1e300;
      for (double y : {(double)r0, (double)(r0 + n - 1)})
        for (double x : {0.0, (double)(W - 1)}) {
          const double den = (map.coef[6] * x + map.coef[7] * y) + 1.0;
          double yd = ((map.coef[3] * x + map.coef[4] * y) + map.coef[5]) / den;
          double xd = ((map.coef[0] * x + map.coef[1] * y) + map.coef[2]) / den;
          if (!(yd >= 0.0)) yd = 0.0;
          if (yd > (double)(H - 1)) yd = (double)(H - 1);
          if (!(xd >= 0.0)) xd = 0.0;
          if (xd > (double)(W - 1)) xd = (double)(W - 1);
          ylo = std::min(ylo, yd);
          yhi = std::max(yhi, yd);
          xlo = std::min(xlo, xd);
          xhi = std::max(xhi, xd);
        }
      if (kind == dcp::kFused) {
        host_row_band_rect(map, H, xlo - 0.01, xhi + 0.01, ylo - 0.01, yhi + 0.01, b0, b1);
        return;
      }
      *b0 = std::max<int64_t>(0, (int64_t)std::floor(ylo) - 1);
      *b1 = std::min<int64_t>(H, (int64_t)std::floor(yhi) + 3);
    };
    return run_host_direct(src, dst_alias, H, W, sizeof(float), (size_t)rs * sizeof(float), band_rows,
                           [&](void* dsrc, void* dband, int64_t r0, int64_t n, hipStream_t s) {
                             dcp::ImageArgs b;
                             memset(&b, 0, sizeof(b));
                             b.H = (int32_t)H;
                             b.W = (int32_t)W;
                             b.src = (const float*)dsrc;
                             b.dst = (float*)dband;
                             b.src_stride = (int32_t)W;
                             b.src_col_stride = 1;
                             b.src_bytes = (uint32_t)((size_t)H * (size_t)W * 4);
                             b.y_origin = (int32_t)r0;
                             b.rows_out = (int32_t)n;
                             return dcp::launch_image(kind, b, map, sampler, round_f32, opts, s);
                           });
  }
  if ((kind == dcp::kPersp || kind == dcp::kFused) && map.fast_div && cs == 1 && H >= 512 && W >= 2 && g_host_duplex.load() &&
      (double)H * (double)W * 4.0 >= 16.0 * 1048576.0 && (double)H * (double)W * 4.0 <= 4294967040.0 &&
      (g_host_duplex.load() == 2 || runtime_overlaps_directions())) {
    // homography with a denominator of one sign over the frame ("tame", checked by the caller): yd is monotone along
    // every segment, so over a band of output rows it takes its extremes at the band's four corners
    // (the same holds for xd; the fused map then evaluates the radial model inside that clipped rectangle of positions)
    auto band = [&](int64_t r0, int64_t n, int64_t* b0, int64_t* b1) {
      double ylo = 1e300, yhi = -1e300, xlo = 1e300, xhi = -1e300;
      for (double y : {(double)r0, (double)(r0 + n - 1)})
        for (double x : {0.0, (double)(W - 1)}) {
          const double den = (map.coef[6] * x + map.coef[7] * y) + 1.0;
          double yd = ((map.coef[3] * x + map.coef[4] * y) + map.coef[5]) / den;
          double xd = ((map.coef[0] * x + map.coef[1] * y) + map.coef[2]) / den;
          if (!(yd >= 0.0)) yd = 0.0;
          if (yd > (double)(H - 1)) yd = (double)(H - 1);
          if (!(xd >= 0.0)) xd = 0.0;
          if (xd > (double)(W - 1)) xd = (double)(W - 1);
          ylo = std::min(ylo, yd);
          yhi = std::max(yhi, yd);
          xlo = std::min(xlo, xd);
          xhi = std::max(xhi, xd);
        }
      if (kind == dcp::kFused) {
        // float32 rounding of the perspective position moves it by < 1e-3 px: widen the rectangle a little
        host_row_band_rect(map, H, xlo - 0.01, xhi + 0.01, ylo - 0.01, yhi + 0.01, b0, b1);
        return;
      }
      *b0 = std::max<int64_t>(0, (int64_t)std::floor(ylo) - 1);
      *b1 = std::min<int64_t>(H, (int64_t)std::floor(yhi) + 3);
    };
    return run_host_banded(src, dst, H, W, sizeof(float), (size_t)rs * sizeof(float), band,
                           [&](void* dsrc, void* dband, int64_t r0, int64_t n, hipStream_t s) {
                             dcp::ImageArgs b;
                             memset(&b, 0, sizeof(b));
                             b.H = (int32_t)H;
                             b.W = (int32_t)W;
                             b.src = (const float*)dsrc;
                             b.dst = (float*)dband;
                             b.src_stride = (int32_t)W;
                             b.src_col_stride = 1;
                             b.src_bytes = (uint32_t)((size_t)H * (size_t)W * 4);
                             b.y_origin = (int32_t)r0;
                             b.rows_out = (int32_t)n;
                             return dcp::launch_image(kind, b, map, sampler, round_f32, opts, s);
                           });
  }
  if (kind == dcp::kRadial && sampler != dcp::kNearest && round_f32 && cs == 1 && H >= 512 && W >= 2 && g_host_duplex.load() &&
      (double)H * (double)W * 4.0 >= 16.0 * 1048576.0 && (double)H * (double)W * 4.0 <= 4294967040.0 &&
      (g_host_duplex.load() == 2 || runtime_overlaps_directions()))
    return run_host_banded(src, dst, H, W, sizeof(float), (size_t)rs * sizeof(float),
                                  [&](int64_t r0, int64_t n, int64_t* b0, int64_t* b1) { host_row_band(map, H, W, (double)r0, n, b0, b1); },
                                  [&](void* dsrc, void* dband, int64_t r0, int64_t n, hipStream_t s) {
                                    // a band of image rows = a chunk of rows of a one-projection stack
                                    dcp::StackArgs st;
                                    memset(&st, 0, sizeof(st));
                                    st.D = 1;
                                    st.H = (int32_t)H;
                                    st.W = (int32_t)W;
                                    st.row_start = (double)r0;
                                    st.nrows = (int32_t)n;
                                    st.vol = (const float*)dsrc;
                                    st.out = (float*)dband;
                                    st.proj_stride = H * W;
                                    st.row_stride = (int32_t)W;
                                    st.proj_bytes = (uint32_t)((size_t)H * (size_t)W * 4);
                                    return dcp::launch_stack(st, map, sampler, true, opts, s);
                                  });
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

// Orders 0..5 on any element type: 0/1 through typed_kernels.hip, 2..5 through the spline path.
// map_kind 0 radial, 1 perspective, 2 fused, 3 explicit coordinates.
int run_typed(int map_kind, const void* src, void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs,
              const dcp::MapArgs& map, const void* ycoord, const void* xcoord, int coord_dtype, int64_t npts, int order,
              int mode, int mem_kind, int device, void* stream) {
  int rc;
  if (order < 0 || order > 5) return fail(DCP_ERR_INVALID_ARG, "spline order %d outside [0, 5]", order);
  if (mode < 0 || (mode & ~DCP_SPLINE_SCIPY_SUM) > 7) return fail(DCP_ERR_INVALID_ARG, "unknown boundary mode %d", mode);
  if (order < 2) mode &= ~DCP_SPLINE_SCIPY_SUM;        // (orders 0 / 1 of the typed entry points always blend in scipy's order)
  if (order >= 2) {   // run_spline numbers the maps 0 radial, 1 perspective, 2 coordinates, 3 fused
    return run_spline(map_kind == 3 ? 2 : map_kind == 2 ? 3 : map_kind, src, dst, dtype, H, W, rs, cs, map, ycoord, xcoord, coord_dtype, npts,
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
  ca.mode = mode;
  const int64_t nout = map_kind == 3 ? npts : H * W;
  // 8- / 16-bit integers, radial or perspective map, certified: the workgroup-box kernel (same arithmetic, LDS-staged)
  dcp::MapArgs mapc = map;
  auto staged_launch = [&](const void* dsrc_, void* ddst_, int64_t rs_, bool* taken) -> hipError_t {
    *taken = false;
    if ((map_kind != 0 && map_kind != 1) || cs != 1 || (double)extent_bytes_typed(H, W, rs_, 1, dtype) > 4294900000.0) return hipSuccess;
    const dcp::MapKind kind = map_kind == 0 ? dcp::kRadial : dcp::kPersp;
    mapc.tile_dev_ok = g_tile_cert.load() ? tile_deviation_certified(kind, mapc, H, W) : 0;
    if (kind == dcp::kPersp) mapc.fast_div = homography_is_tame(mapc.coef, H, W);
    dcp::ImageArgs im;
    memset(&im, 0, sizeof(im));
    im.H = (int32_t)H;
    im.W = (int32_t)W;
    im.src = (const float*)dsrc_;
    im.dst = (float*)ddst_;
    im.src_stride = (int32_t)rs_;
    im.src_col_stride = 1;
    im.src_bytes = (uint32_t)extent_bytes_typed(H, W, rs_, 1, dtype);
    im.xcd_remap = current_opts().xcd_remap;
    if (kind == dcp::kRadial && (dtype == dcp::kF64 || dtype == dcp::kI32 || dtype == dcp::kU32)) {
      // 4- and 8-byte element types: the interleaved-pixel kernel with one channel (color_kernels.hip)
      im.src_col_stride = 1;
      return dcp::launch_color(im, mapc, 1, dtype, order == 0 ? dcp::kNearest : dcp::kScipy, current_opts(), st, taken);
    }
    return dcp::launch_wg_typed(kind, im, mapc, order, dtype, current_opts(), st, taken);
  };
  if (mem_kind == DCP_MEM_DEVICE) {
    bool taken = false;
    DCP_HIP(staged_launch(src, dst, rs, &taken));
    if (taken) return DCP_OK;
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
  bool taken = false;
  DCP_HIP(staged_launch(dsrc, ddst, rs, &taken));
  if (!taken) DCP_HIP(dcp::launch_typed_image(map_kind, a, map, ca, st));
  DCP_HIP(hipMemcpyAsync(dst, ddst, (size_t)nout * esz, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

}  // namespace

namespace dcpapi {
bool host_direct_applies() { return host_direct_applies_here(); }
}  // namespace dcpapi

extern "C" {

int dcp_unwarp_image_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                         int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                         int nfact, int order, int coord_round_f32, int blend_mode, int mem_kind, int device,
                         void* stream) {
  int rc, sampler;
  if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  if (beyond_32bit_offsets(height, width, src_row_stride, src_col_stride) && coord_round_f32)
    return run_typed(0, src, dst, dcp::kF32, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0, order,
                     0, mem_kind, device, stream);
  if ((rc = check_image(src, dst, height, width, src_row_stride, src_col_stride)) != DCP_OK) return rc;
  {
    int tall = 0;
    map.tile_dev_ok = g_tile_cert.load() ? tile_deviation_certified(dcp::kRadial, map, height, width, &tall) : 0;
    map.tall_ok = g_tile_cert.load() ? tall : 0;
  }
  return run_image(dcp::kRadial, src, dst, height, width, src_row_stride, src_col_stride, map, sampler,
                   coord_round_f32 != 0, mem_kind, device, stream);
}

int dcp_unwarp_images_f32(const float* const* srcs, float* const* dsts, int nframes, int64_t height, int64_t width,
                          int64_t src_row_stride, int64_t src_col_stride, const double* xcenters, const double* ycenters,
                          const double* list_facts, int nfact, int order, int coord_round_f32, int blend_mode, int mem_kind,
                          int device, void* stream) {
  int rc, sampler;
  if (nframes < 0) return fail(DCP_ERR_INVALID_ARG, "nframes < 0");
  if (nframes == 0) return DCP_OK;
  if (!srcs || !dsts || !xcenters || !ycenters) return fail(DCP_ERR_INVALID_ARG, "null frame / centre array");
  if (nfact < 0 || nfact > dcp::kMaxFact)
    return fail(DCP_ERR_INVALID_ARG, "nfact = %d outside [0, %d] (DCP_MAX_FACT is a limit of this library, not of the reference)", nfact, dcp::kMaxFact);
  if (nfact > 0 && !list_facts) return fail(DCP_ERR_INVALID_ARG, "null coefficient pointer");
  if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
  auto one_by_one = [&](int first) {      // frames first .. nframes-1 through the single-frame entry point
    for (int i = first; i < nframes; ++i) {
      const int r = dcp_unwarp_image_f32(srcs[i], dsts[i], height, width, src_row_stride, src_col_stride, xcenters[i], ycenters[i],
                                         list_facts ? list_facts + (size_t)i * (size_t)nfact : nullptr, nfact, order, coord_round_f32,
                                         blend_mode, mem_kind, device, stream);
      if (r != DCP_OK) return r;
    }
    return DCP_OK;
  };
  // host frames are bound by PCIe: each goes through the single-frame host path (bands of rows, uploads and downloads
  // overlapped); float64 coordinates and sources beyond 32-bit offsets have no multi-frame kernel either
  if (mem_kind != DCP_MEM_DEVICE || !coord_round_f32 || nframes == 1 || nfact > 10 ||
      beyond_32bit_offsets(height, width, src_row_stride, src_col_stride))
    return one_by_one(0);
  for (int i = 0; i < nframes; ++i)
    if ((rc = check_image(srcs[i], dsts[i], height, width, src_row_stride, src_col_stride)) != DCP_OK) return rc;
  // Frames of ONE calibration, evenly spaced, results dense -- a (n, height, width) array, which is what the reference's callers
  // loop over (the channels of demo_06.py:111-113, the frames of demo_07.py:25,60): the projections of a stack, every row wanted.
  if (order == 1 && src_col_stride == 1) {
    bool same = true;
    const ptrdiff_t pitch = srcs[1] - srcs[0];
    const int64_t extent = (height - 1) * src_row_stride + width;
    for (int i = 1; i < nframes && same; ++i)
      same = xcenters[i] == xcenters[0] && ycenters[i] == ycenters[0] && srcs[i] - srcs[i - 1] == pitch &&
             dsts[i] - dsts[i - 1] == (ptrdiff_t)(height * width) &&
             (nfact == 0 || memcmp(list_facts + (size_t)i * (size_t)nfact, list_facts, (size_t)nfact * sizeof(double)) == 0);
    if (same && (int64_t)pitch >= extent) {
      bool taken = false;
      rc = frames_as_stack(srcs[0], dsts[0], nframes, height, width, (int64_t)pitch, src_row_stride, xcenters[0], ycenters[0],
                           list_facts, nfact, blend_mode, device, stream, &taken);
      if (rc != DCP_OK || taken) return rc;
    }
  }
  // every frame's calibration must hold the level-2 tile certificate (one box per 128 x 32 workgroup tile)
  std::vector<dcp::BatchFrame> fr((size_t)nframes);
  bool all_certified = g_tile_cert.load() != 0;
  for (int i = 0; i < nframes && all_certified; ++i) {
    dcp::MapArgs map;
    const double* f = list_facts ? list_facts + (size_t)i * (size_t)nfact : nullptr;
    if ((rc = fill_map(&map, xcenters[i], ycenters[i], f, nfact, nullptr)) != DCP_OK) return rc;
    all_certified = tile_deviation_certified(dcp::kRadial, map, height, width) >= 2;
    fr[(size_t)i] = dcp::BatchFrame{srcs[i], dsts[i], xcenters[i], ycenters[i], f};
  }
  if (!all_certified) return one_by_one(0);
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  dcp::ImageArgs img;
  memset(&img, 0, sizeof(img));
  img.H = (int32_t)height;
  img.W = (int32_t)width;
  img.src_stride = (int32_t)src_row_stride;
  img.src_col_stride = (int32_t)src_col_stride;
  img.src_bytes = extent_bytes(height, width, src_row_stride, src_col_stride);
  bool taken = false;
  DCP_HIP(dcp::launch_image_batch(img, fr.data(), nframes, nfact, sampler, current_opts(), (hipStream_t)stream, &taken));
  return taken ? DCP_OK : one_by_one(0);
}

int dcp_perspective_image_f32(const float* src, float* dst, int64_t height, int64_t width,
                              int64_t src_row_stride, int64_t src_col_stride, const double* list_coef,
                              int order, int blend_mode, int mem_kind, int device, void* stream) {
  int rc, sampler;
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
  dcp::MapArgs map;
  if ((rc = fill_map(&map, 0.0, 0.0, nullptr, 0, list_coef)) != DCP_OK) return rc;
  if (height > 0 && width > 0) map.fast_div = homography_is_tame(list_coef, height, width);
  if (beyond_32bit_offsets(height, width, src_row_stride, src_col_stride))
    return run_typed(1, src, dst, dcp::kF32, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0, order,
                     0, mem_kind, device, stream);
  if ((rc = check_image(src, dst, height, width, src_row_stride, src_col_stride)) != DCP_OK) return rc;
  map.tile_dev_ok = g_tile_cert.load() ? tile_deviation_certified(dcp::kPersp, map, height, width) : 0;
  return run_image(dcp::kPersp, src, dst, height, width, src_row_stride, src_col_stride, map, sampler, true,
                   mem_kind, device, stream);
}

int dcp_unwarp_fused_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                         int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                         int nfact, const double* list_coef, int order, int blend_mode, int mem_kind,
                         int device, void* stream) {
  int rc, sampler;
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, list_coef)) != DCP_OK) return rc;
  if (height > 0 && width > 0) map.fast_div = homography_is_tame(list_coef, height, width);
  if (beyond_32bit_offsets(height, width, src_row_stride, src_col_stride))
    return run_typed(2, src, dst, dcp::kF32, height, width, src_row_stride, src_col_stride, map, nullptr, nullptr, 0, 0, order,
                     0, mem_kind, device, stream);
  if ((rc = check_image(src, dst, height, width, src_row_stride, src_col_stride)) != DCP_OK) return rc;
  map.tile_dev_ok = (g_tile_cert.load() && g_fused_wg.load()) ? tile_deviation_certified(dcp::kFused, map, height, width) : 0;
  return run_image(dcp::kFused, src, dst, height, width, src_row_stride, src_col_stride, map, sampler, true,
                   mem_kind, device, stream);
}

int dcp_remap_coords_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                         int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype,
                         int64_t npts, int order, int blend_mode, int mem_kind, int device, void* stream) {
  return dcp_remap_coords_mode_f32(src, dst, height, width, src_row_stride, src_col_stride, ycoord, xcoord, coord_dtype, npts, order,
                                   dcp::kModeNearest, blend_mode, mem_kind, device, stream);
}

int dcp_remap_coords_mode_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                              int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype,
                              int64_t npts, int order, int boundary_mode, int blend_mode, int mem_kind, int device, void* stream) {
  int rc, sampler;
  if (boundary_mode < 0 || boundary_mode > 7) return fail(DCP_ERR_INVALID_ARG, "unknown boundary mode %d", boundary_mode);
  if (beyond_32bit_offsets(height, width, src_row_stride, src_col_stride)) {
    dcp::MapArgs none;
    memset(&none, 0, sizeof(none));
    if ((rc = sampler_of(order, blend_mode, &sampler)) != DCP_OK) return rc;
    return run_typed(3, src, dst, dcp::kF32, height, width, src_row_stride, src_col_stride, none, ycoord, xcoord, coord_dtype,
                     npts, order, boundary_mode, mem_kind, device, stream);
  }
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
  ca.mode = boundary_mode;
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

int dcp_unwarp_image_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width, int64_t src_row_stride,
                           int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact, int nfact,
                           int order, int boundary_mode, int mem_kind, int device, void* stream) {
  int rc;
  // a large dense host frame, orders 0 / 1: the one-channel case of the interleaved entry point, whose host path moves
  // the frame in bands of rows with uploads and downloads overlapped (same arithmetic: scipy's exact blend)
  if (mem_kind == DCP_MEM_HOST && order >= 0 && order <= 1 && boundary_mode >= 0 && boundary_mode <= 7 && src_col_stride == 1 && height >= 512 && dtype >= 0 &&
      dtype < dcp::kNumElemTypes && (double)height * (double)width * (double)dcp::elem_size(dtype) >= 16.0 * 1048576.0)
    return dcp_unwarp_image_channels(src, dst, dtype, height, width, 1, src_row_stride, 1, xcenter, ycenter, list_fact, nfact,
                                     order, mem_kind, device, stream);
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

int dcp_unwarp_image_channels(const void* src, void* dst, int dtype, int64_t height, int64_t width, int channels,
                              int64_t src_row_stride, int64_t src_pixel_stride, double xcenter, double ycenter,
                              const double* list_fact, int nfact, int order, int mem_kind, int device, void* stream) {
  return dcp_unwarp_color_image(src, dst, dtype, height, width, channels, src_row_stride, src_pixel_stride, xcenter, ycenter, list_fact, nfact,
                                order, DCP_BLEND_SCIPY, mem_kind, device, stream);
}

int dcp_unwarp_color_image(const void* src, void* dst, int dtype, int64_t height, int64_t width, int channels, int64_t src_row_stride,
                           int64_t src_pixel_stride, double xcenter, double ycenter, const double* list_fact, int nfact, int order,
                           int blend_mode, int mem_kind, int device, void* stream) {
  int rc;
  if (channels < 1 || channels > 64) return fail(DCP_ERR_INVALID_ARG, "channels = %d outside [1, 64]", channels);
  if (order < 0 || order > 1) return fail(DCP_ERR_UNSUPPORTED, "the interleaved-channel kernels take orders 0 and 1 (got %d)", order);
  if (blend_mode != DCP_BLEND_SCIPY && blend_mode != DCP_BLEND_F64LERP)
    return fail(DCP_ERR_UNSUPPORTED, "interleaved channels blend as scipy does (DCP_BLEND_SCIPY) or within one ulp of it (DCP_BLEND_F64LERP); got %d", blend_mode);
  if (src_pixel_stride < channels) return fail(DCP_ERR_INVALID_ARG, "pixel stride %lld smaller than %d channels", (long long)src_pixel_stride, channels);
  if ((rc = check_image_typed(src, dst, dtype, height, width, src_row_stride, src_pixel_stride)) != DCP_OK) return rc;
  if (src_row_stride < (width - 1) * src_pixel_stride + channels && height > 1)
    return fail(DCP_ERR_INVALID_ARG, "row stride %lld overlaps rows of %lld pixels", (long long)src_row_stride, (long long)width);
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipStream_t st = (hipStream_t)stream;
  // float32 only: the one-ulp factorisation; integer types always blend in scipy's exact order (their rounding ties depend on it)
  const int sampler = order == 0 ? dcp::kNearest : (blend_mode == DCP_BLEND_F64LERP && dtype == dcp::kF32 ? dcp::kF64Lerp : dcp::kScipy);
  dcp::TypedImageArgs a;
  memset(&a, 0, sizeof(a));
  a.H = (int32_t)height;
  a.W = (int32_t)width;
  a.src_stride = src_row_stride;
  a.src_cstride = src_pixel_stride;
  a.order = order;
  a.dtype = dtype;
  a.blend = sampler;
  a.y0 = 0;
  a.rows = (int32_t)height;
  const size_t esz = (size_t)dcp::elem_size(dtype);
  const dcp::LaunchOpts opts = current_opts();
  map.tile_dev_ok = g_tile_cert.load() ? tile_deviation_certified(dcp::kRadial, map, height, width) : 0;
  // rows [r0, r0 + n) of the result from the whole-frame source at dsrc: the workgroup-box kernel (remap_wg_color_kernel) where the
  // call qualifies -- dense pixels of 3 / 4 channels, float32 / uint8 / uint16, certified map --, else one thread per pixel
  auto launch_rows = [&](const void* dsrc, void* drows, int64_t rs_el, int64_t ps_el, int64_t r0, int64_t n, hipStream_t s) -> hipError_t {
    const double ext = ((double)(height - 1) * (double)rs_el + (double)(width - 1) * (double)ps_el + (double)channels) * (double)esz;
    if (ext <= 4294900000.0 && rs_el < (1ll << 31)) {
      dcp::ImageArgs im;
      memset(&im, 0, sizeof(im));
      im.H = (int32_t)height;
      im.W = (int32_t)width;
      im.src = (const float*)dsrc;
      im.dst = (float*)drows;
      im.src_stride = (int32_t)rs_el;
      im.src_col_stride = (int32_t)ps_el;
      im.src_bytes = (uint32_t)ext;
      im.y_origin = (int32_t)r0;
      im.rows_out = (int32_t)n;
      bool taken = false;
      const hipError_t e = dcp::launch_color(im, map, channels, dtype, sampler, opts, s, &taken);
      if (e != hipSuccess || taken) return e;
    }
    dcp::TypedImageArgs b = a;
    b.src = dsrc;
    b.dst = drows;
    b.src_stride = rs_el;
    b.src_cstride = ps_el;
    b.y0 = (int32_t)r0;
    b.rows = (int32_t)n;
    return dcp::launch_typed_channels(b, map, channels, s);
  };
  if (mem_kind == DCP_MEM_DEVICE) {
    DCP_HIP(launch_rows(src, dst, src_row_stride, src_pixel_stride, 0, height, st));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  if (src_pixel_stride == channels && height >= 512 && (double)height * (double)width * (double)channels * (double)esz >= 16.0 * 1048576.0 &&
      g_host_duplex.load() && (g_host_duplex.load() == 2 || runtime_overlaps_directions())) {
    // dense interleaved frame: bands of rows, uploads and downloads overlapped (see run_host_banded)
    return run_host_banded(src, dst, height, width, (size_t)channels * esz, (size_t)src_row_stride * esz,
                                  [&](int64_t r0, int64_t n, int64_t* b0, int64_t* b1) { host_row_band(map, height, width, (double)r0, n, b0, b1); },
                                  [&](void* dsrc, void* dband, int64_t r0, int64_t n, hipStream_t s) {
                                    return launch_rows(dsrc, dband, width * channels, channels, r0, n, s);
                                  });
  }
  const size_t ext = (size_t)((height - 1) * src_row_stride + (width - 1) * src_pixel_stride + channels) * esz;
  const size_t obytes = (size_t)height * (size_t)width * (size_t)channels * esz;
  void *dsrc, *ddst;
  DCP_HIP(g_staging.get(0, ext, &dsrc));
  DCP_HIP(g_staging.get(1, obytes, &ddst));
  DCP_HIP(hipMemcpyAsync(dsrc, src, ext, hipMemcpyHostToDevice, st));
  DCP_HIP(launch_rows(dsrc, ddst, src_row_stride, src_pixel_stride, 0, height, st));
  DCP_HIP(hipMemcpyAsync(dst, ddst, obytes, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

int dcp_map_points_f64(const double* yx_in, double* yx_out, int64_t npts, double xcenter, double ycenter,
                       const double* list_fact, int nfact, int mem_kind, int device, void* stream) {
  int rc;
  if (npts < 0) return fail(DCP_ERR_INVALID_ARG, "npts < 0");
  if (npts > 0 && (!yx_in || !yx_out)) return fail(DCP_ERR_INVALID_ARG, "null point pointer");
  dcp::MapArgs map;
  if ((rc = fill_map(&map, xcenter, ycenter, list_fact, nfact, nullptr)) != DCP_OK) return rc;
  if (npts == 0) return DCP_OK;
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipStream_t st = (hipStream_t)stream;
  if (mem_kind == DCP_MEM_DEVICE) {
    DCP_HIP(dcp::launch_map_points(yx_in, yx_out, npts, map, st));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  void *din, *dout;
  const size_t bytes = (size_t)npts * 16;
  DCP_HIP(g_staging.get(0, bytes, &din));
  DCP_HIP(g_staging.get(1, bytes, &dout));
  DCP_HIP(hipMemcpyAsync(din, yx_in, bytes, hipMemcpyHostToDevice, st));
  DCP_HIP(dcp::launch_map_points((const double*)din, (double*)dout, npts, map, st));
  DCP_HIP(hipMemcpyAsync(yx_out, dout, bytes, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
}

int dcp_map_points_perspective_f64(const double* yx_in, double* yx_out, int64_t npts, const double* list_coef, int mem_kind, int device,
                                   void* stream) {
  int rc;
  if (npts < 0) return fail(DCP_ERR_INVALID_ARG, "npts < 0");
  if (npts > 0 && (!yx_in || !yx_out)) return fail(DCP_ERR_INVALID_ARG, "null point pointer");
  if (!list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  dcp::MapArgs map;
  if ((rc = fill_map(&map, 0.0, 0.0, nullptr, 0, list_coef)) != DCP_OK) return rc;
  if (npts == 0) return DCP_OK;
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipStream_t st = (hipStream_t)stream;
  if (mem_kind == DCP_MEM_DEVICE) {
    DCP_HIP(dcp::launch_map_points_persp(yx_in, yx_out, npts, map, st));
    return DCP_OK;
  }
  if (mem_kind != DCP_MEM_HOST) return fail(DCP_ERR_INVALID_ARG, "unknown mem_kind %d", mem_kind);
  void *din, *dout;
  const size_t bytes = (size_t)npts * 16;
  DCP_HIP(g_staging.get(0, bytes, &din));
  DCP_HIP(g_staging.get(1, bytes, &dout));
  DCP_HIP(hipMemcpyAsync(din, yx_in, bytes, hipMemcpyHostToDevice, st));
  DCP_HIP(dcp::launch_map_points_persp((const double*)din, (double*)dout, npts, map, st));
  DCP_HIP(hipMemcpyAsync(yx_out, dout, bytes, hipMemcpyDeviceToHost, st));
  DCP_HIP(hipStreamSynchronize(st));
  return DCP_OK;
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

}  // extern "C"
