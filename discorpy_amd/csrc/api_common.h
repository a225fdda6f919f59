// api_common.h -- shared by the C-ABI translation units (api_*.cpp): error reporting, device selection,
// per-thread device scratch and streams for DCP_MEM_HOST callers, tuning knobs, argument checks.
// Not installed; the public surface is include/discorpy_hip.h.
#pragma once
#include "../../include/discorpy_hip.h"
#include "dcp_internal.h"

#include <atomic>
#include <cstddef>
#include <cstdint>

namespace dcpapi {

// sets the calling thread's last-error message and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* last_error();

#define DCP_HIP(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) {                                                                                \
      (void)hipGetLastError(); /* reported here: must not resurface as the next launch's hipGetLastError */ \
      return dcpapi::fail(DCP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                     \
    }                                                                                                      \
  } while (0)

extern std::atomic<int> g_tile_rows, g_xcd_remap, g_coef_lds, g_d_chunk, g_pipe_depth, g_lds_gather, g_stack_chunk_kb, g_stack_lds, g_host_duplex, g_host_bands, g_tile_cert, g_wg_box, g_wg_per_cu, g_stack_wg, g_int_exact, g_host_direct, g_tall_tiles, g_store_wait, g_fused_wg, g_any_order, g_host_band_sync;
dcp::LaunchOpts current_opts();

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
extern thread_local Staging g_staging;

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
extern thread_local HostStreams g_host_streams;

bool host_direct_applies();      // api_image.cpp
int sampler_of(int order, int blend_mode, int* sampler);
int check_image(const void* src, const void* dst, int64_t H, int64_t W, int64_t rs, int64_t cs);
int check_image_typed(const void* src, const void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs);
uint32_t extent_bytes(int64_t H, int64_t W, int64_t rs, int64_t cs);
size_t extent_bytes_typed(int64_t H, int64_t W, int64_t rs, int64_t cs, int dtype);
int fill_map(dcp::MapArgs* m, double xc, double yc, const double* fact, int nfact, const double* coef);
int homography_is_tame(const double* c, int64_t H, int64_t W);
// level 1: inside any 64 x 16 output tile the map stays within 0.95 px of the bilinear interpolant of the tile's corners;
// level 2: the same for 128 x 32 tiles (kind: dcp::kRadial or dcp::kPersp; anything else 0) -- lets the staged kernels
// take a tile's source box from its corner pixels alone, without a per-pixel containment vote
// *tall_ok (optional): radial maps -- the bound also holds for 64 x 32 tiles and (nearly) all their source boxes fit 80 x 56
// (remap_wg_color_kernel's tile shape for sheared maps)
int tile_deviation_certified(int kind, const dcp::MapArgs& m, int64_t H, int64_t W, int* tall_ok = nullptr);
// yd = yc + B(r) yu increases with yu at every x over the frame (sufficient test): no row of a chunk can then leave the
// band the reference crops from the chunk's first and last rows
bool radial_monotone_in_y(const dcp::MapArgs& m, int64_t H, int64_t W);
// that band, [*b0, *b1), with the reference's own arithmetic (postprocessing.py:289-301)
void reference_chunk_band(const dcp::MapArgs& m, int64_t H, int64_t W, double row_first, double row_last, int64_t* b0, int64_t* b1);
void host_row_band(const dcp::MapArgs& m, int64_t H, int64_t W, double row_start, int64_t nrows, int64_t* b0, int64_t* b1);
void host_row_band_rect(const dcp::MapArgs& m, int64_t H, double x_lo, double x_hi, double y_lo, double y_hi, int64_t* b0,
                        int64_t* b1);

// api_stack.cpp: frames of one calibration at a constant pitch as the projections of a stack (see there)
int frames_as_stack(const float* src0, float* dst0, int nframes, int64_t height, int64_t width, int64_t pitch, int64_t row_stride,
                    double xcenter, double ycenter, const double* list_fact, int nfact, int blend_mode, int device, void* stream,
                    bool* taken);

// api_spline.cpp: frees the coefficient planes of every device (waits for the devices first)
int release_spline_workspace();

// api_spline.cpp: orders 2..5 on any element type; map_kind 0 radial, 1 perspective, 2 explicit coordinates, 3 fused
int run_spline(int map_kind, const void* src, void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs,
               const dcp::MapArgs& map, const void* ycoord, const void* xcoord, int coord_dtype, int64_t npts, int order,
               int mode, int mem_kind, int device, void* stream);

}  // namespace dcpapi
