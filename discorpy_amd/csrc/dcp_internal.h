// dcp_internal.h -- shared between the HIP kernels (unwarp_kernels.hip) and the
// C-ABI layer (api_*.cpp).  Not installed; the public surface is
// include/discorpy_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcp {

constexpr int kMaxFact = 32;       // longest radial coefficient vector accepted
constexpr int kInlineFact = 10;    // lengths 1..kInlineFact get a fully unrolled, SGPR-resident polynomial
constexpr int kMaxTileRows = 64;   // rows one workgroup walks (runtime option tile_rows <= this)

// sampler selector: order 0, or order 1 with one of the three blend arithmetics
enum Sampler : int { kNearest = 0, kScipy = 1, kF64Lerp = 2, kF32Lerp = 3 };

// what maps an output pixel to a source coordinate
enum MapKind : int { kRadial = 0, kPersp = 1, kFused = 2 };

struct ImageArgs {
  const float* src;
  float* dst;
  int32_t H, W;            // output == source shape
  int32_t src_stride;      // elements between source rows
  int32_t src_col_stride;  // elements between source columns (1 on the fast path)
  uint32_t src_bytes;      // extent of the source for the buffer descriptor
  int32_t tiles_x, tiles_y;
  int32_t tile_rows;       // rows per workgroup
  int32_t xcd_remap;       // 1: contiguous band of tiles per XCD
  int32_t pipe_depth;      // rows of gathers in flight per thread (1, 2 or 4)
  int32_t lds_gather;      // 1: stage the source box of each wave tile in LDS (remap_lds_kernel)
  int32_t wg_box;          // 1: one box per workgroup (remap_wg_kernel) when the certificate allows it
  int32_t wg_per_cu;       // remap_wg_kernel: workgroups resident per CU (0 = as many as fit: 6)
  int32_t y_origin;        // a launch may cover only output rows [y_origin, y_origin + rows_out) of the H x W map;
  int32_t rows_out;        // dst then points at row y_origin (0 / 0 = the whole image)
  int32_t int_exact;       // integer element types: 1 = tiles at coordinates >= 32 take the exact factorised blend (exact_lerp_pairs)
  const int32_t* boxes = nullptr;   // remap_wg_kernel: corner hulls (x0, x1, y0, y1) of every 128 x 32 tile, frame-major, from box_table_kernel
                                    // (nullptr: every wave evaluates its workgroup's corners itself)
};

struct MapArgs {
  double xc, yc;
  double fact[kMaxFact];
  double coef[8];          // perspective c1..c8
  int32_t nfact;
  int32_t fast_div;        // 1: operands of the homography division stay in the normal range over the image
  int32_t tall_ok;         // host certificate for 64 x 32 tiles under an 80 x 56 box (sheared maps: remap_wg_color_kernel's second tile
                           // shape): the deviation bound holds for that tile and (nearly) every such tile's box fits
  int32_t tile_dev_ok;     // host certificate -- inside any output tile every source coordinate stays within 0.95 px of
                           // the bilinear interpolant of the tile's corner coordinates: 1 for 64 x 16 tiles, 2 also
                           // for 128 x 32 tiles (radial and perspective maps; api_core.cpp tile_deviation_certified)
};

struct StackArgs {
  const float* vol;
  float* out;
  int64_t proj_stride;     // elements between projections
  int32_t D, H, W;
  int32_t row_stride;      // elements between rows of a projection
  double row_start;
  int32_t nrows;
  int32_t d_chunk;         // projections walked by one thread
  uint32_t proj_bytes;
  int32_t rb0, rbh;        // the reference's row band [rb0, rb0 + rbh) of unwarp_chunk_slices_backward (postprocessing.py:289-312):
                           // a row coordinate outside it is reflected inside it, as scipy does with the cropped band.  rbh = 0:
                           // the host has shown that no coordinate can leave the band (or the call is not a chunk): no check
  int32_t int_exact;       // as ImageArgs::int_exact (stack_wg_kernel on integer element types)
  int32_t store_wait = 0;  // stack_wg_kernel: LaunchOpts::store_wait
  int32_t wg_per_cu = 0;   // stack_wg_kernel: LaunchOpts::wg_per_cu (1..3: workgroups per CU capped through unused dynamic LDS; A/B)
  int32_t xcd_order = 0;   // stack_wg_kernel: tiles dealt to the XCDs in contiguous runs (LaunchOpts::xcd_remap != 0)
};

struct CoordArgs {
  const void* ycoord;
  const void* xcoord;
  int64_t npts;
  int32_t is_f64;
  int32_t mode;            // BoundaryMode applied to coordinates OUTSIDE the image at orders 0 / 1 (kModeNearest = clamp)
};

// scipy boundary modes in the order of the reference's docstrings (postprocessing.py:128-130)
enum BoundaryMode : int { kModeReflect = 0, kModeGridMirror, kModeConstant, kModeGridConstant, kModeNearest, kModeMirror,
                          kModeGridWrap, kModeWrap };
enum SplineFilterKind : int { kSplMirror = 0, kSplReflect = 1, kSplWrap = 2 };

// element types of the *_typed entry points (DCP_DTYPE_* of include/discorpy_hip.h)
enum ElemType : int { kF32 = 0, kF64, kU8, kI8, kU16, kI16, kU32, kI32, kI64, kU64, kBool, kNumElemTypes };
__host__ __device__ inline int elem_size(int dtype) {
  return (dtype == kF64 || dtype == kI64 || dtype == kU64) ? 8 : (dtype == kF32 || dtype == kU32 || dtype == kI32) ? 4 : (dtype == kU16 || dtype == kI16) ? 2 : 1;
}
// numpy's bool_: one byte holding 0 or 1, read as a double and stored by a C cast of the double (scipy CASE_INTERP_OUT(NPY_BOOL))
struct Bool8 {
  uint8_t v;
  __host__ __device__ operator double() const { return (double)v; }
};

// orders 0 / 1 on any element type (typed_kernels.hip); strides in elements
struct TypedImageArgs {
  const void* src;
  void* dst;
  int64_t src_stride, src_cstride;
  int32_t H, W;
  int32_t order;           // 0 or 1
  int32_t dtype;
  int32_t blend;           // interleaved-channel kernel, float32: kF64Lerp = the one-ulp factorisation (else scipy's exact order)
  int32_t y0, rows;        // interleaved-channel kernel only: output rows [y0, y0 + rows), dst = first row of that band
};

struct TypedStackArgs {
  const void* vol;
  void* out;
  int64_t proj_stride, row_stride;
  int32_t D, H, W;
  int32_t nrows, d_chunk;
  int32_t dtype;
  int32_t out_f32;         // 1: out is float32 holding the value already converted to `dtype` (unwarp_slice_backward)
  int32_t round_f32;       // 1: coordinates rounded to float32 (unwarp_chunk_slices_backward)
  int32_t rb0, rbh;        // as in StackArgs
  double row_start;
};

struct SplineArgs {
  const void* src;
  int32_t src_dtype, dst_dtype;
  double* coef;            // (Hp x Wp) float64 workspace: padded image -> B-spline coefficients
  double* scratch;         // second plane of the same size (out-of-place filter passes, transposes)
  int32_t H, W, src_stride, src_cstride;
  int32_t Hp, Wp, pad;     // pad = 12 for 'nearest' / 'grid-constant', else 0
  int32_t order, mode, filter_kind, npoles;
  double poles[2];
  double zpow[2][2];       // [axis][pole]: z^n (reflect) or z^(n-1) (mirror), evaluated on the host
  int32_t exact_sum;       // 1: the taps are accumulated in scipy's order, t += (c wy) wx; 0: factorised and fused (the LDS-staged gather only)
  int32_t xcd_remap;       // spline_wg_kernel: 1 = every XCD owns a run of neighbouring tile columns (set by launch_spline; option "spline_xcd")
};

struct LaunchOpts {
  int tile_rows = 16;
  int xcd_remap = 2;
  int pipe_depth = 2;
  int lds_gather = 1;
  int coef_lds = 0;        // 1: force the LDS-staged coefficient path even for short vectors
  int d_chunk = 16;
  int stack_lds = 1;       // LDS-staged stack kernel: 0 never, 1 when the launch has enough wave tiles, 2 always
  int wg_box = 1;          // 1: remap_wg_kernel (one source box per 128 x 32 workgroup tile) for certified maps
  int wg_per_cu = 0;       // remap_wg_kernel: cap on resident workgroups per CU (0 = none)
  int store_wait = 1;      // stack_wg_kernel: 1 waits for the fill only (the stores of the previous projection stay in flight), 0 for everything
  int tall_tiles = 0;      // (A/B option, off: slower on config 5) sheared radial maps (level-1 certificate, MapArgs::tall_ok): 64 x 32 workgroup tiles (remap_wg_color_kernel) instead of per-wave boxes
  int int_exact = 1;       // integer element types: the exact factorised blend where it is provably exact (0: scipy's operation order everywhere)
  int stack_wg = 1;        // stack_wg_kernel (one box per workgroup, two slabs) for chunks of rows under a certified map: 0 never,
                           // 1 when the launch has enough workgroups (float32 and 8- / 16-bit integers), 2 whenever eligible
  int any_order = 0;       // 1: whole-frame launches leave with the AQL barrier bit cleared (hipExtAnyOrderLaunch): the packet may start
                           // while earlier packets of its stream still run (DCP_MEM_DEVICE_UNORDERED; the caller vouches for independence)
};

// launchers (unwarp_kernels.hip)
hipError_t launch_image(MapKind kind, const ImageArgs& img, const MapArgs& map, int sampler,
                        bool round_f32, const LaunchOpts& opts, hipStream_t stream);
// one frame of launch_image_batch (host side): its own source, destination, centre and coefficient vector
struct BatchFrame {
  const float* src;
  float* dst;
  double xc, yc;
  const double* fact;
};
// n dense float32 frames of ONE shape (img: H, W, src_stride, src_bytes; src / dst ignored), radial map, float32 coordinates, every
// frame with its own calibration of `nfact` <= 10 coefficients, through remap_wg_batch_kernel (blockIdx.z = frame) in
// ceil(n / 55) launches (35 per launch above 5 coefficients).  The caller has certified every frame at level 2
// (tile_deviation_certified).  *taken = false: the shape / options do not qualify, launch the frames one by one
hipError_t launch_image_batch(const ImageArgs& img, const BatchFrame* frames, int n, int nfact, int sampler, const LaunchOpts& opts,
                              hipStream_t stream, bool* taken);
// 8- and 16-bit integer images on remap_wg_kernel (img.src / dst reinterpreted, src_stride in elements, src_bytes the extent
// in bytes); *taken = false: the call does not qualify, use launch_typed_image
hipError_t launch_wg_typed(MapKind kind, const ImageArgs& img, const MapArgs& map, int order, int dtype, const LaunchOpts& opts,
                           hipStream_t stream, bool* taken);
hipError_t launch_stack_wg_typed(const StackArgs& st, const MapArgs& map, int dtype, const LaunchOpts& opts, hipStream_t stream, bool* taken);
hipError_t launch_coords(const ImageArgs& img, const CoordArgs& ca, int sampler, hipStream_t stream);
hipError_t launch_coord_map(MapKind kind, const ImageArgs& img, const MapArgs& map, float* ymap, float* xmap,
                            hipStream_t stream);
bool stack_wg_would_take(const StackArgs& st, const MapArgs& map, const LaunchOpts& opts);
hipError_t launch_stack(const StackArgs& st, const MapArgs& map, int sampler, bool round_f32,
                        const LaunchOpts& opts, hipStream_t stream);

// the rows of `st` under `ncentres` calibrations that differ in the centre only (map.xc / map.yc ignored), one launch per 224
// centres; st.out = (ncentres, D, nrows, W).  st.rbh must be 0 (no band check: the caller sends folding models call by call)
hipError_t launch_stack_centres(const StackArgs& st, const MapArgs& map, const double* xcs, const double* ycs, int ncentres, int sampler,
                                bool round_f32, const LaunchOpts& opts, hipStream_t stream);

hipError_t read_lds_stats(unsigned long long* out, bool reset);
// -DDCP_DEBUG_BOUNDS builds: LDS taps outside their slab (count, first byte offset, slab bytes, site) per translation unit; zeros otherwise
hipError_t read_bounds_unwarp(unsigned long long* out, bool reset);
hipError_t read_bounds_color(unsigned long long* out, bool reset);
hipError_t read_bounds_spline(unsigned long long* out, bool reset);
void set_last_kernel_name(const char* name);   // for the launchers of the other translation units
const char* last_kernel_name();   // unwarp_kernels.hip: the kernel the calling thread launched last (float32 image / stack launchers)
// spline_kernels.hip: map_kind 0 radial, 1 perspective, 2 explicit coordinates
void set_spline_wg(int v);      // 0: spline taps always from global memory (option "spline_wg")
void set_box_table(int v);      // option "box_table": remap_wg_kernel reads its tile hulls from box_table_kernel's table (0 never, 1 where it pays)
int get_box_table();
int get_spline_wg();
void set_spline_tiled(int v);   // 0: chunked prefilter passes + transposes even where the one-pass tiles qualify (option "spline_tiled")
int get_spline_tiled();
void set_pf2d_chunk(int v);     // rows per chunk of spline_prefilter2d_kernel (option "pf2d_chunk"; 0 = automatic)
int get_pf2d_chunk();
void set_pf2d_xcd(int v);       // 0: spline_prefilter2d_kernel's tiles in plain launch order (option "pf2d_xcd")
int get_pf2d_xcd();
void set_pf2d_two_pole(int v);  // 0: the two-pole spline orders (4, 5) on spline_tile_filter_kernel, one launch per axis (option "pf2d_two_pole")
int get_pf2d_two_pole();
void set_spline_xcd(int v);     // 0: spline_wg_kernel's tiles in plain launch order (option "spline_xcd")
int get_spline_xcd();
hipError_t launch_spline(const SplineArgs& a, int map_kind, const MapArgs& map, const CoordArgs& ca, void* dst,
                         hipStream_t stream);
// typed_kernels.hip: map_kind 0 radial, 1 perspective, 2 fused, 3 explicit coordinates
hipError_t launch_typed_image(int map_kind, const TypedImageArgs& img, const MapArgs& map, const CoordArgs& ca,
                              hipStream_t stream);
hipError_t launch_typed_stack(const TypedStackArgs& st, const MapArgs& map, hipStream_t stream);
// n points (y, x) -> centre + B(r) (p - centre), float64
hipError_t launch_map_points(const double* yx_in, double* yx_out, int64_t n, const MapArgs& map, hipStream_t stream);
// n points (y, x) through the homography map.coef (numpy's operation order), float64
hipError_t launch_map_points_persp(const double* yx_in, double* yx_out, int64_t n, const MapArgs& map, hipStream_t stream);
// interleaved (H, W, C) image, radial map, orders 0 / 1; src_cstride = elements between pixels
hipError_t launch_typed_channels(const TypedImageArgs& img, const MapArgs& map, int channels, hipStream_t stream);
// color_kernels.hip: the same on remap_wg_kernel's data path (3 / 4 dense channels of float32 / uint8 / uint16, certified radial map);
// img as for launch_wg_typed with src_col_stride = channels; *taken = false: does not qualify, use launch_typed_channels
hipError_t launch_color(const ImageArgs& img, const MapArgs& map, int channels, int dtype, int sampler, const LaunchOpts& opts, hipStream_t stream,
                        bool* taken);

}  // namespace dcp
