// unwarp_kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels for the backward
// unwarp path of discorpy (reference: /root/reference/discorpy/post/postprocessing.py).
//
//   remap_wg_kernel / remap_lds_kernel / remap_tile_kernel <Radial>  K1  unwarp_image_backward      postprocessing.py:137-148
//   remap_wg_kernel / remap_lds_kernel / remap_tile_kernel <Persp>   K2  correct_perspective_image  postprocessing.py:448-459,486-492
//   remap_lds_kernel / remap_tile_kernel <Fused>   K3  perspective o radial in one pass
//   stack_wg_kernel / stack_lds_kernel / stack_rows_kernel   K4  unwarp_slice_backward / unwarp_chunk_slices_backward  :211-229,:281-313
//   remap_coords_kernel         K5  map_index= / _mapping            postprocessing.py:250-251,489-491
//   coord_map_kernel            K6  the float32 (yd, xd) planes      postprocessing.py:144-145,444-459
//
// Design (see DESIGN.md section 4 for the selection table and the numbers):
//  * one thread per output pixel column, 64-wide wavefronts along x: a wave's store is 256 B
//    contiguous.  The order-1 frame kernels stage the source in LDS: remap_wg_kernel (certified maps --
//    the host bounds how far a tile's coordinates can stray from the bilinear interpolant of its corner
//    pixels) gives a workgroup of four waves a 128 x 32 output tile and ONE source box, predicted from
//    the tile's four corner pixels and copied with row-contiguous 16-byte LDS-DMA loads that go out
//    between the coordinate rows; remap_lds_kernel gives every wave its own 64 x 16 tile and box
//    (certified, or with every pixel verified against the box: the fused map and uncertified models).
//    The taps of every pixel then come from LDS.  The direct kernel (remap_tile_kernel) gathers tap
//    pairs from global memory with 8-byte buffer loads; it serves float64 coordinates, strided /
//    degenerate sources and the fallback.  stack_wg_kernel does the same for the rows of a stack, with
//    two slabs: the box of projection d + 1 streams in while projection d is blended.
//  * the coordinate polynomial is evaluated in fp64 (the reference computes float64 and only
//    then rounds to float32, postprocessing.py:144-145; an fp32 evaluation changes 34 % of the
//    coordinates by one ulp).  Everything that does not depend on x is staged once per
//    workgroup in LDS (yu, yu^2 per row; c*y products for the homography); everything that
//    does not depend on y lives in registers across the row loop; coefficients are SGPR
//    resident (kernarg) for vectors of <= 10 terms and staged in LDS for longer ones.
//  * sqrt is the correctly rounded fp64 result built from v_rsq_f64 + one coupled Newton
//    step + one residual correction (tools/ubench.hip); the two homography divisions share one
//    refined reciprocal and are correctly rounded too (tools/ubench_div.hip).
//  * every global access goes through a buffer descriptor: a 32-bit byte offset per lane, row
//    steps through the scalar offset, hardware bounds checking.
//  * no MFMA: the op is a remap (8 B of HBM traffic per pixel), not a contraction.
//
// Build with -ffp-contract=off: every fused multiply-add below is written explicitly so that
// the arithmetic is the same sequence of IEEE operations as oracle/unwarp_oracle.c.
#include "dcp_internal.h"
#include "dcp_device.h"
#include "dcp_lab.h"
#include <type_traits>
#include <cstdio>
#include <cstring>

namespace dcp {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;
#ifndef DCP_PIPE_DEPTH
#define DCP_PIPE_DEPTH 2
#endif
constexpr int kPipeDepth = DCP_PIPE_DEPTH;
#ifndef DCP_LDS_UNTRACKED_DMA
#define DCP_LDS_UNTRACKED_DMA 1  // 0: remap_lds_kernel's fill through the compiler's LDS-DMA builtin (rounds 1-3; A/B)
#endif
#ifndef DCP_FILL_AUX
#define DCP_FILL_AUX 0   // cache-policy bits of remap_wg_kernel's LDS-DMA fill (2 = nt)
#endif
#ifndef DCP_ROWS_NT_STORE
#define DCP_ROWS_NT_STORE 1   // stack_rows_body: nt stores (0: plain, rounds 1-3 -- the 121-centre grid 104 -> 87 us, one sinogram 4.0 -> 3.7)
#endif
#ifndef DCP_STORE_AUX
#define DCP_STORE_AUX 2  // cache-policy bits of the output store: 2 = nt (streamed once, never re-read here)
#endif

// ------------------------------------------------------------------ sampler

struct SrcView {
  __amdgpu_buffer_rsrc_t rsrc;
  int32_t W, H;
  int32_t stride;      // elements
  int32_t cstride;     // elements
};

// One bilinear (or nearest) fetch in flight: the issued loads plus the two fractions.
// PAIR: the x0/x1 taps of a row are one 8-byte load (needs unit column stride, W,H >= 2).
template <int SAMPLER, bool PAIR, typename CT>
struct Fetch {
  u32x2 a, b;          // PAIR: rows y0 / y0+1.  !PAIR: a = (v00, v01), b = (v10, v11)
  CT fx, fy;
};

// Coordinates are already inside [0, W-1] x [0, H-1].  The base tap is held at x0 <= W-2,
// y0 <= H-2: at the far edge the fraction becomes 1 on (len-2, len-1) instead of scipy's 0 on
// (len-1, folded len-1) -- the same value for finite data, and no out-of-range tap exists.
template <int SAMPLER, bool PAIR, typename CT>
__device__ __forceinline__ Fetch<SAMPLER, PAIR, CT> fetch(const SrcView& s, CT xc, CT yc) {
  Fetch<SAMPLER, PAIR, CT> f;
  int xi = (int)xc, yi = (int)yc;   // truncation == floor for non-negative coordinates
  if constexpr (SAMPLER == kNearest) {
    // order 0: index = floor(c + 0.5)  (round half up, not rint)
    xi += (xc - (CT)xi >= (CT)0.5) ? 1 : 0;
    yi += (yc - (CT)yi >= (CT)0.5) ? 1 : 0;
    uint32_t off;
    if constexpr (PAIR) off = (__umul24(yi, s.stride) + (uint32_t)xi) << 2;
    else off = ((uint32_t)yi * (uint32_t)s.stride + (uint32_t)xi * (uint32_t)s.cstride) * 4u;
    f.a.x = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, off, 0, 0);
    f.fx = f.fy = (CT)0;
    return f;
  } else if constexpr (PAIR) {
    xi = min(xi, s.W - 2);
    yi = min(yi, s.H - 2);
    f.fx = xc - (CT)xi;             // exact
    f.fy = yc - (CT)yi;
    const uint32_t off = (__umul24(yi, s.stride) + (uint32_t)xi) << 2;
    f.a = __builtin_amdgcn_raw_buffer_load_b64(s.rsrc, off, 0, 0);
    f.b = __builtin_amdgcn_raw_buffer_load_b64(s.rsrc, off, s.stride * 4, 0);
    return f;
  } else {
    // any stride, any size: same base-tap rule per axis, four 4-byte loads
    xi = min(xi, max(s.W - 2, 0));
    yi = min(yi, max(s.H - 2, 0));
    f.fx = xc - (CT)xi;
    f.fy = yc - (CT)yi;
    const uint32_t dx = s.W >= 2 ? (uint32_t)s.cstride * 4u : 0u;
    const uint32_t dy = s.H >= 2 ? (uint32_t)s.stride * 4u : 0u;
    const uint32_t off = ((uint32_t)yi * (uint32_t)s.stride + (uint32_t)xi * (uint32_t)s.cstride) * 4u;
    f.a.x = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, off, 0, 0);
    f.a.y = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, off + dx, 0, 0);
    f.b.x = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, off + dy, 0, 0);
    f.b.y = __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, off + dy + dx, 0, 0);
    return f;
  }
}

// The three order-1 arithmetics (same as sample() in oracle/unwarp_oracle.c).
// EDGE = false: the caller knows both fractions are below 1 (tiles strictly inside the image)
template <int SAMPLER, bool PAIR, typename CT, bool EDGE = true>
__device__ __forceinline__ float finish(const Fetch<SAMPLER, PAIR, CT>& f) {
  const float v00 = __uint_as_float(f.a.x), v01 = __uint_as_float(f.a.y);
  const float v10 = __uint_as_float(f.b.x), v11 = __uint_as_float(f.b.y);
  if constexpr (SAMPLER == kNearest) {
    return v00;
  } else if constexpr (SAMPLER == kScipy) {
    // scipy NI_GeometricTransform: w0 = 1 - f, w1 = 1 - w0; ((v*wy)*wx) summed left to right.
    // At the far edge the gather holds the base tap at len - 2 and the fraction is 1, where scipy's taps are (len - 1,
    // len - 1 folded) with fraction 0: the same value for finite data, but scipy's zero-weight tap is the edge pixel itself.
    // Take it too, so that a non-finite pixel at len - 2 does not reach a clipped coordinate (0 * NaN).
    float v00 = __uint_as_float(f.a.x), v01 = __uint_as_float(f.a.y);
    float v10 = __uint_as_float(f.b.x), v11 = __uint_as_float(f.b.y);
    if constexpr (EDGE) {
      if (f.fx == (CT)1) {
        v00 = v01;
        v10 = v11;
      }
      if (f.fy == (CT)1) {
        v00 = v10;
        v01 = v11;
      }
    }
    const double fx = (double)f.fx, fy = (double)f.fy;
    const double wy0 = 1.0 - fy, wy1 = 1.0 - wy0;
    const double wx0 = 1.0 - fx, wx1 = 1.0 - wx0;
    double acc = ((double)v00 * wy0) * wx0;
    acc += ((double)v01 * wy0) * wx1;
    acc += ((double)v10 * wy1) * wx0;
    acc += ((double)v11 * wy1) * wx1;
    return (float)acc;
  } else if constexpr (SAMPLER == kF64Lerp) {
    const double fx = (double)f.fx, fy = (double)f.fy;
    const double a = (double)v00, b = (double)v01, c = (double)v10, d = (double)v11;
    const double top = __builtin_fma(fx, b - a, a);
    const double bot = __builtin_fma(fx, d - c, c);
    return (float)__builtin_fma(fy, bot - top, top);
  } else {
    const float fx = (float)f.fx, fy = (float)f.fy;
    const float top = __builtin_fmaf(fx, v01 - v00, v00);
    const float bot = __builtin_fmaf(fx, v11 - v10, v10);
    return __builtin_fmaf(fy, bot - top, top);
  }
}

// ------------------------------------------------------------------ tile bookkeeping

// blockIdx.x -> tile (tx, ty).  The dispatcher places workgroup b on XCD b % 8 (observed, used for
// speed only).  xcd_remap:
//   0  row-major tile order: horizontally adjacent tiles land on different XCDs
//   1  each XCD gets a contiguous run of row-major tiles (a horizontal band of the image)
//   2  each XCD gets a vertical stripe of tile columns and sweeps it row by row: neighbours in x
//      and y share an L2 (their source boxes overlap), while the eight XCDs together still touch
//      one full-width band of rows at a time (even spread over the HBM channels)
__device__ __forceinline__ void logical_tile(int xcd_remap, int tiles_x, int* tx, int* ty) {
  const int b = blockIdx.x;
  if (xcd_remap == 0) {
    *ty = b / tiles_x;
    *tx = b - *ty * tiles_x;
    return;
  }
  const int nb = gridDim.x;
  const int per = nb >> 3, rem = nb & 7;
  const int xcd = b & 7, j = b >> 3;
  const int lin = xcd * per + min(xcd, rem) + j;     // position in the XCD-contiguous enumeration
  if (xcd_remap == 1) {
    *ty = lin / tiles_x;
    *tx = lin - *ty * tiles_x;
    return;
  }
  // enumeration: stripe by stripe; inside a stripe row by row.  Stripe s spans columns
  // [s*wq + min(s, wr), ...) with width wq + (s < wr), wq = tiles_x / 8, wr = tiles_x % 8.
  const int tiles_y = nb / tiles_x;
  const int wq = tiles_x >> 3, wr = tiles_x & 7;
  const int big = (wq + 1) * tiles_y;                 // tiles in one of the first wr (wider) stripes
  int s, off;
  if (lin < wr * big) {
    s = lin / big;
    off = lin - s * big;
  } else {
    const int l2 = lin - wr * big;
    const int small = max(wq, 1) * tiles_y;
    s = wr + l2 / small;
    off = l2 - (s - wr) * small;
  }
  const int w = wq + (s < wr ? 1 : 0);
  const int x0 = s * wq + min(s, wr);
  *ty = off / w;
  *tx = x0 + (off - *ty * w);
}

__device__ __forceinline__ SrcView make_view(const float* base, uint32_t bytes, int W, int H, int stride,
                                             int cstride) {
  SrcView s;
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  s.W = W;
  s.H = H;
  s.stride = stride;
  s.cstride = cstride;
  return s;
}

// ------------------------------------------------------------------ K1 / K2 / K3


// (ColCtx / make_col / fill_row / map_coord -- the hoisted evaluation of the coordinate maps -- live in dcp_device.h: the spline
// gather shares them)

// NF >= 0: polynomial length known at compile time, coefficients read from kernarg (SGPRs).
// NF == -1: runtime length, coefficients staged in LDS.
// PD: rows in flight per thread (software pipeline depth of the gather).
template <int KIND, int NF, int SAMPLER, bool ROUND32, bool PAIR, int PD>
__global__ void __launch_bounds__(kBlock) remap_tile_kernel(const ImageArgs img, const MapArgs map) {
  __shared__ double s_row[kMaxTileRows + PD][4];
  __shared__ double s_coef[kMaxFact];
  using CT = typename std::conditional<ROUND32, float, double>::type;
  using FetchT = Fetch<SAMPLER, PAIR, CT>;

  int tx, ty;
  logical_tile(img.xcd_remap, img.tiles_x, &tx, &ty);
  const int y0 = ty * img.tile_rows;
  const int x = tx * kBlock + (int)threadIdx.x;
  const int rows = min(img.tile_rows, img.rows_out - y0);     // y0: row inside the launch's band of output rows

  // per-row invariants -> LDS.  PD rows past the tile are filled too: the pipeline below runs
  // ahead by PD rows and simply discards what it computed for them.
  if ((int)threadIdx.x < img.tile_rows + PD)
    fill_row<KIND, 4>(map, s_row, threadIdx.x, (double)(img.y_origin + y0 + (int)threadIdx.x));
  if constexpr (NF < 0 && KIND != kPersp) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
  }
  __syncthreads();
  if (x >= img.W) return;

  const SrcView src = make_view(img.src, img.src_bytes, img.W, img.H, img.src_stride, img.src_col_stride);
  const float wmaxf = (float)(img.W - 1), hmaxf = (float)(img.H - 1);
  const double wmaxd = (double)(img.W - 1), hmaxd = (double)(img.H - 1);
  // output rows through one buffer descriptor per row built on the scalar unit: lane offset
  // x*4 in a VGPR; a row past the image gets num_records = 0, i.e. the store is dropped
  const uint32_t row_bytes_out = (uint32_t)img.W * 4u;
  const char* out_base = (const char*)(img.dst + (size_t)y0 * (size_t)img.W);
  const uint32_t xoff = (uint32_t)x * 4u;
  const ColCtx col = make_col<KIND, NF>(map, x);

  // source coordinate of row k of the tile and the gather it starts
  auto issue = [&](int k) -> FetchT {
    double xd, yd;
    map_coord<KIND, NF, 4>(map, s_row, s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
    if constexpr (ROUND32) {
      return fetch<SAMPLER, PAIR, float>(src, round_clip_f32(xd, wmaxf), round_clip_f32(yd, hmaxf));
    } else {
      return fetch<SAMPLER, PAIR, double>(src, clip_f64(xd, wmaxd), clip_f64(yd, hmaxd));
    }
  };

  // software pipeline: PD rows of gathers in flight per thread
  FetchT q[PD];
#pragma unroll
  for (int j = 0; j < PD; ++j) q[j] = issue(j);
  for (int k = 0; k < rows; k += PD) {
#pragma unroll
    for (int j = 0; j < PD; ++j) {
      const float v = finish<SAMPLER, PAIR, CT>(q[j]);
      const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(out_base + (size_t)(k + j) * row_bytes_out), 0, (k + j < rows) ? (int)row_bytes_out : 0,
          0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst, xoff, 0, DCP_STORE_AUX);
      q[j] = issue(k + j + PD);
    }
  }
}

// ------------------------------------------------------------------ K1 / K2 / K3, LDS-staged gather

// The same maps as remap_tile_kernel, float32-rounded coordinates, order 1, unit column stride.
// Each WAVE owns a 64-wide x 16-tall output tile:
//   phase 1a every lane evaluates the source coordinates of the first and last row of its column;
//            the four corner pixels of the tile predict the bounding box of all its taps;
//   fill     the wave starts copying that box -- at most kBoxW x kBoxH floats -- from HBM/L2 into
//            its own LDS slab with row-contiguous 4-byte-per-lane LDS-DMA loads (one 256 B line
//            pair per instruction, the access shape the vector L1 serves fastest);
//   phase 1b the other 14 rows are evaluated while the fill is in flight, and every lane checks
//            that all its taps lie inside the predicted box (one ballot);
//   phase 2  every lane gathers its four taps from LDS (two ds_read2_b32), blends and stores.
// A box that does not fit (strong minification or skew) or a failed containment vote makes the
// wave fall back to the direct global gather for that tile.  No workgroup barrier is needed after the row table is built:
// slabs are wave-private and a wave's LDS operations execute in order.
// statistics of the staged path (bumped only when a wave leaves it): [0] box did not fit,
// [1] containment vote failed.  Read through dcp_debug_counters().
__device__ unsigned long long g_lds_stats[2];

#ifndef DCP_LDS_TH
#define DCP_LDS_TH 16
#endif
constexpr int kLdsTW = 64, kLdsTH = DCP_LDS_TH;
#ifndef DCP_BOXW
#define DCP_BOXW 80
#endif
#ifndef DCP_BOXH
#define DCP_BOXH 24
#endif
#ifndef DCP_LDS_MARGIN
#define DCP_LDS_MARGIN 0   // extra pixels around the corner hull (0: ~1 tile per 16384 fails the vote on cfg2)
#endif
#ifndef DCP_LDS_WAVES
#define DCP_LDS_WAVES 5
#endif
#ifndef DCP_LDS_BLOCK_WAVES
#define DCP_LDS_BLOCK_WAVES 4   // wave tiles (64 x 16 pixels, stacked vertically) per workgroup
#endif
constexpr int kBoxW = DCP_BOXW, kBoxH = DCP_BOXH;
constexpr int kLdsBW = DCP_LDS_BLOCK_WAVES;

// VOTE = false: the host has certified (MapArgs::tile_dev_ok, api_core.cpp: tile_deviation_certified) that inside any
// 64 x 16 tile every source coordinate stays within one pixel of the bilinear interpolant of the tile's four corner
// coordinates.  The box is then the corner hull grown by one pixel on every side, it contains every tap by
// construction, and the per-pixel containment vote (two min3/max3 per pixel and a ballot) is not compiled in.
// VOTE = true: no certificate (fused map, strongly curved models): zero margin, every pixel verified.
// (the fused map with runtime-length coefficients needs > 96 VGPRs: three waves per SIMD is what it gets, and what it declares)
template <int KIND, int NF, int SAMPLER, bool VOTE>
__global__ void __launch_bounds__(64 * kLdsBW, NF < 0 ? 3 : DCP_LDS_WAVES) remap_lds_kernel(const ImageArgs img, const MapArgs map) {
  constexpr int kMargin = VOTE ? DCP_LDS_MARGIN : 1;
  // 4 x 7680 B slabs + row table + coefficients <= 32 KB: five workgroups (20 waves) per CU
  __shared__ float s_box[kLdsBW][kBoxH * kBoxW];
  __shared__ double s_row[kLdsBW * kLdsTH][KIND == kRadial ? 2 : 4];
  __shared__ double s_coef[NF < 0 ? kMaxFact : 1];
  using FetchT = Fetch<SAMPLER, true, float>;

  // the wave index is wave-uniform by construction; say so, or every descriptor derived from it
  // is treated as divergent and wrapped in a waterfall loop
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  DCP_TRACE_WAVE_BEGIN(((unsigned)blockIdx.y * gridDim.x + blockIdx.x) * kLdsBW + (unsigned)wave);
  int tx, ty;
  if (img.xcd_remap == 2) {
    // stripe order without divisions: a 2-D grid whose x extent is 8 * (widest stripe); the
    // linear workgroup id is y * gridDim.x + x, so XCD = blockIdx.x & 7 owns stripe blockIdx.x & 7
    // and all stripes sweep the image row by row together.  Narrower stripes have a few idle
    // workgroups at their right end.
    const int s = blockIdx.x & 7, c = blockIdx.x >> 3;
    const int wq = img.tiles_x >> 3, wr = img.tiles_x & 7;
    if (c >= wq + (s < wr ? 1 : 0)) return;
    tx = s * wq + min(s, wr) + c;
    ty = blockIdx.y;
  } else {
    logical_tile(img.xcd_remap, img.tiles_x, &tx, &ty);
  }
  const int yblk = ty * (kLdsBW * kLdsTH);
  const int y0 = yblk + wave * kLdsTH;            // first row of this wave's tile
  const int x = tx * kLdsTW + lane;

  if ((int)threadIdx.x < kLdsBW * kLdsTH)
    fill_row<KIND, (KIND == kRadial ? 2 : 4)>(map, s_row, threadIdx.x,
                                              (double)(img.y_origin + min(yblk + (int)threadIdx.x, img.rows_out - 1)));
  if constexpr (NF < 0 && KIND != kPersp) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
  }
  __syncthreads();
  DCP_TRACE(1);
  if (y0 >= img.rows_out) return;                 // whole wave past the band of output rows (wave-uniform)
  const int rows = min(kLdsTH, img.rows_out - y0);

  const SrcView src = make_view(img.src, img.src_bytes, img.W, img.H, img.src_stride, 1);
  const float wmaxf = (float)(img.W - 1), hmaxf = (float)(img.H - 1);
  const ColCtx col = make_col<KIND, NF>(map, min(x, img.W - 1));
  const auto* rowtab = s_row + wave * kLdsTH;

  float* box = s_box[wave];
  const uint32_t row_bytes_out = (uint32_t)img.W * 4u;
  const char* out_base = (const char*)(img.dst + (size_t)y0 * (size_t)img.W);
  const uint32_t xoff = (uint32_t)x * 4u;         // lanes with x >= W: the store is out of range and dropped
  const __amdgpu_buffer_rsrc_t dst =
      __builtin_amdgcn_make_buffer_rsrc((void*)out_base, 0, (int)((uint32_t)rows * row_bytes_out), 0x00020000);

  // ---- phase 1a: first and last row of the tile; their end lanes are the tile's four corners
  float xf[kLdsTH], yf[kLdsTH];
  auto eval_row = [&](int k) {
    double xd, yd;
    map_coord<KIND, NF, (KIND == kRadial ? 2 : 4)>(map, rowtab, s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
    xf[k] = round_clip_f32(xd, wmaxf);
    yf[k] = round_clip_f32(yd, hmaxf);
  };
  eval_row(0);
  eval_row(kLdsTH - 1);

  // ---- predicted source box: hull of the corner taps grown by one pixel.  The map is smooth, so
  // the taps of the other 1020 pixels stay inside it (the bulge of a 64-pixel arc is ~0.01 px);
  // phase 1b verifies that for every pixel and the wave falls back to the direct gather if not.
  int cx0, cx1, cy0, cy1;
  {
    const int xa = (int)xf[0], xb = (int)xf[kLdsTH - 1], ya = (int)yf[0], yb = (int)yf[kLdsTH - 1];
    const int xa0 = __builtin_amdgcn_readlane(xa, 0), xa1 = __builtin_amdgcn_readlane(xa, 63);
    const int xb0 = __builtin_amdgcn_readlane(xb, 0), xb1 = __builtin_amdgcn_readlane(xb, 63);
    const int ya0 = __builtin_amdgcn_readlane(ya, 0), ya1 = __builtin_amdgcn_readlane(ya, 63);
    const int yb0 = __builtin_amdgcn_readlane(yb, 0), yb1 = __builtin_amdgcn_readlane(yb, 63);
    cx0 = min(min(xa0, xa1), min(xb0, xb1));
    cx1 = max(max(xa0, xa1), max(xb0, xb1));
    cy0 = min(min(ya0, ya1), min(yb0, yb1));
    cy1 = max(max(ya0, ya1), max(yb0, yb1));
  }
  const int bx0 = max(min(cx0 - kMargin, img.W - 2), 0);
  const int bx1 = min(cx1 + 1 + kMargin, img.W - 1);   // last column held: tap x0+1 of the largest x0 (+ margin)
  const int by0 = max(min(cy0 - kMargin, img.H - 2), 0);
  const int by1 = min(cy1 + 1 + kMargin, img.H - 1);
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  const bool fits = bw <= kBoxW && bh <= kBoxH;
  DCP_TRACE(2);

  // ---- fill: box rows by0..by1 from column bx0, row-contiguous loads that land directly in LDS
  // (buffer_load ... lds, no VGPR round trip).  The row group advances through the scalar offset.
  // All rows are in flight at once and complete underneath phase 1b.  (The slab is filled to its
  // full 80-float pitch: up to 12 columns more than the box needs, from cache lines the
  // neighbouring tile fetches anyway.)
  if (fits) {
    // 16 bytes per lane: 20 lanes cover one slab row (pitch 80 floats), so one instruction fills
    // three consecutive box rows (lanes 0-59) and the LDS image stays lane-linear as LDS-DMA needs
    const uint32_t org = ((uint32_t)by0 * (uint32_t)img.src_stride + (uint32_t)bx0) * 4u;
    const uint32_t rstep = (uint32_t)img.src_stride * 4u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    static_assert(kBoxW == 80, "the fill maps 20 lanes of 16 bytes to one slab row");
    const int lrow = (int)(__umul24((uint32_t)lane, 13u) >> 8);        // lane / 20 for lane < 64
    const int lcol = lane - lrow * 20;
    const uint32_t voff = (uint32_t)lrow * rstep + (uint32_t)lcol * 16u;   // full 32-bit product: a row stride may exceed 2^24 bytes
    [[maybe_unused]] const dcp_rsrc_words src_words = raw_rsrc_words(img.src, img.src_bytes);
    [[maybe_unused]] const uint32_t box_lds = (uint32_t)(uintptr_t)(lds_ptr)box;
    if (lane < 60) {
      // the whole offset goes through the VGPR so that the descriptor's bounds check sees it: the slab pitch can reach past the last
      // image column / the end of the source buffer, where the load must return zeros instead of touching memory.
      // Radial and perspective maps: hidden from the compiler, which otherwise waits for the fill in front of the first row-table
      // read of phase 1b on the unclipped path -- see lds_dma16_untracked; the wave's own wait is the explicit one below (cfg5
      // 115.3 -> 113.5 us).  The fused map, whose instantiation has no such wait, is 3 % slower that way: it keeps the builtin
      auto three_rows = [&](int r) {
        if (r + lrow < bh) {
          if constexpr (DCP_LDS_UNTRACKED_DMA && KIND != kFused)
            lds_dma16_untracked(src_words, box_lds + (uint32_t)(r * kBoxW * 4), voff + (org + (uint32_t)r * rstep));
          else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src.rsrc, (lds_ptr)(box + r * kBoxW), 16, voff + (org + (uint32_t)r * rstep), 0, 0, 0);
        }
      };
      if constexpr (DCP_LDS_UNTRACKED_DMA && KIND != kFused) {
        for (int r = 0; r < bh; r += 6) {       // (two loads per turn, by hand: `#pragma unroll` does not unroll around the asm)
          three_rows(r);
          three_rows(r + 3);
        }
      } else {
#pragma unroll 2
        for (int r = 0; r < bh; r += 3) three_rows(r);
      }
    }
  }

  DCP_TRACE(3);
  // ---- phase 1b: the other rows, and this lane's extremes for the containment vote.
  // When the predicted box lies strictly inside the image the clip is skipped: a coordinate that
  // would have been clipped is then outside the box, fails the vote below and the fallback path
  // clips it.  (Border tiles keep the clip: their clipped coordinates are legitimate box members.)
  const bool box_inside = cx0 - kMargin >= 0 && cx1 + 1 + kMargin <= img.W - 1 &&
                          cy0 - kMargin >= 0 && cy1 + 1 + kMargin <= img.H - 1;
  float xmn = 0.f, xmx = 0.f, ymn = 0.f, ymx = 0.f;
  if constexpr (VOTE) {
    xmn = __builtin_fminf(xf[0], xf[kLdsTH - 1]), xmx = __builtin_fmaxf(xf[0], xf[kLdsTH - 1]);
    ymn = __builtin_fminf(yf[0], yf[kLdsTH - 1]), ymx = __builtin_fmaxf(yf[0], yf[kLdsTH - 1]);
  }
  auto rows_1b = [&](auto noclip, auto fastdiv) {
#pragma unroll
    for (int k = 1; k < kLdsTH - 1; ++k) {
      double xd, yd;
      map_coord<KIND, NF, (KIND == kRadial ? 2 : 4), decltype(fastdiv)::value>(map, rowtab, s_coef, col, k, wmaxf,
                                                                               hmaxf, &xd, &yd);
      if constexpr (decltype(noclip)::value) {
        xf[k] = (float)xd;
        yf[k] = (float)yd;
      } else {
        xf[k] = round_clip_f32(xd, wmaxf);
        yf[k] = round_clip_f32(yd, hmaxf);
      }
      if constexpr (VOTE) {
        xmn = __builtin_fminf(xmn, xf[k]);
        xmx = __builtin_fmaxf(xmx, xf[k]);
        ymn = __builtin_fminf(ymn, yf[k]);
        ymx = __builtin_fmaxf(ymx, yf[k]);
      }
    }
  };
  const bool unclipped = box_inside && fits;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (KIND == kRadial || !map.fast_div) {
    if (unclipped) rows_1b(std::true_type{}, I0{});
    else rows_1b(std::false_type{}, I0{});
  } else {
    if (unclipped) rows_1b(std::true_type{}, I1{});
    else rows_1b(std::false_type{}, I1{});
  }
  // tap columns are min(floor(xf), W-2) and +1: inside [bx0, bx1] iff xf >= bx0 and (xf < bx1 or
  // the coordinate was clipped and the box ends at the image edge)
  bool staged = fits;
  if constexpr (VOTE) {
    const bool inside = xmn >= (float)bx0 && (xmx < (float)bx1 || (!unclipped && bx1 == img.W - 1)) &&
                        ymn >= (float)by0 && (ymx < (float)by1 || (!unclipped && by1 == img.H - 1));
    staged = fits && __builtin_amdgcn_ballot_w64(!inside) == 0;
  }
  if (!staged && lane == 0) atomicAdd(&g_lds_stats[fits ? 1 : 0], 1ull);
  DCP_TRACE(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  DCP_TRACE(5);
  // one descriptor for the rows of the tile that exist; the row offset goes through the scalar
  // offset (not bounds-checked, hence the explicit row and column predicates)
  if (!(x < img.W)) return;                      // no cross-lane work from here on
  if (staged) {
    // ---- phase 2: taps from LDS, blend, store.  Byte address of tap (y0, x0) in the slab:
    // (y0 - by0) * pitch + (x0 - bx0) floats = y0 * (4 pitch) + (4 x0 - 4 (by0 pitch + bx0))
    // the slab's own byte offset inside the workgroup's LDS is folded into the origin term (one scalar add per tile
    // instead of a vector add per pixel)
    const uint32_t negorg4 = (uint32_t)(-(by0 * kBoxW + bx0) * 4) + (uint32_t)(wave * (kBoxH * kBoxW * 4));
    const char* boxb = (const char*)&s_box[0][0];
    // A box that ends before the last image column/row cannot contain the clipped coordinate
    // W-1 / H-1, so interior tiles need no base-tap clamp: floor and fraction are one op each.
    const bool interior = bx1 < img.W - 1 && by1 < img.H - 1;
    // interior tiles form the address in float32 (the only VALU ops that issue a wave in 2 cycles instead of 4):
    // floor = c - fract(c) exactly, address = floor_y * 4 pitch + floor_x * 4 + origin, every term an integer
    // below 2^24 (launch_lds checks H and W), one conversion at the end
    const float negorg4f = (float)(int32_t)negorg4;
    auto tile_rows_loop = [&](auto full, auto inner) {
#pragma unroll
      for (int k = 0; k < kLdsTH; ++k) {
        FetchT f;
        uint32_t addr;
        if constexpr (SAMPLER == kNearest) {
          // order 0: index = floor(c + 0.5); it is one of the two bilinear tap indices, so it lies inside the box
          int xi = (int)xf[k];
          int yi = (int)yf[k];
          xi += (xf[k] - (float)xi >= 0.5f) ? 1 : 0;
          yi += (yf[k] - (float)yi >= 0.5f) ? 1 : 0;
          uint32_t xa;
          asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(xa) : "v"(xi), "s"(negorg4));
          asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(yi), "s"(kBoxW * 4), "v"(xa));
        } else if constexpr (decltype(inner)::value) {
          f.fx = __builtin_amdgcn_fractf(xf[k]);      // x - floor(x), exact
          f.fy = __builtin_amdgcn_fractf(yf[k]);
          const float flx = xf[k] - f.fx, fly = yf[k] - f.fy;
          const float af = __builtin_fmaf(fly, (float)(kBoxW * 4), __builtin_fmaf(flx, 4.0f, negorg4f));
          addr = (uint32_t)(int32_t)af;
        } else {
          const int xi = min((int)xf[k], img.W - 2);
          const int yi = min((int)yf[k], img.H - 2);
          f.fx = xf[k] - (float)xi;
          f.fy = yf[k] - (float)yi;
          uint32_t xa;
          asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(xa) : "v"(xi), "s"(negorg4));
          asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(yi), "s"(kBoxW * 4), "v"(xa));
        }
        DCP_BOUNDS(addr - (uint32_t)wave * (uint32_t)(kBoxH * kBoxW * 4), SAMPLER == kNearest ? 4 : kBoxW * 4 + 8, kBoxH * kBoxW * 4, 1);
        const float* t = (const float*)(boxb + addr);
        float v;
        if constexpr (SAMPLER == kNearest) {
          v = t[0];
        } else {
          f.a.x = __float_as_uint(t[0]);
          f.a.y = __float_as_uint(t[1]);
          f.b.x = __float_as_uint(t[kBoxW]);
          f.b.y = __float_as_uint(t[kBoxW + 1]);
          v = finish<SAMPLER, true, float, !decltype(inner)::value>(f);
        }
        if (decltype(full)::value || k < rows)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst, xoff, (uint32_t)k * row_bytes_out,
                                                DCP_STORE_AUX);
      }
    };
    if (rows == kLdsTH && interior) tile_rows_loop(std::true_type{}, std::true_type{});
    else if (rows == kLdsTH) tile_rows_loop(std::true_type{}, std::false_type{});
    else tile_rows_loop(std::false_type{}, std::false_type{});
    DCP_TRACE(6);
    DCP_TRACE_WAVE_END();
  } else {
    // ---- box too large for the slab, or a tap outside the predicted box: direct global gather
#pragma unroll
    for (int k = 0; k < kLdsTH; ++k) {
      const FetchT f = fetch<SAMPLER, true, float>(src, __builtin_amdgcn_fmed3f(xf[k], 0.0f, wmaxf),
                                                   __builtin_amdgcn_fmed3f(yf[k], 0.0f, hmaxf));
      const float v = finish<SAMPLER, true, float>(f);
      if (k < rows)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst, xoff, (uint32_t)k * row_bytes_out,
                                              DCP_STORE_AUX);
    }
  }
}

// ------------------------------------------------------------------ integer element types: the exact lerp

// Bilinear blend of 8- / 16-bit integer taps when BOTH coordinates are >= 32: a float32 coordinate in [32, 2^24) is a multiple of
// 2^-18, so the fractions carry at most 18 bits, a tap (or a difference of taps) at most 17, and every product and sum of
// scipy's ((v * wy) * wx) accumulation -- and of the factorised form below -- is EXACTLY representable in a double (<= 53
// bits: 16 + 18 + 18 + 1).  No operation rounds, so the two forms agree bit for bit (and with the true bilinear value), and the
// factorised one costs 13 float64 operations fewer per pixel.  `top` / `bot`: the tap pairs (x0, x0 + 1) of rows y0 / y0 + 1 in
// the low bits of a dword (element 0 in the low half); fx, fy: the fractions.  Tiles whose source box reaches a coordinate
// below 32 keep scipy's operation order (where roundings do occur it is their order that has to be reproduced).
template <typename T>
__device__ __forceinline__ double exact_lerp_pairs(uint32_t top, uint32_t bot, double fx, double fy) {
  constexpr int B = (int)sizeof(T) * 8;
  int a, b, c, d;
  if constexpr (std::is_signed<T>::value) {
    a = __builtin_amdgcn_sbfe(top, 0, B);
    b = __builtin_amdgcn_sbfe(top, B, B);
    c = __builtin_amdgcn_sbfe(bot, 0, B);
    d = __builtin_amdgcn_sbfe(bot, B, B);
  } else {
    a = (int)(top & ((1u << B) - 1u));
    c = (int)(bot & ((1u << B) - 1u));
    if constexpr (B == 16) {
      b = (int)(top >> 16);
      d = (int)(bot >> 16);
    } else {
      b = (int)__builtin_amdgcn_ubfe(top, B, B);
      d = (int)__builtin_amdgcn_ubfe(bot, B, B);
    }
  }
  const double tp = __builtin_fma(fx, (double)(b - a), (double)a);
  const double bt = __builtin_fma(fx, (double)(d - c), (double)c);
  return __builtin_fma(fy, bt - tp, tp);
}

// (exact_lerp_taps -- the same blend on narrow LDS reads -- and to_elem_in_range live in dcp_device.h: the colour kernel shares them)

// ------------------------------------------------------------------ K1 / K2, workgroup-shared source box

// The per-CU rate at which streamed source data can be brought in (vector L1 misses served by the L2: ~10 B/clk per
// CU, ~5 TB/s over the chip -- MI355X_MICROARCH.md, "global_load_dwordx4 (HBM-bound)") is what bounds the staged
// kernels, not HBM: with one box per WAVE tile (64 x 16 outputs, 80 x 21..24 floats fetched, every row segment cut
// into 128-byte lines at both ends) the L2 -> L1 traffic is 8 B per output pixel for 4 B of source.  This kernel
// stages ONE box per WORKGROUP: the four waves own a 2 x 2 arrangement of 64 x 16 sub-tiles (a 128 x 32 output tile),
// the box is the hull of the workgroup tile's four corner pixels grown by one pixel (host certificate for that tile
// shape: MapArgs::tile_dev_ok == 2), and the four waves copy it into one shared slab with 16-byte LDS-DMA loads --
// longer row segments (fewer partial lines) and half the halo rows: ~5.9 B per output pixel.
//   corners   every wave evaluates the workgroup tile's four corner pixels itself (lanes 0..3): no exchange, no barrier;
//   fill      the slab is a linear array of 16-byte chunks, 36 per row of pitch 144 floats; chunk (4 j + wave) 64 + lane
//             belongs to lane `lane` of wave `wave` in its j-th load (at most 6 per wave);
//   phase 1   the source coordinates of the sub-tile's 16 rows while the loads are in flight (row table per wave);
//   barrier   every wave waits for its own loads, then for the other waves' (the only barrier of the kernel);
//   phase 2   taps from the shared slab, blend, store -- as in remap_lds_kernel.
// A box that does not fit the slab (magnification > ~1.1) sends the whole workgroup to the direct global gather.
constexpr int kWgTW = 128, kWgTH = 32;            // outputs per workgroup: 2 x 2 wave tiles of kLdsTW x kLdsTH
constexpr int kWgBoxW = 144, kWgBoxH = 42;        // largest box: 36 chunks of 16 B per row, 42 rows (api_core.cpp: kWgBoxRows / kWgBoxCols)
constexpr int kWgSlabRows = 42;                   // 24 192 B: six workgroups per CU
static_assert(kLdsTW == 64 && kLdsTH == 16, "remap_wg_kernel assumes 64 x 16 wave tiles");

#ifndef DCP_WG_UNTRACKED_DMA
#define DCP_WG_UNTRACKED_DMA 1   // 0: remap_wg_kernel's fill through the compiler's LDS-DMA builtin (rounds 2-3; A/B)
#endif
#ifndef DCP_WG_FILL_START
#define DCP_WG_FILL_START 0   // phase 1 row in front of which the first load of the fill is issued
#endif
#ifndef DCP_WG_FILL_EVERY
#define DCP_WG_FILL_EVERY 1   // phase 1 rows between two loads of the fill (0: all six loads in one burst in front of phase 1)
#endif
#ifndef DCP_WG_WAVES
#define DCP_WG_WAVES 6      // waves per SIMD the register allocation aims at (six 24 KB workgroups fit a CU's LDS)
#endif
// Corner `corner` (bit 0: right, bit 1: bottom) of workgroup tile (tx, yblk / kWgTH): its source pixel, rounded and clipped as the
// taps are.  Pixels past the image are clamped to the last valid ones, so the corners span exactly the valid part of the tile.
template <int KIND, int NF>
__device__ __forceinline__ void wg_corner_tap(const ImageArgs& img, const MapArgs& map, int tx, int yblk, int corner, int* cxi, int* cyi) {
  const float wmaxf = (float)(img.W - 1), hmaxf = (float)(img.H - 1);
  const double X = (double)min(tx * kWgTW + (corner & 1) * (kWgTW - 1), img.W - 1);
  const double Y = (double)(img.y_origin + min(yblk + ((corner >> 1) & 1) * (kWgTH - 1), img.rows_out - 1));
  double xd, yd;
  if constexpr (KIND == kFused) {
    // Lanes 0..3 together (corner = lane).  The homography maps the tile onto a convex quadrilateral (denominator of one sign: the
    // host's certificate), so the float32-rounded, clipped perspective positions of ALL its pixels -- rounding and clipping are
    // monotone -- lie in the bounding box Q of the four corners' positions.  The radial map is then taken at the corners of Q: its
    // deviation from the bilinear interpolant of those four values over Q is what the certificate bounds, so their hull grown by
    // one pixel holds every tap of the tile -- also of a tile that the inner clip cuts through, where the composed map has a kink.
    double px, py;
    corner_coord<kPersp, NF>(map, X, Y, &px, &py);
    const float pxf = round_clip_f32(px, wmaxf), pyf = round_clip_f32(py, hmaxf);
    float qx[4], qy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qx[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxf), i));
      qy[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pyf), i));
    }
    const float qx0 = fminf(fminf(qx[0], qx[1]), fminf(qx[2], qx[3])), qx1 = fmaxf(fmaxf(qx[0], qx[1]), fmaxf(qx[2], qx[3]));
    const float qy0 = fminf(fminf(qy[0], qy[1]), fminf(qy[2], qy[3])), qy1 = fmaxf(fmaxf(qy[0], qy[1]), fmaxf(qy[2], qy[3]));
    corner_coord<kRadial, NF>(map, (double)((corner & 1) ? qx1 : qx0), (double)((corner & 2) ? qy1 : qy0), &xd, &yd);
  } else {
    corner_coord<KIND, NF>(map, X, Y, &xd, &yd);
  }
  *cxi = (int)round_clip_f32(xd, wmaxf);
  *cyi = (int)round_clip_f32(yd, hmaxf);
}

// T: element type of source and result.  float is the tuned float32 path (any blend).  uint8 / int8 / uint16 / int16
// (what detectors and cameras deliver) run the same kernel on narrower slab rows -- 16-bit: 20 chunks of 16 bytes =
// 160 elements, 8-bit: 10 chunks = 160 elements, the box's first column rounded down to a 4-byte boundary -- read their
// taps with ds_read_u16 / _i16 / _u8 / _i8, blend in scipy's exact float64 operation order (SAMPLER = kScipy; the
// integer result depends on it at rounding ties) and convert as scipy does: round half away from zero, saturate.
// ImageArgs::src / dst / src_stride / src_bytes keep their meaning (pointers reinterpreted, stride in ELEMENTS, extent in bytes).
// (the body is a device function: the single-frame kernel and the multi-frame kernel remap_wg_batch_kernel share it)
template <int KIND, int NF, int SAMPLER, typename T>
__device__ __forceinline__ void remap_wg_body(const ImageArgs& img, const MapArgs& map) {
  constexpr bool kIsF32 = std::is_same<T, float>::value;
  constexpr int ES = (int)sizeof(T);                           // element size in bytes
  constexpr int CH = ES == 4 ? 36 : (ES == 2 ? 20 : 10);       // 16-byte chunks per slab row
  constexpr int PB = CH * 16;                                  // slab pitch in bytes
  constexpr int kBoxWEl = PB / ES;                             // widest box in elements: 144 / 160 / 160
  constexpr int NJ = (kWgBoxH * CH + 255) / 256;               // loads per wave that cover the slab: 6 / 4 / 2
  static_assert(kIsF32 || SAMPLER == kNearest || SAMPLER == kScipy, "integer element types blend in scipy's exact order");
  // (the slab holds every chunk the NJ loads of the four waves can write -- 43 rows of float32, 52 of the narrower types --
  // so that no lane of a load has to be masked: rows past the box are past the fill descriptor's extent and arrive as zeros)
  constexpr int kSlabChunks = NJ * 256;
  __shared__ __attribute__((aligned(16))) unsigned char s_box[kSlabChunks * 16];
  static_assert(kSlabChunks * 16 >= kWgSlabRows * PB, "the slab covers the largest box");
  __shared__ double s_row[4][kLdsTH][KIND == kRadial ? 2 : 4];     // one row table per wave: no barrier before it is read
  __shared__ double s_coef[NF < 0 ? kMaxFact : 1];
  using FetchT = Fetch<SAMPLER, true, float>;
  constexpr int RW = KIND == kRadial ? 2 : 4;
  const T* const srcT = (const T*)img.src;
  T* const dstT = (T*)img.dst;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int wx = wave & 1, wy = wave >> 1;
  DCP_TRACE_WAVE_BEGIN((((unsigned)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4u + (unsigned)wave);
  // tile order: XCD blockIdx.x & 7 owns a vertical stripe of tile columns and sweeps it row by row (see remap_lds_kernel)
  int tx, ty;
  if (img.xcd_remap == 2) {
    const int s = blockIdx.x & 7, c = blockIdx.x >> 3;
    const int wq = img.tiles_x >> 3, wr = img.tiles_x & 7;
    if (c >= wq + (s < wr ? 1 : 0)) return;
    tx = s * wq + min(s, wr) + c;
    ty = blockIdx.y;
  } else {                       // plain row-major order (grid = tiles_x x tiles_y): x-neighbours on different XCDs, every XCD the same load
    tx = blockIdx.x;
    ty = blockIdx.y;
  }
  const int yblk = ty * kWgTH;
  // (wave-uniform by construction; said explicitly, or the store descriptor derived from it lands in VGPRs and every
  // store is wrapped in a waterfall loop)
  const int y0 = __builtin_amdgcn_readfirstlane(yblk + wy * kLdsTH);   // first row of this wave's sub-tile (inside the band of output rows)
  const int x = tx * kWgTW + wx * kLdsTW + lane;
  const float wmaxf = (float)(img.W - 1), hmaxf = (float)(img.H - 1);

  // ---- the workgroup tile's four corner pixels, one per lane 0..3, by every wave for itself (no exchange, no barrier:
  // the fill below can start a few hundred cycles into the wave's life).  Pixels past the image are clamped to the
  // last valid ones, so the corners span exactly the valid part of the tile.  The values may differ from the
  // row-hoisted evaluation of phase 1 in the last bits; the certificate leaves 0.05 px for that.
  int cx0, cx1, cy0, cy1;
  if (img.boxes != nullptr) {
    // (the hull was computed once per tile by box_table_kernel with the same code: four scalars from one scalar load,
    // instead of ~70 vector instructions in each of the four waves)
    typedef __attribute__((address_space(4))) const int32_t* const_ptr;
    const const_ptr b = (const_ptr)(img.boxes + 4 * ((size_t)blockIdx.z * img.tiles_y * img.tiles_x + (size_t)ty * img.tiles_x + tx));
    cx0 = b[0];
    cx1 = b[1];
    cy0 = b[2];
    cy1 = b[3];
  } else {
    int cxi, cyi;
    wg_corner_tap<KIND, NF>(img, map, tx, yblk, lane, &cxi, &cyi);
    const int xa = __builtin_amdgcn_readlane(cxi, 0), xb = __builtin_amdgcn_readlane(cxi, 1);
    const int xc_ = __builtin_amdgcn_readlane(cxi, 2), xd_ = __builtin_amdgcn_readlane(cxi, 3);
    const int ya = __builtin_amdgcn_readlane(cyi, 0), yb = __builtin_amdgcn_readlane(cyi, 1);
    const int yc_ = __builtin_amdgcn_readlane(cyi, 2), yd_ = __builtin_amdgcn_readlane(cyi, 3);
    cx0 = min(min(xa, xb), min(xc_, xd_));
    cx1 = max(max(xa, xb), max(xc_, xd_));
    cy0 = min(min(ya, yb), min(yc_, yd_));
    cy1 = max(max(ya, yb), max(yc_, yd_));
  }
  // hull of the corner taps grown by one pixel (the certified deviation is below one pixel)
  // (narrow element types: the first column rounded down to a 4-byte boundary of the source row, as the 16-byte copies need)
  const int bx0 = max(min(cx0 - 1, img.W - 2), 0) & ~(4 / ES - 1);
  const int bx1 = min(cx1 + 2, img.W - 1);
  const int by0 = max(min(cy0 - 1, img.H - 2), 0);
  const int by1 = min(cy1 + 2, img.H - 1);
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  const bool fits = bw <= kBoxWEl && bh <= kWgBoxH;          // workgroup-uniform
  DCP_TRACE(1);
  DCP_TRACE(2);

  // ---- fill: this wave's share of the box.  The slab is a linear array of 16-byte chunks, 36 per row of pitch 144
  // floats; chunk (4 j + wave) 64 + lane belongs to lane `lane` of wave `wave` in its j-th load (at most 6 per wave).
  // Advancing j adds 256 chunks = 7 rows + 4 chunks: row and byte offset of load j follow from those of load 0 with
  // one wrap test.  (A/B-tested against dealing whole row triples to the waves through per-load descriptors -- no
  // vector arithmetic per load at all, but 5 us slower per frame.)  The loads are not issued in one burst: load j
  // goes out in front of coordinate row DCP_WG_FILL_EVERY * j of phase 1 (a wave stuck at a full memory queue does no arithmetic).
  typedef __attribute__((address_space(3))) void* lds_ptr;
  static_assert(kWgBoxW == 144 && kWgBoxH * 36 <= 6 * 256, "six loads of 64 chunks per wave cover the float32 slab");
  const uint32_t rstep = (uint32_t)img.src_stride * (uint32_t)ES;      // source row pitch in bytes
  // the fill's descriptor ends with the box's last row: a chunk of a later row is out of range -- zeros, and no memory access --
  // which replaces the per-lane row test of every load (7 -> 3 vector instructions per load)
  const unsigned long long rows_end = (unsigned long long)(by1 + 1) * rstep;
  const uint32_t fill_extent = rows_end < (unsigned long long)img.src_bytes ? (uint32_t)rows_end : img.src_bytes;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)img.src, 0, (int)fill_extent, 0x00020000);
  [[maybe_unused]] const dcp_rsrc_words src_words = raw_rsrc_words(img.src, fill_extent);
  [[maybe_unused]] const uint32_t box0 = (uint32_t)(uintptr_t)(lds_ptr)&s_box[0];
  const int fc = wave * 64 + lane;
  const int crow0 = fc / CH;                                            // (constant divisor: a multiply and a shift)
  const int c160 = fc - crow0 * CH;
  const uint32_t off0 = ((uint32_t)by0 * (uint32_t)img.src_stride + (uint32_t)bx0) * (uint32_t)ES + (uint32_t)crow0 * rstep + (uint32_t)c160 * 16u;
  const int nchunk = bh * CH;
  auto issue_fill = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < NJ) {
      if (fits && (j * 4 + wave) * 64 < nchunk) {             // wave-uniform: does any chunk of this load lie inside the box?
        // 256 j chunks further on: (256 j) / CH whole rows, and the column wraps into the next row at most once
        constexpr int qrow = (256 * j) / CH, rem = (256 * j) % CH;
        const bool wrap = c160 >= CH - rem;
        // (the two alternatives are scalars: one compare, one select, one add per load)
        const uint32_t step_nowrap = (uint32_t)qrow * rstep + (uint32_t)rem * 16u, step_wrap = step_nowrap + rstep - (uint32_t)PB;
#if DCP_WG_UNTRACKED_DMA
        // (hidden from the compiler, which otherwise puts `s_waitcnt vmcnt(0)` in front of a row-table read in the middle of phase 1
        // on the interior-tile path: see lds_dma16_untracked.  The wait for the fill is the explicit one in front of the barrier below)
        lds_dma16_untracked(src_words, box0 + (uint32_t)((j * 4 + wave) * 1024), off0 + (wrap ? step_wrap : step_nowrap));
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lds_ptr)(s_box + (j * 4 + wave) * 1024), 16, off0 + (wrap ? step_wrap : step_nowrap), 0, 0,
                                                 DCP_FILL_AUX);
#endif
      }
    }
  };

  DCP_TRACE(3);
  // ---- row table of this wave's 16 rows (lanes 0..15; same-wave LDS traffic is ordered, no barrier)
  if (lane < kLdsTH)
    fill_row<KIND, RW>(map, s_row[wave], lane, (double)(img.y_origin + min(y0 + lane, img.rows_out - 1)));
  if constexpr (NF < 0 && KIND != kPersp) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
    __syncthreads();
  }
  // (max(0, min(..)) selects v_med3_i32, a VALU-only instruction: without the readfirstlane the value -- and the store
  // descriptor built from it -- would live in VGPRs)
  const int rows = __builtin_amdgcn_readfirstlane(max(0, min(kLdsTH, img.rows_out - y0)));
  const int ybase = __builtin_amdgcn_readfirstlane(min(y0, img.rows_out - 1));
  const ColCtx col = make_col<KIND, NF>(map, min(x, img.W - 1));
  const auto* rowtab = s_row[wave];
  const uint32_t row_bytes_out = (uint32_t)img.W * (uint32_t)ES;
  const char* out_base = (const char*)(dstT + (size_t)ybase * (size_t)img.W);
  const uint32_t xoff = (uint32_t)x * (uint32_t)ES;         // lanes with x >= W: the store is out of range and dropped
  const __amdgpu_buffer_rsrc_t dst =
      __builtin_amdgcn_make_buffer_rsrc((void*)out_base, 0, (int)((uint32_t)rows * row_bytes_out), 0x00020000);

  // ---- phase 1: the source coordinates of the sub-tile's 16 rows, while the loads are in flight.  A box strictly
  // inside the image needs no clip (every coordinate lies inside the box).
  float xf[kLdsTH], yf[kLdsTH];
  const bool box_inside = cx0 - 1 >= 0 && cx1 + 2 <= img.W - 1 && cy0 - 1 >= 0 && cy1 + 2 <= img.H - 1;
  const bool unclipped = box_inside && fits;
  auto rows_1 = [&](auto noclip, auto fastdiv) {
    if constexpr (DCP_WG_FILL_EVERY == 0) {
      issue_fill(std::integral_constant<int, 0>{});
      issue_fill(std::integral_constant<int, 1>{});
      issue_fill(std::integral_constant<int, 2>{});
      issue_fill(std::integral_constant<int, 3>{});
      issue_fill(std::integral_constant<int, 4>{});
      issue_fill(std::integral_constant<int, 5>{});
    }
#pragma unroll
    for (int k = 0; k < kLdsTH; ++k) {
      if constexpr (DCP_WG_FILL_EVERY > 0) {
        if (k == DCP_WG_FILL_START + 0 * DCP_WG_FILL_EVERY) issue_fill(std::integral_constant<int, 0>{});
        if (k == DCP_WG_FILL_START + 1 * DCP_WG_FILL_EVERY) issue_fill(std::integral_constant<int, 1>{});
        if (k == DCP_WG_FILL_START + 2 * DCP_WG_FILL_EVERY) issue_fill(std::integral_constant<int, 2>{});
        if (k == DCP_WG_FILL_START + 3 * DCP_WG_FILL_EVERY) issue_fill(std::integral_constant<int, 3>{});
        if (k == DCP_WG_FILL_START + 4 * DCP_WG_FILL_EVERY) issue_fill(std::integral_constant<int, 4>{});
        if (k == DCP_WG_FILL_START + 5 * DCP_WG_FILL_EVERY) issue_fill(std::integral_constant<int, 5>{});
      }
      double xd, yd;
      map_coord<KIND, NF, RW, decltype(fastdiv)::value>(map, rowtab, s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
      if constexpr (decltype(noclip)::value) {
        xf[k] = (float)xd;
        yf[k] = (float)yd;
      } else {
        xf[k] = round_clip_f32(xd, wmaxf);
        yf[k] = round_clip_f32(yd, hmaxf);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (rows > 0) {
    if (KIND == kRadial || !map.fast_div) {
      if (unclipped) rows_1(std::true_type{}, I0{});
      else rows_1(std::false_type{}, I0{});
    } else {
      if (unclipped) rows_1(std::true_type{}, I1{});
      else rows_1(std::false_type{}, I1{});
    }
  } else {                                         // nothing to compute for this wave: only its share of the copy
    issue_fill(std::integral_constant<int, 0>{});
    issue_fill(std::integral_constant<int, 1>{});
    issue_fill(std::integral_constant<int, 2>{});
    issue_fill(std::integral_constant<int, 3>{});
    issue_fill(std::integral_constant<int, 4>{});
    issue_fill(std::integral_constant<int, 5>{});
  }
  if (!fits && lane == 0 && wave == 0) atomicAdd(&g_lds_stats[0], 1ull);
  DCP_TRACE(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                 // every wave's share of the box has landed
  DCP_TRACE(5);

  if (rows == 0 || !(x < img.W)) return;           // no cross-lane work from here on
  if (fits) {
    // ---- phase 2: taps from the shared slab, blend, store
    const uint32_t negorg4 = (uint32_t)(-(by0 * PB + bx0 * ES));
    const char* boxb = (const char*)s_box;
    const bool interior = bx1 < img.W - 1 && by1 < img.H - 1;
    const float negorg4f = (float)(int32_t)negorg4;
    auto tap_addr = [&](int xi, int yi) -> uint32_t {            // yi * pitch + xi * element size + origin, 24-bit multiply
      uint32_t xa, a_;
      if constexpr (ES == 4) asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(xa) : "v"(xi), "s"(negorg4));
      else if constexpr (ES == 2) asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(xa) : "v"(xi), "s"(negorg4));
      else xa = (uint32_t)xi + negorg4;
      asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a_) : "v"(yi), "s"(PB), "v"(xa));
      return a_;
    };
    auto tile_rows_loop = [&](auto full, auto inner, auto exact) {
#pragma unroll
      for (int k = 0; k < kLdsTH; ++k) {
        FetchT f;
        uint32_t addr;
        if constexpr (SAMPLER == kNearest) {
          int xi = (int)xf[k];
          int yi = (int)yf[k];
          xi += (xf[k] - (float)xi >= 0.5f) ? 1 : 0;
          yi += (yf[k] - (float)yi >= 0.5f) ? 1 : 0;
          addr = tap_addr(xi, yi);
        } else if constexpr (decltype(inner)::value) {
          f.fx = __builtin_amdgcn_fractf(xf[k]);      // x - floor(x), exact
          f.fy = __builtin_amdgcn_fractf(yf[k]);
          const float flx = xf[k] - f.fx, fly = yf[k] - f.fy;
          const float af = __builtin_fmaf(fly, (float)PB, __builtin_fmaf(flx, (float)ES, negorg4f));
          addr = (uint32_t)(int32_t)af;
        } else {
          const int xi = min((int)xf[k], img.W - 2);
          const int yi = min((int)yf[k], img.H - 2);
          f.fx = xf[k] - (float)xi;
          f.fy = yf[k] - (float)yi;
          addr = tap_addr(xi, yi);
        }
        DCP_BOUNDS(addr, SAMPLER == kNearest ? ES : PB + 2 * ES, kSlabChunks * 16, 2);
        const T* t = (const T*)(boxb + addr);
        if constexpr (kIsF32) {
          float v;
          if constexpr (SAMPLER == kNearest) {
            v = t[0];
          } else {
            f.a.x = __float_as_uint(t[0]);
            f.a.y = __float_as_uint(t[1]);
            f.b.x = __float_as_uint(t[kBoxWEl]);
            f.b.y = __float_as_uint(t[kBoxWEl + 1]);
            v = finish<SAMPLER, true, float, !decltype(inner)::value>(f);
          }
          if (decltype(full)::value || k < rows)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst, xoff, (uint32_t)k * row_bytes_out, DCP_STORE_AUX);
        } else {
          T v;
          if constexpr (SAMPLER == kNearest) {
            v = t[0];
          } else {
            const double fx = (double)f.fx, fy = (double)f.fy;
            double acc;
            if constexpr (decltype(exact)::value) {
              // every coordinate of the tile >= 32: no operation rounds (exact_lerp_pairs), no clamps.  The pairs come as aligned
              // dwords here: four narrow LDS reads (exact_lerp_taps) are 7 % faster in the stack kernel, whose addresses are
              // loop-invariant, and 1.5 % slower in this one (tools/ab_int_exact.py, same box)
              const uint32_t* q = (const uint32_t*)(boxb + (addr & ~3u));
              const uint32_t sh = (addr & 3u) * 8u;
              acc = exact_lerp_pairs<T>(__builtin_amdgcn_alignbit(q[1], q[0], sh), __builtin_amdgcn_alignbit(q[PB / 4 + 1], q[PB / 4], sh), fx, fy);
              v = to_elem_in_range<T>(acc);
            } else {
            // The tap pair (x0, x0 + 1) of a row starts at any multiple of the element size: read the two ALIGNED dwords
            // around it (one ds_read2_b32) and shift the pair down -- a dword read at a 2- or 1-byte boundary works on this
            // hardware but costs 40 us per frame.
            const uint32_t* q = (const uint32_t*)(boxb + (addr & ~3u));
            const uint32_t sh = (addr & 3u) * 8u;
            const uint32_t top = __builtin_amdgcn_alignbit(q[1], q[0], sh), bot = __builtin_amdgcn_alignbit(q[PB / 4 + 1], q[PB / 4], sh);
              // scipy NI_GeometricTransform: w0 = 1 - f, w1 = 1 - w0; ((v*wy)*wx) summed left to right, then its integer store
              const double wy0 = 1.0 - fy, wy1 = 1.0 - wy0;
              const double wx0 = 1.0 - fx, wx1 = 1.0 - wx0;
              auto tap = [](uint32_t w, int i) -> double {            // element i (0 / 1) of the pair, as scipy reads it: a double
                if constexpr (std::is_signed<T>::value) return (double)(int32_t)__builtin_amdgcn_sbfe(w, i * ES * 8, ES * 8);
                else return (double)__builtin_amdgcn_ubfe(w, i * ES * 8, ES * 8);
              };
              acc = (tap(top, 0) * wy0) * wx0;
              acc += (tap(top, 1) * wy0) * wx1;
              acc += (tap(bot, 0) * wy1) * wx0;
              acc += (tap(bot, 1) * wy1) * wx1;
              v = to_elem<T>(acc);
            }
          }
          if (decltype(full)::value || k < rows) {
            if constexpr (ES == 2)
              __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, dst, xoff, (uint32_t)k * row_bytes_out, DCP_STORE_AUX);
            else
              __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, dst, xoff, (uint32_t)k * row_bytes_out, DCP_STORE_AUX);
          }
        }
      }
    };
    if constexpr (kIsF32 || SAMPLER == kNearest) {
      if (rows == kLdsTH && interior) tile_rows_loop(std::true_type{}, std::true_type{}, std::false_type{});
      else if (rows == kLdsTH) tile_rows_loop(std::true_type{}, std::false_type{}, std::false_type{});
      else tile_rows_loop(std::false_type{}, std::false_type{}, std::false_type{});
    } else {
      // integer elements, order 1: whole interior tiles whose box lies at coordinates >= 32 take the exact factorised blend
      // (every coordinate of the tile is >= its box origin); the others scipy's operation order
      if (rows == kLdsTH && interior && img.int_exact && bx0 >= (int)kExactLerpMinCoord && by0 >= (int)kExactLerpMinCoord)
        tile_rows_loop(std::true_type{}, std::true_type{}, std::true_type{});
      else tile_rows_loop(std::false_type{}, std::false_type{}, std::false_type{});
    }
    DCP_TRACE(6);
    DCP_TRACE_WAVE_END();
  } else {
    // ---- box too large for the slab: direct global gather
#pragma unroll
    for (int k = 0; k < kLdsTH; ++k) {
      const float xc = __builtin_amdgcn_fmed3f(xf[k], 0.0f, wmaxf), yc = __builtin_amdgcn_fmed3f(yf[k], 0.0f, hmaxf);
      if constexpr (kIsF32) {
        const SrcView src = make_view(img.src, img.src_bytes, img.W, img.H, img.src_stride, 1);
        const FetchT f = fetch<SAMPLER, true, float>(src, xc, yc);
        const float v = finish<SAMPLER, true, float>(f);
        if (k < rows)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst, xoff, (uint32_t)k * row_bytes_out, DCP_STORE_AUX);
      } else {
        T v;
        if constexpr (SAMPLER == kNearest) {
          int xi = (int)xc, yi = (int)yc;
          xi += (xc - (float)xi >= 0.5f) ? 1 : 0;
          yi += (yc - (float)yi >= 0.5f) ? 1 : 0;
          v = srcT[(size_t)yi * (size_t)img.src_stride + (size_t)xi];
        } else {
          const int xi = min((int)xc, img.W - 2), yi = min((int)yc, img.H - 2);
          const double fx = (double)(xc - (float)xi), fy = (double)(yc - (float)yi);
          const T* t = srcT + (size_t)yi * (size_t)img.src_stride + (size_t)xi;
          const double wy0 = 1.0 - fy, wy1 = 1.0 - wy0;
          const double wx0 = 1.0 - fx, wx1 = 1.0 - wx0;
          double acc = ((double)t[0] * wy0) * wx0;
          acc += ((double)t[1] * wy0) * wx1;
          acc += ((double)t[img.src_stride] * wy1) * wx0;
          acc += ((double)t[img.src_stride + 1] * wy1) * wx1;
          v = to_elem<T>(acc);
        }
        if (k < rows) {
          if constexpr (ES == 2)
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, dst, xoff, (uint32_t)k * row_bytes_out, DCP_STORE_AUX);
          else
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, dst, xoff, (uint32_t)k * row_bytes_out, DCP_STORE_AUX);
        }
      }
    }
  }
}

// (runtime-length coefficient vectors -- NF < 0 -- walk the vector in LDS and need more registers than six waves per SIMD leave: at six
// they spilled 116-148 bytes into scratch, and every reload of a spill is a VMEM wait that also waits for the untracked fill;
// tools/isa_hazards.py keeps the shipped instantiations free of scratch)
template <int KIND, int NF, int SAMPLER, typename T = float>
__global__ void __launch_bounds__(256, (NF < 0 ? 4 : (NF == 1 ? 5 : DCP_WG_WAVES))) remap_wg_kernel(const ImageArgs img, const MapArgs map) {
  remap_wg_body<KIND, NF, SAMPLER, T>(img, map);
}

// ------------------------------------------------------------------ K1, many frames in one launch

// remap_wg_kernel over a batch of frames of one shape, each with its OWN source, destination, centre and coefficient
// vector: blockIdx.z = frame.  A per-frame launch spends ~3.5 us before its first tile completes and ~9 us draining
// its last workgroups (profiles/rounds_1-4/r02b_phase_timeline_wg_f64lerp.txt: a third of a 4096^2 launch); here the tail of
// frame z runs under the head of frame z + 1.  The per-frame arguments travel in the KERNEL ARGUMENTS (a table of
// BatchEntry<NF>, indexed by the workgroup's frame: scalar loads from the kernarg segment, no device-side table whose
// lifetime would have to outlast the launch), which bounds a launch to BatchTable<NF>::kMax frames; longer batches are
// cut into several launches by the host.  Radial map, float32, certified at level 2 for EVERY frame (the host checks).
template <int NF>
struct BatchEntry {
  const float* src;
  float* dst;
  double xc, yc;
  double fact[NF];
};
template <int NF>
struct BatchTable {
  // kernel arguments are limited to 4 KB: ImageArgs (72 B) + the table
  static constexpr int kMax = (4096 - 128) / (int)sizeof(BatchEntry<NF>) > 64 ? 64 : (4096 - 128) / (int)sizeof(BatchEntry<NF>);
  BatchEntry<NF> e[kMax];
};

template <int NF>
__device__ __forceinline__ void batch_frame_args(const ImageArgs& img0, const BatchTable<NF>& tab, int frame, ImageArgs* img, MapArgs* map) {
  const BatchEntry<NF>& e = tab.e[frame];
  *img = img0;
  img->src = e.src;
  img->dst = e.dst;
  map->xc = e.xc;
  map->yc = e.yc;
#pragma unroll
  for (int i = 0; i < NF; ++i) map->fact[i] = e.fact[i];
  map->nfact = NF;
  map->fast_div = 0;
  map->tile_dev_ok = 2;
}

template <int NF, int SAMPLER>
__global__ void __launch_bounds__(256, DCP_WG_WAVES) remap_wg_batch_kernel(const ImageArgs img0, const BatchTable<NF> tab) {
  ImageArgs img;
  MapArgs map;
  batch_frame_args<NF>(img0, tab, blockIdx.z, &img, &map);
  remap_wg_body<kRadial, NF, SAMPLER, float>(img, map);
}

// The corner hulls of every workgroup tile of every frame of a batch, once per tile (one thread each, the four corners in
// turn) with the code the waves of remap_wg_kernel would otherwise run four times per tile: (x0, x1, y0, y1) of the rounded,
// clipped corner taps.  ~2 us for 24 frames of 4096 tiles; the table lives in stream-ordered scratch of the batch launch.
template <int KIND, int NF>
__device__ __forceinline__ void tile_hull(const ImageArgs& img, const MapArgs& map, int frame, int32_t* boxes) {
  const int ntiles = img.tiles_x * img.tiles_y;
  const int t = (int)blockIdx.x * 64 + (int)threadIdx.x;
  if (t >= ntiles) return;
  const int ty = t / img.tiles_x, tx = t - ty * img.tiles_x;
  int x0 = INT32_MAX, x1 = INT32_MIN, y0 = INT32_MAX, y1 = INT32_MIN;
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    int cxi, cyi;
    wg_corner_tap<KIND, NF>(img, map, tx, ty * kWgTH, c, &cxi, &cyi);
    x0 = min(x0, cxi);
    x1 = max(x1, cxi);
    y0 = min(y0, cyi);
    y1 = max(y1, cyi);
  }
  int4* const o = (int4*)(boxes + 4 * ((size_t)frame * ntiles + t));
  *o = make_int4(x0, x1, y0, y1);
}

template <int NF>
__global__ void __launch_bounds__(64) box_table_kernel(const ImageArgs img0, const BatchTable<NF> tab, int32_t* boxes) {
  ImageArgs img;
  MapArgs map;
  batch_frame_args<NF>(img0, tab, blockIdx.z, &img, &map);
  tile_hull<kRadial, NF>(img, map, blockIdx.z, boxes);
}

// ------------------------------------------------------------------ K5: explicit coordinates

template <int SAMPLER, bool PAIR, typename CT>
__global__ void __launch_bounds__(kBlock) remap_coords_kernel(const ImageArgs img, const CoordArgs ca) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= ca.npts) return;
  const SrcView src = make_view(img.src, img.src_bytes, img.W, img.H, img.src_stride, img.src_col_stride);
  CT yc = ((const CT*)ca.ycoord)[i];
  CT xc = ((const CT*)ca.xcoord)[i];
  const CT wmax = (CT)(img.W - 1), hmax = (CT)(img.H - 1);
  if (ca.mode != kModeNearest && (yc < (CT)0 || yc > hmax || xc < (CT)0 || xc > wmax)) {
    // outside the image under a mode other than 'nearest': scipy's coordinate mapping and tap folding, in double and in
    // scipy's operation order whatever the blend (rare points: plain 64-bit addressing)
    const float* base = img.src;
    const int64_t rs = img.src_stride, cs = img.src_col_stride;
    img.dst[i] = (float)mc_sample_outside([&](long long r, long long c) -> double { return (double)base[r * rs + c * cs]; }, img.H, img.W,
                                          (double)yc, (double)xc, SAMPLER == kNearest ? 0 : 1, ca.mode);
    return;
  }
  xc = xc < (CT)0 ? (CT)0 : xc;
  xc = xc > wmax ? wmax : xc;
  yc = yc < (CT)0 ? (CT)0 : yc;
  yc = yc > hmax ? hmax : yc;
  const Fetch<SAMPLER, PAIR, CT> f = fetch<SAMPLER, PAIR, CT>(src, xc, yc);
  img.dst[i] = finish<SAMPLER, PAIR, CT>(f);
}

// ------------------------------------------------------------------ K6: coordinate maps

// The float32 (yd, xd) planes themselves -- _generate_perspective_map (postprocessing.py:444-459)
// and the xd_mat / yd_mat of unwarp_image_backward (:144-145) -- for callers that reuse a map
// (map_index=, cv2.remap).  Same coordinate code as the remap kernels, no sampling.
template <int KIND>
__global__ void __launch_bounds__(kBlock) coord_map_kernel(const ImageArgs img, const MapArgs map, float* ymap,
                                                           float* xmap) {
  __shared__ double s_row[kMaxTileRows][4];
  __shared__ double s_coef[kMaxFact];
  const int ty = blockIdx.x / img.tiles_x;
  const int tx = blockIdx.x - ty * img.tiles_x;
  const int y0 = ty * img.tile_rows;
  const int x = tx * kBlock + (int)threadIdx.x;
  const int rows = min(img.tile_rows, img.H - y0);
  if ((int)threadIdx.x < rows) fill_row<KIND, 4>(map, s_row, threadIdx.x, (double)(y0 + (int)threadIdx.x));
  if constexpr (KIND != kPersp) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
  }
  __syncthreads();
  if (x >= img.W) return;
  const float wmaxf = (float)(img.W - 1), hmaxf = (float)(img.H - 1);
  const ColCtx col = make_col<KIND, -1>(map, x);
  for (int k = 0; k < rows; ++k) {
    double xd, yd;
    map_coord<KIND, -1, 4>(map, s_row, s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
    const size_t o = (size_t)(y0 + k) * (size_t)img.W + (size_t)x;
    ymap[o] = round_clip_f32(yd, hmaxf);
    xmap[o] = round_clip_f32(xd, wmaxf);
  }
}

// ------------------------------------------------------------------ K4: rows of a (D,H,W) stack

// out[d, r, x] = projection d sampled at the radial source coordinate of (row_start + r, x).
// The coordinate and the tap weights are computed once per (r, x) and reused for d_chunk
// projections; the per-projection work is two 8-byte gathers, the blend and a 4-byte store.
// (body shared by stack_rows_kernel -- one centre, from the kernel arguments -- and stack_centres_kernel -- the centre of
// the workgroup's calibration from a table: xc / yc and the output block of that centre are parameters)
template <int NF, int SAMPLER, bool ROUND32>
__device__ __forceinline__ void stack_rows_body(const StackArgs& st, const MapArgs& map, const double xc_, const double yc_, const int r,
                                                float* __restrict__ out_block) {
  __shared__ double s_coef[kMaxFact];
  if constexpr (NF < 0) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
    __syncthreads();
  }
  const int x = blockIdx.x * kBlock + (int)threadIdx.x;
  const int d0 = blockIdx.z * st.d_chunk;
  const int d1 = min(st.D, d0 + st.d_chunk);
  if (x >= st.W) return;

  const double xu = (double)x - xc_;
  const double yu = (st.row_start + (double)r) - yc_;
  const double r2 = xu * xu + yu * yu;
  const double ru = sqrt_rn(r2);
  double f;
  if constexpr (NF >= 0) {
    double lead_e, lead_o;
    poly_leads<NF>(map.fact, &lead_e, &lead_o);
    f = poly_inline<NF>(map.fact, lead_e, lead_o, r2, ru);
  } else {
    f = poly_lds(s_coef, map.nfact, r2, ru);
  }
  const double xd = __builtin_fma(f, xu, xc_);
  const double yd = __builtin_fma(f, yu, yc_);

  using CT = typename std::conditional<ROUND32, float, double>::type;
  CT xc, yc;
  if constexpr (ROUND32) {
    xc = round_clip_f32(xd, (float)(st.W - 1));
    yc = round_clip_f32(yd, (float)(st.H - 1));
  } else {
    xc = clip_f64(xd, (double)(st.W - 1));
    yc = clip_f64(yd, (double)(st.H - 1));
  }
  if constexpr (ROUND32) {
    if (st.rbh > 0 && (yc < (float)st.rb0 || yc > (float)(st.rb0 + st.rbh - 1))) {
      // a folding model: the row coordinate lies outside the band the reference crops (rows rb0 .. rb0 + rbh - 1); scipy
      // reflects it inside the band (mode='reflect' on the cropped array, postprocessing.py:308-312) -- double arithmetic in
      // scipy's order whatever the blend, 64-bit addressing (rare pixels of a model nobody should use)
      const float yrel = yc - (float)st.rb0;             // float32, exact: the reference subtracts yd_min from the float32 plane
      const float* proj = st.vol + (size_t)d0 * (size_t)st.proj_stride + (size_t)st.rb0 * (size_t)st.row_stride;
      float* o = out_block + ((size_t)d0 * (size_t)st.nrows + (size_t)r) * (size_t)st.W + (size_t)x;
      for (int d = d0; d < d1; ++d) {
        const int64_t rs = st.row_stride;
        *o = (float)mc_sample_outside([&](long long rr, long long cc) -> double { return (double)proj[rr * rs + cc]; }, st.rbh, st.W,
                                      (double)yrel, (double)xc, 1, kModeReflect);
        proj += st.proj_stride;
        o += (size_t)st.nrows * (size_t)st.W;
      }
      return;
    }
  }
  // base tap and fractions once; per projection only the descriptor base moves
  int xi = min((int)xc, st.W - 2), yi = min((int)yc, st.H - 2);
  Fetch<SAMPLER, true, CT> ft;
  ft.fx = xc - (CT)xi;
  ft.fy = yc - (CT)yi;
  const uint32_t off = ((uint32_t)yi * (uint32_t)st.row_stride + (uint32_t)xi) << 2;   // full 32-bit product: a row stride may exceed 2^24
  const int row_bytes = st.row_stride * 4;
  const float* base = st.vol + (size_t)d0 * (size_t)st.proj_stride;
  float* out = out_block + ((size_t)d0 * (size_t)st.nrows + (size_t)r) * (size_t)st.W + (size_t)x;
  const size_t out_step = (size_t)st.nrows * (size_t)st.W;
#pragma unroll 4
  for (int d = d0; d < d1; ++d) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)st.proj_bytes, 0x00020000);
    ft.a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
    ft.b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, row_bytes, 0);
#if DCP_ROWS_NT_STORE
    __builtin_nontemporal_store(finish<SAMPLER, true, CT>(ft), out);     // (written once, never re-read here)
#else
    *out = finish<SAMPLER, true, CT>(ft);
#endif
    base += st.proj_stride;
    out += out_step;
  }
}

template <int NF, int SAMPLER, bool ROUND32>
__global__ void __launch_bounds__(kBlock) stack_rows_kernel(const StackArgs st, const MapArgs map) {
  stack_rows_body<NF, SAMPLER, ROUND32>(st, map, map.xc, map.yc, (int)blockIdx.y, st.out);
}

// K4 for a grid search over the centre of distortion (examples/example_05.py:62-65 calls unwarp_slice_backward 121 times on
// one stack, each time with another centre): the same rows of the same stack under `ncentres` calibrations that differ in
// (xcenter, ycenter) only, in ONE launch.  blockIdx.y = centre * nrows + row; the centres travel in the kernel arguments
// (scalar loads indexed by the workgroup's centre); out = (ncentres, depth, nrows, width).  The workgroups of one depth chunk
// run back to back over all centres (x fastest, then y), so the two or three source rows every centre needs of each of
// its projections are fetched from HBM once and then served by the L2.
struct CentreTable {
  static constexpr int kMax = 224;                  // 224 x 16 B = 3.5 KB of the 4 KB of kernel arguments
  double xc[kMax], yc[kMax];
};

template <int NF, int SAMPLER, bool ROUND32>
__global__ void __launch_bounds__(kBlock) stack_centres_kernel(const StackArgs st, const MapArgs map, const CentreTable tab, const int k0) {
  const int k = (int)blockIdx.y / st.nrows, r = (int)blockIdx.y - k * st.nrows;
  float* out_block = st.out + (size_t)(k0 + k) * (size_t)st.D * (size_t)st.nrows * (size_t)st.W;
  stack_rows_body<NF, SAMPLER, ROUND32>(st, map, tab.xc[k], tab.yc[k], r, out_block);
}

// K4 with the LDS-staged gather of K1: a wave owns a 64 x 16 tile of (x, row) positions; the coordinates, the
// source box and the per-pixel LDS tap address and fractions are computed ONCE and kept in registers, then for
// each of the d_chunk projections the box is filled by LDS-DMA from that projection and the 1024 pixels are
// blended out of LDS.  Per voxel that leaves two ds_read2, the blend and a store (~15 VALU instructions against
// ~42 for a single frame), and row-contiguous 16-byte fills instead of scattered 8-byte gathers.  A tile whose box
// does not fit the slab, or whose containment vote fails, gathers directly as stack_rows_kernel does.
// float32 coordinates only (unwarp_chunk_slices_backward); the slice path keeps stack_rows_kernel.
template <int NF, int SAMPLER>
__global__ void __launch_bounds__(64 * kLdsBW, (NF < 0 || SAMPLER == kScipy) ? 3 : DCP_LDS_WAVES) stack_lds_kernel(const StackArgs st, const MapArgs map) {
  __shared__ float s_box[kLdsBW][kBoxH * kBoxW];
  __shared__ double s_row[kLdsBW * kLdsTH][2];
  __shared__ double s_coef[NF < 0 ? kMaxFact : 1];
  using FetchT = Fetch<SAMPLER, true, float>;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int rblk = blockIdx.y * (kLdsBW * kLdsTH);
  const int r0 = rblk + wave * kLdsTH;              // first output row (index into the requested rows) of this wave
  const int x = blockIdx.x * kLdsTW + lane;
  const int d0 = blockIdx.z * st.d_chunk, d1 = min(st.D, d0 + st.d_chunk);
  if ((int)threadIdx.x < kLdsBW * kLdsTH)
    fill_row<kRadial, 2>(map, s_row, threadIdx.x, st.row_start + (double)min(rblk + (int)threadIdx.x, st.nrows - 1));
  if constexpr (NF < 0) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
  }
  __syncthreads();
  if (r0 >= st.nrows) return;                       // wave-uniform
  const int rows = min(kLdsTH, st.nrows - r0);
  const float wmaxf = (float)(st.W - 1), hmaxf = (float)(st.H - 1);
  const ColCtx col = make_col<kRadial, NF>(map, min(x, st.W - 1));
  const auto* rowtab = s_row + wave * kLdsTH;

  // ---- coordinates of the tile (once for all projections)
  float xf[kLdsTH], yf[kLdsTH];
#pragma unroll
  for (int k = 0; k < kLdsTH; ++k) {
    double xd, yd;
    map_coord<kRadial, NF, 2>(map, rowtab, s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
    xf[k] = round_clip_f32(xd, wmaxf);
    yf[k] = round_clip_f32(yd, hmaxf);
  }
  // ---- source box: hull of the four corner pixels grown by one pixel, verified for every pixel of the tile
  int cx0, cx1, cy0, cy1;
  {
    const int xa = (int)xf[0], xb = (int)xf[kLdsTH - 1], ya = (int)yf[0], yb = (int)yf[kLdsTH - 1];
    const int xa0 = __builtin_amdgcn_readlane(xa, 0), xa1 = __builtin_amdgcn_readlane(xa, 63);
    const int xb0 = __builtin_amdgcn_readlane(xb, 0), xb1 = __builtin_amdgcn_readlane(xb, 63);
    const int ya0 = __builtin_amdgcn_readlane(ya, 0), ya1 = __builtin_amdgcn_readlane(ya, 63);
    const int yb0 = __builtin_amdgcn_readlane(yb, 0), yb1 = __builtin_amdgcn_readlane(yb, 63);
    cx0 = min(min(xa0, xa1), min(xb0, xb1));
    cx1 = max(max(xa0, xa1), max(xb0, xb1));
    cy0 = min(min(ya0, ya1), min(yb0, yb1));
    cy1 = max(max(ya0, ya1), max(yb0, yb1));
  }
  const int bx0 = max(min(cx0 - DCP_LDS_MARGIN, st.W - 2), 0);
  const int bx1 = min(cx1 + 1 + DCP_LDS_MARGIN, st.W - 1);
  const int by0 = max(min(cy0 - DCP_LDS_MARGIN, st.H - 2), 0);
  const int by1 = min(cy1 + 1 + DCP_LDS_MARGIN, st.H - 1);
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  const bool fits = bw <= kBoxW && bh <= kBoxH;
  float xmn = xf[0], xmx = xf[0], ymn = yf[0], ymx = yf[0];
#pragma unroll
  for (int k = 1; k < kLdsTH; ++k) {
    xmn = __builtin_fminf(xmn, xf[k]);
    xmx = __builtin_fmaxf(xmx, xf[k]);
    ymn = __builtin_fminf(ymn, yf[k]);
    ymx = __builtin_fmaxf(ymx, yf[k]);
  }
  const bool inside = xmn >= (float)bx0 && (xmx < (float)bx1 || bx1 == st.W - 1) && ymn >= (float)by0 &&
                      (ymx < (float)by1 || by1 == st.H - 1);
  const bool staged = fits && __builtin_amdgcn_ballot_w64(!inside) == 0;
  if (!staged && lane == 0) atomicAdd(&g_lds_stats[fits ? 1 : 0], 1ull);

  // ---- per-pixel tap address in the slab and fractions (base tap held at x0 <= W-2, y0 <= H-2)
  uint32_t addr[kLdsTH];
  float fx[kLdsTH], fy[kLdsTH];
#pragma unroll
  for (int k = 0; k < kLdsTH; ++k) {
    const int xi = min((int)xf[k], st.W - 2), yi = min((int)yf[k], st.H - 2);
    fx[k] = xf[k] - (float)xi;
    fy[k] = yf[k] - (float)yi;
    // slab byte address when staged, byte offset inside the projection otherwise
    addr[k] = staged ? (uint32_t)(((yi - by0) * kBoxW + (xi - bx0)) * 4) : ((uint32_t)yi * (uint32_t)st.row_stride + (uint32_t)xi) << 2;
  }
  const bool active = x < st.W;
  float* box = s_box[wave];
  const char* boxb = (const char*)box;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  static_assert(kBoxW == 80, "the fill maps 20 lanes of 16 bytes to one slab row");
  const uint32_t org = ((uint32_t)by0 * (uint32_t)st.row_stride + (uint32_t)bx0) * 4u;
  const uint32_t rstep = (uint32_t)st.row_stride * 4u;
  const int lrow = (int)(__umul24((uint32_t)lane, 13u) >> 8);        // lane / 20 for lane < 64
  const int lcol = lane - lrow * 20;
  const uint32_t voff = (uint32_t)lrow * rstep + (uint32_t)lcol * 16u;      // full 32-bit product (rows of 16 MB and more)
  const int row_bytes = st.row_stride * 4;
  const float* base = st.vol + (size_t)d0 * (size_t)st.proj_stride;
  float* out = st.out + ((size_t)d0 * (size_t)st.nrows + (size_t)r0) * (size_t)st.W;
  const size_t out_step = (size_t)st.nrows * (size_t)st.W;
  const uint32_t out_row = (uint32_t)st.W * 4u, xoff = (uint32_t)x * 4u;
  for (int d = d0; d < d1; ++d) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)st.proj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst =
        __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)((uint32_t)rows * out_row), 0x00020000);
    if (staged) {
      // the previous projection's taps have been consumed (their values fed the stores): refill the slab
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane < 60) {
#pragma unroll 2
        for (int r = 0; r < bh; r += 3) {
          if (r + lrow < bh)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(box + r * kBoxW), 16, voff + (org + (uint32_t)r * rstep), 0,
                                                     0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (active) {
#pragma unroll
        for (int k = 0; k < kLdsTH; ++k) {
          FetchT f;
          f.fx = fx[k];
          f.fy = fy[k];
          DCP_BOUNDS(addr[k], kBoxW * 4 + 8, kBoxH * kBoxW * 4, 3);
          const float* t = (const float*)(boxb + addr[k]);
          f.a.x = __float_as_uint(t[0]);
          f.a.y = __float_as_uint(t[1]);
          f.b.x = __float_as_uint(t[kBoxW]);
          f.b.y = __float_as_uint(t[kBoxW + 1]);
          const float v = finish<SAMPLER, true, float>(f);
          if (k < rows) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
        }
      }
    } else if (active) {
#pragma unroll
      for (int k = 0; k < kLdsTH; ++k) {
        FetchT f;
        f.fx = fx[k];
        f.fy = fy[k];
        f.a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, addr[k], 0, 0);
        f.b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, addr[k], row_bytes, 0);
        const float v = finish<SAMPLER, true, float>(f);
        if (k < rows) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
      }
    }
    base += st.proj_stride;
    out += out_step;
  }
}

// ------------------------------------------------------------------ K4, workgroup-shared source box

// stack_lds_kernel's scheme with remap_wg_kernel's box: a workgroup owns a 128 x 32 tile of (x, row) positions of the
// requested rows (four waves, 2 x 2 sub-tiles of 64 x 16); corner pixels -> box (certificate level 2), per-pixel slab
// address and fractions ONCE; then for every projection of the depth chunk the four waves copy the box of THAT
// projection into one of two slabs (LDS-DMA, the loads of projection d + 1 issued before projection d is blended, so the
// copy runs under the arithmetic inside the workgroup) and blend their 1024 pixels each out of the other.  One barrier
// per projection.  T as in remap_wg_kernel: float (any blend), an 8- / 16- / 32-bit integer type or double (scipy's blend and its
// store) -- tomography detectors deliver uint16.  float32 coordinates only (unwarp_chunk_slices_backward).
#ifndef DCP_STACK_UNTRACKED_DMA
#define DCP_STACK_UNTRACKED_DMA 1   // 0: the fill through the compiler's LDS-DMA builtin (rounds 2-3; A/B) -- see lds_dma16_untracked
#endif
#ifndef DCP_STACK_UNTRACKED_F32
#define DCP_STACK_UNTRACKED_F32 0   // 1: float32 stacks too (A/B: 1.5 % slower)
#endif
#ifndef DCP_STACK_UNTRACKED_I32
#define DCP_STACK_UNTRACKED_I32 1   // 0: int32 / uint32 stacks through the builtin as float32 ones (A/B)
#endif
#ifndef DCP_STACK_INT_WAVES
#define DCP_STACK_INT_WAVES 4   // waves per SIMD the integer instantiations are allocated for (128 VGPRs; at 5 = 96 VGPRs the projection loop spills: 575 us against 435 per uint16 shard)
#endif
template <int NF, int SAMPLER, typename T = float>
__global__ void __launch_bounds__(256, (sizeof(T) >= 4 ? 3 : DCP_STACK_INT_WAVES)) stack_wg_kernel(const StackArgs st, const MapArgs map) {
  constexpr bool kIsF32 = std::is_same<T, float>::value;
  constexpr int ES = (int)sizeof(T);
  constexpr int CH = ES == 8 ? 72 : ES == 4 ? 36 : (ES == 2 ? 20 : 10);
  constexpr int PB = CH * 16;
  constexpr int kBoxWEl = PB / ES;
  constexpr int NJ = (kWgBoxH * CH + 255) / 256;
  // float64: ONE slab of 48 KB (two would leave a CU a single workgroup) -- fill, wait, blend, the other two workgroups of the CU
  // covering the wait
  constexpr int NSLAB = ES == 8 ? 1 : 2;
  constexpr int kAlignEl = ES >= 4 ? 1 : 4 / ES;               // the box's first column: a dword-aligned byte offset
  static_assert(kIsF32 || SAMPLER == kScipy, "integer and float64 element types blend in scipy's exact order");
  __shared__ __attribute__((aligned(16))) unsigned char s_box[NSLAB][kWgSlabRows * PB];
  __shared__ double s_row[4][kLdsTH][2];
  __shared__ double s_coef[NF < 0 ? kMaxFact : 1];
  using FetchT = Fetch<SAMPLER, true, float>;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int wx = wave & 1, wy = wave >> 1;
  // ---- which (tile column, tile row, depth chunk) this workgroup owns.  The dispatcher deals workgroups to the eight XCDs round robin
  // by their linear id (see logical_tile): in grid order the tiles left and right of this one stream THEIR boxes -- which share the
  // lines at this box's edges and, above and below, whole rows -- through other L2s.  xcd_order 1: XCD k takes the k-th eighth of the
  // (x fastest, then rows, then depth chunks) enumeration, so that the ~100 workgroups an XCD holds at a time are neighbouring tiles
  // of one depth chunk and walk its projections together; 2: whole tile rows (wg_stack_xcd_order has the measurements)
  int tile_x = (int)blockIdx.x, tile_y = (int)blockIdx.y, tile_z = (int)blockIdx.z;
  if (st.xcd_order) {
    const uint32_t gx = gridDim.x, gxy = gx * gridDim.y, n = gxy * gridDim.z;
    const uint32_t b = blockIdx.x + gx * blockIdx.y + gxy * blockIdx.z;
    const uint32_t xcd = b & 7u;
    if (st.xcd_order == 2) {
      // XCD k takes the tile rows k, k + 8, ... of every depth chunk (gridDim.y is the number of tile rows rounded up to a multiple of
      // eight: the workgroups of the rows that do not exist leave).  Left and right neighbours share an L2, the eight XCDs stay
      // inside one band of rows of one depth chunk as in grid order
      const uint32_t l = b - gxy * blockIdx.z, j = l >> 3, q = j / gx;
      tile_y = (int)(q * 8u + xcd);
      tile_x = (int)(j - q * gx);
      if (tile_y * kWgTH >= st.nrows) return;
    } else {
      uint32_t lin = xcd * (n >> 3) + min(xcd, n & 7u) + (b >> 3);
      tile_z = (int)(lin / gxy);
      lin -= (uint32_t)tile_z * gxy;
      tile_y = (int)(lin / gx);
      tile_x = (int)(lin - (uint32_t)tile_y * gx);
    }
  }
  tile_x = __builtin_amdgcn_readfirstlane(tile_x);
  tile_y = __builtin_amdgcn_readfirstlane(tile_y);
  tile_z = __builtin_amdgcn_readfirstlane(tile_z);
  const int rblk = tile_y * kWgTH;
  const int r0 = __builtin_amdgcn_readfirstlane(rblk + wy * kLdsTH);   // first requested row of this wave's sub-tile
  const int x = tile_x * kWgTW + wx * kLdsTW + lane;
  const int d0 = tile_z * st.d_chunk, d1 = min(st.D, d0 + st.d_chunk);
  const float wmaxf = (float)(st.W - 1), hmaxf = (float)(st.H - 1);
  const T* const volT = (const T*)st.vol;
  T* const outT = (T*)st.out;

  // ---- the tile's four corner pixels (lanes 0..3, every wave for itself) -> box
  int cx0, cx1, cy0, cy1;
  {
    const double X = (double)min(tile_x * kWgTW + (lane & 1) * (kWgTW - 1), st.W - 1);
    const double Y = st.row_start + (double)min(rblk + ((lane >> 1) & 1) * (kWgTH - 1), st.nrows - 1);
    const double xu = X - map.xc, yu = Y - map.yc;
    const double r2 = xu * xu + yu * yu;
    const double ru = sqrt_rn(r2);
    double f;
    if constexpr (NF >= 0) {
      double le, lo;
      poly_leads<NF>(map.fact, &le, &lo);
      f = poly_inline<NF>(map.fact, le, lo, r2, ru);
    } else {
      f = poly_lds(map.fact, map.nfact, r2, ru);
    }
    const int cxi = (int)round_clip_f32(__builtin_fma(f, xu, map.xc), wmaxf), cyi = (int)round_clip_f32(__builtin_fma(f, yu, map.yc), hmaxf);
    const int xa = __builtin_amdgcn_readlane(cxi, 0), xb = __builtin_amdgcn_readlane(cxi, 1);
    const int xc_ = __builtin_amdgcn_readlane(cxi, 2), xd_ = __builtin_amdgcn_readlane(cxi, 3);
    const int ya = __builtin_amdgcn_readlane(cyi, 0), yb = __builtin_amdgcn_readlane(cyi, 1);
    const int yc_ = __builtin_amdgcn_readlane(cyi, 2), yd_ = __builtin_amdgcn_readlane(cyi, 3);
    cx0 = min(min(xa, xb), min(xc_, xd_));
    cx1 = max(max(xa, xb), max(xc_, xd_));
    cy0 = min(min(ya, yb), min(yc_, yd_));
    cy1 = max(max(ya, yb), max(yc_, yd_));
  }
  const int bx0 = max(min(cx0 - 1, st.W - 2), 0) & ~(kAlignEl - 1);
  const int bx1 = min(cx1 + 2, st.W - 1);
  const int by0 = max(min(cy0 - 1, st.H - 2), 0);
  const int by1 = min(cy1 + 2, st.H - 1);
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  const bool fits = bw <= kBoxWEl && bh <= kWgBoxH;          // workgroup-uniform
  if (!fits && lane == 0 && wave == 0) atomicAdd(&g_lds_stats[0], 1ull);
  // integer elements: the tile's coordinates are all >= its box origin; at >= 32 the factorised blend is exact (exact_lerp_pairs)
  // (32-bit integers: the products of the factorised blend do not fit a float64 exactly -- scipy's order, as for float32)
  const bool exact = !kIsF32 && ES < 4 && SAMPLER != kNearest && fits && st.int_exact && bx0 >= (int)kExactLerpMinCoord && by0 >= (int)kExactLerpMinCoord;

  // ---- coordinates of this wave's 16 rows, once for all projections: slab address (or byte offset inside a projection
  // when the box does not fit) and fractions
  if (lane < kLdsTH) fill_row<kRadial, 2>(map, s_row[wave], lane, st.row_start + (double)min(r0 + lane, st.nrows - 1));
  if constexpr (NF < 0) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
    __syncthreads();
  }
  const int rows = __builtin_amdgcn_readfirstlane(max(0, min(kLdsTH, st.nrows - r0)));
  const ColCtx col = make_col<kRadial, NF>(map, min(x, st.W - 1));
  uint32_t addr[kLdsTH];
  // the two float32 fractions per pixel stay in registers for all projections.  (Integer element types: round 2 kept scipy's four
  // float64 weights instead -- 128 of 168 VGPRs, three waves per SIMD and spills; the conversions back to float64 cost two
  // instructions per voxel and buy five waves per SIMD.)
  float fx[kLdsTH], fy[kLdsTH];
  const uint32_t negorg = (uint32_t)(-(by0 * PB + bx0 * ES));
#pragma unroll
  for (int k = 0; k < kLdsTH; ++k) {
    double xd, yd;
    map_coord<kRadial, NF, 2>(map, s_row[wave], s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
    const float xc = round_clip_f32(xd, wmaxf), yc = round_clip_f32(yd, hmaxf);
    int xi, yi;
    if constexpr (SAMPLER == kNearest) {
      xi = (int)xc;
      yi = (int)yc;
      xi += (xc - (float)xi >= 0.5f) ? 1 : 0;
      yi += (yc - (float)yi >= 0.5f) ? 1 : 0;
      fx[k] = fy[k] = 0.0f;
    } else {
      xi = min((int)xc, st.W - 2);
      yi = min((int)yc, st.H - 2);
      fx[k] = xc - (float)xi;
      fy[k] = yc - (float)yi;
    }
    addr[k] = fits ? (uint32_t)yi * (uint32_t)PB + (uint32_t)xi * (uint32_t)ES + negorg
                   : ((uint32_t)yi * (uint32_t)st.row_stride + (uint32_t)xi) * (uint32_t)ES;
  }

  // ---- the fill of one projection: this wave's chunks (see remap_wg_kernel)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const uint32_t rstep = (uint32_t)st.row_stride * (uint32_t)ES;
  const int fc = wave * 64 + lane;
  const int crow0 = fc / CH;
  const int c160 = fc - crow0 * CH;
  const uint32_t off0 = ((uint32_t)by0 * (uint32_t)st.row_stride + (uint32_t)bx0) * (uint32_t)ES + (uint32_t)crow0 * rstep + (uint32_t)c160 * 16u;
  const int nchunk = bh * CH;
  // lds_dma16_untracked: seen by the compiler, the stream of projection d + 1 is waited for before the blend of projection d.  Integer
  // stacks (bound by that chain's latency) gain 3-9 % without the wait; float32 stacks (at the rate the box copies memory at) lose 1 %:
  // they keep the builtin
  // (one slab -- float64: the wait behind the fill is for everything, the builtin says so to the compiler)
  constexpr bool kUntrackedFill = DCP_STACK_UNTRACKED_DMA && NSLAB == 2 && ((!kIsF32 && (ES < 4 || DCP_STACK_UNTRACKED_I32)) || DCP_STACK_UNTRACKED_F32);
  [[maybe_unused]] const uint32_t slab0 = (uint32_t)(uintptr_t)(lds_ptr)&s_box[0][0];
  auto fill = [&](const T* proj, int slab) {
    [[maybe_unused]] const dcp_rsrc_words rs = raw_rsrc_words(proj, st.proj_bytes);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_seen = __builtin_amdgcn_make_buffer_rsrc((void*)proj, 0, (int)st.proj_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if ((j * 4 + wave) * 64 < nchunk) {
        const int qrow = (256 * j) / CH, rem = (256 * j) % CH;
        const bool wrap = c160 >= CH - rem;
        const int crow = crow0 + qrow + (wrap ? 1 : 0);
        const uint32_t voff = off0 + (wrap ? rstep - (uint32_t)PB : 0u) + (uint32_t)qrow * rstep + (uint32_t)rem * 16u;
        if (crow < bh) {
          if constexpr (kUntrackedFill) lds_dma16_untracked(rs, slab0 + (uint32_t)(slab * (kWgSlabRows * PB) + (j * 4 + wave) * 1024), voff);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_seen, (lds_ptr)(s_box[slab] + (j * 4 + wave) * 1024), 16, voff, 0, 0, 0);
        }
      }
    }
  };

  const bool active = x < st.W && rows > 0;
  const uint32_t out_row = (uint32_t)st.W * (uint32_t)ES, xoff = (uint32_t)x * (uint32_t)ES;
  const T* proj = volT + (size_t)d0 * (size_t)st.proj_stride;
  T* out = outT + ((size_t)d0 * (size_t)st.nrows + (size_t)min(r0, st.nrows - 1)) * (size_t)st.W;
  const size_t out_step = (size_t)st.nrows * (size_t)st.W;
  // (bounds-checking build: the vector-memory instructions this lane's wave has issued since its share of the last fill -- the lazy
  // wait below is right only if there were at least kLdsTH of them)
  DCP_VM_COUNTER(vm_since_fill);
  auto blend_store = [&](const T* t_lo, const T* t_hi, const __amdgpu_buffer_rsrc_t& dst, int k) {
    // t_lo / t_hi: the tap pairs of the two rows (LDS or global); a_: the byte address (for the aligned-dword extraction)
    if constexpr (kIsF32) {
      FetchT f;
      f.fx = fx[k];
      f.fy = fy[k];
      // (scipy's order needs four float64 weights per pixel: hoisted out of the projection loop they are 128 registers and the
      // kernel spills -- the empty asm keeps their computation inside the loop)
      if constexpr (SAMPLER == kScipy) asm volatile("" : "+v"(f.fx), "+v"(f.fy));
      f.a.x = __float_as_uint(t_lo[0]);
      if constexpr (SAMPLER != kNearest) {
        f.a.y = __float_as_uint(t_lo[1]);
        f.b.x = __float_as_uint(t_hi[0]);
        f.b.y = __float_as_uint(t_hi[1]);
      }
      const float v = finish<SAMPLER, true, float>(f);
      if (k < rows) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
        DCP_VM_ISSUED(vm_since_fill);
      }
    } else {
      // scipy NI_GeometricTransform: w0 = 1 - f, w1 = 1 - w0 (not f itself where 1 - f rounds); ((v * wy) * wx) summed left to right
      // (the empty asm keeps the conversions and the weights inside the projection loop: hoisted, they are 128 registers again)
      float fy_ = fy[k], fx_ = fx[k];
      asm volatile("" : "+v"(fy_), "+v"(fx_));
      const double wy0_ = 1.0 - (double)fy_, wy1_ = 1.0 - wy0_, wx0_ = 1.0 - (double)fx_, wx1_ = 1.0 - wx0_;
      double acc = ((double)t_lo[0] * wy0_) * wx0_;
      acc += ((double)t_lo[1] * wy0_) * wx1_;
      acc += ((double)t_hi[0] * wy1_) * wx0_;
      acc += ((double)t_hi[1] * wy1_) * wx1_;
      const T v = to_elem<T>(acc);
      if (k < rows) {
        DCP_VM_ISSUED(vm_since_fill);
        if constexpr (ES == 8) {
          typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
          const unsigned long long b = (unsigned long long)__double_as_longlong((double)v);
          const u32x2_t pk = {(uint32_t)b, (uint32_t)(b >> 32)};
          __builtin_amdgcn_raw_buffer_store_b64(pk, dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
        } else if constexpr (ES == 4) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v, dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
        else if constexpr (ES == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
        else __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
      }
    }
  };

  static_assert(kLdsTH == 16, "the partial wait below counts the stores of one projection");
  const bool lazy_wait = kUntrackedFill && st.store_wait && rows == kLdsTH && __builtin_amdgcn_readfirstlane((int)(__ballot(active) != 0ull));
  if (fits) {
    if constexpr (NSLAB == 2) fill(proj, 0);
    for (int d = d0; d < d1; ++d) {
      const int cur = NSLAB == 2 ? ((d - d0) & 1) : 0;
      if constexpr (NSLAB == 1) {
        __syncthreads();                                    // everyone is done with the slab's previous projection
        fill(proj, 0);
      }
      // this wave's share of projection d.  Vector memory operations of a wave complete in issue order (one counter for loads and stores
      // on gfx9-class hardware), and what the wave issued AFTER that share are the stores of projection d - 1: exactly kLdsTH of them
      // when the wave has all its rows and a lane inside the image -- then "at most kLdsTH outstanding" means the fill has landed and
      // the stores stay in flight under the barrier, the next fill and the blend (waiting for them too cost a write round trip per
      // projection).  Fewer stores than that (ragged waves): wait for everything
      if (lazy_wait && d > d0) {
        DCP_VM_CHECK(active, vm_since_fill, kLdsTH);                                     // (checking build only)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // (d0: no stores behind the first fill yet)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();                                      // everyone's share has landed; everyone is done with the other slab
      if constexpr (NSLAB == 2) {
        if (d + 1 < d1) fill(proj + st.proj_stride, cur ^ 1); // projection d + 1 streams in under the blend of d
        DCP_VM_RESET(vm_since_fill);
      }
      if (active) {
        const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)((uint32_t)rows * out_row), 0x00020000);
        const char* boxb = (const char*)s_box[cur];
        // (the pointers advance below, in uniform control flow: advanced inside this per-lane branch they -- and the store descriptor built
        // from them -- become per-lane values and every store a waterfall loop)
        [[maybe_unused]] bool done = false;
        if constexpr (!kIsF32 && ES < 4) {
          if (exact) {
            // every coordinate of the tile >= 32: the factorised blend is exact (exact_lerp_pairs) and scipy's w1 = 1 - (1 - f) IS the fraction
#pragma unroll
            for (int k = 0; k < kLdsTH; ++k) {
              DCP_BOUNDS(addr[k], PB + 2 * ES, kWgSlabRows * PB, 4);
              const T* t = (const T*)(boxb + addr[k]);
              float fy_ = fy[k], fx_ = fx[k];
              asm volatile("" : "+v"(fy_), "+v"(fx_));            // (see blend_store)
              const T v = to_elem_in_range<T>(exact_lerp_taps<T>(t, t + kBoxWEl, (double)fx_, (double)fy_));
              if (k < rows) {
                DCP_VM_ISSUED(vm_since_fill);
                if constexpr (ES == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
                else __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, dst, xoff, (uint32_t)k * out_row, DCP_STORE_AUX);
              }
            }
            done = true;
          }
        }
        if (!done) {
#pragma unroll
        for (int k = 0; k < kLdsTH; ++k) {
          DCP_BOUNDS(addr[k], PB + 2 * ES, kWgSlabRows * PB, 5);
          if constexpr (ES >= 4) {                          // float32, int32, uint32, float64: the taps are aligned elements
            const T* t = (const T*)(boxb + addr[k]);
            blend_store(t, t + kBoxWEl, dst, k);
          } else {
            // tap pairs as the two aligned dwords around them, shifted down (see remap_wg_kernel)
            const uint32_t* q = (const uint32_t*)(boxb + (addr[k] & ~3u));
            const uint32_t sh = (addr[k] & 3u) * 8u;
            const uint32_t top = __builtin_amdgcn_alignbit(q[1], q[0], sh), bot = __builtin_amdgcn_alignbit(q[PB / 4 + 1], q[PB / 4], sh);
            T lo[2], hi[2];
            if constexpr (std::is_signed<T>::value) {
              lo[0] = (T)__builtin_amdgcn_sbfe(top, 0, ES * 8); lo[1] = (T)__builtin_amdgcn_sbfe(top, ES * 8, ES * 8);
              hi[0] = (T)__builtin_amdgcn_sbfe(bot, 0, ES * 8); hi[1] = (T)__builtin_amdgcn_sbfe(bot, ES * 8, ES * 8);
            } else {
              lo[0] = (T)__builtin_amdgcn_ubfe(top, 0, ES * 8); lo[1] = (T)__builtin_amdgcn_ubfe(top, ES * 8, ES * 8);
              hi[0] = (T)__builtin_amdgcn_ubfe(bot, 0, ES * 8); hi[1] = (T)__builtin_amdgcn_ubfe(bot, ES * 8, ES * 8);
            }
            blend_store(lo, hi, dst, k);
          }
        }
        }
      }
      proj += st.proj_stride;
      out += out_step;
    }
  } else if (active) {
    // ---- box too large for the slab: direct global gather, projection by projection
    for (int d = d0; d < d1; ++d) {
      const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)((uint32_t)rows * out_row), 0x00020000);
#pragma unroll
      for (int k = 0; k < kLdsTH; ++k) {
        const T* t = (const T*)((const char*)proj + addr[k]);
        blend_store(t, t + st.row_stride, dst, k);
      }
      proj += st.proj_stride;
      out += out_step;
    }
  }
}

// ------------------------------------------------------------------ launchers

// Name of the kernel the calling thread launched last (dcp_debug_last_kernel): tests and bench.py use it to state --
// and assert -- which kernel a call really took.
static thread_local char g_last_kernel[96] = "";
static const char* kind_name(int k) { return k == kRadial ? "Radial" : k == kPersp ? "Persp" : "Fused"; }
static const char* sampler_name(int s) { return s == kNearest ? "nearest" : s == kScipy ? "scipy" : s == kF64Lerp ? "f64lerp" : "f32lerp"; }
static void note_kernel(const char* kernel, int kind, int nf, int sampler, const char* extra = "") {
  if (kind >= 0) snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<%s,NF=%d,%s%s>", kernel, kind_name(kind), nf, sampler_name(sampler), extra);
  else snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<NF=%d,%s%s>", kernel, nf, sampler_name(sampler), extra);
}
const char* last_kernel_name() { return g_last_kernel; }

// Whole-frame launches of the staged kernels.  LaunchOpts::any_order (DCP_MEM_DEVICE_UNORDERED of the C ABI) clears the barrier bit
// of the dispatch packet (hipExtAnyOrderLaunch): the command processor may then start the workgroups of this frame while the last
// workgroups of the previous packet of the SAME stream still run -- the drain of frame z under the ramp of frame z + 1, what
// remap_wg_batch_kernel gets from blockIdx.z, for callers that hand over one frame per call.  Set per launch_image / launch_wg_typed call.
static thread_local int t_any_order = 0;
#include <hip/hip_ext.h>
#define DCP_LAUNCH_FRAME(kernel, grid, block, lds, stream, ...)                                                                  \
  do {                                                                                                                            \
    if (t_any_order) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, nullptr, hipExtAnyOrderLaunch, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                       \
  } while (0)
void set_last_kernel_name(const char* name) { snprintf(g_last_kernel, sizeof(g_last_kernel), "%s", name); }

template <int KIND, int NF, int SAMPLER, bool ROUND32, bool PAIR>
static hipError_t launch_one(const ImageArgs& img, const MapArgs& map, hipStream_t stream) {
  const int nb = img.tiles_x * img.tiles_y;
  note_kernel("remap_tile_kernel", KIND, NF, SAMPLER, ROUND32 ? (PAIR ? "" : ",strided") : ",f64coords");
  if constexpr (KIND == kRadial && NF >= 0) {
    // the headline path also exists with 1 and 4 rows in flight (option pipe_depth) for A/B runs
    if (img.pipe_depth == 1) {
      hipLaunchKernelGGL((remap_tile_kernel<KIND, NF, SAMPLER, ROUND32, PAIR, 1>), dim3(nb), dim3(kBlock), 0,
                         stream, img, map);
      return hipGetLastError();
    }
    if (img.pipe_depth >= 4) {
      hipLaunchKernelGGL((remap_tile_kernel<KIND, NF, SAMPLER, ROUND32, PAIR, 4>), dim3(nb), dim3(kBlock), 0,
                         stream, img, map);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((remap_tile_kernel<KIND, NF, SAMPLER, ROUND32, PAIR, kPipeDepth>), dim3(nb), dim3(kBlock),
                     0, stream, img, map);
  return hipGetLastError();
}

// The staged kernel forms LDS addresses in float32: every term must be an integer below 2^24.
static bool lds_addressable(const ImageArgs& img) {
  return (int64_t)img.H * (kWgBoxW * 4) + (int64_t)img.W * 4 + 4 * (int64_t)(kLdsBW * kBoxH * kBoxW * 4) < (1 << 24);
}

template <int KIND, int NF, int SAMPLER, bool VOTE>
static hipError_t launch_lds(const ImageArgs& img_in, const MapArgs& map, hipStream_t stream) {
  ImageArgs img = img_in;
  img.tiles_x = (img.W + kLdsTW - 1) / kLdsTW;
  img.tiles_y = (img.rows_out + kLdsBW * kLdsTH - 1) / (kLdsBW * kLdsTH);
  dim3 grid(img.tiles_x * img.tiles_y);
  if (img.xcd_remap == 2) grid = dim3(8 * ((img.tiles_x + 7) / 8), img.tiles_y);   // see the kernel's tile order
  note_kernel("remap_lds_kernel", KIND, NF, SAMPLER, VOTE ? ",vote" : ",certified");
  DCP_LAUNCH_FRAME((remap_lds_kernel<KIND, NF, SAMPLER, VOTE>), grid, dim3(64 * kLdsBW), 0, stream, img, map);
  return hipGetLastError();
}

template <int KIND, int NF, int SAMPLER>
static hipError_t launch_wg(const ImageArgs& img_in, const MapArgs& map, hipStream_t stream) {
  ImageArgs img = img_in;
  img.tiles_x = (img.W + kWgTW - 1) / kWgTW;
  img.tiles_y = (img.rows_out + kWgTH - 1) / kWgTH;
  // XCD stripes of whole tile columns share source lines inside an L2 (4096 px: 3 % faster than the plain order), but only
  // balance when the number of tile columns divides by eight: 20 columns (2560 px) put 3 on four XCDs and 2 on the other
  // four, 7 % slower than the plain order.  Stripes when the widest XCD carries at most 7 % more than the average.
  if (img.xcd_remap == 2 && 8 * ((img.tiles_x + 7) / 8) * 100 > img.tiles_x * 107) img.xcd_remap = 0;
  const dim3 grid(img.xcd_remap == 2 ? 8 * ((img.tiles_x + 7) / 8) : img.tiles_x, img.tiles_y);
  // workgroups per CU capped through unused dynamic LDS (the static 23.5 KB allow six): img.wg_per_cu in 1..5
  unsigned pad = 0;
  if (img.wg_per_cu >= 1 && img.wg_per_cu <= 5) pad = (unsigned)(160 * 1024 / img.wg_per_cu - 24 * 1024) & ~255u;
  note_kernel("remap_wg_kernel", KIND, NF, SAMPLER);
  if (pad > 32768u) {        // beyond the 64 KB a launch may ask for by default (static 24.6 KB + the padding)
    const hipError_t e = hipFuncSetAttribute((const void*)remap_wg_kernel<KIND, NF, SAMPLER, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
    if (e != hipSuccess) return e;
  }
  DCP_LAUNCH_FRAME((remap_wg_kernel<KIND, NF, SAMPLER>), grid, dim3(256), pad, stream, img, map);
  return hipGetLastError();
}

// Many frames of one shape, each with its own calibration, in as few launches as the 4 KB of kernel arguments allow
// (55 frames of <= 5 coefficients, 35 of <= 10).  Coefficient vectors shorter than the instantiated length are padded with
// zeros: fma(r2, 0, a) = a exactly, so every intermediate of the even / odd Horner chains is unchanged.
static int g_batch_box_table = 1;      // option box_table (A/B switch)
void set_box_table(int v) { g_batch_box_table = v; }
int get_box_table() { return g_batch_box_table; }

template <int NF, int SAMPLER>
static hipError_t launch_wg_batch_t(const ImageArgs& img_in, const BatchFrame* fr, int n, int nfact, hipStream_t stream) {
  using Tab = BatchTable<NF>;
  static_assert(sizeof(ImageArgs) + sizeof(Tab) + 16 <= 4096, "kernel arguments are limited to 4 KB");
  ImageArgs img = img_in;
  img.tiles_x = (img.W + kWgTW - 1) / kWgTW;
  img.tiles_y = (img.rows_out + kWgTH - 1) / kWgTH;
  if (img.xcd_remap == 2 && 8 * ((img.tiles_x + 7) / 8) * 100 > img.tiles_x * 107) img.xcd_remap = 0;       // see launch_wg
  note_kernel("remap_wg_batch_kernel", kRadial, NF, SAMPLER);
  for (int f0 = 0; f0 < n; f0 += Tab::kMax) {
    const int m = n - f0 < Tab::kMax ? n - f0 : Tab::kMax;
    Tab tab;
    memset(&tab, 0, sizeof(tab));
    for (int i = 0; i < m; ++i) {
      const BatchFrame& f = fr[f0 + i];
      tab.e[i].src = f.src;
      tab.e[i].dst = f.dst;
      tab.e[i].xc = f.xc;
      tab.e[i].yc = f.yc;
      for (int k = 0; k < nfact; ++k) tab.e[i].fact[k] = f.fact[k];
    }
    const dim3 grid(img.xcd_remap == 2 ? 8 * ((img.tiles_x + 7) / 8) : img.tiles_x, img.tiles_y, m);
    // the tiles' corner hulls once per tile instead of in every wave, when there are enough frames to pay for the extra launch
    // (stream-ordered scratch: allocated, written, read and freed in the stream's order; without it the waves evaluate the corners)
    const int ntiles = img.tiles_x * img.tiles_y;
    int32_t* boxes = nullptr;
    if (g_batch_box_table && m >= 4 && hipMallocAsync((void**)&boxes, (size_t)m * ntiles * 16, stream) != hipSuccess) {
      (void)hipGetLastError();
      boxes = nullptr;
    }
    img.boxes = boxes;
    if (boxes) hipLaunchKernelGGL((box_table_kernel<NF>), dim3((ntiles + 63) / 64, 1, m), dim3(64), 0, stream, img, tab, boxes);
    hipLaunchKernelGGL((remap_wg_batch_kernel<NF, SAMPLER>), grid, dim3(256), 0, stream, img, tab);
    const hipError_t e = hipGetLastError();
    if (boxes) (void)hipFreeAsync(boxes, stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <int NF>
static hipError_t launch_wg_batch_s(const ImageArgs& img, const BatchFrame* fr, int n, int nfact, int sampler, hipStream_t stream) {
  switch (sampler) {
    case kNearest: return launch_wg_batch_t<NF, kNearest>(img, fr, n, nfact, stream);
    case kScipy: return launch_wg_batch_t<NF, kScipy>(img, fr, n, nfact, stream);
    case kF64Lerp: return launch_wg_batch_t<NF, kF64Lerp>(img, fr, n, nfact, stream);
    default: return launch_wg_batch_t<NF, kF32Lerp>(img, fr, n, nfact, stream);
  }
}

hipError_t launch_image_batch(const ImageArgs& img_in, const BatchFrame* frames, int n, int nfact, int sampler, const LaunchOpts& opts,
                              hipStream_t stream, bool* taken) {
  *taken = false;
  ImageArgs img = img_in;
  if (img.rows_out <= 0) {
    img.y_origin = 0;
    img.rows_out = img.H;
  }
  img.xcd_remap = opts.xcd_remap;
  img.wg_box = opts.wg_box;
  img.wg_per_cu = 0;
  // what remap_wg_kernel needs (launch_image / launch_lds_vote): the LDS-staged pair gather with one box per workgroup
  if (n <= 0 || nfact < 0 || nfact > 10 || !opts.wg_box || !opts.lds_gather || opts.coef_lds || opts.xcd_remap == 1 || img.src_col_stride != 1 ||
      img.W < 2 || img.H < 2 || img.src_stride >= (1 << 22) || img.H >= (1 << 24) || !lds_addressable(img))
    return hipSuccess;
  *taken = true;
  if (nfact <= 5) return launch_wg_batch_s<5>(img, frames, n, nfact, sampler, stream);
  return launch_wg_batch_s<10>(img, frames, n, nfact, sampler, stream);
}

// Narrow integer element types on remap_wg_kernel (typed entry points of the C ABI, orders 0 / 1): taken = false when the
// call does not qualify (no level-2 certificate, column-strided or unaligned source, image too wide for float32 LDS
// addresses) and the generic one-thread-per-pixel kernels of typed_kernels.hip must serve it.
template <int KIND, int NF, typename T>
static hipError_t launch_wg_typed_t(const ImageArgs& img_in, const MapArgs& map, int order, hipStream_t stream) {
  ImageArgs img = img_in;
  img.tiles_x = (img.W + kWgTW - 1) / kWgTW;
  img.tiles_y = (img.rows_out + kWgTH - 1) / kWgTH;
  if (8 * ((img.tiles_x + 7) / 8) * 100 > img.tiles_x * 107) img.xcd_remap = 0;       // see launch_wg
  const dim3 grid(img.xcd_remap == 2 ? 8 * ((img.tiles_x + 7) / 8) : img.tiles_x, img.tiles_y);
  if (order == 0) {
    note_kernel("remap_wg_kernel", KIND, NF, kNearest, sizeof(T) == 2 ? ",16-bit" : ",8-bit");
    DCP_LAUNCH_FRAME((remap_wg_kernel<KIND, NF, kNearest, T>), grid, dim3(256), 0, stream, img, map);
  } else {
    note_kernel("remap_wg_kernel", KIND, NF, kScipy, sizeof(T) == 2 ? ",16-bit" : ",8-bit");
    DCP_LAUNCH_FRAME((remap_wg_kernel<KIND, NF, kScipy, T>), grid, dim3(256), 0, stream, img, map);
  }
  return hipGetLastError();
}

// A coefficient vector of fewer than four terms through the NF = 4 instantiations: padded with zeros -- fma(r2, 0, a) = a exactly, so
// every intermediate of the even / odd Horner chains, and the result, is unchanged (as in launch_wg_batch_t) -- instead of the
// run-time-length form (coefficients from LDS, more registers).
static MapArgs pad4(const MapArgs& m) {
  MapArgs p = m;
  for (int i = p.nfact < 0 ? 0 : p.nfact; i < 4; ++i) p.fact[i] = 0.0;
  if (p.nfact < 4) p.nfact = 4;
  return p;
}

template <typename T>
static hipError_t launch_wg_typed_k(MapKind kind, const ImageArgs& img, const MapArgs& map, int order, hipStream_t stream) {
  if (kind == kPersp) return launch_wg_typed_t<kPersp, -1, T>(img, map, order, stream);
  if (map.nfact == 5) return launch_wg_typed_t<kRadial, 5, T>(img, map, order, stream);
  if (map.nfact <= 4) return launch_wg_typed_t<kRadial, 4, T>(img, pad4(map), order, stream);
  return launch_wg_typed_t<kRadial, -1, T>(img, map, order, stream);
}

hipError_t launch_wg_typed(MapKind kind, const ImageArgs& img_in, const MapArgs& map, int order, int dtype, const LaunchOpts& opts,
                           hipStream_t stream, bool* taken) {
  *taken = false;
  t_any_order = opts.any_order;
  const int es = elem_size(dtype);
  if ((kind != kRadial && kind != kPersp) || (order != 0 && order != 1) || map.tile_dev_ok < 2 || !opts.wg_box || !opts.lds_gather ||
      opts.xcd_remap != 2 || opts.coef_lds)
    return hipSuccess;
  if (dtype != kU8 && dtype != kI8 && dtype != kU16 && dtype != kI16) return hipSuccess;
  ImageArgs img = img_in;
  if (img.rows_out <= 0) {
    img.y_origin = 0;
    img.rows_out = img.H;
  }
  img.int_exact = opts.int_exact;
  // unit column stride, at least 2 x 2, 4-byte aligned rows (the 16-byte LDS-DMA copies), 24-bit products, float32 LDS addresses
  if (img.src_col_stride != 1 || img.W < 2 || img.H < 2 || ((uintptr_t)img.src & 3u) || (((int64_t)img.src_stride * es) & 3) ||
      img.src_stride >= (1 << 22) || img.H >= (1 << 24) || !lds_addressable(img))
    return hipSuccess;
  *taken = true;
  switch (dtype) {
    case kU8: return launch_wg_typed_k<uint8_t>(kind, img, map, order, stream);
    case kI8: return launch_wg_typed_k<int8_t>(kind, img, map, order, stream);
    case kU16: return launch_wg_typed_k<uint16_t>(kind, img, map, order, stream);
    default: return launch_wg_typed_k<int16_t>(kind, img, map, order, stream);
  }
}

// VOTE = false needs the host's certificate (MapArgs::tile_dev_ok); without it the runtime-length polynomial
// variant with the per-pixel vote runs (one uncertified instantiation per map kind and sampler instead of eleven).
// The fused map always votes: its inner clip (the perspective position clipped to the frame before the radial model
// is applied) makes the composed map non-smooth, so no curvature bound holds for it.
template <int KIND, int NF, int SAMPLER>
static hipError_t launch_lds_vote(const ImageArgs& img, const MapArgs& map, hipStream_t stream) {
  if constexpr (KIND == kFused) {
    // (round 5) a tame homography in front of a radial model of certified curvature: one box per 128 x 32 workgroup tile from the
    // corners of the tile's perspective bounding box (wg_corner_tap) -- no per-pixel vote.  Anything else votes.
    if constexpr (NF >= 0) {
      if (map.tile_dev_ok >= 2 && img.wg_box && img.xcd_remap != 1) return launch_wg<KIND, NF, SAMPLER>(img, map, stream);
    }
    return launch_lds<KIND, NF, SAMPLER, true>(img, map, stream);
  } else {
    // certificate levels: 2 = holds for 128 x 32 tiles (workgroup-shared box), 1 = for 64 x 16 tiles only
    if (map.tile_dev_ok >= 2 && img.wg_box && img.xcd_remap != 1) return launch_wg<KIND, NF, SAMPLER>(img, map, stream);
    if (map.tile_dev_ok) return launch_lds<KIND, NF, SAMPLER, false>(img, map, stream);
    return launch_lds<KIND, -1, SAMPLER, true>(img, map, stream);
  }
}

template <int KIND, int NF>
static hipError_t launch_lds_any(const ImageArgs& img, const MapArgs& map, int sampler, hipStream_t stream) {
  switch (sampler) {
    case kNearest: return launch_lds_vote<KIND, NF, kNearest>(img, map, stream);
    case kScipy: return launch_lds_vote<KIND, NF, kScipy>(img, map, stream);
    case kF64Lerp: return launch_lds_vote<KIND, NF, kF64Lerp>(img, map, stream);
    default: return launch_lds_vote<KIND, NF, kF32Lerp>(img, map, stream);
  }
}

template <int KIND, int NF>
static hipError_t launch_fast(const ImageArgs& img, const MapArgs& map, int sampler, hipStream_t stream) {
  if (img.lds_gather) return launch_lds_any<KIND, NF>(img, map, sampler, stream);
  switch (sampler) {
    case kNearest: return launch_one<KIND, NF, kNearest, true, true>(img, map, stream);
    case kScipy: return launch_one<KIND, NF, kScipy, true, true>(img, map, stream);
    case kF64Lerp: return launch_one<KIND, NF, kF64Lerp, true, true>(img, map, stream);
    default: return launch_one<KIND, NF, kF32Lerp, true, true>(img, map, stream);
  }
}

template <int KIND, bool ROUND32, bool PAIR>
static hipError_t launch_generic(const ImageArgs& img, const MapArgs& map, int sampler, hipStream_t stream) {
  switch (sampler) {
    case kNearest: return launch_one<KIND, -1, kNearest, ROUND32, PAIR>(img, map, stream);
    case kScipy: return launch_one<KIND, -1, kScipy, ROUND32, PAIR>(img, map, stream);
    case kF64Lerp: return launch_one<KIND, -1, kF64Lerp, ROUND32, PAIR>(img, map, stream);
    default: return launch_one<KIND, -1, kF32Lerp, ROUND32, PAIR>(img, map, stream);
  }
}

hipError_t launch_image(MapKind kind, const ImageArgs& img_in, const MapArgs& map, int sampler, bool round_f32,
                        const LaunchOpts& opts, hipStream_t stream) {
  ImageArgs img = img_in;
  t_any_order = opts.any_order;
  int tr = opts.tile_rows;
  if (tr < 1) tr = 1;
  if (tr > kMaxTileRows) tr = kMaxTileRows;
  img.tile_rows = tr;
  if (img.rows_out <= 0) {          // the whole image
    img.y_origin = 0;
    img.rows_out = img.H;
  }
  img.tiles_x = (img.W + kBlock - 1) / kBlock;
  img.tiles_y = (img.rows_out + tr - 1) / tr;
  img.xcd_remap = opts.xcd_remap;
  img.pipe_depth = opts.pipe_depth;
  img.lds_gather = opts.lds_gather && lds_addressable(img);
  img.wg_box = opts.wg_box;
  img.wg_per_cu = opts.wg_per_cu;
  // the 8-byte pair gather needs unit column stride and at least a 2x2 image; its row offsets are 24-bit products
  // (v_mul_u32_u24), so a source whose row stride reaches 2^22 elements (16 MB rows: a column-plane view of a large
  // volume) or whose height reaches 2^24 takes the strided kernels with full 32-bit multiplies instead
  const bool pair = img.src_col_stride == 1 && img.W >= 2 && img.H >= 2 && img.src_stride < (1 << 22) && img.H < (1 << 24);
  const int nf = map.nfact;

  // order-1 remaps of dense float32 images with float32 coordinates: LDS-staged gather
  const bool lds = pair && round_f32 && img.lds_gather;
  if (kind == kPersp) {
    if (lds) return launch_lds_any<kPersp, -1>(img, map, sampler, stream);
    if (pair) return launch_generic<kPersp, true, true>(img, map, sampler, stream);
    return launch_generic<kPersp, true, false>(img, map, sampler, stream);
  }
  if (kind == kRadial) {
    // a sheared map (the certificate holds, but the boxes of 128 x 32 tiles overflow the slab: a fisheye model far from its
    // centre) on 64 x 32 workgroup tiles under an 80 x 56 box -- color_kernels.hip, one channel -- instead of per-wave boxes
    if (pair && round_f32 && !opts.coef_lds && img.lds_gather && img.wg_box && opts.tall_tiles && map.tile_dev_ok == 1 && map.tall_ok &&
        opts.xcd_remap != 1 && sampler != kF32Lerp) {
      ImageArgs t = img;
      t.tile_rows = 64;
      bool taken = false;
      const hipError_t e = launch_color(t, map, 1, kF32, sampler, opts, stream, &taken);
      if (e != hipSuccess || taken) return e;
    }
    if (pair && round_f32 && !opts.coef_lds && img.lds_gather && img.wg_box && opts.tall_tiles == 2 && map.tile_dev_ok >= 2 && opts.xcd_remap != 1 &&
        sampler != kF32Lerp) {        // A/B: the generic interleaved-pixel kernel with one channel on 128 x 16 tiles in remap_wg_kernel's place
      ImageArgs t = img;
      t.tile_rows = 128;
      bool taken = false;
      const hipError_t e = launch_color(t, map, 1, kF32, sampler, opts, stream, &taken);
      if (e != hipSuccess || taken) return e;
    }
    if (pair && round_f32 && !opts.coef_lds) {
      switch (nf) {
        case 1: return launch_fast<kRadial, 1>(img, map, sampler, stream);
        case 2: return launch_fast<kRadial, 2>(img, map, sampler, stream);
        case 3: return launch_fast<kRadial, 3>(img, map, sampler, stream);
        case 4: return launch_fast<kRadial, 4>(img, map, sampler, stream);
        case 5: return launch_fast<kRadial, 5>(img, map, sampler, stream);
        case 6: return launch_fast<kRadial, 6>(img, map, sampler, stream);
        case 7: return launch_fast<kRadial, 7>(img, map, sampler, stream);
        case 8: return launch_fast<kRadial, 8>(img, map, sampler, stream);
        case 9: return launch_fast<kRadial, 9>(img, map, sampler, stream);
        case 10: return launch_fast<kRadial, 10>(img, map, sampler, stream);
        default: break;
      }
    }
    if (lds) return launch_lds_any<kRadial, -1>(img, map, sampler, stream);
    if (pair) {
      if (round_f32) return launch_generic<kRadial, true, true>(img, map, sampler, stream);
      return launch_generic<kRadial, false, true>(img, map, sampler, stream);
    }
    if (round_f32) return launch_generic<kRadial, true, false>(img, map, sampler, stream);
    return launch_generic<kRadial, false, false>(img, map, sampler, stream);
  }
  // fused
  if (pair && !opts.coef_lds) {
    switch (nf) {
      case 4: return launch_fast<kFused, 4>(img, map, sampler, stream);
      case 5: return launch_fast<kFused, 5>(img, map, sampler, stream);
      default: break;
    }
  }
  if (lds) return launch_lds_any<kFused, -1>(img, map, sampler, stream);
  if (pair) return launch_generic<kFused, true, true>(img, map, sampler, stream);
  return launch_generic<kFused, true, false>(img, map, sampler, stream);
}

template <int SAMPLER, bool PAIR>
static hipError_t launch_coords_t(const ImageArgs& img, const CoordArgs& ca, hipStream_t stream) {
  const int64_t nb = (ca.npts + kBlock - 1) / kBlock;
  if (nb == 0) return hipSuccess;
  if (ca.is_f64)
    hipLaunchKernelGGL((remap_coords_kernel<SAMPLER, PAIR, double>), dim3((unsigned)nb), dim3(kBlock), 0, stream,
                       img, ca);
  else
    hipLaunchKernelGGL((remap_coords_kernel<SAMPLER, PAIR, float>), dim3((unsigned)nb), dim3(kBlock), 0, stream,
                       img, ca);
  return hipGetLastError();
}

hipError_t launch_coords(const ImageArgs& img, const CoordArgs& ca, int sampler, hipStream_t stream) {
  const bool pair = img.src_col_stride == 1 && img.W >= 2 && img.H >= 2 && img.src_stride < (1 << 22) && img.H < (1 << 24);   // see launch_image
#define DCP_COORDS(S)                                                   \
  case S:                                                               \
    return pair ? launch_coords_t<S, true>(img, ca, stream) : launch_coords_t<S, false>(img, ca, stream);
  switch (sampler) {
    DCP_COORDS(kNearest)
    DCP_COORDS(kScipy)
    DCP_COORDS(kF64Lerp)
    default:
      return pair ? launch_coords_t<kF32Lerp, true>(img, ca, stream)
                  : launch_coords_t<kF32Lerp, false>(img, ca, stream);
  }
#undef DCP_COORDS
}

hipError_t launch_coord_map(MapKind kind, const ImageArgs& img_in, const MapArgs& map, float* ymap, float* xmap,
                            hipStream_t stream) {
  ImageArgs img = img_in;
  img.tile_rows = 16;
  img.tiles_x = (img.W + kBlock - 1) / kBlock;
  img.tiles_y = (img.H + img.tile_rows - 1) / img.tile_rows;
  const dim3 grid(img.tiles_x * img.tiles_y);
  switch (kind) {
    case kRadial: hipLaunchKernelGGL((coord_map_kernel<kRadial>), grid, dim3(kBlock), 0, stream, img, map, ymap, xmap); break;
    case kPersp: hipLaunchKernelGGL((coord_map_kernel<kPersp>), grid, dim3(kBlock), 0, stream, img, map, ymap, xmap); break;
    default: hipLaunchKernelGGL((coord_map_kernel<kFused>), grid, dim3(kBlock), 0, stream, img, map, ymap, xmap); break;
  }
  return hipGetLastError();
}

template <int NF, bool ROUND32>
static hipError_t launch_stack_t(const StackArgs& st, const MapArgs& map, int sampler, hipStream_t stream) {
  note_kernel("stack_rows_kernel", -1, NF, sampler, ROUND32 ? "" : ",f64coords");
  const dim3 grid((st.W + kBlock - 1) / kBlock, st.nrows, (st.D + st.d_chunk - 1) / st.d_chunk);
  switch (sampler) {
    case kScipy:
      hipLaunchKernelGGL((stack_rows_kernel<NF, kScipy, ROUND32>), grid, dim3(kBlock), 0, stream, st, map);
      break;
    case kF64Lerp:
      hipLaunchKernelGGL((stack_rows_kernel<NF, kF64Lerp, ROUND32>), grid, dim3(kBlock), 0, stream, st, map);
      break;
    default:
      hipLaunchKernelGGL((stack_rows_kernel<NF, kF32Lerp, ROUND32>), grid, dim3(kBlock), 0, stream, st, map);
      break;
  }
  return hipGetLastError();
}

template <int NF, bool ROUND32>
static hipError_t launch_centres_t(const StackArgs& st, const MapArgs& map, const double* xcs, const double* ycs, int ncentres, int sampler,
                                   hipStream_t stream) {
  static_assert(sizeof(StackArgs) + sizeof(MapArgs) + sizeof(CentreTable) + 16 <= 4096, "kernel arguments are limited to 4 KB");
  note_kernel("stack_centres_kernel", -1, NF, sampler, ROUND32 ? "" : ",f64coords");
  // blockIdx.y = centre * nrows + row stays below 65536
  const int per_launch = st.nrows >= 65535 ? 1 : (65535 / st.nrows < CentreTable::kMax ? 65535 / st.nrows : CentreTable::kMax);
  for (int k0 = 0; k0 < ncentres; k0 += per_launch) {
    const int m = ncentres - k0 < per_launch ? ncentres - k0 : per_launch;
    CentreTable tab;
    memset(&tab, 0, sizeof(tab));
    for (int i = 0; i < m; ++i) {
      tab.xc[i] = xcs[k0 + i];
      tab.yc[i] = ycs[k0 + i];
    }
    const dim3 grid((st.W + kBlock - 1) / kBlock, (unsigned)(m * st.nrows), (st.D + st.d_chunk - 1) / st.d_chunk);
    switch (sampler) {
      case kScipy: hipLaunchKernelGGL((stack_centres_kernel<NF, kScipy, ROUND32>), grid, dim3(kBlock), 0, stream, st, map, tab, k0); break;
      case kF64Lerp: hipLaunchKernelGGL((stack_centres_kernel<NF, kF64Lerp, ROUND32>), grid, dim3(kBlock), 0, stream, st, map, tab, k0); break;
      default: hipLaunchKernelGGL((stack_centres_kernel<NF, kF32Lerp, ROUND32>), grid, dim3(kBlock), 0, stream, st, map, tab, k0); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_stack_centres(const StackArgs& st_in, const MapArgs& map, const double* xcs, const double* ycs, int ncentres, int sampler,
                                bool round_f32, const LaunchOpts& opts, hipStream_t stream) {
  StackArgs st = st_in;
  if (st.D == 0 || st.nrows == 0 || ncentres == 0) return hipSuccess;
  // one thread keeps its coordinate for d_chunk projections; with many centres the launch is large whatever the chunk
  st.d_chunk = opts.d_chunk < 1 ? 1 : opts.d_chunk;
  while (st.d_chunk < 64 && (int64_t)((st.W + kBlock - 1) / kBlock) * st.nrows * ncentres * ((st.D + 2 * st.d_chunk - 1) / (2 * st.d_chunk)) >= 4096)
    st.d_chunk *= 2;
  if ((st.D + st.d_chunk - 1) / st.d_chunk > 65535) st.d_chunk = (st.D + 65534) / 65535;
  if (!opts.coef_lds) {
    if (map.nfact == 5)
      return round_f32 ? launch_centres_t<5, true>(st, map, xcs, ycs, ncentres, sampler, stream)
                       : launch_centres_t<5, false>(st, map, xcs, ycs, ncentres, sampler, stream);
    if (map.nfact <= 4)
      return round_f32 ? launch_centres_t<4, true>(st, pad4(map), xcs, ycs, ncentres, sampler, stream)
                       : launch_centres_t<4, false>(st, pad4(map), xcs, ycs, ncentres, sampler, stream);
  }
  return round_f32 ? launch_centres_t<-1, true>(st, map, xcs, ycs, ncentres, sampler, stream)
                   : launch_centres_t<-1, false>(st, map, xcs, ycs, ncentres, sampler, stream);
}

DCP_LAB_HOST_DEFINITIONS_UNWARP

DCP_DEFINE_BOUNDS_READER(read_bounds_unwarp)

hipError_t read_lds_stats(unsigned long long* out, bool reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lds_stats), sizeof(g_lds_stats));
  if (e != hipSuccess || !reset) return e;
  const unsigned long long zero[2] = {0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_lds_stats), zero, sizeof(zero));
}

template <int NF>
static hipError_t launch_stack_lds(const StackArgs& st, const MapArgs& map, int sampler, hipStream_t stream) {
  note_kernel("stack_lds_kernel", -1, NF, sampler);
  const dim3 grid((unsigned)((st.W + kLdsTW - 1) / kLdsTW), (unsigned)((st.nrows + kLdsBW * kLdsTH - 1) / (kLdsBW * kLdsTH)),
                  (unsigned)((st.D + st.d_chunk - 1) / st.d_chunk));
  const dim3 block(64 * kLdsBW);
  switch (sampler) {
    case kScipy: hipLaunchKernelGGL((stack_lds_kernel<NF, kScipy>), grid, block, 0, stream, st, map); break;
    case kF64Lerp: hipLaunchKernelGGL((stack_lds_kernel<NF, kF64Lerp>), grid, block, 0, stream, st, map); break;
    default: hipLaunchKernelGGL((stack_lds_kernel<NF, kF32Lerp>), grid, block, 0, stream, st, map); break;
  }
  return hipGetLastError();
}

// Depth chunk for stack_wg_kernel: a workgroup streams its projections through two slabs, so it wants several of
// them, and the launch wants a few thousand workgroups (three fit a CU).  0: too little work even at 4 per workgroup.
static int wg_stack_chunk(const StackArgs& st, int d_chunk, bool force = false) {
  const int64_t tiles = (int64_t)((st.W + kWgTW - 1) / kWgTW) * ((st.nrows + kWgTH - 1) / kWgTH);
  int dc = d_chunk < 4 ? 4 : d_chunk;
  auto wgs = [&](int c) { return tiles * ((st.D + c - 1) / c); };
  while (dc > 4 && wgs(dc) < 3072) dc >>= 1;
  if (dc < 4) dc = 4;
  if ((wgs(dc) < 1024 && !force) || (st.D + dc - 1) / dc > 65535) return 0;
  return dc;
}

// stack_wg_kernel's tile order: 0 grid order, 1 XCD runs, 2 XCD tile rows (see the kernel).  Measured (tools/ab_stack_order.py,
// tools/pmc_stack_order.sh; cfg4 shards of 64 / 256 / 1024 projections, 2560^2): the L2s fetch 1.50 x the algorithmic reads of a
// float32 shard in grid order, 1.10 x by tile rows (what is left is the four rows two vertical neighbours share), 1.07 x by runs;
// uint16: 1.95 / 1.09 / 0.99.  The kernel is not bound by those fetches, so time moves little: float32 by rows 0-1 % faster than grid
// order in every process tried, by runs anywhere between 11 % faster and 9 % slower from process to process (eight XCDs streaming
// eight distant regions); uint16 by runs 3-5 % faster, by rows 2-4 %.  So: float32 by tile rows -- the same speed for a fifth less
// HBM traffic, which the exchange of a sharded stack runs beside -- unless the padding to eight rows costs more than 7 % (then grid
// order); integer element types by runs.  Option xcd_remap: 0 grid order, 1 runs for every type.
static int wg_stack_xcd_order(const StackArgs& st, const LaunchOpts& opts, int es) {
  const int64_t ty = (st.nrows + kWgTH - 1) / kWgTH, ty8 = 8 * ((ty + 7) / 8);
  const int64_t n = (int64_t)((st.W + kWgTW - 1) / kWgTW) * ty8 * ((st.D + st.d_chunk - 1) / st.d_chunk);
  if (opts.xcd_remap == 0 || n >= (int64_t(1) << 31)) return 0;      // (the kernel enumerates the grid in 32 bits)
  if (es < 4 || opts.xcd_remap == 1) return 1;
  return ty8 * 100 <= ty * 107 && ty8 <= 65535 ? 2 : 0;
}

static bool wg_stack_eligible(const StackArgs& st, const MapArgs& map, const LaunchOpts& opts, int es) {
  return map.tile_dev_ok >= 2 && opts.wg_box && opts.lds_gather && opts.stack_lds && !opts.coef_lds && st.nrows >= 8 && st.W >= 2 && st.H >= 2 &&
         st.row_stride < (1 << 22) && st.H < (1 << 24) && (((uintptr_t)st.vol) & 3u) == 0 && (((int64_t)st.row_stride * es) & 3) == 0 &&
         (((int64_t)st.proj_stride * es) & 3) == 0;
}

// would launch_stack() hand this float32 call (float32 coordinates) to stack_wg_kernel?  (dcp_unwarp_images_f32 routes frames of one
// calibration here only when it does: the generic stack kernels are slower than remap_wg_batch_kernel)
bool stack_wg_would_take(const StackArgs& st, const MapArgs& map, const LaunchOpts& opts) {
  if (st.D == 0 || st.nrows == 0 || !opts.stack_wg || !wg_stack_eligible(st, map, opts, 4)) return false;
  const int dc = opts.d_chunk < 1 ? 1 : opts.d_chunk;
  return wg_stack_chunk(st, (dc + 1) / 2, opts.stack_wg >= 2) > 0;
}

template <int NF, typename T>
static hipError_t launch_stack_wg_t(const StackArgs& st, const MapArgs& map, int sampler, hipStream_t stream) {
  dim3 grid((unsigned)((st.W + kWgTW - 1) / kWgTW), (unsigned)((st.nrows + kWgTH - 1) / kWgTH), (unsigned)((st.D + st.d_chunk - 1) / st.d_chunk));
  if (st.xcd_order == 2) grid.y = 8 * ((grid.y + 7) / 8);            // see the kernel's tile order
  // (A/B) workgroups per CU capped through unused dynamic LDS
  unsigned pad = 0;
  if (st.wg_per_cu >= 1 && st.wg_per_cu <= 3) pad = (unsigned)(160 * 1024 / st.wg_per_cu - 56 * 1024) & ~255u;
  if constexpr (std::is_same<T, float>::value) {
    note_kernel("stack_wg_kernel", -1, NF, sampler);
    switch (sampler) {
      case kScipy: hipLaunchKernelGGL((stack_wg_kernel<NF, kScipy, float>), grid, dim3(256), pad, stream, st, map); break;
      case kF64Lerp: hipLaunchKernelGGL((stack_wg_kernel<NF, kF64Lerp, float>), grid, dim3(256), pad, stream, st, map); break;
      default: hipLaunchKernelGGL((stack_wg_kernel<NF, kF32Lerp, float>), grid, dim3(256), pad, stream, st, map); break;
    }
  } else {
    note_kernel("stack_wg_kernel", -1, NF, kScipy, sizeof(T) == 8 ? ",float64" : sizeof(T) == 4 ? ",32-bit" : sizeof(T) == 2 ? ",16-bit" : ",8-bit");
    hipLaunchKernelGGL((stack_wg_kernel<NF, kScipy, T>), grid, dim3(256), pad, stream, st, map);
  }
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_stack_wg_n(const StackArgs& st, const MapArgs& map, int sampler, hipStream_t stream) {
  if (map.nfact == 5) return launch_stack_wg_t<5, T>(st, map, sampler, stream);
  if (map.nfact <= 4) return launch_stack_wg_t<4, T>(st, pad4(map), sampler, stream);
  return launch_stack_wg_t<-1, T>(st, map, sampler, stream);
}

// 8- / 16- / 32-bit integer stacks, float32 coordinates, result of the input's type (unwarp_chunk_slices_backward): st.vol / out
// reinterpreted, strides in elements, proj_bytes the extent of a projection in bytes.  *taken = false: use launch_typed_stack.
hipError_t launch_stack_wg_typed(const StackArgs& st_in, const MapArgs& map, int dtype, const LaunchOpts& opts, hipStream_t stream,
                                 bool* taken) {
  *taken = false;
  if (dtype != kU8 && dtype != kI8 && dtype != kU16 && dtype != kI16 && dtype != kU32 && dtype != kI32 && dtype != kF64) return hipSuccess;
  if (st_in.D == 0 || st_in.nrows == 0 || !wg_stack_eligible(st_in, map, opts, elem_size(dtype))) return hipSuccess;
  StackArgs st = st_in;
  // (32-bit integers move float32's bytes: half the depth chunk as there -- tools/ab_int32_stack.py: 8 projections per workgroup 0-2 %
  // faster than 16 in every process, 32 slower by 2-3 %; untracked against tracked fill: equal to 1 % better)
  st.d_chunk = wg_stack_chunk(st, elem_size(dtype) >= 4 ? (opts.d_chunk + 1) / 2 : opts.d_chunk, opts.stack_wg >= 2);
  if (st.d_chunk == 0) return hipSuccess;
  st.int_exact = opts.int_exact;
  st.xcd_order = wg_stack_xcd_order(st, opts, elem_size(dtype));
  st.store_wait = opts.store_wait;
  st.wg_per_cu = opts.wg_per_cu;
  *taken = true;
  switch (dtype) {
    case kU8: return launch_stack_wg_n<uint8_t>(st, map, kScipy, stream);
    case kI8: return launch_stack_wg_n<int8_t>(st, map, kScipy, stream);
    case kU16: return launch_stack_wg_n<uint16_t>(st, map, kScipy, stream);
    case kI16: return launch_stack_wg_n<int16_t>(st, map, kScipy, stream);
    case kU32: return launch_stack_wg_n<uint32_t>(st, map, kScipy, stream);
    case kF64: return launch_stack_wg_n<double>(st, map, kScipy, stream);
    default: return launch_stack_wg_n<int32_t>(st, map, kScipy, stream);
  }
}

hipError_t launch_stack(const StackArgs& st_in, const MapArgs& map, int sampler, bool round_f32,
                        const LaunchOpts& opts, hipStream_t stream) {
  StackArgs st = st_in;
  st.d_chunk = opts.d_chunk < 1 ? 1 : opts.d_chunk;
  if (st.D == 0 || st.nrows == 0) return hipSuccess;
  // chunks of rows under a certified map: one box per workgroup, two slabs (stack_wg_kernel)
  if (round_f32 && opts.stack_wg && wg_stack_eligible(st, map, opts, 4)) {
    // (float32: half the generic depth chunk -- 8 projections per workgroup at the default.  Measured on cfg4 shards of 64 and 256
    // projections, five processes: 1-4 % faster than 16 in every one, 4 as fast as 8, 32 and 64 slower by 2 and 4 %; the integer
    // instantiations, whose fill streams in under the blend, are fastest at 16)
    const int dc = wg_stack_chunk(st, (st.d_chunk + 1) / 2, opts.stack_wg >= 2);
    if (dc > 0) {
      st.d_chunk = dc;
      st.xcd_order = wg_stack_xcd_order(st, opts, 4);
      st.store_wait = opts.store_wait;
      st.wg_per_cu = opts.wg_per_cu;
      return launch_stack_wg_n<float>(st, map, sampler, stream);
    }
  }
  // chunks of rows (float32 coordinates): the LDS-staged kernel; a few rows only, or the float64-coordinate
  // slice path: the direct gather
  if (opts.lds_gather && opts.stack_lds && round_f32 && (st.nrows >= 8 || opts.stack_lds == 2) && !opts.coef_lds) {
    // a wave walks its projections one after the other, so the launch needs enough wave tiles to fill the
    // chip (256 CUs x 20 waves): shorten the depth chunk until it does; with too little work even at 4
    // projections per wave the finer-grained direct kernel is the faster one (measured on cfg4: 64 rows of
    // a depth-64 shard 21.8 us direct, 39 us staged; all 2560 rows 876 us direct, 740 us staged)
    const int64_t tiles = (int64_t)((st.W + kLdsTW - 1) / kLdsTW) * ((st.nrows + kLdsTH - 1) / kLdsTH);
    int dc = st.d_chunk;
    auto waves = [&](int c) { return tiles * ((st.D + c - 1) / c); };
    while (dc > 4 && waves(dc) < 8192 && opts.stack_lds == 1 && (st.D + (dc >> 1) - 1) / (dc >> 1) <= 65535) dc >>= 1;
    if (waves(dc) >= 4096 || opts.stack_lds == 2) {
      st.d_chunk = dc;
      if (map.nfact == 5) return launch_stack_lds<5>(st, map, sampler, stream);
      if (map.nfact <= 4) return launch_stack_lds<4>(st, pad4(map), sampler, stream);
      return launch_stack_lds<-1>(st, map, sampler, stream);
    }
  }
  if (!opts.coef_lds) {
    if (map.nfact == 5)
      return round_f32 ? launch_stack_t<5, true>(st, map, sampler, stream)
                       : launch_stack_t<5, false>(st, map, sampler, stream);
    if (map.nfact <= 4)
      return round_f32 ? launch_stack_t<4, true>(st, pad4(map), sampler, stream)
                       : launch_stack_t<4, false>(st, pad4(map), sampler, stream);
  }
  return round_f32 ? launch_stack_t<-1, true>(st, map, sampler, stream)
                   : launch_stack_t<-1, false>(st, map, sampler, stream);
}

}  // namespace dcp
