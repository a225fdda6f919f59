// color_kernels.hip -- interleaved (H, W, C) images: util.unwarp_color_image_backward (reference
// discorpy/util/utility.py:278-342).  The reference unwarps a colour image channel by
// channel -- C calls of scipy.ndimage.map_coordinates on the views mat_pad[:, :, i], all at the SAME coordinates
// (utility.py:320-341).  Here a pixel's coordinate is evaluated once and its C channels are blended together.
//
//   remap_wg_color_kernel<NF, SAMPLER, T, NC>   remap_wg_kernel's data path (unwarp_kernels.hip) for pixels of NC
//                                               interleaved elements of type T
//
// A workgroup of four waves owns a 128 x 16 output tile (2 x 2 wave sub-tiles of 64 x 8 pixels).  Its source box --
// the hull of the tile's four corner pixels grown by one pixel, under the host's level-2 tile certificate
// (MapArgs::tile_dev_ok, api_core.cpp) -- is copied into ONE shared slab with row-contiguous 16-byte LDS-DMA loads of
// whole interleaved pixels (a slab row holds up to 144 pixels = 144 NC elements), the loads going out between the rows
// of the coordinate evaluation; after the only barrier every pixel reads its 4 NC taps from LDS, blends its NC channels
// and stores them with one NC-element store, so a wave's store is 64 NC sizeof(T) contiguous bytes.
// Bound: HBM / the L2 -> L1 stream, 2 NC sizeof(T) algorithmic bytes per pixel (24 for float32 RGB); ~80 VALU
// instructions per RGB pixel against ~45 for one plane, so the arithmetic hides under the memory stream more easily
// than in the single-plane kernel.  No MFMA: a remap, not a contraction.
//
// Arithmetic: identical to remap_wg_kernel's per channel -- float64 coordinate, rounded to float32 and clipped
// (utility.py:316-317), then scipy's order-1 blend (SAMPLER = kScipy: bit-equal to the reference), its one-ulp
// factorisation (kF64Lerp, the default of the float32 entry points) or order 0 (kNearest).  Integer element types
// blend in scipy's exact operation order and store as scipy does (to_elem).
#include "dcp_internal.h"
#include "dcp_device.h"
#include <type_traits>
#include <cstdio>

namespace dcp {

#ifndef DCP_COLOR_STORE_AUX
#define DCP_COLOR_STORE_AUX 2   // nt: the result is streamed once
#endif
#ifndef DCP_COLOR_RPW
#define DCP_COLOR_RPW 8         // rows per wave sub-tile: the workgroup tile is 128 x (2 * DCP_COLOR_RPW)
#endif

// SHAPE 0: 128 x 16 output tile, the four waves 2 x 2 (each 64 x 8); box <= 144 pixels x 26 rows.
// SHAPE 1: 64 x 32 output tile, the four waves stacked (each 64 x 8); box <= 80 pixels x 56 rows -- for maps whose tiles are
//          SHEARED (a fisheye model far from the centre: 128 pixels along x climb ~20 source rows, which overflows the
//          height of every 128-wide box): half the width under more than twice the height.
constexpr int kColRPW = DCP_COLOR_RPW;           // rows of a wave sub-tile
template <int SHAPE>
struct TileShape {
  static constexpr int WX = SHAPE == 0 ? 2 : 1, WY = 4 / WX;             // waves along x / y
  static constexpr int TW = 64 * WX, TH = kColRPW * WY;                  // workgroup tile
  static constexpr int BoxPx = SHAPE == 0 ? 144 : 80;                    // widest box in pixels (plus the alignment slack of narrow pixels)
  static constexpr int BoxH = SHAPE == 0 ? TH + TH / 4 + 6 : 56;         // tallest box in rows
};

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2c __attribute__((ext_vector_type(2)));

template <typename T, int NC, int SHAPE = 0>
struct ColorGeom {
  static constexpr int ES = (int)sizeof(T);
  static constexpr int PS = ES * NC;                                  // bytes per pixel
  // the box's first column is rounded down until its byte offset in the row is a multiple of 4 (the 16-byte copies
  // need dword-aligned addresses): a multiple of `kAlignPx` pixels
  static constexpr int kAlignPx = (PS % 4 == 0) ? 1 : ((PS % 2 == 0) ? 2 : 4);
  static constexpr int kBoxWPx = TileShape<SHAPE>::BoxPx + kAlignPx - 1;
  static constexpr int CH = (kBoxWPx * PS + 15) / 16;                 // 16-byte chunks per slab row
  static constexpr int PB = CH * 16;                                  // slab pitch in bytes
  static constexpr int NJ = (TileShape<SHAPE>::BoxH * CH + 255) / 256;   // loads per wave that cover the slab
  static constexpr int kSlabBytes = NJ * 256 * 16;
};

// One blend of the four taps of a channel, in the arithmetic of `SAMPLER` (finish() of unwarp_kernels.hip, restated for
// any element type; EDGE: the far-edge rule of scipy's zero-weight tap, see there).
template <int SAMPLER, typename T, bool EDGE>
__device__ __forceinline__ T blend4(T t00, T t01, T t10, T t11, float fxf, float fyf) {
  if constexpr (SAMPLER == kNearest) {
    return t00;
  } else if constexpr (SAMPLER == kScipy || !std::is_same<T, float>::value) {
    if constexpr (EDGE) {
      if (fxf == 1.0f) {
        t00 = t01;
        t10 = t11;
      }
      if (fyf == 1.0f) {
        t00 = t10;
        t01 = t11;
      }
    }
    const double fx = (double)fxf, fy = (double)fyf;
    const double wy0 = 1.0 - fy, wy1 = 1.0 - wy0;
    const double wx0 = 1.0 - fx, wx1 = 1.0 - wx0;
    double acc = ((double)t00 * wy0) * wx0;
    acc += ((double)t01 * wy0) * wx1;
    acc += ((double)t10 * wy1) * wx0;
    acc += ((double)t11 * wy1) * wx1;
    return to_elem<T>(acc);
  } else if constexpr (SAMPLER == kF64Lerp) {
    const double fx = (double)fxf, fy = (double)fyf;
    const double a = (double)t00, b = (double)t01, c = (double)t10, d = (double)t11;
    const double top = __builtin_fma(fx, b - a, a);
    const double bot = __builtin_fma(fx, d - c, c);
    return (float)__builtin_fma(fy, bot - top, top);
  } else {
    const float top = __builtin_fmaf(fxf, t01 - t00, t00);
    const float bot = __builtin_fmaf(fxf, t11 - t10, t10);
    return __builtin_fmaf(fyf, bot - top, top);
  }
}

// A buffer store of MORE than 64 bits whose data registers are overwritten by the very next VALU instruction stored the NEW
// contents on gfx950 (seen with buffer_store_dwordx3 ... sN offen nt: 500-700 of 4.6e6 values of a frame carried the next row's
// v_cvt_i32_f32 result, differently from run to run).  LLVM's hazard recognizer pads this case only when soffset is NOT a
// register (GCNHazardRecognizer::createsVALUHazard), hipcc therefore left no gap behind the stores with an SGPR row offset.
// The pad keeps the data registers live across one `s_nop 1` (two wait states) behind the store.
#ifndef DCP_WIDE_STORE_PAD_ON
#define DCP_WIDE_STORE_PAD_ON 1
#endif
#if DCP_WIDE_STORE_PAD_ON
#define DCP_WIDE_STORE_PAD(p) asm volatile("s_nop 1" : "+v"(p))
#else
#define DCP_WIDE_STORE_PAD(p) do { } while (0)
#endif

// NC elements of type T to dst + voff (bytes) + soff: one store of the pixel where the hardware has one of that width
template <typename T, int NC>
__device__ __forceinline__ void store_pixel(const T (&v)[NC], __amdgpu_buffer_rsrc_t dst, uint32_t voff, uint32_t soff) {
  constexpr int PS = (int)sizeof(T) * NC;
  if constexpr (std::is_same<T, float>::value && NC == 3) {
    u32x3 p = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2])};
    __builtin_amdgcn_raw_buffer_store_b96(p, dst, voff, soff, DCP_COLOR_STORE_AUX);
    DCP_WIDE_STORE_PAD(p);
  } else if constexpr (std::is_same<T, float>::value && NC == 4) {
    u32x4 p = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(p, dst, voff, soff, DCP_COLOR_STORE_AUX);
    DCP_WIDE_STORE_PAD(p);
  } else if constexpr (std::is_same<T, float>::value && NC == 2) {
    u32x2c p = {__float_as_uint(v[0]), __float_as_uint(v[1])};
    __builtin_amdgcn_raw_buffer_store_b64(p, dst, voff, soff, DCP_COLOR_STORE_AUX);
  } else if constexpr (std::is_same<T, double>::value && NC == 1) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v[0]);
    u32x2c p = {(uint32_t)b, (uint32_t)(b >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(p, dst, voff, soff, DCP_COLOR_STORE_AUX);
  } else if constexpr (std::is_same<T, float>::value && NC == 1) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), dst, voff, soff, DCP_COLOR_STORE_AUX);
  } else if constexpr (sizeof(T) == 4 && NC == 1) {                      // int32 / uint32
    __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v[0], dst, voff, soff, DCP_COLOR_STORE_AUX);
  } else if constexpr (PS == 4) {                                        // 4 x 8-bit, 2 x 16-bit
    uint32_t p = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) p |= ((uint32_t)v[c] & ((1u << (8 * sizeof(T))) - 1u)) << (8 * sizeof(T) * c);
    __builtin_amdgcn_raw_buffer_store_b32(p, dst, voff, soff, DCP_COLOR_STORE_AUX);
  } else if constexpr (PS == 8) {                                        // 4 x 16-bit
    u32x2c p;
    p.x = ((uint32_t)v[0] & 0xffffu) | (((uint32_t)v[1] & 0xffffu) << 16);
    p.y = ((uint32_t)v[2] & 0xffffu) | (((uint32_t)v[3] & 0xffffu) << 16);
    __builtin_amdgcn_raw_buffer_store_b64(p, dst, voff, soff, DCP_COLOR_STORE_AUX);
  } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if constexpr (sizeof(T) == 1)
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v[c], dst, voff + (uint32_t)c, soff, DCP_COLOR_STORE_AUX);
      else if constexpr (sizeof(T) == 2)
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v[c], dst, voff + 2u * (uint32_t)c, soff, DCP_COLOR_STORE_AUX);
      else
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)v[c]), dst, voff + 4u * (uint32_t)c, soff, DCP_COLOR_STORE_AUX);
    }
  }
}

// ImageArgs as for remap_wg_kernel, with src / dst reinterpreted as T*, src_stride = ELEMENTS between source rows, src_col_stride =
// NC (dense pixels), src_bytes the extent in bytes, W / H in PIXELS; the result is dense (W NC elements per row).
template <int NF, int SAMPLER, typename T, int NC, int SHAPE = 0>
__global__ void __launch_bounds__(256, (ColorGeom<T, NC, SHAPE>::kSlabBytes <= 53 * 1024 ? 3 : 2)) remap_wg_color_kernel(const ImageArgs img, const MapArgs map) {
  using G = ColorGeom<T, NC, SHAPE>;
  using S = TileShape<SHAPE>;
  constexpr int kColTW = S::TW, kColTH = S::TH, kColBoxH = S::BoxH;
  constexpr int ES = G::ES, PS = G::PS, CH = G::CH, PB = G::PB, NJ = G::NJ;
  constexpr int RPW = kColRPW;
  __shared__ __attribute__((aligned(16))) unsigned char s_box[G::kSlabBytes];
  __shared__ double s_row[4][RPW][2];
  __shared__ double s_coef[NF < 0 ? kMaxFact : 1];
  const T* const srcT = (const T*)img.src;
  T* const dstT = (T*)img.dst;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int wx = S::WX == 2 ? (wave & 1) : 0, wy = S::WX == 2 ? (wave >> 1) : wave;
  // tile order as remap_wg_kernel: XCD blockIdx.x & 7 owns a stripe of tile columns and sweeps it row by row
  int tx, ty;
  if (img.xcd_remap == 2) {
    const int s = blockIdx.x & 7, c = blockIdx.x >> 3;
    const int wq = img.tiles_x >> 3, wr = img.tiles_x & 7;
    if (c >= wq + (s < wr ? 1 : 0)) return;
    tx = s * wq + min(s, wr) + c;
    ty = blockIdx.y;
  } else {
    tx = blockIdx.x;
    ty = blockIdx.y;
  }
  const int yblk = ty * kColTH;
  const int y0 = __builtin_amdgcn_readfirstlane(yblk + wy * RPW);       // first row of this wave's sub-tile (inside the band)
  const int x = tx * kColTW + wx * 64 + lane;
  const float wmaxf = (float)(img.W - 1), hmaxf = (float)(img.H - 1);

  // ---- the tile's four corner pixels, one per lane 0..3, by every wave for itself (no exchange, no barrier)
  int cx0, cx1, cy0, cy1;
  {
    const double X = (double)min(tx * kColTW + (lane & 1) * (kColTW - 1), img.W - 1);
    const double Y = (double)(img.y_origin + min(yblk + ((lane >> 1) & 1) * (kColTH - 1), img.rows_out - 1));
    double xd, yd;
    corner_coord<kRadial, NF>(map, X, Y, &xd, &yd);
    const int cxi = (int)round_clip_f32(xd, wmaxf), cyi = (int)round_clip_f32(yd, hmaxf);
    const int xa = __builtin_amdgcn_readlane(cxi, 0), xb = __builtin_amdgcn_readlane(cxi, 1);
    const int xc_ = __builtin_amdgcn_readlane(cxi, 2), xd_ = __builtin_amdgcn_readlane(cxi, 3);
    const int ya = __builtin_amdgcn_readlane(cyi, 0), yb = __builtin_amdgcn_readlane(cyi, 1);
    const int yc_ = __builtin_amdgcn_readlane(cyi, 2), yd_ = __builtin_amdgcn_readlane(cyi, 3);
    cx0 = min(min(xa, xb), min(xc_, xd_));
    cx1 = max(max(xa, xb), max(xc_, xd_));
    cy0 = min(min(ya, yb), min(yc_, yd_));
    cy1 = max(max(ya, yb), max(yc_, yd_));
  }
  const int bx0 = max(min(cx0 - 1, img.W - 2), 0) & ~(G::kAlignPx - 1);
  const int bx1 = min(cx1 + 2, img.W - 1);
  const int by0 = max(min(cy0 - 1, img.H - 2), 0);
  const int by1 = min(cy1 + 2, img.H - 1);
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  const bool fits = bw <= G::kBoxWPx && bh <= kColBoxH;          // workgroup-uniform

  // ---- fill: the slab is a linear array of 16-byte chunks, CH per row; chunk (4 j + wave) 64 + lane belongs to lane `lane`
  // of wave `wave` in its j-th load.  Rows past the box are past the fill descriptor's extent: zeros, no memory access.
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const uint32_t rstep = (uint32_t)img.src_stride * (uint32_t)ES;      // source row pitch in bytes
  const unsigned long long rows_end = (unsigned long long)(by1 + 1) * rstep;
  const uint32_t fill_extent = rows_end < (unsigned long long)img.src_bytes ? (uint32_t)rows_end : img.src_bytes;
  const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)img.src, 0, (int)fill_extent, 0x00020000);
  const int fc = wave * 64 + lane;
  const int crow0 = fc / CH;
  const int ccol0 = fc - crow0 * CH;
  const uint32_t off0 = (uint32_t)by0 * rstep + (uint32_t)bx0 * (uint32_t)PS + (uint32_t)crow0 * rstep + (uint32_t)ccol0 * 16u;
  const int nchunk = bh * CH;
  auto issue_fill = [&](const int j) {           // (j is a constant after unrolling)
    if (j < NJ && fits && (j * 4 + wave) * 64 < nchunk) {
      const int qrow = (256 * j) / CH, rem = (256 * j) % CH;
      const bool wrap = ccol0 >= CH - rem;
      const uint32_t step_nowrap = (uint32_t)qrow * rstep + (uint32_t)rem * 16u, step_wrap = step_nowrap + rstep - (uint32_t)PB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lds_ptr)(s_box + (j * 4 + wave) * 1024), 16, off0 + (wrap ? step_wrap : step_nowrap), 0, 0, 0);
    }
  };

  // ---- row table of this wave's rows (same-wave LDS traffic is ordered: no barrier)
  if (lane < RPW) fill_row<kRadial, 2>(map, s_row[wave], lane, (double)(img.y_origin + min(y0 + lane, img.rows_out - 1)));
  if constexpr (NF < 0) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
    __syncthreads();
  }
  const int rows = __builtin_amdgcn_readfirstlane(max(0, min(RPW, img.rows_out - y0)));
  const int ybase = __builtin_amdgcn_readfirstlane(min(y0, img.rows_out - 1));
  const ColCtx col = make_col<kRadial, NF>(map, min(x, img.W - 1));
  const auto* rowtab = s_row[wave];
  const uint32_t row_bytes_out = (uint32_t)img.W * (uint32_t)PS;
  const char* out_base = (const char*)dstT + (size_t)ybase * (size_t)row_bytes_out;
  const uint32_t xoff = (uint32_t)x * (uint32_t)PS;         // lanes with x >= W: guarded below (a row of the result may exceed 4 GiB / rows)
  const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc((void*)out_base, 0, (int)((uint32_t)rows * row_bytes_out), 0x00020000);

  // ---- phase 1: the source coordinates of the sub-tile's rows, the loads of the fill going out between them
  float xf[RPW], yf[RPW];
  constexpr int kPerRow = (NJ + RPW - 1) / RPW;              // loads in front of every coordinate row
  if (rows > 0) {
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
#pragma unroll
      for (int q = 0; q < kPerRow; ++q) issue_fill(k * kPerRow + q);
      double xd, yd;
      map_coord<kRadial, NF, 2, 0>(map, rowtab, s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
      xf[k] = round_clip_f32(xd, wmaxf);
      yf[k] = round_clip_f32(yd, hmaxf);
    }
  } else {
#pragma unroll
    for (int j = 0; j < kPerRow * RPW; ++j) issue_fill(j);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                 // every wave's share of the box has landed

  if (rows == 0 || !(x < img.W)) return;
  if (fits) {
    // ---- phase 2: taps from the shared slab, blend, store
    const uint32_t negorg = (uint32_t)(-(by0 * PB + bx0 * PS));
    const char* boxb = (const char*)s_box;
    const bool interior = bx1 < img.W - 1 && by1 < img.H - 1;
    auto tile_rows_loop = [&](auto inner, auto exact) {
#pragma unroll
      for (int k = 0; k < RPW; ++k) {
        int xi = (int)xf[k], yi = (int)yf[k];
        float fx = 0.0f, fy = 0.0f;
        if constexpr (SAMPLER == kNearest) {
          xi += (xf[k] - (float)xi >= 0.5f) ? 1 : 0;
          yi += (yf[k] - (float)yi >= 0.5f) ? 1 : 0;
        } else {
          if constexpr (!decltype(inner)::value) {
            xi = min(xi, img.W - 2);
            yi = min(yi, img.H - 2);
          }
          fx = xf[k] - (float)xi;
          fy = yf[k] - (float)yi;
        }
        uint32_t addr;
        {
          const uint32_t xa = (uint32_t)xi * (uint32_t)PS + negorg;
          asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(yi), "s"(PB), "v"(xa));
        }
        DCP_BOUNDS(addr, SAMPLER == kNearest ? PS : PB + 2 * PS, G::kSlabBytes, 6);
        const T* t = (const T*)(boxb + addr);
        T v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if constexpr (SAMPLER == kNearest) {
            v[c] = t[c];
          } else if constexpr (decltype(exact)::value) {
            // 8- / 16-bit taps at coordinates >= 32: no operation of the factorised blend rounds, it equals scipy's (dcp_device.h)
            const int ta = (int)t[c], tb = (int)t[NC + c], tc = (int)t[PB / ES + c], td = (int)t[PB / ES + NC + c];
            const double fxd = (double)fx, fyd = (double)fy;
            const double tp = __builtin_fma(fxd, (double)(tb - ta), (double)ta);
            const double bt = __builtin_fma(fxd, (double)(td - tc), (double)tc);
            v[c] = to_elem_in_range<T>(__builtin_fma(fyd, bt - tp, tp));
          } else {
            v[c] = blend4<SAMPLER, T, !decltype(inner)::value>(t[c], t[NC + c], t[PB / ES + c], t[PB / ES + NC + c], fx, fy);
          }
        }
        if (k < rows) store_pixel<T, NC>(v, dst, xoff, (uint32_t)k * row_bytes_out);
      }
    };
    constexpr bool kNarrowInt = std::is_integral<T>::value && sizeof(T) <= 2;
    bool done = false;
    if constexpr (kNarrowInt && SAMPLER != kNearest) {
      if (interior && img.int_exact && bx0 >= (int)kExactLerpMinCoord && by0 >= (int)kExactLerpMinCoord) {
        tile_rows_loop(std::true_type{}, std::true_type{});
        done = true;
      }
    }
    if (!done) {
      if (interior) tile_rows_loop(std::true_type{}, std::false_type{});
      else tile_rows_loop(std::false_type{}, std::false_type{});
    }
  } else {
    // ---- box too large for the slab (magnification above ~1.1): direct global gather, same arithmetic
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      int xi = (int)xf[k], yi = (int)yf[k];
      float fx = 0.0f, fy = 0.0f;
      if constexpr (SAMPLER == kNearest) {
        xi += (xf[k] - (float)xi >= 0.5f) ? 1 : 0;
        yi += (yf[k] - (float)yi >= 0.5f) ? 1 : 0;
      } else {
        xi = min(xi, img.W - 2);
        yi = min(yi, img.H - 2);
        fx = xf[k] - (float)xi;
        fy = yf[k] - (float)yi;
      }
      const T* t = srcT + (size_t)yi * (size_t)img.src_stride + (size_t)xi * NC;
      const size_t rs = (size_t)img.src_stride;
      T v[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if constexpr (SAMPLER == kNearest) v[c] = t[c];
        else v[c] = blend4<SAMPLER, T, true>(t[c], t[NC + c], t[rs + c], t[rs + NC + c], fx, fy);
      }
      if (k < rows) store_pixel<T, NC>(v, dst, xoff, (uint32_t)k * row_bytes_out);
    }
  }
}

// ------------------------------------------------------------------ launchers

DCP_DEFINE_BOUNDS_READER(read_bounds_color)

template <int NF, int SAMPLER, typename T, int NC, int SHAPE = 0>
static hipError_t launch_color_t(const ImageArgs& img_in, const MapArgs& map, hipStream_t stream) {
  using S = TileShape<SHAPE>;
  ImageArgs img = img_in;
  img.tiles_x = (img.W + S::TW - 1) / S::TW;
  img.tiles_y = (img.rows_out + S::TH - 1) / S::TH;
  // XCD stripes of whole tile columns only when they balance (see launch_wg in unwarp_kernels.hip)
  if (img.xcd_remap != 2 || 8 * ((img.tiles_x + 7) / 8) * 100 > img.tiles_x * 107) img.xcd_remap = 0;
  const dim3 grid(img.xcd_remap == 2 ? 8 * ((img.tiles_x + 7) / 8) : img.tiles_x, img.tiles_y);
  char name[96];
  snprintf(name, sizeof(name), "remap_wg_color_kernel<NF=%d,%s,%s x %d%s>", NF, SAMPLER == kNearest ? "nearest" : SAMPLER == kScipy ? "scipy" : "f64lerp",
           std::is_same<T, float>::value ? "float32" : std::is_same<T, double>::value ? "float64" : std::is_same<T, int32_t>::value ? "int32"
           : std::is_same<T, uint32_t>::value ? "uint32" : sizeof(T) == 2 ? "uint16" : "uint8", NC, SHAPE == 1 ? ",64x32 tiles" : "");
  set_last_kernel_name(name);
  hipLaunchKernelGGL((remap_wg_color_kernel<NF, SAMPLER, T, NC, SHAPE>), grid, dim3(256), 0, stream, img, map);
  return hipGetLastError();
}

// coefficient vectors shorter than the instantiated length are padded with zeros: fma(r2, 0, a) = a exactly, so every
// intermediate of the even / odd Horner chains, and the result, is unchanged (as launch_wg_batch_t / pad4)
static MapArgs pad_to(const MapArgs& m, int n) {
  MapArgs p = m;
  for (int i = p.nfact < 0 ? 0 : p.nfact; i < n; ++i) p.fact[i] = 0.0;
  if (p.nfact < n) p.nfact = n;
  return p;
}

template <int SAMPLER, typename T, int NC, int SHAPE = 0>
static hipError_t launch_color_n(const ImageArgs& img, const MapArgs& map, hipStream_t stream) {
  if (map.nfact <= 5) return launch_color_t<5, SAMPLER, T, NC, SHAPE>(img, pad_to(map, 5), stream);
  if (map.nfact <= 10) return launch_color_t<10, SAMPLER, T, NC, SHAPE>(img, pad_to(map, 10), stream);
  return launch_color_t<-1, SAMPLER, T, NC, SHAPE>(img, map, stream);
}

template <typename T, int NC, int SHAPE = 0>
static hipError_t launch_color_s(const ImageArgs& img, const MapArgs& map, int sampler, hipStream_t stream) {
  if (sampler == kNearest) return launch_color_n<kNearest, T, NC, SHAPE>(img, map, stream);
  if constexpr (std::is_same<T, float>::value) {
    if (sampler == kF64Lerp) return launch_color_n<kF64Lerp, T, NC, SHAPE>(img, map, stream);
  }
  return launch_color_n<kScipy, T, NC, SHAPE>(img, map, stream);
}

template <typename T>
static hipError_t launch_color_c(const ImageArgs& img, const MapArgs& map, int channels, int sampler, hipStream_t stream) {
  if (channels == 3) return launch_color_s<T, 3>(img, map, sampler, stream);
  return launch_color_s<T, 4>(img, map, sampler, stream);
}
// single-plane frames of the 4- and 8-byte element types the single-plane kernel (remap_wg_kernel: float32, 8- / 16-bit) does not take
template <typename T>
static hipError_t launch_plane(const ImageArgs& img, const MapArgs& map, int sampler, hipStream_t stream) {
  return launch_color_s<T, 1>(img, map, sampler, stream);
}

// Interleaved pixels of 3 or 4 channels, float32 / uint8 / uint16 -- or ONE channel of float64 / int32 / uint32 (the single-plane
// frames remap_wg_kernel has no instantiation for) --, dense (pixel stride = channels), radial map under the level-2
// certificate, orders 0 / 1.  *taken = false: the call does not qualify and the one-thread-per-pixel kernels must serve it.
hipError_t launch_color(const ImageArgs& img_in, const MapArgs& map, int channels, int dtype, int sampler, const LaunchOpts& opts, hipStream_t stream,
                        bool* taken) {
  *taken = false;
  if (!opts.wg_box || !opts.lds_gather || opts.coef_lds || opts.xcd_remap == 1) return hipSuccess;
  if (!(map.tile_dev_ok >= 2 || (channels == 1 && dtype == kF32 && img_in.tile_rows == 64 && map.tall_ok))) return hipSuccess;
  if (channels == 1 && dtype == kF32 && img_in.tile_rows == 128 && map.tile_dev_ok < 2) return hipSuccess;
  const bool colour = (channels == 3 || channels == 4) && (dtype == kF32 || dtype == kU8 || dtype == kU16);
  // (float32 single planes: only the 64 x 32 tile shape for sheared maps, img.tile_rows = 64 -- launch_plane_tall below; the
  // 128-wide shapes of float32 belong to remap_wg_kernel)
  const bool tall = channels == 1 && dtype == kF32 && img_in.tile_rows == 64;
  const bool flat = channels == 1 && dtype == kF32 && img_in.tile_rows == 128;      // (A/B only: option tall_tiles = 2)
  const bool plane = channels == 1 && (dtype == kF64 || dtype == kI32 || dtype == kU32 || tall || flat);
  if (!colour && !plane) return hipSuccess;
  if (sampler != kNearest && sampler != kScipy && !(sampler == kF64Lerp && dtype == kF32)) return hipSuccess;
  const int es = elem_size(dtype);
  ImageArgs img = img_in;
  if (img.rows_out <= 0) {
    img.y_origin = 0;
    img.rows_out = img.H;
  }
  img.xcd_remap = opts.xcd_remap;
  img.int_exact = opts.int_exact;
  // dense pixels, at least 2 x 2, dword-aligned rows (the 16-byte LDS-DMA copies), 24-bit row products, rows of the result below 4 GiB
  if (img.src_col_stride != channels || img.W < 2 || img.H < 2 || ((uintptr_t)img.src & 3u) || (((int64_t)img.src_stride * es) & 3) ||
      (int64_t)img.src_stride * es >= (1ll << 31) || img.H >= (1 << 24) || (int64_t)img.W * channels * es >= (1ll << 28))
    return hipSuccess;
  *taken = true;
  if (tall) return launch_color_s<float, 1, 1>(img, map, sampler, stream);
  if (flat) return launch_color_s<float, 1, 0>(img, map, sampler, stream);
  switch (dtype) {
    case kF32: return launch_color_c<float>(img, map, channels, sampler, stream);
    case kU8: return launch_color_c<uint8_t>(img, map, channels, sampler, stream);
    case kU16: return launch_color_c<uint16_t>(img, map, channels, sampler, stream);
    case kF64: return launch_plane<double>(img, map, sampler, stream);
    case kI32: return launch_plane<int32_t>(img, map, sampler, stream);
    default: return launch_plane<uint32_t>(img, map, sampler, stream);
  }
}

}  // namespace dcp
