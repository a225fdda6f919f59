// unwarp_kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels for the backward
// unwarp path of discorpy (reference: /root/reference/discorpy/post/postprocessing.py).
//
//   remap_tile_kernel<Radial>   K1  unwarp_image_backward            postprocessing.py:137-148
//   remap_tile_kernel<Persp>    K2  correct_perspective_image        postprocessing.py:448-459,486-492
//   remap_tile_kernel<Fused>    K3  perspective o radial in one pass
//   stack_rows_kernel           K4  unwarp_slice_backward / unwarp_chunk_slices_backward  :211-229,:281-313
//   remap_coords_kernel         K5  map_index= / _mapping            postprocessing.py:250-251,489-491
//
// Design (see DESIGN.md for the numbers):
//  * one thread per output pixel column, 64-wide wavefronts along x, a workgroup of 4 waves
//    walks `tile_rows` rows of a 256-pixel-wide tile: stores are 256 B contiguous per wave
//    instruction, the two source rows a wave gathers from are ~260 B contiguous each.
//  * the coordinate polynomial is evaluated in fp64 (the reference computes float64 and only
//    then rounds to float32, postprocessing.py:144-145; an fp32 evaluation changes 34 % of the
//    coordinates by one ulp).  Everything that does not depend on x is staged once per
//    workgroup in LDS (yu, yu^2 per row; c*y products for the homography); everything that
//    does not depend on y lives in registers across the row loop; coefficients are SGPR
//    resident (kernarg) for vectors of <= 10 terms and staged in LDS for longer ones.
//  * sqrt is the correctly rounded fp64 result built from v_rsq_f64 + one coupled Newton
//    step + one residual correction (validated against the host sqrt in tools/ubench.hip).
//  * the gather uses raw buffer loads: a 32-bit byte offset per lane, the second row through
//    the scalar offset, hardware bounds checking; the two taps of a row are one 8-byte load.
//  * no MFMA: the op is a remap (8 B of HBM traffic per pixel), not a contraction.
//
// Build with -ffp-contract=off: every fused multiply-add below is written explicitly so that
// the arithmetic is the same sequence of IEEE operations as oracle/unwarp_oracle.c.
#include "dcp_internal.h"
#include <type_traits>

namespace dcp {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;

// ------------------------------------------------------------------ fp64 helpers

// Correctly rounded sqrt for finite x >= 0 (no range scaling: r2 is bounded by the image size).
__device__ __forceinline__ double sqrt_rn(double x) {
  double y = __builtin_amdgcn_rsq(x);          // ~2^-24 relative
  double g = x * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);                  // ~2^-47
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);          // exact residual
  g = __builtin_fma(d, h, g);                  // correctly rounded
  return x == 0.0 ? 0.0 : g;
}

// B(ru) = sum a_i ru^i, split into even and odd powers so that only one multiply by ru is
// needed: E = a0 + r2 (a2 + r2 (a4 + ...)), O = a1 + r2 (a3 + ...), B = fma(ru, O, E).
// Same operation order as poly_kernel() in oracle/unwarp_oracle.c.
template <int NF>
__device__ __forceinline__ double poly_inline(const double* __restrict__ a, double r2, double ru) {
  if constexpr (NF == 0) {
    return 0.0;
  } else {
    constexpr int ne = (NF + 1) / 2, no = NF / 2;
    double E = a[2 * (ne - 1)];
#pragma unroll
    for (int k = ne - 2; k >= 0; --k) E = __builtin_fma(r2, E, a[2 * k]);
    if constexpr (no == 0) {
      return E;
    } else {
      double O = a[2 * (no - 1) + 1];
#pragma unroll
      for (int k = no - 2; k >= 0; --k) O = __builtin_fma(r2, O, a[2 * k + 1]);
      return __builtin_fma(ru, O, E);
    }
  }
}

__device__ __forceinline__ double poly_lds(const double* s_coef, int nf, double r2, double ru) {
  if (nf <= 0) return 0.0;
  const int ne = (nf + 1) >> 1, no = nf >> 1;
  double E = s_coef[2 * (ne - 1)];
  for (int k = ne - 2; k >= 0; --k) E = __builtin_fma(r2, E, s_coef[2 * k]);
  if (no == 0) return E;
  double O = s_coef[2 * (no - 1) + 1];
  for (int k = no - 2; k >= 0; --k) O = __builtin_fma(r2, O, s_coef[2 * k + 1]);
  return __builtin_fma(ru, O, E);
}

// ------------------------------------------------------------------ sampler

struct SrcView {
  __amdgpu_buffer_rsrc_t rsrc;
  int32_t W, H;
  int32_t stride;      // elements
  int32_t cstride;     // elements
};

template <typename WT>
struct Pos {
  uint32_t off;        // byte offset of tap (y0, x0)
  uint32_t dx, dy;     // byte distance to the x+1 / y+1 taps (safe path only)
  WT fx, fy;
};

// Coordinates are already inside [0, W-1] x [0, H-1].
// PAIR: the x0/x1 taps are one 8-byte load, so x0 is held <= W-2 (and y0 <= H-2): at the
// far edge the weight pair becomes (0, 1) on (W-2, W-1) instead of scipy's (1, 0) on
// (W-1, reflected W-1) -- the same value for finite data.
template <bool PAIR, typename CT>
__device__ __forceinline__ Pos<CT> locate(const SrcView& s, CT xc, CT yc) {
  Pos<CT> p;
  int xi = (int)xc, yi = (int)yc;   // truncation == floor for non-negative coordinates
  if constexpr (PAIR) {
    xi = min(xi, s.W - 2);
    yi = min(yi, s.H - 2);
    p.dx = 4u;
    p.dy = (uint32_t)s.stride * 4u;
  } else {
    // any stride, any size: same base-tap rule per axis, four 4-byte loads
    xi = min(xi, max(s.W - 2, 0));
    yi = min(yi, max(s.H - 2, 0));
    p.dx = s.W >= 2 ? (uint32_t)s.cstride * 4u : 0u;
    p.dy = s.H >= 2 ? (uint32_t)s.stride * 4u : 0u;
  }
  p.fx = xc - (CT)xi;               // exact
  p.fy = yc - (CT)yi;
  p.off = ((uint32_t)yi * (uint32_t)s.stride + (uint32_t)xi * (uint32_t)s.cstride) * 4u;
  return p;
}

// order 0: index = floor(c + 0.5)  (round half up, not rint)
template <typename CT>
__device__ __forceinline__ uint32_t locate_nearest(const SrcView& s, CT xc, CT yc) {
  int xi = (int)xc, yi = (int)yc;
  xi += (xc - (CT)xi >= (CT)0.5) ? 1 : 0;
  yi += (yc - (CT)yi >= (CT)0.5) ? 1 : 0;
  return ((uint32_t)yi * (uint32_t)s.stride + (uint32_t)xi * (uint32_t)s.cstride) * 4u;
}

struct Taps {
  float v00, v01, v10, v11;
};

template <bool PAIR, typename WT>
__device__ __forceinline__ Taps gather(const __amdgpu_buffer_rsrc_t rsrc, const Pos<WT>& p, int row_bytes) {
  Taps t;
  if constexpr (PAIR) {
    u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, p.off, 0, 0);
    u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, p.off, row_bytes, 0);
    t.v00 = __uint_as_float(a.x);
    t.v01 = __uint_as_float(a.y);
    t.v10 = __uint_as_float(b.x);
    t.v11 = __uint_as_float(b.y);
  } else {
    t.v00 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, p.off, 0, 0));
    t.v01 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, p.off + p.dx, 0, 0));
    t.v10 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, p.off + p.dy, 0, 0));
    t.v11 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, p.off + p.dy + p.dx, 0, 0));
  }
  return t;
}

// The three order-1 arithmetics (same as sample() in oracle/unwarp_oracle.c).
template <int SAMPLER, typename WT>
__device__ __forceinline__ float blend(const Taps& t, WT fx_, WT fy_) {
  if constexpr (SAMPLER == kScipy) {
    // scipy NI_GeometricTransform: w0 = 1 - f, w1 = 1 - w0; ((v*wy)*wx) summed left to right
    double fx = (double)fx_, fy = (double)fy_;
    double wy0 = 1.0 - fy, wy1 = 1.0 - wy0;
    double wx0 = 1.0 - fx, wx1 = 1.0 - wx0;
    double acc = ((double)t.v00 * wy0) * wx0;
    acc += ((double)t.v01 * wy0) * wx1;
    acc += ((double)t.v10 * wy1) * wx0;
    acc += ((double)t.v11 * wy1) * wx1;
    return (float)acc;
  } else if constexpr (SAMPLER == kF64Lerp) {
    double fx = (double)fx_, fy = (double)fy_;
    double a = (double)t.v00, b = (double)t.v01, c = (double)t.v10, d = (double)t.v11;
    double top = __builtin_fma(fx, b - a, a);
    double bot = __builtin_fma(fx, d - c, c);
    return (float)__builtin_fma(fy, bot - top, top);
  } else {
    float fx = (float)fx_, fy = (float)fy_;
    float top = __builtin_fmaf(fx, t.v01 - t.v00, t.v00);
    float bot = __builtin_fmaf(fx, t.v11 - t.v10, t.v10);
    return __builtin_fmaf(fy, bot - top, top);
  }
}

template <int SAMPLER, bool PAIR, typename CT>
__device__ __forceinline__ float sample(const SrcView& s, CT xc, CT yc) {
  if constexpr (SAMPLER == kNearest) {
    uint32_t off = locate_nearest<CT>(s, xc, yc);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(s.rsrc, off, 0, 0));
  } else {
    Pos<CT> p = locate<PAIR, CT>(s, xc, yc);
    Taps t = gather<PAIR, CT>(s.rsrc, p, s.stride * 4);
    return blend<SAMPLER, CT>(t, p.fx, p.fy);
  }
}

// np.float32(np.clip(v, 0, len-1)) == clip(float32(v), 0, len-1): rounding is monotone and
// both bounds are float32 numbers, so the clip is done after the conversion, in fp32.
__device__ __forceinline__ float round_clip_f32(double v, float hi) {
  return __builtin_amdgcn_fmed3f((float)v, 0.0f, hi);
}
__device__ __forceinline__ double clip_f64(double v, double hi) {
  v = v < 0.0 ? 0.0 : v;
  return v > hi ? hi : v;
}

// ------------------------------------------------------------------ tile bookkeeping

// blockIdx.x -> tile.  The dispatcher places workgroup b on XCD b % 8; with xcd_remap each XCD
// gets one contiguous band of tiles, so the one-row halo shared by vertically adjacent tiles is
// re-read from that XCD's own L2 instead of from another die's (speed only, never correctness).
__device__ __forceinline__ int logical_tile(int xcd_remap) {
  int b = blockIdx.x;
  if (!xcd_remap) return b;
  int nb = gridDim.x;
  int per = nb >> 3, rem = nb & 7;
  int xcd = b & 7, j = b >> 3;
  return xcd * per + min(xcd, rem) + j;
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

// NF >= 0: polynomial length known at compile time, coefficients read from kernarg (SGPRs).
// NF == -1: runtime length, coefficients staged in LDS.
template <int KIND, int NF, int SAMPLER, bool ROUND32, bool PAIR>
__global__ void __launch_bounds__(kBlock) remap_tile_kernel(const ImageArgs img, const MapArgs map) {
  __shared__ double s_row[kMaxTileRows][4];
  __shared__ double s_coef[kMaxFact];

  const int tile = logical_tile(img.xcd_remap);
  const int ty = tile / img.tiles_x;
  const int tx = tile - ty * img.tiles_x;
  const int y0 = ty * img.tile_rows;
  const int x = tx * kBlock + (int)threadIdx.x;
  const int rows = min(img.tile_rows, img.H - y0);

  // per-row invariants -> LDS
  if ((int)threadIdx.x < rows) {
    const double y = (double)(y0 + (int)threadIdx.x);
    if constexpr (KIND == kRadial) {
      const double yu = y - map.yc;
      s_row[threadIdx.x][0] = yu;
      s_row[threadIdx.x][1] = yu * yu;
    } else {
      s_row[threadIdx.x][0] = map.coef[7] * y;   // c8*y
      s_row[threadIdx.x][1] = map.coef[1] * y;   // c2*y
      s_row[threadIdx.x][2] = map.coef[4] * y;   // c5*y
    }
  }
  if constexpr (NF < 0 && KIND != kPersp) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
  }
  __syncthreads();
  if (x >= img.W) return;

  const SrcView src = make_view(img.src, img.src_bytes, img.W, img.H, img.src_stride, img.src_col_stride);
  const float wmaxf = (float)(img.W - 1), hmaxf = (float)(img.H - 1);
  const double wmaxd = (double)(img.W - 1), hmaxd = (double)(img.H - 1);
  float* __restrict__ out = img.dst + (size_t)y0 * (size_t)img.W + (size_t)x;

  // per-column invariants -> registers
  const double xd_ = (double)x;
  double cx0, cx1, cx2;
  if constexpr (KIND == kRadial) {
    cx0 = xd_ - map.xc;        // xu
    cx1 = cx0 * cx0;           // xu^2
    cx2 = 0.0;
  } else {
    cx0 = map.coef[6] * xd_;   // c7*x
    cx1 = map.coef[0] * xd_;   // c1*x
    cx2 = map.coef[3] * xd_;   // c4*x
  }

#pragma unroll 2
  for (int k = 0; k < rows; ++k) {
    double xd, yd;             // source coordinate, float64, not yet clipped
    if constexpr (KIND == kRadial) {
      const double yu = s_row[k][0];
      const double r2 = cx1 + s_row[k][1];
      const double ru = sqrt_rn(r2);
      double f;
      if constexpr (NF >= 0) f = poly_inline<NF>(map.fact, r2, ru);
      else f = poly_lds(s_coef, map.nfact, r2, ru);
      const double px = f * cx0;
      const double py = f * yu;
      xd = map.xc + px;
      yd = map.yc + py;
    } else {
      // postprocessing.py:453-455, numpy order: (c7*x + c8*y) + 1.0 etc., true divisions
      const double den = (cx0 + s_row[k][0]) + 1.0;
      const double nx = (cx1 + s_row[k][1]) + map.coef[2];
      const double ny = (cx2 + s_row[k][2]) + map.coef[5];
      xd = nx / den;
      yd = ny / den;
      if constexpr (KIND == kFused) {
        // float32-rounded perspective coordinate, then the radial map evaluated there
        const double xp = (double)round_clip_f32(xd, wmaxf);
        const double yp = (double)round_clip_f32(yd, hmaxf);
        const double xu = xp - map.xc;
        const double yu = yp - map.yc;
        const double xx = xu * xu;
        const double yy = yu * yu;
        const double r2 = xx + yy;
        const double ru = sqrt_rn(r2);
        double f;
        if constexpr (NF >= 0) f = poly_inline<NF>(map.fact, r2, ru);
        else f = poly_lds(s_coef, map.nfact, r2, ru);
        const double px = f * xu;
        const double py = f * yu;
        xd = map.xc + px;
        yd = map.yc + py;
      }
    }
    float v;
    if constexpr (ROUND32) {
      v = sample<SAMPLER, PAIR, float>(src, round_clip_f32(xd, wmaxf), round_clip_f32(yd, hmaxf));
    } else {
      v = sample<SAMPLER, PAIR, double>(src, clip_f64(xd, wmaxd), clip_f64(yd, hmaxd));
    }
    out[(size_t)k * (size_t)img.W] = v;
  }
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
  xc = xc < (CT)0 ? (CT)0 : xc;
  xc = xc > wmax ? wmax : xc;
  yc = yc < (CT)0 ? (CT)0 : yc;
  yc = yc > hmax ? hmax : yc;
  img.dst[i] = sample<SAMPLER, PAIR, CT>(src, xc, yc);
}

// ------------------------------------------------------------------ K4: rows of a (D,H,W) stack

// out[d, r, x] = projection d sampled at the radial source coordinate of (row_start + r, x).
// The coordinate and the tap weights are computed once per (r, x) and reused for d_chunk
// projections; the per-projection work is two 8-byte gathers, the blend and a 4-byte store.
template <int NF, int SAMPLER, bool ROUND32>
__global__ void __launch_bounds__(kBlock) stack_rows_kernel(const StackArgs st, const MapArgs map) {
  __shared__ double s_coef[kMaxFact];
  if constexpr (NF < 0) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
    __syncthreads();
  }
  const int x = blockIdx.x * kBlock + (int)threadIdx.x;
  const int r = blockIdx.y;
  const int d0 = blockIdx.z * st.d_chunk;
  const int d1 = min(st.D, d0 + st.d_chunk);
  if (x >= st.W) return;

  const double xu = (double)x - map.xc;
  const double yu = (st.row_start + (double)r) - map.yc;
  const double r2 = xu * xu + yu * yu;
  const double ru = sqrt_rn(r2);
  double f;
  if constexpr (NF >= 0) f = poly_inline<NF>(map.fact, r2, ru);
  else f = poly_lds(s_coef, map.nfact, r2, ru);
  const double px = f * xu;
  const double py = f * yu;
  const double xd = map.xc + px;
  const double yd = map.yc + py;

  SrcView s;
  s.W = st.W;
  s.H = st.H;
  s.stride = st.row_stride;
  s.cstride = 1;
  using CT = typename std::conditional<ROUND32, float, double>::type;
  CT xc, yc;
  if constexpr (ROUND32) {
    xc = round_clip_f32(xd, (float)(st.W - 1));
    yc = round_clip_f32(yd, (float)(st.H - 1));
  } else {
    xc = clip_f64(xd, (double)(st.W - 1));
    yc = clip_f64(yd, (double)(st.H - 1));
  }
  const Pos<CT> p = locate<true, CT>(s, xc, yc);
  const int row_bytes = st.row_stride * 4;
  const float* base = st.vol + (size_t)d0 * (size_t)st.proj_stride;
  float* out = st.out + ((size_t)d0 * (size_t)st.nrows + (size_t)r) * (size_t)st.W + (size_t)x;
  const size_t out_step = (size_t)st.nrows * (size_t)st.W;
#pragma unroll 4
  for (int d = d0; d < d1; ++d) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)st.proj_bytes, 0x00020000);
    const Taps t = gather<true, CT>(rsrc, p, row_bytes);
    *out = blend<SAMPLER, CT>(t, p.fx, p.fy);
    base += st.proj_stride;
    out += out_step;
  }
}

// ------------------------------------------------------------------ launchers

template <int KIND, int NF, int SAMPLER, bool ROUND32, bool PAIR>
static hipError_t launch_one(const ImageArgs& img, const MapArgs& map, hipStream_t stream) {
  const int nb = img.tiles_x * img.tiles_y;
  hipLaunchKernelGGL((remap_tile_kernel<KIND, NF, SAMPLER, ROUND32, PAIR>), dim3(nb), dim3(kBlock), 0, stream,
                     img, map);
  return hipGetLastError();
}

template <int KIND, int NF>
static hipError_t launch_fast(const ImageArgs& img, const MapArgs& map, int sampler, hipStream_t stream) {
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
  int tr = opts.tile_rows;
  if (tr < 1) tr = 1;
  if (tr > kMaxTileRows) tr = kMaxTileRows;
  img.tile_rows = tr;
  img.tiles_x = (img.W + kBlock - 1) / kBlock;
  img.tiles_y = (img.H + tr - 1) / tr;
  img.xcd_remap = opts.xcd_remap;
  // the 8-byte pair gather needs unit column stride and at least a 2x2 image
  const bool pair = img.src_col_stride == 1 && img.W >= 2 && img.H >= 2;
  const int nf = map.nfact;

  if (kind == kPersp) {
    if (pair) return launch_generic<kPersp, true, true>(img, map, sampler, stream);
    return launch_generic<kPersp, true, false>(img, map, sampler, stream);
  }
  if (kind == kRadial) {
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
  const bool pair = img.src_col_stride == 1 && img.W >= 2 && img.H >= 2;
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

template <int NF, bool ROUND32>
static hipError_t launch_stack_t(const StackArgs& st, const MapArgs& map, int sampler, hipStream_t stream) {
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

hipError_t launch_stack(const StackArgs& st_in, const MapArgs& map, int sampler, bool round_f32,
                        const LaunchOpts& opts, hipStream_t stream) {
  StackArgs st = st_in;
  st.d_chunk = opts.d_chunk < 1 ? 1 : opts.d_chunk;
  if (st.D == 0 || st.nrows == 0) return hipSuccess;
  if (!opts.coef_lds) {
    if (map.nfact == 5)
      return round_f32 ? launch_stack_t<5, true>(st, map, sampler, stream)
                       : launch_stack_t<5, false>(st, map, sampler, stream);
    if (map.nfact == 4)
      return round_f32 ? launch_stack_t<4, true>(st, map, sampler, stream)
                       : launch_stack_t<4, false>(st, map, sampler, stream);
  }
  return round_f32 ? launch_stack_t<-1, true>(st, map, sampler, stream)
                   : launch_stack_t<-1, false>(st, map, sampler, stream);
}

}  // namespace dcp
