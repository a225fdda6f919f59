// typed_kernels.hip -- spline orders 0 / 1 of the unwarp path for element types other than float32
// ("output dtype = input dtype"; section 8(f2)).  The reference hands
// whatever array it is given to scipy.ndimage.map_coordinates (discorpy/post/postprocessing.py:147,
// 227, 251, 491), which reads every element as a double, blends in double in its fixed order
//     t = 0; t += (v00*wy0)*wx0; t += (v01*wy0)*wx1; t += (v10*wy1)*wx0; t += (v11*wy1)*wx1
// and converts t to the input's type: C cast for floats, round-half-away-from-zero + saturation for
// integers.  These kernels do exactly that (bit-equal to oracle/unwarp_oracle.c's
// orc_map_coordinates_typed, which golden G12 pins to the reference):
//
//   typed_image_kernel<T, KIND>   one thread per output pixel; KIND radial / perspective / fused map,
//                                 or 3 = caller-supplied coordinates (float32 or float64)
//   typed_stack_kernel<T>         rows of a (D, H, W) stack: the coordinate of (row, x) once, then
//                                 d_chunk projections; unwarp_slice_backward keeps float64
//                                 coordinates and stores float32, unwarp_chunk_slices_backward rounds
//                                 the coordinates to float32 and stores T
//
// Plain 64-bit addressing (no 4 GiB limit), no LDS staging: float32 is the hot path and has its own
// kernels (unwarp_kernels.hip); these are HBM-bound gathers with 1..8-byte taps.
#include "dcp_internal.h"
#include "dcp_device.h"

namespace dcp {

constexpr int kTypedBlock = 256;

// t of map_coordinates at the clamped coordinate (yc, xc), order 0 or 1, taps of type T
template <typename T>
__device__ __forceinline__ double sample_typed(const T* __restrict__ src, int64_t rs, int64_t cs, int H, int W, double yc,
                                               double xc, int order) {
  if (order == 0) {
    int iy = (int)__builtin_floor(yc + 0.5), ix = (int)__builtin_floor(xc + 0.5);
    iy = min(max(iy, 0), H - 1);
    ix = min(max(ix, 0), W - 1);
    return (double)src[(int64_t)iy * rs + (int64_t)ix * cs];
  }
  const double y0 = __builtin_floor(yc), x0 = __builtin_floor(xc);
  const double wy0 = 1.0 - (yc - y0), wy1 = 1.0 - wy0;
  const double wx0 = 1.0 - (xc - x0), wx1 = 1.0 - wx0;
  // coordinates are inside [0, len-1]; the neighbour of len-1 folds back onto it and weighs 0
  const int iy0 = min((int)y0, H - 1), ix0 = min((int)x0, W - 1);
  const int iy1 = min(iy0 + 1, H - 1), ix1 = min(ix0 + 1, W - 1);
  const T* r0 = src + (int64_t)iy0 * rs;
  const T* r1 = src + (int64_t)iy1 * rs;
  const double v00 = (double)r0[(int64_t)ix0 * cs], v01 = (double)r0[(int64_t)ix1 * cs];
  const double v10 = (double)r1[(int64_t)ix0 * cs], v11 = (double)r1[(int64_t)ix1 * cs];
  double t = 0.0;
  t += (v00 * wy0) * wx0;
  t += (v01 * wy0) * wx1;
  t += (v10 * wy1) * wx0;
  t += (v11 * wy1) * wx1;
  return t;
}

template <typename T, int KIND>
__global__ void __launch_bounds__(kTypedBlock) typed_image_kernel(const TypedImageArgs a, const MapArgs map,
                                                                 const CoordArgs ca) {
  // explicit coordinates: a 1-D launch over the points; maps: blockIdx.y walks the rows (no 64-bit division)
  int64_t i;
  int x = 0, y = 0;
  if constexpr (KIND == 3) {
    i = (int64_t)blockIdx.x * kTypedBlock + threadIdx.x;
    if (i >= ca.npts) return;
  } else {
    x = blockIdx.x * kTypedBlock + (int)threadIdx.x;
    y = blockIdx.y + blockIdx.z * 65535;
    if (x >= a.W || y >= a.H) return;
    i = (int64_t)y * a.W + x;
  }
  double yc, xc;
  if constexpr (KIND == 3) {
    if (ca.is_f64) {
      yc = ((const double*)ca.ycoord)[i];
      xc = ((const double*)ca.xcoord)[i];
    } else {
      yc = (double)((const float*)ca.ycoord)[i];
      xc = (double)((const float*)ca.xcoord)[i];
    }
    if (ca.mode != kModeNearest && (yc < 0.0 || yc > (double)(a.H - 1) || xc < 0.0 || xc > (double)(a.W - 1))) {
      const T* base = (const T*)a.src;
      const int64_t rs = a.src_stride, cs = a.src_cstride;
      ((T*)a.dst)[i] = to_elem<T>(mc_sample_outside([&](long long r, long long c) -> double { return (double)base[r * rs + c * cs]; }, a.H,
                                                    a.W, yc, xc, a.order, ca.mode));
      return;
    }
    yc = clip_f64(yc, (double)(a.H - 1));
    xc = clip_f64(xc, (double)(a.W - 1));
  } else {
    const float wmaxf = (float)(a.W - 1), hmaxf = (float)(a.H - 1);
    double xd, yd;
    pixel_coord<KIND>(map, (double)x, (double)y, wmaxf, hmaxf, &xd, &yd);
    xc = (double)round_clip_f32(xd, wmaxf);     // np.float32(np.clip(...)), postprocessing.py:144-145, 456-457
    yc = (double)round_clip_f32(yd, hmaxf);
  }
  const double t = sample_typed<T>((const T*)a.src, a.src_stride, a.src_cstride, a.H, a.W, yc, xc, a.order);
  ((T*)a.dst)[i] = to_elem<T>(t);
}

// Interleaved (H, W, C) image, every channel sampled at the SAME coordinate -- what
// util.unwarp_color_image_backward does channel by channel (discorpy/util/utility.py:327-341): one
// coordinate evaluation and C blends per pixel, taps of a pixel's C channels contiguous in memory.
template <typename T>
__global__ void __launch_bounds__(kTypedBlock) typed_channels_kernel(const TypedImageArgs a, const MapArgs map, int C) {
  const int x = blockIdx.x * kTypedBlock + (int)threadIdx.x;
  const int yl = blockIdx.y + blockIdx.z * 65535;          // row inside the band [y0, y0 + rows)
  if (x >= a.W || yl >= a.rows) return;
  const int y = a.y0 + yl;
  const int64_t i = (int64_t)yl * a.W + x;
  const float wmaxf = (float)(a.W - 1), hmaxf = (float)(a.H - 1);
  double xd, yd;
  pixel_coord<kRadial>(map, (double)x, (double)y, wmaxf, hmaxf, &xd, &yd);
  const double xc = (double)round_clip_f32(xd, wmaxf), yc = (double)round_clip_f32(yd, hmaxf);
  const T* src = (const T*)a.src;
  T* dst = (T*)a.dst + i * C;
  const int64_t rs = a.src_stride, cs = a.src_cstride;
  if (a.order == 0) {
    int iy = (int)__builtin_floor(yc + 0.5), ix = (int)__builtin_floor(xc + 0.5);
    iy = min(max(iy, 0), a.H - 1);
    ix = min(max(ix, 0), a.W - 1);
    const T* p = src + (int64_t)iy * rs + (int64_t)ix * cs;
    for (int c = 0; c < C; ++c) dst[c] = to_elem<T>((double)p[c]);
    return;
  }
  if constexpr (std::is_same<T, float>::value) {
    if (a.blend == kF64Lerp && a.W >= 2 && a.H >= 2) {
      // the one-ulp factorisation with the staged kernels' edge rule (base tap held at len - 2, fraction 1 at the far edge):
      // bit-equal to remap_wg_color_kernel / remap_wg_kernel under the same blend
      const float xcf = (float)xc, ycf = (float)yc;
      const int xi = min((int)xcf, a.W - 2), yi = min((int)ycf, a.H - 2);
      const double fx = (double)(xcf - (float)xi), fy = (double)(ycf - (float)yi);
      const T* p = src + (int64_t)yi * rs + (int64_t)xi * cs;
      for (int c = 0; c < C; ++c) {
        const double ta = (double)p[c], tb = (double)p[cs + c], tc = (double)p[rs + c], td = (double)p[rs + cs + c];
        const double top = __builtin_fma(fx, tb - ta, ta);
        const double bot = __builtin_fma(fx, td - tc, tc);
        dst[c] = (float)__builtin_fma(fy, bot - top, top);
      }
      return;
    }
  }
  const double y0 = __builtin_floor(yc), x0 = __builtin_floor(xc);
  const double wy0 = 1.0 - (yc - y0), wy1 = 1.0 - wy0;
  const double wx0 = 1.0 - (xc - x0), wx1 = 1.0 - wx0;
  const int iy0 = min((int)y0, a.H - 1), ix0 = min((int)x0, a.W - 1);
  const int iy1 = min(iy0 + 1, a.H - 1), ix1 = min(ix0 + 1, a.W - 1);
  const T* p00 = src + (int64_t)iy0 * rs + (int64_t)ix0 * cs;
  const T* p01 = src + (int64_t)iy0 * rs + (int64_t)ix1 * cs;
  const T* p10 = src + (int64_t)iy1 * rs + (int64_t)ix0 * cs;
  const T* p11 = src + (int64_t)iy1 * rs + (int64_t)ix1 * cs;
  for (int c = 0; c < C; ++c) {
    double t = 0.0;
    t += ((double)p00[c] * wy0) * wx0;
    t += ((double)p01[c] * wy0) * wx1;
    t += ((double)p10[c] * wy1) * wx0;
    t += ((double)p11[c] * wy1) * wx1;
    dst[c] = to_elem<T>(t);
  }
}

template <typename T>
__global__ void __launch_bounds__(kTypedBlock) typed_stack_kernel(const TypedStackArgs st, const MapArgs map) {
  const int x = blockIdx.x * kTypedBlock + (int)threadIdx.x;
  const int r = blockIdx.y;
  const int d0 = blockIdx.z * st.d_chunk;
  const int d1 = min(st.D, d0 + st.d_chunk);
  if (x >= st.W) return;
  double xd, yd;
  pixel_coord<kRadial>(map, (double)x, st.row_start + (double)r, 0.0f, 0.0f, &xd, &yd);
  double xc, yc;
  if (st.round_f32) {
    xc = (double)round_clip_f32(xd, (float)(st.W - 1));
    yc = (double)round_clip_f32(yd, (float)(st.H - 1));
  } else {
    xc = clip_f64(xd, (double)(st.W - 1));
    yc = clip_f64(yd, (double)(st.H - 1));
  }
  const T* proj = (const T*)st.vol + (int64_t)d0 * st.proj_stride;
  const int64_t o0 = ((int64_t)d0 * st.nrows + r) * (int64_t)st.W + x;
  const int64_t out_step = (int64_t)st.nrows * st.W;
  // (a folding model, chunk semantics: a row coordinate outside the reference's band is reflected inside it -- see
  // stack_rows_kernel)
  const bool outside = st.round_f32 && st.rbh > 0 && (yc < (double)st.rb0 || yc > (double)(st.rb0 + st.rbh - 1));
  for (int d = d0; d < d1; ++d) {
    double t;
    if (outside) {
      const T* band = proj + (int64_t)st.rb0 * st.row_stride;
      const int64_t rs = st.row_stride;
      t = mc_sample_outside([&](long long rr, long long cc) -> double { return (double)band[rr * rs + cc]; }, st.rbh, st.W,
                            (double)((float)yc - (float)st.rb0), xc, 1, kModeReflect);      // (a float32 subtraction in the reference)
    } else {
      t = sample_typed<T>(proj, st.row_stride, 1, st.H, st.W, yc, xc, 1);
    }
    const T v = to_elem<T>(t);
    const int64_t o = o0 + (int64_t)(d - d0) * out_step;
    if (st.out_f32) ((float*)st.out)[o] = (float)(double)v;   // sino[i] = ... into a float32 array, postprocessing.py:224-227
    else ((T*)st.out)[o] = v;
    proj += st.proj_stride;
  }
}

// Closed-form radial mapping of a list of points (unwarp_line_forward, discorpy/post/postprocessing.py:36-64;
// find_point_to_point, discorpy/util/utility.py:192-230): out = centre + B(r) (p - centre), float64 throughout,
// B evaluated as the reference's sum a_i r^i accumulated left to right (powers by repeated multiplication).
__global__ void __launch_bounds__(kTypedBlock) map_points_kernel(const double* __restrict__ yx_in, double* __restrict__ yx_out,
                                                                int64_t n, const MapArgs map) {
  const int64_t i = (int64_t)blockIdx.x * kTypedBlock + threadIdx.x;
  if (i >= n) return;
  const double y = yx_in[2 * i], x = yx_in[2 * i + 1];
  const double xd = x - map.xc, yd = y - map.yc;
  const double rd = sqrt_rn(xd * xd + yd * yd);
  double factor = 0.0, p = 1.0;
  for (int k = 0; k < map.nfact; ++k) {
    factor += map.fact[k] * p;
    p *= rd;
  }
  yx_out[2 * i] = map.yc + factor * yd;
  yx_out[2 * i + 1] = map.xc + factor * xd;
}

// The homography applied to a list of points (correct_perspective_line, discorpy/post/postprocessing.py:414-441):
// xn = (c1 x + c2 y + c3) / (c7 x + c8 y + 1), yn = (c4 x + c5 y + c6) / (c7 x + c8 y + 1) -- numpy's operation order, IEEE
// divisions (every operation correctly rounded: bit-equal to the reference).
__global__ void __launch_bounds__(kTypedBlock) map_points_persp_kernel(const double* __restrict__ yx_in, double* __restrict__ yx_out,
                                                                      int64_t n, const MapArgs map) {
  const int64_t i = (int64_t)blockIdx.x * kTypedBlock + threadIdx.x;
  if (i >= n) return;
  const double y = yx_in[2 * i], x = yx_in[2 * i + 1];
  const double den = (map.coef[6] * x + map.coef[7] * y) + 1.0;
  yx_out[2 * i] = ((map.coef[3] * x + map.coef[4] * y) + map.coef[5]) / den;
  yx_out[2 * i + 1] = ((map.coef[0] * x + map.coef[1] * y) + map.coef[2]) / den;
}

hipError_t launch_map_points_persp(const double* yx_in, double* yx_out, int64_t n, const MapArgs& map, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(map_points_persp_kernel, dim3((unsigned)((n + kTypedBlock - 1) / kTypedBlock)), dim3(kTypedBlock), 0, stream, yx_in, yx_out, n,
                     map);
  return hipGetLastError();
}

hipError_t launch_map_points(const double* yx_in, double* yx_out, int64_t n, const MapArgs& map, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(map_points_kernel, dim3((unsigned)((n + kTypedBlock - 1) / kTypedBlock)), dim3(kTypedBlock), 0, stream,
                     yx_in, yx_out, n, map);
  return hipGetLastError();
}

// ------------------------------------------------------------------ launchers

// x tiles of 256 pixels, one row per blockIdx.y (65535 per grid.z slice)
static dim3 image_grid(int H, int W) {
  return dim3((unsigned)((W + kTypedBlock - 1) / kTypedBlock), (unsigned)(H < 65535 ? H : 65535), (unsigned)((H + 65534) / 65535));
}

template <typename T>
static hipError_t launch_image_t(int map_kind, const TypedImageArgs& a, const MapArgs& map, const CoordArgs& ca,
                                 hipStream_t stream) {
  const int64_t total = map_kind == 3 ? ca.npts : (int64_t)a.H * a.W;
  if (total == 0) return hipSuccess;
  const dim3 block(kTypedBlock);
  const dim3 grid = map_kind == 3 ? dim3((unsigned)((total + kTypedBlock - 1) / kTypedBlock)) : image_grid(a.H, a.W);
  switch (map_kind) {
    case 0: hipLaunchKernelGGL((typed_image_kernel<T, kRadial>), grid, block, 0, stream, a, map, ca); break;
    case 1: hipLaunchKernelGGL((typed_image_kernel<T, kPersp>), grid, block, 0, stream, a, map, ca); break;
    case 2: hipLaunchKernelGGL((typed_image_kernel<T, kFused>), grid, block, 0, stream, a, map, ca); break;
    default: hipLaunchKernelGGL((typed_image_kernel<T, 3>), grid, block, 0, stream, a, map, ca); break;
  }
  return hipGetLastError();
}

#define DCP_TYPED_DISPATCH(dtype, CALL)             \
  switch (dtype) {                                  \
    case kF32: return CALL(float);                  \
    case kF64: return CALL(double);                 \
    case kU8: return CALL(uint8_t);                 \
    case kI8: return CALL(int8_t);                  \
    case kU16: return CALL(uint16_t);               \
    case kI16: return CALL(int16_t);                \
    case kU32: return CALL(uint32_t);               \
    case kI32: return CALL(int32_t);                \
    case kI64: return CALL(int64_t);                \
    case kU64: return CALL(uint64_t);               \
    case kBool: return CALL(Bool8);                 \
    default: return hipErrorInvalidValue;           \
  }

hipError_t launch_typed_image(int map_kind, const TypedImageArgs& a, const MapArgs& map, const CoordArgs& ca,
                              hipStream_t stream) {
  set_last_kernel_name("typed_image_kernel (one thread per pixel, any element type)");
#define DCP_CALL(T) launch_image_t<T>(map_kind, a, map, ca, stream)
  DCP_TYPED_DISPATCH(a.dtype, DCP_CALL)
#undef DCP_CALL
}

template <typename T>
static hipError_t launch_channels_t(const TypedImageArgs& a, const MapArgs& map, int channels, hipStream_t stream) {
  hipLaunchKernelGGL((typed_channels_kernel<T>), image_grid(a.rows, a.W), dim3(kTypedBlock), 0, stream, a, map, channels);
  return hipGetLastError();
}

hipError_t launch_typed_channels(const TypedImageArgs& a, const MapArgs& map, int channels, hipStream_t stream) {
  set_last_kernel_name("typed_channels_kernel (interleaved channels, one thread per pixel)");
#define DCP_CALL(T) launch_channels_t<T>(a, map, channels, stream)
  DCP_TYPED_DISPATCH(a.dtype, DCP_CALL)
#undef DCP_CALL
}

template <typename T>
static hipError_t launch_stack_t(const TypedStackArgs& st, const MapArgs& map, hipStream_t stream) {
  const dim3 grid((unsigned)((st.W + kTypedBlock - 1) / kTypedBlock), (unsigned)st.nrows,
                  (unsigned)((st.D + st.d_chunk - 1) / st.d_chunk));
  hipLaunchKernelGGL((typed_stack_kernel<T>), grid, dim3(kTypedBlock), 0, stream, st, map);
  return hipGetLastError();
}

hipError_t launch_typed_stack(const TypedStackArgs& st, const MapArgs& map, hipStream_t stream) {
  set_last_kernel_name("typed_stack_kernel (one thread per (row, x), any element type)");
  if (st.D == 0 || st.nrows == 0) return hipSuccess;
#define DCP_CALL(T) launch_stack_t<T>(st, map, stream)
  DCP_TYPED_DISPATCH(st.dtype, DCP_CALL)
#undef DCP_CALL
}

}  // namespace dcp
