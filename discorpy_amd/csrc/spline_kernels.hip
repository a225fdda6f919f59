// spline_kernels.hip -- spline orders 2..5 of the unwarp path: what
// scipy.ndimage.map_coordinates does for order >= 2, which the reference reaches through the
// `order` argument of unwarp_image_backward / correct_perspective_image
// (discorpy/post/postprocessing.py:111,147,462,491; order=3 in examples/readthedocs_demo/demo_07.py:60).
//
//   spline_expand_kernel   float32 image -> float64 plane, padded by 12 for 'nearest' / 'grid-constant'
//   spline_filter_kernel   in-place recursive B-spline prefilter along one axis (one line per thread)
//   spline_remap_kernel    (order+1)^2-tap gather at the radial / perspective / explicit coordinates
//
// The arithmetic is operation for operation that of the spline section of oracle/unwarp_oracle.c
// (same expressions, same order, no contraction; pow(z, n) is evaluated on the host and passed in),
// so GPU and oracle agree bit for bit; the oracle is within one float32 ulp of scipy.
// This is the first, correctness-first version: the prefilter walks lines serially (a 4096^2 image
// takes ~1 ms), the gather reads its taps straight from global memory.
#include "dcp_internal.h"
#include "dcp_device.h"

namespace dcp {

constexpr int kSplBlock = 256;

__global__ void __launch_bounds__(kSplBlock) spline_expand_kernel(const SplineArgs a) {
  const int64_t i = (int64_t)blockIdx.x * kSplBlock + threadIdx.x;
  if (i >= (int64_t)a.Hp * a.Wp) return;
  const int y = (int)(i / a.Wp), x = (int)(i - (int64_t)y * a.Wp);
  int sy = y - a.pad, sx = x - a.pad;
  double v;
  if (a.mode == kModeGridConstant && (sy < 0 || sy >= a.H || sx < 0 || sx >= a.W)) {
    v = 0.0;
  } else {
    sy = sy < 0 ? 0 : (sy > a.H - 1 ? a.H - 1 : sy);
    sx = sx < 0 ? 0 : (sx > a.W - 1 ? a.W - 1 : sx);
    v = (double)a.src[(size_t)sy * a.src_stride + (size_t)sx * a.src_cstride];
  }
  a.coef[i] = v;
}

// One thread filters one line of n samples (stride s) in place; `axis` selects the pow() table.
__global__ void __launch_bounds__(kSplBlock) spline_filter_kernel(const SplineArgs a, int axis) {
  const int64_t line = (int64_t)blockIdx.x * kSplBlock + threadIdx.x;
  const int64_t n = axis == 0 ? a.Hp : a.Wp;
  const int64_t count = axis == 0 ? a.Wp : a.Hp;
  if (line >= count || n < 2) return;
  const int64_t s = axis == 0 ? a.Wp : 1;
  double* c = a.coef + (axis == 0 ? line : line * (int64_t)a.Wp);
  double lam = 1.0;
  for (int p = 0; p < a.npoles; ++p) lam *= (1.0 - a.poles[p]) * (1.0 - 1.0 / a.poles[p]);
  for (int64_t i = 0; i < n; ++i) c[i * s] *= lam;
  for (int p = 0; p < a.npoles; ++p) {
    const double z = a.poles[p];
    const double zpow = a.zpow[axis][p];       // reflect / wrap: unused or z^n ; mirror: z^(n-1)
    if (a.filter_kind == kSplReflect) {
      double z_i = z;
      const double z_n = zpow;
      const double c0 = c[0];
      double acc = c[0] + z_n * c[(n - 1) * s];
      for (int64_t i = 1; i < n; ++i) {
        acc += z_i * (c[i * s] + z_n * c[(n - 1 - i) * s]);
        z_i *= z;
      }
      c[0] = acc * z / (1.0 - z_i * z_i) + c0;
      for (int64_t i = 1; i < n; ++i) c[i * s] += z * c[(i - 1) * s];
      c[(n - 1) * s] *= z / (z - 1.0);
      for (int64_t i = n - 2; i >= 0; --i) c[i * s] = z * (c[(i + 1) * s] - c[i * s]);
    } else if (a.filter_kind == kSplMirror) {
      double z_i = z;
      const double z_n_1 = zpow;
      double acc = c[0] + z_n_1 * c[(n - 1) * s];
      for (int64_t i = 1; i < n - 1; ++i) {
        acc += z_i * (c[i * s] + z_n_1 * c[(n - 1 - i) * s]);
        z_i *= z;
      }
      c[0] = acc / (1.0 - z_n_1 * z_n_1);
      for (int64_t i = 1; i < n; ++i) c[i * s] += z * c[(i - 1) * s];
      c[(n - 1) * s] = (z / (z * z - 1.0)) * (c[(n - 1) * s] + z * c[(n - 2) * s]);
      for (int64_t i = n - 2; i >= 0; --i) c[i * s] = z * (c[(i + 1) * s] - c[i * s]);
    } else {
      double z_i = z, acc = c[0];
      for (int64_t i = n - 1; i > 0; --i) {
        acc += z_i * c[i * s];
        z_i *= z;
      }
      c[0] = acc / (1.0 - z_i);
      for (int64_t i = 1; i < n; ++i) c[i * s] += z * c[(i - 1) * s];
      z_i = z;
      acc = c[(n - 1) * s];
      for (int64_t i = 0; i < n - 1; ++i) {
        acc += z_i * c[i * s];
        z_i *= z;
      }
      c[(n - 1) * s] = acc * z / (z_i - 1.0);
      for (int64_t i = n - 2; i >= 0; --i) c[i * s] = z * (c[(i + 1) * s] - c[i * s]);
    }
  }
}

// centred B-spline weights (the expressions of spline_weights() in the oracle); returns the first tap
template <int ORDER>
__device__ __forceinline__ int spline_weights(double x, double* w) {
  double s;
  if constexpr (ORDER & 1) s = __builtin_floor(x);
  else s = __builtin_floor(x + 0.5);
  const double t = x - s;
  const int start = (int)s - ORDER / 2;
  double y = t, z = 1.0 - t, t2;
  if constexpr (ORDER == 2) {
    w[1] = 0.75 - t * t;
    y = 0.5 + t;
    w[2] = 0.5 * y * y;
    w[0] = 1.0 - w[1] - w[2];
  } else if constexpr (ORDER == 3) {
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = z * z * z / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
  } else if constexpr (ORDER == 4) {
    t2 = t * t;
    w[2] = t2 * (t2 * 0.25 - 0.625) + 115.0 / 192.0;
    y = 1.0 + t;
    z = 1.0 - t;
    w[1] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
    w[3] = z * (z * (z * (5.0 - z) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
    y = 0.5 - t;
    y *= y;
    w[0] = y * y / 24.0;
    w[4] = 1.0 - w[0] - w[1] - w[2] - w[3];
  } else {
    t2 = y * y;
    w[2] = t2 * (t2 * (0.25 - y / 12.0) - 0.5) + 0.55;
    t2 = z * z;
    w[3] = t2 * (t2 * (0.25 - z / 12.0) - 0.5) + 0.55;
    y += 1.0;
    w[1] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
    y = z + 1.0;
    w[4] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
    t2 = z * z;
    w[0] = t2 * t2 * z / 120.0;
    w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4];
  }
  return start;
}

__device__ __forceinline__ int spline_fold(int i, int n, int mode) {
  if (i >= 0 && i < n) return i;
  if (mode == kModeReflect || mode == kModeGridMirror) {
    const int s2 = 2 * n;
    i %= s2;
    if (i < 0) i += s2;
    return i < n ? i : s2 - 1 - i;
  }
  if (mode == kModeGridWrap) {
    i %= n;
    return i < 0 ? i + n : i;
  }
  if (mode == kModeNearest || mode == kModeGridConstant) return i < 0 ? 0 : n - 1;
  if (n == 1) return 0;
  const int s2 = 2 * n - 2;
  i %= s2;
  if (i < 0) i += s2;
  return i < n ? i : s2 - i;
}

// MAPKIND 0 radial, 1 perspective, 2 explicit coordinates (dst[i] for point i)
template <int MAPKIND, int ORDER>
__global__ void __launch_bounds__(kSplBlock) spline_remap_kernel(const SplineArgs a, const MapArgs map, const CoordArgs ca,
                                                                float* dst) {
  const int64_t i = (int64_t)blockIdx.x * kSplBlock + threadIdx.x;
  const int64_t total = MAPKIND == 2 ? ca.npts : (int64_t)a.H * a.W;
  if (i >= total) return;
  double yc, xc;   // float32-rounded (or caller-supplied) coordinates in the unpadded image
  if constexpr (MAPKIND == 2) {
    if (ca.is_f64) {
      yc = ((const double*)ca.ycoord)[i];
      xc = ((const double*)ca.xcoord)[i];
    } else {
      yc = (double)((const float*)ca.ycoord)[i];
      xc = (double)((const float*)ca.xcoord)[i];
    }
    yc = clip_f64(yc, (double)(a.H - 1));
    xc = clip_f64(xc, (double)(a.W - 1));
  } else {
    const int y = (int)(i / a.W), x = (int)(i - (int64_t)y * a.W);
    const float wmaxf = (float)(a.W - 1), hmaxf = (float)(a.H - 1);
    double xd, yd;
    if constexpr (MAPKIND == 0) {
      const double xu = (double)x - map.xc, yu = (double)y - map.yc;
      const double xx = xu * xu, yy = yu * yu;
      const double r2 = xx + yy;
      const double ru = sqrt_rn(r2);
      const double f = poly_lds(map.fact, map.nfact, r2, ru);
      xd = __builtin_fma(f, xu, map.xc);
      yd = __builtin_fma(f, yu, map.yc);
    } else {
      const double X = (double)x, Y = (double)y;
      const double den = (map.coef[6] * X + map.coef[7] * Y) + 1.0;
      const double nx = (map.coef[0] * X + map.coef[1] * Y) + map.coef[2];
      const double ny = (map.coef[3] * X + map.coef[4] * Y) + map.coef[5];
      xd = nx / den;
      yd = ny / den;
    }
    xc = (double)round_clip_f32(xd, wmaxf);
    yc = (double)round_clip_f32(yd, hmaxf);
  }
  double wy[6], wx[6];
  const int sy = spline_weights<ORDER>(yc + (double)a.pad, wy);
  const int sx = spline_weights<ORDER>(xc + (double)a.pad, wx);
  int ix[ORDER + 1];
#pragma unroll
  for (int k = 0; k <= ORDER; ++k) ix[k] = spline_fold(sx + k, a.Wp, a.mode);
  double t = 0.0;
#pragma unroll
  for (int j = 0; j <= ORDER; ++j) {
    const double* row = a.coef + (size_t)spline_fold(sy + j, a.Hp, a.mode) * (size_t)a.Wp;
#pragma unroll
    for (int k = 0; k <= ORDER; ++k) t += (row[ix[k]] * wy[j]) * wx[k];
  }
  dst[i] = (float)t;
}

template <int MAPKIND>
static hipError_t launch_remap_order(const SplineArgs& a, const MapArgs& map, const CoordArgs& ca, float* dst,
                                     int64_t total, hipStream_t stream) {
  const dim3 grid((unsigned)((total + kSplBlock - 1) / kSplBlock));
  switch (a.order) {
    case 2: hipLaunchKernelGGL((spline_remap_kernel<MAPKIND, 2>), grid, dim3(kSplBlock), 0, stream, a, map, ca, dst); break;
    case 3: hipLaunchKernelGGL((spline_remap_kernel<MAPKIND, 3>), grid, dim3(kSplBlock), 0, stream, a, map, ca, dst); break;
    case 4: hipLaunchKernelGGL((spline_remap_kernel<MAPKIND, 4>), grid, dim3(kSplBlock), 0, stream, a, map, ca, dst); break;
    default: hipLaunchKernelGGL((spline_remap_kernel<MAPKIND, 5>), grid, dim3(kSplBlock), 0, stream, a, map, ca, dst); break;
  }
  return hipGetLastError();
}

hipError_t launch_spline(const SplineArgs& a, int map_kind, const MapArgs& map, const CoordArgs& ca, float* dst,
                         hipStream_t stream) {
  const int64_t plane = (int64_t)a.Hp * a.Wp;
  hipLaunchKernelGGL(spline_expand_kernel, dim3((unsigned)((plane + kSplBlock - 1) / kSplBlock)), dim3(kSplBlock), 0,
                     stream, a);
  hipLaunchKernelGGL(spline_filter_kernel, dim3((unsigned)((a.Wp + kSplBlock - 1) / kSplBlock)), dim3(kSplBlock), 0,
                     stream, a, 0);
  hipLaunchKernelGGL(spline_filter_kernel, dim3((unsigned)((a.Hp + kSplBlock - 1) / kSplBlock)), dim3(kSplBlock), 0,
                     stream, a, 1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int64_t total = map_kind == 2 ? ca.npts : (int64_t)a.H * a.W;
  if (total == 0) return hipSuccess;
  if (map_kind == 0) return launch_remap_order<0>(a, map, ca, dst, total, stream);
  if (map_kind == 1) return launch_remap_order<1>(a, map, ca, dst, total, stream);
  return launch_remap_order<2>(a, map, ca, dst, total, stream);
}

}  // namespace dcp
