// dcp_device.h -- device-side fp64 building blocks shared by unwarp_kernels.hip and
// spline_kernels.hip: correctly rounded sqrt and homography division, the even/odd polynomial,
// the float32 round-and-clip.  Every fused multiply-add is explicit (build with -ffp-contract=off)
// so that the arithmetic is the same sequence of IEEE operations as oracle/unwarp_oracle.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits>
#include <type_traits>
#include "dcp_internal.h"

namespace dcp {

// ------------------------------------------------------------------ debug build: every LDS tap checked against its slab
// `make -C discorpy_amd/csrc bounds` (-DDCP_DEBUG_BOUNDS, a separate library: lib/libdiscorpy_hip_bounds.so) compiles a check
// in front of every tap the staged kernels read from LDS: DCP_BOUNDS(first byte, bytes spanned, slab bytes, site).  A tap outside
// its slab is counted (dcp_debug_bounds; the first offender is kept) instead of trapping, so a whole randomised campaign
// (tools/fuzz_parity.py --bounds) runs through and reports.  The certificate (api_core.cpp tile_deviation_certified) says
// the count must stay 0.  Nothing of this exists in the product build.
#ifdef DCP_DEBUG_BOUNDS
static __device__ unsigned long long g_bounds[4];      // per translation unit: violations, first offset, slab bytes, site
__device__ __forceinline__ void check_lds_tap(uint32_t first, uint32_t span, uint32_t slab, int site) {
  if (first > slab || span > slab - first) {
    if (atomicAdd(&g_bounds[0], 1ull) == 0ull) {
      g_bounds[1] = first;
      g_bounds[2] = slab;
      g_bounds[3] = (unsigned long long)site;
    }
  }
}
#define DCP_BOUNDS(first, span, slab, site) ::dcp::check_lds_tap((uint32_t)(first), (uint32_t)(span), (uint32_t)(slab), (site))
// stack_wg_kernel's partial wait `s_waitcnt vmcnt(N)` stands for "the fill has landed" only if the wave issued at least N
// vector-memory instructions AFTER its share of the fill: counted here, a shortfall reported as a violation at site 9
#define DCP_VM_COUNTER(name) int name = 0
#define DCP_VM_ISSUED(name) (++(name))
#define DCP_VM_RESET(name) ((name) = 0)
#define DCP_VM_CHECK(lane_is_active, name, need)                               \
  do {                                                                         \
    if ((lane_is_active) && (name) < (need)) ::dcp::check_lds_tap(1u + (uint32_t)(name), 0u, 0u, 9); \
  } while (0)
#define DCP_DEFINE_BOUNDS_READER(name)                                                                       \
  hipError_t name(unsigned long long* out, bool reset) {                                                     \
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds), sizeof(g_bounds));                         \
    if (e != hipSuccess || !reset) return e;                                                                 \
    const unsigned long long zero[4] = {0, 0, 0, 0};                                                         \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bounds), zero, sizeof(zero));                                      \
  }
#else
#define DCP_BOUNDS(first, span, slab, site) do { } while (0)
#define DCP_VM_COUNTER(name) do { } while (0)
#define DCP_VM_ISSUED(name) do { } while (0)
#define DCP_VM_RESET(name) do { } while (0)
#define DCP_VM_CHECK(lane_is_active, name, need) do { } while (0)
#define DCP_DEFINE_BOUNDS_READER(name)                                     \
  hipError_t name(unsigned long long* out, bool) {                         \
    out[0] = out[1] = out[2] = out[3] = 0;                                 \
    return hipSuccess;                                                     \
  }
#endif

// ------------------------------------------------------------------ LDS-DMA the compiler does not know about
// hipcc tracks a `buffer_load ... lds` it issued as a pending write to ALL of LDS and puts `s_waitcnt vmcnt(0)` in front of the next LDS
// read of the wave, whichever slab that read is from (SIInsertWaitcnts needs alias scopes to tell slabs apart; a slab picked by a
// run-time index has none): a kernel that streams the NEXT box into one slab while it blends out of the other was made to wait for the
// stream before every blend.  These two hide the load in an asm statement (M0 -- the LDS address of lane 0's 16 bytes -- saved and
// restored around it); the caller owns the wait (`s_waitcnt vmcnt(n)` + barrier before the slab is read).
typedef int dcp_rsrc_words __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dcp_rsrc_words raw_rsrc_words(const void* base, uint32_t bytes) {       // as make_buffer_rsrc(base, 0, bytes, 0x00020000)
  const uint64_t a = (uint64_t)(uintptr_t)base;
  dcp_rsrc_words r = {(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  return r;
}
__device__ __forceinline__ void lds_dma16_untracked(dcp_rsrc_words rs, uint32_t lds_addr, uint32_t voffset) {
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved)
               : "s"(lds_addr), "v"(voffset), "s"(rs)
               : "memory");
}

// ------------------------------------------------------------------ fp64 helpers

constexpr double kTinyR2 = 1e-300;   // keeps rsq finite at the centre pixel; absorbed everywhere else

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

// The same for x > 0 (no zero select) in the per-pixel coordinate loops of the frame and stack kernels, with the refinement of h
// left out: after the coupled step g is good to ~2^-52 and the exact residual d = x - g^2 is of that relative size, so the
// correction d * h needs h only to the seed's accuracy.  tools/ubench_sqrt.hip, 4.3e9 operands r^2 of a frame: 0 results differ
// from the correctly rounded sqrt with h refined, 79 (1.8e-8) differ by one ulp without -- a third of a pixel per 4096^2 frame
// whose ru is one ulp off, which moves a float32-rounded coordinate with probability < 1e-16 per pixel (ru enters B through
// ru * O(r^2) ~ 0.1 and the coordinate is rounded to 24 bits): outputs stay equal to the oracle's (correctly rounded sqrt).
// One float64 instruction of 44 per pixel: 28.8 -> 28.1 us per frame in the multi-frame launch, same box.  DCP_SQRT_REFINE_H=1
// restores the refinement (A/B builds).
#ifndef DCP_SQRT_REFINE_H
#define DCP_SQRT_REFINE_H 0
#endif
__device__ __forceinline__ double sqrt_pos(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
#if DCP_SQRT_REFINE_H
  h = __builtin_fma(h, r, h);
#endif
  double d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}

// B(ru) = sum a_i ru^i, split into even and odd powers so that only one multiply by ru is
// needed: E = a0 + r2 (a2 + r2 (a4 + ...)), O = a1 + r2 (a3 + ...), B = fma(ru, O, E).
// Same operation order as poly_kernel() in oracle/unwarp_oracle.c.
template <int NF>
__device__ __forceinline__ double poly_inline(const double* __restrict__ a, double lead_e, double lead_o,
                                              double r2, double ru) {
  // lead_e / lead_o: the highest even / odd coefficient, passed separately so the caller can pin
  // them in VGPRs (a VOP3 fma takes one SGPR operand; the first Horner step would need two)
  if constexpr (NF == 0) {
    return 0.0;
  } else {
    constexpr int ne = (NF + 1) / 2, no = NF / 2;
    double E = lead_e;
#pragma unroll
    for (int k = ne - 2; k >= 0; --k) E = __builtin_fma(r2, E, a[2 * k]);
    if constexpr (no == 0) {
      return E;
    } else {
      double O = lead_o;
#pragma unroll
      for (int k = no - 2; k >= 0; --k) O = __builtin_fma(r2, O, a[2 * k + 1]);
      return __builtin_fma(ru, O, E);
    }
  }
}
template <int NF>
__device__ __forceinline__ void poly_leads(const double* __restrict__ a, double* lead_e, double* lead_o) {
  *lead_e = 0.0;
  *lead_o = 0.0;
  if constexpr (NF > 0) {
    double e = a[2 * ((NF + 1) / 2 - 1)];
    asm volatile("" : "+v"(e));
    *lead_e = e;
  }
  if constexpr (NF > 1) {
    double o = a[2 * (NF / 2 - 1) + 1];
    asm volatile("" : "+v"(o));
    *lead_o = o;
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

// np.float32(np.clip(v, 0, len-1)) == clip(float32(v), 0, len-1): rounding is monotone and
// both bounds are float32 numbers, so the clip is done after the conversion, in fp32.
__device__ __forceinline__ float round_clip_f32(double v, float hi) {
  return __builtin_amdgcn_fmed3f((float)v, 0.0f, hi);
}
__device__ __forceinline__ double clip_f64(double v, double hi) {
  v = v < 0.0 ? 0.0 : v;
  return v > hi ? hi : v;
}

// nx/den and ny/den, both correctly rounded, from ONE refined reciprocal (v_rcp_f64 + one Newton
// step, then an exact residual and a correction per quotient: the correction term is good to ~2^-90 of
// the quotient, so the result is the correctly rounded one unless the true quotient lies that close
// to a rounding boundary).  Valid while no intermediate leaves the normal range -- the C ABI checks the
// homography and the image size on the host and otherwise leaves MapArgs::fast_div at 0 (compiler's
// full IEEE division).  tools/ubench_div.hip: 0 mismatches against the IEEE division on 2.1e9 quotients
// (round 2 used two Newton steps: same count, two fused multiply-adds more per pixel).
__device__ __forceinline__ void div2_rn(double nx, double ny, double den, double* qx, double* qy) {
  double r = __builtin_amdgcn_rcp(den);
  double e = __builtin_fma(-den, r, 1.0);
  r = __builtin_fma(r, e, r);
  double q = nx * r;
  double t = __builtin_fma(-den, q, nx);
  *qx = __builtin_fma(t, r, q);
  q = ny * r;
  t = __builtin_fma(-den, q, ny);
  *qy = __builtin_fma(t, r, q);
}

// ------------------------------------------------------------------ one pixel of a coordinate map

// Source coordinate (float64, unclipped) of output pixel (x, y): the radial map of
// postprocessing.py:138-145, the homography of :448-455, or the homography's float32-rounded
// result fed to the radial map (the fused map).  Same operations as map_coord() of
// unwarp_kernels.hip without the per-row / per-column hoisting -- for the kernels that are not
// the float32 hot path (spline orders, other element types).
template <int KIND>
__device__ __forceinline__ void pixel_coord(const MapArgs& map, double X, double Y, float wmaxf, float hmaxf,
                                            double* xd_out, double* yd_out) {
  double xu, yu;
  if constexpr (KIND == kRadial) {
    xu = X - map.xc;
    yu = Y - map.yc;
  } else {
    const double den = (map.coef[6] * X + map.coef[7] * Y) + 1.0;
    const double nx = (map.coef[0] * X + map.coef[1] * Y) + map.coef[2];
    const double ny = (map.coef[3] * X + map.coef[4] * Y) + map.coef[5];
    double xd, yd;
    if (map.fast_div) {
      div2_rn(nx, ny, den, &xd, &yd);
    } else {
      xd = nx / den;
      yd = ny / den;
    }
    if constexpr (KIND == kPersp) {
      *xd_out = xd;
      *yd_out = yd;
      return;
    }
    xu = (double)round_clip_f32(xd, wmaxf) - map.xc;
    yu = (double)round_clip_f32(yd, hmaxf) - map.yc;
  }
  const double xx = xu * xu, yy = yu * yu;
  const double r2 = xx + yy;
  const double ru = sqrt_rn(r2);
  const double f = poly_lds(map.fact, map.nfact, r2, ru);
  *xd_out = __builtin_fma(f, xu, map.xc);
  *yd_out = __builtin_fma(f, yu, map.yc);
}

// ------------------------------------------------------------------ the hoisted evaluation (tiled kernels)

// Per-column (thread) invariants of the coordinate map.
struct ColCtx {
  double cx0, cx1, cx2;      // radial: xu, xu^2, -.  perspective/fused: c7*x, c1*x, c4*x
  double lead_e, lead_o;     // leading even / odd polynomial coefficient pinned in VGPRs
};

template <int KIND, int NF>
__device__ __forceinline__ ColCtx make_col(const MapArgs& map, int x) {
  ColCtx c;
  const double xd_ = (double)x;
  if constexpr (KIND == kRadial) {
    c.cx0 = xd_ - map.xc;
    c.cx1 = c.cx0 * c.cx0;
    c.cx2 = 0.0;
  } else {
    c.cx0 = map.coef[6] * xd_;
    c.cx1 = map.coef[0] * xd_;
    c.cx2 = map.coef[3] * xd_;
  }
  c.lead_e = c.lead_o = 0.0;
  if constexpr (NF >= 0 && KIND != kPersp) poly_leads<NF>(map.fact, &c.lead_e, &c.lead_o);
  return c;
}

// Per-row invariants written to LDS by one thread per row: radial (yu, max(yu^2, tiny)),
// perspective (c8*y, c2*y, c5*y).
template <int KIND, int RW>
__device__ __forceinline__ void fill_row(const MapArgs& map, double (*row)[RW], int slot, double y) {
  if constexpr (KIND == kRadial) {
    const double yu = y - map.yc;
    const double yu2 = yu * yu;
    row[slot][0] = yu;
    // r2 = xu^2 + yu^2 must stay > 0 for rsq; the bias is absorbed by the addition unless
    // xu == yu == 0, where the coordinate is xc + B*0 whatever B is.
    row[slot][1] = yu2 > kTinyR2 ? yu2 : kTinyR2;
  } else {
    row[slot][0] = map.coef[7] * y;   // c8*y
    row[slot][1] = map.coef[1] * y;   // c2*y
    row[slot][2] = map.coef[4] * y;   // c5*y
  }
}

// Source coordinate (float64, unclipped) of the pixel in column ctx / LDS row slot k.
template <int KIND, int NF, int RW, int FASTDIV = -1>   // FASTDIV: 1 / 0 fixed at compile time, -1 = map.fast_div
__device__ __forceinline__ void map_coord(const MapArgs& map, const double (*s_row)[RW], const double* s_coef,
                                          const ColCtx& c, int k, float wmaxf, float hmaxf, double* xd_out,
                                          double* yd_out) {
  double xd, yd;
  if constexpr (KIND == kRadial) {
    const double yu = s_row[k][0];
    const double r2 = c.cx1 + s_row[k][1];
    const double g = sqrt_pos(r2);                 // r2 > 0 (row table)
    double f;
    if constexpr (NF >= 0) f = poly_inline<NF>(map.fact, c.lead_e, c.lead_o, r2, g);
    else f = poly_lds(s_coef, map.nfact, r2, g);
    xd = __builtin_fma(f, c.cx0, map.xc);
    yd = __builtin_fma(f, yu, map.yc);
  } else {
    // postprocessing.py:453-455, numpy order: (c7*x + c8*y) + 1.0 etc., true divisions
    const double den = (c.cx0 + s_row[k][0]) + 1.0;
    const double nx = (c.cx1 + s_row[k][1]) + map.coef[2];
    const double ny = (c.cx2 + s_row[k][2]) + map.coef[5];
    if (FASTDIV == 1 || (FASTDIV < 0 && map.fast_div)) {
      div2_rn(nx, ny, den, &xd, &yd);
    } else {
      xd = nx / den;
      yd = ny / den;
    }
    if constexpr (KIND == kFused) {
      // float32-rounded perspective coordinate, then the radial map evaluated there
      const double xp = (double)round_clip_f32(xd, wmaxf);
      const double yp = (double)round_clip_f32(yd, hmaxf);
      const double xu = xp - map.xc;
      const double yu = yp - map.yc;
      const double xx = xu * xu;
      const double yy = yu * yu;
      // as in the radial branch: keep rsq finite at the centre; there xu == yu == 0 and the
      // coordinate is xc + B*0 whatever B is
      const double r2 = __builtin_fmax(xx + yy, kTinyR2);
      const double ru = sqrt_pos(r2);
      double f;
      if constexpr (NF >= 0) f = poly_inline<NF>(map.fact, c.lead_e, c.lead_o, r2, ru);
      else f = poly_lds(s_coef, map.nfact, r2, ru);
      xd = __builtin_fma(f, xu, map.xc);
      yd = __builtin_fma(f, yu, map.yc);
    }
  }
  *xd_out = xd;
  *yd_out = yd;
}

// One pixel of the radial or perspective map without hoisting, polynomial length fixed at compile time (NF >= 0: coefficients
// straight from the kernel arguments) -- the corner pixels of a workgroup tile (remap_wg_kernel, spline_wg_kernel).
template <int KIND, int NF>
__device__ __forceinline__ void corner_coord(const MapArgs& map, double X, double Y, double* xd, double* yd) {
  if constexpr (KIND == kRadial) {
    const double xu = X - map.xc, yu = Y - map.yc;
    const double r2 = xu * xu + yu * yu;
    const double ru = sqrt_rn(r2);
    double f;
    if constexpr (NF >= 0) {
      double le, lo;
      poly_leads<NF>(map.fact, &le, &lo);
      f = poly_inline<NF>(map.fact, le, lo, r2, ru);
    } else {
      f = poly_lds(map.fact, map.nfact, r2, ru);          // straight from the kernel arguments (uniform loads)
    }
    *xd = __builtin_fma(f, xu, map.xc);
    *yd = __builtin_fma(f, yu, map.yc);
  } else {
    const double den = (map.coef[6] * X + map.coef[7] * Y) + 1.0;
    *xd = ((map.coef[0] * X + map.coef[1] * Y) + map.coef[2]) / den;
    *yd = ((map.coef[3] * X + map.coef[4] * Y) + map.coef[5]) / den;
  }
}

// ------------------------------------------------------------------ explicit coordinates outside the image

// What scipy.ndimage.map_coordinates does with a coordinate outside [0, len - 1] at orders 0 and 1 (the reference hands
// `map_index` to it with the caller's `mode`, postprocessing.py:489-491): ni_interpolation.c map_coordinate() moves the
// coordinate into the extended image, the taps that still fall outside fold by the same mode, and the two constant modes
// read cval = 0 there.  Mode numbers: BoundaryMode in dcp_internal.h (0 reflect, 1 grid-mirror, 2 constant,
// 3 grid-constant, 4 nearest, 5 mirror, 6 grid-wrap, 7 wrap).  Restated in oracle/unwarp_oracle.c; both are tested
// against scipy itself.
__device__ __forceinline__ double mc_map_coordinate(double in, int len, int mode) {
  const double n = (double)len;
  if (in < 0.0) {
    switch (mode) {
      case 5: {                                      // mirror
        if (len <= 1) return 0.0;
        const double sz2 = 2.0 * n - 2.0;
        in = sz2 * (double)(long long)(-in / sz2) + in;
        return in <= 1.0 - n ? in + sz2 : -in;
      }
      case 0:
      case 1: {                                      // reflect / grid-mirror
        if (len <= 1) return 0.0;
        const double sz2 = 2.0 * n;
        if (in < -sz2) in = sz2 * (double)(long long)(-in / sz2) + in;
        return in < -n ? in + sz2 : (in > -1e-15 ? 1e-15 : -in) - 1.0;
      }
      case 7: {                                      // wrap
        if (len <= 1) return 0.0;
        const double sz = n - 1.0;
        return in + sz * ((double)(long long)(-in / sz) + 1.0);
      }
      case 6: {                                      // grid-wrap
        if (len <= 1) return 0.0;
        return in + n * ((double)(long long)((-1.0 - in) / n) + 1.0);
      }
      case 4: return 0.0;                            // nearest
      case 2: return -1.0;                           // constant
      default: return in;                            // grid-constant
    }
  }
  if (in > n - 1.0) {
    switch (mode) {
      case 5: {
        if (len <= 1) return 0.0;
        const double sz2 = 2.0 * n - 2.0;
        in -= sz2 * (double)(long long)(in / sz2);
        return in >= n ? sz2 - in : in;
      }
      case 0:
      case 1: {
        if (len <= 1) return 0.0;
        const double sz2 = 2.0 * n;
        in -= sz2 * (double)(long long)(in / sz2);
        return in >= n ? sz2 - in - 1.0 : in;
      }
      case 7: {
        if (len <= 1) return 0.0;
        const double sz = n - 1.0;
        return in - sz * (double)(long long)(in / sz);
      }
      case 6: {
        if (len <= 1) return 0.0;
        return in - n * (double)(long long)((in + 1.0) / n);
      }
      case 4: return n - 1.0;
      case 2: return -1.0;
      default: return in;
    }
  }
  return in;
}

// index of tap i of a line of len samples under `mode`; -1: outside, reads cval (the two constant modes)
__device__ __forceinline__ long long mc_fold_tap(long long i, long long len, int mode) {
  if (i >= 0 && i < len) return i;
  switch (mode) {
    case 0:
    case 1: {
      const long long s2 = 2 * len;
      i %= s2;
      if (i < 0) i += s2;
      return i < len ? i : s2 - 1 - i;
    }
    case 6: {
      i %= len;
      return i < 0 ? i + len : i;
    }
    case 4: return i < 0 ? 0 : len - 1;
    case 5: {
      if (len == 1) return 0;
      const long long s2 = 2 * len - 2;
      i %= s2;
      if (i < 0) i += s2;
      return i < len ? i : s2 - i;
    }
    case 7: {
      if (len == 1) return 0;
      const long long s = len - 1;
      i %= s;
      return i < 0 ? i + s : i;
    }
    default: return -1;
  }
}

// t of map_coordinates(src, (yc, xc), order, mode) for a point with a coordinate outside the image (any point, in fact):
// double arithmetic in scipy's order; LOAD(row, col) returns the element as a double
template <typename LOAD>
__device__ __forceinline__ double mc_sample_outside(LOAD&& load, int H, int W, double yc, double xc, int order, int mode) {
  const double y = mc_map_coordinate(yc, H, mode), x = mc_map_coordinate(xc, W, mode);
  if (mode == 2 && (y <= -1.0 || x <= -1.0)) return 0.0;
  if (order == 0) {
    const long long iy = mc_fold_tap((long long)__builtin_floor(y + 0.5), H, mode), ix = mc_fold_tap((long long)__builtin_floor(x + 0.5), W, mode);
    return (iy < 0 || ix < 0) ? 0.0 : load(iy, ix);
  }
  const double y0 = __builtin_floor(y), x0 = __builtin_floor(x);
  const double wy0 = 1.0 - (y - y0), wy1 = 1.0 - wy0;
  const double wx0 = 1.0 - (x - x0), wx1 = 1.0 - wx0;
  const long long iy0 = mc_fold_tap((long long)y0, H, mode), iy1 = mc_fold_tap((long long)y0 + 1, H, mode);
  const long long ix0 = mc_fold_tap((long long)x0, W, mode), ix1 = mc_fold_tap((long long)x0 + 1, W, mode);
  auto tap = [&](long long iy, long long ix) -> double { return (iy < 0 || ix < 0) ? 0.0 : load(iy, ix); };
  double t = 0.0;
  t += (tap(iy0, ix0) * wy0) * wx0;
  t += (tap(iy0, ix1) * wy0) * wx1;
  t += (tap(iy1, ix0) * wy1) * wx0;
  t += (tap(iy1, ix1) * wy1) * wx1;
  return t;
}

// ------------------------------------------------------------------ element types

// scipy reads every element as a double and converts the double result on the way out
// (ni_interpolation.c CASE_INTERP_OUT*): floats by a C cast; unsigned integers t > 0 ? t + 0.5 : 0,
// clamped to the maximum, truncated; signed integers rounded half away from zero, clamped, truncated.
// 64-bit integers: scipy clamps against NPY_MAX_INT64 / NPY_MAX_UINT64 converted to double -- 2^63 / 2^64, which the C cast
// behind the clamp cannot represent (undefined behaviour).  The reference's result on x86-64 is what cvttsd2si gives: the
// "integer indefinite" 0x8000000000000000 for every double outside [-2^63, 2^63); the unsigned cast is cvttsd2si(t) below 2^63,
// else cvttsd2si(t - 2^63) ^ 2^63, so 2^64 stores 0 (oracle/unwarp_oracle.c x86_cvttsd2si, pinned by golden G12).
__device__ __forceinline__ long long x86_cvttsd2si(double t) {
  return (t >= -9223372036854775808.0 && t < 9223372036854775808.0) ? (long long)t : (long long)0x8000000000000000ull;
}
template <typename T>
__device__ __forceinline__ T to_elem(double t) {
  if constexpr (std::is_same<T, float>::value) {
    return (float)t;
  } else if constexpr (std::is_same<T, double>::value) {
    return t;
  } else if constexpr (std::is_same<T, Bool8>::value) {
    Bool8 b;
    b.v = (t >= 0.0 && t < 256.0) ? (uint8_t)(int)t : (uint8_t)0;        // a C cast of the double: truncation
    return b;
  } else if constexpr (std::is_same<T, int64_t>::value || std::is_same<T, long long>::value) {
    // (round 5, golden G12b at order 3: a value STRICTLY above 2^63 takes scipy's clamp branch, whose out-of-range conversion the
    // reference's compiler folded at build time to INT64_MAX; exactly 2^63 is not clamped and converts at run time: INT64_MIN)
    double th = t > 0.0 ? t + 0.5 : t - 0.5;
    if (th > 9223372036854775808.0) return (T)0x7fffffffffffffffll;
    th = th < -9223372036854775808.0 ? -9223372036854775808.0 : th;
    return (T)x86_cvttsd2si(th);
  } else if constexpr (std::is_same<T, uint64_t>::value || std::is_same<T, unsigned long long>::value) {
    double th = t > 0.0 ? t + 0.5 : 0.0;
    if (th > 18446744073709551616.0) return (T)0xffffffffffffffffull;          // (as above: the clamp branch saturates, 2^64 itself stores 0)
    if (th < 9223372036854775808.0) return (T)(unsigned long long)x86_cvttsd2si(th);
    return (T)((unsigned long long)x86_cvttsd2si(th - 9223372036854775808.0) ^ 0x8000000000000000ull);
  } else if constexpr (std::is_unsigned<T>::value) {
    // scipy: t > 0 ? t + 0.5 : 0, clamped to the maximum, truncated.  v_cvt_u32_f64 truncates and saturates (negative
    // and NaN -> 0, >= 2^32 -> 0xffffffff), so one add, one conversion and one integer minimum say the same: for
    // t <= 0, t + 0.5 <= 0.5 truncates to 0; for t > 0 it is the same sum; anything above the maximum is clamped after.
    uint32_t r;
    const double th = t + 0.5;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(th));
    constexpr uint32_t hi = (uint32_t)std::numeric_limits<T>::max();
    return (T)(r < hi ? r : hi);
  } else {
    // scipy: rounded half away from zero (t > 0 ? t + 0.5 : t - 0.5), clamped, truncated.  v_cvt_i32_f64 truncates and
    // saturates to the int32 range, which contains every narrower type's bounds.
    int32_t r;
    const double th = t + (t > 0.0 ? 0.5 : -0.5);
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(th));
    constexpr int32_t lo = (int32_t)std::numeric_limits<T>::min(), hi = (int32_t)std::numeric_limits<T>::max();
    r = r > hi ? hi : r;
    r = r < lo ? lo : r;
    return (T)r;
  }
}

// ------------------------------------------------------------------ integer element types: the exact lerp (see unwarp_kernels.hip, exact_lerp_pairs)
constexpr float kExactLerpMinCoord = 32.0f;
// The same blend with the four taps read as NARROW elements straight from LDS (ds_read_u16 / _i16 / _u8 / _i8: naturally aligned,
// no pair extraction): 4 LDS reads instead of 2, and 6 VALU instructions fewer per pixel (no aligned-dword address, shift count,
// two alignbit, two masks) -- the integer kernels are bound by VALU issue (SQ_INSTS_VALU 51 per pixel at 85 % utilisation), not
// by LDS.  t_lo / t_hi: the tap pairs of rows y0 / y0 + 1.
template <typename T>
__device__ __forceinline__ double exact_lerp_taps(const T* t_lo, const T* t_hi, double fx, double fy) {
  const int a = (int)t_lo[0], b = (int)t_lo[1], c = (int)t_hi[0], d = (int)t_hi[1];
  const double tp = __builtin_fma(fx, (double)(b - a), (double)a);
  const double bt = __builtin_fma(fx, (double)(d - c), (double)c);
  return __builtin_fma(fy, bt - tp, tp);
}
// scipy's integer store (to_elem) of a value that is an EXACT convex combination of elements of T: it lies inside T's range, so
// the clamps cannot act and are left out (v_cvt_u32_f64 / v_cvt_i32_f64 truncate).
template <typename T>
__device__ __forceinline__ T to_elem_in_range(double t) {
  if constexpr (std::is_unsigned<T>::value) {
    uint32_t r;
    const double th = t + 0.5;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(th));
    return (T)r;
  } else {
    int32_t r;
    const double th = t + (t > 0.0 ? 0.5 : -0.5);
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(th));
    return (T)r;
  }
}

// runtime-typed access for the kernels where the element type is not worth a template parameter
__device__ __forceinline__ double load_any(const void* p, int dtype, size_t i) {
  switch (dtype) {
    case kF32: return (double)((const float*)p)[i];
    case kF64: return ((const double*)p)[i];
    case kU8: return (double)((const uint8_t*)p)[i];
    case kI8: return (double)((const int8_t*)p)[i];
    case kU16: return (double)((const uint16_t*)p)[i];
    case kI16: return (double)((const int16_t*)p)[i];
    case kU32: return (double)((const uint32_t*)p)[i];
    case kI64: return (double)((const int64_t*)p)[i];
    case kU64: return (double)((const uint64_t*)p)[i];
    case kBool: return (double)((const uint8_t*)p)[i];
    default: return (double)((const int32_t*)p)[i];
  }
}
#ifndef DCP_ANY_NT_STORE
#define DCP_ANY_NT_STORE 1   // store_any: nt stores -- every caller writes a final result that no kernel of the call reads again (0: plain, A/B)
#endif
template <typename E>
__device__ __forceinline__ void store_final(E* p, E v) {
#if DCP_ANY_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void store_any(void* p, int dtype, size_t i, double t) {
  switch (dtype) {
    case kF32: store_final((float*)p + i, to_elem<float>(t)); break;
    case kF64: store_final((double*)p + i, t); break;
    case kU8: store_final((uint8_t*)p + i, to_elem<uint8_t>(t)); break;
    case kI8: store_final((int8_t*)p + i, to_elem<int8_t>(t)); break;
    case kU16: store_final((uint16_t*)p + i, to_elem<uint16_t>(t)); break;
    case kI16: store_final((int16_t*)p + i, to_elem<int16_t>(t)); break;
    case kU32: store_final((uint32_t*)p + i, to_elem<uint32_t>(t)); break;
    case kI64: store_final((int64_t*)p + i, to_elem<int64_t>(t)); break;
    case kU64: store_final((uint64_t*)p + i, to_elem<uint64_t>(t)); break;
    case kBool: store_final((uint8_t*)p + i, to_elem<Bool8>(t).v); break;
    default: store_final((int32_t*)p + i, to_elem<int32_t>(t)); break;
  }
}

}  // namespace dcp
