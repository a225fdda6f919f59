// api_core.cpp -- library / device / option / memory / stream / event entry points of the C ABI
// (include/discorpy_hip.h) and the helpers every other api_*.cpp uses (api_common.h).
#include "api_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace dcpapi {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  // a HIP failure is reported HERE; the runtime's sticky "last error" must not resurface as the hipGetLastError() of the next,
  // unrelated launch
  if (code == DCP_ERR_HIP) (void)hipGetLastError();
  return code;
}

const char* last_error() { return g_err; }

std::atomic<int> g_tile_rows{16}, g_xcd_remap{2}, g_coef_lds{0}, g_d_chunk{16}, g_pipe_depth{2}, g_lds_gather{1}, g_stack_chunk_kb{24576}, g_stack_lds{1}, g_host_duplex{1}, g_host_bands{6}, g_tile_cert{1}, g_wg_box{1}, g_wg_per_cu{0}, g_stack_wg{1}, g_int_exact{1}, g_host_direct{1}, g_tall_tiles{0}, g_store_wait{1}, g_fused_wg{1}, g_any_order{0}, g_host_band_sync{0};

dcp::LaunchOpts current_opts() {
  dcp::LaunchOpts o;
  o.tile_rows = g_tile_rows.load();
  o.xcd_remap = g_xcd_remap.load();
  o.coef_lds = g_coef_lds.load();
  o.d_chunk = g_d_chunk.load();
  o.pipe_depth = g_pipe_depth.load();
  o.lds_gather = g_lds_gather.load();
  o.stack_lds = g_stack_lds.load();
  o.wg_box = g_wg_box.load();
  o.wg_per_cu = g_wg_per_cu.load();
  o.stack_wg = g_stack_wg.load();
  o.int_exact = g_int_exact.load();
  o.tall_tiles = g_tall_tiles.load();
  o.store_wait = g_store_wait.load();
  o.any_order = g_any_order.load();
  return o;
}

thread_local Staging g_staging;
thread_local HostStreams g_host_streams;

int sampler_of(int order, int blend_mode, int* sampler) {
  if (order == 0) {
    *sampler = dcp::kNearest;
    return DCP_OK;
  }
  if (order != 1)
    return fail(DCP_ERR_UNSUPPORTED, "spline order %d is not implemented on the GPU path (only 0 and 1)", order);
  switch (blend_mode) {
    case DCP_BLEND_SCIPY: *sampler = dcp::kScipy; return DCP_OK;
    case DCP_BLEND_F64LERP: *sampler = dcp::kF64Lerp; return DCP_OK;
    case DCP_BLEND_F32LERP: *sampler = dcp::kF32Lerp; return DCP_OK;
    default: return fail(DCP_ERR_INVALID_ARG, "unknown blend_mode %d", blend_mode);
  }
}

int check_image(const void* src, const void* dst, int64_t H, int64_t W, int64_t rs, int64_t cs) {
  if (!src || !dst) return fail(DCP_ERR_INVALID_ARG, "null image pointer");
  if (H <= 0 || W <= 0) return fail(DCP_ERR_INVALID_ARG, "image must be non-empty (got %lld x %lld)", (long long)H, (long long)W);
  if (cs < 1 || rs < 1) return fail(DCP_ERR_INVALID_ARG, "strides must be positive (row %lld, col %lld)", (long long)rs, (long long)cs);
  if (rs < (W - 1) * cs + 1 && H > 1) return fail(DCP_ERR_INVALID_ARG, "row stride %lld overlaps rows of width %lld", (long long)rs, (long long)W);
  // the gather addresses the source with 32-bit byte offsets
  const double extent = ((double)(H - 1) * (double)rs + (double)(W - 1) * (double)cs + 1.0) * 4.0;
  if (extent > 4294967040.0 || H > 2147483647LL / 2 || W > 2147483647LL / 2)
    return fail(DCP_ERR_UNSUPPORTED, "source extent %.0f bytes exceeds the 4 GiB the 32-bit gather offsets address", extent);
  return DCP_OK;
}

int fill_map(dcp::MapArgs* m, double xc, double yc, const double* fact, int nfact, const double* coef) {
  memset(m, 0, sizeof(*m));
  m->xc = xc;
  m->yc = yc;
  if (nfact < 0 || nfact > dcp::kMaxFact)
    return fail(DCP_ERR_INVALID_ARG, "nfact = %d outside [0, %d] (DCP_MAX_FACT is a limit of this library: the reference takes a coefficient list of any "
                "length, discorpy/post/postprocessing.py:142-143, and never ships more than 5)", nfact, dcp::kMaxFact);
  if (nfact > 0 && !fact) return fail(DCP_ERR_INVALID_ARG, "null coefficient pointer");
  for (int i = 0; i < nfact; ++i) m->fact[i] = fact[i];
  m->nfact = nfact;
  if (coef)
    for (int i = 0; i < 8; ++i) m->coef[i] = coef[i];
  return DCP_OK;
}

// The shared-reciprocal division of the perspective kernels (div2_rn) is exact while nothing
// leaves the normal range: every coefficient is 0 or of moderate magnitude, and the denominator
// c7*x + c8*y + 1 keeps one sign and a moderate magnitude over the whole image (checked at the four
// corners; it is affine in x and y).
int homography_is_tame(const double* c, int64_t H, int64_t W) {
  for (int i = 0; i < 8; ++i) {
    const double a = std::fabs(c[i]);
    if (!(a == 0.0 || (a > 1e-100 && a < 1e100))) return 0;
  }
  const double xs[2] = {0.0, (double)(W - 1)}, ys[2] = {0.0, (double)(H - 1)};
  double lo = 1e300, hi = -1e300;
  for (double x : xs)
    for (double y : ys) {
      const double d = (c[6] * x + c[7] * y) + 1.0;
      lo = d < lo ? d : lo;
      hi = d > hi ? d : hi;
    }
  if (!(lo > 0.0 || hi < 0.0)) return 0;
  const double m = std::fabs(lo) < std::fabs(hi) ? std::fabs(lo) : std::fabs(hi);
  const double M = std::fabs(lo) > std::fabs(hi) ? std::fabs(lo) : std::fabs(hi);
  return m > 1e-6 && M < 1e6;
}

// ---- tile deviation certificate (MapArgs::tile_dev_ok) ---------------------------------------------------------
// remap_lds_kernel predicts the source box of a 64 x 16 output tile from the tile's four corner pixels.  For a map
// f in C^{1,1} the bilinear interpolant of the corner values deviates from f inside the tile by at most
//     (w^2 sup|f_xx| + h^2 sup|f_yy|) / 8,   w = 63, h = 15  (pixel spans of the tile),
// and the interpolant itself lies inside the hull of the corner values.  When that bound is below one pixel, the
// corner hull grown by one pixel contains every tap of the tile and the kernel needs no per-pixel check.  The answer
// is a level: 1 = holds for 64 x 16 tiles, 2 = also for the 128 x 32 tiles of remap_wg_kernel (w = 127, h = 31).
//   radial map (x - xc) B(r):  |f_xx|, |f_yy| <= 4 |B'(r)| + r |B''(r)|   (the x and the y coordinate alike;
//     f_xx = 3 xu B'/r + xu^3 (B''/r^2 - B'/r^3), f_yy = xu B'/r + xu yu^2 (B''/r^2 - B'/r^3), |xu|, |yu| <= r).
//     The supremum over [0, rmax] (rmax: farthest frame corner from the centre) is taken on 1024 midpoints plus the
//     half-interval times a triangle-inequality bound of the derivative -- rigorous, and tight enough because the
//     intervals are short.
//   homography N(x, y) / D(x, y), N and D affine, D of one sign over the frame ("tame"):
//     f_xx = -2 c7 (N_x D - c7 N) / D^3, f_yy = -2 c8 (N_y D - c8 N) / D^3 with |N|, |D| bounded at the frame corners.
// The final clip to the frame and the float32 rounding are monotone, so they keep a coordinate inside the (clipped,
// rounded) hull +- the deviation; 0.95 px leaves room for the rounding (one float32 ulp of a coordinate < 2^24).
namespace {
constexpr double kTileDevLimit = 0.95;
constexpr double kWgBoxCols = 144.0, kWgBoxRows = 42.0;     // remap_wg_kernel's slab (unwarp_kernels.hip: kWgBoxW, kWgBoxH)
// (w^2, h^2) / 8 of the two tile shapes: the 64 x 16 wave tile of remap_lds_kernel, the 128 x 32 workgroup tile of remap_wg_kernel
constexpr double kSpanX2[2] = {63.0 * 63.0 / 8.0, 127.0 * 127.0 / 8.0}, kSpanY2[2] = {15.0 * 15.0 / 8.0, 31.0 * 31.0 / 8.0};

double radial_curvature_bound(const dcp::MapArgs& m, double rmax) {
  const int n = m.nfact;
  if (n <= 1) return 0.0;                      // B constant: the map is affine
  if (!(rmax >= 0.0) || !std::isfinite(rmax)) return INFINITY;
  // P1 = B' = sum i a_i r^(i-1),  P2 = r B'' = sum i (i-1) a_i r^(i-1); their derivatives bounded by the triangle inequality
  double l1 = 0.0, l2 = 0.0, rp = 1.0;         // rp = rmax^(i-2)
  for (int i = 2; i < n; ++i) {
    const double a = std::fabs(m.fact[i]);
    l1 += (double)i * (i - 1) * a * rp;
    l2 += (double)i * (i - 1) * (i - 1) * a * rp;
    rp *= rmax;
  }
  constexpr int kNodes = 1024;      // (the answer is cached per calibration: ~20 us once)
  const double h = rmax / kNodes;
  double sup = 0.0;
  for (int k = 0; k < kNodes; ++k) {
    const double r = (k + 0.5) * h;
    double p1 = 0.0, p2 = 0.0;
    for (int i = n - 1; i >= 1; --i) {
      p1 = p1 * r + (double)i * m.fact[i];
      p2 = p2 * r + (double)i * (i - 1) * m.fact[i];
    }
    const double v = 4.0 * std::fabs(p1) + std::fabs(p2);
    sup = v > sup ? v : sup;
  }
  sup += 0.5 * h * (4.0 * l1 + l2);
  return std::isfinite(sup) ? sup : INFINITY;
}

// Level 2 is only worth taking when (nearly) every 128 x 32 tile's source box fits remap_wg_kernel's slab: a tile
// that does not sends its whole workgroup to the direct gather, and with a fifth of the tiles doing so (the 9-term
// fisheye model of config 5, whose tiles are sheared) the per-wave kernel is the faster one.  The boxes of a 24 x 24
// lattice of tiles are formed exactly as the kernel forms them (corner hull, one pixel of margin, tap width).
bool wg_boxes_mostly_fit(int kind, const dcp::MapArgs& m, int64_t H, int64_t W, int tw = 128, int th = 32, double box_cols = kWgBoxCols,
                         double box_rows = kWgBoxRows) {
  const int64_t ntx = (W + tw - 1) / tw, nty = (H + th - 1) / th;
  const int sx = (int)std::min<int64_t>(ntx, 24), sy = (int)std::min<int64_t>(nty, 24);
  int bad = 0;
  for (int iy = 0; iy < sy; ++iy)
    for (int ix = 0; ix < sx; ++ix) {
      const int64_t tx = sx > 1 ? (int64_t)ix * (ntx - 1) / (sx - 1) : 0, ty = sy > 1 ? (int64_t)iy * (nty - 1) / (sy - 1) : 0;
      double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300;
      for (int cy = 0; cy < 2; ++cy)
        for (int cx = 0; cx < 2; ++cx) {
          const double X = (double)std::min<int64_t>(tx * tw + cx * (tw - 1), W - 1), Y = (double)std::min<int64_t>(ty * th + cy * (th - 1), H - 1);
          double xd, yd;
          auto radial = [&](double px, double py) {
            const double xu = px - m.xc, yu = py - m.yc, r = std::hypot(xu, yu);
            double b = 0.0;
            for (int i = m.nfact - 1; i >= 0; --i) b = b * r + m.fact[i];
            xd = m.xc + b * xu;
            yd = m.yc + b * yu;
          };
          auto persp = [&](double px, double py) {
            const double den = (m.coef[6] * px + m.coef[7] * py) + 1.0;
            xd = ((m.coef[0] * px + m.coef[1] * py) + m.coef[2]) / den;
            yd = ((m.coef[3] * px + m.coef[4] * py) + m.coef[5]) / den;
          };
          if (kind == dcp::kRadial) {
            radial(X, Y);
          } else if (kind == dcp::kPersp) {
            persp(X, Y);
          } else {
            // fused, as wg_corner_tap forms the box: the radial map at the corners of the bounding box of the tile's four (clipped)
            // perspective positions
            const double x0 = (double)(tx * tw), x1 = (double)std::min<int64_t>(tx * tw + tw - 1, W - 1);
            const double y0 = (double)(ty * th), y1 = (double)std::min<int64_t>(ty * th + th - 1, H - 1);
            double qx0 = 1e300, qx1 = -1e300, qy0 = 1e300, qy1 = -1e300;
            for (double px : {x0, x1})
              for (double py : {y0, y1}) {
                persp(px, py);
                if (!(xd == xd) || !(yd == yd)) return false;
                xd = std::min(std::max(xd, 0.0), (double)(W - 1));
                yd = std::min(std::max(yd, 0.0), (double)(H - 1));
                qx0 = std::min(qx0, xd), qx1 = std::max(qx1, xd), qy0 = std::min(qy0, yd), qy1 = std::max(qy1, yd);
              }
            radial(cx ? qx1 : qx0, cy ? qy1 : qy0);
          }
          xd = std::min(std::max(xd, 0.0), (double)(W - 1));
          yd = std::min(std::max(yd, 0.0), (double)(H - 1));
          if (!(xd == xd) || !(yd == yd)) return false;
          xlo = std::min(xlo, std::floor(xd));
          xhi = std::max(xhi, std::floor(xd));
          ylo = std::min(ylo, std::floor(yd));
          yhi = std::max(yhi, std::floor(yd));
        }
      if (xhi - xlo + 4.0 > box_cols || yhi - ylo + 4.0 > box_rows) ++bad;
    }
  return bad * 50 <= sx * sy;          // at most 2 % of the sampled tiles
}

// the same calibrations are applied to frame after frame (one per camera / channel / grid-search candidate): keep the
// answers of the last kCertSlots distinct ones, replaced round-robin
constexpr int kCertSlots = 256;     // (a batch call certifies every frame's calibration: four launches' worth of distinct ones stay cached)
struct CertEntry {
  int kind = -1, nfact = -1;
  int64_t H = 0, W = 0;
  double xc = 0, yc = 0, fact[dcp::kMaxFact], coef[8];
  int ok = 0, tall = 0;
};
thread_local CertEntry g_cert_cache[kCertSlots];
thread_local int g_cert_next = 0, g_cert_last = 0;
}  // namespace

int tile_deviation_certified(int kind, const dcp::MapArgs& m, int64_t H, int64_t W, int* tall_ok) {
  if (tall_ok) *tall_ok = 0;
  if (H < 1 || W < 1) return 0;
  auto same = [&](const CertEntry& c) {
    return c.kind == kind && c.nfact == m.nfact && c.H == H && c.W == W && c.xc == m.xc && c.yc == m.yc &&
           memcmp(c.fact, m.fact, sizeof(double) * (size_t)(m.nfact > 0 ? m.nfact : 0)) == 0 && memcmp(c.coef, m.coef, sizeof(c.coef)) == 0;
  };
  if (same(g_cert_cache[g_cert_last])) {      // the common case: the calibration of the previous call
    if (tall_ok) *tall_ok = g_cert_cache[g_cert_last].tall;
    return g_cert_cache[g_cert_last].ok;
  }
  for (int i = 0; i < kCertSlots; ++i)
    if (same(g_cert_cache[i])) {
      g_cert_last = i;
      if (tall_ok) *tall_ok = g_cert_cache[i].tall;
      return g_cert_cache[i].ok;
    }
  CertEntry& c = g_cert_cache[g_cert_next];
  g_cert_last = g_cert_next;
  g_cert_next = (g_cert_next + 1) % kCertSlots;
  int ok = 0, tall = 0;
  if (kind == dcp::kRadial) {
    double rmax = 0.0;
    for (double x : {0.0, (double)(W - 1)})
      for (double y : {0.0, (double)(H - 1)}) rmax = std::max(rmax, std::hypot(x - m.xc, y - m.yc));
    const double k2 = radial_curvature_bound(m, rmax * (1.0 + 1e-12) + 1e-9);
    for (int lvl = 0; lvl < 2; ++lvl)
      if ((kSpanX2[lvl] + kSpanY2[lvl]) * k2 <= kTileDevLimit) ok = lvl + 1;
    // the 64 x 32 tiles of remap_wg_color_kernel's second shape (sheared maps): the same bound for that span, its boxes under 80 x 56
    tall = (63.0 * 63.0 / 8.0 + 31.0 * 31.0 / 8.0) * k2 <= kTileDevLimit && wg_boxes_mostly_fit(kind, m, H, W, 64, 32, 80.0, 56.0);
  } else if (kind == dcp::kPersp) {
    if (homography_is_tame(m.coef, H, W)) {
      // N / D with N, D affine: d/dx = (N_x D - D_x N) / D^2 and d2/dx2 = -2 D_x (N_x D - D_x N) / D^3.  The numerators
      // are affine in (x, y) too, so over the frame their magnitude peaks at a corner, where |D| is smallest as well.
      double dmin = 1e300;
      double gx[4] = {0, 0, 0, 0};      // max |N_t D - D_t N| for (N, t) = (Nx, x), (Nx, y), (Ny, x), (Ny, y)
      for (double x : {0.0, (double)(W - 1)})
        for (double y : {0.0, (double)(H - 1)}) {
          const double d = (m.coef[6] * x + m.coef[7] * y) + 1.0;
          const double nx = (m.coef[0] * x + m.coef[1] * y) + m.coef[2], ny = (m.coef[3] * x + m.coef[4] * y) + m.coef[5];
          dmin = std::min(dmin, std::fabs(d));
          gx[0] = std::max(gx[0], std::fabs(m.coef[0] * d - m.coef[6] * nx));
          gx[1] = std::max(gx[1], std::fabs(m.coef[1] * d - m.coef[7] * nx));
          gx[2] = std::max(gx[2], std::fabs(m.coef[3] * d - m.coef[6] * ny));
          gx[3] = std::max(gx[3], std::fabs(m.coef[4] * d - m.coef[7] * ny));
        }
      const double d2 = dmin * dmin, d3 = d2 * dmin, c7 = std::fabs(m.coef[6]), c8 = std::fabs(m.coef[7]);
      const double xxx = 2.0 * c7 * gx[0] / d3, xyy = 2.0 * c8 * gx[1] / d3, yxx = 2.0 * c7 * gx[2] / d3, yyy = 2.0 * c8 * gx[3] / d3;
      for (int lvl = 0; lvl < 2; ++lvl) {
        const double devx = kSpanX2[lvl] * xxx + kSpanY2[lvl] * xyy, devy = kSpanX2[lvl] * yxx + kSpanY2[lvl] * yyy;
        if (std::isfinite(devx) && std::isfinite(devy) && devx <= kTileDevLimit && devy <= kTileDevLimit) ok = lvl + 1;
      }
    }
  }
  else if (kind == dcp::kFused) {
    // R o f32clip o P.  The tile's perspective positions lie in the bounding box Q of its corners' positions (P is projective with a
    // denominator of one sign: convex image; clip and rounding are monotone), |Q| <= (127 |dxp/dx| + 31 |dxp/dy|, 127 |dyp/dx| +
    // 31 |dyp/dy|) with the derivative bounds of the perspective branch; the radial map deviates from the bilinear interpolant of
    // its values at Q's corners by at most (qw^2 + qh^2) / 8 * sup(4 |B'| + r |B''|) over the radii of the frame (the positions are
    // clipped into it).  Level 2 only: an uncertified fused map keeps the per-wave kernel with its per-pixel vote.
    if (homography_is_tame(m.coef, H, W)) {
      double dmin = 1e300, gx[4] = {0, 0, 0, 0};
      for (double x : {0.0, (double)(W - 1)})
        for (double y : {0.0, (double)(H - 1)}) {
          const double d = (m.coef[6] * x + m.coef[7] * y) + 1.0;
          const double nx = (m.coef[0] * x + m.coef[1] * y) + m.coef[2], ny = (m.coef[3] * x + m.coef[4] * y) + m.coef[5];
          dmin = std::min(dmin, std::fabs(d));
          gx[0] = std::max(gx[0], std::fabs(m.coef[0] * d - m.coef[6] * nx));
          gx[1] = std::max(gx[1], std::fabs(m.coef[1] * d - m.coef[7] * nx));
          gx[2] = std::max(gx[2], std::fabs(m.coef[3] * d - m.coef[6] * ny));
          gx[3] = std::max(gx[3], std::fabs(m.coef[4] * d - m.coef[7] * ny));
        }
      const double d2 = dmin * dmin;
      // + the float32 rounding of the positions at both ends of the box: two half-ulps at the largest coordinate of the frame (2^-24
      // relative each), never less than the 1e-3 px of frames below 8192 px (a fixed 1e-3 stopped being a bound above that)
      const double slack = std::max(1e-3, 2.0 * (double)std::max(W, H) * 0x1p-23);
      const double qw = (127.0 * gx[0] + 31.0 * gx[1]) / d2 + slack, qh = (127.0 * gx[2] + 31.0 * gx[3]) / d2 + slack;
      double rmax = 0.0;
      for (double x : {0.0, (double)(W - 1)})
        for (double y : {0.0, (double)(H - 1)}) rmax = std::max(rmax, std::hypot(x - m.xc, y - m.yc));
      const double k2 = radial_curvature_bound(m, rmax * (1.0 + 1e-12) + 1e-9);
      const double dev = (qw * qw + qh * qh) / 8.0 * k2;
      if (std::isfinite(dev) && dev <= kTileDevLimit) ok = 2;          // (the A/B switch x_fused_wg is applied by the caller, outside this cache)
    }
  }
  if (ok == 2 && !wg_boxes_mostly_fit(kind, m, H, W)) ok = kind == dcp::kFused ? 0 : 1;
  c.kind = kind;
  c.nfact = m.nfact;
  c.H = H;
  c.W = W;
  c.xc = m.xc;
  c.yc = m.yc;
  memcpy(c.fact, m.fact, sizeof(c.fact));
  memcpy(c.coef, m.coef, sizeof(c.coef));
  c.ok = ok;
  c.tall = tall;
  if (tall_ok) *tall_ok = tall;
  return ok;
}

// d/dyu [B(r) yu] = B + B' yu^2 / r >= B - |B'| r: positive on [0, rmax] => every column's row coordinate increases with
// the row.  1024 samples of B - |B'| r and the same Lipschitz slack as radial_curvature_bound; a folding model fails.
bool radial_monotone_in_y(const dcp::MapArgs& m, int64_t H, int64_t W) {
  const int n = m.nfact;
  if (n <= 0 || H < 1 || W < 1) return false;
  // (the same calibration is applied to chunk after chunk: keep the last answer)
  thread_local struct {
    int nfact = -1;
    int64_t H = 0, W = 0;
    double xc = 0, yc = 0, fact[dcp::kMaxFact];
    bool ok = false;
  } cache;
  if (cache.nfact == n && cache.H == H && cache.W == W && cache.xc == m.xc && cache.yc == m.yc &&
      memcmp(cache.fact, m.fact, sizeof(double) * (size_t)n) == 0)
    return cache.ok;
  auto remember = [&](bool ok) {
    cache.nfact = n;
    cache.H = H;
    cache.W = W;
    cache.xc = m.xc;
    cache.yc = m.yc;
    memcpy(cache.fact, m.fact, sizeof(double) * (size_t)n);
    cache.ok = ok;
    return ok;
  };
  double rmax = 0.0;
  for (double x : {0.0, (double)(W - 1)})
    for (double y : {0.0, (double)(H - 1)}) rmax = std::max(rmax, std::hypot(x - m.xc, y - m.yc));
  rmax = rmax * (1.0 + 1e-12) + 1e-9;
  if (!std::isfinite(rmax)) return remember(false);
  double lip = 0.0, rp = 1.0;                  // |d/dr (B - |B'| r)| <= |B'| + |B'| + |B''| r <= sum (2 i + i (i - 1)) |a_i| rmax^(i-1)
  for (int i = 1; i < n; ++i) {
    lip += (double)(2 * i + i * (i - 1)) * std::fabs(m.fact[i]) * rp;
    rp *= rmax;
  }
  constexpr int kNodes = 1024;
  const double h = rmax / kNodes;
  double lo = INFINITY;
  for (int k = 0; k < kNodes; ++k) {
    const double r = (k + 0.5) * h;
    double b = 0.0, b1 = 0.0;
    for (int i = n - 1; i >= 0; --i) b = b * r + m.fact[i];
    for (int i = n - 1; i >= 1; --i) b1 = b1 * r + (double)i * m.fact[i];
    lo = std::min(lo, b - std::fabs(b1) * r);
  }
  lo -= 0.5 * h * lip;
  return remember(std::isfinite(lo) && lo > 0.0);
}

// yd_min = int16(floor(amin(yd_list1))), yd_max = int16(ceil(amax(yd_list2))) + 1 of postprocessing.py:289-301, the two
// lists being the clipped float64 row coordinates of the chunk's first and last rows, evaluated as numpy evaluates
// them: flist = sum_i a_i * ru**i accumulated from i = 0 (ru**0 = 1, ru**1 = ru, ru**2 = ru * ru, higher powers by pow()).
void reference_chunk_band(const dcp::MapArgs& m, int64_t H, int64_t W, double row_first, double row_last, int64_t* b0, int64_t* b1) {
  auto row_extreme = [&](double row, bool want_min) {
    const double yu = row - m.yc;
    double ext = want_min ? INFINITY : -INFINITY;
    for (int64_t x = 0; x < W; ++x) {
      const double xu = (double)x - m.xc;
      const double ru = std::sqrt(xu * xu + yu * yu);
      double fl = 0.0;
      for (int i = 0; i < m.nfact; ++i) {
        const double p = i == 0 ? 1.0 : i == 1 ? ru : i == 2 ? ru * ru : std::pow(ru, (double)i);
        const double t = m.fact[i] * p;
        fl = i == 0 ? t : fl + t;
      }
      double yd = m.yc + fl * yu;
      yd = yd < 0.0 ? 0.0 : (yd > (double)(H - 1) ? (double)(H - 1) : yd);
      ext = want_min ? std::fmin(ext, yd) : std::fmax(ext, yd);
    }
    return ext;
  };
  const double lo = std::floor(row_extreme(row_first, true)), hi = std::ceil(row_extreme(row_last, false)) + 1.0;
  *b0 = std::isfinite(lo) ? (int64_t)lo : 0;
  *b1 = std::isfinite(hi) ? std::min<int64_t>((int64_t)hi, H) : H;     // (a Python slice stops at the array's end)
  if (*b0 < 0) *b0 = 0;
}

uint32_t extent_bytes(int64_t H, int64_t W, int64_t rs, int64_t cs) {
  return (uint32_t)(((H - 1) * rs + (W - 1) * cs + 1) * 4);
}

// Rows of a projection that the radial map of output rows row_start .. row_start+nrows-1 can
// touch: [*b0, *b1).  yd = yc + yu * B(r) is bilinear in (yu, B), so its range over the rows is spanned
// by the corners of [yu_min, yu_max] x [B_min, B_max], with B's range taken over every radius the rows
// reach.  B is sampled every 1/4 pixel of radius (a few thousand evaluations, instead of one per output
// pixel) and the range is widened by twice the largest step between neighbouring samples; the hull is
// then grown by a safety row on each side.  A non-finite model gets the whole projection.
void host_row_band(const dcp::MapArgs& m, int64_t H, int64_t W, double row_start, int64_t nrows, int64_t* b0,
                   int64_t* b1) {
  host_row_band_rect(m, H, 0.0, (double)(W - 1), row_start, row_start + (double)(nrows - 1), b0, b1);
}

// The same for the radial map evaluated at any position of the rectangle [x_lo, x_hi] x [y_lo, y_hi] (the fused map
// evaluates it at perspective-corrected positions rather than on the pixel grid).
void host_row_band_rect(const dcp::MapArgs& m, int64_t H, double x_lo, double x_hi, double y_lo, double y_hi, int64_t* b0,
                        int64_t* b1) {
  *b0 = 0;
  *b1 = H;
  const int n = m.nfact, ne = (n + 1) / 2, no = n / 2;
  auto B = [&](double ru) {
    if (n <= 0) return 0.0;
    const double r2 = ru * ru;
    double E = m.fact[2 * (ne - 1)];
    for (int k = ne - 2; k >= 0; --k) E = E * r2 + m.fact[2 * k];
    if (no == 0) return E;
    double O = m.fact[2 * (no - 1) + 1];
    for (int k = no - 2; k >= 0; --k) O = O * r2 + m.fact[2 * k + 1];
    return ru * O + E;
  };
  const double yu0 = y_lo - m.yc, yu1 = y_hi - m.yc;
  const double ya = std::fmin(std::fabs(yu0), std::fabs(yu1));
  const double ay_min = (yu0 <= 0.0 && yu1 >= 0.0) ? 0.0 : ya;                    // smallest |yu| over the rows
  const double ay_max = std::fmax(std::fabs(yu0), std::fabs(yu1));
  const double xl = x_lo - m.xc, xr = x_hi - m.xc;
  const double ax_min = (xl <= 0.0 && xr >= 0.0) ? 0.0 : std::fmin(std::fabs(xl), std::fabs(xr));
  const double ax_max = std::fmax(std::fabs(xl), std::fabs(xr));
  const double rlo = std::sqrt(ax_min * ax_min + ay_min * ay_min), rhi = std::sqrt(ax_max * ax_max + ay_max * ay_max);
  if (!std::isfinite(rlo) || !std::isfinite(rhi) || rhi > 1e9) return;
  // a sample every quarter pixel of radius, but never more than ~1e5 of them (a centre far outside the frame would
  // otherwise cost seconds per call): a coarser step only widens the 2 * step safety margin below
  const double dr = std::fmax(0.25, (rhi - rlo) / 1.0e5);
  const int64_t ns = (int64_t)std::ceil((rhi - rlo) / dr) + 1;
  double bmin = 1e300, bmax = -1e300, step = 0.0, prev = 0.0;
  for (int64_t i = 0; i <= ns; ++i) {
    const double r = i == ns ? rhi : rlo + dr * (double)i;
    const double v = B(r < rhi ? r : rhi);
    if (!std::isfinite(v)) return;
    if (i > 0) step = std::fmax(step, std::fabs(v - prev));
    prev = v;
    bmin = std::fmin(bmin, v);
    bmax = std::fmax(bmax, v);
  }
  bmin -= 2.0 * step;
  bmax += 2.0 * step;
  double ymin = 1e300, ymax = -1e300;
  for (double yu : {yu0, yu1})
    for (double bv : {bmin, bmax}) {
      double yd = m.yc + yu * bv;
      if (!(yd >= 0.0)) yd = 0.0;               // also catches NaN
      if (yd > (double)(H - 1)) yd = (double)(H - 1);
      ymin = std::fmin(ymin, yd);
      ymax = std::fmax(ymax, yd);
    }
  int64_t lo = (int64_t)std::floor(ymin) - 1, hi = (int64_t)std::floor(ymax) + 3;
  if (lo < 0) lo = 0;
  if (hi > H) hi = H;
  if (hi - lo < 2) {  // the gather needs two rows
    lo = lo > 0 ? lo - 1 : lo;
    hi = lo + 2 <= H ? (hi > lo + 2 ? hi : lo + 2) : H;
  }
  *b0 = lo;
  *b1 = hi;
}

int check_image_typed(const void* src, const void* dst, int dtype, int64_t H, int64_t W, int64_t rs, int64_t cs) {
  if (dtype < 0 || dtype >= dcp::kNumElemTypes) return fail(DCP_ERR_INVALID_ARG, "unknown element type %d", dtype);
  if (!src || !dst) return fail(DCP_ERR_INVALID_ARG, "null image pointer");
  if (H <= 0 || W <= 0) return fail(DCP_ERR_INVALID_ARG, "image must be non-empty (got %lld x %lld)", (long long)H, (long long)W);
  if (cs < 1 || rs < 1) return fail(DCP_ERR_INVALID_ARG, "strides must be positive (row %lld, col %lld)", (long long)rs, (long long)cs);
  if (rs < (W - 1) * cs + 1 && H > 1) return fail(DCP_ERR_INVALID_ARG, "row stride %lld overlaps rows of width %lld", (long long)rs, (long long)W);
  if (H > 1073741823LL || W > 1073741823LL) return fail(DCP_ERR_UNSUPPORTED, "image too large");
  return DCP_OK;
}

size_t extent_bytes_typed(int64_t H, int64_t W, int64_t rs, int64_t cs, int dtype) {
  return (size_t)((H - 1) * rs + (W - 1) * cs + 1) * (size_t)dcp::elem_size(dtype);
}

}  // namespace dcpapi

using namespace dcpapi;

extern "C" {

int dcp_version(void) { return 100; }

int dcp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* dcp_last_error(void) { return dcpapi::last_error(); }

// The seven documented options are spelt as they are; every other knob is a switch of the measurement lab (A/B runs, parity campaigns,
// tests that force a kernel) and answers only to its name with an "x_" prefix -- undocumented in the header, free to change.
static const char* const kStableOptions[] = {"stack_chunk_kb", "host_duplex", "host_bands", "host_direct", "host_direct_applies", "tile_cert",
                                             "lds_gather"};
// The unprefixed spellings the round-4 header documented for what are now lab switches: accepted for one more release,
// with one warning per process on stderr -- a caller written against that header keeps working and is told what to change.
static const char* const kDeprecatedAliases[] = {"tile_rows", "xcd_remap", "coef_lds", "d_chunk", "pipe_depth", "stack_lds", "wg_box", "wg_per_cu",
                                                 "stack_wg", "spline_tiled", "spline_wg", "int_exact", "box_table", "tall_tiles", "store_wait"};
static const char* option_name(const char* key) {
  bool stable = false;
  const bool lab = !strncmp(key, "x_", 2);
  const char* name = lab ? key + 2 : key;
  for (const char* k : kStableOptions) stable = stable || !strcmp(name, k);
  if (!lab && !stable) {
    for (const char* k : kDeprecatedAliases)
      if (!strcmp(name, k)) {
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))
          fprintf(stderr, "libdiscorpy_hip: option \"%s\" is a lab switch now -- spell it \"x_%s\" (the unprefixed name is accepted until the next release)\n", name, name);
        return name;
      }
  }
  return lab == stable ? "" : name;          // a stable key with the prefix, or an unknown lab key without it: unknown
}

int dcp_set_option(const char* key_in, int value) {
  if (!key_in) return fail(DCP_ERR_INVALID_ARG, "null option key");
  const char* key = option_name(key_in);
  if (!strcmp(key, "tile_rows")) {
    if (value < 1 || value > dcp::kMaxTileRows) return fail(DCP_ERR_INVALID_ARG, "tile_rows must be in [1, %d]", dcp::kMaxTileRows);
    g_tile_rows = value;
  } else if (!strcmp(key, "xcd_remap")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "xcd_remap must be 0, 1 or 2");
    g_xcd_remap = value;
  } else if (!strcmp(key, "coef_lds")) {
    g_coef_lds = value ? 1 : 0;
  } else if (!strcmp(key, "lds_gather")) {
    g_lds_gather = value ? 1 : 0;
  } else if (!strcmp(key, "pipe_depth")) {
    if (value != 1 && value != 2 && value != 4) return fail(DCP_ERR_INVALID_ARG, "pipe_depth must be 1, 2 or 4");
    g_pipe_depth = value;
  } else if (!strcmp(key, "d_chunk")) {
    if (value < 1) return fail(DCP_ERR_INVALID_ARG, "d_chunk must be >= 1");
    g_d_chunk = value;
  } else if (!strcmp(key, "stack_lds")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "stack_lds must be 0, 1 or 2");
    g_stack_lds = value;
  } else if (!strcmp(key, "host_duplex")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "host_duplex must be 0, 1 or 2");
    g_host_duplex = value;
  } else if (!strcmp(key, "host_bands")) {
    if (value < 1 || value > 256) return fail(DCP_ERR_INVALID_ARG, "host_bands must be in [1, 256]");
    g_host_bands = value;
  } else if (!strcmp(key, "stack_wg")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "stack_wg must be 0, 1 or 2");
    g_stack_wg = value;               // 0: round 1's stack kernels; 1: stack_wg_kernel when the launch is large enough; 2: whenever eligible
  } else if (!strcmp(key, "wg_per_cu")) {
    if (value < 0 || value > 6) return fail(DCP_ERR_INVALID_ARG, "wg_per_cu must be in [0, 6]");
    g_wg_per_cu = value;
  } else if (!strcmp(key, "host_direct")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "host_direct must be 0, 1 or 2");
    g_host_direct = value;            // 0: a host frame's result is always staged on the device and copied back; 1: written straight into a
                                      // registered destination when the runtime cannot overlap an upload with a download; 2: whenever registered
  } else if (!strcmp(key, "host_band_sync")) {
    g_host_band_sync = value ? 1 : 0; // 1: the banded host path synchronises its upload stream after every band's kernel (rounds 1-5) instead of handing the downloader an event
  } else if (!strcmp(key, "any_order")) {
    g_any_order = value ? 1 : 0;      // 1: EVERY whole-frame device launch as with DCP_MEM_DEVICE_UNORDERED (A/B runs; the per-call flag is the interface)
  } else if (!strcmp(key, "store_wait")) {
    g_store_wait = value ? 1 : 0;     // 0: stack_wg_kernel waits for its own stores at every projection (rounds 2-3), A/B
  } else if (!strcmp(key, "fused_wg")) {
    g_fused_wg = value ? 1 : 0;       // 0: the fused perspective o radial map always on the per-wave kernel with the per-pixel vote (rounds 1-4), A/B
                                      // (read where the certificate is USED -- dcp_unwarp_fused_f32 --, not where it is cached: every thread sees a change)
  } else if (!strcmp(key, "tall_tiles")) {
    g_tall_tiles = value < 0 ? 0 : (value > 2 ? 2 : value);     // 1: sheared radial maps (level-1 certificate, boxes of 64 x 32 tiles fit 80 x 56) on 64 x 32 workgroup tiles instead of the
                                      // per-wave-box kernel.  Default 0: measured SLOWER on BASELINE config 5 (128-131 us against 113-116, tools/time_cfg5.py)
  } else if (!strcmp(key, "int_exact")) {
    g_int_exact = value ? 1 : 0;      // 0: integer element types blend in scipy's operation order everywhere (A/B and parity runs)
  } else if (!strcmp(key, "wg_box")) {
    g_wg_box = value ? 1 : 0;         // 0: one source box per wave tile (remap_lds_kernel) even when the certificate covers 128 x 32 tiles
  } else if (!strcmp(key, "box_table")) {
    if (value < 0 || value > 2) return fail(DCP_ERR_INVALID_ARG, "box_table must be 0, 1 or 2");
    dcp::set_box_table(value);        // 0: every wave of remap_wg_kernel evaluates its tile's corners; 1: once per tile by box_table_kernel where it pays
  } else if (!strcmp(key, "spline_wg")) {
    dcp::set_spline_wg(value ? 1 : 0);
  } else if (!strcmp(key, "spline_tiled")) {
    dcp::set_spline_tiled(value < 0 ? 0 : (value > 6 ? 6 : value));
  } else if (!strcmp(key, "pf2d_chunk")) {
    dcp::set_pf2d_chunk(value < 0 ? 0 : value);
  } else if (!strcmp(key, "pf2d_xcd")) {
    dcp::set_pf2d_xcd(value ? 1 : 0);
  } else if (!strcmp(key, "pf2d_two_pole")) {
    dcp::set_pf2d_two_pole(value ? 1 : 0);
  } else if (!strcmp(key, "spline_xcd")) {
    dcp::set_spline_xcd(value ? 1 : 0);
  } else if (!strcmp(key, "tile_cert")) {
    g_tile_cert = value ? 1 : 0;      // 0: never use the host's tile-deviation certificate (remap_lds_kernel then votes)
  } else if (!strcmp(key, "stack_chunk_kb")) {
    if (value < 1) return fail(DCP_ERR_INVALID_ARG, "stack_chunk_kb must be >= 1");
    g_stack_chunk_kb = value;
  } else {
    return fail(DCP_ERR_INVALID_ARG, "unknown option '%s'", key_in);
  }
  return DCP_OK;
}

int dcp_get_option(const char* key_in, int* value) {
  if (!key_in || !value) return fail(DCP_ERR_INVALID_ARG, "null argument");
  const char* key = option_name(key_in);
  if (!strcmp(key, "tile_rows")) *value = g_tile_rows;
  else if (!strcmp(key, "xcd_remap")) *value = g_xcd_remap;
  else if (!strcmp(key, "any_order")) *value = g_any_order;
  else if (!strcmp(key, "host_band_sync")) *value = g_host_band_sync;
  else if (!strcmp(key, "coef_lds")) *value = g_coef_lds;
  else if (!strcmp(key, "d_chunk")) *value = g_d_chunk;
  else if (!strcmp(key, "pipe_depth")) *value = g_pipe_depth;
  else if (!strcmp(key, "lds_gather")) *value = g_lds_gather;
  else if (!strcmp(key, "stack_chunk_kb")) *value = g_stack_chunk_kb;
  else if (!strcmp(key, "stack_lds")) *value = g_stack_lds;
  else if (!strcmp(key, "host_duplex")) *value = g_host_duplex;
  else if (!strcmp(key, "host_bands")) *value = g_host_bands;
  else if (!strcmp(key, "tile_cert")) *value = g_tile_cert;
  else if (!strcmp(key, "wg_box")) *value = g_wg_box;
  else if (!strcmp(key, "wg_per_cu")) *value = g_wg_per_cu;
  else if (!strcmp(key, "spline_tiled")) *value = dcp::get_spline_tiled();
  else if (!strcmp(key, "pf2d_chunk")) *value = dcp::get_pf2d_chunk();
  else if (!strcmp(key, "pf2d_xcd")) *value = dcp::get_pf2d_xcd();
  else if (!strcmp(key, "pf2d_two_pole")) *value = dcp::get_pf2d_two_pole();
  else if (!strcmp(key, "spline_xcd")) *value = dcp::get_spline_xcd();
  else if (!strcmp(key, "spline_wg")) *value = dcp::get_spline_wg();
  else if (!strcmp(key, "box_table")) *value = dcp::get_box_table();
  else if (!strcmp(key, "stack_wg")) *value = g_stack_wg;
  else if (!strcmp(key, "int_exact")) *value = g_int_exact;
  else if (!strcmp(key, "host_direct")) *value = g_host_direct;
  else if (!strcmp(key, "tall_tiles")) *value = g_tall_tiles;
  else if (!strcmp(key, "fused_wg")) *value = g_fused_wg;
  else if (!strcmp(key, "store_wait")) *value = g_store_wait;
  else if (!strcmp(key, "host_direct_applies")) {        // read-only: measures the runtime once (needs a device)
    int n = 0;
    // (the one-off probe moves 32 MiB each way: on the device host frames will go to -- DISCORPY_AMD_DEVICE, as the Python front end
    // reads it -- rather than on whatever device happens to be current)
    int dev = -1;
    if (const char* e = getenv("DISCORPY_AMD_DEVICE")) dev = atoi(e);
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      *value = 0;
    } else {
      DeviceScope scope(dev >= 0 && dev < n ? dev : -1);
      *value = host_direct_applies() ? 1 : 0;
    }
  }
  else return fail(DCP_ERR_INVALID_ARG, "unknown option '%s'", key_in);
  return DCP_OK;
}

int dcp_release_scratch(void) {
  // the calling thread's staging buffers and streams (DCP_MEM_HOST calls), and the spline planes of every device
  g_staging.release();
  g_host_streams.release();
  return release_spline_workspace();
}

int dcp_debug_counters(uint64_t* out, int n, int reset) {
  if (!out || n < 2) return fail(DCP_ERR_INVALID_ARG, "need room for 2 counters");
  unsigned long long v[2];
  DCP_HIP(hipDeviceSynchronize());
  DCP_HIP(dcp::read_lds_stats(v, reset != 0));
  out[0] = v[0];
  out[1] = v[1];
  return DCP_OK;
}

int dcp_debug_bounds(uint64_t* out, int n, int reset) {
  if (!out || n < 5) return fail(DCP_ERR_INVALID_ARG, "need room for 5 values");
#ifdef DCP_DEBUG_BOUNDS
  out[4] = 1;
#else
  out[4] = 0;
#endif
  out[0] = out[1] = out[2] = out[3] = 0;
  DCP_HIP(hipDeviceSynchronize());
  hipError_t (*readers[3])(unsigned long long*, bool) = {dcp::read_bounds_unwarp, dcp::read_bounds_color, dcp::read_bounds_spline};
  for (auto rd : readers) {
    unsigned long long v[4];
    DCP_HIP(rd(v, reset != 0));
    if (v[0] && !out[0]) {
      out[1] = v[1];
      out[2] = v[2];
      out[3] = v[3];
    }
    out[0] += v[0];
  }
  return DCP_OK;
}

const char* dcp_debug_last_kernel(void) { return dcp::last_kernel_name(); }

int dcp_debug_tile_certificate(int map_kind, int64_t height, int64_t width, double xcenter, double ycenter, const double* list_fact,
                               int nfact, const double* list_coef) {
  if (map_kind != DCP_MAP_RADIAL && map_kind != DCP_MAP_PERSPECTIVE && map_kind != DCP_MAP_FUSED)
    return fail(DCP_ERR_INVALID_ARG, "map_kind must be radial, perspective or fused");
  if (map_kind != DCP_MAP_RADIAL && !list_coef) return fail(DCP_ERR_INVALID_ARG, "null homography pointer");
  dcp::MapArgs map;
  int rc;
  if ((rc = fill_map(&map, xcenter, ycenter, map_kind != DCP_MAP_PERSPECTIVE ? list_fact : nullptr, map_kind != DCP_MAP_PERSPECTIVE ? nfact : 0,
                     map_kind != DCP_MAP_RADIAL ? list_coef : nullptr)) != DCP_OK)
    return rc;
  return tile_deviation_certified(map_kind == DCP_MAP_RADIAL ? dcp::kRadial : map_kind == DCP_MAP_PERSPECTIVE ? dcp::kPersp : dcp::kFused, map, height,
                                  width);
}

int dcp_malloc(void** ptr, size_t bytes, int device) {
  if (!ptr) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  DCP_HIP(hipMalloc(ptr, bytes ? bytes : 4));
  return DCP_OK;
}

int dcp_free(void* ptr, int device) {
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  DCP_HIP(hipFree(ptr));
  return DCP_OK;
}

int dcp_memcpy(void* dst, const void* src, size_t bytes, int kind, int device, void* stream) {
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipMemcpyKind k;
  switch (kind) {
    case DCP_COPY_H2D: k = hipMemcpyHostToDevice; break;
    case DCP_COPY_D2H: k = hipMemcpyDeviceToHost; break;
    case DCP_COPY_D2D: k = hipMemcpyDeviceToDevice; break;
    default: return fail(DCP_ERR_INVALID_ARG, "unknown copy kind %d", kind);
  }
  DCP_HIP(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
  if (kind != DCP_COPY_D2D) DCP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return DCP_OK;
}

int dcp_host_register(void* ptr, size_t bytes, int device) {
  if (!ptr || bytes == 0) return fail(DCP_ERR_INVALID_ARG, "nothing to register");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  DCP_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return DCP_OK;
}

int dcp_host_unregister(void* ptr) {
  if (!ptr) return DCP_OK;
  DCP_HIP(hipHostUnregister(ptr));
  return DCP_OK;
}

int dcp_stream_create(void** stream, int device) {
  if (!stream) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipStream_t s;
  DCP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return DCP_OK;
}

int dcp_stream_destroy(void* stream) {
  if (stream) DCP_HIP(hipStreamDestroy((hipStream_t)stream));
  return DCP_OK;
}

int dcp_stream_synchronize(int device, void* stream) {
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  DCP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return DCP_OK;
}

int dcp_event_create(void** event, int device) {
  if (!event) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  DeviceScope scope(device);
  if (scope.status != hipSuccess) return fail(DCP_ERR_HIP, "cannot select device %d: %s", device, hipGetErrorString(scope.status));
  hipEvent_t e;
  DCP_HIP(hipEventCreate(&e));
  *event = (void*)e;
  return DCP_OK;
}

int dcp_event_record(void* event, void* stream) {
  DCP_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return DCP_OK;
}

int dcp_stream_wait_event(void* stream, void* event) {
  if (!event) return fail(DCP_ERR_INVALID_ARG, "null event");
  DCP_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return DCP_OK;
}

int dcp_event_synchronize(void* event) {
  DCP_HIP(hipEventSynchronize((hipEvent_t)event));
  return DCP_OK;
}

int dcp_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) return fail(DCP_ERR_INVALID_ARG, "null out pointer");
  DCP_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return DCP_OK;
}

int dcp_event_destroy(void* event) {
  DCP_HIP(hipEventDestroy((hipEvent_t)event));
  return DCP_OK;
}

}  // extern "C"
