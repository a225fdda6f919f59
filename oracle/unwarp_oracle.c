/*
 * oracle/unwarp_oracle.c -- CPU restatement of discorpy's backward-unwarp path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under discorpy_amd/ may import, link or
 * call this file; it exists so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check (never replace) the HIP kernels.
 *
 * Parity status: PINNED.  tools/gen_golden.py imports the reference from
 * /root/reference in the build container and stores its inputs/outputs under
 * tests/golden/; tests/test_oracle_golden.py holds this file bit-equal to
 * those vectors (poly_mode = NUMPY, blend_mode = SCIPY).
 *
 * What is restated (reference = /root/reference, discorpy 1.7.0):
 *   discorpy/post/postprocessing.py:137-148   unwarp_image_backward
 *   discorpy/post/postprocessing.py:211-229   unwarp_slice_backward
 *   discorpy/post/postprocessing.py:281-313   unwarp_chunk_slices_backward (+ _mapping :250-252)
 *   discorpy/post/postprocessing.py:448-459   _generate_perspective_map
 *   discorpy/post/postprocessing.py:486-492   correct_perspective_image
 * and the third-party inner loop those call, scipy.ndimage.map_coordinates
 * (scipy is un-vendored and unpinned in the reference's setup.py:5-12; 1.15.3
 * in the build container).  Its published algorithm for spline order <= 1 --
 * C function NI_GeometricTransform + get_spline_interpolation_weights -- is:
 *   order 1: s = floor(c); f = c - s; w0 = 1 - f; w1 = 1 - w0 (per axis);
 *            taps (y0,x0),(y0,x1),(y1,x0),(y1,x1); each term is
 *            ((double)v * wy) * wx; terms accumulated left to right from 0.0
 *            in a double; the sum is cast to the output dtype.
 *   order 0: index = floor(c + 0.5) per axis; value copied.
 *   an out-of-range neighbour (only when c == len-1 exactly) is folded back
 *   onto the edge sample by the boundary mode and carries weight 0.
 * Because the callers clip every coordinate into [0, len-1] first, the
 * `mode` argument cannot change the result for order <= 1.
 *
 * Two evaluation orders of the radial polynomial are provided:
 *   ORC_POLY_NUMPY  (1): exactly the reference's expression
 *        sum_i a_i * ru**i, left to right, ru**0 = 1, ru**1 = ru,
 *        ru**2 = ru*ru, ru**i = pow(ru, i) for i >= 3
 *        (postprocessing.py:142-143; numpy's scalar-power fast paths).
 *   ORC_POLY_KERNEL (0): the order the HIP kernels use -- even/odd split in
 *        r2 = xu^2 + yu^2 with fused multiply-adds:
 *        E = a0 + r2*(a2 + r2*(a4 + ...)), O = a1 + r2*(a3 + ...),
 *        B = fma(ru, O, E); xd = fma(B, xu, xc), yd = fma(B, yu, yc).
 *   ORC_POLY_KERNEL_MULADD (2): kernel polynomial, but xc + B*xu as a separate
 *        multiply and add (kept for the flip-rate comparison in tools/flip_rate.py).
 *   Both agree to ~2e-16 relative; after the float32 rounding of the
 *   coordinates (postprocessing.py:144-145) they are bit-identical except when
 *   a float64 coordinate lies within ~1e-12 px of a float32 rounding boundary.
 *
 * Three blend modes: ORC_BLEND_SCIPY (0) is scipy's arithmetic above,
 * ORC_BLEND_F64LERP (1) and ORC_BLEND_F32LERP (2) restate the cheaper
 * factorised forms the HIP kernels may be asked to use.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include "unwarp_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ---- radial polynomial ------------------------------------------------- */

/* postprocessing.py:142-143 -- reference order */
static inline double poly_numpy(const double *a, int n, double ru)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double p;
        if (i == 0) p = 1.0;
        else if (i == 1) p = ru;
        else if (i == 2) p = ru * ru;
        else p = pow(ru, (double)i);
        double t = a[i] * p;
        s = (i == 0) ? t : s + t;
    }
    return s;
}

/* kernel order: even/odd Horner in r2, fused */
static inline double poly_kernel(const double *a, int n, double r2, double ru)
{
    if (n <= 0) return 0.0;
    int ne = (n + 1) / 2, no = n / 2;
    double E = a[2 * (ne - 1)];
    for (int k = ne - 2; k >= 0; --k) E = fma(r2, E, a[2 * k]);
    if (no == 0) return E;
    double O = a[2 * (no - 1) + 1];
    for (int k = no - 2; k >= 0; --k) O = fma(r2, O, a[2 * k + 1]);
    return fma(ru, O, E);
}

static inline double clipd(double v, double lo, double hi)
{
    /* np.clip == minimum(maximum(v, lo), hi) */
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

/* one output pixel of the radial backward map; postprocessing.py:138-145 */
static inline void radial_coord(double x, double y, double xc, double yc,
                                const double *a, int n, int poly_mode,
                                double wmax, double hmax, int round_f32,
                                double *xd, double *yd)
{
    double xu = x - xc;
    double yu = y - yc;
    double r2 = xu * xu + yu * yu;          /* xu**2 + yu**2: two products, one add */
    double ru = sqrt(r2);
    double f = poly_mode == ORC_POLY_NUMPY ? poly_numpy(a, n, ru)
                                           : poly_kernel(a, n, r2, ru);
    double sx, sy;
    if (poly_mode == ORC_POLY_KERNEL) {
        sx = fma(f, xu, xc);                 /* kernel order: one rounding instead of two */
        sy = fma(f, yu, yc);
    } else {
        double px = f * xu;
        double py = f * yu;
        sx = xc + px;
        sy = yc + py;
    }
    double cx = clipd(sx, 0.0, wmax);
    double cy = clipd(sy, 0.0, hmax);
    if (round_f32) {
        cx = (double)(float)cx;              /* np.float32(...) then widened by scipy */
        cy = (double)(float)cy;
    }
    *xd = cx;
    *yd = cy;
}

/* _generate_perspective_map; postprocessing.py:448-457 (numpy: no contraction) */
static inline void persp_coord(double x, double y, const double *c,
                               double wmax, double hmax, int round_f32,
                               double *xd, double *yd)
{
    double den = (c[6] * x + c[7] * y) + 1.0;
    double nx = (c[0] * x + c[1] * y) + c[2];
    double ny = (c[3] * x + c[4] * y) + c[5];
    double cx = clipd(nx / den, 0.0, wmax);
    double cy = clipd(ny / den, 0.0, hmax);
    if (round_f32) {
        cx = (double)(float)cx;
        cy = (double)(float)cy;
    }
    *xd = cx;
    *yd = cy;
}

/* ---- scipy.ndimage.map_coordinates, order 0 / 1, coordinate inside [0,len-1] ---- */

static inline int64_t fold_edge(int64_t i, int64_t len)
{
    /* every boundary mode maps index len -> a sample whose weight is 0 here;
       'reflect' (d c b a | a b c d | d c b a) gives len-1 */
    if (i < 0) return 0;
    if (i > len - 1) return len - 1;
    return i;
}

static inline float sample(const float *src, int64_t H, int64_t W,
                           int64_t rs, int64_t cs, double y, double x,
                           int order, int blend_mode)
{
    if (order == 0) {
        int64_t iy = fold_edge((int64_t)floor(y + 0.5), H);
        int64_t ix = fold_edge((int64_t)floor(x + 0.5), W);
        return src[iy * rs + ix * cs];
    }
    double y0 = floor(y), x0 = floor(x);
    double fy = y - y0, fx = x - x0;
    int64_t iy0 = (int64_t)y0, ix0 = (int64_t)x0;
    int64_t iy1 = fold_edge(iy0 + 1, H), ix1 = fold_edge(ix0 + 1, W);
    iy0 = fold_edge(iy0, H);
    ix0 = fold_edge(ix0, W);
    if (blend_mode != ORC_BLEND_SCIPY) {
        /* the factorised forms are defined on the base tap the HIP gather uses: x0 <= W-2,
           y0 <= H-2, so that at the far edge the fraction is 1 on (len-2, len-1) rather than
           0 on (len-1, folded len-1) */
        if (W >= 2 && ix0 > W - 2) { ix0 = W - 2; ix1 = W - 1; fx = x - (double)ix0; }
        if (H >= 2 && iy0 > H - 2) { iy0 = H - 2; iy1 = H - 1; fy = y - (double)iy0; }
    }
    float v00 = src[iy0 * rs + ix0 * cs], v01 = src[iy0 * rs + ix1 * cs];
    float v10 = src[iy1 * rs + ix0 * cs], v11 = src[iy1 * rs + ix1 * cs];
    if (blend_mode == ORC_BLEND_SCIPY) {
        double wy0 = 1.0 - fy, wy1 = 1.0 - wy0;
        double wx0 = 1.0 - fx, wx1 = 1.0 - wx0;
        double t = 0.0;
        t += ((double)v00 * wy0) * wx0;
        t += ((double)v01 * wy0) * wx1;
        t += ((double)v10 * wy1) * wx0;
        t += ((double)v11 * wy1) * wx1;
        return (float)t;
    } else if (blend_mode == ORC_BLEND_F64LERP) {
        double a = (double)v00, b = (double)v01, c = (double)v10, d = (double)v11;
        double top = fma(fx, b - a, a);
        double bot = fma(fx, d - c, c);
        return (float)fma(fy, bot - top, top);
    } else {
        float gx = (float)fx, gy = (float)fy;
        float top = fmaf(gx, v01 - v00, v00);
        float bot = fmaf(gx, v11 - v10, v10);
        return fmaf(gy, bot - top, top);
    }
}

/* ---- public entry points ----------------------------------------------- */

int orc_radial_coords(int64_t H, int64_t W, double xc, double yc,
                      const double *fact, int nfact, int poly_mode,
                      int round_f32, double *yd, double *xd)
{
    if (H < 0 || W < 0 || nfact < 0) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x)
            radial_coord((double)x, (double)y, xc, yc, fact, nfact, poly_mode,
                         (double)(W - 1), (double)(H - 1), round_f32,
                         &xd[y * W + x], &yd[y * W + x]);
    return 0;
}

/* the same for nrows output rows row_start, row_start + 1, ... of an (H, W) frame -- row_start any real number, as the
   slice function's `index` (not validated by the reference, postprocessing.py:215) */
int orc_radial_coords_rows(int64_t H, int64_t W, double xc, double yc, const double *fact, int nfact, int poly_mode,
                           int round_f32, double row_start, int64_t nrows, double *yd, double *xd)
{
    if (H < 0 || W < 0 || nfact < 0 || nrows < 0) return -1;
    for (int64_t r = 0; r < nrows; ++r)
        for (int64_t x = 0; x < W; ++x)
            radial_coord((double)x, row_start + (double)r, xc, yc, fact, nfact, poly_mode,
                         (double)(W - 1), (double)(H - 1), round_f32, &xd[r * W + x], &yd[r * W + x]);
    return 0;
}

int orc_perspective_coords(int64_t H, int64_t W, const double *coef,
                           int round_f32, double *yd, double *xd)
{
    if (H < 0 || W < 0) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x)
            persp_coord((double)x, (double)y, coef, (double)(W - 1),
                        (double)(H - 1), round_f32, &xd[y * W + x], &yd[y * W + x]);
    return 0;
}

/* unwarp_image_backward; postprocessing.py:137-148 */
int orc_unwarp_image_f32(const float *src, float *dst, int64_t H, int64_t W,
                         int64_t src_row_stride, double xc, double yc,
                         const double *fact, int nfact, int order,
                         int coord_round_f32, int poly_mode, int blend_mode)
{
    if (H <= 0 || W <= 0 || nfact < 0 || order < 0 || order > 1) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t y = 0; y < H; ++y) {
        for (int64_t x = 0; x < W; ++x) {
            double xd, yd;
            radial_coord((double)x, (double)y, xc, yc, fact, nfact, poly_mode,
                         (double)(W - 1), (double)(H - 1), coord_round_f32, &xd, &yd);
            dst[y * W + x] = sample(src, H, W, src_row_stride, 1, yd, xd, order, blend_mode);
        }
    }
    return 0;
}

/* correct_perspective_image with map_index=None; postprocessing.py:486-492 */
int orc_perspective_image_f32(const float *src, float *dst, int64_t H, int64_t W,
                              int64_t src_row_stride, const double *coef,
                              int order, int blend_mode)
{
    if (H <= 0 || W <= 0 || order < 0 || order > 1) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t y = 0; y < H; ++y) {
        for (int64_t x = 0; x < W; ++x) {
            double xd, yd;
            persp_coord((double)x, (double)y, coef, (double)(W - 1), (double)(H - 1), 1, &xd, &yd);
            dst[y * W + x] = sample(src, H, W, src_row_stride, 1, yd, xd, order, blend_mode);
        }
    }
    return 0;
}

/*
 * Fused perspective -> radial map, ONE resampling (BASELINE config 3; definition
 * SURVEY.md section 8(d) cfg3): (xp,yp) = float32(clip(Hmg(x,y))) as
 * postprocessing.py:453-457, then the radial map of :141-145 evaluated at the
 * (non-integer) position (xp,yp), float32-rounded, one sample of src.
 * Its reference value is one map_coordinates call at these coordinates.
 */
int orc_unwarp_fused_f32(const float *src, float *dst, int64_t H, int64_t W,
                         int64_t src_row_stride, double xc, double yc,
                         const double *fact, int nfact, const double *coef,
                         int order, int poly_mode, int blend_mode)
{
    if (H <= 0 || W <= 0 || nfact < 0 || order < 0 || order > 1) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t y = 0; y < H; ++y) {
        for (int64_t x = 0; x < W; ++x) {
            double xp, yp, xd, yd;
            persp_coord((double)x, (double)y, coef, (double)(W - 1), (double)(H - 1), 1, &xp, &yp);
            radial_coord(xp, yp, xc, yc, fact, nfact, poly_mode,
                         (double)(W - 1), (double)(H - 1), 1, &xd, &yd);
            dst[y * W + x] = sample(src, H, W, src_row_stride, 1, yd, xd, order, blend_mode);
        }
    }
    return 0;
}

/* ---- scipy.ndimage.map_coordinates, order 0 / 1, a coordinate OUTSIDE [0, len-1] ----
   The reference hands a caller's map_index and `mode` straight to scipy (postprocessing.py:489-491).  scipy's
   ni_interpolation.c first moves such a coordinate into the extended image (map_coordinate()), the taps that still fall
   outside fold by the same mode, and the two constant modes read cval = 0 there.  Restated here from scipy's documented
   behaviour and pinned against scipy itself (tests/test_oracle_golden.py, every mode, orders 0 and 1) and against the
   reference's own call (golden G16).  mode: index in ORC mode order (0 reflect, 1 grid-mirror, 2 constant,
   3 grid-constant, 4 nearest, 5 mirror, 6 grid-wrap, 7 wrap). */
static double mc_map_coordinate(double in, int64_t len, int mode)
{
    const double n = (double)len;
    if (in < 0.0) {
        switch (mode) {
        case 5: {
            if (len <= 1) return 0.0;
            const double sz2 = 2.0 * n - 2.0;
            in = sz2 * (double)(int64_t)(-in / sz2) + in;
            return in <= 1.0 - n ? in + sz2 : -in;
        }
        case 0: case 1: {
            if (len <= 1) return 0.0;
            const double sz2 = 2.0 * n;
            if (in < -sz2) in = sz2 * (double)(int64_t)(-in / sz2) + in;
            return in < -n ? in + sz2 : (in > -1e-15 ? 1e-15 : -in) - 1.0;
        }
        case 7: {
            if (len <= 1) return 0.0;
            const double sz = n - 1.0;
            return in + sz * ((double)(int64_t)(-in / sz) + 1.0);
        }
        case 6: {
            if (len <= 1) return 0.0;
            return in + n * ((double)(int64_t)((-1.0 - in) / n) + 1.0);
        }
        case 4: return 0.0;
        case 2: return -1.0;
        default: return in;
        }
    }
    if (in > n - 1.0) {
        switch (mode) {
        case 5: {
            if (len <= 1) return 0.0;
            const double sz2 = 2.0 * n - 2.0;
            in -= sz2 * (double)(int64_t)(in / sz2);
            return in >= n ? sz2 - in : in;
        }
        case 0: case 1: {
            if (len <= 1) return 0.0;
            const double sz2 = 2.0 * n;
            in -= sz2 * (double)(int64_t)(in / sz2);
            return in >= n ? sz2 - in - 1.0 : in;
        }
        case 7: {
            if (len <= 1) return 0.0;
            const double sz = n - 1.0;
            return in - sz * (double)(int64_t)(in / sz);
        }
        case 6: {
            if (len <= 1) return 0.0;
            return in - n * (double)(int64_t)((in + 1.0) / n);
        }
        case 4: return n - 1.0;
        case 2: return -1.0;
        default: return in;
        }
    }
    return in;
}

static int64_t mc_fold_tap(int64_t i, int64_t len, int mode)
{
    if (i >= 0 && i < len) return i;
    switch (mode) {
    case 0: case 1: {
        const int64_t s2 = 2 * len;
        i %= s2;
        if (i < 0) i += s2;
        return i < len ? i : s2 - 1 - i;
    }
    case 6:
        i %= len;
        return i < 0 ? i + len : i;
    case 4: return i < 0 ? 0 : len - 1;
    case 5: {
        if (len == 1) return 0;
        const int64_t s2 = 2 * len - 2;
        i %= s2;
        if (i < 0) i += s2;
        return i < len ? i : s2 - i;
    }
    case 7: {
        if (len == 1) return 0;
        const int64_t s = len - 1;
        i %= s;
        return i < 0 ? i + s : i;
    }
    default: return -1;          /* constant / grid-constant: cval */
    }
}

/* element (row, col) as a double, or cval = 0 for a tap outside */
static double mc_tap(const void *src, int dtype, int64_t rs, int64_t iy, int64_t ix);

static double mc_sample_outside(const void *src, int dtype, int64_t H, int64_t W, int64_t rs, double yc, double xc,
                                int order, int mode)
{
    const double y = mc_map_coordinate(yc, H, mode), x = mc_map_coordinate(xc, W, mode);
    if (mode == 2 && (y <= -1.0 || x <= -1.0)) return 0.0;
    if (order == 0)
        return mc_tap(src, dtype, rs, mc_fold_tap((int64_t)floor(y + 0.5), H, mode), mc_fold_tap((int64_t)floor(x + 0.5), W, mode));
    const double y0 = floor(y), x0 = floor(x);
    const double wy0 = 1.0 - (y - y0), wy1 = 1.0 - wy0;
    const double wx0 = 1.0 - (x - x0), wx1 = 1.0 - wx0;
    const int64_t iy0 = mc_fold_tap((int64_t)y0, H, mode), iy1 = mc_fold_tap((int64_t)y0 + 1, H, mode);
    const int64_t ix0 = mc_fold_tap((int64_t)x0, W, mode), ix1 = mc_fold_tap((int64_t)x0 + 1, W, mode);
    double t = 0.0;
    t += (mc_tap(src, dtype, rs, iy0, ix0) * wy0) * wx0;
    t += (mc_tap(src, dtype, rs, iy0, ix1) * wy0) * wx1;
    t += (mc_tap(src, dtype, rs, iy1, ix0) * wy1) * wx0;
    t += (mc_tap(src, dtype, rs, iy1, ix1) * wy1) * wx1;
    return t;
}

/* map_coordinates(mat, (ycoord, xcoord)) with caller-supplied coordinates:
   correct_perspective_image(map_index=...) :489-491 and _mapping :250-251.
   Coordinates outside [0,len-1]: scipy's `mode` (mode 4 = 'nearest' clamps them). */
int orc_remap_coords_f32(const float *src, float *dst, int64_t H, int64_t W,
                         int64_t src_row_stride, const void *ycoord,
                         const void *xcoord, int coord_is_f64, int64_t npts,
                         int order, int blend_mode, int mode)
{
    if (H <= 0 || W <= 0 || npts < 0 || order < 0 || order > 1 || mode < 0 || mode > 7) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < npts; ++i) {
        double y = coord_is_f64 ? ((const double *)ycoord)[i] : (double)((const float *)ycoord)[i];
        double x = coord_is_f64 ? ((const double *)xcoord)[i] : (double)((const float *)xcoord)[i];
        if (mode != 4 && (y < 0.0 || y > (double)(H - 1) || x < 0.0 || x > (double)(W - 1))) {
            dst[i] = (float)mc_sample_outside(src, ORC_DT_F32, H, W, src_row_stride, y, x, order, mode);
            continue;
        }
        y = clipd(y, 0.0, (double)(H - 1));
        x = clipd(x, 0.0, (double)(W - 1));
        dst[i] = sample(src, H, W, src_row_stride, 1, y, x, order, blend_mode);
    }
    return 0;
}

/*
 * unwarp_slice_backward (:211-229, coord_round_f32 = 0, nrows = 1) and
 * unwarp_chunk_slices_backward (:281-313, coord_round_f32 = 1) over a
 * (D,H,W) stack: out[d, r, x] for rows row_start .. row_start+nrows-1.
 * The reference samples a row band mat3D[i, yd_min:yd_max, :] with
 * band-relative coordinates; subtracting the integer yd_min is exact, so for
 * a coordinate inside the band the result equals sampling the whole
 * projection at the absolute coordinate.  The chunk function takes the band
 * from its first row's minimum and its last row's maximum (:289-301): under a
 * folding model a row in between can leave it, and scipy then reflects the
 * coordinate inside the CROPPED array (mode='reflect', :308-312) -- restated
 * below (golden G15).  The slice function's band comes from the row itself.
 * row_start may be any double-representable number (the reference does not
 * validate `index`, :215); rows are row_start + r.
 */
/* [*b0, *b1): yd_min, yd_max of postprocessing.py:289-301 for the rows row_first .. row_last of a chunk (the reference's
   arithmetic: float64, numpy's polynomial order, clipped, not rounded to float32) */
void orc_chunk_band(int64_t H, int64_t W, double xc, double yc, const double *fact, int nfact, double row_first,
                    double row_last, int64_t *b0, int64_t *b1)
{
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t x = 0; x < W; ++x) {
        double xd, yd;
        radial_coord((double)x, row_first, xc, yc, fact, nfact, ORC_POLY_NUMPY, (double)(W - 1), (double)(H - 1), 0, &xd, &yd);
        lo = fmin(lo, yd);
        radial_coord((double)x, row_last, xc, yc, fact, nfact, ORC_POLY_NUMPY, (double)(W - 1), (double)(H - 1), 0, &xd, &yd);
        hi = fmax(hi, yd);
    }
    *b0 = (int64_t)floor(lo);
    *b1 = (int64_t)ceil(hi) + 1;
    if (*b0 < 0) *b0 = 0;
    if (*b1 > H) *b1 = H;            /* a Python slice stops at the array's end */
}

int orc_unwarp_stack_rows_f32(const float *vol, float *out, int64_t D, int64_t H,
                              int64_t W, double xc, double yc, const double *fact,
                              int nfact, double row_start, int64_t nrows,
                              int coord_round_f32, int poly_mode, int blend_mode)
{
    if (D < 0 || H <= 0 || W <= 0 || nrows < 0 || nfact < 0) return -1;
    int64_t b0 = 0, b1 = H;
    if (coord_round_f32 && nrows > 0) {
        orc_chunk_band(H, W, xc, yc, fact, nfact, row_start, row_start + (double)(nrows - 1), &b0, &b1);
        if (b1 <= b0) return -1;
    }
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t r = 0; r < nrows; ++r) {
        for (int64_t x = 0; x < W; ++x) {
            double xd, yd;
            radial_coord((double)x, row_start + (double)r, xc, yc, fact, nfact, poly_mode,
                         (double)(W - 1), (double)(H - 1), coord_round_f32, &xd, &yd);
            const int outside = coord_round_f32 && (yd < (double)b0 || yd > (double)(b1 - 1));
            for (int64_t d = 0; d < D; ++d)
                out[(d * nrows + r) * W + x] = outside
                    /* (yd_mat - yd_min is a float32 subtraction in the reference, :305-307: exact inside the band, rounded
                       when the difference is larger in magnitude than the coordinate) */
                    ? (float)mc_sample_outside(vol + d * H * W + b0 * W, ORC_DT_F32, b1 - b0, W, W, (double)((float)yd - (float)b0), xd, 1, ORC_MODE_REFLECT)
                    : sample(vol + d * H * W, H, W, W, 1, yd, xd, 1, blend_mode);
        }
    }
    return 0;
}

/* ======================================================================================
 * Spline orders 2..5: scipy.ndimage.map_coordinates with prefilter.
 *
 * Published algorithm (scipy ndimage: ni_splines.c / ni_interpolation.c, after Thevenaz et al.):
 *  1. modes 'nearest' and 'grid-constant' pad the input by 12 samples (edge values / zeros);
 *  2. the (padded) image is turned into B-spline coefficients, float64, by a separable recursive
 *     filter, axis 0 then axis 1: gain prod (1-z)(1-1/z) over the poles z of the order, then per
 *     pole a causal and an anti-causal pass with EXACT initial values for the boundary extension
 *     'reflect' (half-sample symmetric; modes reflect, grid-mirror), periodic (grid-wrap) or
 *     'mirror' (whole-sample symmetric; every other mode);
 *  3. a point is the (order+1)^2-tap sum  t += (c * wy) * wx,  taps row-major from
 *     floor(c) - order/2 (odd orders) or floor(c + 0.5) - order/2 (even orders), with the
 *     centred B-spline weights; taps outside the array fold back by the boundary mode
 *     ('constant' and 'wrap' fold like 'mirror' for in-range coordinates).
 * Pinned against scipy 1.15.3 for all 8 modes x 4 orders by tests/golden/g11_* (float32-equal).
 * The reference reaches this path through order=3 in examples/readthedocs_demo/demo_07.py:60.
 * ====================================================================================== */

static int spline_poles(int order, double *z)
{
    switch (order) {
    case 2: z[0] = sqrt(8.0) - 3.0; return 1;
    case 3: z[0] = sqrt(3.0) - 2.0; return 1;
    case 4: z[0] = sqrt(664.0 - sqrt(438976.0)) + sqrt(304.0) - 19.0;
            z[1] = sqrt(664.0 + sqrt(438976.0)) - sqrt(304.0) - 19.0; return 2;
    case 5: z[0] = sqrt(67.5 - sqrt(4436.25)) + sqrt(26.25) - 6.5;
            z[1] = sqrt(67.5 + sqrt(4436.25)) - sqrt(26.25) - 6.5; return 2;
    default: return 0;
    }
}

/* filter kinds */
#define SPL_MIRROR 0
#define SPL_REFLECT 1
#define SPL_WRAP 2

static int spline_filter_kind(int mode)
{
    /* ('nearest' prefilters its edge-padded array with the reflect boundary too: scipy.ndimage.spline_filter1d(x,
       mode='nearest') == (x, mode='reflect') to the last bit; invisible inside the image -- twelve samples of constant
       padding away -- but it is what coordinates outside the image see) */
    if (mode == ORC_MODE_REFLECT || mode == ORC_MODE_GRID_MIRROR || mode == ORC_MODE_NEAREST) return SPL_REFLECT;
    if (mode == ORC_MODE_GRID_WRAP) return SPL_WRAP;
    return SPL_MIRROR;
}

/* The exact initial sums run over the whole line in scipy; their terms decay like |z|^i with
   |z| <= 0.431, so everything past SPL_HORIZON terms is below 4e-24 of the leading ones and cannot be
   seen in float64.  The sums stop there (here and in spline_kernels.hip), which bounds the serial
   part of a line. */
#define SPL_HORIZON 64

/* in-place recursive filter of n samples with stride s */
static void spline_filter_line(double *c, int64_t n, int64_t s, const double *poles, int npoles, int kind)
{
    if (n < 2) return;
    double lam = 1.0;
    for (int p = 0; p < npoles; ++p) lam *= (1.0 - poles[p]) * (1.0 - 1.0 / poles[p]);
    for (int64_t i = 0; i < n; ++i) c[i * s] *= lam;
    for (int p = 0; p < npoles; ++p) {
        const double z = poles[p];
        if (kind == SPL_REFLECT) {
            double z_i = z;
            const double z_n = pow(z, (double)n);
            const double c0 = c[0];
            double acc = c[0] + z_n * c[(n - 1) * s];
            const int64_t m = n - 1 < SPL_HORIZON ? n - 1 : SPL_HORIZON;
            for (int64_t i = 1; i <= m; ++i) {
                /* scipy accumulates this sum IN c[0] (ni_splines.c _init_causal_reflect), so its last term, i = n - 1,
                   reads the partial sum where the formula wants sample 0 -- a z^(2n-1) effect, invisible from a dozen
                   samples on, 6e-3 for a 2-sample line.  The reference's results contain it; so do ours. */
                const double far = (n - 1 - i == 0) ? acc : c[(n - 1 - i) * s];
                acc += z_i * (c[i * s] + z_n * far);
                z_i *= z;
            }
            c[0] = acc * z / (1.0 - z_i * z_i) + c0;
            for (int64_t i = 1; i < n; ++i) c[i * s] += z * c[(i - 1) * s];
            c[(n - 1) * s] *= z / (z - 1.0);
            for (int64_t i = n - 2; i >= 0; --i) c[i * s] = z * (c[(i + 1) * s] - c[i * s]);
        } else if (kind == SPL_MIRROR) {
            double z_i = z;
            const double z_n_1 = pow(z, (double)(n - 1));
            double acc = c[0] + z_n_1 * c[(n - 1) * s];
            const int64_t m = n - 2 < SPL_HORIZON ? n - 2 : SPL_HORIZON;
            for (int64_t i = 1; i <= m; ++i) {
                acc += z_i * (c[i * s] + z_n_1 * c[(n - 1 - i) * s]);
                z_i *= z;
            }
            c[0] = acc / (1.0 - z_n_1 * z_n_1);
            for (int64_t i = 1; i < n; ++i) c[i * s] += z * c[(i - 1) * s];
            c[(n - 1) * s] = (z / (z * z - 1.0)) * (c[(n - 1) * s] + z * c[(n - 2) * s]);
            for (int64_t i = n - 2; i >= 0; --i) c[i * s] = z * (c[(i + 1) * s] - c[i * s]);
        } else {
            double z_i = z, acc = c[0];
            const int64_t m = n - 1 < SPL_HORIZON ? n - 1 : SPL_HORIZON;
            for (int64_t k = 0; k < m; ++k) {
                acc += z_i * c[(n - 1 - k) * s];
                z_i *= z;
            }
            c[0] = acc / (1.0 - z_i);
            for (int64_t i = 1; i < n; ++i) c[i * s] += z * c[(i - 1) * s];
            z_i = z;
            acc = c[(n - 1) * s];
            for (int64_t i = 0; i < m; ++i) {
                acc += z_i * c[i * s];
                z_i *= z;
            }
            c[(n - 1) * s] = acc * z / (z_i - 1.0);
            for (int64_t i = n - 2; i >= 0; --i) c[i * s] = z * (c[(i + 1) * s] - c[i * s]);
        }
    }
}

int orc_spline_pad(int mode) { return (mode == ORC_MODE_NEAREST || mode == ORC_MODE_GRID_CONSTANT) ? 12 : 0; }

/* ---- element types other than float32 ("output dtype = input dtype") ----
 * scipy reads every input element as a double (ni_interpolation.c CASE_INTERP / the line buffers of
 * spline_filter) and converts the double result on the way out (CASE_INTERP_OUT*):
 *   float32 / float64: C cast;
 *   unsigned: t = t > 0 ? t + 0.5 : 0, clamped to the type's maximum, truncated;
 *   signed:   t = t > 0 ? t + 0.5 : t - 0.5 (round half away from zero), clamped, truncated.
 * Pinned by tests/golden/g12_* (reference outputs for uint8 .. float64 inputs). */
static inline double load_typed(const void *src, int dtype, int64_t i)
{
    switch (dtype) {
    case ORC_DT_F32: return (double)((const float *)src)[i];
    case ORC_DT_F64: return ((const double *)src)[i];
    case ORC_DT_U8: return (double)((const uint8_t *)src)[i];
    case ORC_DT_I8: return (double)((const int8_t *)src)[i];
    case ORC_DT_U16: return (double)((const uint16_t *)src)[i];
    case ORC_DT_I16: return (double)((const int16_t *)src)[i];
    case ORC_DT_U32: return (double)((const uint32_t *)src)[i];
    case ORC_DT_I64: return (double)((const int64_t *)src)[i];      /* rounds to nearest above 2^53, as scipy's cast does */
    case ORC_DT_U64: return (double)((const uint64_t *)src)[i];
    case ORC_DT_BOOL: return (double)((const uint8_t *)src)[i];     /* npy_bool is an unsigned char */
    default: return (double)((const int32_t *)src)[i];
    }
}

static double mc_tap(const void *src, int dtype, int64_t rs, int64_t iy, int64_t ix)
{
    return (iy < 0 || ix < 0) ? 0.0 : load_typed(src, dtype, iy * rs + ix);
}

static inline double round_unsigned(double t, double hi)
{
    t = t > 0 ? t + 0.5 : 0.0;
    return t > hi ? hi : t;
}

static inline double round_signed(double t, double lo, double hi)
{
    t = t > 0 ? t + 0.5 : t - 0.5;
    t = t > hi ? hi : t;
    return t < lo ? lo : t;
}

/* 64-bit integers: scipy's clamps compare against NPY_MAX_INT64 / NPY_MAX_UINT64 converted to double, i.e. 2^63 / 2^64, which
 * the cast that follows cannot represent -- undefined behaviour in C.  What the reference DOES on x86-64 (scipy 1.15.3 wheels,
 * probed by tools/gen_golden.py, golden G12): cvttsd2si returns the "integer indefinite" 0x8000000000000000 for every double
 * outside [-2^63, 2^63), and the unsigned cast is cvttsd2si(t) below 2^63, else cvttsd2si(t - 2^63) ^ 2^63 -- so a result of
 * 2^63 (any blend of taps near INT64_MAX) stores INT64_MIN and 2^64 stores 0.  Restated with defined operations only. */
static inline int64_t x86_cvttsd2si(double t)
{
    return (t >= -9223372036854775808.0 && t < 9223372036854775808.0) ? (int64_t)t : INT64_MIN;
}

static inline uint64_t x86_double_to_u64(double t)
{
    if (t < 9223372036854775808.0) return (uint64_t)x86_cvttsd2si(t);
    return (uint64_t)x86_cvttsd2si(t - 9223372036854775808.0) ^ 0x8000000000000000ull;
}

static inline void store_typed(void *dst, int dtype, int64_t i, double t)
{
    switch (dtype) {
    case ORC_DT_I64: {
        /* (round 5, golden G12b at order 3) a value STRICTLY above 2^63 takes scipy's clamp branch, `_t = NPY_MAX_INT64; (npy_int64)_t`,
         * whose out-of-range conversion the reference's compiler folded at build time -- to INT64_MAX, saturating --, while a value
         * of exactly 2^63 is not clamped and goes through the run-time cvttsd2si: INT64_MIN.  Likewise 2^64 / UINT64_MAX below. */
        const double r_ = round_signed(t, -9223372036854775808.0, 1.0e300);
        ((int64_t *)dst)[i] = r_ > 9223372036854775808.0 ? INT64_MAX : x86_cvttsd2si(r_);
        break;
    }
    case ORC_DT_U64: {
        const double r_ = round_unsigned(t, 1.0e300);
        ((uint64_t *)dst)[i] = r_ > 18446744073709551616.0 ? UINT64_MAX : x86_double_to_u64(r_);
        break;
    }
    case ORC_DT_BOOL: ((uint8_t *)dst)[i] = (uint8_t)(t >= 0.0 && t < 256.0 ? t : 0.0); break;   /* CASE_INTERP_OUT(NPY_BOOL): a C cast, truncation */
    case ORC_DT_F32: ((float *)dst)[i] = (float)t; break;
    case ORC_DT_F64: ((double *)dst)[i] = t; break;
    case ORC_DT_U8: ((uint8_t *)dst)[i] = (uint8_t)round_unsigned(t, 255.0); break;
    case ORC_DT_I8: ((int8_t *)dst)[i] = (int8_t)round_signed(t, -128.0, 127.0); break;
    case ORC_DT_U16: ((uint16_t *)dst)[i] = (uint16_t)round_unsigned(t, 65535.0); break;
    case ORC_DT_I16: ((int16_t *)dst)[i] = (int16_t)round_signed(t, -32768.0, 32767.0); break;
    case ORC_DT_U32: ((uint32_t *)dst)[i] = (uint32_t)round_unsigned(t, 4294967295.0); break;
    default: ((int32_t *)dst)[i] = (int32_t)round_signed(t, -2147483648.0, 2147483647.0); break;
    }
}

/* coefficients of the (padded) image: coef is (H + 2 pad) x (W + 2 pad) doubles */
static int spline_coefficients_any(const void *src, int dtype, int64_t H, int64_t W, int64_t src_row_stride, int order,
                                   int mode, double *coef)
{
    double poles[2];
    const int np = spline_poles(order, poles);
    if (np == 0 || H <= 0 || W <= 0) return -1;
    const int pad = orc_spline_pad(mode);
    const int64_t Hp = H + 2 * pad, Wp = W + 2 * pad;
    for (int64_t y = 0; y < Hp; ++y)
        for (int64_t x = 0; x < Wp; ++x) {
            int64_t sy = y - pad, sx = x - pad;
            double v;
            if (mode == ORC_MODE_GRID_CONSTANT && (sy < 0 || sy >= H || sx < 0 || sx >= W)) v = 0.0;
            else {
                sy = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
                sx = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx);
                v = load_typed(src, dtype, sy * src_row_stride + sx);
            }
            coef[y * Wp + x] = v;
        }
    const int kind = spline_filter_kind(mode);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t x = 0; x < Wp; ++x) spline_filter_line(coef + x, Hp, Wp, poles, np, kind);   /* axis 0 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t y = 0; y < Hp; ++y) spline_filter_line(coef + y * Wp, Wp, 1, poles, np, kind); /* axis 1 */
    return 0;
}

int orc_spline_coefficients_f32(const float *src, int64_t H, int64_t W, int64_t src_row_stride, int order,
                                int mode, double *coef)
{
    return spline_coefficients_any(src, ORC_DT_F32, H, W, src_row_stride, order, mode, coef);
}

/* centred B-spline weights; returns the first tap index */
static int64_t spline_weights(int order, double x, double *w)
{
    double s, t;
    if (order & 1) s = floor(x);
    else s = floor(x + 0.5);
    t = x - s;
    const int64_t start = (int64_t)s - order / 2;
    double y = t, z = 1.0 - t, t2;
    switch (order) {
    case 2:
        w[1] = 0.75 - t * t;
        y = 0.5 + t;
        w[2] = 0.5 * y * y;
        w[0] = 1.0 - w[1] - w[2];
        break;
    case 3:
        w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
        w[0] = z * z * z / 6.0;
        w[3] = 1.0 - w[0] - w[1] - w[2];
        break;
    case 4:
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
        break;
    default: /* 5 */
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
        break;
    }
    return start;
}

static inline int64_t spline_fold(int64_t i, int64_t n, int mode)
{
    if (i >= 0 && i < n) return i;
    if (mode == ORC_MODE_REFLECT || mode == ORC_MODE_GRID_MIRROR) {
        const int64_t s2 = 2 * n;
        i %= s2;
        if (i < 0) i += s2;
        return i < n ? i : s2 - 1 - i;
    }
    if (mode == ORC_MODE_GRID_WRAP) {
        i %= n;
        return i < 0 ? i + n : i;
    }
    if (mode == ORC_MODE_NEAREST || mode == ORC_MODE_GRID_CONSTANT) /* padded: cannot happen for in-range points */
        return i < 0 ? 0 : n - 1;
    if (n == 1) return 0;                     /* mirror, constant, wrap */
    const int64_t s2 = 2 * n - 2;
    i %= s2;
    if (i < 0) i += s2;
    return i < n ? i : s2 - i;
}

/* one point of the padded coefficient array; (y, x) are coordinates in the UNPADDED image */
static inline double spline_sample_d(const double *coef, int64_t Hp, int64_t Wp, int pad, double y, double x,
                                     int order, int mode)
{
    double wy[6], wx[6];
    const int64_t sy = spline_weights(order, y + (double)pad, wy);
    const int64_t sx = spline_weights(order, x + (double)pad, wx);
    double t = 0.0;
    for (int j = 0; j <= order; ++j) {
        const int64_t iy = spline_fold(sy + j, Hp, mode);
        /* 'grid-constant': a tap outside the padded plane reads cval = 0 (only reachable from coordinates outside the image) */
        const int yout = mode == ORC_MODE_GRID_CONSTANT && (sy + j < 0 || sy + j >= Hp);
        for (int i = 0; i <= order; ++i) {
            const int64_t ix = spline_fold(sx + i, Wp, mode);
            const int out = yout || (mode == ORC_MODE_GRID_CONSTANT && (sx + i < 0 || sx + i >= Wp));
            t += ((out ? 0.0 : coef[iy * Wp + ix]) * wy[j]) * wx[i];
        }
    }
    return t;
}

/* Spline orders, a caller's coordinate that may lie OUTSIDE the image: what scipy does before it evaluates the spline.
   'constant' returns cval, 'nearest' and 'grid-constant' evaluate where the coordinate is (taps outside the padded plane
   clamp to its edge / read cval), every other mode moves the coordinate into the extended image with map_coordinate()
   of the order 0/1 section.  Returns 1 when the result is cval. */
static inline int spline_map_point(double *y, double *x, int64_t H, int64_t W, int pad, int mode)
{
    if (*y >= 0.0 && *y <= (double)(H - 1) && *x >= 0.0 && *x <= (double)(W - 1)) return 0;
    (void)pad;
    /* 'nearest' and 'grid-constant' evaluate where the coordinate is: their taps outside the padded plane clamp to its
       edge / read cval (scipy does not move the coordinate for these two at the spline orders -- checked against it) */
    if (mode == ORC_MODE_NEAREST || mode == ORC_MODE_GRID_CONSTANT) return 0;
    *y = mc_map_coordinate(*y, H, mode);
    *x = mc_map_coordinate(*x, W, mode);
    return mode == ORC_MODE_CONSTANT && (*y <= -1.0 || *x <= -1.0);
}

static inline float spline_sample(const double *coef, int64_t Hp, int64_t Wp, int pad, double y, double x,
                                  int order, int mode)
{
    return (float)spline_sample_d(coef, Hp, Wp, pad, y, x, order, mode);
}

/* map_kind: 0 radial (unwarp_image_backward), 1 perspective (correct_perspective_image), 2 explicit
   coordinates (float32 / float64, clamped to the image) */
int orc_remap_spline_f32(const float *src, float *dst, int64_t H, int64_t W, int64_t src_row_stride, int map_kind,
                         double xc, double yc, const double *fact, int nfact, const double *coef8,
                         const void *ycoord, const void *xcoord, int coord_is_f64, int64_t npts, int order,
                         int mode, int poly_mode, double *workspace)
{
    if (order < 2 || order > 5 || mode < 0 || mode > 7) return -1;
    if (orc_spline_coefficients_f32(src, H, W, src_row_stride, order, mode, workspace) != 0) return -1;
    const int pad = orc_spline_pad(mode);
    const int64_t Hp = H + 2 * pad, Wp = W + 2 * pad;
    if (map_kind == 2) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (int64_t i = 0; i < npts; ++i) {
            double y = coord_is_f64 ? ((const double *)ycoord)[i] : (double)((const float *)ycoord)[i];
            double x = coord_is_f64 ? ((const double *)xcoord)[i] : (double)((const float *)xcoord)[i];
            dst[i] = spline_map_point(&y, &x, H, W, pad, mode) ? 0.0f : spline_sample(workspace, Hp, Wp, pad, y, x, order, mode);
        }
        return 0;
    }
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x) {
            double xd, yd;
            if (map_kind == 0)
                radial_coord((double)x, (double)y, xc, yc, fact, nfact, poly_mode, (double)(W - 1), (double)(H - 1), 1,
                             &xd, &yd);
            else
                persp_coord((double)x, (double)y, coef8, (double)(W - 1), (double)(H - 1), 1, &xd, &yd);
            dst[y * W + x] = spline_sample(workspace, Hp, Wp, pad, yd, xd, order, mode);
        }
    return 0;
}

/* scipy.ndimage.map_coordinates(src, (ycoord, xcoord), order, mode) for a 2-D `src` of any of the
 * element types above and npts coordinates clamped into the image, output of the same type --
 * what the reference's four functions reduce to once the coordinates exist
 * (postprocessing.py:146-147, 226-228, 250-251, 490-491).  Orders 0..5; `workspace` holds
 * (H + 2 pad) x (W + 2 pad) doubles for order >= 2. */
int orc_map_coordinates_typed(const void *src, void *dst, int dtype, int64_t H, int64_t W, int64_t src_row_stride,
                              const void *ycoord, const void *xcoord, int coord_is_f64, int64_t npts, int order,
                              int mode, double *workspace)
{
    if (H <= 0 || W <= 0 || npts < 0 || order < 0 || order > 5 || mode < 0 || mode > 7) return -1;
    if (dtype < ORC_DT_F32 || dtype > ORC_DT_BOOL) return -1;
    const int pad = orc_spline_pad(mode);
    const int64_t Hp = H + 2 * pad, Wp = W + 2 * pad;
    if (order >= 2 && spline_coefficients_any(src, dtype, H, W, src_row_stride, order, mode, workspace) != 0) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < npts; ++i) {
        double y = coord_is_f64 ? ((const double *)ycoord)[i] : (double)((const float *)ycoord)[i];
        double x = coord_is_f64 ? ((const double *)xcoord)[i] : (double)((const float *)xcoord)[i];
        if (order <= 1 && mode != 4 && (y < 0.0 || y > (double)(H - 1) || x < 0.0 || x > (double)(W - 1))) {
            store_typed(dst, dtype, i, mc_sample_outside(src, dtype, H, W, src_row_stride, y, x, order, mode));
            continue;
        }
        if (order >= 2) {
            store_typed(dst, dtype, i, spline_map_point(&y, &x, H, W, pad, mode) ? 0.0 : spline_sample_d(workspace, Hp, Wp, pad, y, x, order, mode));
            continue;
        }
        y = clipd(y, 0.0, (double)(H - 1));
        x = clipd(x, 0.0, (double)(W - 1));
        double t;
        if (order >= 2) {
            t = spline_sample_d(workspace, Hp, Wp, pad, y, x, order, mode);
        } else if (order == 0) {
            const int64_t iy = fold_edge((int64_t)floor(y + 0.5), H), ix = fold_edge((int64_t)floor(x + 0.5), W);
            t = load_typed(src, dtype, iy * src_row_stride + ix);
        } else {
            const double y0 = floor(y), x0 = floor(x);
            const double wy0 = 1.0 - (y - y0), wy1 = 1.0 - wy0;
            const double wx0 = 1.0 - (x - x0), wx1 = 1.0 - wx0;
            const int64_t iy0 = fold_edge((int64_t)y0, H), ix0 = fold_edge((int64_t)x0, W);
            const int64_t iy1 = fold_edge((int64_t)y0 + 1, H), ix1 = fold_edge((int64_t)x0 + 1, W);
            t = 0.0;
            t += (load_typed(src, dtype, iy0 * src_row_stride + ix0) * wy0) * wx0;
            t += (load_typed(src, dtype, iy0 * src_row_stride + ix1) * wy0) * wx1;
            t += (load_typed(src, dtype, iy1 * src_row_stride + ix0) * wy1) * wx0;
            t += (load_typed(src, dtype, iy1 * src_row_stride + ix1) * wy1) * wx1;
        }
        store_typed(dst, dtype, i, t);
    }
    return 0;
}
