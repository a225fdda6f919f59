/* oracle/unwarp_oracle.h -- declarations for the CPU restatement (test infrastructure only). */
#ifndef UNWARP_ORACLE_H
#define UNWARP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_POLY_KERNEL 0
#define ORC_POLY_NUMPY 1
#define ORC_POLY_KERNEL_MULADD 2 /* flip-rate comparison only (tools/flip_rate.py) */
#define ORC_BLEND_SCIPY 0
#define ORC_BLEND_F64LERP 1
#define ORC_BLEND_F32LERP 2

/* scipy boundary modes, in the order of the reference's docstrings (postprocessing.py:128-130) */
#define ORC_MODE_REFLECT 0
#define ORC_MODE_GRID_MIRROR 1
#define ORC_MODE_CONSTANT 2
#define ORC_MODE_GRID_CONSTANT 3
#define ORC_MODE_NEAREST 4
#define ORC_MODE_MIRROR 5
#define ORC_MODE_GRID_WRAP 6
#define ORC_MODE_WRAP 7

/* element types of orc_map_coordinates_typed (same codes as DCP_DTYPE_* in include/discorpy_hip.h) */
#define ORC_DT_F32 0
#define ORC_DT_F64 1
#define ORC_DT_U8 2
#define ORC_DT_I8 3
#define ORC_DT_U16 4
#define ORC_DT_I16 5
#define ORC_DT_U32 6
#define ORC_DT_I32 7
#define ORC_DT_I64 8
#define ORC_DT_U64 9
#define ORC_DT_BOOL 10

void orc_set_threads(int n);
int orc_get_threads(void);
int orc_max_threads(void);

int orc_radial_coords_rows(int64_t H, int64_t W, double xc, double yc, const double *fact, int nfact, int poly_mode,
                           int round_f32, double row_start, int64_t nrows, double *yd, double *xd);
int orc_radial_coords(int64_t H, int64_t W, double xc, double yc, const double *fact, int nfact,
                      int poly_mode, int round_f32, double *yd, double *xd);
int orc_perspective_coords(int64_t H, int64_t W, const double *coef, int round_f32, double *yd, double *xd);
int orc_unwarp_image_f32(const float *src, float *dst, int64_t H, int64_t W, int64_t src_row_stride,
                         double xc, double yc, const double *fact, int nfact, int order,
                         int coord_round_f32, int poly_mode, int blend_mode);
int orc_perspective_image_f32(const float *src, float *dst, int64_t H, int64_t W, int64_t src_row_stride,
                              const double *coef, int order, int blend_mode);
int orc_unwarp_fused_f32(const float *src, float *dst, int64_t H, int64_t W, int64_t src_row_stride,
                         double xc, double yc, const double *fact, int nfact, const double *coef,
                         int order, int poly_mode, int blend_mode);
int orc_remap_coords_f32(const float *src, float *dst, int64_t H, int64_t W, int64_t src_row_stride,
                         const void *ycoord, const void *xcoord, int coord_is_f64, int64_t npts,
                         int order, int blend_mode, int mode);
void orc_chunk_band(int64_t H, int64_t W, double xc, double yc, const double *fact, int nfact, double row_first,
                    double row_last, int64_t *b0, int64_t *b1);
int orc_unwarp_stack_rows_f32(const float *vol, float *out, int64_t D, int64_t H, int64_t W,
                              double xc, double yc, const double *fact, int nfact, double row_start,
                              int64_t nrows, int coord_round_f32, int poly_mode, int blend_mode);
int orc_spline_pad(int mode);
int orc_spline_coefficients_f32(const float *src, int64_t H, int64_t W, int64_t src_row_stride, int order, int mode,
                                double *coef);
int orc_remap_spline_f32(const float *src, float *dst, int64_t H, int64_t W, int64_t src_row_stride, int map_kind,
                         double xc, double yc, const double *fact, int nfact, const double *coef8,
                         const void *ycoord, const void *xcoord, int coord_is_f64, int64_t npts, int order,
                         int mode, int poly_mode, double *workspace);
int orc_map_coordinates_typed(const void *src, void *dst, int dtype, int64_t H, int64_t W, int64_t src_row_stride,
                              const void *ycoord, const void *xcoord, int coord_is_f64, int64_t npts, int order,
                              int mode, double *workspace);
#ifdef __cplusplus
}
#endif
#endif
