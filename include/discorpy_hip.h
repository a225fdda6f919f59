/*
 * discorpy_hip.h -- C ABI of libdiscorpy_hip.so: the MI355X (gfx950) implementation of the
 * backward-unwarp path of discorpy.post.postprocessing.
 *
 * The reference is pure Python (no FFI of its own); each entry point below replaces the body of
 * one reference function, cited as file:line under /root/reference.  INTEGRATION.md shows the
 * ctypes stub a discorpy maintainer would add to call them.
 *
 * Conventions
 *   - every function returns DCP_OK (0) or a negative DCP_ERR_* code; dcp_last_error() returns a
 *     thread-local, human-readable message for the last failure on the calling thread.
 *   - images are float32, row-major; `src_row_stride` / `src_col_stride` are in ELEMENTS, the
 *     output is always dense (height x width).  x = column index, y = row index.
 *   - mem_kind DCP_MEM_HOST: pointers are host memory, the library stages H2D/D2H itself and
 *     returns after the result is in `dst`.  DCP_MEM_DEVICE: pointers are device memory on
 *     `device`, the kernel is enqueued on `stream` (a hipStream_t, NULL = default stream) and the
 *     call returns without synchronising; the caller owns the memory and its lifetime.
 *   - device < 0 means "the calling thread's current HIP device".
 *   - order: 0 nearest (floor(c + 0.5)), 1 bilinear; orders 2..5 go through the *_spline_* entry points.
 *   - blend_mode (order 1 only): DCP_BLEND_SCIPY reproduces scipy.ndimage.map_coordinates'
 *     float64 arithmetic bit for bit; DCP_BLEND_F64LERP is the factorised float64 form (equal to
 *     it to <= 1 float32 ulp, in practice bit-equal); DCP_BLEND_F32LERP is float32 arithmetic
 *     (<= 2 float32 ulp of the largest tap; opt-in).
 */
#ifndef DISCORPY_HIP_H
#define DISCORPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden and an export list (csrc/exports.map): exactly these entry points are visible */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define DCP_OK 0
#define DCP_ERR_INVALID_ARG (-1)
#define DCP_ERR_HIP (-2)
#define DCP_ERR_UNSUPPORTED (-3)
#define DCP_ERR_NO_DEVICE (-4)

#define DCP_MEM_HOST 0
#define DCP_MEM_DEVICE 1
/* DCP_MEM_DEVICE for callers that hand over INDEPENDENT frames one call at a time (a camera loop, the channels of demo_06.py:111-113):
 * the kernel is enqueued on `stream` with the barrier bit of its dispatch packet cleared (hipExtAnyOrderLaunch), so its first
 * workgroups may start while the last workgroups of the PREVIOUS kernel on that stream still run -- the drain of one frame under
 * the ramp of the next, which is what dcp_unwarp_images_f32 gets from one launch (profiles/r06a_dispatch_modes.txt).  The caller
 * vouches that the call reads nothing an unfinished earlier call on the stream writes and writes nothing it reads or writes;
 * events, copies and synchronisations on the stream still wait for every earlier launch.  Accepted by dcp_unwarp_image_f32,
 * dcp_perspective_image_f32 and dcp_unwarp_fused_f32 (orders 0 / 1); DCP_ERR_INVALID_ARG elsewhere. */
#define DCP_MEM_DEVICE_UNORDERED 0x101

#define DCP_BLEND_SCIPY 0
#define DCP_BLEND_F64LERP 1
#define DCP_BLEND_F32LERP 2

#define DCP_COORD_F32 0
#define DCP_COORD_F64 1

#define DCP_MAX_FACT 32

/* ---- library / device ---- */
int dcp_version(void);                 /* major*10000 + minor*100 + patch */
int dcp_device_count(void);            /* number of HIP devices, 0 if none / no driver */
const char* dcp_last_error(void);

/* Device scratch the library keeps between calls -- the calling thread's staging buffers and streams for
 * DCP_MEM_HOST calls (grow-only, otherwise freed when the thread ends) and the per-device float64 planes of the
 * spline path -- is released; the next call allocates again.  For long-running services. */
int dcp_release_scratch(void);

/* Options (process-wide).  The documented ones:
 *   "stack_chunk_kb"  KiB of one projection chunk when a HOST stack is streamed through the GPU (device scratch = 4 chunks; default 24576)
 *   "host_duplex"     host frames of the radial map through the GPU in bands of rows, uploads and downloads at the same time:
 *                     0 never, 1 (default) when a one-off probe finds that the HIP runtime overlaps the two directions, 2 always
 *   "host_bands"      number of those bands (default 6; where uploads and downloads are both runtime copies the frame travels in twice as
 *                     many, the first and last ones shorter -- 128, 256, 512 .. rows -- so that the stretch with one direction idle is short)
 *   "host_direct"     0: never write a host frame's result straight into registered host memory; 1 (default): when the runtime
 *                     cannot overlap an upload with a download; 2: whenever the destination is registered
 *   "host_direct_applies"   read-only: 1 if "host_direct" = 1 takes effect on this runtime (measured once; needs a device)
 *   "tile_cert"       0: never trust the host's tile-deviation certificate -- every staged kernel then verifies every pixel's taps
 *                     against its source box (default 1)
 *   "lds_gather"      0: no LDS-staged kernels at all: every tap a global load (default 1)
 * Every other switch of the library (which kernel takes a call, tile orders, chunk sizes: what the A/B runs under tools/ and the
 * parity campaigns flip) answers only to its name prefixed with "x_"; those are not part of this interface and may change
 * (list: csrc/api_core.cpp, dcp_set_option).  Returns DCP_ERR_INVALID_ARG for an unknown key. */
int dcp_set_option(const char* key, int value);
int dcp_get_option(const char* key, int* value);

/* ---- the hot path ---- */

/* discorpy/post/postprocessing.py:111-148  unwarp_image_backward
 * dst[y,x] = sample(src, clip(yc + B*yu), clip(xc + B*xu)),  B = sum_i list_fact[i] * ru^i,
 * xu = x - xcenter, yu = y - ycenter, ru = sqrt(xu^2 + yu^2), evaluated in float64.
 * coord_round_f32 = 1 rounds the clipped coordinates to float32 as :144-145 does. */
int dcp_unwarp_image_f32(const float* src, float* dst, int64_t height, int64_t width,
                         int64_t src_row_stride, int64_t src_col_stride, double xcenter, double ycenter,
                         const double* list_fact, int nfact, int order, int coord_round_f32, int blend_mode,
                         int mem_kind, int device, void* stream);

/* unwarp_image_backward (discorpy/post/postprocessing.py:111-148) over `nframes` images of ONE shape in one call, every
 * frame with its own source, destination, centre and coefficient vector -- the reference's callers loop over channels
 * (examples/readthedocs_demo/demo_06.py:111-113, demo_07.py:58-60), cameras or candidate calibrations (example_05.py:62-65),
 * one call per image.  srcs[i] / dsts[i]: the frames (all with the strides given); xcenters[i], ycenters[i]; list_facts:
 * nframes x nfact, row-major (pad shorter vectors with zeros: the result does not change).  Device-resident dense
 * float32 frames (unit column stride, coord_round_f32 = 1, nfact <= 10) whose calibrations all hold the tile certificate
 * run as ONE launch per 55 frames (35 above 5 coefficients) with the workgroup's frame in blockIdx.z, so that the drain of
 * one frame overlaps the ramp of the next (a 4096 x 4096 frame: ~25 us instead of ~31 us per frame); anything else is
 * processed frame by frame through dcp_unwarp_image_f32.  Results are bit-identical to nframes separate calls. */
int dcp_unwarp_images_f32(const float* const* srcs, float* const* dsts, int nframes, int64_t height, int64_t width,
                          int64_t src_row_stride, int64_t src_col_stride, const double* xcenters, const double* ycenters,
                          const double* list_facts, int nfact, int order, int coord_round_f32, int blend_mode, int mem_kind,
                          int device, void* stream);

/* discorpy/post/postprocessing.py:444-459 (_generate_perspective_map) + :486-492
 * (correct_perspective_image, map_index=None).  list_coef = c1..c8 of the backward homography in
 * (x, y) convention (discorpy/proc/processing.py:1254-1270). */
int dcp_perspective_image_f32(const float* src, float* dst, int64_t height, int64_t width,
                              int64_t src_row_stride, int64_t src_col_stride, const double* list_coef,
                              int order, int blend_mode, int mem_kind, int device, void* stream);

/* One-pass composition (BASELINE config 3): the float32-rounded perspective coordinate of
 * :453-457 is fed to the radial map of :141-145 and the source is sampled once.  NOT equal to
 * calling the two functions above in sequence (that is two resamplings, demo_05.py:127,147). */
int dcp_unwarp_fused_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                         int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                         int nfact, const double* list_coef, int order, int blend_mode, int mem_kind,
                         int device, void* stream);

/* discorpy/post/postprocessing.py:489-491 (map_index given) and :250-251 (_mapping):
 * dst[i] = sample(src, ycoord[i], xcoord[i]) for npts caller-supplied coordinates of type
 * coord_dtype (DCP_COORD_F32 / DCP_COORD_F64), clamped to the image (scipy's mode='nearest'). */
int dcp_remap_coords_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                         int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype,
                         int64_t npts, int order, int blend_mode, int mem_kind, int device, void* stream);

/* The same with the reference's `mode` argument honoured for coordinates OUTSIDE the image, as
 * scipy.ndimage.map_coordinates -- which postprocessing.py:489-491 hands `map_index` and `mode` to -- treats them at
 * orders 0 and 1: boundary_mode is the index of the mode in the reference's docstring order (0 reflect, 1 grid-mirror,
 * 2 constant, 3 grid-constant, 4 nearest, 5 mirror, 6 grid-wrap, 7 wrap; cval = 0).  Such points are computed in double
 * in scipy's operation order whatever blend_mode says; points inside the image are what dcp_remap_coords_f32 returns. */
int dcp_remap_coords_mode_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                              int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype,
                              int64_t npts, int order, int boundary_mode, int blend_mode, int mem_kind, int device,
                              void* stream);

/* discorpy/post/postprocessing.py:188-229 (unwarp_slice_backward: nrows = 1, coord_round_f32 = 0)
 * and :255-313 (unwarp_chunk_slices_backward: coord_round_f32 = 1) over a (depth, height, width)
 * stack: out[d, r, x] (dense, depth x nrows x width) = projection d sampled bilinearly at the
 * radial source coordinate of output pixel (row_start + r, x).  proj_stride / row_stride are the
 * element strides of `vol` between projections / rows.
 * coord_round_f32 = 2: float32 coordinates under unwarp_image_backward's semantics (:141-148) -- the projections are the FRAMES of a
 * (n, height, width) array corrected with one calibration: every coordinate is clipped to the whole frame and the chunk function's
 * row band [yd_min, yd_max) (:289-301) plays no part, so a model that folds rows gives exactly what n calls of
 * dcp_unwarp_image_f32 give.  The requested rows must lie inside the frame.  (dcp_unwarp_images_f32 routes frames of one
 * calibration this way by itself; dcp_unwarp_stack_rows_typed accepts the value too.) */
int dcp_unwarp_stack_rows_f32(const float* vol, float* out, int64_t depth, int64_t height, int64_t width,
                              int64_t proj_stride, int64_t row_stride, double xcenter, double ycenter,
                              const double* list_fact, int nfact, double row_start, int64_t nrows,
                              int coord_round_f32, int blend_mode, int mem_kind, int device, void* stream);

/* dcp_unwarp_stack_rows_f32 under `ncentres` calibrations that differ in the centre of distortion only, in one call -- the grid
 * search of examples/example_05.py:62-65 (unwarp_slice_backward 11 x 11 = 121 times on one stack of 600 projections, one call per
 * candidate centre).  out = (ncentres, depth, nrows, width), dense; block k is what dcp_unwarp_stack_rows_f32 returns for
 * (xcenters[k], ycenters[k]), bit for bit.  One launch per 224 centres: the two or three source rows a sinogram needs of each
 * projection are fetched from HBM once and served to every centre by the L2 (device-resident stacks); a host stack is shipped
 * once -- the union of the centres' row bands -- instead of once per centre.  coord_round_f32 = 1 under a model that may fold
 * rows out of the reference's band is processed centre by centre. */
int dcp_unwarp_stack_rows_centres_f32(const float* vol, float* out, int64_t depth, int64_t height, int64_t width, int64_t proj_stride,
                                      int64_t row_stride, const double* xcenters, const double* ycenters, int ncentres,
                                      const double* list_fact, int nfact, double row_start, int64_t nrows, int coord_round_f32,
                                      int blend_mode, int mem_kind, int device, void* stream);

/* The same over a HOST-resident stack sharded across `ndev` GPUs of this process (SURVEY.md section
 * 8(e): projections are independent).  devices[i] takes the i-th contiguous depth shard (sizes as
 * numpy.array_split), stages it, runs the kernel and copies its block of `out` (depth x nrows x width,
 * host) back; the shards run concurrently on one host thread each.  For DEVICE-resident shards use one
 * process per GPU and an RCCL all-gather (discorpy_amd/stack.py). */
int dcp_unwarp_stack_rows_multi_f32(const float* vol, float* out, int64_t depth, int64_t height, int64_t width,
                                    int64_t proj_stride, int64_t row_stride, double xcenter, double ycenter,
                                    const double* list_fact, int nfact, double row_start, int64_t nrows,
                                    int coord_round_f32, int blend_mode, const int* devices, int ndev);

/* Device-resident depth shards on the GPUs of ONE process and the exchange by direct peer copies (no torch, no RCCL):
 * slot g holds projections [d0_g, d1_g) -- the split of numpy.array_split(range(depth), ndev) -- at vol[g] on
 * devices[g]; its kernel writes its (d1_g - d0_g, nrows, width) block, and with gather != 0 every slot's block is
 * pushed into the depth-outer (depth, nrows, width) result out[h] of every other slot with hipMemcpyPeerAsync (on an
 * 8-GPU node: seven pushes per device, one per xGMI link).  out[g] holds (depth, nrows, width) floats when gathering
 * (slot g writes at depth offset d0_g), (d1_g - d0_g, nrows, width) otherwise.  A device may appear in several slots.
 * Returns after every stream has drained.  Reference: the loops over depth of postprocessing.py:226-228, 310-312 carry no
 * state, so the projections may be split anywhere; the reference itself has no multi-device code. */
int dcp_unwarp_stack_rows_peer_f32(const float* const* vol, float* const* out, int64_t depth, int64_t height, int64_t width,
                                   int64_t proj_stride, int64_t row_stride, double xcenter, double ycenter,
                                   const double* list_fact, int nfact, double row_start, int64_t nrows, int coord_round_f32,
                                   int blend_mode, const int* devices, int ndev, int gather);

/* ---- one process per GPU: the depth-sharded stack and ONE RCCL all-gather over xGMI, without torch ----
 * (north star / SURVEY.md section 8(e); the reference has no multi-device code: its loops over depth,
 * discorpy/post/postprocessing.py:226-228 and 310-312, carry no state, so the projections may be split anywhere.)
 * librccl.so is loaded on first use from the directory of the HIP runtime the process runs on (DCP_RCCL_PATH overrides);
 * every function below returns DCP_ERR_UNSUPPORTED when it cannot be loaded.
 *   dcp_rccl_unique_id      ncclGetUniqueId: 128 bytes that ONE rank creates and the caller ships to the others (MPI, a file, a socket)
 *   dcp_rccl_comm_create    ncclCommInitRank on `device` (collective: every rank of the world calls it), plus the side stream of the
 *                           pipelined exchange; dcp_rccl_comm_destroy releases both
 *   dcp_unwarp_stack_rows_rccl_f32
 *       COLLECTIVE: every rank of the communicator makes the call.  This rank holds `depth_local` projections at `vol` and a
 *       result buffer `out` of (sum of all ranks' depth_local, nrows, width) floats on the communicator's device.  The ranks first
 *       agree on their shard shapes (one all-gather of 40 bytes per rank on the communicator's side stream; the caller's stream is
 *       not waited for): nrows, width and pipeline must be the same everywhere, depth_local may differ from rank to rank and may
 *       be 0 (the shards are laid down in rank order), and a rank whose own arguments are unusable says so there -- a
 *       disagreement makes EVERY rank return DCP_ERR_INVALID_ARG from this call, none is left waiting in a collective.  Then the
 *       stack kernel (dcp_unwarp_stack_rows_f32) writes this rank's block in place at its depth offset and the exchange fills in
 *       the other ranks' blocks -- depth is the outer axis of the result, so every contribution is one contiguous block: equal
 *       shards, pipeline <= 1: ONE ncclAllGather in place on `stream` behind the kernel; ragged shards: one group of
 *       ncclBroadcasts with per-rank counts.  pipeline > 1: every shard is cut into that many depth sub-blocks and the exchange of
 *       sub-block s (grouped ncclBroadcasts on the side stream) overlaps the kernel of sub-block s + 1.  Stream-ordered: returns
 *       without waiting for the result, `stream` is complete when the whole (depth, nrows, width) result is.  After a HIP / RCCL
 *       failure on one rank the others may be blocked in their collectives: destroy the communicator (the caller's stream is
 *       re-joined to the side stream on every path, so nothing is left running behind the caller's back).
 *       The agreement is a HOST wait on the side stream, and RCCL serialises the operations of one communicator: a call therefore also
 *       waits for the previous call's exchange (back-to-back calls do not overlap) -- unless
 *   dcp_rccl_comm_fixed_shards(comm, 1)
 *       has been set, by which the caller vouches ON EVERY RANK that the following calls repeat the depth_local / nrows / width /
 *       pipeline of the last agreed call: the agreement is skipped (no collective, no host wait) and the call is stream-ordered end to
 *       end.  A rank that then passes other shapes gets DCP_ERR_INVALID_ARG; its peers are not told (destroy the communicator).
 *   dcp_rccl_comm_info
 *       what the communicator ITSELF reports, so that a multi-GPU run can prove what RCCL saw rather than what the launcher claimed:
 *       info[0] ncclCommCount, [1] ncclCommUserRank, [2] ncclCommCuDevice, [3] ncclGetVersion (-1 each where the loaded library lacks
 *       the query), [4] / [5] the world size / rank given to dcp_rccl_comm_create, [6] the HIP device, [7] 1 if a shard agreement is
 *       in force, [8] exchanges done, [9] agreements done (up to `ninfo` values are written); shard_depths[r] = depth_local of rank r
 *       as agreed in the last exchange (-1 without one; up to `nshards`); librccl_path = the file the ncclXxx symbols were bound
 *       from (DCP_RCCL_PATH, the librccl.so next to the loaded HIP runtime, or the search path).  Not a collective. */
int dcp_rccl_available(void);
int dcp_rccl_unique_id(void* id, size_t bytes);
int dcp_rccl_comm_create(void** comm, int world_size, int rank, const void* id, int device);
int dcp_rccl_comm_destroy(void* comm);
int dcp_rccl_comm_fixed_shards(void* comm, int on);
int dcp_rccl_comm_info(void* comm, int64_t* info, int ninfo, int64_t* shard_depths, int nshards, char* librccl_path, size_t path_bytes);
int dcp_unwarp_stack_rows_rccl_f32(const float* vol, float* out, int64_t depth_local, int64_t height, int64_t width, int64_t proj_stride,
                                   int64_t row_stride, double xcenter, double ycenter, const double* list_fact, int nfact, double row_start,
                                   int64_t nrows, int coord_round_f32, int blend_mode, void* comm, int pipeline, void* stream);

/* ---- spline orders 2..5 (scipy.ndimage.map_coordinates with its B-spline prefilter) ----
 * The same maps as dcp_unwarp_image_f32 / dcp_perspective_image_f32 / dcp_remap_coords_f32 for
 * `order` in 2..5 -- what the reference computes when a caller passes order >= 2
 * (discorpy/post/postprocessing.py:147, 491; examples/readthedocs_demo/demo_07.py:60 uses 3).  Here the
 * boundary mode matters (prefilter boundary condition, tap folding, 12-sample padding for 'nearest'
 * and 'grid-constant'):  0 reflect, 1 grid-mirror, 2 constant, 3 grid-constant, 4 nearest, 5 mirror,
 * 6 grid-wrap, 7 wrap.  Results are within one float32 ulp of scipy's.  A float64 coefficient plane
 * of (height + 2 pad) x (width + 2 pad) is kept per device by the library. */
/* OR-ed into `boundary_mode` of the spline entry points (and of the *_typed ones at orders >= 2): accumulate the (order + 1)^2 taps
 * of a point in scipy's operation order, t += (c * wy) * wx tap by tap.  Without it the LDS-staged gather of certified maps on
 * whole frames sums them factorised and fused -- sum_j wy_j (sum_q c_jq wx_q): 20 float64 operations instead of 48 at order 3 --,
 * which differs from scipy's sum in the last bits of the float64 value, i.e. in the float32 result of ~1 pixel in 1e8 (the
 * spline orders' contract is "within one float32 ulp of scipy's" either way).  Python: blend="scipy". */
#define DCP_SPLINE_SCIPY_SUM 0x100
#define DCP_MODE_REFLECT 0
#define DCP_MODE_GRID_MIRROR 1
#define DCP_MODE_CONSTANT 2
#define DCP_MODE_GRID_CONSTANT 3
#define DCP_MODE_NEAREST 4
#define DCP_MODE_MIRROR 5
#define DCP_MODE_GRID_WRAP 6
#define DCP_MODE_WRAP 7
int dcp_unwarp_image_spline_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                                int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                                int nfact, int order, int boundary_mode, int mem_kind, int device, void* stream);
int dcp_perspective_image_spline_f32(const float* src, float* dst, int64_t height, int64_t width,
                                     int64_t src_row_stride, int64_t src_col_stride, const double* list_coef, int order,
                                     int boundary_mode, int mem_kind, int device, void* stream);
int dcp_remap_coords_spline_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                                int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype,
                                int64_t npts, int order, int boundary_mode, int mem_kind, int device, void* stream);
/* the one-pass perspective -> radial map of dcp_unwarp_fused_f32 at a spline order */
int dcp_unwarp_fused_spline_f32(const float* src, float* dst, int64_t height, int64_t width, int64_t src_row_stride,
                                int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact,
                                int nfact, const double* list_coef, int order, int boundary_mode, int mem_kind, int device,
                                void* stream);

/* ---- element types other than float32 ----
 * The reference hands `mat` to scipy.ndimage.map_coordinates whatever its dtype
 * (discorpy/post/postprocessing.py:147, 227, 251, 491): every element is read as a double, the blend
 * runs in double in scipy's operation order, and the result is converted to the INPUT's type -- a C
 * cast for floats; for integers round-half-away-from-zero and saturation.  The *_typed entry points
 * do that for `dtype` = DCP_DTYPE_*, orders 0..5 (`boundary_mode` matters for orders >= 2 only); `src` and `dst` hold elements of `dtype`, strides in elements.
 * DCP_DTYPE_F32 is accepted too (exact scipy blend; the *_f32 entry points are the fast path).
 * dcp_unwarp_stack_rows_typed: out_float32 = 1 stores float32 values of the result already converted
 * to `dtype`, as unwarp_slice_backward assigns into its float32 sinogram (:224-227). */
#define DCP_DTYPE_F32 0
#define DCP_DTYPE_F64 1
#define DCP_DTYPE_U8 2
#define DCP_DTYPE_I8 3
#define DCP_DTYPE_U16 4
#define DCP_DTYPE_I16 5
#define DCP_DTYPE_U32 6
#define DCP_DTYPE_I32 7
#define DCP_DTYPE_I64 8  /* read as a double (rounds above 2^53), stored as scipy's cast does on x86-64: see dcp_device.h to_elem */
#define DCP_DTYPE_U64 9
#define DCP_DTYPE_BOOL 10 /* numpy bool_: one byte, 0 / 1; the double result is truncated on the way out */
int dcp_unwarp_image_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width, int64_t src_row_stride,
                           int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact, int nfact,
                           int order, int boundary_mode, int mem_kind, int device, void* stream);
int dcp_perspective_image_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width,
                                int64_t src_row_stride, int64_t src_col_stride, const double* list_coef, int order,
                                int boundary_mode, int mem_kind, int device, void* stream);
int dcp_unwarp_fused_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width, int64_t src_row_stride,
                           int64_t src_col_stride, double xcenter, double ycenter, const double* list_fact, int nfact,
                           const double* list_coef, int order, int boundary_mode, int mem_kind, int device, void* stream);
int dcp_remap_coords_typed(const void* src, void* dst, int dtype, int64_t height, int64_t width, int64_t src_row_stride,
                           int64_t src_col_stride, const void* ycoord, const void* xcoord, int coord_dtype, int64_t npts,
                           int order, int boundary_mode, int mem_kind, int device, void* stream);
int dcp_unwarp_stack_rows_typed(const void* vol, void* out, int dtype, int out_float32, int64_t depth, int64_t height,
                                int64_t width, int64_t proj_stride, int64_t row_stride, double xcenter, double ycenter,
                                const double* list_fact, int nfact, double row_start, int64_t nrows, int coord_round_f32,
                                int mem_kind, int device, void* stream);

/* discorpy/util/utility.py:278-342 (unwarp_color_image_backward), the part after np.pad: an interleaved
 * (height, width, channels) image of `dtype`, every channel sampled at the same radial coordinate
 * (:320-341 loops map_coordinates over mat_pad[:, :, i]) -- one coordinate evaluation and `channels`
 * blends per pixel, orders 0 / 1.  src_pixel_stride = elements between pixels (>= channels); dst is dense
 * (height, width, channels).
 *   dcp_unwarp_color_image     blend_mode DCP_BLEND_SCIPY (scipy's exact arithmetic: bit-equal to the reference) or, for
 *                              float32 pixels, DCP_BLEND_F64LERP (within one float32 ulp of it: what dcp_unwarp_image_f32 computes
 *                              per plane by default); integer element types always blend in scipy's order.  Dense pixels of 3 or 4
 *                              float32 / uint8 / uint16 channels under a certified calibration take the workgroup-box kernel
 *                              (remap_wg_color_kernel: the source box of a 128 x 16 tile staged in LDS once for all channels),
 *                              everything else one thread per pixel -- the same values either way
 *   dcp_unwarp_image_channels  the same with DCP_BLEND_SCIPY (kept for callers of the earlier ABI) */
int dcp_unwarp_color_image(const void* src, void* dst, int dtype, int64_t height, int64_t width, int channels, int64_t src_row_stride,
                           int64_t src_pixel_stride, double xcenter, double ycenter, const double* list_fact, int nfact, int order,
                           int blend_mode, int mem_kind, int device, void* stream);
int dcp_unwarp_image_channels(const void* src, void* dst, int dtype, int64_t height, int64_t width, int channels,
                              int64_t src_row_stride, int64_t src_pixel_stride, double xcenter, double ycenter,
                              const double* list_fact, int nfact, int order, int mem_kind, int device, void* stream);

/* ---- out-of-core stacks ----
 * The reference never touches more of a projection than mat3D[i, yd_min:yd_max, :]
 * (discorpy/post/postprocessing.py:221-228, 295-301), which is what lets it run on an HDF5 dataset
 * larger than memory.  dcp_stack_row_band reports that band for a request (rows row_start ..
 * row_start + nrows - 1): source rows [band_start, band_start + band_rows).  dcp_unwarp_stack_band is
 * dcp_unwarp_stack_rows_f32 / _typed for a buffer that holds ONLY the rows [band_start, band_start +
 * band_rows) of each of `depth` projections (row_stride / proj_stride: element strides inside that
 * buffer); it fails with DCP_ERR_INVALID_ARG if the request needs rows outside the band.  dtype
 * DCP_DTYPE_F32 with out_float32 = 0 runs the tuned kernel with `blend_mode`; every other combination
 * uses scipy's exact blend. */
int dcp_stack_row_band(int64_t height, int64_t width, double xcenter, double ycenter, const double* list_fact, int nfact,
                       double row_start, int64_t nrows, int64_t* band_start, int64_t* band_rows);
int dcp_unwarp_stack_band(const void* band, void* out, int dtype, int out_float32, int64_t depth, int64_t height,
                          int64_t width, int64_t band_start, int64_t band_rows, int64_t proj_stride, int64_t row_stride,
                          double xcenter, double ycenter, const double* list_fact, int nfact, double row_start,
                          int64_t nrows, int coord_round_f32, int blend_mode, int mem_kind, int device, void* stream);

/* The float32 source-coordinate planes themselves, ymap / xmap of height*width floats each:
 * DCP_MAP_RADIAL       yd_mat / xd_mat of unwarp_image_backward, discorpy/post/postprocessing.py:141-145
 * DCP_MAP_PERSPECTIVE  _generate_perspective_map, :444-459 (list_fact ignored)
 * DCP_MAP_FUSED        the composed map of dcp_unwarp_fused_f32
 * for callers that reuse a map (map_index= of correct_perspective_image :478-479, cv2.remap as in
 * discorpy/util/utility.py:425-435).  Feeding the planes to dcp_remap_coords_f32 reproduces the
 * corresponding image entry point bit for bit. */
#define DCP_MAP_RADIAL 0
#define DCP_MAP_PERSPECTIVE 1
#define DCP_MAP_FUSED 2
int dcp_coordinate_map_f32(float* ymap, float* xmap, int64_t height, int64_t width, int map_kind, double xcenter,
                           double ycenter, const double* list_fact, int nfact, const double* list_coef, int mem_kind,
                           int device, void* stream);

/* discorpy/post/postprocessing.py:36-64 (unwarp_line_forward) and discorpy/util/utility.py:192-230
 * (find_point_to_point): the radial model applied to npts points given as (y, x) pairs of doubles,
 * out = centre + B(r) * (p - centre) with B(r) = sum_i list_fact[i] * r^i.  Float64; agrees with the
 * reference's numpy/libm evaluation to a few units in the last place. */
int dcp_map_points_f64(const double* yx_in, double* yx_out, int64_t npts, double xcenter, double ycenter,
                       const double* list_fact, int nfact, int mem_kind, int device, void* stream);

/* discorpy/post/postprocessing.py:414-441 (correct_perspective_line): the homography applied to npts points given as (y, x) pairs of
 * doubles -- xn = (c1 x + c2 y + c3) / (c7 x + c8 y + 1), yn = (c4 x + c5 y + c6) / (c7 x + c8 y + 1) in numpy's operation order with
 * IEEE divisions: bit-equal to the reference. */
int dcp_map_points_perspective_f64(const double* yx_in, double* yx_out, int64_t npts, const double* list_coef, int mem_kind, int device,
                                   void* stream);

/* Diagnostics of the LDS-staged gather on the current device: out[0] = wave tiles whose source
 * box did not fit the LDS slab, out[1] = wave tiles whose containment vote failed (both fall back
 * to the direct gather).  Synchronises the device. */
int dcp_debug_counters(uint64_t* out, int n, int reset);

/* The bounds-checking debug build (make -C discorpy_amd/csrc bounds -> lib/libdiscorpy_hip_bounds.so, loaded through DCP_LIB_PATH;
 * SURVEY.md section 5): every tap the staged kernels read from LDS is checked against its slab.  out[0] = taps outside their
 * slab since the last reset (must be 0: the tile certificate says so), out[1..3] = byte offset, slab size and site number of the
 * first one, out[4] = 1 if this library IS the checking build (0: the product build, which compiles no check and reports zeros).
 * n >= 5.  Synchronises the device. */
int dcp_debug_bounds(uint64_t* out, int n, int reset);

/* Name of the float32 image / stack kernel the calling thread launched last, e.g.
 * "remap_wg_kernel<Radial,NF=5,f64lerp>" (empty before the first launch).  For tests and benchmarks that must say
 * -- and assert -- which kernel a call took; the reference has no counterpart. */
const char* dcp_debug_last_kernel(void);

/* The host's tile-deviation certificate for a calibration on a height x width frame (needs no GPU): 0 = none (the staged
 * kernels verify every pixel), 1 = holds for 64 x 16 wave tiles, 2 = also for 128 x 32 workgroup tiles (remap_wg_kernel,
 * stack_wg_kernel, remap_wg_batch_kernel).  map_kind DCP_MAP_RADIAL (list_fact), DCP_MAP_PERSPECTIVE (list_coef) or DCP_MAP_FUSED (both:
 * 0 or 2).
 * Negative: a DCP_ERR_* code.  For tests and for callers that want to know which kernel their calibration will take. */
int dcp_debug_tile_certificate(int map_kind, int64_t height, int64_t width, double xcenter, double ycenter, const double* list_fact,
                               int nfact, const double* list_coef);

/* ---- device memory / stream / event helpers (so a host language needs no other GPU runtime) ---- */
int dcp_malloc(void** ptr, size_t bytes, int device);
int dcp_free(void* ptr, int device);
#define DCP_COPY_H2D 0
#define DCP_COPY_D2H 1
#define DCP_COPY_D2D 2
int dcp_memcpy(void* dst, const void* src, size_t bytes, int kind, int device, void* stream);
/* Host memory the GPU can address (hipHostRegister): a DCP_MEM_HOST frame call whose `dst` lies in registered memory -- these, or
 * anything from hipHostMalloc -- has its result written straight into it by the kernels, band by band while the source is still
 * uploading (no device copy of the result, no download; option "host_direct" = 0 switches that off).  Registration costs ~0.8 ms
 * per 64 MiB: for buffers that are reused.  Unregister before the memory is freed.  Register WHOLE PAGES that nothing else lives in
 * (an mmap'ed or page-aligned allocation): a malloc'ed block shares its edge pages with other heap objects, and a pageable copy of
 * one of those is pinned and unpinned by the runtime under the registration (GPU memory access faults, round 4). */
int dcp_host_register(void* ptr, size_t bytes, int device);
int dcp_host_unregister(void* ptr);
int dcp_stream_create(void** stream, int device);   /* a non-blocking stream for DCP_MEM_DEVICE calls */
int dcp_stream_destroy(void* stream);
int dcp_stream_synchronize(int device, void* stream);
int dcp_event_create(void** event, int device);
int dcp_event_record(void* event, void* stream);
/* work enqueued on `stream` after this call starts only when `event` (recorded on any stream of the device) has completed --
 * hipStreamWaitEvent: how a caller who spreads independent frames over two streams (two frames in flight: the drain of one under the
 * ramp of the other, profiles/r06a_dispatch_modes.txt) forks from and joins back into one stream without a host synchronisation */
int dcp_stream_wait_event(void* stream, void* event);
int dcp_event_synchronize(void* event);
int dcp_event_elapsed_ms(void* start, void* stop, float* ms);
int dcp_event_destroy(void* event);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DISCORPY_HIP_H */
