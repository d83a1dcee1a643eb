/*
 * gfremap.h -- C ABI of libgfremap.so: the MI355X (gfx950) implementation of gyroflow's per-frame
 * fisheye-undistort + rotation remap hot path.
 *
 * The reference (ElvinC/gyroflow, legacy Python) has no FFI for this path: it calls opencv-python
 * from Python.  Each entry point below therefore cites the Python call site (file:line under the
 * reference tree) whose arithmetic it replaces; the Python modules under gyroflow_amd/ keep the reference's Python call
 * surface on top of this ABI (INTEGRATION.md shows the ctypes binding a maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Pointers are DEVICE pointers unless the name ends
 *     in _host or the comment says "host"; the caller owns every buffer; *_step are row strides in BYTES.
 *   - Every call returns GF_OK (0) or a negative GF_ERR_* and never throws/aborts;
 *     gf_last_error() returns a thread-local message for the last failure on this thread.
 *   - Work is enqueued on the caller's hipStream_t (passed as void*; NULL = default stream) and is
 *     asynchronous.  Only gf_ctx_create/gf_ctx_destroy allocate or free device memory (two small constant tables
 *     and four 16 KB lens-table slots per context); the compute entry points neither allocate nor synchronise,
 *     except the first call that sees a new lens D (see gf_frame_xform below).
 *   - Images are interleaved 3-channel (BGR/RGB order is irrelevant to the arithmetic), depth
 *     GF_DEPTH_U8 or GF_DEPTH_U16.  All image dimensions must be < 32767 (cv::remap's own assert).
 */
#ifndef GFREMAP_H
#define GFREMAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define GF_API __attribute__((visibility("default")))
#else
#define GF_API
#endif

#define GF_VERSION 100

#define GF_OK               0
#define GF_ERR_INVALID_ARG -1
#define GF_ERR_HIP         -2
#define GF_ERR_UNSUPPORTED -3
#define GF_ERR_SINGULAR    -4

/* cv2 depth codes */
#define GF_DEPTH_U8  0   /* CV_8U  */
#define GF_DEPTH_U16 2   /* CV_16U */

/* cv2 border codes as the reference selects them (stabilizer.py:1235-1244) */
#define GF_BORDER_CONSTANT    0   /* cv2.BORDER_CONSTANT    */
#define GF_BORDER_REPLICATE   1   /* cv2.BORDER_REPLICATE   ("REPLICATE") */
#define GF_BORDER_TRANSPARENT 5   /* cv2.BORDER_TRANSPARENT ("HISTORY": dst keeps its previous pixels) */

typedef void* gf_stream_t;        /* hipStream_t */

GF_API int         gf_version(void);
GF_API const char* gf_last_error(void);

/* Opaque per-device context: owns the bicubic weight table and the lens-table cache in device memory. */
GF_API int gf_ctx_create(int device, void** ctx);
GF_API int gf_ctx_destroy(void* ctx);

/* Kernel variant used by gf_undistort_rotate_c3, gf_undistort_rotate_blend_c3 and gf_fisheye_maps_16sc2 of this context (no
 * counterpart in the reference: an A/B switch; results are the same to the tolerance documented in DESIGN.md, and fused ==
 * maps + remap bit for bit in every mode).  RP (the default): coordinates from per-tile polynomials on 64 x 64 tiles, taps
 * gathered from the frame with aligned 12-byte loads.  DIRECT: per-pixel lens-table coordinates, same taps.  RL: RP
 * coordinates + source taps staged through LDS by LDS-DMA (u8 frames with 16-byte aligned base / pitch / frame stride; other
 * inputs run as RP).  The switch takes the context's mutex; launches already queued keep the variant they were queued with. */
#define GF_KERNEL_DIRECT 0
#define GF_KERNEL_RP     1
#define GF_KERNEL_RL     2
GF_API int gf_ctx_set_kernel_mode(void* ctx, int mode);

/* Per-frame transform consumed by the kernels: everything cv2.fisheye.initUndistortRectifyMap derives
 * from (K, R, P) before its pixel loop: iR = (P[:, :3] * R)^-1, f/c = focal length/centre of K.  The lens distortion
 * coefficients D = (k1..k4) belong to the lens, not the frame: the calls below take them as 4 HOST doubles and keep a
 * per-context table of the radial scale function theta_d(atan r)/r for each distinct D (up to 4 lenses cached; the
 * first call with a new D builds the 16 KB table on the host, uploads it on the call's stream and waits for that
 * upload -- every later call is asynchronous; a fifth distinct D evicts the least recently used table after waiting, on
 * per-slot events, for the launches that read it).  A context may be used from several threads and streams at once; the
 * context's mutex is held from a call's cache lookup until its kernel is queued, and a miss (table build, upload, wait; an
 * eviction's wait) happens under it, so other threads' calls on the same context queue behind a miss. */
typedef struct gf_frame_xform {
    double iR[9];
    double f[2];
    double c[2];
} gf_frame_xform;

/* HOST helper (no GPU work).  Builds one gf_frame_xform from the four matrices the reference hands to
 * cv2.fisheye.initUndistortRectifyMap at calibrate_video.py:376/383 (K=scaled_K, R, P=new_K), all
 * row-major host doubles.  GF_ERR_SINGULAR if P*R is not invertible. */
GF_API int gf_make_frame_xform(const double K_host[9], const double R_host[9], const double P_host[9],
                        gf_frame_xform* out_host);

/* HOST helper: gf_make_frame_xform for the n frames of a batch in one call -- the lens matrix K (scaled_K) is the same for
 * every frame, R_host and P_host hold n row-major 3x3 matrices each (per-frame rotation and new_K,
 * calibrate_video.py:351-370).  Element i is bit-identical to gf_make_frame_xform(K, R[i], P[i]). */
GF_API int gf_make_frame_xforms(const double K_host[9], const double* R_host, const double* P_host, size_t n,
                         gf_frame_xform* out_host);

/* cv2.fisheye.initUndistortRectifyMap(K, D, R, P, (out_w,out_h), cv2.CV_16SC2)
 *   replaces calibrate_video.py:383 (and :322 with R = I).  map1: int16 [out_h][out_w][2],
 *   map2: uint16 [out_h][out_w] = (fy<<5)|fx, INTER_BITS = 5.  xf is a HOST pointer. */
GF_API int gf_fisheye_maps_16sc2(void* ctx, const double D_host[4], const gf_frame_xform* xf_host, int out_w, int out_h,
                          int16_t* map1, size_t map1_step, uint16_t* map2, size_t map2_step, gf_stream_t stream);

/* cv2.fisheye.initUndistortRectifyMap(..., cv2.CV_32FC1) -- calibrate_video.py:376 (stretched presets). */
GF_API int gf_fisheye_maps_32f(void* ctx, const double D_host[4], const gf_frame_xform* xf_host, int out_w, int out_h,
                        float* mapx, size_t mapx_step, float* mapy, size_t mapy_step, gf_stream_t stream);

/* cv2.convertMaps(mapx * sx_num / sx_den, mapy * sy_num / sy_den, cv2.CV_16SC2) -- calibrate_video.py:378.
 * The two scalings are done in float32, multiply first then divide, each rounded, exactly as numpy
 * evaluates `map1 * orig_w / calib_w` on a float32 array.  Pass num = den = 1 for a plain convertMaps. */
GF_API int gf_convert_maps_32f_16sc2(void* ctx, const float* mapx, size_t mapx_step, const float* mapy, size_t mapy_step,
                              float sx_num, float sx_den, float sy_num, float sy_den, int w, int h,
                              int16_t* map1, size_t map1_step, uint16_t* map2, size_t map2_step, gf_stream_t stream);

/* cv2.remap(src, map1, map2, cv2.INTER_LINEAR, dst, borderMode, borderValue) with CV_16SC2 + CV_16UC1 maps
 *   replaces stabilizer.py:1442-1443 (also :1825, :2423, calibrate_video.py:325, gyroflow.py:275,283).
 * u8: Q15 fixed-point weights, (sum + 2^14) >> 15.  u16: float32 weights, left-to-right sum, round-half-even.
 * src and dst must not overlap.  With GF_BORDER_TRANSPARENT dst pixels whose source falls outside stay as they are.
 * border_value_host: 4 host doubles (cv2 Scalar), only [0..2] used. */
GF_API int gf_remap_linear_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep,
                       void* dst, int dw, int dh, size_t dstep,
                       const int16_t* map1, size_t map1_step, const uint16_t* map2, size_t map2_step,
                       int border, const double border_value_host[4], gf_stream_t stream);

/* cv2.remap(src_u8c3, mapx_f32, mapy_f32, cv2.INTER_CUBIC, borderMode, borderValue)
 *   replaces nonlinear_stretch.py:130 (NonlinearStretch.apply_stretch). */
GF_API int gf_remap_cubic_u8c3_f32maps(void* ctx, const uint8_t* src, int sw, int sh, size_t sstep,
                                uint8_t* dst, int dw, int dh, size_t dstep,
                                const float* mapx, size_t mapx_step, const float* mapy, size_t mapy_step,
                                int border, const double border_value_host[4], gf_stream_t stream);

/* NonlinearStretch.apply_stretch (nonlinear_stretch.py:120-130) for a batch of frames, with the maps recompute_maps()
 * builds (nonlinear_stretch.py:58-100) passed in their separable form: map1 depends on the column only, map2 on the row
 * only, so mapx_row = map1[0, :] (dw floats) and mapy_col = map2[:, 0] (dh floats), both DEVICE pointers.  Same
 * arithmetic and result as gf_remap_cubic_u8c3_f32maps with the full maps; no 2-D map is read (8 bytes per pixel less).
 * Frame i reads src + i*src_frame_stride and writes dst + i*dst_frame_stride. */
GF_API int gf_superview_cubic_u8c3(void* ctx, const uint8_t* src, int sw, int sh, size_t sstep, size_t src_frame_stride,
                                   uint8_t* dst, int dw, int dh, size_t dstep, size_t dst_frame_stride,
                                   const float* mapx_row, const float* mapy_col, int n_frames,
                                   int border, const double border_value_host[4], gf_stream_t stream);

/* The fused hot path: for each of n_frames frames, get_maps() + cv2.remap(INTER_LINEAR) in one kernel; the
 * fixed-point map is computed per output pixel in registers and never stored
 *   replaces calibrate_video.py:383 + stabilizer.py:1442-1443 (the body of the render loop, stabilizer.py:1423-1443).
 * Frame i reads src + i*src_frame_stride, writes dst + i*dst_frame_stride (strides in bytes; 0 is allowed for src
 * to reuse one input) and uses xf_dev[i].  xf_dev is a DEVICE array of n_frames gf_frame_xform.
 * Output is bit-identical to gf_fisheye_maps_16sc2 followed by gf_remap_linear_c3. */
GF_API int gf_undistort_rotate_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t src_frame_stride,
                           void* dst, int dw, int dh, size_t dstep, size_t dst_frame_stride,
                           const double D_host[4], const gf_frame_xform* xf_dev, int n_frames,
                           int border, const double border_value_host[4], gf_stream_t stream);

/* cv2.rotate codes as the reference passes them (`video_rotation`, stabilizer.py:89,153-164): -1 means "do not rotate". */
#define GF_ROTATE_90_CLOCKWISE        0   /* cv2.ROTATE_90_CLOCKWISE        */
#define GF_ROTATE_180                 1   /* cv2.ROTATE_180                 */
#define GF_ROTATE_90_COUNTERCLOCKWISE 2   /* cv2.ROTATE_90_COUNTERCLOCKWISE */

/* cv2.rotate(frame, rotate_code) for a batch of frames
 *   replaces stabilizer.py:1385-1386 (`frame = cv2.rotate(frame, self.video_rotate_code)` in the render loop; also :1352-1353).
 * src: n_frames stored frames of sw x sh; dst: sh x sw for the two 90 degree codes, sw x sh for ROTATE_180 (the caller
 * sizes it; dstep >= a rotated row).  Pure index work: bit-exact. */
GF_API int gf_rotate_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t src_frame_stride,
                 void* dst, size_t dstep, size_t dst_frame_stride, int n_frames, int rotate_code, gf_stream_t stream);

/* The fused hot path on a source the reference would have rotated first:
 *   replaces stabilizer.py:1385-1386 (cv2.rotate) + calibrate_video.py:383 + stabilizer.py:1442-1443 in one kernel.
 * src holds the STORED frames (sw x sh, as decoded); the kernel remaps the frame cv2.rotate(frame, rotate_code) would produce
 * (sh x sw for the 90 degree codes) without building it: the rotation is an index permutation of the four taps, folded into
 * the tap addresses and the choice of weights.  Output is bit-identical to gf_rotate_c3 followed by gf_undistort_rotate_c3.
 * Runs on the RP kernel (the default mode) only: GF_ERR_UNSUPPORTED in the DIRECT / RL modes -- rotate with gf_rotate_c3 then. */
GF_API int gf_undistort_rotate_rot_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t src_frame_stride,
                               void* dst, int dw, int dh, size_t dstep, size_t dst_frame_stride,
                               const double D_host[4], const gf_frame_xform* xf_dev, int n_frames, int rotate_code,
                               int border, const double border_value_host[4], gf_stream_t stream);

/* The fused hot path for STRETCHED presets (input_horizontal_stretch != 1): get_maps' other branch,
 *   replaces calibrate_video.py:376-380 (CV_32FC1 maps, `map1 * orig_w / calib_w`, `map2 * orig_h / calib_h` in float32,
 *   cv2.convertMaps) + stabilizer.py:1442-1443, for a batch of frames.
 * Per pixel the coordinate goes through float32, is multiplied by s*_num and divided by s*_den in float32 (two roundings
 * each, as numpy evaluates it) and rounded to Q5 like cv::convertMaps.  Output is bit-identical to gf_fisheye_maps_32f +
 * gf_convert_maps_32f_16sc2 + gf_remap_linear_c3.  Other arguments as gf_undistort_rotate_c3. */
GF_API int gf_undistort_rotate_stretched_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t src_frame_stride,
                                     void* dst, int dw, int dh, size_t dstep, size_t dst_frame_stride,
                                     const double D_host[4], const gf_frame_xform* xf_dev, int n_frames,
                                     float sx_num, float sx_den, float sy_num, float sy_den,
                                     int border, const double border_value_host[4], gf_stream_t stream);

/* BASELINE config C4 in one launch: gf_undistort_rotate_c3 (u8) into an intermediate image of mid_w x mid_h, then
 * gf_superview_cubic_u8c3 (BORDER_CONSTANT 0, as NonlinearStretch.apply_stretch calls cv2.remap) to dst -- with the
 * intermediate image living tile by tile in LDS instead of HBM
 *   replaces stabilizer.py:1423-1443 followed by nonlinear_stretch.py:130 on the frame it produced.
 * Output is bit-identical to the two calls (the uint8 rounding between the two resampling steps is kept).
 * mapx_row_host / mapy_col_host: HOST arrays of dw / dh floats (map1[0, :] and map2[:, 0] of recompute_maps,
 * nonlinear_stretch.py:58-100); their fixed-point form is cached on the device per context (the first call with new maps
 * waits for a small upload).  `border` / border_value_host apply to the remap step.  GF_ERR_UNSUPPORTED (use the two calls)
 * unless the vertical map is a unit-step shift -- output row y reads intermediate rows y + c .. y + c + 3 with one fraction
 * for all rows, which is what recompute_maps gives when the output height equals the intermediate height -- and the
 * horizontal map is monotone; GF_BORDER_TRANSPARENT is unsupported as well. */
GF_API int gf_undistort_rotate_superview_u8c3(void* ctx, const uint8_t* src, int sw, int sh, size_t sstep, size_t src_frame_stride,
                                       int mid_w, int mid_h,
                                       uint8_t* dst, int dw, int dh, size_t dstep, size_t dst_frame_stride,
                                       const double D_host[4], const gf_frame_xform* xf_dev, int n_frames,
                                       const float* mapx_row_host, const float* mapy_col_host,
                                       int border, const double border_value_host[4], gf_stream_t stream);

/* gf_undistort_rotate_c3 for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  (No counterpart in the reference.)  Same arithmetic, same kernels and bit-identical output, but no
 * context: the device memory the kernel needs beyond src and dst (lens table, RP table, the batch's transforms: 28 KB)
 * lives in a static device array inside the library, which every GPU the library runs on holds until the process ends.
 * The call allocates and frees nothing, copies nothing with hipMemcpy and never synchronises: the tables and xf_host (a
 * HOST array of n_frames transforms, read before the call returns) reach the device as kernel arguments of small
 * launches queued on `stream` ahead of the fused kernel.  n_frames <= GF_NOALLOC_MAX_FRAMES, else GF_ERR_INVALID_ARG.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two threads) race;
 * calls queued on one stream do not.  Kernel variant: RP (DIRECT if the process runs with GF_K1=direct).
 * The first call on a device resolves the array's address (hipGetSymbolAddress, which loads the library's code object on that
 * device -- the runtime may allocate there): make one warm-up call per device before capturing a graph.  Host cost of a call:
 * about 14 argument-carrying launches of <= 2 KB each; the host copy of the lens table is rebuilt (about 1 ms) only when
 * D_host differs from the previous call's.  The calls of one process are serialised by a mutex. */
#define GF_NOALLOC_MAX_FRAMES 64
GF_API int gf_undistort_rotate_c3_noalloc(int device, int depth, const void* src, int sw, int sh, size_t sstep, size_t src_frame_stride,
                                   void* dst, int dw, int dh, size_t dstep, size_t dst_frame_stride,
                                   const double D_host[4], const gf_frame_xform* xf_host, int n_frames,
                                   int border, const double border_value_host[4], gf_stream_t stream);

/* Hyperlapse blend, replaces the numpy float64 passes at stabilizer.py:1434 (reset), :1447 (accumulate)
 * and :1468 (truncating cast).  acc is double [n] with n = w*h*3. */
GF_API int gf_blend_accumulate(void* ctx, int depth, const void* frame, double* acc, double weight, int reset,
                        size_t n, gf_stream_t stream);
GF_API int gf_blend_finalize(void* ctx, int depth, const double* acc, void* out, size_t n, gf_stream_t stream);

/* Fused hot path + hyperlapse blend: output frame g = trunc( sum_{b<n_blend} (1/n_blend) * remap(frame g*group_stride+b) ),
 * accumulated in float64 in registers in the reference's order (stabilizer.py:1431-1447,1464-1468); the
 * intermediate remapped frames and the float64 accumulator never touch HBM.
 * src holds n_out*... input frames: input frame index for (g,b) is g*frames_per_group + b. */
GF_API int gf_undistort_rotate_blend_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t src_frame_stride,
                                 void* dst, int dw, int dh, size_t dstep, size_t dst_frame_stride,
                                 const double D_host[4], const gf_frame_xform* xf_dev, int n_out, int frames_per_group, int n_blend,
                                 int border, const double border_value_host[4], gf_stream_t stream);

/* cv2.fisheye.undistortPoints(pts, K, D, R=R_s, P=P) for n_sets rotations R_s of ONE list of n_pts distorted points
 *   replaces adaptive_zoom.py:220 (AdaptiveZoom.boundingPolygon, called once per frame at adaptive_zoom.py:104).
 * pts: DEVICE double [n_pts][2]; PR: DEVICE double [n_sets][9], row-major P*R_s (the matrix OpenCV forms before its loop;
 * pass P when R is absent); out: DEVICE double [n_sets][n_pts][2].  Points whose Newton iteration does not converge or
 * flips sign come back as (-1e6, -1e6), as in OpenCV 4.5.2. */
GF_API int gf_fisheye_undistort_points(void* ctx, const double* pts, size_t n_pts, const double K_host[9], const double D_host[4],
                                       const double* PR, size_t n_sets, double* out, gf_stream_t stream);

/* HOST (no GPU work): the sequential quaternion recurrences that produce the per-frame stabilisation quaternions, the
 * hot path's other per-frame input.  They stay on the CPU; these entries only take the Python
 * interpreter out of the per-sample loops.  Quaternions are (w,x,y,z) doubles; all pointers are host pointers.
 *   gf_host_integrate_gyro       gyro_integrator.py:125-179 (GyroIntegrator.integrate_all, gyro only): gyro = n rows of
 *                                [time s, wx, wy, wz rad/s]; out_q = n orientations starting from q0.
 *   gf_host_interp_orientations  gyro_integrator.py:262-301 (get_interpolated_orientations): resamples at start + k*interval;
 *                                returns the number of samples (call with NULL outputs to size the arrays), < 0 on error.
 *   gf_host_plain_slerp          smoothing_algos.py:235-246 (PlainSlerp's forward + reverse exponential slerp), given alpha.
 *   gf_host_rot_between          quaternion.py:83-87 row-wise: out[i] = inverse(q1[i]) * q2[i] (gyro_integrator.py:323-325). */
GF_API int gf_host_integrate_gyro(const double* gyro_host, size_t n, const double q0_host[4], double* out_q_host);
GF_API long long gf_host_interp_orientations(const double* times_host, const double* quats_host, size_t n, double start, double interval,
                                             double* out_times_host, double* out_quats_host, size_t capacity);
GF_API int gf_host_plain_slerp(const double* quats_host, size_t n, double alpha, double* out_host);
GF_API int gf_host_rot_between(const double* q1_host, const double* q2_host, size_t n, double* out_host);

/* Test hook (no counterpart in the reference): the constant table of the RP coordinate engine, GF_RP_TABLE_DOUBLES doubles:
 * the 8 x 8 inverse Vandermonde matrix of the Chebyshev nodes (row-major), the 8 nodes, and the 64 x 8 Lagrange weights of
 * the tile rows.  Host only. */
#define GF_RP_TABLE_DOUBLES (8*8 + 8 + 64*8)
GF_API int gf_get_rp_table(double* out_host);

/* HOST (no GPU work): the kernels' table approximation of the radial scale S(t) = theta_d(atan sqrt t)/sqrt t of
 * cv::fisheye::initUndistortRectifyMap, evaluated exactly as on the device, for n values t = x^2 + y^2 in [0, 255)
 * (NaN outside: there the kernels use the exact formula).  Lets CPU tests bound the table's error. */
GF_API int gf_lens_scale_host(const double D_host[4], const double* t_host, size_t n, double* out_host);

/* HOST: copy out the 1024x16 int16 bicubic weight table the library uses (tests compare it with the oracle's). */
GF_API int gf_get_cubic_table(int16_t* out_host);

#ifdef __cplusplus
}
#endif
#endif /* GFREMAP_H */
