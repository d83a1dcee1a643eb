/*
 * oracle/cv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar) of the arithmetic the reference's hot path
 * executes inside opencv-python 4.5.2.54 (pinned at /root/reference/poetry.lock:195-196):
 *
 *   calibrate_video.py:322,376,383  cv2.fisheye.initUndistortRectifyMap   -> orc_fisheye_maps()
 *   calibrate_video.py:378          cv2.convertMaps(32F,32F -> 16SC2)     -> orc_convert_maps_32f_16sc2()
 *   stabilizer.py:1442,1825,2423    cv2.remap(INTER_LINEAR, 16SC2 maps)   -> orc_remap_linear_c3()
 *   nonlinear_stretch.py:130        cv2.remap(INTER_CUBIC, 32FC1 maps)    -> orc_remap_cubic_u8c3_f32maps()
 *   adaptive_zoom.py:220            cv2.fisheye.undistortPoints           -> orc_undistort_points()
 *   stabilizer.py:1434,1447,1468    float64 hyperlapse blend              -> orc_blend_*()
 *
 * PARITY UNPINNED: OpenCV's sources are a third-party dependency that is not under
 * /root/reference, cv2 is not importable in the build container, and none of the
 * reference's own files hold a map entry or an output pixel.  This file
 * restates OpenCV 4.5.2's modules/calib3d/src/fisheye.cpp and modules/imgproc/src/imgwarp.cpp
 * from their published algorithm; every behavioural quirk is isolated (Q1..Q6 below) and
 * has its own unit test in tests/test_oracle.py so it can be flipped if a real cv2 disagrees.
 * The one exception: orc_undistort_points() is pinned (to ~1e-4, the precision the file stores) by the fcorr column
 * and min_fcorr of the reference's plugins/blender/GX016015.MP4.gyroflow, which real cv2 produced
 * (tests/test_oracle.py::test_undistort_points_pinned_by_gyroflow_sample).  Maps and pixels remain unpinned.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: see oracle/Makefile  (-O2 -ffp-contract=off: OpenCV's baseline x86-64 build has
 * no FMA contraction and the u16 / cubic-table paths depend on float rounding order).
 */
#include <emmintrin.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <limits.h>

#define ORC_API __attribute__((visibility("default")))

enum { INTER_BITS = 5, INTER_TAB_SIZE = 32, INTER_TAB_SIZE2 = 1024,
       INTER_REMAP_COEF_BITS = 15, INTER_REMAP_COEF_SCALE = 32768 };
enum { ORC_BORDER_CONSTANT = 0, ORC_BORDER_REPLICATE = 1, ORC_BORDER_TRANSPARENT = 5 };
enum { ORC_CV_32FC1 = 5, ORC_CV_16SC2 = 11 };

/* cvRound on x86-64 = cvtsd2si / cvtss2si: round-half-to-even, INT_MIN on NaN/overflow. */
static inline int cv_round_d(double v) { return _mm_cvtsd_si32(_mm_set_sd(v)); }
static inline int cv_round_f(float v)  { return _mm_cvtss_si32(_mm_set_ss(v)); }
static inline short sat_s16(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static inline uint16_t sat_u16(int v) { return (uint16_t)(v < 0 ? 0 : v > 65535 ? 65535 : v); }

ORC_API int orc_cv_round_d(double v) { return cv_round_d(v); }
ORC_API int orc_cv_round_f(float v) { return cv_round_f(v); }

/* ------------------------------------------------------------------------------------ */
/* 3x3 helpers                                                                           */

/* Matx33d * Matx33d: s = 0; s += a(i,k)*b(k,j), k ascending (cv::Matx MatxMulOp).        */
ORC_API void orc_matmul3(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a[i*3+k] * b[k*3+j];
            c[i*3+j] = s;
        }
}

/* OpenCV inverts with DECOMP_SVD (Jacobi or LAPACK, build dependent), which cannot be
 * reproduced bit-for-bit; the oracle returns the adjugate inverse evaluated in 80-bit
 * long double and rounded once to double (agrees with any backward-stable f64 inverse to
 * a few ulp => <1e-12 px on the maps).                                                   */
ORC_API int orc_invert3(const double* m, double* inv)
{
    long double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    long double A = e*i - f*h, B = -(d*i - f*g), C = d*h - e*g;
    long double det = a*A + b*B + c*C;
    if (det == 0) return -1;
    long double r = 1.0L / det;
    inv[0] = (double)(A*r);            inv[1] = (double)(-(b*i - c*h)*r); inv[2] = (double)((b*f - c*e)*r);
    inv[3] = (double)(B*r);            inv[4] = (double)((a*i - c*g)*r);  inv[5] = (double)(-(a*f - c*d)*r);
    inv[6] = (double)(C*r);            inv[7] = (double)(-(a*h - b*g)*r); inv[8] = (double)((a*e - b*d)*r);
    return 0;
}

/* iR = (P[:, :3] * R)^-1  -- fisheye.cpp initUndistortRectifyMap */
ORC_API int orc_compute_iR(const double* R, const double* P, double* iR)
{
    double PR[9];
    orc_matmul3(P, R, PR);
    return orc_invert3(PR, iR);
}

/* ------------------------------------------------------------------------------------ */
/* cv::fisheye::initUndistortRectifyMap (CV_16SC2 or CV_32FC1)                            */
/*  flags bit0: Q2 -- apply the `_w <= 0 -> +-inf` branch of newer 4.5.x builds.          */
/*  iR_in: when non-NULL use it instead of inverting P*R (lets tests share one inverse).  */
/*  nthreads: rows are independent (each starts from i*iR[.1] + iR[.2]), so striping them over threads
 *  gives the same bits; OpenCV itself runs this loop on one thread (nthreads = 1 is the faithful
 *  timing, > 1 is bench.py's "best CPU" figure).                                          */
static int fisheye_maps_impl(const double* K, const double* D, const double* R, const double* P,
                             const double* iR_in, int width, int height, int m1type, int flags,
                             void* map1, size_t map1_step, void* map2, size_t map2_step, int nthreads)
{
    if (m1type != ORC_CV_16SC2 && m1type != ORC_CV_32FC1) return -2;
    const double f0 = K[0], f1 = K[4], c0 = K[2], c1 = K[5];
    const double k0 = D[0], k1 = D[1], k2 = D[2], k3 = D[3];
    double iR[9];
    if (iR_in) memcpy(iR, iR_in, sizeof iR);
    else if (orc_compute_iR(R, P, iR)) return -1;
    const int wcheck = flags & 1;
    if (nthreads < 1) nthreads = 1;

    #pragma omp parallel for num_threads(nthreads) schedule(static) if(nthreads > 1)
    for (int i = 0; i < height; ++i) {
        float* m1f = (float*)((char*)map1 + (size_t)i * map1_step);
        float* m2f = (float*)((char*)map2 + (size_t)i * map2_step);
        short* m1 = (short*)m1f;
        uint16_t* m2 = (uint16_t*)m2f;

        double _x = i*iR[1] + iR[2], _y = i*iR[4] + iR[5], _w = i*iR[7] + iR[8];

        for (int j = 0; j < width; ++j) {
            double u, v;
            if (wcheck && _w <= 0) {
                u = (_x > 0) ? -INFINITY : INFINITY;
                v = (_y > 0) ? -INFINITY : INFINITY;
            } else {
                double x = _x/_w, y = _y/_w;
                double r = sqrt(x*x + y*y);
                double theta = atan(r);
                double theta2 = theta*theta, theta4 = theta2*theta2, theta6 = theta4*theta2, theta8 = theta4*theta4;
                double theta_d = theta * (1 + k0*theta2 + k1*theta4 + k2*theta6 + k3*theta8);
                double scale = (r == 0) ? 1.0 : theta_d / r;     /* Q3 */
                u = f0*x*scale + c0;
                v = f1*y*scale + c1;
            }
            if (m1type == ORC_CV_16SC2) {
                int iu = cv_round_d(u*INTER_TAB_SIZE);
                int iv = cv_round_d(v*INTER_TAB_SIZE);
                m1[j*2+0] = (short)(iu >> INTER_BITS);            /* Q1: wrapping cast, not saturate */
                m1[j*2+1] = (short)(iv >> INTER_BITS);
                m2[j] = (uint16_t)((iv & (INTER_TAB_SIZE-1))*INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE-1)));
            } else {
                m1f[j] = (float)u;
                m2f[j] = (float)v;
            }
            _x += iR[0]; _y += iR[3]; _w += iR[6];                /* incremental, f64 */
        }
    }
    return 0;
}

ORC_API int orc_fisheye_maps(const double* K, const double* D, const double* R, const double* P,
                             const double* iR_in, int width, int height, int m1type, int flags,
                             void* map1, size_t map1_step, void* map2, size_t map2_step)
{
    return fisheye_maps_impl(K, D, R, P, iR_in, width, height, m1type, flags, map1, map1_step, map2, map2_step, 1);
}

ORC_API int orc_fisheye_maps_mt(const double* K, const double* D, const double* R, const double* P,
                                const double* iR_in, int width, int height, int m1type, int flags,
                                void* map1, size_t map1_step, void* map2, size_t map2_step, int nthreads)
{
    return fisheye_maps_impl(K, D, R, P, iR_in, width, height, m1type, flags, map1, map1_step, map2, map2_step, nthreads);
}

/* cv::convertMaps(CV_32FC1, CV_32FC1 -> CV_16SC2 + CV_16UC1), nninterpolate=false        */
ORC_API int orc_convert_maps_32f_16sc2(const float* mx, size_t mx_step, const float* my, size_t my_step,
                                       int width, int height,
                                       short* map1, size_t map1_step, uint16_t* map2, size_t map2_step)
{
    for (int y = 0; y < height; y++) {
        const float* sx = (const float*)((const char*)mx + y*mx_step);
        const float* sy = (const float*)((const char*)my + y*my_step);
        short* d1 = (short*)((char*)map1 + y*map1_step);
        uint16_t* d2 = (uint16_t*)((char*)map2 + y*map2_step);
        for (int x = 0; x < width; x++) {
            int ix = cv_round_f(sx[x]*(float)INTER_TAB_SIZE);     /* float multiply, then round */
            int iy = cv_round_f(sy[x]*(float)INTER_TAB_SIZE);
            d1[x*2]   = sat_s16(ix >> INTER_BITS);
            d1[x*2+1] = sat_s16(iy >> INTER_BITS);
            d2[x] = (uint16_t)((iy & (INTER_TAB_SIZE-1))*INTER_TAB_SIZE + (ix & (INTER_TAB_SIZE-1)));
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Interpolation tables -- imgwarp.cpp initInterTab1D / initInterTab2D                    */

static void interpolate_linear(float x, float* c) { c[0] = 1.f - x; c[1] = x; }

static void interpolate_cubic(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A*(x + 1) - 5*A)*(x + 1) + 8*A)*(x + 1) - 4*A;
    c[1] = ((A + 2)*x - (A + 3))*x*x + 1;
    c[2] = ((A + 2)*(1 - x) - (A + 3))*(1 - x)*(1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

/* ksize = 2 (linear) or 4 (cubic).  ftab / itab hold 1024 * ksize*ksize entries.
 * The sum fix-up loop reproduces OpenCV's bounds `k1,k2 in [ksize/2, ksize/2+2)`:
 *   Q6 (cubic): the searched 2x2 block is rows/cols {2,3}, not the centre {1,2};
 *   Q4 (linear): for ksize==2 the loop indexes past the entry (k==2); those reads land
 *       in the *next*, not-yet-written entry of the zero-initialised static table, so they
 *       compare as 0.  Net effect: entry 0 becomes {32767,0,0,1}.                         */
static void init_inter_tab_2d(int ksize, float* ftab, short* itab)
{
    float t1[INTER_TAB_SIZE*4];
    const float scale = 1.f/INTER_TAB_SIZE;
    for (int i = 0; i < INTER_TAB_SIZE; i++) {
        if (ksize == 2) interpolate_linear(i*scale, t1 + i*ksize);
        else            interpolate_cubic(i*scale, t1 + i*ksize);
    }
    const int kk = ksize*ksize;
    memset(itab, 0, sizeof(short)*(INTER_TAB_SIZE2*kk + 2*kk));   /* incl. padding */
    for (int i = 0; i < INTER_TAB_SIZE; i++)
        for (int j = 0; j < INTER_TAB_SIZE; j++) {
            float* tab = ftab + (i*INTER_TAB_SIZE + j)*kk;
            short* it = itab + (i*INTER_TAB_SIZE + j)*kk;
            int isum = 0;
            for (int k1 = 0; k1 < ksize; k1++) {
                float vy = t1[i*ksize + k1];
                for (int k2 = 0; k2 < ksize; k2++) {
                    float v = vy*t1[j*ksize + k2];
                    tab[k1*ksize + k2] = v;
                    isum += it[k1*ksize + k2] = sat_s16(cv_round_f(v*INTER_REMAP_COEF_SCALE));
                }
            }
            if (isum != INTER_REMAP_COEF_SCALE) {
                int diff = isum - INTER_REMAP_COEF_SCALE;
                int ksize2 = ksize/2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (int k1 = ksize2; k1 < ksize2+2; k1++)
                    for (int k2 = ksize2; k2 < ksize2+2; k2++) {
                        /* for ksize==2 this indexes past the entry into the next, still-zero one
                         * (callers pad the table so the last entry's reads stay in bounds) */
                        short val = it[k1*ksize + k2];
                        if (val < it[mk1*ksize+mk2]) mk1 = k1, mk2 = k2;
                        else if (val > it[Mk1*ksize+Mk2]) Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0) it[Mk1*ksize + Mk2] = (short)(it[Mk1*ksize + Mk2] - diff);
                else          it[mk1*ksize + mk2] = (short)(it[mk1*ksize + mk2] - diff);
            }
        }
}

static float  g_lin_f[INTER_TAB_SIZE2*4];
static short  g_lin_i[INTER_TAB_SIZE2*4 + 2*4];
static float  g_cub_f[INTER_TAB_SIZE2*16];
static short  g_cub_i[INTER_TAB_SIZE2*16 + 2*16];
static int g_tabs_ready = 0;

static void ensure_tabs(void)
{
    if (g_tabs_ready) return;
    init_inter_tab_2d(2, g_lin_f, g_lin_i);
    init_inter_tab_2d(4, g_cub_f, g_cub_i);
    g_tabs_ready = 1;
}

ORC_API void orc_linear_tab(float* ftab, short* itab)
{
    ensure_tabs();
    if (ftab) memcpy(ftab, g_lin_f, sizeof g_lin_f);
    if (itab) memcpy(itab, g_lin_i, sizeof(short)*INTER_TAB_SIZE2*4);
}
ORC_API void orc_cubic_tab(float* ftab, short* itab)
{
    ensure_tabs();
    if (ftab) memcpy(ftab, g_cub_f, sizeof g_cub_f);
    if (itab) memcpy(itab, g_cub_i, sizeof(short)*INTER_TAB_SIZE2*16);
}

/* ------------------------------------------------------------------------------------ */
/* cv::remap INTER_LINEAR, 3 channels, map1 CV_16SC2 + map2 CV_16UC1 (imgwarp.cpp
 * RemapInvoker + remapBilinear).  depth: 0 = CV_8U (fixed-point Q15 weights),
 * 2 = CV_16U (float32 weights, left-to-right float sum, cvRound).
 * dst must already hold the previous contents for BORDER_TRANSPARENT.
 * Rows [row0,row1) only -- lets the timed baseline stripe rows over threads exactly like
 * cv::parallel_for_ does; results do not depend on the striping.                         */
static void remap_linear_rows(int depth, const void* src_, int sw, int sh, size_t sstep,
                              void* dst_, int dw, size_t dstep,
                              const short* map1, size_t m1step, const uint16_t* map2, size_t m2step,
                              int border, const double* bv, int row0, int row1)
{
    const int cn = 3;
    uint8_t cval8[3]; uint16_t cval16[3];
    for (int k = 0; k < 3; k++) {
        cval8[k] = sat_u8(cv_round_d(bv[k]));
        cval16[k] = sat_u16(cv_round_d(bv[k]));
    }
    for (int dy = row0; dy < row1; dy++) {
        const short* XY = (const short*)((const char*)map1 + dy*m1step);
        const uint16_t* FXY = (const uint16_t*)((const char*)map2 + dy*m2step);
        for (int dx = 0; dx < dw; dx++) {
            int sx = XY[dx*2], sy = XY[dx*2+1];
            unsigned a = FXY[dx] & (INTER_TAB_SIZE2-1);
            int x0, x1, y0, y1;             /* tap coordinates, -1 = use cval */
            if ((unsigned)sx < (unsigned)(sw > 1 ? sw-1 : 0) && (unsigned)sy < (unsigned)(sh > 1 ? sh-1 : 0)) {
                x0 = sx; x1 = sx+1; y0 = sy; y1 = sy+1;                     /* inlier: Q5 */
            } else if (border == ORC_BORDER_CONSTANT) {
                if (sx >= sw || sx+1 < 0 || sy >= sh || sy+1 < 0) {
                    if (depth == 0) { uint8_t* D = (uint8_t*)dst_ + dy*dstep + dx*cn; D[0]=cval8[0]; D[1]=cval8[1]; D[2]=cval8[2]; }
                    else { uint16_t* D = (uint16_t*)((char*)dst_ + dy*dstep) + dx*cn; D[0]=cval16[0]; D[1]=cval16[1]; D[2]=cval16[2]; }
                    continue;
                }
                x0 = (sx >= 0 && sx < sw) ? sx : -1;   x1 = (sx+1 >= 0 && sx+1 < sw) ? sx+1 : -1;
                y0 = (sy >= 0 && sy < sh) ? sy : -1;   y1 = (sy+1 >= 0 && sy+1 < sh) ? sy+1 : -1;
            } else if (border == ORC_BORDER_TRANSPARENT) {
                if ((unsigned)sx >= (unsigned)(sw-1) || (unsigned)sy >= (unsigned)(sh-1)) continue;
                x0 = sx; x1 = sx+1; y0 = sy; y1 = sy+1;
            } else { /* REPLICATE: borderInterpolate clamps */
                x0 = sx < 0 ? 0 : sx >= sw ? sw-1 : sx;       x1 = sx+1 < 0 ? 0 : sx+1 >= sw ? sw-1 : sx+1;
                y0 = sy < 0 ? 0 : sy >= sh ? sh-1 : sy;       y1 = sy+1 < 0 ? 0 : sy+1 >= sh ? sh-1 : sy+1;
            }
            if (depth == 0) {
                const short* w = g_lin_i + a*4;
                const uint8_t* S = (const uint8_t*)src_;
                uint8_t* D = (uint8_t*)dst_ + dy*dstep + dx*cn;
                const uint8_t* v0 = (x0 >= 0 && y0 >= 0) ? S + y0*sstep + x0*cn : cval8;
                const uint8_t* v1 = (x1 >= 0 && y0 >= 0) ? S + y0*sstep + x1*cn : cval8;
                const uint8_t* v2 = (x0 >= 0 && y1 >= 0) ? S + y1*sstep + x0*cn : cval8;
                const uint8_t* v3 = (x1 >= 0 && y1 >= 0) ? S + y1*sstep + x1*cn : cval8;
                for (int k = 0; k < cn; k++) {
                    int t = v0[k]*w[0] + v1[k]*w[1] + v2[k]*w[2] + v3[k]*w[3];
                    D[k] = sat_u8((t + (1 << (INTER_REMAP_COEF_BITS-1))) >> INTER_REMAP_COEF_BITS);
                }
            } else {
                const float* w = g_lin_f + a*4;
                const char* S = (const char*)src_;
                uint16_t* D = (uint16_t*)((char*)dst_ + dy*dstep) + dx*cn;
                const uint16_t* v0 = (x0 >= 0 && y0 >= 0) ? (const uint16_t*)(S + y0*sstep) + x0*cn : cval16;
                const uint16_t* v1 = (x1 >= 0 && y0 >= 0) ? (const uint16_t*)(S + y0*sstep) + x1*cn : cval16;
                const uint16_t* v2 = (x0 >= 0 && y1 >= 0) ? (const uint16_t*)(S + y1*sstep) + x0*cn : cval16;
                const uint16_t* v3 = (x1 >= 0 && y1 >= 0) ? (const uint16_t*)(S + y1*sstep) + x1*cn : cval16;
                for (int k = 0; k < cn; k++) {
                    float t = v0[k]*w[0] + v1[k]*w[1] + v2[k]*w[2] + v3[k]*w[3];   /* float, no FMA */
                    D[k] = sat_u16(cv_round_f(t));
                }
            }
        }
    }
}

ORC_API int orc_remap_linear_c3(int depth, const void* src, int sw, int sh, size_t sstep,
                                void* dst, int dw, int dh, size_t dstep,
                                const short* map1, size_t m1step, const uint16_t* map2, size_t m2step,
                                int border, const double* bv, int nthreads)
{
    if (depth != 0 && depth != 2) return -2;
    if (border != ORC_BORDER_CONSTANT && border != ORC_BORDER_REPLICATE && border != ORC_BORDER_TRANSPARENT) return -3;
    if (dw >= SHRT_MAX || dh >= SHRT_MAX || sw >= SHRT_MAX || sh >= SHRT_MAX || sw <= 0 || sh <= 0) return -4;
    ensure_tabs();
    if (nthreads <= 1) {
        remap_linear_rows(depth, src, sw, sh, sstep, dst, dw, dstep, map1, m1step, map2, m2step, border, bv, 0, dh);
    } else {
        #pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int s = 0; s < nthreads; s++) {
            int r0 = (int)((long long)dh * s / nthreads), r1 = (int)((long long)dh * (s+1) / nthreads);
            remap_linear_rows(depth, src, sw, sh, sstep, dst, dw, dstep, map1, m1step, map2, m2step, border, bv, r0, r1);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* cv::remap INTER_CUBIC, CV_8UC3, map1/map2 CV_32FC1 (RemapInvoker planar branch +
 * remapBicubic<FixedPtCast<int,uchar,15>, short, 32768>).                                */
static inline int border_interp(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (border == ORC_BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (border == 4 /* REFLECT_101, used for TRANSPARENT */) {
        if (len == 1) return 0;
        do { if (p < 0) p = -p; else p = len - 1 - (p - len) - 1; } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    return -1; /* CONSTANT */
}

static void remap_cubic_rows(const uint8_t* S0, int sw, int sh, size_t sstep, uint8_t* dst, int dw, size_t dstep,
                             const float* mx, size_t mxstep, const float* my, size_t mystep,
                             int border, const double* bv, int row0, int row1)
{
    const int cn = 3;
    const int border1 = border != ORC_BORDER_TRANSPARENT ? border : 4;
    int cval[3];
    for (int k = 0; k < 3; k++) cval[k] = sat_u8(cv_round_d(bv[k]));
    const unsigned width1 = sw-3 > 0 ? sw-3 : 0, height1 = sh-3 > 0 ? sh-3 : 0;
    for (int dy = row0; dy < row1; dy++) {
        const float* sX = (const float*)((const char*)mx + dy*mxstep);
        const float* sY = (const float*)((const char*)my + dy*mystep);
        uint8_t* D = dst + dy*dstep;
        for (int dx = 0; dx < dw; dx++, D += cn) {
            int isx = cv_round_f(sX[dx]*(float)INTER_TAB_SIZE);
            int isy = cv_round_f(sY[dx]*(float)INTER_TAB_SIZE);
            int a = (isy & (INTER_TAB_SIZE-1))*INTER_TAB_SIZE + (isx & (INTER_TAB_SIZE-1));
            int sx = sat_s16(isx >> INTER_BITS) - 1, sy = sat_s16(isy >> INTER_BITS) - 1;
            const short* w = g_cub_i + a*16;
            if ((unsigned)sx < width1 && (unsigned)sy < height1) {
                const uint8_t* S = S0 + sy*sstep + sx*cn;
                for (int k = 0; k < cn; k++) {
                    int sum = 0;
                    for (int r = 0; r < 4; r++)
                        sum += S[r*sstep+k]*w[r*4] + S[r*sstep+cn+k]*w[r*4+1] + S[r*sstep+cn*2+k]*w[r*4+2] + S[r*sstep+cn*3+k]*w[r*4+3];
                    D[k] = sat_u8((sum + (1 << 14)) >> 15);
                }
            } else {
                int x[4], y[4];
                if (border == ORC_BORDER_TRANSPARENT &&
                    ((unsigned)(sx+1) >= (unsigned)sw || (unsigned)(sy+1) >= (unsigned)sh))
                    continue;
                if (border1 == ORC_BORDER_CONSTANT && (sx >= sw || sx+4 <= 0 || sy >= sh || sy+4 <= 0)) {
                    for (int k = 0; k < cn; k++) D[k] = (uint8_t)cval[k];
                    continue;
                }
                for (int i = 0; i < 4; i++) {
                    x[i] = border_interp(sx + i, sw, border1)*cn;
                    y[i] = border_interp(sy + i, sh, border1);
                }
                for (int k = 0; k < cn; k++) {
                    int cv = cval[k], sum = cv*INTER_REMAP_COEF_SCALE;
                    for (int i = 0; i < 4; i++) {
                        int yi = y[i];
                        if (yi < 0) continue;
                        const uint8_t* S = S0 + yi*sstep + k;
                        for (int q = 0; q < 4; q++)
                            if (x[q] >= 0) sum += (S[x[q]] - cv)*w[i*4+q];
                    }
                    D[k] = sat_u8((sum + (1 << 14)) >> 15);
                }
            }
        }
    }
}

ORC_API int orc_remap_cubic_u8c3_f32maps(const uint8_t* src, int sw, int sh, size_t sstep,
                                         uint8_t* dst, int dw, int dh, size_t dstep,
                                         const float* mx, size_t mxstep, const float* my, size_t mystep,
                                         int border, const double* bv, int nthreads)
{
    if (border != ORC_BORDER_CONSTANT && border != ORC_BORDER_REPLICATE && border != ORC_BORDER_TRANSPARENT) return -3;
    if (dw >= SHRT_MAX || dh >= SHRT_MAX || sw >= SHRT_MAX || sh >= SHRT_MAX || sw <= 0 || sh <= 0) return -4;
    ensure_tabs();
    if (nthreads <= 1) {
        remap_cubic_rows(src, sw, sh, sstep, dst, dw, dstep, mx, mxstep, my, mystep, border, bv, 0, dh);
    } else {
        #pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int s = 0; s < nthreads; s++) {
            int r0 = (int)((long long)dh * s / nthreads), r1 = (int)((long long)dh * (s+1) / nthreads);
            remap_cubic_rows(src, sw, sh, sstep, dst, dw, dstep, mx, mxstep, my, mystep, border, bv, r0, r1);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* cv::fisheye::undistortPoints (fisheye.cpp, 4.5.2) for N points, f64.                   */
ORC_API int orc_undistort_points(const double* pts, int n, const double* K, const double* D,
                                 const double* R, const double* P, double* out)
{
    const double f0 = K[0], f1 = K[4], c0 = K[2], c1 = K[5];
    double RR[9];
    orc_matmul3(P, R, RR);                       /* RR = PP * RR */
    for (int i = 0; i < n; i++) {
        double pix = pts[2*i], piy = pts[2*i+1];
        double pwx = (pix - c0)/f0, pwy = (piy - c1)/f1;
        double scale = 1.0;
        double theta_d = sqrt(pwx*pwx + pwy*pwy);
        /* the current camera model is only valid up to 180 FOV; clip */
        theta_d = fmin(fmax(-M_PI/2., theta_d), M_PI/2.);
        int converged = 0;
        double theta = theta_d;
        if (theta_d > 1e-8) {
            for (int j = 0; j < 10; j++) {
                double theta2 = theta*theta, theta4 = theta2*theta2, theta6 = theta4*theta2, theta8 = theta6*theta2;
                double k0_theta2 = D[0]*theta2, k1_theta4 = D[1]*theta4, k2_theta6 = D[2]*theta6, k3_theta8 = D[3]*theta8;
                double theta_fix = (theta*(1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                   (1 + 3*k0_theta2 + 5*k1_theta4 + 7*k2_theta6 + 9*k3_theta8);
                theta = theta - theta_fix;
                if (fabs(theta_fix) < 1e-8) { converged = 1; break; }
            }
            scale = tan(theta)/theta_d;
        } else {
            converged = 1;
        }
        int theta_flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
        if (converged && !theta_flipped) {
            double pux = pwx*scale, puy = pwy*scale;
            double prx = RR[0]*pux + RR[1]*puy + RR[2];
            double pry = RR[3]*pux + RR[4]*puy + RR[5];
            double prz = RR[6]*pux + RR[7]*puy + RR[8];
            out[2*i] = prx/prz; out[2*i+1] = pry/prz;
        } else {
            out[2*i] = -1000000.0; out[2*i+1] = -1000000.0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Hyperlapse blend (stabilizer.py:1434,1447,1468): acc(f64) += (1/N)*frame ; out = trunc */
ORC_API void orc_blend_accumulate(int depth, const void* frame, double* acc, double weight, int reset, size_t n)
{
    if (reset) for (size_t i = 0; i < n; i++) acc[i] = acc[i]*0.0;
    if (depth == 0) { const uint8_t* f = (const uint8_t*)frame; for (size_t i = 0; i < n; i++) acc[i] += weight*(double)f[i]; }
    else { const uint16_t* f = (const uint16_t*)frame; for (size_t i = 0; i < n; i++) acc[i] += weight*(double)f[i]; }
}
ORC_API void orc_blend_finalize(int depth, const double* acc, void* out, size_t n)
{
    /* numpy astype(uint8/uint16) of an in-range non-negative f64: truncation toward zero */
    if (depth == 0) { uint8_t* o = (uint8_t*)out; for (size_t i = 0; i < n; i++) o[i] = (uint8_t)(long long)acc[i]; }
    else { uint16_t* o = (uint16_t*)out; for (size_t i = 0; i < n; i++) o[i] = (uint16_t)(long long)acc[i]; }
}
