// gfremap.hip -- HIP (gfx950 / CDNA4) kernels + C ABI of libgfremap.so.  See include/gfremap.h.
//
// Path replaced (reference = ElvinC/gyroflow, Python):
//   calibrate_video.py:330-384  FisheyeCalibrator.get_maps  -> cv2.fisheye.initUndistortRectifyMap
//   stabilizer.py:1423-1447     render-loop body            -> cv2.remap(INTER_LINEAR) + float64 blend
//   nonlinear_stretch.py:130    apply_stretch               -> cv2.remap(INTER_CUBIC)
//
// Design (DESIGN.md has the numbers):
//  * The fused kernel is bound by the vector-memory return path and by VALU issue, not by HBM, so the lens model is evaluated
//    as  u = fx * (X/W) * S(t) + cx,  t = (X^2+Y^2)/W^2,  S(t) = theta_d(atan sqrt t)/sqrt t,  with S read from a per-lens
//    table of degree-6 polynomials (256 pieces, geometric in 1+t, 16 KB in LDS, relative error < 2e-14) and 1/W from
//    v_rcp_f64 + one Newton step.  Pixels whose ray leaves the table (more than ~86 degrees off axis, or behind the camera)
//    take the exact formula.
//  * u8 bilinear taps use v_dot4_u32_u8 on the raw loaded dwords (no byte unpacking): a horizontal pass with the
//    5-bit x weights, then a vertical 24-bit mad pass -- the same integers as OpenCV's Q15 table.
//  * One thread produces PXT=4 horizontally adjacent output pixels (12 B of u8 BGR = one dwordx3 store; a wave64
//    writes two 384-B row segments = whole 128-B lines); a 256-thread workgroup covers a 128x32 output tile in four
//    passes of 8 rows; workgroups are persistent and walk (frame, tile) work items.  The fixed-point map (INTER_BITS=5)
//    lives in registers only.
//  * Source taps are gathered through L1/L2 with dword-aligned 12-byte loads (16 bytes for u16) that v_alignbyte_b32 shifts into
//    place; workgroups are dealt to XCDs in contiguous bands of tile rows so that each XCD's L2 sees one horizontal band of
//    the source frame.
//  * Three variants of the fused kernel (gf_ctx_set_kernel_mode): RP -- the default -- takes the coordinates from per-tile
//    polynomials (64 x 64 tiles) instead of the per-pixel table lookup described above (DIRECT); RL additionally stages the
//    source box of a tile in LDS by LDS-DMA.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <string.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <new>

#include "../../include/gfremap.h"

namespace {

constexpr int INTER_BITS = 5;
constexpr int INTER_TAB_SIZE = 32;
constexpr int PXT = 4;            // pixels per thread (horizontal)
#ifndef GF_TILE_W
#define GF_TILE_W 128
#endif
#ifndef GF_TILE_H
#define GF_TILE_H 32
#endif
constexpr int BLOCK = 256;
constexpr int TILE_W = GF_TILE_W;             // output tile width  (TILE_W/4 threads x 4 px)
constexpr int TILE_H = GF_TILE_H;             // output tile height (passes of PASS_H rows)
constexpr int TPR = TILE_W/PXT;               // threads per tile row
constexpr int PASS_H = BLOCK/TPR;             // rows per pass
static_assert(TILE_W % PXT == 0 && BLOCK % TPR == 0 && TILE_H % PASS_H == 0 && (TPR & (TPR - 1)) == 0, "tile shape");
constexpr int NXCD = 8;
const unsigned WG_PER_CU = getenv("GF_WG_PER_CU") ? (unsigned)atoi(getenv("GF_WG_PER_CU")) : 24u;  // persistent fused kernel: workgroups per CU (4 are resident; the surplus evens out the tail, measured +4 %)

// lens table: q = 1 + t in [1, 2^LENS_OCT), 2^LENS_SUB_BITS pieces per octave; per piece the centre q_mid and the
// LENS_DEG+1 coefficients of a polynomial in (q - q_mid): 8 doubles = 64 bytes
constexpr int LENS_OCT = 8;
constexpr int LENS_SUB_BITS = 5;
constexpr int LENS_N = LENS_OCT << LENS_SUB_BITS;       // 256 pieces
constexpr int LENS_DEG = 6;
constexpr int LENS_COEF = 8;                            // doubles per piece: q_mid, c0..c6
constexpr int LENS_DOUBLES = LENS_N*LENS_COEF;          // 2048 doubles = 16 KB
constexpr unsigned LENS_HI_LIMIT = 0x3FF00000u + ((unsigned)LENS_OCT << 20);   // high dword of 2^LENS_OCT
constexpr int LENS_SLOTS = 4;
constexpr double ROUND_MAGIC = 6755399441055744.0;      // 1.5 * 2^52: low dword of (x + magic) = rint(x), |x| < 2^31

// RP (row polynomials, see k_undistort_rotate_rp): output tiles of RT x RT pixels; per tile row the radial factor
// G = S(t)/_w is evaluated at RP_N Chebyshev nodes and interpolated by a polynomial of degree RP_N - 1 in
// z = (j - centre)/RP_HALF.
constexpr int RT = 64;                                  // tile side
constexpr int RT_TPR = RT/PXT;                          // 16 threads per tile row
constexpr int RT_RPP = BLOCK/RT_TPR;                    // 16 rows per pass
constexpr int RT_NPASS = RT/RT_RPP;                     // 4 passes
constexpr int RP_N = 8;
constexpr int RP_TAB_DOUBLES = RP_N*RP_N + RP_N + RT*RP_N;  // inverse Vandermonde matrix (row-major), the nodes, row weights [RT][RP_N]
constexpr double RP_HALF = RT/2.0;                      // the nodes span [-0.5, RT - 0.5]
constexpr double RP_GUARD = 8.0;                        // singularities of G at least this many half tiles away
// RL (LDS-staged source box of the RP kernel, u8): raw source rows at a fixed pitch, four rows per LDS-DMA instruction
constexpr int RL_PITCH = 256;                           // bytes per box row in LDS (16 lanes x 16 bytes)
#ifndef GF_RL_WGS
#define GF_RL_WGS 2              /* workgroups per CU the LDS budget below is sized for */
#endif
#ifndef GF_RL_ROWS
#define GF_RL_ROWS (GF_RL_WGS >= 3 ? 68 : 120)
#endif
constexpr int RL_ROWS = GF_RL_ROWS;                     // rows per box buffer (two buffers)
static_assert(RL_ROWS % 4 == 0, "four rows per LDS-DMA instruction");

thread_local char g_err[512] = "";
// Kernel variants of the fused kernel and the CV_16SC2 map kernel (gf_ctx_set_kernel_mode; the environment variable GF_K1
// = direct | rp | rl, read once when the library is loaded, sets the default of new contexts): rp = RP coordinates (per-tile
// polynomials) with taps gathered from the frame -- the default and the fastest as measured on MI355X; direct = per-pixel
// lens-table coordinates; rl = RP coordinates and, for u8 frames, source taps staged through LDS by LDS-DMA.  The kernels of a
// mode use the same coordinate arithmetic, so fused output == maps + remap bit for bit in every mode.
const int g_k1_mode = !getenv("GF_K1") ? GF_KERNEL_RP : !strcmp(getenv("GF_K1"), "direct") ? GF_KERNEL_DIRECT : !strcmp(getenv("GF_K1"), "rl") ? GF_KERNEL_RL : GF_KERNEL_RP;
// Persistent grid of the RP kernel, workgroups per CU (4 are resident).  0 = by launch size: 16, or 32 once that still leaves
// every workgroup 24 or more work items -- a finer split shortens the tail of partly idle CUs at the end of a launch (measured on
// C3: 128 frames 19.1 vs 19.55 us per frame, 32 frames 21.9 vs 21.5).
const unsigned WG_PER_CU_RP = getenv("GF_WG_PER_CU_RP") ? (unsigned)atoi(getenv("GF_WG_PER_CU_RP")) : 0u;
const unsigned WG_PER_CU_RL = getenv("GF_WG_PER_CU_RL") ? (unsigned)atoi(getenv("GF_WG_PER_CU_RL")) : 8u;

int fail(int code, const char* fmt, const char* a = "", long long b = 0)
{
    snprintf(g_err, sizeof g_err, fmt, a, b);
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) return fail(GF_ERR_HIP, "%s failed: HIP error %lld", #expr, (long long)e_); \
    } while (0)

constexpr int LENS_EVENTS = 4;
struct LensSlot {
    double D[4];
    bool valid;
    double* dev;          // LENS_DOUBLES doubles
    double* host;         // pinned staging
    unsigned long long last_use;
    // the last launch that read this table on each of up to LENS_EVENTS streams: what an eviction has to wait for
    hipEvent_t ev[LENS_EVENTS];
    hipStream_t ev_stream[LENS_EVENTS];
    bool ev_used[LENS_EVENTS];
    bool ev_overflow;     // more streams than events have used the slot: an eviction falls back to a device synchronise
};

// Fixed-point form of one pair of separable stretch maps (gf_undistort_rotate_superview_u8c3), cached like a lens table:
// colq[w] = cvRound(32 mapx_row), then tcol[tiles_x + 1].  One slot per context: a render uses one stretch configuration.
constexpr int STRETCH_MAX_INTS = 32768 + 1024;
struct StretchSlot {
    bool valid;
    int w, h, mid_w;
    float* key;           // host copy of mapx_row (w floats) and mapy_col (h floats) the slot was built from
    int* dev;             // STRETCH_MAX_INTS ints
    int* host;            // pinned staging
    int c0row, fy;
    hipEvent_t ev[LENS_EVENTS];
    hipStream_t ev_stream[LENS_EVENTS];
    bool ev_used[LENS_EVENTS];
    bool ev_overflow;
};

struct Ctx {
    int device;
    int n_cu;             // compute units of the device (sizes the persistent grid of the fused kernel)
    int k1_mode;          // GF_KERNEL_*
    int16_t* cubic_tab;   // device, 1024*16 int16 (cv2 INTER_CUBIC Q15 table)
    double* rp_tab;       // device, RP_TAB_DOUBLES doubles (build_rp_table)
    LensSlot lens[LENS_SLOTS];
    StretchSlot stretch;
    unsigned long long tick;
    std::mutex mu;
};

// ---------------------------------------------------------------------------------------------
// Host: bicubic Q15 table, restating cv::initInterTab2D(INTER_CUBIC, fixpt=true) (imgwarp.cpp):
// float 1-D coefficients (A=-0.75) at t=i/32, 2-D products rounded half-even to int16, then the
// sum-to-32768 fix-up over the sub-block rows/cols {2,3}.  Built with no FMA contraction.

void cubic_coeffs(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A*(x + 1) - 5*A)*(x + 1) + 8*A)*(x + 1) - 4*A;
    c[1] = ((A + 2)*x - (A + 3))*x*x + 1;
    c[2] = ((A + 2)*(1 - x) - (A + 3))*(1 - x)*(1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

int16_t g_cubic_host[1024*16];
std::once_flag g_cubic_once;

void build_cubic_host()
{
    float t1[32][4];
    for (int i = 0; i < 32; i++) cubic_coeffs(i*(1.f/32), t1[i]);
    for (int fy = 0; fy < 32; fy++)
        for (int fx = 0; fx < 32; fx++) {
            int16_t* e = g_cubic_host + (fy*32 + fx)*16;
            int isum = 0;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) {
                    float v = t1[fy][r]*t1[fx][c];
                    long q = lrintf(v*32768.f);                       // round-half-even
                    q = q > 32767 ? 32767 : q < -32768 ? -32768 : q;
                    e[r*4+c] = (int16_t)q;
                    isum += (int)q;
                }
            if (isum != 32768) {
                int diff = isum - 32768, M = 2*4+2, m = 2*4+2;
                for (int r = 2; r < 4; r++)
                    for (int c = 2; c < 4; c++) {
                        if (e[r*4+c] < e[m]) m = r*4+c;
                        else if (e[r*4+c] > e[M]) M = r*4+c;
                    }
                if (diff < 0) e[M] = (int16_t)(e[M] - diff);
                else          e[m] = (int16_t)(e[m] - diff);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Host: lens table.  S(t) = theta_d / r with r = sqrt(t), theta = atan r, theta_d = theta (1 + k1 theta^2 + ... +
// k4 theta^8) is the radial scale of cv::fisheye::initUndistortRectifyMap (fisheye.cpp) as a function of
// t = x^2 + y^2.  It is analytic in t with its nearest singularity at t = -1 (q = 0), so on pieces whose half-width is
// 1/64 .. 1/128 of their distance to that singularity (q = 1 + t split by binary exponent and the top 5 mantissa bits)
// a degree-6 interpolant at Chebyshev nodes is accurate to ~2 (1/128)^7 = 3.6e-15 relative.  Built in 80-bit long double.
// Piece (e, m) covers q in [2^e (1 + m/32), 2^e (1 + (m+1)/32)).  Its slot in the table is the 8 bits of q's high dword
// below bit 23 (three exponent bits, five mantissa bits), so the kernel indexes with a shift and a mask; the polynomial
// variable is q - q_mid, an exact subtraction (q_mid has 7 significant bits and q is within 2 % of it).

long double lens_scale_ld(const double* D, long double t)
{
    if (t <= 0) return 1.0L;
    const long double r = sqrtl(t), th = atanl(r), th2 = th*th;
    return th*(1.0L + th2*((long double)D[0] + th2*((long double)D[1] + th2*((long double)D[2] + th2*(long double)D[3]))))/r;
}

inline unsigned lens_slot_of_hi(unsigned hi) { return (hi >> (20 - LENS_SUB_BITS)) & (unsigned)(LENS_N - 1); }

void build_lens_table(const double* D, double* tab)
{
    constexpr int NC = LENS_DEG + 1;
    long double zn[NC];
    for (int k = 0; k < NC; k++) zn[k] = cosl(M_PIl*(2*k + 1)/(2.0L*NC));       // nodes in [-1, 1]
    for (int e = 0; e < LENS_OCT; e++)
        for (int m = 0; m < (1 << LENS_SUB_BITS); m++) {
            const long double hh = ldexpl(1.0L, e - LENS_SUB_BITS - 1);             // half width (a power of two)
            const long double qmid = ldexpl(1.0L, e) + (2*m + 1)*hh;
            long double A[NC][NC + 1];
            for (int k = 0; k < NC; k++) {
                long double p = 1;
                for (int j = 0; j < NC; j++) { A[k][j] = p; p *= zn[k]; }
                A[k][NC] = lens_scale_ld(D, qmid + zn[k]*hh - 1.0L);
            }
            for (int c = 0; c < NC; c++) {                   // Gaussian elimination, partial pivoting
                int piv = c;
                for (int r = c + 1; r < NC; r++) if (fabsl(A[r][c]) > fabsl(A[piv][c])) piv = r;
                if (piv != c) for (int j = 0; j <= NC; j++) { long double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
                for (int r = c + 1; r < NC; r++) {
                    const long double f = A[r][c]/A[c][c];
                    for (int j = c; j <= NC; j++) A[r][j] -= f*A[c][j];
                }
            }
            long double a[NC];
            for (int r = NC - 1; r >= 0; r--) {
                long double sum = A[r][NC];
                for (int j = r + 1; j < NC; j++) sum -= A[r][j]*a[j];
                a[r] = sum/A[r][r];
            }
            const double qm = (double)qmid;
            uint64_t b; memcpy(&b, &qm, 8);
            double* o = tab + (size_t)lens_slot_of_hi((unsigned)(b >> 32))*LENS_COEF;
            o[0] = qm;
            for (int j = 0; j < NC; j++) o[1 + j] = (double)ldexpl(a[j], -j*(e - LENS_SUB_BITS - 1));   // z = (q - q_mid)/hh
        }
}

// Host: RP table.  Nodes z_n = cos((2n+1) pi / 16); M = V^-1 with V[n][a] = z_n^a, so that the interpolating
// polynomial's monomial coefficients are c = M * (values at the nodes).  Long double, rounded once.
void build_rp_table(double* out)
{
    long double z[RP_N], A[RP_N][2*RP_N];
    for (int n = 0; n < RP_N; n++) z[n] = cosl(M_PIl*(2*n + 1)/(2.0L*RP_N));
    for (int n = 0; n < RP_N; n++) {
        long double p = 1;
        for (int a = 0; a < RP_N; a++) { A[n][a] = p; p *= z[n]; }
        for (int a = 0; a < RP_N; a++) A[n][RP_N + a] = a == n ? 1.0L : 0.0L;
    }
    for (int c = 0; c < RP_N; c++) {                     // Gauss-Jordan with partial pivoting
        int piv = c;
        for (int r = c + 1; r < RP_N; r++) if (fabsl(A[r][c]) > fabsl(A[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 2*RP_N; j++) { long double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        const long double d = 1.0L/A[c][c];
        for (int j = 0; j < 2*RP_N; j++) A[c][j] *= d;
        for (int r = 0; r < RP_N; r++)
            if (r != c) {
                const long double f = A[r][c];
                for (int j = 0; j < 2*RP_N; j++) A[r][j] -= f*A[c][j];
            }
    }
    for (int a = 0; a < RP_N; a++)
        for (int n = 0; n < RP_N; n++) out[a*RP_N + n] = (double)A[a][RP_N + n];
    for (int n = 0; n < RP_N; n++) out[RP_N*RP_N + n] = (double)z[n];
    // row weights: Lagrange basis polynomial m (nodes z) at tile row r, i.e. at (r - (RP_HALF - 0.5))/RP_HALF
    for (int r = 0; r < RT; r++) {
        const long double eta = ((long double)r - ((long double)RP_HALF - 0.5L))/(long double)RP_HALF;
        for (int m = 0; m < RP_N; m++) {
            long double l = 1;
            for (int k = 0; k < RP_N; k++) if (k != m) l *= (eta - z[k])/(z[m] - z[k]);
            out[RP_N*RP_N + RP_N + r*RP_N + m] = (double)l;
        }
    }
}

// the kernel's table lookup, restated for the host (gf_lens_scale_host; CPU tests check it against long double)
double lens_eval_host(const double* tab, double q)
{
    uint64_t b; memcpy(&b, &q, 8);
    const unsigned hi = (unsigned)(b >> 32);
    if (!(hi >= 0x3FF00000u && hi < LENS_HI_LIMIT)) return NAN;
    const double* c = tab + (size_t)lens_slot_of_hi(hi)*LENS_COEF;
    const double dz = q - c[0];
    double S = c[LENS_COEF - 1];
    for (int j = LENS_COEF - 2; j >= 1; j--) S = fma(S, dz, c[j]);
    return S;
}

// ---------------------------------------------------------------------------------------------
// Device helpers

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
struct __attribute__((packed, aligned(2))) U16x6 { uint32_t a, b, c; };

// cvRound(double) on x86-64 (cvtsd2si): round-half-even, INT_MIN when the result does not fit / NaN.
__device__ __forceinline__ int cv_round_d(double v)
{
    double r = rint(v);
    return (r >= -2147483648.0 && r <= 2147483647.0) ? (int)r : INT_MIN;
}
__device__ __forceinline__ int cv_round_f(float v)
{
    float r = rintf(v);
    return (r >= -2147483648.f && r < 2147483648.f) ? (int)r : INT_MIN;
}
__device__ __forceinline__ int sat_s16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

struct Xf {                // kernel-side copy of gf_frame_xform
    double iR[9];
    double f[2], c[2];
};
struct LensK { double k[4]; double bound; };      // bound >= |theta_d| wherever the lens table applies (t < 255)

// Exact per-pixel body of cv::fisheye::initUndistortRectifyMap: 32*u, 32*v from the homogeneous ray (_x,_y,_w).
// Used where the table does not apply.  Never inlined: it is the cold path.
__device__ __noinline__ double2 fisheye_uv32_exact(double f0, double f1, double c0, double c1, double k0, double k1, double k2, double k3,
                                                   double _x, double _y, double _w)
{
    double u, v;
    if (_w <= 0) {
        u = (_x > 0) ? -INFINITY : INFINITY;
        v = (_y > 0) ? -INFINITY : INFINITY;
    } else {
        const double x = _x/_w, y = _y/_w;
        const double r = sqrt(x*x + y*y);
        const double theta = atan(r);
        const double t2 = theta*theta, t4 = t2*t2, t6 = t4*t2, t8 = t4*t4;
        const double theta_d = theta*(1 + k0*t2 + k1*t4 + k2*t6 + k3*t8);
        const double scale = (r == 0) ? 1.0 : theta_d/r;
        u = f0*x*scale + c0;
        v = f1*y*scale + c1;
    }
    return make_double2(u*INTER_TAB_SIZE, v*INTER_TAB_SIZE);
}

// Per-thread, per-(frame,row) constants of the coordinate computation.
struct RowC {
    double rx, ry, rw;        // i*iR[.1] + iR[.2]: the reference loop's row start (mul then add, two roundings)
    double ax, ay, aw;        // iR[.0]: per-column step
    double fx32, fy32, cx32, cy32;
};

__device__ __forceinline__ RowC row_consts(const Xf& xf, int i)
{
    RowC r;
    const double di = (double)i;
    r.rx = __dadd_rn(__dmul_rn(di, xf.iR[1]), xf.iR[2]);
    r.ry = __dadd_rn(__dmul_rn(di, xf.iR[4]), xf.iR[5]);
    r.rw = __dadd_rn(__dmul_rn(di, xf.iR[7]), xf.iR[8]);
    r.ax = xf.iR[0]; r.ay = xf.iR[3]; r.aw = xf.iR[6];
    r.fx32 = xf.f[0]*32.0; r.fy32 = xf.f[1]*32.0; r.cx32 = xf.c[0]*32.0; r.cy32 = xf.c[1]*32.0;   // exact scalings
    return r;
}

// 32*u, 32*v of output pixel column j through the lens table.  Branch-free; the caller decides from W and qhi (the
// high dword of q = 1 + t) whether the result is usable: the pixel needs the exact formula when the ray is behind the
// camera (_w <= 0), t is outside [0, 255), or anything is NaN (lens_ok below).
// The column term uses one fma instead of the reference's j sequential additions (differs by < 1e-12 px).
__device__ __forceinline__ void coord_fast(const RowC& r, const double2* __restrict__ tab, int j, double& X, double& Y, double& W,
                                           unsigned& qhi, double& tu, double& tv)
{
    const double dj = (double)j;
#ifdef GF_ABL_NOMATH   /* tools/ ablation builds only: a cheap 3-degree rotation instead of the lens model */
    X = Y = W = 1.0; qhi = 0x3FF00000u;
    tu = (dj*1.003 + 2.3 + 0.05*r.rw)*32.0; tv = (r.rw*0.998 - dj*0.04 + 160.7)*32.0;
    return;
#endif
    X = __fma_rn(dj, r.ax, r.rx);
    Y = __fma_rn(dj, r.ay, r.ry);
    W = __fma_rn(dj, r.aw, r.rw);
    double iw = __builtin_amdgcn_rcp(W);                       // v_rcp_f64: 2^-24.8 relative
    iw = __fma_rn(__fma_rn(-W, iw, 1.0), iw, iw);              // one Newton step: ~2^-49
    const double x = X*iw, y = Y*iw;
    const double q = __fma_rn(x, x, __fma_rn(y, y, 1.0));      // 1 + t
    qhi = (unsigned)__double2hiint(q);
    // byte offset of the piece: 8 bits below bit 23 of the high dword, times 64 (always inside the 16 KB table)
    const unsigned off = (qhi >> (20 - LENS_SUB_BITS - 6)) & ((unsigned)(LENS_N - 1) << 6);
    const double2* c = (const double2*)((const char*)tab + off);
    const double2 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];  // (q_mid, a0) (a1, a2) (a3, a4) (a5, a6)
    const double dz = q - c0.x;
    double S = __fma_rn(c3.y, dz, c3.x);
    S = __fma_rn(S, dz, c2.y); S = __fma_rn(S, dz, c2.x); S = __fma_rn(S, dz, c1.y); S = __fma_rn(S, dz, c1.x);
    S = __fma_rn(S, dz, c0.y);
    tu = __fma_rn(x*S, r.fx32, r.cx32);
    tv = __fma_rn(y*S, r.fy32, r.cy32);
}

__device__ __forceinline__ bool lens_ok(double W, unsigned qhi, unsigned hi_limit) { return qhi < hi_limit && W > 0; }   // q >= 1 unless NaN

// The table path rounds with the 1.5*2^52 trick, which needs |32 u|, |32 v| < 2^31; u = f*(x S) + c with |x S| = |theta_d| <=
// lk.bound.  For a frame whose focal length or principal point is absurd (or NaN) the limit comes back 0: every pixel then
// takes the exact formula and cvRound's overflow rule (INT_MIN), like the reference.  Uniform per frame.
__device__ __forceinline__ unsigned lens_hi_limit(const Xf& xf, const LensK& lk)
{
    const double mu = fabs(xf.f[0])*lk.bound + fabs(xf.c[0]), mv = fabs(xf.f[1])*lk.bound + fabs(xf.c[1]);
    // a non-finite column step makes column 0 special (the reference adds the step j times: zero times there), see ray_cold
    const double step0 = xf.iR[0]*0.0 + xf.iR[3]*0.0 + xf.iR[6]*0.0;             // 0, or NaN if any step is inf/NaN
    return (mu < 16777216.0 && mv < 16777216.0 && step0 == 0.0) ? LENS_HI_LIMIT : 0u;           // 2^24 px: 32*u stays below 2^29
}

// The homogeneous ray of column j for the cold path: column 0 is the row start itself (the reference's loop has not yet
// added the column step), which matters when the step is inf/NaN (0*inf != 0).
__device__ __forceinline__ void ray_cold(const RowC& r, int j, double& X, double& Y, double& W)
{
    const double dj = (double)j;
    X = j == 0 ? r.rx : __fma_rn(dj, r.ax, r.rx);
    Y = j == 0 ? r.ry : __fma_rn(dj, r.ay, r.ry);
    W = j == 0 ? r.rw : __fma_rn(dj, r.aw, r.rw);
}

// Q5 integers of one pixel, fast or exact.
__device__ __forceinline__ void coord_q5(const RowC& r, const Xf& xf, const LensK& lk, unsigned hi_limit, const double2* __restrict__ tab, int j,
                                         int& iu, int& iv)
{
    double X, Y, W, tu, tv;
    unsigned qhi;
    coord_fast(r, tab, j, X, Y, W, qhi, tu, tv);
    if (lens_ok(W, qhi, hi_limit)) {
        iu = __double2loint(tu + ROUND_MAGIC);
        iv = __double2loint(tv + ROUND_MAGIC);
    } else {
        ray_cold(r, j, X, Y, W);
        const double2 o = fisheye_uv32_exact(xf.f[0], xf.f[1], xf.c[0], xf.c[1], lk.k[0], lk.k[1], lk.k[2], lk.k[3], X, Y, W);
        iu = cv_round_d(o.x); iv = cv_round_d(o.y);
    }
}

// Q5 integers of PXT adjacent pixels: the table path for all of them in straight-line code (independent chains
// interleave), one combined validity test (W is linear in j, so its sign is decided at the two ends; the largest high
// dword decides the range), then the exact formula for the rare pixels that need it.
__device__ __forceinline__ void coord_q5x4(const RowC& r, const Xf& xf, const LensK& lk, unsigned hi_limit, const double2* __restrict__ tab, int j0,
                                           int* iu, int* iv)
{
    unsigned qhi[PXT];
    double Wj[PXT];
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        double X, Y, tu, tv;
        coord_fast(r, tab, j0 + i, X, Y, Wj[i], qhi[i], tu, tv);
        iu[i] = __double2loint(tu + ROUND_MAGIC);
        iv[i] = __double2loint(tv + ROUND_MAGIC);
#if GF_COORD_PAIRS
        if (i == 1) __builtin_amdgcn_sched_barrier(0);     // two pixels at a time: fewer live registers
#endif
    }
    static_assert(PXT == 4, "combined validity test is written for 4 pixels");
    const unsigned qmax = max(max(qhi[0], qhi[1]), max(qhi[2], qhi[3]));
    const bool all_ok = qmax < hi_limit && fmin(Wj[0], Wj[PXT - 1]) > 0;
    if (!all_ok) {
#pragma unroll
        for (int i = 0; i < PXT; i++)
            if (!lens_ok(Wj[i], qhi[i], hi_limit)) {
                double X, Y, W;
                ray_cold(r, j0 + i, X, Y, W);
                const double2 o = fisheye_uv32_exact(xf.f[0], xf.f[1], xf.c[0], xf.c[1], lk.k[0], lk.k[1], lk.k[2], lk.k[3], X, Y, W);
                iu[i] = cv_round_d(o.x); iv[i] = cv_round_d(o.y);
            }
    }
}

// Q5 integers of PXT adjacent pixels for a stretched preset: the float32 maps of k_fisheye_maps<false> (per-pixel table
// coordinates, exact formula where the table does not apply), numpy's float32 rescale and cv::convertMaps of k_convert_maps,
// then the CV_16SC2 map entry put back together as remap reads it -- the same values, pixel by pixel, as the three launches.
__device__ __forceinline__ void coord_q5x4_stretched(const RowC& r, const Xf& xf, const LensK& lk, unsigned hi_limit, const double2* __restrict__ tab, int j0,
                                                     float sxn, float sxd, float syn, float syd, int* iu, int* iv)
{
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        double X, Y, W;
        double2 o;
        unsigned qhi;
        coord_fast(r, tab, j0 + i, X, Y, W, qhi, o.x, o.y);
        if (!lens_ok(W, qhi, hi_limit)) {
            ray_cold(r, j0 + i, X, Y, W);
            o = fisheye_uv32_exact(xf.f[0], xf.f[1], xf.c[0], xf.c[1], lk.k[0], lk.k[1], lk.k[2], lk.k[3], X, Y, W);
        }
        float fx = (float)(o.x*(1.0/32)), fy = (float)(o.y*(1.0/32));
        fx = __fdiv_rn(__fmul_rn(fx, sxn), sxd);
        fy = __fdiv_rn(__fmul_rn(fy, syn), syd);
        const int ix = cv_round_f(__fmul_rn(fx, 32.f)), iy = cv_round_f(__fmul_rn(fy, 32.f));
        iu[i] = (sat_s16(ix >> INTER_BITS) << INTER_BITS) | (ix & 31);
        iv[i] = (sat_s16(iy >> INTER_BITS) << INTER_BITS) | (iy & 31);
    }
}

__device__ __forceinline__ void load_lens_table(const double* __restrict__ g, double2* s_tab)
{
    const double2* g2 = (const double2*)g;
#pragma unroll
    for (int k = 0; k < LENS_DOUBLES/2/BLOCK; k++) s_tab[threadIdx.x + k*BLOCK] = g2[threadIdx.x + k*BLOCK];
    __syncthreads();
}

struct SrcImg {
    const uint8_t* base;   // frame base (bytes)
    int w, h;
    unsigned step;         // bytes (< 2^24; frame < 2^31 bytes)
};

template <int DEPTH> struct PixT;
template <> struct PixT<GF_DEPTH_U8>  { typedef uint8_t  T; };
template <> struct PixT<GF_DEPTH_U16> { typedef uint16_t T; };

// cv2.rotate(frame, code) of the render loop (stabilizer.py:1385-1386) folded into the tap addressing: the kernel remaps the
// ROTATED frame (SrcImg.w / .h are its size) without ever building it.  With p the rotated coordinate that runs along the
// stored frame's rows (x, or y for the 90 degree codes) and q the other one, rotated pixel (x, y) is stored pixel
// (column cs > 0 ? p : stw - 1 - p, row rs > 0 ? q : sth - 1 - q).
struct RotMap {
    int swap;              // 90 degree codes: the rotated x axis runs along the stored frame's columns (down or up a column)
    int cs, rs;            // +1 / -1: direction of p along a stored row, of q down a stored column
    int stw, sth;          // stored frame size
};
struct NoRot {};
template <bool ROT> struct RotT;
template <> struct RotT<false> { typedef NoRot T; };
template <> struct RotT<true>  { typedef RotMap T; };

__device__ __forceinline__ unsigned rot_pixel_offset(const SrcImg& s, const NoRot&, int x, int y, int esz) { return (unsigned)y*s.step + (unsigned)(x*3*esz); }
__device__ __forceinline__ unsigned rot_pixel_offset(const SrcImg& s, const RotMap& m, int x, int y, int esz)
{
    const int p = m.swap ? y : x, q = m.swap ? x : y;
    const int col = m.cs > 0 ? p : m.stw - 1 - p, row = m.rs > 0 ? q : m.sth - 1 - q;
    return (unsigned)row*s.step + (unsigned)(col*3*esz);
}

// One bilinear output pixel, cv::remapBilinear semantics (imgwarp.cpp), any border case.  (sx, sy) are the map's
// int16 coordinates, a = fy*32 + fx.  Returns false when the pixel must be left untouched (BORDER_TRANSPARENT
// outside).  This is the general (slow) path; interior pixels go through interp_u8_interior / interp_u16_interior.
struct Px3 { int c0, c1, c2, ok; };

template <int DEPTH, int BORDER, class ROTM = NoRot>
__device__ __noinline__ Px3 bilinear_px(SrcImg s, int sx, int sy, int a, int cv0, int cv1, int cv2, ROTM rm = ROTM())
{
    typedef typename PixT<DEPTH>::T T;
    const int cval[3] = {cv0, cv1, cv2};
    int out[3];
    const int fx = a & 31, fy = a >> 5;
    int p[4][3];                                    // taps: 00, 01, 10, 11
    int x0 = sx, x1 = sx + 1, y0 = sy, y1 = sy + 1;
    bool vx0 = true, vx1 = true, vy0 = true, vy1 = true;
    const bool inlier = (unsigned)sx < (unsigned)(s.w - 1) && (unsigned)sy < (unsigned)(s.h - 1);
    if (!inlier) {
        if (BORDER == GF_BORDER_CONSTANT) {
            if (sx >= s.w || sx + 1 < 0 || sy >= s.h || sy + 1 < 0) {
                return Px3{cv0, cv1, cv2, 1};
            }
            vx0 = (unsigned)x0 < (unsigned)s.w; vx1 = (unsigned)x1 < (unsigned)s.w;
            vy0 = (unsigned)y0 < (unsigned)s.h; vy1 = (unsigned)y1 < (unsigned)s.h;
        } else if (BORDER == GF_BORDER_TRANSPARENT) {
            return Px3{0, 0, 0, 0};                 // (unsigned)sx >= w-1 || (unsigned)sy >= h-1
        } else {                                    // REPLICATE: clamp
            x0 = min(max(x0, 0), s.w - 1); x1 = min(max(x1, 0), s.w - 1);
            y0 = min(max(y0, 0), s.h - 1); y1 = min(max(y1, 0), s.h - 1);
        }
    }
    const int xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
    const bool vs[4] = {vx0 && vy0, vx1 && vy0, vx0 && vy1, vx1 && vy1};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (vs[t]) {
            const T* q = (const T*)(s.base + rot_pixel_offset(s, rm, xs[t], ys[t], (int)sizeof(T)));
            p[t][0] = q[0]; p[t][1] = q[1]; p[t][2] = q[2];
        } else {
            p[t][0] = cval[0]; p[t][1] = cval[1]; p[t][2] = cval[2];
        }
    }
    if (DEPTH == GF_DEPTH_U8) {
        // Q15 table entry = {(32-fy)(32-fx), (32-fy)fx, fy(32-fx), fy*fx} * 32; (sum + 2^14) >> 15 equals
        // (sum/32 + 512) >> 10 exactly.  The table's one irregular entry a==0 ({32767,0,0,1}) gives the same
        // 8-bit result as {32768,0,0,0} (|p11 - p00| < 2^14), checked exhaustively in tests/test_oracle.py.
        const int w00 = (32 - fy)*(32 - fx), w01 = (32 - fy)*fx, w10 = fy*(32 - fx), w11 = fy*fx;
#pragma unroll
        for (int c = 0; c < 3; c++)
            out[c] = (p[0][c]*w00 + p[1][c]*w01 + p[2][c]*w10 + p[3][c]*w11 + 512) >> 10;
    } else {
        // float32 table entry = (1-fy/32 | fy/32) * (1-fx/32 | fx/32), all exact; sum left to right, no FMA.
        const float wy1 = fy*(1.f/32), wy0 = 1.f - wy1, wx1 = fx*(1.f/32), wx0 = 1.f - wx1;
        const float w00 = wy0*wx0, w01 = wy0*wx1, w10 = wy1*wx0, w11 = wy1*wx1;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float t = __fmul_rn((float)p[0][c], w00);
            t = __fadd_rn(t, __fmul_rn((float)p[1][c], w01));
            t = __fadd_rn(t, __fmul_rn((float)p[2][c], w10));
            t = __fadd_rn(t, __fmul_rn((float)p[3][c], w11));
            const int r = (int)rintf(t);
            out[c] = r < 0 ? 0 : r > 65535 ? 65535 : r;
        }
    }
    return Px3{out[0], out[1], out[2], 1};
}

// a*b + c with a, b < 2^24, kept as its own v_mad_u32_u24: the empty asm only stops the compiler from re-associating
// two chained mads into mul + mul + add3.  (No instruction may sit inside an asm statement here: the hazard recogniser
// cannot see it, and gfx950 needs wait states between v_dot4 results and a dependent VALU instruction.)
__device__ __forceinline__ unsigned umad24(unsigned a, unsigned b, unsigned c)
{
    unsigned d = __umul24(a, b) + c;
    asm("" : "+v"(d));
    return d;
}

// Interior pixel, u8.  One unaligned 8-byte load per source row covers the 6 bytes B0 G0 R0 B1 G1 R1 (+2 spare).
// Horizontal pass: v_dot4_u32_u8 of the byte windows starting at B0, G0, R0 (the row's 8 bytes shifted by 0, 8, 16 bits)
// with one weight word (32-fx on byte 0, fx on byte 3).  Vertical pass: two 24-bit mads per channel with the weights
// scaled by 64, so that  sum(S wy wx)*64 + 2^15  carries the 8-bit result exactly in bits 16..23:
//   ((sum + 512) >> 10) & 255  ==  OpenCV's (sum(S * w_q15) + 2^14) >> 15   (its Q15 weights are 32 wy wx).
// HI16 = true leaves the result in byte 2 (store_px picks it up with v_perm_b32), false shifts it down.
template <bool HI16>
__device__ __forceinline__ void bilin_u8_weights(uint64_t ta, uint64_t tb, unsigned wx, unsigned wy1, int* out);

template <bool HI16>
__device__ __forceinline__ void bilin_u8_rows(uint64_t ta, uint64_t tb, unsigned iu, unsigned iv, int* out)
{
    const unsigned wx = __umul24(iu & 31u, 0xFFFFFFu) + 32u;               // (32 - fx) | fx << 24
    const unsigned wy1 = (iv << 6) & 0x7C0u;                                // 64 fy
    bilin_u8_weights<HI16>(ta, tb, wx, wy1, out);
}

// wx = (weight of the left pixel) | (weight of the right pixel) << 24, wy1 = 64 * (weight of the lower row); weights 0 .. 32.
template <bool HI16>
__device__ __forceinline__ void bilin_u8_weights(uint64_t ta, uint64_t tb, unsigned wx, unsigned wy1, int* out)
{
    const unsigned wy0 = 2048u - wy1;                                       // 64 (32 - fy)
    const uint32_t a0 = (uint32_t)ta, a1 = (uint32_t)(ta >> 8), a2 = (uint32_t)(ta >> 16);
    const uint32_t b0 = (uint32_t)tb, b1 = (uint32_t)(tb >> 8), b2 = (uint32_t)(tb >> 16);
    const unsigned tB = __builtin_amdgcn_udot4(a0, wx, 0u, false), tG = __builtin_amdgcn_udot4(a1, wx, 0u, false), tR = __builtin_amdgcn_udot4(a2, wx, 0u, false);
    const unsigned bB = __builtin_amdgcn_udot4(b0, wx, 0u, false), bG = __builtin_amdgcn_udot4(b1, wx, 0u, false), bR = __builtin_amdgcn_udot4(b2, wx, 0u, false);
    const unsigned rB = umad24(bB, wy1, umad24(tB, wy0, 32768u));
    const unsigned rG = umad24(bG, wy1, umad24(tG, wy0, 32768u));
    const unsigned rR = umad24(bR, wy1, umad24(tR, wy0, 32768u));
    out[0] = (int)(HI16 ? rB : rB >> 16); out[1] = (int)(HI16 ? rG : rG >> 16); out[2] = (int)(HI16 ? rR : rR >> 16);
}

// (iu, iv): non-negative Q5 coordinates with sx = iu >> 5 in [0, w-4], sy = iv >> 5 in [0, h-2].
// The caller routes the frame's last three source columns to the general path, so the spare bytes of each 12-byte
// load never lie past the end of the row.  32-bit offsets from the uniform frame base (frames are < 2 GiB).
struct TapsU8 { uint64_t a, b; };          // the 8 bytes at (sx, sy) and at (sx, sy+1)

__device__ __forceinline__ TapsU8 fetch_u8_interior(const uint8_t* __restrict__ base, unsigned step, unsigned iu, unsigned iv)
{
#ifdef GF_ABL_NOGATHER   /* tools/ ablation builds only: every tap hits the same cache line */
    const unsigned off = (((iu ^ iv) >> 5) & 3)*4;
#else
    const unsigned off = __umul24(iv >> 5, step) + __umul24(iu >> 5, 3u);
#endif
#ifdef GF_ABL_NOLOAD     /* tools/ ablation builds only: no tap loads at all (VALU-only floor of the kernel) */
    const uint64_t ta = (uint64_t)off*0x9E3779B97F4A7C15ull;
    return TapsU8{ta, ta ^ (uint64_t)step};
#else
#ifndef GF_EXP_UNALIGNED_TAPS
    // Aligned 12-byte loads + v_alignbyte_b32: the vector memory path splits a load that is not dword aligned, which made the
    // former unaligned 8-byte loads the kernel's bottleneck (26.3 -> 22.5 us per 4K frame on MI355X).  The 12 bytes from the
    // aligned address cover the 6 tap bytes at any byte phase; the caller keeps sx <= w - 4 so that they stay inside the row.
    struct alignas(4) D3 { uint32_t a, b, c; };
    const unsigned o0 = off & ~3u, o1 = (off + step) & ~3u;
    const D3 r0 = *(const D3*)(base + o0), r1 = *(const D3*)(base + o1);
    const uint64_t ta = (uint64_t)__builtin_amdgcn_alignbyte(r0.b, r0.a, off) | ((uint64_t)__builtin_amdgcn_alignbyte(r0.c, r0.b, off) << 32);
    const uint64_t tb = (uint64_t)__builtin_amdgcn_alignbyte(r1.b, r1.a, off + step) | ((uint64_t)__builtin_amdgcn_alignbyte(r1.c, r1.b, off + step) << 32);
    return TapsU8{ta, tb};
#else
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    return TapsU8{*(const u64_unaligned*)(base + off), *(const u64_unaligned*)(base + (off + step))};
#endif
#endif
}

template <bool HI16>
__device__ __forceinline__ void interp_u8_interior(const uint8_t* __restrict__ base, unsigned step, unsigned iu, unsigned iv, int* out)
{
    const TapsU8 t = fetch_u8_interior(base, step, iu, iv);
    bilin_u8_rows<HI16>(t.a, t.b, iu, iv, out);
}

// u16 taps: OpenCV's float path -- float32 weights, ((p00 w00 + p01 w01) + p10 w10) + p11 w11 with every product and
// sum rounded (no FMA), cvRound, saturate.  r0/r1 = the 12 bytes (two pixels) of the upper/lower source row.
__device__ __forceinline__ void bilin_u16_vals(const unsigned* p00, const unsigned* p01, const unsigned* p10, const unsigned* p11,
                                               int fx, int fy, int* out)
{
    const float wy1 = fy*(1.f/32), wy0 = 1.f - wy1, wx1 = fx*(1.f/32), wx0 = 1.f - wx1;
    const float w00 = wy0*wx0, w01 = wy0*wx1, w10 = wy1*wx0, w11 = wy1*wx1;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float t = __fmul_rn((float)p00[c], w00);
        t = __fadd_rn(t, __fmul_rn((float)p01[c], w01));
        t = __fadd_rn(t, __fmul_rn((float)p10[c], w10));
        t = __fadd_rn(t, __fmul_rn((float)p11[c], w11));
        const int r = (int)rintf(t);
        out[c] = min(r, 65535);                                 // t >= 0 always
    }
}

__device__ __forceinline__ void bilin_u16_float(const U16x6& r0, const U16x6& r1, int fx, int fy, int* out)
{
    const unsigned p00[3] = {r0.a & 0xffff, r0.a >> 16, r0.b & 0xffff}, p01[3] = {r0.b >> 16, r0.c & 0xffff, r0.c >> 16};
    const unsigned p10[3] = {r1.a & 0xffff, r1.a >> 16, r1.b & 0xffff}, p11[3] = {r1.b >> 16, r1.c & 0xffff, r1.c >> 16};
    bilin_u16_vals(p00, p01, p10, p11, fx, fy, out);
}

struct TapsU16 { U16x6 a, b; };             // the 12 bytes (two pixels) at (sx, sy) and at (sx, sy+1)

__device__ __forceinline__ TapsU16 fetch_u16_interior(const uint8_t* __restrict__ base, unsigned step, unsigned iu, unsigned iv)
{
#ifdef GF_ABL_NOGATHER
    const unsigned off = (((iu ^ iv) >> 5) & 3)*4;
#else
    const unsigned off = __umul24(iv >> 5, step) + __umul24(iu >> 5, 6u);
#endif
    TapsU16 t;
#ifndef GF_EXP_UNALIGNED_TAPS
    // aligned 16-byte loads (see fetch_u8_interior); off is even, so the 12 tap bytes start at byte 0 or 2 of the first dword.
    // The caller keeps sx <= w - 3 so that the 16 bytes stay inside the row.
    struct alignas(4) D4 { uint32_t a, b, c, d; };
    const unsigned s1 = off + step;
    const D4 r0 = *(const D4*)(base + (off & ~3u)), r1 = *(const D4*)(base + (s1 & ~3u));
    t.a.a = __builtin_amdgcn_alignbyte(r0.b, r0.a, off); t.a.b = __builtin_amdgcn_alignbyte(r0.c, r0.b, off); t.a.c = __builtin_amdgcn_alignbyte(r0.d, r0.c, off);
    t.b.a = __builtin_amdgcn_alignbyte(r1.b, r1.a, s1);  t.b.b = __builtin_amdgcn_alignbyte(r1.c, r1.b, s1);  t.b.c = __builtin_amdgcn_alignbyte(r1.d, r1.c, s1);
#else
    __builtin_memcpy(&t.a, base + off, 12);
    __builtin_memcpy(&t.b, base + (off + step), 12);
#endif
    return t;
}

__device__ __forceinline__ void interp_u16_interior(const uint8_t* __restrict__ base, unsigned step, unsigned iu, unsigned iv, int* out)
{
    const TapsU16 t = fetch_u16_interior(base, step, iu, iv);
    bilin_u16_float(t.a, t.b, (int)(iu & 31u), (int)(iv & 31u), out);
}

#ifdef GF_RP_PREFETCH
// One dword per lane, GF_RP_PREFETCH source rows below the lane's first tap, copied into an LDS sink nobody reads
// (global_load_lds_dword: no destination register, nothing waits for it by name; it only has to have been served by the
// L1 -- which then holds the line -- before the next pass's loads return, and those are behind it in the queue anyway).
typedef __attribute__((address_space(3))) void* lds_sink_ptr;
__device__ __forceinline__ void prefetch_rows_ahead(const SrcImg& s, unsigned iu, unsigned iv)
{
    __shared__ unsigned sink[BLOCK];
    const unsigned row = min((iv >> 5) + (unsigned)(GF_RP_PREFETCH), (unsigned)s.h - 1u);
    const unsigned off = (__umul24(row, s.step) + __umul24(iu >> 5, 3u)) & ~3u;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_sink_ptr)sink + ((unsigned)threadIdx.x & ~63u)*4u));
    const uintptr_t b = (uintptr_t)s.base;           // uniform in every caller; say so
    const uintptr_t ub = (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b) | ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) << 32);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(dst), "s"(ub) : "memory");
}
#endif

template <int DEPTH> struct TapsT;
template <> struct TapsT<GF_DEPTH_U8>  { typedef TapsU8  T; };
template <> struct TapsT<GF_DEPTH_U16> { typedef TapsU16 T; };


// PXT pixels of one thread from their Q5 coordinates (iu, iv) = rint(32 u), rint(32 v).  The CV_16SC2 map would carry
// (short)(iu >> 5): whenever the unwrapped value is inside the image it equals the wrapped one, so the interior test
// runs on the unwrapped integers and only the general path applies the wrap.  Returns the keep mask.
// HI16 (u8 only): pixel values are returned in bits 16..23 of px (other bits undefined); store_px<DEPTH, true> expects that.
template <int DEPTH, int BORDER, bool HI16 = false>
__device__ __forceinline__ unsigned remap4(const SrcImg& s, const int* iu, const int* iv, const int* cval, int (*px)[3])
{
    static_assert(PXT == 4, "interior test is written for 4 pixels");
    static_assert(!HI16 || DEPTH == GF_DEPTH_U8, "HI16 is a u8 representation");
    unsigned keep = 0;
    // Taps are fetched with aligned 12-byte (u8) / 16-byte (u16) loads: keeping the last source columns out of the fast path
    // means the spare bytes are always inside the row, never past the end of the frame.
    // Interior: 0 <= sx < w1 and 0 <= sy < h1 for all four pixels, tested on the Q5 values (negative ones are huge as unsigned).
    const unsigned w1 = (unsigned)max(s.w - (DEPTH == GF_DEPTH_U8 ? 3 : 2), 0), h1 = (unsigned)(s.h - 1);
    const unsigned umax = max(max((unsigned)iu[0], (unsigned)iu[1]), max((unsigned)iu[2], (unsigned)iu[3]));
    const unsigned vmax = max(max((unsigned)iv[0], (unsigned)iv[1]), max((unsigned)iv[2], (unsigned)iv[3]));
    if (umax < (w1 << INTER_BITS) && vmax < (h1 << INTER_BITS)) {
#ifdef GF_RP_PREFETCH    /* experiment (tools/exp_r03.sh): touch the rows this thread will need GF_RP_PREFETCH source rows further down, behind this pass's tap loads */
        if constexpr (DEPTH == GF_DEPTH_U8) {
            TapsU8 T[PXT];
#pragma unroll
            for (int i = 0; i < PXT; i++) T[i] = fetch_u8_interior(s.base, s.step, (unsigned)iu[i], (unsigned)iv[i]);
            prefetch_rows_ahead(s, (unsigned)iu[0], (unsigned)iv[0]);
#pragma unroll
            for (int i = 0; i < PXT; i++) bilin_u8_rows<HI16>(T[i].a, T[i].b, (unsigned)iu[i], (unsigned)iv[i], px[i]);
            return 0;
        }
#endif
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            if (DEPTH == GF_DEPTH_U8) interp_u8_interior<HI16>(s.base, s.step, (unsigned)iu[i], (unsigned)iv[i], px[i]);
            else                      interp_u16_interior(s.base, s.step, (unsigned)iu[i], (unsigned)iv[i], px[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            const int sx = (int)(int16_t)(iu[i] >> INTER_BITS), sy = (int)(int16_t)(iv[i] >> INTER_BITS);   // Q1: wrapping cast
            const int a = (iv[i] & 31)*32 + (iu[i] & 31);
            const Px3 o = bilinear_px<DEPTH, BORDER>(s, sx, sy, a, cval[0], cval[1], cval[2]);
            if (!o.ok) keep |= 1u << i;
            px[i][0] = HI16 ? o.c0 << 16 : o.c0; px[i][1] = HI16 ? o.c1 << 16 : o.c1; px[i][2] = HI16 ? o.c2 << 16 : o.c2;
        }
    }
    return keep;
}

// remap4 on a frame that cv2.rotate would have turned first (RotMap).  Interior pixels: the 2 x 2 taps of the rotated frame are
// a 2 x 2 block of the stored frame -- left column (p, or stw - 2 - p when p runs right to left), upper row likewise -- so the
// same aligned loads serve, with the horizontal / vertical weights taken from the fraction of whichever rotated axis runs
// along the stored rows / columns and mirrored where the axis is reversed.  u8: the weighted sum is the same set of integer
// products, so the result is bit-identical to remapping the rotated copy.  u16: OpenCV sums the four float products in the
// rotated frame's tap order, which is kept here by permuting the taps.  Everything else takes the general path, whose border
// decisions are made on the rotated coordinates and whose pixel fetch goes through rot_pixel_offset.
template <int DEPTH, int BORDER, bool HI16 = false>
__device__ __forceinline__ unsigned remap4_rot(const SrcImg& s, const RotMap& m, const int* iu, const int* iv, const int* cval, int (*px)[3])
{
    static_assert(PXT == 4, "interior test is written for 4 pixels");
    unsigned keep = 0;
    int ip[PXT], iq[PXT];
#pragma unroll
    for (int i = 0; i < PXT; i++) { ip[i] = m.swap ? iv[i] : iu[i]; iq[i] = m.swap ? iu[i] : iv[i]; }
    const int pmin = min(min(ip[0], ip[1]), min(ip[2], ip[3])), pmax = max(max(ip[0], ip[1]), max(ip[2], ip[3]));
    const int qmin = min(min(iq[0], iq[1]), min(iq[2], iq[3])), qmax = max(max(iq[0], iq[1]), max(iq[2], iq[3]));
    // the aligned 12-byte (u8) / 16-byte (u16) loads need their left column <= stw - 4 (see remap4); rows 0 .. sth - 2
    const int plo = m.cs > 0 ? 0 : 2, phi = m.cs > 0 ? m.stw - 4 : m.stw - 2;
    if (pmin >= (plo << INTER_BITS) && pmax < ((phi + 1) << INTER_BITS) && qmin >= 0 && qmax < ((m.sth - 1) << INTER_BITS)) {
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            const unsigned sp = (unsigned)ip[i] >> INTER_BITS, sq = (unsigned)iq[i] >> INTER_BITS, fp = (unsigned)ip[i] & 31u, fq = (unsigned)iq[i] & 31u;
            const unsigned col = m.cs > 0 ? sp : (unsigned)(m.stw - 2) - sp, row = m.rs > 0 ? sq : (unsigned)(m.sth - 2) - sq;
            if constexpr (DEPTH == GF_DEPTH_U8) {
                const unsigned hx = m.cs > 0 ? fp : 32u - fp, vy = m.rs > 0 ? fq : 32u - fq;      // weight of the right pixel / the lower row
                const TapsU8 t = fetch_u8_interior(s.base, s.step, col << INTER_BITS, row << INTER_BITS);
                bilin_u8_weights<HI16>(t.a, t.b, __umul24(hx, 0xFFFFFFu) + 32u, vy << 6, px[i]);
            } else {
                const TapsU16 t = fetch_u16_interior(s.base, s.step, col << INTER_BITS, row << INTER_BITS);
                // T[r][c]: stored block, r = upper / lower row, c = left / right pixel
                unsigned T[2][2][3] = {{{t.a.a & 0xffff, t.a.a >> 16, t.a.b & 0xffff}, {t.a.b >> 16, t.a.c & 0xffff, t.a.c >> 16}},
                                       {{t.b.a & 0xffff, t.b.a >> 16, t.b.b & 0xffff}, {t.b.b >> 16, t.b.c & 0xffff, t.b.c >> 16}}};
                // rotated tap (i along x, j along y) sits at p-step tp = swap ? j : i and q-step tq = swap ? i : j, i.e. stored column
                // c = cs > 0 ? tp : 1 - tp and stored row r = rs > 0 ? tq : 1 - tq
                unsigned P[2][2][3];            // [j][i]
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int ii = 0; ii < 2; ii++)
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            const unsigned a00 = T[0][0][ch], a01 = T[0][1][ch], a10 = T[1][0][ch], a11 = T[1][1][ch];
                            // value at (tq, tp) for both choices of swap, then the mirrorings
                            const int tp0 = ii, tq0 = j, tp1 = j, tq1 = ii;
                            auto pick = [&](int tq, int tp) -> unsigned {
                                const unsigned up = m.cs > 0 ? (tp ? a01 : a00) : (tp ? a00 : a01);     // stored upper row, column by tp
                                const unsigned lo = m.cs > 0 ? (tp ? a11 : a10) : (tp ? a10 : a11);
                                return m.rs > 0 ? (tq ? lo : up) : (tq ? up : lo);
                            };
                            P[j][ii][ch] = m.swap ? pick(tq1, tp1) : pick(tq0, tp0);
                        }
                bilin_u16_vals(P[0][0], P[0][1], P[1][0], P[1][1], iu[i] & 31, iv[i] & 31, px[i]);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            const int sx = (int)(int16_t)(iu[i] >> INTER_BITS), sy = (int)(int16_t)(iv[i] >> INTER_BITS);   // Q1: wrapping cast
            const int a = (iv[i] & 31)*32 + (iu[i] & 31);
            const Px3 o = bilinear_px<DEPTH, BORDER, RotMap>(s, sx, sy, a, cval[0], cval[1], cval[2], m);
            if (!o.ok) keep |= 1u << i;
            px[i][0] = HI16 ? o.c0 << 16 : o.c0; px[i][1] = HI16 ? o.c1 << 16 : o.c1; px[i][2] = HI16 ? o.c2 << 16 : o.c2;
        }
    }
    return keep;
}

// remap4 in two phases, so that a thread can put other work (the next row's coordinates) between issuing its tap loads
// and using them.  fetch4 returns whether all PXT pixels are interior (then T holds their taps, loads in flight).
template <int DEPTH>
__device__ __forceinline__ bool fetch4(const SrcImg& s, const int* iu, const int* iv, typename TapsT<DEPTH>::T* T)
{
    const unsigned w1 = (unsigned)max(s.w - (DEPTH == GF_DEPTH_U8 ? 3 : 2), 0), h1 = (unsigned)(s.h - 1);
    const unsigned umax = max(max((unsigned)iu[0], (unsigned)iu[1]), max((unsigned)iu[2], (unsigned)iu[3]));
    const unsigned vmax = max(max((unsigned)iv[0], (unsigned)iv[1]), max((unsigned)iv[2], (unsigned)iv[3]));
    const bool interior = umax < (w1 << INTER_BITS) && vmax < (h1 << INTER_BITS);
    if (interior) {
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            if constexpr (DEPTH == GF_DEPTH_U8) T[i] = fetch_u8_interior(s.base, s.step, (unsigned)iu[i], (unsigned)iv[i]);
            else                                T[i] = fetch_u16_interior(s.base, s.step, (unsigned)iu[i], (unsigned)iv[i]);
        }
    }
    return interior;
}

template <int DEPTH, int BORDER, bool HI16>
__device__ __forceinline__ unsigned finish4(const SrcImg& s, const int* iu, const int* iv, const int* cval, bool interior,
                                            const typename TapsT<DEPTH>::T* T, int (*px)[3])
{
    unsigned keep = 0;
    if (interior) {
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            if constexpr (DEPTH == GF_DEPTH_U8) bilin_u8_rows<HI16>(T[i].a, T[i].b, (unsigned)iu[i], (unsigned)iv[i], px[i]);
            else                                bilin_u16_float(T[i].a, T[i].b, iu[i] & 31, iv[i] & 31, px[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            const int sx = (int)(int16_t)(iu[i] >> INTER_BITS), sy = (int)(int16_t)(iv[i] >> INTER_BITS);   // Q1: wrapping cast
            const int a = (iv[i] & 31)*32 + (iu[i] & 31);
            const Px3 o = bilinear_px<DEPTH, BORDER>(s, sx, sy, a, cval[0], cval[1], cval[2]);
            if (!o.ok) keep |= 1u << i;
            px[i][0] = HI16 ? o.c0 << 16 : o.c0; px[i][1] = HI16 ? o.c1 << 16 : o.c1; px[i][2] = HI16 ? o.c2 << 16 : o.c2;
        }
    }
    return keep;
}

// Store PXT pixels (3 channels each) of one thread.  `keep` bit i set = pixel i untouched.
// HI16: u8 values sit in byte 2 of each px word; four of them are packed into a dword with three v_perm_b32.
__device__ __forceinline__ uint32_t pack4_hi16(int a, int b, int c, int d)
{
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0602u);     // a.b2 | b.b2 << 8
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x0c0c0602u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

template <int DEPTH, bool HI16 = false>
__device__ __forceinline__ void store_px(uint8_t* drow, int x0, int dw, const int (*px)[3], unsigned keep, bool vec_ok)
{
    typedef typename PixT<DEPTH>::T T;
#ifdef GF_ABL_NOSTORE    /* tools/ ablation builds only: results are computed but (practically) never stored */
    if (px[0][0] != 0x7fffffff) return;
#endif
    if (vec_ok && keep == 0 && x0 + PXT <= dw) {
        if (DEPTH == GF_DEPTH_U8) {
            uint32_t w0, w1, w2;
            if (HI16) {
                w0 = pack4_hi16(px[0][0], px[0][1], px[0][2], px[1][0]);
                w1 = pack4_hi16(px[1][1], px[1][2], px[2][0], px[2][1]);
                w2 = pack4_hi16(px[2][2], px[3][0], px[3][1], px[3][2]);
            } else {
                w0 = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
                w1 = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
                w2 = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
            }
            uint32_t* d = (uint32_t*)(drow + (size_t)x0*3);
            struct alignas(4) W3 { uint32_t a, b, c; };
            *(W3*)d = W3{w0, w1, w2};
        } else {
            uint32_t w[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int e = i*2;          // element index in the 12-element run
                w[i] = (uint32_t)px[e/3][e%3] | ((uint32_t)px[(e+1)/3][(e+1)%3] << 16);
            }
            uint32_t* d = (uint32_t*)(drow + (size_t)x0*6);
            struct alignas(8) W2 { uint32_t a, b; };
            *(W2*)(d) = W2{w[0], w[1]}; *(W2*)(d + 2) = W2{w[2], w[3]}; *(W2*)(d + 4) = W2{w[4], w[5]};
        }
        return;
    }
    T* d = (T*)drow + (size_t)x0*3;
#pragma unroll
    for (int i = 0; i < PXT; i++)
        if (x0 + i < dw && !((keep >> i) & 1)) {
            d[i*3+0] = (T)(HI16 ? px[i][0] >> 16 : px[i][0]); d[i*3+1] = (T)(HI16 ? px[i][1] >> 16 : px[i][1]);
            d[i*3+2] = (T)(HI16 ? px[i][2] >> 16 : px[i][2]);
        }
}

// Workgroup -> (tile_x, tile_y): hardware deals consecutive workgroup ids round-robin over the 8 XCDs;
// give XCD k the k-th contiguous band of tile rows so neighbouring tiles (which share source rows)
// share an L2.  Bijective for any tile count (placement is a speed matter only).
__device__ __forceinline__ void tile_of_block(int bid, int tiles_x, int tiles_y, int& tx, int& ty)
{
    const int n = tiles_x*tiles_y;
    const int q = n / NXCD, r = n % NXCD;
    const int xcd = bid % NXCD, k = bid / NXCD;
    const int t = xcd < r ? xcd*(q + 1) + k : r*(q + 1) + (xcd - r)*q + k;
    ty = t / tiles_x;
    tx = t - ty*tiles_x;
}

struct RemapArgs {
    const uint8_t* src; int sw, sh; unsigned sstep; size_t sframe;
    uint8_t* dst; int dw, dh; size_t dstep, dframe;
    int tiles_x, tiles_y;
    int cval[3];
    int dst_vec_ok;        // dst base/step/frame stride allow 4-byte (u8) / 8-byte (u16) vector stores
    // stretched presets (calibrate_video.py:375-380): map coordinates go through float32, are rescaled as numpy does it
    // (multiply, round, divide, round) and converted like cv::convertMaps.  Only the DIRECT kernel reads these.
    int stretch_on;
    float sxn, sxd, syn, syd;
};

// K1: fused fisheye map + rotation + bilinear remap, batched over frames.
// Persistent workgroups: the grid is sized to the machine (a multiple of the 8 XCDs), every workgroup loads the 16 KB
// lens table into LDS once and then walks the work list v = blockIdx.x, blockIdx.x + gridDim.x, ... with a fixed stride, so
// it always terminates.  v enumerates (frame, padded tile id): tiles_pad = tiles rounded up to a multiple of 8 keeps
// v % 8 == blockIdx.x % 8, i.e. a workgroup stays on its XCD's band of every frame (tile_of_block); ids past a
// band's end are skipped.
__device__ __forceinline__ bool tile_of_padded_id(unsigned w, int tiles_x, int tiles_y, int& tx, int& ty)
{
    const unsigned n = (unsigned)(tiles_x*tiles_y);
    const unsigned q = n / NXCD, r = n % NXCD;
    const unsigned xcd = w % NXCD, k = w / NXCD;
    if (k >= q + (xcd < r ? 1u : 0u)) return false;
    const unsigned t = xcd < r ? xcd*(q + 1) + k : r*(q + 1) + (xcd - r)*q + k;
    ty = (int)(t / (unsigned)tiles_x);
    tx = (int)(t - (unsigned)ty*(unsigned)tiles_x);
    return true;
}

#ifndef GF_K1_PIPELINE
#define GF_K1_PIPELINE 0
#endif
#ifndef GF_K1_WAVES
#define GF_K1_WAVES 4       /* 121 VGPRs; at 5 (96 VGPRs) the kernel spills 21 registers since it also carries the stretched-preset
                               rescale, and runs at half the speed (43.9 vs 21.5 us per C3 frame, tools/direct_ab.sh) */
#endif
template <int DEPTH, int BORDER>
__global__ void __launch_bounds__(BLOCK, GF_K1_WAVES) k_undistort_rotate(RemapArgs A, const Xf* __restrict__ xfs, const double* __restrict__ lens, LensK lk,
                                                             unsigned tiles_pad, unsigned n_work)
{
    __shared__ double2 s_tab[LENS_DOUBLES/2];
    load_lens_table(lens, s_tab);
    constexpr bool HI16 = DEPTH == GF_DEPTH_U8;
    const int lx = threadIdx.x & (TPR - 1), ly = threadIdx.x / TPR;
#pragma unroll 1
    for (unsigned v = blockIdx.x; v < n_work; v += gridDim.x) {
        const unsigned frame = v / tiles_pad;
        int tx, ty;
        if (!tile_of_padded_id(v - frame*tiles_pad, A.tiles_x, A.tiles_y, tx, ty)) continue;
        const int x0 = tx*TILE_W + lx*PXT;
        if (x0 >= A.dw) continue;
        const Xf& xf = xfs[frame];
        const unsigned hi_limit = lens_hi_limit(xf, lk);
        const SrcImg s{A.src + (size_t)frame*A.sframe, A.sw, A.sh, A.sstep};
#if GF_K1_PIPELINE
        // Software pipeline over the tile's passes: the taps of row y are in flight while the coordinates of row
        // y + PASS_H are computed, so a wave overlaps its own gather latency with its own arithmetic.
        int y = ty*TILE_H + ly;
        if (y >= A.dh) continue;
        int iu[PXT], iv[PXT];
        coord_q5x4(row_consts(xf, y), xf, lk, hi_limit, s_tab, x0, iu, iv);
#pragma unroll 1
        for (int pass = 0; pass < TILE_H/PASS_H; pass++) {
            typename TapsT<DEPTH>::T T[PXT];
            const bool interior = fetch4<DEPTH>(s, iu, iv, T);
            const int yn = y + PASS_H;
            const bool more = pass + 1 < TILE_H/PASS_H && yn < A.dh;
            int nu[PXT], nv[PXT];
            if (more) coord_q5x4(row_consts(xf, yn), xf, lk, hi_limit, s_tab, x0, nu, nv);
            int px[PXT][3];
            const unsigned keep = finish4<DEPTH, BORDER, HI16>(s, iu, iv, A.cval, interior, T, px);
            uint8_t* drow = A.dst + (size_t)frame*A.dframe + (size_t)y*A.dstep;
            store_px<DEPTH, HI16>(drow, x0, A.dw, px, keep, A.dst_vec_ok);
            if (!more) break;
#pragma unroll
            for (int i = 0; i < PXT; i++) { iu[i] = nu[i]; iv[i] = nv[i]; }
            y = yn;
        }
#else
#pragma unroll 1
        for (int pass = 0; pass < TILE_H/PASS_H; pass++) {
            const int y = ty*TILE_H + pass*PASS_H + ly;
            if (y >= A.dh) break;
            int iu[PXT], iv[PXT];
            if (A.stretch_on) coord_q5x4_stretched(row_consts(xf, y), xf, lk, hi_limit, s_tab, x0, A.sxn, A.sxd, A.syn, A.syd, iu, iv);
            else coord_q5x4(row_consts(xf, y), xf, lk, hi_limit, s_tab, x0, iu, iv);
            int px[PXT][3];
            const unsigned keep = remap4<DEPTH, BORDER, HI16>(s, iu, iv, A.cval, px);
            uint8_t* drow = A.dst + (size_t)frame*A.dframe + (size_t)y*A.dstep;
            store_px<DEPTH, HI16>(drow, x0, A.dw, px, keep, A.dst_vec_ok);
        }
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// RP: coordinates from per-row polynomials.
// 32u = fx32 * _x * G + cx32 and 32v = fy32 * _y * G + cy32 with G = S(t) / _w.  _x, _y are linear in the column j, and G is
// analytic in the output position with its nearest singularities (q = 1 + t = 0 and _w = 0) |P| / |step of iR| pixels away
// from a pixel with ray P = (_x, _y, _w) -- about one focal length of the output camera.  On a 64 x 64 tile whose half width
// is at most 1/8 of that distance the tensor-product interpolant through 8 x 8 Chebyshev nodes is accurate to 2e-10
// relative at worst (Bernstein ellipse parameter >= 16) and to rounding error for the BASELINE configurations (parameter
// 110 for a 4K frame, 55 for 1080p: 55^-8 = 1e-14), the same class of error as the lens table itself.  Per tile
//   rp_nodes: one wave evaluates G at the 64 nodes through the lens table (lane = node column a + 8 * node row m), checks
//             every node (ray in front of the camera and inside the table, singularities far enough, sane focal data) and
//             turns each node row into monomial coefficients in z = (j - centre)/32 (an 8-term dot product with a row of
//             the inverse Vandermonde matrix; the eight nodes of a row sit in eight adjacent lanes, so no barrier);
//   rp_rows:  the coefficient vector of every tile row = the node rows' vectors weighted with the Lagrange weights of that
//             row (8 fma per coefficient);
//   rp_coords: a pixel costs a degree-7 Horner evaluation (7 fma) instead of the reciprocal, the table lookup and the
//             degree-6 piece: 15 f64 operations instead of 31.
// A tile with a node that fails a check takes the per-pixel path (coord_q5x4).  k_undistort_rotate_rp, k_undistort_rotate_rl
// and k_fisheye_maps_rp use the same functions on the same tile grid, so their coordinates are identical bit for bit.
static_assert(RP_N == 8 && RT == 64 && BLOCK == 256, "lane <-> node and thread <-> row mappings below");
#ifndef GF_RP_WAVES
#define GF_RP_WAVES 4            /* waves per SIMD the RP kernel is compiled for: keeps it within 128 VGPRs */
#endif

struct RpConst {                    // LDS copies of build_rp_table's constants
    double M[RP_N][RP_N];           // inverse Vandermonde matrix
    double Ly[RT][RP_N];            // Lagrange weights of the tile rows
};
struct RpShared {
    RpConst K;
    double U[RP_N][RP_N];           // scratch of the wave that runs rp_nodes
    double C2[2][RP_N][RP_N];       // [buffer][node row][power]
    double C[2][RT][RP_N];          // [buffer][tile row][power]: monomial coefficients in z
    int ok[4];                      // [slot]: the tile may use its polynomials
};

// Copies the constants into LDS (the caller's next barrier orders that before their first use).
__device__ __forceinline__ void rp_load_consts(const double* __restrict__ rp_tab, RpConst& K, int tid, int nthreads)
{
    for (int i = tid; i < RP_N*RP_N; i += nthreads) (&K.M[0][0])[i] = rp_tab[i];
    for (int i = tid; i < RT*RP_N; i += nthreads) (&K.Ly[0][0])[i] = rp_tab[RP_N*RP_N + RP_N + i];
}
__device__ __forceinline__ double rp_node_off(const double* __restrict__ rp_tab, int n) { return (RP_HALF - 0.5) + RP_HALF*rp_tab[RP_N*RP_N + n]; }

// One wave, all 64 lanes: lane = a + 8 m evaluates node (column a, row m) of tile (tx, ty); offx / offy = rp_node_off(a / m).
// Results: C2[m][a] (coefficient a of node row m), *okflag (every node passed), and with BOX the source positions of the
// tile's rim nodes merged into boxslot[0..3] (min u, max u, min v, max v).
#ifdef GF_LB
__device__ __forceinline__ void lb_node_boxes(const Xf& xf, double X, double Y, double G, int a, int m, int all_ok, int* slot);
#endif
template <int BOX>          // 0: none; 1: the tile's box (RL); 2: the boxes of its two halves (LB, lb_node_boxes)
__device__ __forceinline__ void rp_nodes(const Xf& xf, const LensK& lk, const double2* __restrict__ tab, double offx, double offy, int a, int m,
                                         int tx, int ty, double (*U)[RP_N], const RpConst& K, double (*C2)[RP_N], int* okflag, int* boxslot)
{
    const unsigned hi_limit = lens_hi_limit(xf, lk);
    const double dj = (double)(tx*RT) + offx, di = (double)(ty*RT) + offy;
    const double rx = __dadd_rn(__dmul_rn(di, xf.iR[1]), xf.iR[2]), ry = __dadd_rn(__dmul_rn(di, xf.iR[4]), xf.iR[5]);
    const double rw = __dadd_rn(__dmul_rn(di, xf.iR[7]), xf.iR[8]);
    const double X = __fma_rn(dj, xf.iR[0], rx), Y = __fma_rn(dj, xf.iR[3], ry), W = __fma_rn(dj, xf.iR[6], rw);
    double iw = __builtin_amdgcn_rcp(W);
    iw = __fma_rn(__fma_rn(-W, iw, 1.0), iw, iw);
    const double x = X*iw, y = Y*iw;
    const double q = __fma_rn(x, x, __fma_rn(y, y, 1.0));
    const unsigned qhi = (unsigned)__double2hiint(q);
    const unsigned off = (qhi >> (20 - LENS_SUB_BITS - 6)) & ((unsigned)(LENS_N - 1) << 6);
    const double2* c = (const double2*)((const char*)tab + off);
    const double2 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
    const double dz = q - c0.x;
    double Sc = __fma_rn(c3.y, dz, c3.x);
    Sc = __fma_rn(Sc, dz, c2.y); Sc = __fma_rn(Sc, dz, c2.x); Sc = __fma_rn(Sc, dz, c1.y); Sc = __fma_rn(Sc, dz, c1.x);
    Sc = __fma_rn(Sc, dz, c0.y);
    const double G = Sc*iw;
    // distance to the nearest singularity of G along x and along y, in pixels: |P| / |step| (q = 0) and _w / |step of _w| (_w = 0)
    const double ax2 = xf.iR[0]*xf.iR[0] + xf.iR[3]*xf.iR[3] + xf.iR[6]*xf.iR[6], ay2 = xf.iR[1]*xf.iR[1] + xf.iR[4]*xf.iR[4] + xf.iR[7]*xf.iR[7];
    const double G2 = (RP_GUARD*RP_HALF)*(RP_GUARD*RP_HALF), W2 = W*W;
    const bool ok = lens_ok(W, qhi, hi_limit) && W2*q >= G2*fmax(ax2, ay2) && W2 >= G2*fmax(xf.iR[6]*xf.iR[6], xf.iR[7]*xf.iR[7]);
    const int all_ok = __all(ok ? 1 : 0);
    if (a == 0 && m == 0) *okflag = all_ok;
    // Box: the corner nodes and the end nodes of the two middle node rows (8 samples; LDS atomics from many lanes to one
    // address are serialised and cost more than the tile itself).  The margin of rl_box covers the bulge between them.
#ifdef GF_LB
    if constexpr (BOX == 2) lb_node_boxes(xf, X, Y, G, a, m, all_ok, boxslot);
#endif
    if (BOX == 1 && all_ok && (a == 0 || a == RP_N - 1) && (m == 0 || m == RP_N/2 - 1 || m == RP_N/2 || m == RP_N - 1)) {
        const int su = __double2loint(__fma_rn(X*G, xf.f[0]*32.0, xf.c[0]*32.0) + ROUND_MAGIC) >> INTER_BITS;
        const int sv = __double2loint(__fma_rn(Y*G, xf.f[1]*32.0, xf.c[1]*32.0) + ROUND_MAGIC) >> INTER_BITS;
        atomicMin(&boxslot[0], su); atomicMax(&boxslot[1], su); atomicMin(&boxslot[2], sv); atomicMax(&boxslot[3], sv);
    }
    // node values -> coefficient a of node row m, through the scratch (LDS operations of one wave execute in program order;
    // the empty asm statements keep the compiler from moving the reads above the writes and back)
    U[m][a] = G;
    asm volatile("" ::: "memory");
    const double2* u = (const double2*)U[m];
    const double2* mm = (const double2*)K.M[a];
    const double2 u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3], m0 = mm[0], m1 = mm[1], m2 = mm[2], m3 = mm[3];
    asm volatile("" ::: "memory");
    double cf = m0.x*u0.x;
    cf = __fma_rn(m0.y, u0.y, cf); cf = __fma_rn(m1.x, u1.x, cf); cf = __fma_rn(m1.y, u1.y, cf);
    cf = __fma_rn(m2.x, u2.x, cf); cf = __fma_rn(m2.y, u2.y, cf); cf = __fma_rn(m3.x, u3.x, cf); cf = __fma_rn(m3.y, u3.y, cf);
    C2[m][a] = cf;
}

// Coefficient a of tile rows r0, r0 + rstride, ... (nr of them) from the node rows' coefficients: C[r][a] = sum_m Ly[r][m] C2[m][a].
// One coefficient of one tile row: the node rows' coefficients c2[0..7] weighted with the row's Lagrange weights (this order).
__device__ __forceinline__ double rp_row_coef(const double* c2, const double* ly_row)
{
    const double2* w = (const double2*)ly_row;
    const double2 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    double cf = w0.x*c2[0];
    cf = __fma_rn(w0.y, c2[1], cf); cf = __fma_rn(w1.x, c2[2], cf); cf = __fma_rn(w1.y, c2[3], cf);
    cf = __fma_rn(w2.x, c2[4], cf); cf = __fma_rn(w2.y, c2[5], cf); cf = __fma_rn(w3.x, c2[6], cf); cf = __fma_rn(w3.y, c2[7], cf);
    return cf;
}

__device__ __forceinline__ void rp_rows(const double (*C2)[RP_N], const RpConst& K, int a, int r0, int rstride, int nr, double (*C)[RP_N])
{
    double c2[RP_N];
#pragma unroll
    for (int m = 0; m < RP_N; m++) c2[m] = C2[m][a];
#pragma unroll 2
    for (int j = 0; j < nr; j++) {
        const int r = r0 + j*rstride;
        C[r][a] = rp_row_coef(c2, K.Ly[r]);
    }
}

// Frame / row constants of the RP coordinates.  32u = fx32 (j ax + (i iR1 + iR2)) G + cx32 is evaluated as
// fma(fma(j, ax', rx'), G, cx32) with ax' = ax fx32 and rx' = fma(i, iR1 fx32, iR2 fx32): the focal length is folded into the
// ray once per frame (RpFrameC) and once per row (RpRowC), which leaves 3 f64 operations per coordinate and pixel.  Every
// RP kernel (fused, LDS-box, maps, blend) goes through these functions, so their coordinates agree bit for bit.
struct RpFrameC { double axp, ayp, b1x, b2x, b1y, b2y, cx32, cy32; };
struct RpRowC { double axp, ayp, rxp, ryp, cx32, cy32; };

__device__ __forceinline__ RpFrameC rp_frame_consts(const Xf& xf)
{
    RpFrameC t;
    const double fx32 = xf.f[0]*32.0, fy32 = xf.f[1]*32.0;       // exact scalings
    t.axp = xf.iR[0]*fx32; t.b1x = xf.iR[1]*fx32; t.b2x = xf.iR[2]*fx32;
    t.ayp = xf.iR[3]*fy32; t.b1y = xf.iR[4]*fy32; t.b2y = xf.iR[5]*fy32;
    t.cx32 = xf.c[0]*32.0; t.cy32 = xf.c[1]*32.0;
    return t;
}

__device__ __forceinline__ RpRowC rp_row_consts(const RpFrameC& t, int i)
{
    RpRowC r;
    const double di = (double)i;
    r.axp = t.axp; r.ayp = t.ayp; r.cx32 = t.cx32; r.cy32 = t.cy32;
    r.rxp = __fma_rn(di, t.b1x, t.b2x);
    r.ryp = __fma_rn(di, t.b1y, t.b2y);
    return r;
}

// Q5 integers of the thread's PXT pixels (columns x0 .. x0 + 3 of the row with constants rc) from the row's coefficients:
// low dwords of 32u + mu and 32v + mv.  mu = mv = ROUND_MAGIC gives rint(32u), rint(32v); ROUND_MAGIC - 32 k (k an integer)
// gives them relative to source position k with the same ties (the offset is even).
__device__ __forceinline__ void rp_coords(const RpRowC& rc, const double* crow, const double* z, int x0, double mu, double mv, int* iu, int* iv)
{
    static_assert(RP_N == 8, "Horner chain is written for degree 7");
    const double2* c = (const double2*)crow;
    const double2 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
#pragma unroll
    for (int e = 0; e < PXT; e++) {
        double G = __fma_rn(c3.y, z[e], c3.x);
        G = __fma_rn(G, z[e], c2.y); G = __fma_rn(G, z[e], c2.x); G = __fma_rn(G, z[e], c1.y); G = __fma_rn(G, z[e], c1.x);
        G = __fma_rn(G, z[e], c0.y); G = __fma_rn(G, z[e], c0.x);
        const double dj = (double)(x0 + e);
        const double X = __fma_rn(dj, rc.axp, rc.rxp), Y = __fma_rn(dj, rc.ayp, rc.ryp);
        iu[e] = __double2loint(__fma_rn(X, G, rc.cx32) + mu);
        iv[e] = __double2loint(__fma_rn(Y, G, rc.cy32) + mv);
    }
}

// The per-pixel fallback of the RP kernels as a call: tiles that fail rp_nodes' checks are rare, and inlined this code costs
// the hot loop about 30 VGPRs.  Results come back in registers.
struct Q5x4 { int iu[PXT], iv[PXT]; };
__device__ __noinline__ Q5x4 coord_q5x4_cold(const Xf* xfp, double k0, double k1, double k2, double k3, double bound, const double2* tab, int y, int x0)
{
    const Xf xf = *xfp;
    LensK lk; lk.k[0] = k0; lk.k[1] = k1; lk.k[2] = k2; lk.k[3] = k3; lk.bound = bound;
    Q5x4 q = {};
    const RowC rc = row_consts(xf, y);
    const unsigned hi_limit = lens_hi_limit(xf, lk);
#pragma unroll 1
    for (int e = 0; e < PXT; e++) {                      // one pixel at a time: few live registers (same arithmetic as coord_q5x4)
        int u, v;
        coord_q5(rc, xf, lk, hi_limit, tab, x0 + e, u, v);
        // q.iu[e] with a dynamic e would go through memory: select into the four registers instead
        q.iu[0] = e == 0 ? u : q.iu[0]; q.iu[1] = e == 1 ? u : q.iu[1]; q.iu[2] = e == 2 ? u : q.iu[2]; q.iu[3] = e == 3 ? u : q.iu[3];
        q.iv[0] = e == 0 ? v : q.iv[0]; q.iv[1] = e == 1 ? v : q.iv[1]; q.iv[2] = e == 2 ? v : q.iv[2]; q.iv[3] = e == 3 ? v : q.iv[3];
    }
    return q;
}

// The tile's source box (wave-uniform): tap positions bx0 .. bx1, by0 .. by1 (with their right and lower neighbours) are held
// in LDS, row by0 + i at byte i * RL_PITCH, column byte offset (3 sx - bxB).
struct RlBox {
    bool on;
    int bx0, by0;          // first tap position held
    unsigned NXq, NYq;     // Q5 extents: positions with (iu - 32 bx0) < NXq and (iv - 32 by0) < NYq are held
    unsigned bxB;          // byte offset within a source row of the box's first byte (16-byte aligned)
    unsigned xoffB;        // 3 bx0 - bxB
    int nrows;             // source rows held
};

__device__ __forceinline__ RlBox rl_box(const int* slot, int sw, int sh)
{
    RlBox B;
    const int mnu = __builtin_amdgcn_readfirstlane(slot[0]), mxu = __builtin_amdgcn_readfirstlane(slot[1]);
    const int mnv = __builtin_amdgcn_readfirstlane(slot[2]), mxv = __builtin_amdgcn_readfirstlane(slot[3]);
    // +-2 positions of margin (the nodes stop 0.6 px short of the tile's edge columns), clamped to the positions whose
    // taps all lie inside the frame; the subtraction order avoids overflow for the reset values INT_MAX / INT_MIN
    B.bx0 = max(mnu, 2) - 2; const int bx1 = min(mxu, sw - 4) + 2;
    B.by0 = max(mnv, 2) - 2; const int by1 = min(mxv, sh - 4) + 2;
    B.bxB = ((unsigned)B.bx0*3u) & ~15u;
    B.xoffB = (unsigned)B.bx0*3u - B.bxB;
    B.nrows = by1 - B.by0 + 2;
    const long long nbytes = (long long)(bx1 + 2)*3 - (long long)B.bxB;
    B.on = sw >= 4 && sh >= 4 && B.bx0 <= bx1 && B.by0 <= by1 && nbytes <= RL_PITCH && B.nrows <= RL_ROWS;
    B.NXq = B.on ? (unsigned)(bx1 - B.bx0 + 1) << INTER_BITS : 0u;
    B.NYq = B.on ? (unsigned)(by1 - B.by0 + 1) << INTER_BITS : 0u;
    return B;
}

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// One wave (the producer): copy the box rows (four per instruction: lane l fetches 16 bytes at column chunk l % 16 of row
// 4 i + l / 16) from the frame into the LDS buffer, asynchronously (global_load_lds_dwordx4: the destination is M0 +
// 16 * lane).  A chunk that would start at or past `limit` -- the last 16-byte chunk of the frame -- is read from there
// instead: its bytes are never used.
// The instruction is issued from an asm statement on purpose: for the builtin the compiler assumes that every later read of
// the box array may depend on the copy in flight and waits for it -- and for every store issued since -- before each group
// of tap reads, which serialises the whole kernel.  Hidden from its bookkeeping, the copy is waited for once, by rl_wait()
// before the barrier that precedes the first read of that buffer.  M0 is saved and restored inside the statement.
__device__ __forceinline__ void rl_issue(const RlBox& B, const uint8_t* __restrict__ frame, unsigned sstep, unsigned limit, uint4* buf, int lane)
{
    if (!B.on) return;
    const unsigned lr = (unsigned)lane >> 4, lc = (unsigned)lane & 15u;
    const int ngroups = (B.nrows + 3) >> 2;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_ptr)buf;
#pragma unroll 1
    for (int i = 0; i < ngroups; i++) {
        const unsigned row = (unsigned)B.by0 + min(4u*(unsigned)i + lr, (unsigned)B.nrows - 1u);
        const unsigned off = min(row*sstep + B.bxB + lc*16u, limit);
        const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)i*1024u));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(lds_dst), "s"(frame) : "memory");
    }
}
__device__ __forceinline__ void rl_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Workgroup barrier that orders LDS traffic only (no wait for outstanding global stores, unlike __syncthreads()).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

// Interior pixel from the LDS box, u8: ru, rv = Q5 coordinates relative to (bx0, by0).  Three aligned dwords per source row
// cover the 6 tap bytes at any byte phase; v_alignbyte_b32 shifts them into the 8-byte window bilin_u8_rows expects.
template <bool HI16>
__device__ __forceinline__ void interp_u8_raw(const uint32_t* box, unsigned ru, unsigned rv, unsigned xoffB, int* out)
{
    const unsigned o = ((rv >> INTER_BITS) << 8) + __umul24(ru >> INTER_BITS, 3u) + xoffB;
    static_assert(RL_PITCH == 256, "row shift");
    const uint32_t* p = box + (o >> 2);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], e0 = p[RL_PITCH/4], e1 = p[RL_PITCH/4 + 1], e2 = p[RL_PITCH/4 + 2];
    const uint64_t ta = (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, o) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, o) << 32);
    const uint64_t tb = (uint64_t)__builtin_amdgcn_alignbyte(e1, e0, o) | ((uint64_t)__builtin_amdgcn_alignbyte(e2, e1, o) << 32);
    bilin_u8_rows<HI16>(ta, tb, ru, rv, out);
}

// K1 with RP coordinates, taps gathered from the frame.  Persistent; work items are (frame, 64 x 64 tile); thread (lx, ly) =
// (tid % 16, tid / 16) produces columns 4 lx .. 4 lx + 3 of tile rows ly, ly + 16, ly + 32, ly + 48.  Software pipeline with
// one barrier per work item k: one wave (in turn) runs rp_nodes for item k + 2, all threads run rp_rows for item k + 1
// (thread t: coefficient t % 8 of rows t / 8 and 32 + t / 8), then the pixels of item k.
#ifdef GF_TRACE_WG   /* tools/ builds only: start time, end time (wall clock, 100 MHz) and hardware id of every workgroup of the last launch */
__device__ unsigned long long g_wg_trace[8192][3];
#endif

#ifndef GF_RP_WAVES_U16
#define GF_RP_WAVES_U16 GF_RP_WAVES
#endif
template <int DEPTH, int BORDER, bool ROT = false>
__global__ void __launch_bounds__(BLOCK, DEPTH == GF_DEPTH_U16 ? GF_RP_WAVES_U16 : GF_RP_WAVES) k_undistort_rotate_rp(RemapArgs A, const Xf* __restrict__ xfs, const double* __restrict__ lens, LensK lk,
                                                                const double* __restrict__ rp_tab, unsigned tiles_pad, unsigned n_work,
                                                                typename RotT<ROT>::T rot)
{
#ifdef GF_RP_LENS_GLOBAL
    const double2* s_tab = (const double2*)lens;       // experiment: the lens table read where it lies (rp_nodes and the rare per-pixel tiles only)
#else
    __shared__ double2 s_tab[LENS_DOUBLES/2];
#endif
    __shared__ RpShared S;
#ifdef GF_TRACE_WG
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wg_trace[blockIdx.x][0] = wall_clock64(); g_wg_trace[blockIdx.x][2] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
#ifndef GF_RP_LENS_GLOBAL
    load_lens_table(lens, s_tab);
#endif
    constexpr bool HI16 = DEPTH == GF_DEPTH_U8;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    rp_load_consts(rp_tab, S.K, tid, BLOCK);
    const int lx = tid & (RT_TPR - 1), ly = tid / RT_TPR;
    const double offx = rp_node_off(rp_tab, lane & 7), offy = rp_node_off(rp_tab, lane >> 3);
    double z[PXT];
#pragma unroll
    for (int e = 0; e < PXT; e++) z[e] = ((double)(lx*PXT + e) - (RP_HALF - 0.5))*(1.0/RP_HALF);      // exact
    const unsigned n_it = (n_work - blockIdx.x + gridDim.x - 1)/gridDim.x;       // work items of this workgroup (>= 1)
    auto decode = [&](unsigned k, unsigned& frame, int& tx, int& ty) -> bool {   // uniform
        if (k >= n_it) return false;
        const unsigned v = blockIdx.x + k*gridDim.x;
        frame = v / tiles_pad;
        return tile_of_padded_id(v - frame*tiles_pad, A.tiles_x, A.tiles_y, tx, ty);
    };
    auto nodes = [&](unsigned k) {               // one wave
        unsigned frame; int tx, ty;
        if (decode(k, frame, tx, ty)) rp_nodes<0>(xfs[frame], lk, s_tab, offx, offy, lane & 7, lane >> 3, tx, ty, S.U, S.K, S.C2[k & 1], &S.ok[k & 3], nullptr);
    };
    auto rows = [&](unsigned k) {                // all threads
        unsigned frame; int tx, ty;
        if (decode(k, frame, tx, ty)) rp_rows(S.C2[k & 1], S.K, tid & 7, tid >> 3, RT/2, 2, S.C[k & 1]);
    };
    __syncthreads();
    if (wave == 0) { nodes(0); asm volatile("" ::: "memory"); nodes(1); }      // one wave: rp_nodes' scratch is shared
    __syncthreads();
    rows(0);
    lds_barrier();
#pragma unroll 1
    for (unsigned k = 0; k < n_it; k++) {
        if (wave == (int)(k & 3)) nodes(k + 2);
        rows(k + 1);
        unsigned frame; int tx, ty;
        if (decode(k, frame, tx, ty)) {
            const int x0 = tx*RT + lx*PXT;
            const Xf& xf = xfs[frame];
            const SrcImg s{A.src + (size_t)frame*A.sframe, A.sw, A.sh, A.sstep};
            const bool ok = S.ok[k & 3] != 0;
            const int cb = (int)(k & 1);
            if (x0 < A.dw) {
                const RpFrameC fc = rp_frame_consts(xf);
#pragma unroll 1
                for (int pass = 0; pass < RT_NPASS; pass++) {
                    const int rr = pass*RT_RPP + ly, y = ty*RT + rr;
                    if (y >= A.dh) break;
                    int iu[PXT], iv[PXT], px[PXT][3];
                    if (ok) rp_coords(rp_row_consts(fc, y), S.C[cb][rr], z, x0, ROUND_MAGIC, ROUND_MAGIC, iu, iv);
#ifdef GF_RP_COLD_CALL
                    else {
                        const Q5x4 q = coord_q5x4_cold(&xf, lk.k[0], lk.k[1], lk.k[2], lk.k[3], lk.bound, s_tab, y, x0);
#pragma unroll
                        for (int e = 0; e < PXT; e++) { iu[e] = q.iu[e]; iv[e] = q.iv[e]; }
                    }
#else
                    else coord_q5x4(row_consts(xf, y), xf, lk, lens_hi_limit(xf, lk), s_tab, x0, iu, iv);
#endif
                    unsigned keep;
                    if constexpr (ROT) keep = remap4_rot<DEPTH, BORDER, HI16>(s, rot, iu, iv, A.cval, px);       // A.sw, A.sh: the rotated frame's size
                    else               keep = remap4<DEPTH, BORDER, HI16>(s, iu, iv, A.cval, px);
                    uint8_t* drow = A.dst + (size_t)frame*A.dframe + (size_t)y*A.dstep;
                    store_px<DEPTH, HI16>(drow, x0, A.dw, px, keep, A.dst_vec_ok);
                }
            }
        }
#ifndef GF_ABL_NOBARRIER     /* tools/ ablation builds only (racy coefficient buffers, wrong pixels): the per-item barrier costs 2.8 % */
        lds_barrier();
#endif
    }
#ifdef GF_TRACE_WG
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_wg_trace[blockIdx.x][1] = wall_clock64();
#endif
}

// K1-RL: RP coordinates + source taps staged through LDS (u8).  A workgroup is four consumer waves and one producer wave:
//   consumers (threads 0..255): the pixels of work item k -- coordinates from C[k % 3], taps from box buffer k & 1 when all
//       four pixels of a thread lie inside the box, otherwise straight from the frame (the box is an estimate from the
//       tile's rim nodes; results never depend on it) -- then the barrier;
//   producer (wave 4), during the same interval: starts the LDS-DMA of work item k + 1's source box into buffer (k + 1) & 1
//       (rl_issue; the box extent was found one interval earlier), computes coefficients, flag and box extent of work item
//       k + 2 (rp_nodes + rp_rows; lens table read from global memory) while the DMA is in flight, waits for the DMA
//       (s_waitcnt vmcnt(0): the producer has no stores of its own), then the barrier.
// One barrier per work item orders everything: C is triple-buffered (k read, k + 1 waiting, k + 2 written), the box
// double-buffered, flags and box extents live in four slots.  The consumers never wait on memory and issue no LDS-DMA (each
// 1 KB LDS-DMA instruction holds the issuing wave for ~100 cycles).
constexpr int BLOCK_RL = BLOCK + 64;
struct RlShared {
    RpConst K;
    double U[RP_N][RP_N];           // producer's scratch
    double C2[RP_N][RP_N];
    double C[3][RT][RP_N];
    int ok[4];
    int box[4][4];                  // [slot][min u, max u, min v, max v]
};

template <int BORDER>
__global__ void __launch_bounds__(BLOCK_RL, (GF_RL_WGS*(BLOCK_RL/64) + 3)/4) k_undistort_rotate_rl(RemapArgs A, const Xf* __restrict__ xfs, const double* __restrict__ lens, LensK lk,
                                                                   const double* __restrict__ rp_tab, unsigned tiles_pad, unsigned n_work)
{
    constexpr int DEPTH = GF_DEPTH_U8;
    constexpr bool HI16 = true;
    __shared__ RlShared S;
    __shared__ uint4 s_box[2][RL_ROWS*RL_PITCH/16];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const bool producer = __builtin_amdgcn_readfirstlane(tid >> 6) == BLOCK/64;
    rp_load_consts(rp_tab, S.K, tid, BLOCK_RL);
    if (tid < 16) (&S.box[0][0])[tid] = (tid & 1) ? INT_MIN : INT_MAX;
    __syncthreads();
    const unsigned n_it = (n_work - blockIdx.x + gridDim.x - 1)/gridDim.x;       // work items of this workgroup (>= 1)
    auto decode = [&](unsigned k, unsigned& frame, int& tx, int& ty) -> bool {   // uniform
        if (k >= n_it) return false;
        const unsigned v = blockIdx.x + k*gridDim.x;
        frame = v / tiles_pad;
        return tile_of_padded_id(v - frame*tiles_pad, A.tiles_x, A.tiles_y, tx, ty);
    };

    if (producer) {
        const double2* tab = (const double2*)lens;
        const int a = lane & (RP_N - 1), m = lane >> 3;
        const double offx = rp_node_off(rp_tab, a), offy = rp_node_off(rp_tab, m);
        const unsigned limit = ((unsigned)(A.sh - 1)*A.sstep + (unsigned)A.sw*3u - 16u) & ~15u;      // frames hold >= 32 bytes (host-checked)
        auto prepare = [&](unsigned k) {         // coefficients, flag and box extent of work item k (if it exists)
            unsigned frame; int tx, ty;
            if (!decode(k, frame, tx, ty)) return;
            int* slot = S.box[k & 3];
            if (lane < 4) slot[lane] = (lane & 1) ? INT_MIN : INT_MAX;     // same wave as the atomics: LDS keeps program order
            asm volatile("" ::: "memory");
            rp_nodes<1>(xfs[frame], lk, tab, offx, offy, a, m, tx, ty, S.U, S.K, S.C2, &S.ok[k & 3], slot);
            asm volatile("" ::: "memory");
            rp_rows(S.C2, S.K, a, m, RP_N, RT/RP_N, S.C[k % 3]);
            asm volatile("" ::: "memory");
        };
        auto issue = [&](unsigned k) {           // LDS-DMA of work item k's box
            unsigned frame; int tx, ty;
            if (!decode(k, frame, tx, ty)) return;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the slot's atomics have landed
            rl_issue(rl_box(S.box[k & 3], A.sw, A.sh), A.src + (size_t)frame*A.sframe, A.sstep, limit, s_box[k & 1], lane);
        };
        prepare(0);
        issue(0);
        prepare(1);
        rl_wait();
        lds_barrier();
#pragma unroll 1
        for (unsigned k = 0; k < n_it; k++) {
            issue(k + 1);
            prepare(k + 2);
            rl_wait();
            lds_barrier();
        }
        return;
    }

    // consumers
    double z[PXT];
    const int lx = tid & (RT_TPR - 1), ly = tid / RT_TPR;
#pragma unroll
    for (int e = 0; e < PXT; e++) z[e] = ((double)(lx*PXT + e) - (RP_HALF - 0.5))*(1.0/RP_HALF);
    lds_barrier();
#pragma unroll 1
    for (unsigned k = 0; k < n_it; k++) {
        unsigned frame; int tx, ty;
        if (decode(k, frame, tx, ty)) {
            const int x0 = tx*RT + lx*PXT;
            const Xf xf = xfs[frame];
            const SrcImg s{A.src + (size_t)frame*A.sframe, A.sw, A.sh, A.sstep};
            const bool ok = S.ok[k & 3] != 0;
            RlBox B; B.on = false; B.bx0 = B.by0 = 0; B.NXq = B.NYq = 0; B.xoffB = 0;
            if (ok) B = rl_box(S.box[k & 3], A.sw, A.sh);
            const int X0q = B.on ? B.bx0 << INTER_BITS : 0, Y0q = B.on ? B.by0 << INTER_BITS : 0;
            const double mu = ROUND_MAGIC - (double)X0q, mv = ROUND_MAGIC - (double)Y0q;
            const uint32_t* box = (const uint32_t*)s_box[k & 1];
            const int cb = (int)(k % 3);
            if (x0 < A.dw) {
#pragma unroll 1
                for (int pass = 0; pass < RT_NPASS; pass++) {
                    const int rr = pass*RT_RPP + ly, y = ty*RT + rr;
                    if (y >= A.dh) break;
                    int iu[PXT], iv[PXT], px[PXT][3];
                    unsigned keep = 0;
                    bool done = false;
                    if (ok) {
                        rp_coords(rp_row_consts(rp_frame_consts(xf), y), S.C[cb][rr], z, x0, mu, mv, iu, iv);
                        const unsigned rumax = max(max((unsigned)iu[0], (unsigned)iu[1]), max((unsigned)iu[2], (unsigned)iu[3]));
                        const unsigned rvmax = max(max((unsigned)iv[0], (unsigned)iv[1]), max((unsigned)iv[2], (unsigned)iv[3]));
                        if (rumax < B.NXq && rvmax < B.NYq) {
#pragma unroll
                            for (int e = 0; e < PXT; e++) interp_u8_raw<HI16>(box, (unsigned)iu[e], (unsigned)iv[e], B.xoffB, px[e]);
                            done = true;
                        } else {
#pragma unroll
                            for (int e = 0; e < PXT; e++) { iu[e] += X0q; iv[e] += Y0q; }
                        }
                    } else {
                        coord_q5x4(row_consts(xf, y), xf, lk, lens_hi_limit(xf, lk), (const double2*)lens, x0, iu, iv);
                    }
                    if (!done) keep = remap4<DEPTH, BORDER, HI16>(s, iu, iv, A.cval, px);
                    uint8_t* drow = A.dst + (size_t)frame*A.dframe + (size_t)y*A.dstep;
                    store_px<DEPTH, HI16>(drow, x0, A.dw, px, keep, A.dst_vec_ok);
                }
            }
        }
        lds_barrier();
    }
}

#ifdef GF_LB
// K1-LB (experiment, tools/build.sh lb): the RP kernel with its taps read from LDS.  Four waves, no producer: per HALF tile
// (64 x 32 output pixels) the workgroup copies the source box of the next half tile into the other of two LDS buffers with
// LDS-DMA (every wave issues its share, nothing waits for it by name) while all four waves compute the current half tile from
// the current buffer; s_waitcnt vmcnt(0) + one barrier per half tile hand the buffers over.  The boxes of a tile's two halves
// come from the wave that runs its rp_nodes, two work items ahead: source positions of the rim nodes of the half's four node
// rows and of the point between node rows 3 and 4 (tile row 31.5), merged with LDS atomics; +- 3 positions of margin.  A thread
// whose taps are not all inside takes the general per-pixel path on the frame, so results never depend on the estimate.  The
// lens table stays in global memory (only rp_nodes and the per-pixel fallback read it): 2 x 12 KB of boxes + 14.8 KB of
// coefficients = 4 workgroups per CU.
constexpr int LB_ROWS = 46;
#ifdef GF_LB_STATS
__device__ unsigned long long g_lb_stats[4];       // thread-passes from LDS, from the frame (box missed), half tiles with a box, without
#endif
__device__ __forceinline__ RlBox lb_box(const int* slot, int sw, int sh)     // slot: min u, max u, min v, max v (wave-uniform values)
{
    const int4 e4 = *(const int4*)slot;
    const int e0 = __builtin_amdgcn_readfirstlane(e4.x), e1 = __builtin_amdgcn_readfirstlane(e4.y);
    const int e2 = __builtin_amdgcn_readfirstlane(e4.z), e3 = __builtin_amdgcn_readfirstlane(e4.w);
    RlBox B;
    B.bx0 = max(e0, 3) - 3; const int bx1 = min(e1, sw - 5) + 3;
    B.by0 = max(e2, 3) - 3; const int by1 = min(e3, sh - 5) + 3;
    B.bxB = ((unsigned)B.bx0*3u) & ~15u;
    B.xoffB = (unsigned)B.bx0*3u - B.bxB;
    B.nrows = by1 - B.by0 + 2;
    const long long nbytes = (long long)(bx1 + 2)*3 - (long long)B.bxB;
    B.on = sw >= 8 && sh >= 8 && B.bx0 <= bx1 && B.by0 <= by1 && nbytes <= RL_PITCH && B.nrows <= LB_ROWS;
    B.NXq = B.on ? (unsigned)(bx1 - B.bx0 + 1) << INTER_BITS : 0u;
    B.NYq = B.on ? (unsigned)(by1 - B.by0 + 1) << INTER_BITS : 0u;
    return B;
}

// rp_nodes' wave, after the node values: half boxes of the tile into slot[0..3] (tile rows 0..31) and slot[4..7] (rows 32..63).
// Node row m sits at tile row 31.5 + 32 cos((2m + 1) pi / 16): rows 4..7 are the upper half, 0..3 the lower one.
__device__ __forceinline__ void lb_node_boxes(const Xf& xf, double X, double Y, double G, int a, int m, int all_ok, int* slot)
{
    const double u32 = __fma_rn(X*G, xf.f[0]*32.0, xf.c[0]*32.0), v32 = __fma_rn(Y*G, xf.f[1]*32.0, xf.c[1]*32.0);
    const int su = (int)fmin(fmax(u32*(1.0/32), -1e6), 1e6), sv = (int)fmin(fmax(v32*(1.0/32), -1e6), 1e6);
    const int su_n = __shfl_down(su, 8), sv_n = __shfl_down(sv, 8);        // node row m + 1, same column
    if (all_ok && (a == 0 || a == RP_N - 1)) {
        int* h = slot + (m >= RP_N/2 ? 0 : 4);
        atomicMin(&h[0], su - 1); atomicMax(&h[1], su + 1); atomicMin(&h[2], sv - 1); atomicMax(&h[3], sv + 1);     // (int) truncates towards zero
        if (m == RP_N/2 - 1) {              // tile row 31.5: between this node row (37.7) and the next (25.3); goes into both halves
            const int mu = (su + su_n) >> 1, mv = (sv + sv_n) >> 1;
            atomicMin(&slot[0], mu - 1); atomicMax(&slot[1], mu + 2); atomicMin(&slot[2], mv - 1); atomicMax(&slot[3], mv + 2);
            atomicMin(&slot[4], mu - 1); atomicMax(&slot[5], mu + 2); atomicMin(&slot[6], mv - 1); atomicMax(&slot[7], mv + 2);
        }
    }
}

// This wave's share of the copy of box B (groups of four rows: wave w takes groups w, w + 4, w + 8).
__device__ __forceinline__ void lb_issue(const RlBox& B, const uint8_t* __restrict__ frame, unsigned sstep, unsigned limit, uint4* buf, int lane, int wave)
{
    if (!B.on) return;
    const unsigned lr = (unsigned)lane >> 4, lc = (unsigned)lane & 15u;
    const int ngroups = (B.nrows + 3) >> 2;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_ptr)buf;
#pragma unroll 1
    for (int i = wave; i < ngroups; i += BLOCK/64) {
        const unsigned row = (unsigned)B.by0 + min(4u*(unsigned)i + lr, (unsigned)B.nrows - 1u);
        const unsigned off = min(row*sstep + B.bxB + lc*16u, limit);
        const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)i*1024u));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(lds_dst), "s"(frame) : "memory");
    }
}

// rp_nodes as a call, so that its registers do not count against the half-tile loop
__device__ __noinline__ void lb_nodes_cold(const Xf* xf, const LensK* lk, const double2* tab, const double* rp_tab, int lane, int tx, int ty,
                                           RpShared* S, int c2buf, int okslot, int* slot)
{
    rp_nodes<2>(*xf, *lk, tab, rp_node_off(rp_tab, lane & 7), rp_node_off(rp_tab, lane >> 3), lane & 7, lane >> 3, tx, ty, S->U, S->K, S->C2[c2buf], &S->ok[okslot], slot);
}
#ifndef GF_LB_WAVES
#define GF_LB_WAVES GF_RP_WAVES
#endif
template <int BORDER>
__global__ void __launch_bounds__(BLOCK, GF_LB_WAVES) k_undistort_rotate_lb(RemapArgs A, const Xf* __restrict__ xfs, const double* __restrict__ lens, LensK lk,
                                                                           const double* __restrict__ rp_tab, unsigned tiles_pad, unsigned n_work)
{
    constexpr int DEPTH = GF_DEPTH_U8;
    constexpr bool HI16 = true;
    __shared__ RpShared S;
    __shared__ __attribute__((aligned(16))) int s_bx[4][8];     // [slot][half][min u, max u, min v, max v]
    __shared__ uint4 s_box[2][(LB_ROWS + 2)*RL_PITCH/16];       // groups of four rows: the last group may run two rows past LB_ROWS
    const double2* s_tab = (const double2*)lens;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    rp_load_consts(rp_tab, S.K, tid, BLOCK);
    const int lx = tid & (RT_TPR - 1), ly = tid / RT_TPR;
    double z[PXT];
#pragma unroll
    for (int e = 0; e < PXT; e++) z[e] = ((double)(lx*PXT + e) - (RP_HALF - 0.5))*(1.0/RP_HALF);      // exact
    const unsigned limit = ((unsigned)(A.sh - 1)*A.sstep + (unsigned)A.sw*3u - 16u) & ~15u;            // frames hold >= 32 bytes (host-checked)
    const unsigned n_it = (n_work - blockIdx.x + gridDim.x - 1)/gridDim.x;       // work items of this workgroup (>= 1)
    auto decode = [&](unsigned k, unsigned& frame, int& tx, int& ty) -> bool {   // uniform
        if (k >= n_it) return false;
        const unsigned v = blockIdx.x + k*gridDim.x;
        frame = v / tiles_pad;
        return tile_of_padded_id(v - frame*tiles_pad, A.tiles_x, A.tiles_y, tx, ty);
    };
    auto nodes = [&](unsigned k) {               // one wave: coefficients, flag and half boxes of work item k
        unsigned frame; int tx, ty;
        int* slot = s_bx[k & 3];
        if (lane < 8) slot[lane] = (lane & 1) ? INT_MIN : INT_MAX;      // LDS operations of one wave execute in program order
        asm volatile("" ::: "memory");
#ifndef GF_LB_NONODES     /* register-count probe only */
        if (decode(k, frame, tx, ty))
            lb_nodes_cold(&xfs[frame], &lk, s_tab, rp_tab, lane, tx, ty, &S, (int)(k & 1), (int)(k & 3), slot);
#endif
    };
    __syncthreads();
    if (wave == 0) { nodes(0); asm volatile("" ::: "memory"); nodes(1); }
    __syncthreads();
    {
        unsigned f0; int a0, b0;
        if (decode(0, f0, a0, b0)) rp_rows(S.C2[0], S.K, tid & 7, tid >> 3, RT/2, 2, S.C[0]);
    }
    // work item k and k + 1 as decoded once: (have, frame, tx, ty)
    unsigned frame = 0, frame1 = 0; int tx = 0, ty = 0, tx1 = 0, ty1 = 0;
    bool have = decode(0, frame, tx, ty), have1 = false;
    RlBox Bc; Bc.on = false; Bc.bx0 = Bc.by0 = 0; Bc.NXq = Bc.NYq = 0; Bc.bxB = 0; Bc.xoffB = 0; Bc.nrows = 0;
    if (have && S.ok[0] != 0) Bc = lb_box(&s_bx[0][0], A.sw, A.sh);
    if (have) lb_issue(Bc, A.src + (size_t)frame*A.sframe, A.sstep, limit, s_box[0], lane, wave);
    rl_wait();
    lds_barrier();
    int buf = 0;
#pragma unroll 1
    for (unsigned k = 0; k < n_it; k++) {
        have1 = decode(k + 1, frame1, tx1, ty1);
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            // the next half tile's box: start its copy into the other buffer, then this half tile's work
            RlBox Bn; Bn.on = false; Bn.bx0 = Bn.by0 = 0; Bn.NXq = Bn.NYq = 0; Bn.bxB = 0; Bn.xoffB = 0; Bn.nrows = 0;
            if (half == 0) {
                if (have && S.ok[k & 3] != 0) Bn = lb_box(&s_bx[k & 3][4], A.sw, A.sh);
                if (have) lb_issue(Bn, A.src + (size_t)frame*A.sframe, A.sstep, limit, s_box[buf ^ 1], lane, wave);
                if (wave == (int)(k & 3)) nodes(k + 2);
                if (have1) rp_rows(S.C2[(k + 1) & 1], S.K, tid & 7, tid >> 3, RT/2, 2, S.C[(k + 1) & 1]);
            } else {
                if (have1 && S.ok[(k + 1) & 3] != 0) Bn = lb_box(&s_bx[(k + 1) & 3][0], A.sw, A.sh);
                if (have1) lb_issue(Bn, A.src + (size_t)frame1*A.sframe, A.sstep, limit, s_box[buf ^ 1], lane, wave);
            }
            if (have) {
                const int x0 = tx*RT + lx*PXT;
                const Xf& xf = xfs[frame];
                const SrcImg s{A.src + (size_t)frame*A.sframe, A.sw, A.sh, A.sstep};
                const bool ok = S.ok[k & 3] != 0;
                const int cb = (int)(k & 1);
                const int X0q = Bc.on ? Bc.bx0 << INTER_BITS : 0, Y0q = Bc.on ? Bc.by0 << INTER_BITS : 0;
                const double mu = ROUND_MAGIC - (double)X0q, mv = ROUND_MAGIC - (double)Y0q;
                const uint32_t* box = (const uint32_t*)s_box[buf];
                if (x0 < A.dw) {
                    const RpFrameC fc = rp_frame_consts(xf);
#pragma unroll 1
                    for (int pass = 2*half; pass < 2*half + 2; pass++) {
                        const int rr = pass*RT_RPP + ly, y = ty*RT + rr;
                        if (y >= A.dh) break;
                        int iu[PXT], iv[PXT], px[PXT][3];
                        unsigned keep = 0;
                        bool done = false;
                        if (ok) {
                            rp_coords(rp_row_consts(fc, y), S.C[cb][rr], z, x0, mu, mv, iu, iv);         // relative to the box
                            const unsigned rumax = max(max((unsigned)iu[0], (unsigned)iu[1]), max((unsigned)iu[2], (unsigned)iu[3]));
                            const unsigned rvmax = max(max((unsigned)iv[0], (unsigned)iv[1]), max((unsigned)iv[2], (unsigned)iv[3]));
                            if (rumax < Bc.NXq && rvmax < Bc.NYq) {
#pragma unroll
                                for (int e = 0; e < PXT; e++) interp_u8_raw<HI16>(box, (unsigned)iu[e], (unsigned)iv[e], Bc.xoffB, px[e]);
                                done = true;
#ifdef GF_LB_STATS
                                atomicAdd(&g_lb_stats[0], 1ull);
#endif
                            } else {
#pragma unroll
                                for (int e = 0; e < PXT; e++) { iu[e] += X0q; iv[e] += Y0q; }
#ifdef GF_LB_STATS
                                atomicAdd(&g_lb_stats[1], 1ull);
#endif
                            }
                        } else {
                            const Q5x4 q = coord_q5x4_cold(&xf, lk.k[0], lk.k[1], lk.k[2], lk.k[3], lk.bound, s_tab, y, x0);
#pragma unroll
                            for (int e = 0; e < PXT; e++) { iu[e] = q.iu[e]; iv[e] = q.iv[e]; }
                        }
                        if (!done) {            // the general per-pixel path on the frame (a call per pixel: rare)
#pragma unroll 1
                            for (int e = 0; e < PXT; e++) {
                                const int sx = (int)(int16_t)(iu[e] >> INTER_BITS), sy = (int)(int16_t)(iv[e] >> INTER_BITS);   // Q1: wrapping cast
                                const Px3 o = bilinear_px<DEPTH, BORDER>(s, sx, sy, (iv[e] & 31)*32 + (iu[e] & 31), A.cval[0], A.cval[1], A.cval[2]);
                                const int c0 = o.c0 << 16, c1 = o.c1 << 16, c2 = o.c2 << 16;
                                // px[e] with a dynamic e would go through memory: select into the registers instead
#pragma unroll
                                for (int j = 0; j < PXT; j++) if (j == e) { px[j][0] = c0; px[j][1] = c1; px[j][2] = c2; }
                                if (!o.ok) keep |= 1u << e;
                            }
                        }
                        uint8_t* drow = A.dst + (size_t)frame*A.dframe + (size_t)y*A.dstep;
                        store_px<DEPTH, HI16>(drow, x0, A.dw, px, keep, A.dst_vec_ok);
                    }
                }
            }
#ifdef GF_LB_STATS
            if (tid == 0) atomicAdd(&g_lb_stats[Bc.on ? 2 : 3], 1ull);
#endif
            // this wave's share of the copy must have landed.  Vector-memory operations retire in order, and the copy was issued
            // before this half tile's two stores: when both were issued (a tile that lies fully inside the output) it is enough to
            // wait for everything but the two youngest operations, so that the stores' acknowledgements are not waited for
            if (have && tx*RT + RT <= A.dw && ty*RT + half*(RT/2) + RT/2 <= A.dh) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else rl_wait();
            lds_barrier();          // everybody is done with `buf`; the other buffer is complete
            buf ^= 1;
            Bc = Bn;
        }
        have = have1; frame = frame1; tx = tx1; ty = ty1;
    }
}
#endif

// K2 with RP coordinates (CV_16SC2 maps): one 64 x 64 tile per workgroup.
__global__ void __launch_bounds__(BLOCK) k_fisheye_maps_rp(Xf xf, LensK lk, const double* __restrict__ lens, const double* __restrict__ rp_tab, int w, int h,
                                                            uint8_t* map1, size_t m1step, uint8_t* map2, size_t m2step)
{
    __shared__ double2 s_tab[LENS_DOUBLES/2];
    __shared__ RpShared S;
    load_lens_table(lens, s_tab);
    const int tid = (int)threadIdx.x, lane = tid & 63;
    rp_load_consts(rp_tab, S.K, tid, BLOCK);
    const int lx = tid & (RT_TPR - 1), ly = tid / RT_TPR;
    const int tx = blockIdx.x, ty = blockIdx.y;
    __syncthreads();
    if (tid < 64) rp_nodes<0>(xf, lk, s_tab, rp_node_off(rp_tab, lane & 7), rp_node_off(rp_tab, lane >> 3), lane & 7, lane >> 3, tx, ty, S.U, S.K, S.C2[0], &S.ok[0], nullptr);
    __syncthreads();
    rp_rows(S.C2[0], S.K, tid & 7, tid >> 3, RT/2, 2, S.C[0]);
    __syncthreads();
    const int x0 = tx*RT + lx*PXT;
    if (x0 >= w) return;
    const bool ok = S.ok[0] != 0;
    double z[PXT];
#pragma unroll
    for (int e = 0; e < PXT; e++) z[e] = ((double)(lx*PXT + e) - (RP_HALF - 0.5))*(1.0/RP_HALF);
#pragma unroll 1
    for (int pass = 0; pass < RT_NPASS; pass++) {
        const int rr = pass*RT_RPP + ly, y = ty*RT + rr;
        if (y >= h) break;
        int iu[PXT], iv[PXT];
        if (ok) rp_coords(rp_row_consts(rp_frame_consts(xf), y), S.C[0][rr], z, x0, ROUND_MAGIC, ROUND_MAGIC, iu, iv);
        else coord_q5x4(row_consts(xf, y), xf, lk, lens_hi_limit(xf, lk), s_tab, x0, iu, iv);
        int16_t* m1 = (int16_t*)(map1 + (size_t)y*m1step);
        uint16_t* m2 = (uint16_t*)(map2 + (size_t)y*m2step);
#pragma unroll
        for (int e = 0; e < PXT; e++)
            if (x0 + e < w) {
                *(uint32_t*)(m1 + 2*(x0 + e)) = (uint32_t)(uint16_t)(iu[e] >> INTER_BITS) | ((uint32_t)(uint16_t)(iv[e] >> INTER_BITS) << 16);   // (short) cast wraps (Q1)
                m2[x0 + e] = (uint16_t)((iv[e] & 31)*32 + (iu[e] & 31));
            }
    }
}

// K1b: fused + hyperlapse blend: out = trunc(sum_b (1/N) * remap_b) in float64, reference order.
// POW2 (n_blend a power of two): integer sum and shift, exactly the float64 result -- see k_undistort_rotate_blend_rp.
template <int DEPTH, int BORDER, bool POW2>
__global__ void __launch_bounds__(BLOCK) k_undistort_rotate_blend(RemapArgs A, const Xf* __restrict__ xfs, const double* __restrict__ lens,
                                                                   LensK lk, int frames_per_group, int n_blend)
{
    __shared__ double2 s_tab[LENS_DOUBLES/2];
    load_lens_table(lens, s_tab);
    int tx, ty;
    tile_of_block(blockIdx.x, A.tiles_x, A.tiles_y, tx, ty);
    const int g = blockIdx.y;
    const int lx = threadIdx.x & (TPR - 1), ly = threadIdx.x / TPR;
    const int x0 = tx*TILE_W + lx*PXT;
    if (x0 >= A.dw) return;
    const double wgt = 1.0/(double)n_blend;       // Python: 1/(self.hyperlapse_num_blended_frames)
    const int shift = 31 - __builtin_clz((unsigned)n_blend);      // log2(n_blend) when POW2
    typedef typename std::conditional<POW2, unsigned, double>::type AccT;
#pragma unroll 1
    for (int pass = 0; pass < TILE_H/PASS_H; pass++) {
        const int y = ty*TILE_H + pass*PASS_H + ly;
        if (y >= A.dh) break;
        AccT acc[PXT][3];
#pragma unroll
        for (int i = 0; i < PXT; i++) acc[i][0] = acc[i][1] = acc[i][2] = 0;
#pragma unroll 1
        for (int b = 0; b < n_blend; b++) {
            const int frame = g*frames_per_group + b;
            const Xf& xf = xfs[frame];
            const unsigned hi_limit = lens_hi_limit(xf, lk);
            const SrcImg s{A.src + (size_t)frame*A.sframe, A.sw, A.sh, A.sstep};
            const RowC rc = row_consts(xf, y);
            int iu[PXT], iv[PXT], o[PXT][3];
            coord_q5x4(rc, xf, lk, hi_limit, s_tab, x0, iu, iv);
            remap4<DEPTH, BORDER>(s, iu, iv, A.cval, o);
#pragma unroll
            for (int i = 0; i < PXT; i++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if constexpr (POW2) acc[i][c] += (unsigned)o[i][c];
                    else                acc[i][c] = __dadd_rn(acc[i][c], __dmul_rn(wgt, (double)o[i][c]));
                }
        }
        int px[PXT][3];
#pragma unroll
        for (int i = 0; i < PXT; i++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if constexpr (POW2) px[i][c] = (int)(acc[i][c] >> shift);
                else                px[i][c] = (int)acc[i][c];          // astype(uint): truncation
            }
        uint8_t* drow = A.dst + (size_t)g*A.dframe + (size_t)y*A.dstep;
        store_px<DEPTH>(drow, x0, A.dw, px, 0u, A.dst_vec_ok);
    }
}

// K4: remap from materialised CV_16SC2 + CV_16UC1 maps.
template <int DEPTH, int BORDER>
__global__ void __launch_bounds__(BLOCK) k_remap_linear(RemapArgs A, const uint8_t* __restrict__ map1, size_t m1step,
                                                         const uint8_t* __restrict__ map2, size_t m2step)
{
    int tx, ty;
    tile_of_block(blockIdx.x, A.tiles_x, A.tiles_y, tx, ty);
    const int lx = threadIdx.x & (TPR - 1), ly = threadIdx.x / TPR;
    const int x0 = tx*TILE_W + lx*PXT;
    if (x0 >= A.dw) return;
    const SrcImg s{A.src, A.sw, A.sh, A.sstep};
#pragma unroll 1
    for (int pass = 0; pass < TILE_H/PASS_H; pass++) {
        const int y = ty*TILE_H + pass*PASS_H + ly;
        if (y >= A.dh) break;
        const int16_t* m1 = (const int16_t*)(map1 + (size_t)y*m1step);
        const uint16_t* m2 = (const uint16_t*)(map2 + (size_t)y*m2step);
        int iu[PXT], iv[PXT];
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            const int x = min(x0 + i, A.dw - 1);
            const int a = m2[x] & 1023;
            iu[i] = ((int)m1[x*2] << INTER_BITS) | (a & 31);          // back to the Q5 integers the map was made from
            iv[i] = ((int)m1[x*2+1] << INTER_BITS) | (a >> 5);
        }
        int px[PXT][3];
        const unsigned keep = remap4<DEPTH, BORDER>(s, iu, iv, A.cval, px);
        uint8_t* drow = A.dst + (size_t)y*A.dstep;
        store_px<DEPTH>(drow, x0, A.dw, px, keep, A.dst_vec_ok);
    }
}

// K1b with RP coordinates (the same functions and tile grid as k_undistort_rotate_rp, so that the fused blend equals the
// separate passes bit for bit in the RP / RL modes too): one 64 x 64 tile of one output frame per workgroup.  The
// coefficients of all n_blend <= RB (4 or RB_MAX) input frames are prepared first (four frames at a time, one wave per frame), then
// the pixels are produced row pass by row pass with the blend loop innermost, as in k_undistort_rotate_blend.
// POW2: n_blend is a power of two.  Then 1/n_blend, every product (1/n_blend) * pixel and every partial sum of the reference's
// float64 accumulation (frame_temp += (1/N) * frame.astype(float64)) are exact, so the truncated result is (sum of the pixels)
// >> log2(n_blend), and the sum is kept as an integer: 12 additions per input frame and pass instead of 12 conversions, 12
// float64 multiplies and 12 float64 additions.  Other n_blend keep the float64 arithmetic in the reference's order.
constexpr int RB_MAX = 8;
template <int DEPTH, int BORDER, int RB, bool POW2>
__global__ void __launch_bounds__(BLOCK) k_undistort_rotate_blend_rp(RemapArgs A, const Xf* __restrict__ xfs, const double* __restrict__ lens, LensK lk,
                                                                      const double* __restrict__ rp_tab, int frames_per_group, int n_blend)
{
    __shared__ double2 s_tab[LENS_DOUBLES/2];
    __shared__ RpConst K;
    __shared__ double U[BLOCK/64][RP_N][RP_N];
    __shared__ double C2[BLOCK/64][RP_N][RP_N];
    __shared__ double C[RB][RT][RP_N];
    __shared__ int okf[RB];
    __shared__ RpFrameC FC[RB];                  // the frames' constants, once per workgroup instead of once per pass and frame
    load_lens_table(lens, s_tab);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    rp_load_consts(rp_tab, K, tid, BLOCK);
    int tx, ty;
    tile_of_block(blockIdx.x, A.tiles_x, A.tiles_y, tx, ty);
    const int g = blockIdx.y;
    if (tid < n_blend) FC[tid] = rp_frame_consts(xfs[g*frames_per_group + tid]);
    const double offx = rp_node_off(rp_tab, lane & 7), offy = rp_node_off(rp_tab, lane >> 3);
    __syncthreads();
#pragma unroll 1
    for (int b0 = 0; b0 < n_blend; b0 += BLOCK/64) {
        if (b0 + wave < n_blend)
            rp_nodes<0>(xfs[g*frames_per_group + b0 + wave], lk, s_tab, offx, offy, lane & 7, lane >> 3, tx, ty, U[wave], K, C2[wave], &okf[b0 + wave], nullptr);
        __syncthreads();
        for (int j = 0; j < BLOCK/64 && b0 + j < n_blend; j++) rp_rows(C2[j], K, tid & 7, tid >> 3, RT/2, 2, C[b0 + j]);
        __syncthreads();
    }
    const int lx = tid & (RT_TPR - 1), ly = tid / RT_TPR;
    const int x0 = tx*RT + lx*PXT;
    if (x0 >= A.dw) return;
    double z[PXT];
#pragma unroll
    for (int e = 0; e < PXT; e++) z[e] = ((double)(lx*PXT + e) - (RP_HALF - 0.5))*(1.0/RP_HALF);
    const double wgt = 1.0/(double)n_blend;       // Python: 1/(self.hyperlapse_num_blended_frames)
    const int shift = 31 - __builtin_clz((unsigned)n_blend);      // log2(n_blend) when POW2
    typedef typename std::conditional<POW2, unsigned, double>::type AccT;
#pragma unroll 1
    for (int pass = 0; pass < RT_NPASS; pass++) {
        const int rr = pass*RT_RPP + ly, y = ty*RT + rr;
        if (y >= A.dh) break;
        AccT acc[PXT][3];
#pragma unroll
        for (int i = 0; i < PXT; i++) acc[i][0] = acc[i][1] = acc[i][2] = 0;
#pragma unroll 1
        for (int b = 0; b < n_blend; b++) {
            const int frame = g*frames_per_group + b;
            const Xf& xf = xfs[frame];
            const SrcImg s{A.src + (size_t)frame*A.sframe, A.sw, A.sh, A.sstep};
            int iu[PXT], iv[PXT], o[PXT][3];
            if (okf[b]) rp_coords(rp_row_consts(FC[b], y), C[b][rr], z, x0, ROUND_MAGIC, ROUND_MAGIC, iu, iv);
            else coord_q5x4(row_consts(xf, y), xf, lk, lens_hi_limit(xf, lk), s_tab, x0, iu, iv);
            remap4<DEPTH, BORDER>(s, iu, iv, A.cval, o);
#pragma unroll
            for (int i = 0; i < PXT; i++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if constexpr (POW2) acc[i][c] += (unsigned)o[i][c];
                    else                acc[i][c] = __dadd_rn(acc[i][c], __dmul_rn(wgt, (double)o[i][c]));
                }
        }
        int px[PXT][3];
#pragma unroll
        for (int i = 0; i < PXT; i++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if constexpr (POW2) px[i][c] = (int)(acc[i][c] >> shift);
                else                px[i][c] = (int)acc[i][c];          // astype(uint): truncation
            }
        uint8_t* drow = A.dst + (size_t)g*A.dframe + (size_t)y*A.dstep;
        store_px<DEPTH>(drow, x0, A.dw, px, 0u, A.dst_vec_ok);
    }
}

// K2: materialised maps.  One thread per output pixel and pass, 64 x 64 tiles (16 KB lens table per workgroup).
template <bool FIXED>
__global__ void __launch_bounds__(BLOCK) k_fisheye_maps(Xf xf, LensK lk, const double* __restrict__ lens, int w, int h,
                                                         uint8_t* map1, size_t m1step, uint8_t* map2, size_t m2step)
{
    __shared__ double2 s_tab[LENS_DOUBLES/2];
    load_lens_table(lens, s_tab);
    const unsigned hi_limit = lens_hi_limit(xf, lk);
    const int j = blockIdx.x*64 + (threadIdx.x & 63);
    if (j >= w) return;
#pragma unroll 1
    for (int pass = 0; pass < 16; pass++) {
        const int i = blockIdx.y*64 + pass*4 + (threadIdx.x >> 6);
        if (i >= h) break;
        const RowC rc = row_consts(xf, i);
        if (FIXED) {
            int iu, iv;
            coord_q5(rc, xf, lk, hi_limit, s_tab, j, iu, iv);
            int16_t* m1 = (int16_t*)(map1 + (size_t)i*m1step);
            uint16_t* m2 = (uint16_t*)(map2 + (size_t)i*m2step);
            // one 4-byte store for the (x,y) pair; (short) cast wraps (Q1)
            *(uint32_t*)(m1 + 2*j) = (uint32_t)(uint16_t)(iu >> INTER_BITS) | ((uint32_t)(uint16_t)(iv >> INTER_BITS) << 16);
            m2[j] = (uint16_t)((iv & 31)*32 + (iu & 31));
        } else {
            double X, Y, W;
            double2 o;
            unsigned qhi;
            coord_fast(rc, s_tab, j, X, Y, W, qhi, o.x, o.y);
            if (!lens_ok(W, qhi, hi_limit)) { ray_cold(rc, j, X, Y, W);
                o = fisheye_uv32_exact(xf.f[0], xf.f[1], xf.c[0], xf.c[1], lk.k[0], lk.k[1], lk.k[2], lk.k[3], X, Y, W); }
            ((float*)(map1 + (size_t)i*m1step))[j] = (float)(o.x*(1.0/32));
            ((float*)(map2 + (size_t)i*m2step))[j] = (float)(o.y*(1.0/32));
        }
    }
}

// K3: convertMaps(32F,32F -> 16SC2) with the reference's float32 pre-scaling.
__global__ void __launch_bounds__(BLOCK) k_convert_maps(const uint8_t* mx, size_t mxstep, const uint8_t* my, size_t mystep,
                                                        float sxn, float sxd, float syn, float syd, int scale_on,
                                                        int w, int h, uint8_t* map1, size_t m1step, uint8_t* map2, size_t m2step)
{
    const int j = blockIdx.x*64 + (threadIdx.x & 63), i = blockIdx.y*4 + (threadIdx.x >> 6);
    if (j >= w || i >= h) return;
    float fx = ((const float*)(mx + (size_t)i*mxstep))[j], fy = ((const float*)(my + (size_t)i*mystep))[j];
    if (scale_on) {
        fx = __fdiv_rn(__fmul_rn(fx, sxn), sxd);
        fy = __fdiv_rn(__fmul_rn(fy, syn), syd);
    }
    const int ix = cv_round_f(__fmul_rn(fx, 32.f)), iy = cv_round_f(__fmul_rn(fy, 32.f));
    int16_t* m1 = (int16_t*)(map1 + (size_t)i*m1step);
    *(uint32_t*)(m1 + 2*j) = (uint32_t)(uint16_t)sat_s16(ix >> INTER_BITS) | ((uint32_t)(uint16_t)sat_s16(iy >> INTER_BITS) << 16);
    ((uint16_t*)(map2 + (size_t)i*m2step))[j] = (uint16_t)((iy & 31)*32 + (ix & 31));
}

// K5: cv::remapBicubic (u8c3, Q15 table) from float32 maps.  One thread per output pixel.
__device__ __forceinline__ int border_interp(int p, int len, int border1)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (border1 == GF_BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (border1 == 4) {                                  // BORDER_REFLECT_101 (stands in for TRANSPARENT)
        if (len == 1) return 0;
        do { p = p < 0 ? -p : 2*len - 2 - p; } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    return -1;
}

// One bicubic output pixel, cv::remapBicubic semantics (imgwarp.cpp), any border case.  (isx, isy) = cvRound(32 * map).
__device__ __forceinline__ void cubic_px(const uint8_t* __restrict__ src, int sw, int sh, size_t sstep, uint8_t* D, int isx, int isy,
                                         const int16_t* __restrict__ tab, int border, int c0, int c1, int c2)
{
    const int a = (isy & 31)*32 + (isx & 31);
    const int sx = sat_s16(isx >> INTER_BITS) - 1, sy = sat_s16(isy >> INTER_BITS) - 1;
    const int16_t* w = tab + a*16;
    const int cval[3] = {c0, c1, c2};
    const unsigned width1 = max(sw - 3, 0), height1 = max(sh - 3, 0);
    if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        const uint8_t* S = src + (size_t)sy*sstep + sx*3;
        int sum[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // 12 bytes = 4 pixels x 3 channels: three unaligned dwords, all inside the row
            const uint8_t* q = S + (size_t)r*sstep;
            const uint32_t d0 = *(const u32_unaligned*)q, d1 = *(const u32_unaligned*)(q + 4), d2 = *(const u32_unaligned*)(q + 8);
            const int b[12] = {(int)(d0 & 0xff), (int)((d0 >> 8) & 0xff), (int)((d0 >> 16) & 0xff), (int)(d0 >> 24),
                               (int)(d1 & 0xff), (int)((d1 >> 8) & 0xff), (int)((d1 >> 16) & 0xff), (int)(d1 >> 24),
                               (int)(d2 & 0xff), (int)((d2 >> 8) & 0xff), (int)((d2 >> 16) & 0xff), (int)(d2 >> 24)};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int wt = w[r*4 + t];
                sum[0] += __mul24(b[t*3], wt); sum[1] += __mul24(b[t*3+1], wt); sum[2] += __mul24(b[t*3+2], wt);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { int v = (sum[k] + (1 << 14)) >> 15; D[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
        return;
    }
    const int border1 = border != GF_BORDER_TRANSPARENT ? border : 4;
    if (border == GF_BORDER_TRANSPARENT && ((unsigned)(sx + 1) >= (unsigned)sw || (unsigned)(sy + 1) >= (unsigned)sh)) return;
    if (border1 == GF_BORDER_CONSTANT && (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0)) {
        D[0] = (uint8_t)c0; D[1] = (uint8_t)c1; D[2] = (uint8_t)c2;
        return;
    }
    int xs[4], ys[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { xs[i] = border_interp(sx + i, sw, border1); ys[i] = border_interp(sy + i, sh, border1); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int cv = cval[k];
        int sum = cv*32768;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (ys[i] < 0) continue;
            const uint8_t* S = src + (size_t)ys[i]*sstep + k;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (xs[q] >= 0) sum += ((int)S[xs[q]*3] - cv)*w[i*4 + q];
        }
        int v = (sum + (1 << 14)) >> 15;
        D[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

__global__ void __launch_bounds__(BLOCK) k_remap_cubic_u8c3(const uint8_t* __restrict__ src, int sw, int sh, size_t sstep,
                                                             uint8_t* dst, int dw, int dh, size_t dstep,
                                                             const uint8_t* __restrict__ mx, size_t mxstep,
                                                             const uint8_t* __restrict__ my, size_t mystep,
                                                             const int16_t* __restrict__ tab, int border, int c0, int c1, int c2)
{
    const int dx = blockIdx.x*64 + (threadIdx.x & 63), dy = blockIdx.y*4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const float fxm = ((const float*)(mx + (size_t)dy*mxstep))[dx], fym = ((const float*)(my + (size_t)dy*mystep))[dx];
    cubic_px(src, sw, sh, sstep, dst + (size_t)dy*dstep + dx*3, cv_round_f(__fmul_rn(fxm, 32.f)), cv_round_f(__fmul_rn(fym, 32.f)),
             tab, border, c0, c1, c2);
}

// K5s: NonlinearStretch's bicubic remap with separable maps (map x depends on the column, map y on the row), batched.
// A thread owns one output column of a 256-column x SV_ROWS-row strip and walks down it.  Because the source row of an
// output row is the same for every column, consecutive output rows share source rows *uniformly*: the thread keeps the
// four source rows of its 4x4 window in registers -- each as six (tap, tap) halfword pairs ready for v_dot2_i32_i16 --
// and, for groups of SV_G output rows that read consecutive source rows, fetches only the rows that are new (one 12-byte
// load per output pixel when the vertical scale is 1, instead of four; the group's loads are in flight together) and no
// weight-table traffic: the 32 KB Q15 table sits in LDS.  Four neighbouring lanes pack their 12 output bytes into three
// dword stores.  Rows that do not step by one (other vertical scales) and pixels whose window touches the frame's edge
// take cubic_px.
constexpr int SV_ROWS = 48;
constexpr int SV_G = 4;                   // output rows per group of the fast path
typedef short short2_t __attribute__((ext_vector_type(2)));

struct SvRow { uint32_t p[6]; };          // (B0,B1) (B2,B3) (G0,G1) (G2,G3) (R0,R1) (R2,R3) as halfword pairs
struct __attribute__((packed, aligned(1))) SvRaw { uint32_t a, b, c; };     // 12 bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3

__device__ __forceinline__ SvRow sv_unpack(const SvRaw d)
{
    SvRow r;
    r.p[0] = __builtin_amdgcn_perm(d.a, d.a, 0x0c030c00u);      // B0, B1 = a.b0, a.b3
    r.p[1] = __builtin_amdgcn_perm(d.c, d.b, 0x0c050c02u);      // B2, B3 = b.b2, c.b1
    r.p[2] = __builtin_amdgcn_perm(d.b, d.a, 0x0c040c01u);      // G0, G1 = a.b1, b.b0
    r.p[3] = __builtin_amdgcn_perm(d.c, d.b, 0x0c060c03u);      // G2, G3 = b.b3, c.b2
    r.p[4] = __builtin_amdgcn_perm(d.b, d.a, 0x0c050c02u);      // R0, R1 = a.b2, b.b1
    r.p[5] = __builtin_amdgcn_perm(d.c, d.c, 0x0c030c00u);      // R2, R3 = c.b0, c.b3
    return r;
}

__device__ __forceinline__ int sv_dot(uint32_t a, uint32_t w, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, w), acc, false);
}

// One pixel from the window rows U0..U3 (source rows sy..sy+3) and the table entry's 16 halfwords (w0, w1):
// returns B | G << 8 | R << 16.
// Buffer addressing for the superview fast loop: the frame is one buffer resource (scalar), a row is the scalar offset
// row * step and the lane supplies a 32-bit byte offset, so a load or store needs no per-lane 64-bit address arithmetic (with
// plain pointers the compiler folds the lane offset into a per-lane 64-bit base and pays a v_mad_u64_u32 or v_lshl_add_u64 per
// access).  Reads past the resource's size return 0 and such writes are dropped, but nothing here relies on that.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ uint32_t sv_pixel(const SvRow& U0, const SvRow& U1, const SvRow& U2, const SvRow& U3, const uint4 w0, const uint4 w1)
{
    int sb = 1 << 14, sg = 1 << 14, sr = 1 << 14;
#define SV_ROW(U, WA, WB)                                                                   \
    sb = sv_dot(U.p[0], WA, sb); sb = sv_dot(U.p[1], WB, sb);                               \
    sg = sv_dot(U.p[2], WA, sg); sg = sv_dot(U.p[3], WB, sg);                               \
    sr = sv_dot(U.p[4], WA, sr); sr = sv_dot(U.p[5], WB, sr);
    SV_ROW(U0, w0.x, w0.y) SV_ROW(U1, w0.z, w0.w) SV_ROW(U2, w1.x, w1.y) SV_ROW(U3, w1.z, w1.w)
#undef SV_ROW
    uint32_t B = (uint32_t)min(max(sb >> 15, 0), 255), G = (uint32_t)min(max(sg >> 15, 0), 255), R = (uint32_t)min(max(sr >> 15, 0), 255);
    // The empty asm keeps the compiler from fusing two clamps and the byte packing into v_ashr_pk_u8_i32 (gfx950): the
    // code it generated around that instruction relies on the destination's upper half being zero, and pixels came out
    // wrong on MI355X (tests/test_gpu_parity.py::test_superview_separable_maps_batched).
    asm("" : "+v"(G));
    return B | (G << 8) | (R << 16);
}

__global__ void __launch_bounds__(BLOCK) k_superview_cubic(const uint8_t* __restrict__ src, int sw, int sh, unsigned sstep, size_t sframe,
                                                            uint8_t* __restrict__ dst, int dw, int dh, size_t dstep, size_t dframe,
                                                            const float* __restrict__ mx, const float* __restrict__ my,
                                                            const int16_t* __restrict__ tab, int border, int c0, int c1, int c2,
                                                            unsigned strips_x, unsigned strips_y, unsigned n_work, int dst_vec_ok, int src_vec_ok)
{
    static_assert(SV_ROWS <= 64, "one lane per strip row holds that row's map value");
    __shared__ uint4 s_tab[2048];                                   // 1024 entries x 16 halfwords
    for (int i = threadIdx.x; i < 2048; i += BLOCK) s_tab[i] = ((const uint4*)tab)[i];
    __syncthreads();
    const unsigned per_frame = strips_x*strips_y;
    const int lane = threadIdx.x & 63;
    const bool small_frames = (uint64_t)sstep*(unsigned)sh < (1ull << 31) && (uint64_t)dstep*(unsigned)dh < (1ull << 31);      // 32-bit buffer offsets
    const bool transparent = border == GF_BORDER_TRANSPARENT;
    const int border1 = transparent ? 4 : border;                      // taps of a written pixel reflect (as cubic_px)
    const uint32_t cpix = (uint32_t)(c0 & 255) | ((uint32_t)(c1 & 255) << 8) | ((uint32_t)(c2 & 255) << 16);
    SvRaw cpat;                                                         // four pixels of the constant colour
    cpat.a = cpix | (cpix << 24); cpat.b = (cpix >> 8) | (cpix << 16); cpat.c = (cpix >> 16) | (cpix << 8);
#pragma unroll 1
    for (unsigned v = blockIdx.x; v < n_work; v += gridDim.x) {
#ifdef GF_SV_STRIP_SYNC
        // Measurement switch, off: a barrier per strip keeps the four waves on the same strip (no data is exchanged: this is about
        // time).  Their row windows share 128-byte lines, and without it the waves drift apart by whole strips over the persistent
        // loop, so the shared lines have left L1 and L2 when the sibling wave asks: L2 misses to memory 38.2 M -> 26.6 M per 128
        // C4 frames with the barrier (2.05x -> 1.42x the source bytes) -- but 2.75 -> 2.86 ms, because the kernel is bound by VALU
        // issue and the waves now wait for each other.  Speed was chosen.
        __builtin_amdgcn_s_barrier();
#endif
        // Plain row-major strip order.  Measured (C4, 128 frames): the XCD-banded order of the fused kernels' tiles is 9 % slower
        // here (3.33 vs 3.06 ms) -- this kernel is bound by VALU issue, not by what its strips share in L2.
        const unsigned frame = v / per_frame, t = v - frame*per_frame;
        const unsigned syi = t / strips_x, sxi = t - syi*strips_x;
        const uint8_t* S = src + (size_t)frame*sframe;
        uint8_t* Dst = dst + (size_t)frame*dframe;
        const int x = (int)(sxi*BLOCK + threadIdx.x);
        const bool col_ok = x < dw;
        const int isx = cv_round_f(__fmul_rn(mx[col_ok ? x : dw - 1], 32.f));
        const int sx = sat_s16(isx >> INTER_BITS) - 1, fx = isx & 31;
        // A lane whose four tap columns lie inside the row fetches them as one 12-byte load; an edge lane resolves each
        // tap column through the border rule once per strip (xs3 = byte offset in the row, -1 = the constant colour).
        const bool direct = sx >= 0 && sx + 3 <= sw - 1;
        const unsigned off_direct = direct ? (unsigned)sx*3u : 0u;
        // When every lane of the wave is direct and the wave's windows span at most 64 aligned dwords of the row, the wave
        // fetches those dwords once (one aligned dword per lane) and each lane collects its 12 bytes from its neighbours'
        // registers (ds_bpermute): a quarter of the bytes through the vector-memory return path.
        int sx_lo = direct ? sx : INT_MAX, sx_hi = direct ? sx : INT_MIN;
#pragma unroll
        for (int o = 32; o; o >>= 1) { sx_lo = min(sx_lo, __shfl_xor(sx_lo, o)); sx_hi = max(sx_hi, __shfl_xor(sx_hi, o)); }
        const unsigned coop_base = (unsigned)__builtin_amdgcn_readfirstlane(max(sx_lo, 0)*3) & ~3u;
        const unsigned coop_end = (unsigned)__builtin_amdgcn_readfirstlane(max(sx_hi, 0)*3) + 12u;     // one past the last byte needed
        const bool coop = src_vec_ok && __all(direct) && coop_end - coop_base <= 256u && coop_end <= (((unsigned)sw*3u) & ~3u);
        const unsigned coop_ld = coop_base + 4u*(unsigned)lane;        // this lane's dword of the row
        const unsigned coop_bo = off_direct - coop_base;                // this lane's window within the fetched dwords
        const int coop_sel = (int)(coop_bo & ~3u);                      // byte address of its first dword for ds_bpermute
        const unsigned coop_sh = (coop_bo & 3u)*8u;
        int xs3[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int t = border_interp(sx + q, sw, border1); xs3[q] = t < 0 ? -1 : t*3; }
        const bool lane_wr = col_ok && !(transparent && (unsigned)(sx + 1) >= (unsigned)sw);
        // lanes 4q..4q+3 write 12 contiguous bytes: when all four write, three of them store one dword each
        const unsigned long long wr_mask = __ballot(lane_wr);
        const bool quad_full = dst_vec_ok && ((wr_mask >> (lane & ~3)) & 0xFull) == 0xFull;
        const int y0 = (int)(syi*SV_ROWS), nrows = min(SV_ROWS, dh - y0);
        const int isy_lane = cv_round_f(__fmul_rn(my[min(y0 + lane, dh - 1)], 32.f));      // lane i: the strip's row i

        auto load_row = [&](int sy) -> SvRaw {                          // sy is wave-uniform
            const int ry = border_interp(sy, sh, border1);
            SvRaw r = cpat;
            if (ry >= 0) {
                const uint8_t* rowp = S + (size_t)ry*sstep;
                if (direct) r = *(const SvRaw*)(rowp + off_direct);
                else if (col_ok) {
                    uint32_t t[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        t[q] = cpix;
                        if (xs3[q] >= 0) { const uint8_t* e = rowp + xs3[q]; t[q] = (uint32_t)e[0] | ((uint32_t)e[1] << 8) | ((uint32_t)e[2] << 16); }
                    }
                    r.a = t[0] | (t[1] << 24); r.b = (t[1] >> 8) | (t[2] << 16); r.c = (t[2] >> 16) | (t[3] << 8);
                }
            }
            return r;
        };
        auto store_px = [&](int y, uint32_t px, int sy) {
            // right neighbour's pixel within the quad (quad_perm [1,2,3,3]); executed by every lane of the wave
            const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, 0xF9, 0xF, 0xF, true);
            if (transparent && (unsigned)(sy + 1) >= (unsigned)sh) return;          // uniform: the whole row stays untouched
            uint8_t* D = Dst + (size_t)y*dstep + (size_t)x*3;
            if (quad_full) {                                                // the quad's 12 bytes start 4-byte aligned
                const int j = lane & 3;
                if (j < 3) *(uint32_t*)(D + j) = __builtin_amdgcn_alignbit(nb, px << 8, 8u + 8u*(unsigned)j);
            } else if (lane_wr) {
                D[0] = (uint8_t)px; D[1] = (uint8_t)(px >> 8); D[2] = (uint8_t)(px >> 16);
            }
        };

        SvRow R[SV_G + 3];                                              // source rows base .. base + SV_G + 2 of the current group
#pragma unroll
        for (int k = 0; k < SV_G + 3; k++)
#pragma unroll
            for (int j = 0; j < 6; j++) R[k].p[j] = 0;
        // r < 64; uniform, and said so: everything derived from it (row addresses) then stays in scalar registers
        auto row_of = [&](int r) { return __builtin_amdgcn_readfirstlane(sat_s16(__builtin_amdgcn_readlane(isy_lane, r) >> INTER_BITS) - 1); };
        auto group_runs = [&](int r, int sy) {                          // output rows r .. r + SV_G - 1 read source rows sy, sy + 1, ...
            bool ok = true;
#pragma unroll
            for (int k = 1; k < SV_G; k++) ok = ok && row_of(r + k) == sy + k;
            return ok;
        };
        int i = 0;

        // Fast loop: a wave whose lanes all fetch cooperatively and all write.  Straight-line code, no branch around a
        // memory instruction, so that the rows the next group adds are requested *before* this group's stores are queued
        // and waiting for them does not wait for the stores (vector loads and stores complete in issue order).
        if (coop && !transparent && dst_vec_ok && wr_mask == ~0ull && small_frames) {
            const unsigned ld_off = min(coop_ld, (coop_end - 1u) & ~3u);            // lanes past the span re-read its last dword
            const int qj = lane & 3;
            // lanes 0..2 of a quad write dword qj of its 12 bytes; lane 3 repeats lane 2's store (same address, same value)
            const unsigned dl_off = (unsigned)x*3u + (qj < 3 ? (unsigned)qj : ~0u);      // x >= 3 for a quad's lane 3
            const __amdgpu_buffer_rsrc_t srs = frame_rsrc(S, sstep*(unsigned)sh), drs = frame_rsrc(Dst, (unsigned)dstep*(unsigned)dh);
            const unsigned st_sh = qj < 3 ? 8u + 8u*(unsigned)qj : 24u;
            // one vertical fraction for the whole strip (what a unit vertical scale gives): the 16 weights of a lane are then
            // the same for every row and are read from the table once, not per pixel
            const int fy0 = __builtin_amdgcn_readlane(isy_lane, 0) & 31;
            const bool fy_const = __all(lane >= nrows || (isy_lane & 31) == fy0);
            const uint4 cw0 = s_tab[(fy0*32 + fx)*2], cw1 = s_tab[(fy0*32 + fx)*2 + 1];
            auto rows_ok = [&](int sy) { return border1 != GF_BORDER_CONSTANT || (sy >= 0 && sy + SV_G + 2 <= sh - 1); };
            // scalar row base + 32-bit lane offset: no per-lane 64-bit address arithmetic
            auto loadw = [&](int sy) {
                const unsigned ry = (unsigned)__builtin_amdgcn_readfirstlane(max(border_interp(sy, sh, border1), 0));
                return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(srs, (int)ld_off, (int)(ry*sstep), 0);
            };
            auto gather = [&](uint32_t w) {
                const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(coop_sel, (int)w);
                const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(coop_sel + 4, (int)w);
                const uint32_t w2 = (uint32_t)__builtin_amdgcn_ds_bpermute(coop_sel + 8, (int)w);
                const uint32_t w3 = (uint32_t)__builtin_amdgcn_ds_bpermute(coop_sel + 12, (int)w);
                SvRaw r;
                r.a = __builtin_amdgcn_alignbit(w1, w0, coop_sh); r.b = __builtin_amdgcn_alignbit(w2, w1, coop_sh); r.c = __builtin_amdgcn_alignbit(w3, w2, coop_sh);
                return sv_unpack(r);
            };
            auto fast_rows = [&](auto one_fy) {                           // one_fy: std::true_type when fy_const
#pragma unroll 1
            while (i + SV_G <= nrows) {
                int sy0 = row_of(i);
                if (!(group_runs(i, sy0) && rows_ok(sy0))) break;
                uint32_t w[3], nw[SV_G];
#pragma unroll
                for (int k = 0; k < 3; k++) w[k] = loadw(sy0 + k);
#pragma unroll
                for (int k = 0; k < SV_G; k++) nw[k] = loadw(sy0 + 3 + k);
#pragma unroll
                for (int k = 0; k < 3; k++) R[k] = gather(w[k]);
#pragma unroll 1
                for (;;) {
#pragma unroll
                    for (int k = 0; k < SV_G; k++) R[3 + k] = gather(nw[k]);
                    const int i2 = i + SV_G;
                    const int sy1 = row_of(min(i2, 63 - SV_G));
                    const bool more = i2 + SV_G <= nrows && sy1 == sy0 + SV_G && group_runs(min(i2, 63 - SV_G), sy1) && rows_ok(sy1);
#pragma unroll
                    for (int k = 0; k < SV_G; k++) nw[k] = loadw(sy0 + SV_G + 3 + k);      // used only when `more`
#pragma unroll
                    for (int k = 0; k < SV_G; k++) {
                        uint4 wa = cw0, wb = cw1;
                        if constexpr (!decltype(one_fy)::value) {
                            const int e = ((__builtin_amdgcn_readlane(isy_lane, i + k) & 31)*32 + fx)*2;
                            wa = s_tab[e]; wb = s_tab[e + 1];
                        }
                        const uint32_t px = sv_pixel(R[k], R[k + 1], R[k + 2], R[k + 3], wa, wb);
                        const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, 0xF9, 0xF, 0xF, true);      // quad_perm [1,2,3,3]: right neighbour
                        const uint32_t lf = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, 0x90, 0xF, 0xF, true);      // quad_perm [0,0,1,2]: left neighbour
                        __builtin_amdgcn_raw_buffer_store_b32((int)__builtin_amdgcn_alignbit(qj < 3 ? nb : px, (qj < 3 ? px : lf) << 8, st_sh), drs, (int)dl_off,
                                                              (int)((unsigned)__builtin_amdgcn_readfirstlane(y0 + i + k)*(unsigned)dstep), 0);
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) R[k] = R[SV_G + k];
                    i = i2; sy0 = sy1;
                    if (!more) break;
                }
            }
            };
            if (fy_const) fast_rows(std::true_type{}); else fast_rows(std::false_type{});
        }

        // Generic rows: edge waves, the transparent border, vertical steps other than 1, and what the fast loop left.
        int carried = INT_MIN;                                          // R[0..2] hold source rows carried .. carried + 2
#pragma unroll 1
        while (i < nrows) {
            const int sy0 = row_of(i);
            const bool run = i + SV_G <= nrows && group_runs(min(i, 63 - SV_G), sy0);
            const bool fresh = carried != sy0;                          // uniform: the three carried rows are not the ones needed
            if (run) {
                SvRaw raw[SV_G + 3];
                if (fresh) {
#pragma unroll
                    for (int k = 0; k < 3; k++) raw[k] = load_row(sy0 + k);
                }
#pragma unroll
                for (int k = 3; k < SV_G + 3; k++) raw[k] = load_row(sy0 + k);
                if (fresh) {
#pragma unroll
                    for (int k = 0; k < 3; k++) R[k] = sv_unpack(raw[k]);
                }
#pragma unroll
                for (int k = 3; k < SV_G + 3; k++) R[k] = sv_unpack(raw[k]);
#pragma unroll
                for (int k = 0; k < SV_G; k++) {
                    const int e = ((__builtin_amdgcn_readlane(isy_lane, i + k) & 31)*32 + fx)*2;
                    store_px(y0 + i + k, sv_pixel(R[k], R[k + 1], R[k + 2], R[k + 3], s_tab[e], s_tab[e + 1]), sy0 + k);
                }
#pragma unroll
                for (int k = 0; k < 3; k++) R[k] = R[SV_G + k];
                carried = sy0 + SV_G;
                i += SV_G;
            } else {                                                    // one row
                SvRaw raw[4];
                if (fresh) {
#pragma unroll
                    for (int k = 0; k < 3; k++) raw[k] = load_row(sy0 + k);
                }
                raw[3] = load_row(sy0 + 3);
                if (fresh) {
#pragma unroll
                    for (int k = 0; k < 3; k++) R[k] = sv_unpack(raw[k]);
                }
                R[3] = sv_unpack(raw[3]);
                const int e = ((__builtin_amdgcn_readlane(isy_lane, i) & 31)*32 + fx)*2;
                store_px(y0 + i, sv_pixel(R[0], R[1], R[2], R[3], s_tab[e], s_tab[e + 1]), sy0);
#pragma unroll
                for (int k = 0; k < 3; k++) R[k] = R[k + 1];
                carried = sy0 + 1;
                i += 1;
            }
        }
    }
}

// K1s: fused remap + superview stretch (BASELINE config C4): gf_undistort_rotate_c3 followed by gf_superview_cubic_u8c3 without
// the intermediate frame ever reaching HBM.  The reference rounds to uint8 between the two resampling steps
// (stabilizer.py:1442 then nonlinear_stretch.py:130), so the intermediate image must exist -- here one tile of it at a time,
// in LDS.
// Work item = (frame, 64 x 64 tile of the INTERMEDIATE image, i.e. one tile of the RP grid).  The item owns the output
// pixels whose bicubic window starts in that tile: output columns [tcol[tx], tcol[tx+1]) (the horizontal map is monotone;
// host-built table) and, because the vertical map is a unit-step shift (sy = y + c0row, one fraction fy for all rows --
// checked on the host), output rows [64 ty - c0row, 64 ty + 64 - c0row).  Their 4 x 4 windows reach one pixel to the left /
// above the tile and two to the right / below, so the item computes the intermediate region
//   rows 64 ty - 1 .. 64 ty + 65,  columns 64 tx - 4 .. 64 tx + 67   (67 rows of 18 four-pixel groups)
// with exactly the code and the per-tile polynomials of k_undistort_rotate_rp -- the fringe pixels with the polynomials of
// the eight neighbouring tiles they belong to, so that every intermediate pixel has the bits the stand-alone kernel gives
// it -- packs it into LDS (pixels outside the intermediate image are 0, which is what BORDER_CONSTANT 0 of the stretch
// step reads there) and then walks down its output columns as k_superview_cubic does, reading the window rows from LDS.
// Steps per item, one workgroup barrier each: node values of the (up to) nine tiles -> row coefficients of the region's
// rows -> the region's pixels -> the output pixels.
constexpr int FS_QUADS = RT/PXT + 2;              // four-pixel groups per region row: 1 + 16 + 1
constexpr int FS_ROWS = RT + 3;                   // region rows: 1 + 64 + 2
constexpr int FS_PITCH = FS_QUADS*PXT*3;          // bytes per region row in LDS (216)
struct FsShared {
    RpConst K;                                    // 4.6 KB
    double U[BLOCK/64][RP_N][RP_N];               // rp_nodes scratch, one per wave
    double C2[9][RP_N][RP_N];                     // node-row coefficients of the 3 x 3 tiles around (tx, ty)
    double C[3][FS_ROWS][RP_N];                   // [tile column part][region row]: row coefficients
    int ok[12];                                   // per tile of the 3 x 3: 1 = polynomials, 0 = per-pixel path, -1 = no such tile
    uint32_t I[FS_ROWS][FS_PITCH/4];              // the intermediate region, packed BGR bytes (14.5 KB)
};
static_assert(sizeof(FsShared) <= 40*1024, "four workgroups per CU");

struct FsOut {                                    // the final (stretched) frames
    uint8_t* dst; int w, h; size_t step, frame; int vec_ok;
    const int* colq;                              // [w]  cvRound(32 * mapx_row[x])
    const int* tcol;                              // [tiles_x + 1] first output column of every intermediate tile column
    int c0row;                                    // sy = y + c0row  (integer part of the vertical map, minus nothing: window rows are sy-1 .. sy+2)
    int fy;                                       // the vertical fraction (same for every row)
};

template <int BORDER>
__global__ void __launch_bounds__(BLOCK, 4) k_undistort_rotate_superview(RemapArgs A, FsOut F, const Xf* __restrict__ xfs, const double* __restrict__ lens, LensK lk,
                                                                        const double* __restrict__ rp_tab, const int16_t* __restrict__ ctab,
                                                                        unsigned tiles_pad, unsigned n_work)
{
    constexpr int DEPTH = GF_DEPTH_U8;
    __shared__ FsShared S;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    rp_load_consts(rp_tab, S.K, tid, BLOCK);
    const double offx = rp_node_off(rp_tab, lane & 7), offy = rp_node_off(rp_tab, lane >> 3);
    const double2* tab = (const double2*)lens;                  // lens table read from global memory (L2): nine lookups per item
    __syncthreads();
#pragma unroll 1
    for (unsigned v = blockIdx.x; v < n_work; v += gridDim.x) {
        const unsigned frame = v / tiles_pad;
        int tx, ty;
        if (!tile_of_padded_id(v - frame*tiles_pad, A.tiles_x, A.tiles_y, tx, ty)) continue;          // uniform
        const Xf& xf = xfs[frame];
        const SrcImg s{A.src + (size_t)frame*A.sframe, A.sw, A.sh, A.sstep};

#ifndef GF_FS_ABL
#define GF_FS_ABL 0          /* tools/ ablation builds: bit 0 = centre tile's polynomials only, bit 1 = no region pixels, bit 2 = no output pixels */
#endif
        // ---- 1. node values -> node-row coefficients of the 3 x 3 tiles (wave w: tiles w, w + 4, w + 8)
#pragma unroll 1
        for (int t9 = wave; t9 < 9; t9 += BLOCK/64) {
            if ((GF_FS_ABL & 1) && t9 != 4) { if (lane == 0) S.ok[t9] = -1; continue; }
            const int ntx = tx + t9 % 3 - 1, nty = ty + t9/3 - 1;
            if (ntx < 0 || ntx >= A.tiles_x || nty < 0 || nty >= A.tiles_y) { if (lane == 0) S.ok[t9] = -1; continue; }
            rp_nodes<0>(xf, lk, tab, offx, offy, lane & 7, lane >> 3, ntx, nty, S.U[wave], S.K, S.C2[t9], &S.ok[t9], nullptr);
            asm volatile("" ::: "memory");
        }
        lds_barrier();

        // ---- 2. row coefficients of the region's rows, for the three tile columns
#pragma unroll 1
        for (int o = tid; o < 3*FS_ROWS*RP_N; o += BLOCK) {
            const int a = o & (RP_N - 1), rr = (o >> 3) % FS_ROWS, part_x = (o >> 3)/FS_ROWS;
            const int part_y = rr == 0 ? 0 : rr <= RT ? 1 : 2;
            const int lr = rr == 0 ? RT - 1 : rr <= RT ? rr - 1 : rr - RT - 1;        // row of that tile
            const int t9 = part_y*3 + part_x;
            if (S.ok[t9] == 1) {
                double c2[RP_N];
#pragma unroll
                for (int m = 0; m < RP_N; m++) c2[m] = S.C2[t9][m][a];
                S.C[part_x][rr][a] = rp_row_coef(c2, S.K.Ly[lr]);
            }
        }
        lds_barrier();

        // ---- 3. the intermediate region: the pixels of k_undistort_rotate_rp, packed into LDS
        {
            const RpFrameC fc = rp_frame_consts(xf);
#pragma unroll 1
            for (int p = tid; p < FS_QUADS*FS_ROWS; p += BLOCK) {
                const int rr = p/FS_QUADS, q = p - rr*FS_QUADS;
                const int part_x = q == 0 ? 0 : q <= RT/PXT ? 1 : 2, part_y = rr == 0 ? 0 : rr <= RT ? 1 : 2;
                const int okt = S.ok[part_y*3 + part_x];
                const int x0 = tx*RT - PXT + q*PXT, y = ty*RT - 1 + rr;
                uint32_t w0 = 0, w1 = 0, w2 = 0;
                if (!(GF_FS_ABL & 2) && okt >= 0 && y >= 0 && y < A.dh && x0 + PXT > 0 && x0 < A.dw) {
                    const int lxq = (q - 1) & (RT_TPR - 1);              // the group's place in its own tile
                    double z[PXT];
#pragma unroll
                    for (int e = 0; e < PXT; e++) z[e] = ((double)(lxq*PXT + e) - (RP_HALF - 0.5))*(1.0/RP_HALF);
                    int iu[PXT], iv[PXT], px[PXT][3];
                    if (okt == 1) rp_coords(rp_row_consts(fc, y), S.C[part_x][rr], z, x0, ROUND_MAGIC, ROUND_MAGIC, iu, iv);
                    else coord_q5x4(row_consts(xf, y), xf, lk, lens_hi_limit(xf, lk), tab, x0, iu, iv);
                    (void)remap4<DEPTH, BORDER, true>(s, iu, iv, A.cval, px);
                    if (x0 < 0 || x0 + PXT > A.dw) {                     // columns outside the intermediate image read as 0
#pragma unroll
                        for (int e = 0; e < PXT; e++)
                            if (x0 + e < 0 || x0 + e >= A.dw) px[e][0] = px[e][1] = px[e][2] = 0;
                    }
                    w0 = pack4_hi16(px[0][0], px[0][1], px[0][2], px[1][0]);
                    w1 = pack4_hi16(px[1][1], px[1][2], px[2][0], px[2][1]);
                    w2 = pack4_hi16(px[2][2], px[3][0], px[3][1], px[3][2]);
                }
                uint32_t* d = &S.I[rr][q*3];
                d[0] = w0; d[1] = w1; d[2] = w2;
            }
        }
        lds_barrier();

        // ---- 4. the output pixels this item owns, column by column
        {
            const int Xa = F.tcol[tx], Xb = F.tcol[tx + 1];
            const int Ya = ty == 0 ? 0 : min(max(ty*RT - F.c0row, 0), F.h);
            const int Yb = ty == A.tiles_y - 1 ? F.h : min(max(ty*RT + RT - F.c0row, 0), F.h);
            const int xorg = tx*RT - PXT, yorg = ty*RT - 1;             // region origin in the intermediate image
#pragma unroll 1
            for (int cb = Xa & ~3; cb < Xb && Ya < Yb && !(GF_FS_ABL & 4); cb += BLOCK) {
                const int wc4 = min(BLOCK, (Xb - cb + 3) & ~3);          // columns of this chunk, a multiple of 4
                const int nseg = BLOCK/wc4;
                const int seglen = (Yb - Ya + nseg - 1)/nseg;
                const int seg = tid/wc4, col = tid - seg*wc4;
                const int X = cb + col;
                const bool col_ok = seg < nseg && X >= Xa && X < Xb;
                const int y0 = Ya + seg*seglen, y1 = col_ok ? min(Yb, y0 + seglen) : y0;
                const int isx = F.colq[min(max(X, 0), F.w - 1)];
                const int sx = sat_s16(isx >> INTER_BITS) - 1, fx = isx & 31;
                const int cidx = sx - xorg;
                const bool in_x = cidx >= 0 && cidx + 3 < FS_QUADS*PXT;   // otherwise the window lies outside the image: 0
                const unsigned bo = in_x ? (unsigned)cidx*3u : 0u;
                const int e = (F.fy*32 + fx)*2;
                const uint4 wq0 = ((const uint4*)ctab)[e], wq1 = ((const uint4*)ctab)[e + 1];
                // lanes 4q..4q+3 hold four consecutive columns of one row (cb and wc4 are multiples of 4): 12 contiguous bytes
                const bool quad_full = F.vec_ok && ((__ballot(col_ok) >> (lane & ~3)) & 0xFull) == 0xFull;
                auto load_row = [&](int r) -> SvRow {                   // region row r (may be outside: zeros)
                    SvRaw raw; raw.a = raw.b = raw.c = 0;
                    if (in_x && r >= 0 && r < FS_ROWS) {
                        const uint32_t* pr = &S.I[r][bo >> 2];
                        const uint32_t d0 = pr[0], d1 = pr[1], d2 = pr[2], d3 = (bo >> 2) + 3 < FS_PITCH/4 ? pr[3] : 0u;
                        raw.a = __builtin_amdgcn_alignbyte(d1, d0, bo); raw.b = __builtin_amdgcn_alignbyte(d2, d1, bo); raw.c = __builtin_amdgcn_alignbyte(d3, d2, bo);
                    }
                    return sv_unpack(raw);
                };
                auto store = [&](int y, uint32_t px) {
                    const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, 0xF9, 0xF, 0xF, true);     // quad_perm [1,2,3,3]
                    uint8_t* D = F.dst + (size_t)frame*F.frame + (size_t)y*F.step + (size_t)X*3;
                    if (quad_full) {
                        const int j = lane & 3;
                        if (j < 3) *(uint32_t*)(D + j) = __builtin_amdgcn_alignbit(nb, px << 8, 8u + 8u*(unsigned)j);
                    } else if (col_ok) {
                        D[0] = (uint8_t)px; D[1] = (uint8_t)(px >> 8); D[2] = (uint8_t)(px >> 16);
                    }
                };
                if (y0 < y1) {
                    int r = y0 + F.c0row - 1 - yorg;                      // region row of the first window row of output row y0
                    SvRow R[SV_G + 3];
                    R[0] = load_row(r); R[1] = load_row(r + 1); R[2] = load_row(r + 2);
                    int y = y0;
#pragma unroll 1
                    for (; y + SV_G <= y1; y += SV_G, r += SV_G) {
#pragma unroll
                        for (int k = 0; k < SV_G; k++) R[3 + k] = load_row(r + 3 + k);
#pragma unroll
                        for (int k = 0; k < SV_G; k++) store(y + k, in_x ? sv_pixel(R[k], R[k + 1], R[k + 2], R[k + 3], wq0, wq1) : 0u);
#pragma unroll
                        for (int k = 0; k < 3; k++) R[k] = R[SV_G + k];
                    }
#pragma unroll 1
                    for (; y < y1; y++, r++) {
                        R[3] = load_row(r + 3);
                        store(y, in_x ? sv_pixel(R[0], R[1], R[2], R[3], wq0, wq1) : 0u);
                        R[0] = R[1]; R[1] = R[2]; R[2] = R[3];
                    }
                }
            }
        }
        lds_barrier();          // the region and the coefficient buffers are reused by the next item
    }
}

// cv2.rotate (90 degrees clockwise, 180, 90 counter-clockwise) of a batch of frames: the unfused form of what
// k_undistort_rotate_rp<.., true> does in its tap addressing, for the kernels that have no rotated form.
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_rotate_c3(const uint8_t* __restrict__ src, unsigned sstep, size_t sframe, uint8_t* __restrict__ dst, int dw, int dh,
                                                     size_t dstep, size_t dframe, RotMap m)
{
    const int x = blockIdx.x*64 + (threadIdx.x & 63), y = blockIdx.y*4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const SrcImg s{src + (size_t)blockIdx.z*sframe, dw, dh, sstep};
    const T* q = (const T*)(s.base + rot_pixel_offset(s, m, x, y, (int)sizeof(T)));
    T* d = (T*)(dst + (size_t)blockIdx.z*dframe + (size_t)y*dstep) + x*3;
    d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
}

// K6: hyperlapse blend passes (unfused form).
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_blend_acc(const T* __restrict__ f, double* acc, double wgt, int reset, size_t n)
{
    for (size_t i = (size_t)blockIdx.x*BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x*BLOCK) {
        double a = reset ? __dmul_rn(acc[i], 0.0) : acc[i];          // frame_temp * 0.0 (keeps NaN/inf semantics)
        acc[i] = __dadd_rn(a, __dmul_rn(wgt, (double)f[i]));
    }
}
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_blend_fin(const double* __restrict__ acc, T* out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x*BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x*BLOCK)
        out[i] = (T)(long long)acc[i];
}

// K7: cv::fisheye::undistortPoints for n_sets rotations of one point list (AdaptiveZoom's boundary polygon,
// adaptive_zoom.py:193-228): Newton iteration on theta (<= 10 steps, |fix| < 1e-8), scale = tan(theta)/theta_d,
// reprojection through RR = P*R.  One thread per (set, point), all float64.
struct PtsK { double f[2], c[2], k[4]; };
__global__ void __launch_bounds__(BLOCK) k_undistort_points(const double* __restrict__ pts, unsigned n_pts, PtsK P,
                                                            const double* __restrict__ RR, size_t n_total, double* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x*BLOCK + threadIdx.x;
    if (i >= n_total) return;
    const size_t set = i / n_pts;
    const unsigned pi = (unsigned)(i - set*n_pts);
    const double pwx = __ddiv_rn(pts[2*pi] - P.c[0], P.f[0]), pwy = __ddiv_rn(pts[2*pi + 1] - P.c[1], P.f[1]);
    const double PI_2 = 1.57079632679489661923;
    double theta_d = __dsqrt_rn(__dadd_rn(__dmul_rn(pwx, pwx), __dmul_rn(pwy, pwy)));
    theta_d = fmin(fmax(-PI_2, theta_d), PI_2);
    bool converged = false;
    double theta = theta_d, scale = 1.0;
    if (theta_d > 1e-8) {
        for (int j = 0; j < 10; j++) {
            const double t2 = __dmul_rn(theta, theta), t4 = __dmul_rn(t2, t2), t6 = __dmul_rn(t4, t2), t8 = __dmul_rn(t6, t2);
            const double a = __dmul_rn(P.k[0], t2), b = __dmul_rn(P.k[1], t4), c = __dmul_rn(P.k[2], t6), d = __dmul_rn(P.k[3], t8);
            const double num = __dsub_rn(__dmul_rn(theta, __dadd_rn(__dadd_rn(__dadd_rn(__dadd_rn(1.0, a), b), c), d)), theta_d);
            const double den = __dadd_rn(__dadd_rn(__dadd_rn(__dadd_rn(1.0, __dmul_rn(3.0, a)), __dmul_rn(5.0, b)), __dmul_rn(7.0, c)), __dmul_rn(9.0, d));
            const double fix = __ddiv_rn(num, den);
            theta = __dsub_rn(theta, fix);
            if (fabs(fix) < 1e-8) { converged = true; break; }
        }
        scale = __ddiv_rn(tan(theta), theta_d);
    } else {
        converged = true;
    }
    const bool flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
    double ox = -1000000.0, oy = -1000000.0;
    if (converged && !flipped) {
        const double* R = RR + set*9;
        const double pux = __dmul_rn(pwx, scale), puy = __dmul_rn(pwy, scale);
        const double prx = __dadd_rn(__dadd_rn(__dmul_rn(R[0], pux), __dmul_rn(R[1], puy)), R[2]);
        const double pry = __dadd_rn(__dadd_rn(__dmul_rn(R[3], pux), __dmul_rn(R[4], puy)), R[5]);
        const double prz = __dadd_rn(__dadd_rn(__dmul_rn(R[6], pux), __dmul_rn(R[7], puy)), R[8]);
        ox = __ddiv_rn(prx, prz); oy = __ddiv_rn(pry, prz);
    }
    out[2*i] = ox; out[2*i + 1] = oy;
}

// ---------------------------------------------------------------------------------------------
// Host-side argument checking / launch helpers

int check_img(const char* what, const void* p, int w, int h, size_t step, int depth)
{
    const size_t esz = depth == GF_DEPTH_U8 ? 1 : 2;
    if (!p) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", what);
    if (w <= 0 || h <= 0 || w >= SHRT_MAX || h >= SHRT_MAX) return fail(GF_ERR_INVALID_ARG, "%s: bad dimensions (must be 1..32766), w=%lld", what, w);
    if (step < (size_t)w*3*esz) return fail(GF_ERR_INVALID_ARG, "%s: row step %lld smaller than a row", what, (long long)step);
    if (esz == 2 && ((step & 1) || ((uintptr_t)p & 1))) return fail(GF_ERR_INVALID_ARG, "%s: u16 image must be 2-byte aligned", what);
    if (step >= (1u << 24) || step*(size_t)h >= ((size_t)1 << 31))
        return fail(GF_ERR_UNSUPPORTED, "%s: frames of 2 GiB or more (or rows of 16 MiB or more) are not supported (step=%lld)", what, (long long)step);
    return GF_OK;
}

int check_common(int depth, int border)
{
    if (depth != GF_DEPTH_U8 && depth != GF_DEPTH_U16) return fail(GF_ERR_UNSUPPORTED, "depth %s%lld not supported (0=u8, 2=u16)", "", depth);
    if (border != GF_BORDER_CONSTANT && border != GF_BORDER_REPLICATE && border != GF_BORDER_TRANSPARENT)
        return fail(GF_ERR_UNSUPPORTED, "border mode %s%lld not supported (0, 1, 5)", "", border);
    return GF_OK;
}

void cval_from(const double* bv, int depth, int* cval)
{
    for (int k = 0; k < 3; k++) {
        double v = bv ? bv[k] : 0.0;
        long r = (v != v) ? 0 : lrint(v);                       // saturate_cast<T>(double): cvRound then clamp
        const long hi = depth == GF_DEPTH_U8 ? 255 : 65535;
        cval[k] = (int)(r < 0 ? 0 : r > hi ? hi : r);
    }
}

bool ranges_overlap(const void* a, size_t an, const void* b, size_t bn)
{
    const uintptr_t A = (uintptr_t)a, B = (uintptr_t)b;
    return A < B + bn && B < A + an;
}

RemapArgs make_args(int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                    void* dst, int dw, int dh, size_t dstep, size_t dframe, const double* bv)
{
    RemapArgs A;
    A.src = (const uint8_t*)src; A.sw = sw; A.sh = sh; A.sstep = (unsigned)sstep; A.sframe = sframe;
    A.dst = (uint8_t*)dst; A.dw = dw; A.dh = dh; A.dstep = dstep; A.dframe = dframe;
    A.tiles_x = (dw + TILE_W - 1)/TILE_W; A.tiles_y = (dh + TILE_H - 1)/TILE_H;
    cval_from(bv, depth, A.cval);
    const size_t al = depth == GF_DEPTH_U8 ? 4 : 8;
    A.dst_vec_ok = (((uintptr_t)dst % al) == 0 && (dstep % al) == 0 && (dframe % al) == 0) ? 1 : 0;
    A.stretch_on = 0; A.sxn = A.sxd = A.syn = A.syd = 1.f;
    return A;
}

#define DISPATCH_DEPTH_BORDER(KERNEL, depth, border, grid, stream, ...)                                         \
    do {                                                                                                         \
        if (depth == GF_DEPTH_U8) {                                                                              \
            if (border == GF_BORDER_CONSTANT)       hipLaunchKernelGGL((KERNEL<GF_DEPTH_U8, GF_BORDER_CONSTANT>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);    \
            else if (border == GF_BORDER_REPLICATE) hipLaunchKernelGGL((KERNEL<GF_DEPTH_U8, GF_BORDER_REPLICATE>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);   \
            else                                    hipLaunchKernelGGL((KERNEL<GF_DEPTH_U8, GF_BORDER_TRANSPARENT>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__); \
        } else {                                                                                                 \
            if (border == GF_BORDER_CONSTANT)       hipLaunchKernelGGL((KERNEL<GF_DEPTH_U16, GF_BORDER_CONSTANT>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);   \
            else if (border == GF_BORDER_REPLICATE) hipLaunchKernelGGL((KERNEL<GF_DEPTH_U16, GF_BORDER_REPLICATE>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);  \
            else                                    hipLaunchKernelGGL((KERNEL<GF_DEPTH_U16, GF_BORDER_TRANSPARENT>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);\
        }                                                                                                        \
    } while (0)

Xf to_xf(const gf_frame_xform* x)
{
    Xf r;
    static_assert(sizeof(Xf) == sizeof(gf_frame_xform), "layout");
    static_assert(LENS_COEF == 8 && LENS_DOUBLES % (2*BLOCK) == 0, "lens table layout");
    memcpy(&r, x, sizeof r);
    return r;
}

// Device lens table for distortion coefficients D: a hit in the context's 4-slot cache costs nothing; a miss builds the
// table on the host (~1 ms), uploads it on `stream` and waits for the upload (once per lens).
// LensLease holds the context's mutex from the lookup until the launch that uses the table has been queued and then records
// an event behind that launch, so that a later eviction of the slot (a fifth distinct D) waits for exactly the launches
// that may still read it -- not for the whole device, and never for a launch that has not been queued yet.
struct LensLease {
    Ctx* c;
    hipStream_t stream;
    int slot;
    std::unique_lock<std::mutex> lock;
    LensLease(Ctx* ctx, hipStream_t st) : c(ctx), stream(st), slot(-1), lock(ctx->mu) {}
    int mode() const { return c->k1_mode; }
    int acquire(const double* D, const double** dev, LensK* lk);
    bool stretch_used = false;
    int acquire_stretch(const float* mapx, const float* mapy, int w, int h, int mid_w, int mid_h, const int** colq, const int** tcol, int* c0row, int* fy);
    template <class Slot> void mark(Slot& s)
    {
        int e = -1;
        for (int i = 0; i < LENS_EVENTS; i++) if (s.ev_used[i] && s.ev_stream[i] == stream) { e = i; break; }
        if (e < 0) for (int i = 0; i < LENS_EVENTS; i++) if (!s.ev_used[i]) { e = i; break; }
        if (e < 0) { s.ev_overflow = true; return; }
        if (hipEventRecord(s.ev[e], stream) == hipSuccess) { s.ev_used[e] = true; s.ev_stream[e] = stream; }
        else s.ev_overflow = true;
    }
    ~LensLease()
    {
        if (slot >= 0) mark(c->lens[slot]);
        if (stretch_used) mark(c->stretch);
    }
};

int LensLease::acquire(const double* D, const double** dev, LensK* lk)
{
    if (!D) return fail(GF_ERR_INVALID_ARG, "%s: null distortion coefficients", "lens");
    for (int i = 0; i < 4; i++) {
        if (!(D[i] == D[i]) || fabs(D[i]) > 1e6) return fail(GF_ERR_INVALID_ARG, "lens: distortion coefficient %s%lld is not finite", "", i);
        lk->k[i] = D[i];
    }
    {   // |theta_d| <= theta_max (1 + |k1| theta_max^2 + ...) for theta <= atan(sqrt(255))
        const double tm = atan(sqrt((double)((1 << LENS_OCT) - 1))), t2 = tm*tm;
        lk->bound = tm*(1 + t2*(fabs(D[0]) + t2*(fabs(D[1]) + t2*(fabs(D[2]) + t2*fabs(D[3])))));
    }
    c->tick++;
    int victim = -1;
    for (int i = 0; i < LENS_SLOTS; i++) {               // every slot is looked at: a hit may sit behind an empty slot
        LensSlot& s = c->lens[i];
        if (s.valid && memcmp(s.D, D, sizeof s.D) == 0) { s.last_use = c->tick; *dev = s.dev; slot = i; return GF_OK; }
    }
    for (int i = 0; i < LENS_SLOTS; i++) if (!c->lens[i].valid) { victim = i; break; }
    if (victim < 0) {
        victim = 0;
        for (int i = 1; i < LENS_SLOTS; i++) if (c->lens[i].last_use < c->lens[victim].last_use) victim = i;
    }
    LensSlot& s = c->lens[victim];
    if (s.valid) {                       // launches queued earlier (the mutex is ours: none is in the making) may still read the table
        s.valid = false;
        if (s.ev_overflow) HIP_TRY(hipDeviceSynchronize());
        for (int i = 0; i < LENS_EVENTS; i++) if (s.ev_used[i]) HIP_TRY(hipEventSynchronize(s.ev[i]));
    }
    for (int i = 0; i < LENS_EVENTS; i++) s.ev_used[i] = false;
    s.ev_overflow = false;
    build_lens_table(D, s.host);
    HIP_TRY(hipMemcpyAsync(s.dev, s.host, LENS_DOUBLES*sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    memcpy(s.D, D, sizeof s.D);
    s.valid = true; s.last_use = c->tick;
    *dev = s.dev; slot = victim;
    return GF_OK;
}

// The fixed-point stretch tables for (mapx_row, mapy_col): GF_ERR_UNSUPPORTED unless the vertical map is a unit-step shift
// (row y reads source rows y + c .. y + c + 3 with one fraction for all rows: what NonlinearStretch produces when the
// output height equals the input height) and the horizontal map is monotone -- the fused kernel's work split rests on both.
int LensLease::acquire_stretch(const float* mapx, const float* mapy, int w, int h, int mid_w, int mid_h, const int** colq, const int** tcol, int* c0row, int* fy)
{
    (void)mid_h;
    StretchSlot& s = c->stretch;
    const int tiles_x = (mid_w + RT - 1)/RT;
    if (w + tiles_x + 1 > STRETCH_MAX_INTS) return fail(GF_ERR_UNSUPPORTED, "%s: stretch maps too large (%lld columns)", "gf_undistort_rotate_superview_u8c3", w);
    const bool hit = s.valid && s.w == w && s.h == h && s.mid_w == mid_w && memcmp(s.key, mapx, sizeof(float)*(size_t)w) == 0 &&
                     memcmp(s.key + w, mapy, sizeof(float)*(size_t)h) == 0;
    if (!hit) {
        // rows: cvRound(32 my) = 32 y + q0 for every y
        const auto q5 = [](float v) -> long long { const float t = v*32.f; if (!(t == t) || fabsf(t) >= 1048576.f) return LLONG_MIN; return (long long)lrintf(t); };
        const long long q0 = q5(mapy[0]);
        if (q0 == LLONG_MIN) return fail(GF_ERR_UNSUPPORTED, "%s: vertical map out of range", "gf_undistort_rotate_superview_u8c3");
        for (int y = 1; y < h; y++)
            if (q5(mapy[y]) != q0 + 32ll*y) return fail(GF_ERR_UNSUPPORTED, "%s: the vertical map is not a unit-step shift (row %lld): use the two calls", "gf_undistort_rotate_superview_u8c3", y);
        if (s.valid) {
            s.valid = false;
            if (s.ev_overflow) HIP_TRY(hipDeviceSynchronize());
            for (int i = 0; i < LENS_EVENTS; i++) if (s.ev_used[i]) HIP_TRY(hipEventSynchronize(s.ev[i]));
        }
        for (int i = 0; i < LENS_EVENTS; i++) s.ev_used[i] = false;
        s.ev_overflow = false;
        int* H = s.host;
        long long prev = LLONG_MIN;
        for (int x = 0; x < w; x++) {
            const long long q = q5(mapx[x]);
            if (q == LLONG_MIN || q < prev) return fail(GF_ERR_UNSUPPORTED, "%s: the horizontal map is not monotone (column %lld): use the two calls", "gf_undistort_rotate_superview_u8c3", x);
            prev = q; H[x] = (int)q;
        }
        // owner tile column of an output column: the tile of its second tap column (the window is sx1 - 1 .. sx1 + 2), clamped
        int* T = H + w;
        int x = 0;
        for (int k = 0; k <= tiles_x; k++) {
            while (x < w) {
                const int sx1 = (H[x] >> INTER_BITS) < -32768 ? -32768 : (H[x] >> INTER_BITS) > 32767 ? 32767 : (H[x] >> INTER_BITS);
                int owner = sx1 < 0 ? 0 : sx1/RT;
                if (owner > tiles_x - 1) owner = tiles_x - 1;
                if (owner >= k) break;
                x++;
            }
            T[k] = k == tiles_x ? w : x;
        }
        HIP_TRY(hipMemcpyAsync(s.dev, H, sizeof(int)*(size_t)(w + tiles_x + 1), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        memcpy(s.key, mapx, sizeof(float)*(size_t)w); memcpy(s.key + w, mapy, sizeof(float)*(size_t)h);
        s.w = w; s.h = h; s.mid_w = mid_w;
        s.c0row = (int)(q0 >> INTER_BITS); s.fy = (int)(q0 & 31);
        s.valid = true;
    }
    *colq = s.dev; *tcol = s.dev + w; *c0row = s.c0row; *fy = s.fy;
    stretch_used = true;
    return GF_OK;
}

struct DeviceGuard {
    int prev; bool ok;
    explicit DeviceGuard(int dev) : prev(0), ok(false) { if (hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess)) ok = true; }
    ~DeviceGuard() { (void)hipSetDevice(prev); }
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI

extern "C" {

GF_API int gf_version(void) { return GF_VERSION; }
GF_API const char* gf_last_error(void) { return g_err; }

GF_API int gf_get_cubic_table(int16_t* out_host)
{
    if (!out_host) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_get_cubic_table");
    std::call_once(g_cubic_once, build_cubic_host);
    memcpy(out_host, g_cubic_host, sizeof g_cubic_host);
    return GF_OK;
}

GF_API int gf_get_rp_table(double* out_host)
{
    if (!out_host) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_get_rp_table");
    build_rp_table(out_host);
    return GF_OK;
}

GF_API int gf_lens_scale_host(const double D_host[4], const double* t_host, size_t n, double* out_host)
{
    if (!D_host || !t_host || !out_host) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_lens_scale_host");
    double* tab = new (std::nothrow) double[LENS_DOUBLES];
    if (!tab) return fail(GF_ERR_HIP, "%s: out of host memory", "gf_lens_scale_host");
    build_lens_table(D_host, tab);
    for (size_t i = 0; i < n; i++) out_host[i] = lens_eval_host(tab, 1.0 + t_host[i]);
    delete[] tab;
    return GF_OK;
}

GF_API int gf_ctx_create(int device, void** ctx)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx pointer", "gf_ctx_create");
    *ctx = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(GF_ERR_INVALID_ARG, "gf_ctx_create: device %s%lld out of range", "", device);
    DeviceGuard dg(device);
    if (!dg.ok) return fail(GF_ERR_HIP, "gf_ctx_create: cannot select device %s%lld", "", device);
    std::call_once(g_cubic_once, build_cubic_host);
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return fail(GF_ERR_HIP, "%s: out of host memory", "gf_ctx_create");
    c->device = device; c->cubic_tab = nullptr; c->rp_tab = nullptr; c->tick = 0; c->n_cu = 0; c->k1_mode = g_k1_mode;
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu <= 0) c->n_cu = 256;
    for (int i = 0; i < LENS_SLOTS; i++) {
        LensSlot& s = c->lens[i];
        s.valid = false; s.dev = nullptr; s.host = nullptr; s.last_use = 0; s.ev_overflow = false;
        for (int k = 0; k < LENS_EVENTS; k++) { s.ev[k] = nullptr; s.ev_used[k] = false; s.ev_stream[k] = nullptr; }
    }
    hipError_t e = hipMalloc((void**)&c->cubic_tab, sizeof g_cubic_host);
    if (e == hipSuccess) e = hipMemcpy(c->cubic_tab, g_cubic_host, sizeof g_cubic_host, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&c->rp_tab, RP_TAB_DOUBLES*sizeof(double));
    if (e == hipSuccess) {
        double t[RP_TAB_DOUBLES];
        build_rp_table(t);
        e = hipMemcpy(c->rp_tab, t, sizeof t, hipMemcpyHostToDevice);
    }
    {
        StretchSlot& t = c->stretch;
        t.valid = false; t.key = nullptr; t.dev = nullptr; t.host = nullptr; t.ev_overflow = false; t.w = t.h = t.mid_w = 0; t.c0row = t.fy = 0;
        for (int k = 0; k < LENS_EVENTS; k++) { t.ev[k] = nullptr; t.ev_used[k] = false; t.ev_stream[k] = nullptr; }
        t.key = new (std::nothrow) float[2*32768];
        if (!t.key) e = hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMalloc((void**)&t.dev, STRETCH_MAX_INTS*sizeof(int));
        if (e == hipSuccess) e = hipHostMalloc((void**)&t.host, STRETCH_MAX_INTS*sizeof(int), hipHostMallocDefault);
        for (int k = 0; k < LENS_EVENTS && e == hipSuccess; k++) e = hipEventCreateWithFlags(&t.ev[k], hipEventDisableTiming);
    }
    for (int i = 0; i < LENS_SLOTS && e == hipSuccess; i++) {
        e = hipMalloc((void**)&c->lens[i].dev, LENS_DOUBLES*sizeof(double));
        if (e == hipSuccess) e = hipHostMalloc((void**)&c->lens[i].host, LENS_DOUBLES*sizeof(double), hipHostMallocDefault);
        for (int k = 0; k < LENS_EVENTS && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->lens[i].ev[k], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        gf_ctx_destroy(c);
        return fail(GF_ERR_HIP, "gf_ctx_create: %s (HIP error %lld)", "allocation/upload failed", (long long)e);
    }
    *ctx = c;
    return GF_OK;
}

GF_API int gf_ctx_set_kernel_mode(void* ctx, int mode)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_ctx_set_kernel_mode");
    if (mode != GF_KERNEL_DIRECT && mode != GF_KERNEL_RP && mode != GF_KERNEL_RL) return fail(GF_ERR_INVALID_ARG, "gf_ctx_set_kernel_mode: unknown mode %s%lld", "", mode);
    std::lock_guard<std::mutex> g(((Ctx*)ctx)->mu);
    ((Ctx*)ctx)->k1_mode = mode;
    return GF_OK;
}

GF_API int gf_ctx_destroy(void* ctx)
{
    if (!ctx) return GF_OK;
    Ctx* c = (Ctx*)ctx;
    DeviceGuard dg(c->device);
    if (c->cubic_tab) (void)hipFree(c->cubic_tab);
    if (c->rp_tab) (void)hipFree(c->rp_tab);
    delete[] c->stretch.key;
    if (c->stretch.dev) (void)hipFree(c->stretch.dev);
    if (c->stretch.host) (void)hipHostFree(c->stretch.host);
    for (int k = 0; k < LENS_EVENTS; k++) if (c->stretch.ev[k]) (void)hipEventDestroy(c->stretch.ev[k]);
    for (int i = 0; i < LENS_SLOTS; i++) {
        if (c->lens[i].dev) (void)hipFree(c->lens[i].dev);
        if (c->lens[i].host) (void)hipHostFree(c->lens[i].host);
        for (int k = 0; k < LENS_EVENTS; k++) if (c->lens[i].ev[k]) (void)hipEventDestroy(c->lens[i].ev[k]);
    }
    delete c;
    return GF_OK;
}

GF_API int gf_make_frame_xform(const double K[9], const double R[9], const double P[9], gf_frame_xform* out)
{
    if (!K || !R || !P || !out) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_make_frame_xform");
    // PR = P*R (k ascending), iR = PR^-1 by the adjugate in long double, rounded once to double.
    long double m[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P[i*3+k]*R[k*3+j];
            m[i*3+j] = s;
        }
    const long double A = m[4]*m[8] - m[5]*m[7], B = -(m[3]*m[8] - m[5]*m[6]), C = m[3]*m[7] - m[4]*m[6];
    const long double det = m[0]*A + m[1]*B + m[2]*C;
    if (det == 0 || det != det) return fail(GF_ERR_SINGULAR, "%s: P*R is singular", "gf_make_frame_xform");
    const long double r = 1.0L/det;
    out->iR[0] = (double)(A*r); out->iR[1] = (double)(-(m[1]*m[8] - m[2]*m[7])*r); out->iR[2] = (double)((m[1]*m[5] - m[2]*m[4])*r);
    out->iR[3] = (double)(B*r); out->iR[4] = (double)((m[0]*m[8] - m[2]*m[6])*r);  out->iR[5] = (double)(-(m[0]*m[5] - m[2]*m[3])*r);
    out->iR[6] = (double)(C*r); out->iR[7] = (double)(-(m[0]*m[7] - m[1]*m[6])*r); out->iR[8] = (double)((m[0]*m[4] - m[1]*m[3])*r);
    out->f[0] = K[0]; out->f[1] = K[4]; out->c[0] = K[2]; out->c[1] = K[5];
    return GF_OK;
}

GF_API int gf_make_frame_xforms(const double K[9], const double* R, const double* P, size_t n, gf_frame_xform* out)
{
    if (!K || !R || !P || !out) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_make_frame_xforms");
    for (size_t i = 0; i < n; i++) {
        const int rc = gf_make_frame_xform(K, R + 9*i, P + 9*i, out + i);
        if (rc) return fail(rc, "gf_make_frame_xforms: %s (frame %lld)", "P*R is singular", (long long)i);
    }
    return GF_OK;
}

static int maps_common(void* ctx, const double* D, const gf_frame_xform* xf, int w, int h, void* m1, size_t s1, void* m2, size_t s2,
                       bool fixed, gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_fisheye_maps");
    if (!xf || !m1 || !m2) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_fisheye_maps");
    if (w <= 0 || h <= 0 || w >= SHRT_MAX || h >= SHRT_MAX) return fail(GF_ERR_INVALID_ARG, "%s: bad size, w=%lld", "gf_fisheye_maps", w);
    const size_t min1 = (size_t)w*4, min2 = fixed ? (size_t)w*2 : (size_t)w*4;
    if (s1 < min1 || s2 < min2 || (s1 & 3) || (s2 & (fixed ? 1 : 3)) || ((uintptr_t)m1 & 3) || ((uintptr_t)m2 & (fixed ? 1 : 3)))
        return fail(GF_ERR_INVALID_ARG, "%s: map step/alignment invalid (step1=%lld)", "gf_fisheye_maps", (long long)s1);
    Ctx* c = (Ctx*)ctx;
    DeviceGuard dg(c->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "gf_fisheye_maps");
    const double* lens; LensK lk;
    LensLease lease(c, (hipStream_t)stream);
    int rc = lease.acquire(D, &lens, &lk); if (rc) return rc;
    dim3 grid((w + 63)/64, (h + 63)/64);
    if (fixed && c->k1_mode != GF_KERNEL_DIRECT) hipLaunchKernelGGL(k_fisheye_maps_rp, dim3((w + RT - 1)/RT, (h + RT - 1)/RT), dim3(BLOCK), 0, (hipStream_t)stream,
                                                    to_xf(xf), lk, lens, (const double*)c->rp_tab, w, h, (uint8_t*)m1, s1, (uint8_t*)m2, s2);
    else if (fixed) hipLaunchKernelGGL(k_fisheye_maps<true>, grid, dim3(BLOCK), 0, (hipStream_t)stream, to_xf(xf), lk, lens, w, h, (uint8_t*)m1, s1, (uint8_t*)m2, s2);
    else       hipLaunchKernelGGL(k_fisheye_maps<false>, grid, dim3(BLOCK), 0, (hipStream_t)stream, to_xf(xf), lk, lens, w, h, (uint8_t*)m1, s1, (uint8_t*)m2, s2);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_fisheye_maps_16sc2(void* ctx, const double D[4], const gf_frame_xform* xf, int w, int h, int16_t* map1, size_t s1, uint16_t* map2, size_t s2, gf_stream_t stream)
{
    return maps_common(ctx, D, xf, w, h, map1, s1, map2, s2, true, stream);
}

GF_API int gf_fisheye_maps_32f(void* ctx, const double D[4], const gf_frame_xform* xf, int w, int h, float* mx, size_t sx, float* my, size_t sy, gf_stream_t stream)
{
    return maps_common(ctx, D, xf, w, h, mx, sx, my, sy, false, stream);
}

GF_API int gf_convert_maps_32f_16sc2(void* ctx, const float* mx, size_t mxs, const float* my, size_t mys,
                              float sxn, float sxd, float syn, float syd, int w, int h,
                              int16_t* map1, size_t s1, uint16_t* map2, size_t s2, gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_convert_maps_32f_16sc2");
    if (!mx || !my || !map1 || !map2) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_convert_maps_32f_16sc2");
    if (w <= 0 || h <= 0 || w >= SHRT_MAX || h >= SHRT_MAX) return fail(GF_ERR_INVALID_ARG, "%s: bad size, w=%lld", "gf_convert_maps_32f_16sc2", w);
    if (mxs < (size_t)w*4 || mys < (size_t)w*4 || s1 < (size_t)w*4 || s2 < (size_t)w*2 || (mxs & 3) || (mys & 3) || (s1 & 3) || (s2 & 1) ||
        ((uintptr_t)mx & 3) || ((uintptr_t)my & 3) || ((uintptr_t)map1 & 3) || ((uintptr_t)map2 & 1))
        return fail(GF_ERR_INVALID_ARG, "%s: step/alignment invalid (w=%lld)", "gf_convert_maps_32f_16sc2", w);
    if (sxd == 0.f || syd == 0.f) return fail(GF_ERR_INVALID_ARG, "%s: zero scale denominator", "gf_convert_maps_32f_16sc2");
    const int scale_on = !(sxn == 1.f && sxd == 1.f && syn == 1.f && syd == 1.f);
    DeviceGuard dg(((Ctx*)ctx)->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "libgfremap");
    dim3 grid((w + 63)/64, (h + 3)/4);
    hipLaunchKernelGGL(k_convert_maps, grid, dim3(BLOCK), 0, (hipStream_t)stream, (const uint8_t*)mx, mxs, (const uint8_t*)my, mys,
                       sxn, sxd, syn, syd, scale_on, w, h, (uint8_t*)map1, s1, (uint8_t*)map2, s2);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_remap_linear_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep,
                       void* dst, int dw, int dh, size_t dstep,
                       const int16_t* map1, size_t s1, const uint16_t* map2, size_t s2,
                       int border, const double bv[4], gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_remap_linear_c3");
    int rc = check_common(depth, border); if (rc) return rc;
    rc = check_img("gf_remap_linear_c3 src", src, sw, sh, sstep, depth); if (rc) return rc;
    rc = check_img("gf_remap_linear_c3 dst", dst, dw, dh, dstep, depth); if (rc) return rc;
    if (!map1 || !map2) return fail(GF_ERR_INVALID_ARG, "%s: null map", "gf_remap_linear_c3");
    if (s1 < (size_t)dw*4 || s2 < (size_t)dw*2 || (s1 & 1) || (s2 & 1) || ((uintptr_t)map1 & 1) || ((uintptr_t)map2 & 1))
        return fail(GF_ERR_INVALID_ARG, "%s: map step/alignment invalid (step1=%lld)", "gf_remap_linear_c3", (long long)s1);
    if (ranges_overlap(src, sstep*sh, dst, dstep*dh)) return fail(GF_ERR_INVALID_ARG, "%s: src and dst overlap", "gf_remap_linear_c3");
    DeviceGuard dg(((Ctx*)ctx)->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "libgfremap");
    RemapArgs A = make_args(depth, src, sw, sh, sstep, 0, dst, dw, dh, dstep, 0, bv);
    dim3 grid(A.tiles_x*A.tiles_y, 1);
    DISPATCH_DEPTH_BORDER(k_remap_linear, depth, border, grid, (hipStream_t)stream, A, (const uint8_t*)map1, s1, (const uint8_t*)map2, s2);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

static int fused_checks(const char* fn, void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                        void* dst, int dw, int dh, size_t dstep, size_t dframe, const void* xf, long long n_in, long long n_out, int border);

GF_API int gf_remap_cubic_u8c3_f32maps(void* ctx, const uint8_t* src, int sw, int sh, size_t sstep,
                                uint8_t* dst, int dw, int dh, size_t dstep,
                                const float* mx, size_t mxs, const float* my, size_t mys,
                                int border, const double bv[4], gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_remap_cubic_u8c3_f32maps");
    int rc = check_common(GF_DEPTH_U8, border); if (rc) return rc;
    rc = check_img("gf_remap_cubic_u8c3_f32maps src", src, sw, sh, sstep, GF_DEPTH_U8); if (rc) return rc;
    rc = check_img("gf_remap_cubic_u8c3_f32maps dst", dst, dw, dh, dstep, GF_DEPTH_U8); if (rc) return rc;
    if (!mx || !my) return fail(GF_ERR_INVALID_ARG, "%s: null map", "gf_remap_cubic_u8c3_f32maps");
    if (mxs < (size_t)dw*4 || mys < (size_t)dw*4 || (mxs & 3) || (mys & 3) || ((uintptr_t)mx & 3) || ((uintptr_t)my & 3))
        return fail(GF_ERR_INVALID_ARG, "%s: map step/alignment invalid (w=%lld)", "gf_remap_cubic_u8c3_f32maps", dw);
    if (ranges_overlap(src, sstep*sh, dst, dstep*dh)) return fail(GF_ERR_INVALID_ARG, "%s: src and dst overlap", "gf_remap_cubic_u8c3_f32maps");
    int cval[3]; cval_from(bv, GF_DEPTH_U8, cval);
    DeviceGuard dg(((Ctx*)ctx)->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "libgfremap");
    dim3 grid((dw + 63)/64, (dh + 3)/4);
    hipLaunchKernelGGL(k_remap_cubic_u8c3, grid, dim3(BLOCK), 0, (hipStream_t)stream, src, sw, sh, sstep, dst, dw, dh, dstep,
                       (const uint8_t*)mx, mxs, (const uint8_t*)my, mys, ((Ctx*)ctx)->cubic_tab, border, cval[0], cval[1], cval[2]);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_superview_cubic_u8c3(void* ctx, const uint8_t* src, int sw, int sh, size_t sstep, size_t sframe,
                                   uint8_t* dst, int dw, int dh, size_t dstep, size_t dframe,
                                   const float* mapx_row, const float* mapy_col, int n_frames,
                                   int border, const double bv[4], gf_stream_t stream)
{
    int rc = fused_checks("gf_superview_cubic_u8c3", ctx, GF_DEPTH_U8, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, mapx_row, n_frames, n_frames, border);
    if (rc) return rc;
    if (!mapy_col || ((uintptr_t)mapx_row & 3) || ((uintptr_t)mapy_col & 3)) return fail(GF_ERR_INVALID_ARG, "%s: null or misaligned map", "gf_superview_cubic_u8c3");
    int cval[3]; cval_from(bv, GF_DEPTH_U8, cval);
    DeviceGuard dg(((Ctx*)ctx)->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "libgfremap");
    Ctx* c = (Ctx*)ctx;
    const unsigned strips_x = (unsigned)(dw + BLOCK - 1)/BLOCK, strips_y = (unsigned)(dh + SV_ROWS - 1)/SV_ROWS;
    const unsigned long long n_work = (unsigned long long)strips_x*strips_y*(unsigned)n_frames;
    if (n_work >= (1ull << 31)) return fail(GF_ERR_UNSUPPORTED, "%s: too many strips in one call (%lld)", "gf_superview_cubic_u8c3", (long long)n_work);
    static const unsigned sv_wg_per_cu = []{ const char* e = getenv("GF_SV_WG_PER_CU"); int v = e ? atoi(e) : 0; return (unsigned)(v > 0 && v <= 16 ? v : 4); }();
    unsigned blocks = (unsigned)c->n_cu*sv_wg_per_cu;              // persistent: 32 KB of LDS and ~100 VGPRs per workgroup, four resident per CU
    if (blocks > n_work) blocks = (unsigned)n_work;
    hipLaunchKernelGGL(k_superview_cubic, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, src, sw, sh, (unsigned)sstep, sframe, dst, dw, dh, dstep, dframe,
                       mapx_row, mapy_col, c->cubic_tab, border, cval[0], cval[1], cval[2], strips_x, strips_y, (unsigned)n_work,
                       (((uintptr_t)dst | dstep | dframe) & 3) == 0 ? 1 : 0, (((uintptr_t)src | sstep | sframe) & 3) == 0 ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

static int fused_checks(const char* fn, void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                        void* dst, int dw, int dh, size_t dstep, size_t dframe, const void* xf, long long n_in, long long n_out, int border)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", fn);
    int rc = check_common(depth, border); if (rc) return rc;
    rc = check_img(fn, src, sw, sh, sstep, depth); if (rc) return rc;
    rc = check_img(fn, dst, dw, dh, dstep, depth); if (rc) return rc;
    if (!xf) return fail(GF_ERR_INVALID_ARG, "%s: null xf_dev", fn);
    if (n_in <= 0 || n_out <= 0 || n_out > 65535) return fail(GF_ERR_INVALID_ARG, "%s: frame count out of range (1..65535): %lld", fn, n_out);
    if (sframe != 0 && sframe < sstep*(size_t)sh) return fail(GF_ERR_INVALID_ARG, "%s: src_frame_stride smaller than a frame (%lld)", fn, (long long)sframe);
    if (n_out > 1 && dframe < dstep*(size_t)dh) return fail(GF_ERR_INVALID_ARG, "%s: dst_frame_stride smaller than a frame (%lld)", fn, (long long)dframe);
    if (depth == GF_DEPTH_U16 && ((sframe & 1) || (dframe & 1))) return fail(GF_ERR_INVALID_ARG, "%s: u16 frame strides must be even (%lld)", fn, (long long)sframe);
    const size_t sbytes = sframe ? sframe*(size_t)(n_in - 1) + sstep*sh : sstep*sh;
    const size_t dbytes = dframe*(size_t)(n_out - 1) + dstep*dh;
    if (ranges_overlap(src, sbytes, dst, dbytes)) return fail(GF_ERR_INVALID_ARG, "%s: src and dst overlap", fn);
    return GF_OK;
}

// What launch_fused needs from a context (the allocation-free entry point has no Ctx: its tables live in `scratch`).
struct FusedEnv { int k1_mode; int n_cu; const double* rp_tab; };

static int launch_fused(const FusedEnv* c, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                        void* dst, int dw, int dh, size_t dstep, size_t dframe, const double* lens, LensK lk,
                        const gf_frame_xform* xf_dev, int n_frames, int border, const double* bv, hipStream_t stream,
                        const float* stretch = nullptr, const RotMap* rot = nullptr)
{
    // with `rot`, sw x sh is the stored frame and the kernel sees the frame cv2.rotate would have made of it
    if (rot && (c->k1_mode != GF_KERNEL_RP || stretch))
        return fail(GF_ERR_UNSUPPORTED, "%s: the rotated-source form runs on the RP kernel only (rotate with gf_rotate_c3 first)", "gf_undistort_rotate_rot_c3");
    RemapArgs A = make_args(depth, src, rot && rot->swap ? sh : sw, rot && rot->swap ? sw : sh, sstep, sframe, dst, dw, dh, dstep, dframe, bv);
    if (stretch && !(stretch[0] == 1.f && stretch[1] == 1.f && stretch[2] == 1.f && stretch[3] == 1.f)) {
        A.stretch_on = 1; A.sxn = stretch[0]; A.sxd = stretch[1]; A.syn = stretch[2]; A.syd = stretch[3];
    }
    // stretched presets take their coordinates through float32 maps, which get_maps builds from the per-pixel (DIRECT) path
    if (c->k1_mode != GF_KERNEL_DIRECT && !stretch) {
        // RP coordinates on 64 x 64 tiles (mode RP, the default); in mode RL u8 frames with 16-byte aligned base, row pitch and
        // frame stride also stage their source taps through LDS
        A.tiles_x = (dw + RT - 1)/RT; A.tiles_y = (dh + RT - 1)/RT;
        const unsigned tiles_pad = ((unsigned)(A.tiles_x*A.tiles_y) + NXCD - 1) & ~(unsigned)(NXCD - 1);
        const unsigned long long n_work = (unsigned long long)tiles_pad*(unsigned)n_frames;
        if (n_work >= (1ull << 31)) return fail(GF_ERR_UNSUPPORTED, "%s: too many tiles in one call (%lld)", "gf_undistort_rotate_c3", (long long)n_work);
        const bool ldsbox = c->k1_mode == GF_KERNEL_RL && depth == GF_DEPTH_U8 && ((uintptr_t)src & 15) == 0 && (sstep & 15) == 0 && (sframe & 15) == 0 &&
                            (size_t)(sh - 1)*sstep + (size_t)sw*3 >= 32;
        // 32 workgroups per CU once each still gets >= 7 work items, else 16 (tools/wg_sweep.sh, C3: 64 frames 20.3 -> 19.7 us/frame,
        // 32 frames 21.3 -> 21.0, 16 frames 22.6 with 16 against 23.3 with 32; 40..64 per CU change nothing)
        const unsigned wg_rp = WG_PER_CU_RP ? WG_PER_CU_RP : (n_work >= (unsigned long long)c->n_cu*32u*7u ? 32u : 16u);
        unsigned blocks = (unsigned)c->n_cu*(ldsbox ? WG_PER_CU_RL : wg_rp) & ~(unsigned)(NXCD - 1);
        if (blocks == 0) blocks = NXCD;
        if (blocks > n_work) blocks = (unsigned)n_work;
        dim3 grid(blocks, 1);
#ifdef GF_LB
        if (ldsbox) {
            unsigned lb_blocks = (unsigned)c->n_cu*wg_rp & ~(unsigned)(NXCD - 1);
            if (lb_blocks == 0) lb_blocks = NXCD;
            if (lb_blocks > n_work) lb_blocks = (unsigned)n_work;
            dim3 lgrid(lb_blocks, 1), block(BLOCK);
            if (border == GF_BORDER_CONSTANT)       hipLaunchKernelGGL((k_undistort_rotate_lb<GF_BORDER_CONSTANT>), lgrid, block, 0, stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, tiles_pad, (unsigned)n_work);
            else if (border == GF_BORDER_REPLICATE) hipLaunchKernelGGL((k_undistort_rotate_lb<GF_BORDER_REPLICATE>), lgrid, block, 0, stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, tiles_pad, (unsigned)n_work);
            else                                    hipLaunchKernelGGL((k_undistort_rotate_lb<GF_BORDER_TRANSPARENT>), lgrid, block, 0, stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, tiles_pad, (unsigned)n_work);
        } else
#endif
        if (ldsbox) {
            dim3 block(BLOCK_RL);
            if (border == GF_BORDER_CONSTANT)       hipLaunchKernelGGL((k_undistort_rotate_rl<GF_BORDER_CONSTANT>), grid, block, 0, stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, tiles_pad, (unsigned)n_work);
            else if (border == GF_BORDER_REPLICATE) hipLaunchKernelGGL((k_undistort_rotate_rl<GF_BORDER_REPLICATE>), grid, block, 0, stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, tiles_pad, (unsigned)n_work);
            else                                    hipLaunchKernelGGL((k_undistort_rotate_rl<GF_BORDER_TRANSPARENT>), grid, block, 0, stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, tiles_pad, (unsigned)n_work);
        } else {
            if (rot) {
#define LAUNCH_ROT(DEPTH_, BORDER_) hipLaunchKernelGGL((k_undistort_rotate_rp<DEPTH_, BORDER_, true>), grid, dim3(BLOCK), 0, stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, tiles_pad, (unsigned)n_work, *rot)
                if (depth == GF_DEPTH_U8) { if (border == GF_BORDER_CONSTANT) LAUNCH_ROT(GF_DEPTH_U8, GF_BORDER_CONSTANT); else if (border == GF_BORDER_REPLICATE) LAUNCH_ROT(GF_DEPTH_U8, GF_BORDER_REPLICATE); else LAUNCH_ROT(GF_DEPTH_U8, GF_BORDER_TRANSPARENT); }
                else                      { if (border == GF_BORDER_CONSTANT) LAUNCH_ROT(GF_DEPTH_U16, GF_BORDER_CONSTANT); else if (border == GF_BORDER_REPLICATE) LAUNCH_ROT(GF_DEPTH_U16, GF_BORDER_REPLICATE); else LAUNCH_ROT(GF_DEPTH_U16, GF_BORDER_TRANSPARENT); }
#undef LAUNCH_ROT
            } else {
                DISPATCH_DEPTH_BORDER(k_undistort_rotate_rp, depth, border, grid, stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, tiles_pad, (unsigned)n_work, NoRot());
            }
        }
    } else {
        // persistent grid: WG_PER_CU workgroups per CU (what the kernel's register and LDS use allows), a multiple of 8
        const unsigned tiles_pad = ((unsigned)(A.tiles_x*A.tiles_y) + NXCD - 1) & ~(unsigned)(NXCD - 1);
        const unsigned long long n_work = (unsigned long long)tiles_pad*(unsigned)n_frames;
        if (n_work >= (1ull << 31)) return fail(GF_ERR_UNSUPPORTED, "%s: too many tiles in one call (%lld)", "gf_undistort_rotate_c3", (long long)n_work);
        unsigned blocks = (unsigned)c->n_cu*WG_PER_CU & ~(unsigned)(NXCD - 1);
        if (blocks == 0) blocks = NXCD;
        if (blocks > n_work) blocks = (unsigned)n_work;          // n_work is a multiple of 8
        dim3 grid(blocks, 1);
        DISPATCH_DEPTH_BORDER(k_undistort_rotate, depth, border, grid, stream, A, (const Xf*)xf_dev, lens, lk, tiles_pad, (unsigned)n_work);
    }
    HIP_TRY(hipGetLastError());
    return GF_OK;
}


#if defined(GF_LB) && defined(GF_LB_STATS)
GF_API int gf_debug_read_lb_stats(unsigned long long* out /* [4] */)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lb_stats), sizeof(unsigned long long)*4));
    return GF_OK;
}
#endif
#ifdef GF_TRACE_WG
GF_API int gf_debug_read_wg_trace(unsigned long long* out /* [8192][3] */)      // tools/kbench prints the timeline
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_trace), sizeof(unsigned long long)*8192*3));
    return GF_OK;
}
#endif

GF_API int gf_undistort_rotate_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                           void* dst, int dw, int dh, size_t dstep, size_t dframe,
                           const double D[4], const gf_frame_xform* xf_dev, int n_frames, int border, const double bv[4], gf_stream_t stream)
{
    int rc = fused_checks("gf_undistort_rotate_c3", ctx, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, xf_dev, n_frames, n_frames, border);
    if (rc) return rc;
    Ctx* c = (Ctx*)ctx;
    DeviceGuard dg(c->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "gf_undistort_rotate_c3");
    const double* lens; LensK lk;
    LensLease lease(c, (hipStream_t)stream);
    rc = lease.acquire(D, &lens, &lk); if (rc) return rc;
    const FusedEnv env{c->k1_mode, c->n_cu, c->rp_tab};
    return launch_fused(&env, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, lens, lk, xf_dev, n_frames,
                        border, bv, (hipStream_t)stream);
}

static int rot_map_from_code(const char* fn, int code, int sw, int sh, RotMap* m)
{
    m->stw = sw; m->sth = sh;
    if (code == GF_ROTATE_90_CLOCKWISE)             { m->swap = 1; m->cs = 1;  m->rs = -1; }     // R[y][x] = S[sh-1-x][y]
    else if (code == GF_ROTATE_180)                 { m->swap = 0; m->cs = -1; m->rs = -1; }     // R[y][x] = S[sh-1-y][sw-1-x]
    else if (code == GF_ROTATE_90_COUNTERCLOCKWISE) { m->swap = 1; m->cs = -1; m->rs = 1; }      // R[y][x] = S[x][sw-1-y]
    else return fail(GF_ERR_INVALID_ARG, "%s: rotate code %lld is not one of cv2.ROTATE_90_CLOCKWISE (0), ROTATE_180 (1), ROTATE_90_COUNTERCLOCKWISE (2)", fn, code);
    return GF_OK;
}

GF_API int gf_rotate_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                        void* dst, size_t dstep, size_t dframe, int n_frames, int rotate_code, gf_stream_t stream)
{
    const char* fn = "gf_rotate_c3";
    RotMap m;
    int rc = rot_map_from_code(fn, rotate_code, sw, sh, &m); if (rc) return rc;
    const int dw = m.swap ? sh : sw, dh = m.swap ? sw : sh;
    static const double one[13] = {1};
    rc = fused_checks(fn, ctx, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, one, n_frames, n_frames, GF_BORDER_CONSTANT);
    if (rc) return rc;
    DeviceGuard dg(((Ctx*)ctx)->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", fn);
    dim3 grid((dw + 63)/64, (dh + 3)/4, n_frames);
    if (depth == GF_DEPTH_U8) hipLaunchKernelGGL(k_rotate_c3<uint8_t>, grid, dim3(BLOCK), 0, (hipStream_t)stream, (const uint8_t*)src, (unsigned)sstep, sframe, (uint8_t*)dst, dw, dh, dstep, dframe, m);
    else                      hipLaunchKernelGGL(k_rotate_c3<uint16_t>, grid, dim3(BLOCK), 0, (hipStream_t)stream, (const uint8_t*)src, (unsigned)sstep, sframe, (uint8_t*)dst, dw, dh, dstep, dframe, m);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_undistort_rotate_rot_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                                      void* dst, int dw, int dh, size_t dstep, size_t dframe,
                                      const double D[4], const gf_frame_xform* xf_dev, int n_frames, int rotate_code,
                                      int border, const double bv[4], gf_stream_t stream)
{
    const char* fn = "gf_undistort_rotate_rot_c3";
    RotMap m;
    int rc = rot_map_from_code(fn, rotate_code, sw, sh, &m); if (rc) return rc;
    rc = fused_checks(fn, ctx, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, xf_dev, n_frames, n_frames, border);
    if (rc) return rc;
    Ctx* c = (Ctx*)ctx;
    DeviceGuard dg(c->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", fn);
    const double* lens; LensK lk;
    LensLease lease(c, (hipStream_t)stream);
    if (c->k1_mode != GF_KERNEL_RP) return fail(GF_ERR_UNSUPPORTED, "%s: the rotated-source form runs on the RP kernel only (rotate with gf_rotate_c3 first)", fn);
    rc = lease.acquire(D, &lens, &lk); if (rc) return rc;
    const FusedEnv env{c->k1_mode, c->n_cu, c->rp_tab};
    return launch_fused(&env, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, lens, lk, xf_dev, n_frames,
                        border, bv, (hipStream_t)stream, nullptr, &m);
}

GF_API int gf_undistort_rotate_stretched_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                                     void* dst, int dw, int dh, size_t dstep, size_t dframe,
                                     const double D[4], const gf_frame_xform* xf_dev, int n_frames,
                                     float sx_num, float sx_den, float sy_num, float sy_den,
                                     int border, const double bv[4], gf_stream_t stream)
{
    int rc = fused_checks("gf_undistort_rotate_stretched_c3", ctx, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, xf_dev, n_frames, n_frames, border);
    if (rc) return rc;
    if (sx_den == 0.f || sy_den == 0.f) return fail(GF_ERR_INVALID_ARG, "%s: zero scale denominator", "gf_undistort_rotate_stretched_c3");
    Ctx* c = (Ctx*)ctx;
    DeviceGuard dg(c->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "gf_undistort_rotate_stretched_c3");
    const double* lens; LensK lk;
    LensLease lease(c, (hipStream_t)stream);
    rc = lease.acquire(D, &lens, &lk); if (rc) return rc;
    const FusedEnv env{c->k1_mode, c->n_cu, c->rp_tab};
    const float sc[4] = {sx_num, sx_den, sy_num, sy_den};
    return launch_fused(&env, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, lens, lk, xf_dev, n_frames,
                        border, bv, (hipStream_t)stream, sc);
}

GF_API int gf_undistort_rotate_superview_u8c3(void* ctx, const uint8_t* src, int sw, int sh, size_t sstep, size_t sframe,
                                              int mid_w, int mid_h,
                                              uint8_t* dst, int dw, int dh, size_t dstep, size_t dframe,
                                              const double D[4], const gf_frame_xform* xf_dev, int n_frames,
                                              const float* mapx_row_host, const float* mapy_col_host,
                                              int border, const double bv[4], gf_stream_t stream)
{
    const char* fn = "gf_undistort_rotate_superview_u8c3";
    int rc = fused_checks(fn, ctx, GF_DEPTH_U8, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, xf_dev, n_frames, n_frames, border);
    if (rc) return rc;
    if (border == GF_BORDER_TRANSPARENT) return fail(GF_ERR_UNSUPPORTED, "%s: BORDER_TRANSPARENT needs the previous intermediate frame; use the two calls", fn);
    if (mid_w <= 0 || mid_h <= 0 || mid_w >= SHRT_MAX || mid_h >= SHRT_MAX) return fail(GF_ERR_INVALID_ARG, "%s: bad intermediate size, w=%lld", fn, mid_w);
    if (!mapx_row_host || !mapy_col_host) return fail(GF_ERR_INVALID_ARG, "%s: null map", fn);
    Ctx* c = (Ctx*)ctx;
    DeviceGuard dg(c->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", fn);
    const double* lens; LensK lk;
    LensLease lease(c, (hipStream_t)stream);
    rc = lease.acquire(D, &lens, &lk); if (rc) return rc;
    FsOut F;
    rc = lease.acquire_stretch(mapx_row_host, mapy_col_host, dw, dh, mid_w, mid_h, &F.colq, &F.tcol, &F.c0row, &F.fy); if (rc) return rc;
    F.dst = dst; F.w = dw; F.h = dh; F.step = dstep; F.frame = dframe;
    F.vec_ok = (((uintptr_t)dst | dstep | dframe) & 3) == 0 ? 1 : 0;
    RemapArgs A = make_args(GF_DEPTH_U8, src, sw, sh, sstep, sframe, nullptr, mid_w, mid_h, 0, 0, bv);
    A.tiles_x = (mid_w + RT - 1)/RT; A.tiles_y = (mid_h + RT - 1)/RT;
    const unsigned tiles_pad = ((unsigned)(A.tiles_x*A.tiles_y) + NXCD - 1) & ~(unsigned)(NXCD - 1);
    const unsigned long long n_work = (unsigned long long)tiles_pad*(unsigned)n_frames;
    if (n_work >= (1ull << 31)) return fail(GF_ERR_UNSUPPORTED, "%s: too many tiles in one call (%lld)", fn, (long long)n_work);
    static const unsigned fs_wg_per_cu = []{ const char* e = getenv("GF_FS_WG_PER_CU"); int v = e ? atoi(e) : 0; return (unsigned)(v > 0 && v <= 64 ? v : 16); }();
    unsigned blocks = (unsigned)c->n_cu*fs_wg_per_cu & ~(unsigned)(NXCD - 1);
    if (blocks == 0) blocks = NXCD;
    if (blocks > n_work) blocks = (unsigned)n_work;
    if (border == GF_BORDER_CONSTANT) hipLaunchKernelGGL((k_undistort_rotate_superview<GF_BORDER_CONSTANT>), dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, A, F, (const Xf*)xf_dev, lens, lk,
                                                         (const double*)c->rp_tab, (const int16_t*)c->cubic_tab, tiles_pad, (unsigned)n_work);
    else                              hipLaunchKernelGGL((k_undistort_rotate_superview<GF_BORDER_REPLICATE>), dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, A, F, (const Xf*)xf_dev, lens, lk,
                                                         (const double*)c->rp_tab, (const int16_t*)c->cubic_tab, tiles_pad, (unsigned)n_work);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

// ---------------------------------------------------------------------------------------------
// gf_undistort_rotate_c3_noalloc: gf_undistort_rotate_c3 for callers that cannot allocate (HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc).  It needs no context: the device memory the fused kernel uses beyond its inputs and
// outputs -- the lens table, the RP table and the per-frame transforms -- lives in `scratch`, a static device array that
// every GPU the library runs on holds until the process ends.  The tables are built on the host exactly as LensLease::acquire() /
// gf_ctx_create() build them and reach the device as kernel arguments (k_scratch_put: 2 KB per launch), so the call makes
// no hipMalloc / hipFree / hipMemcpy and queues nothing but kernels on the caller's stream.
// `scratch` is a host-side stand-in in host code, not a device address: hipGetSymbolAddress gives the address, once per
// device, and only that pointer is handed to kernels.
constexpr int NOALLOC_MAX_FRAMES = GF_NOALLOC_MAX_FRAMES;
struct NoAllocScratch {
    double lens[LENS_DOUBLES];                 // 16 KB
    double rp[RP_TAB_DOUBLES];                 // 4.6 KB
    double xf[NOALLOC_MAX_FRAMES][13];         // 6.5 KB
};
static_assert(sizeof(Xf) == 13*sizeof(double) && offsetof(NoAllocScratch, rp) % 16 == 0 && offsetof(NoAllocScratch, xf) % 8 == 0, "scratch layout");
}  // extern "C"

static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

namespace {
constexpr int PUT_DOUBLES = 256;               // doubles per k_scratch_put launch (2 KB of kernel arguments)
struct PutChunk { double v[PUT_DOUBLES]; };
__global__ void __launch_bounds__(PUT_DOUBLES) k_scratch_put(double* dst, PutChunk c, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = c.v[threadIdx.x];
}

int scratch_put(double* dst_dev, const double* src_host, int n, hipStream_t stream)
{
    for (int o = 0; o < n; o += PUT_DOUBLES) {
        PutChunk c;
        const int m = n - o < PUT_DOUBLES ? n - o : PUT_DOUBLES;
        memcpy(c.v, src_host + o, (size_t)m*sizeof(double));
        if (m < PUT_DOUBLES) memset(c.v + m, 0, (size_t)(PUT_DOUBLES - m)*sizeof(double));
        hipLaunchKernelGGL(k_scratch_put, dim3(1), dim3(PUT_DOUBLES), 0, stream, dst_dev + o, c, m);
    }
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

constexpr int NOALLOC_MAX_DEVICES = 64;
struct NoAllocDev { NoAllocScratch* ptr; int n_cu; };    // ptr: this device's copy of `scratch` (each GPU has its own, at its own address)
NoAllocDev g_noalloc[NOALLOC_MAX_DEVICES];
std::mutex g_noalloc_mu;
}  // namespace

extern "C" {

GF_API int gf_undistort_rotate_c3_noalloc(int device, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                                          void* dst, int dw, int dh, size_t dstep, size_t dframe,
                                          const double D[4], const gf_frame_xform* xf_host, int n_frames, int border, const double bv[4],
                                          gf_stream_t stream)
{
    static int dummy_ctx;      // fused_checks only tests the context pointer for NULL
    int rc = fused_checks("gf_undistort_rotate_c3_noalloc", &dummy_ctx, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, xf_host, n_frames, n_frames, border);
    if (rc) return rc;
    if (n_frames > NOALLOC_MAX_FRAMES)
        return fail(GF_ERR_INVALID_ARG, "gf_undistort_rotate_c3_noalloc: %s%lld frames exceed GF_NOALLOC_MAX_FRAMES (64); split the batch or use gf_undistort_rotate_c3", "", n_frames);
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return fail(GF_ERR_INVALID_ARG, "gf_undistort_rotate_c3_noalloc: device %s%lld out of range", "", device);
    if (!D) return fail(GF_ERR_INVALID_ARG, "%s: null distortion coefficients", "gf_undistort_rotate_c3_noalloc");
    LensK lk;
    for (int i = 0; i < 4; i++) {
        if (!(D[i] == D[i]) || fabs(D[i]) > 1e6) return fail(GF_ERR_INVALID_ARG, "lens: distortion coefficient %s%lld is not finite", "", i);
        lk.k[i] = D[i];
    }
    {   // as LensLease::acquire()
        const double tm = atan(sqrt((double)((1 << LENS_OCT) - 1))), t2 = tm*tm;
        lk.bound = tm*(1 + t2*(fabs(D[0]) + t2*(fabs(D[1]) + t2*(fabs(D[2]) + t2*fabs(D[3])))));
    }
    DeviceGuard dg(device);
    if (!dg.ok) return fail(GF_ERR_HIP, "gf_undistort_rotate_c3_noalloc: cannot select device %s%lld", "", device);
    std::lock_guard<std::mutex> g(g_noalloc_mu);
    NoAllocDev& nd = g_noalloc[device];
    if (!nd.ptr) {             // first call on this device, with the device current: resolve its copy of the symbol
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n_cu <= 0) n_cu = 256;
        nd.n_cu = n_cu;
        nd.ptr = (NoAllocScratch*)p;
    }
    NoAllocScratch* S = nd.ptr;
    // Every call rewrites what it uses (the array keeps the last call's contents, and a captured graph must be
    // self-contained): lens table, RP table, this call's transforms.  Stream order puts them before the fused kernel.
    // The host copies of the tables are kept between calls (under g_noalloc_mu, held here): the long-double lens table costs
    // about a millisecond to build, and a serving loop calls with the same lens every time.  The device copies are still
    // rewritten by every call.
    static double host_lens[LENS_DOUBLES], host_rp[RP_TAB_DOUBLES], host_lens_D[4];
    static bool host_lens_valid = false, host_rp_valid = false;
    if (!host_lens_valid || memcmp(host_lens_D, D, sizeof host_lens_D) != 0) {
        build_lens_table(D, host_lens);
        memcpy(host_lens_D, D, sizeof host_lens_D);
        host_lens_valid = true;
    }
    if (!host_rp_valid) { build_rp_table(host_rp); host_rp_valid = true; }
    rc = scratch_put(S->lens, host_lens, LENS_DOUBLES, (hipStream_t)stream); if (rc) return rc;
    rc = scratch_put(S->rp, host_rp, RP_TAB_DOUBLES, (hipStream_t)stream); if (rc) return rc;
    rc = scratch_put(&S->xf[0][0], (const double*)xf_host, n_frames*13, (hipStream_t)stream); if (rc) return rc;
    const FusedEnv env{g_k1_mode == GF_KERNEL_DIRECT ? GF_KERNEL_DIRECT : GF_KERNEL_RP, nd.n_cu, S->rp};
    return launch_fused(&env, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, S->lens, lk, (const gf_frame_xform*)&S->xf[0][0], n_frames,
                        border, bv, (hipStream_t)stream);
}

GF_API int gf_undistort_rotate_blend_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                                 void* dst, int dw, int dh, size_t dstep, size_t dframe,
                                 const double D[4], const gf_frame_xform* xf_dev, int n_out, int frames_per_group, int n_blend,
                                 int border, const double bv[4], gf_stream_t stream)
{
    if (n_blend < 1 || frames_per_group < n_blend) return fail(GF_ERR_INVALID_ARG, "%s: need 1 <= n_blend <= frames_per_group (n_blend=%lld)", "gf_undistort_rotate_blend_c3", n_blend);
    if (border == GF_BORDER_TRANSPARENT) return fail(GF_ERR_UNSUPPORTED, "%s: BORDER_TRANSPARENT depends on the previous output frame; use the unfused calls", "gf_undistort_rotate_blend_c3");
    const long long n_in = (long long)(n_out - 1)*frames_per_group + n_blend;
    int rc = fused_checks("gf_undistort_rotate_blend_c3", ctx, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, xf_dev, n_in, n_out, border);
    if (rc) return rc;
    Ctx* c = (Ctx*)ctx;
    DeviceGuard dg(c->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "gf_undistort_rotate_blend_c3");
    const double* lens; LensK lk;
    LensLease lease(c, (hipStream_t)stream);
    rc = lease.acquire(D, &lens, &lk); if (rc) return rc;
    RemapArgs A = make_args(depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, bv);
    if (c->k1_mode != GF_KERNEL_DIRECT && n_blend <= RB_MAX) {      // RP coordinates, as the fused kernel of this mode
        A.tiles_x = (dw + RT - 1)/RT; A.tiles_y = (dh + RT - 1)/RT;
        dim3 grid(A.tiles_x*A.tiles_y, n_out);
#define LAUNCH_BRP2(DEPTH_, BORDER_, RB_, P2_) hipLaunchKernelGGL((k_undistort_rotate_blend_rp<DEPTH_, BORDER_, RB_, P2_>), grid, dim3(BLOCK), 0, (hipStream_t)stream, A, (const Xf*)xf_dev, lens, lk, (const double*)c->rp_tab, frames_per_group, n_blend)
#define LAUNCH_BRP(DEPTH_, BORDER_) do { const bool p2 = (n_blend & (n_blend - 1)) == 0; \
                                          if (n_blend <= 4) { if (p2) LAUNCH_BRP2(DEPTH_, BORDER_, 4, true); else LAUNCH_BRP2(DEPTH_, BORDER_, 4, false); } \
                                          else              { if (p2) LAUNCH_BRP2(DEPTH_, BORDER_, RB_MAX, true); else LAUNCH_BRP2(DEPTH_, BORDER_, RB_MAX, false); } } while (0)
        if (depth == GF_DEPTH_U8) { if (border == GF_BORDER_CONSTANT) LAUNCH_BRP(GF_DEPTH_U8, GF_BORDER_CONSTANT); else LAUNCH_BRP(GF_DEPTH_U8, GF_BORDER_REPLICATE); }
        else                      { if (border == GF_BORDER_CONSTANT) LAUNCH_BRP(GF_DEPTH_U16, GF_BORDER_CONSTANT); else LAUNCH_BRP(GF_DEPTH_U16, GF_BORDER_REPLICATE); }
#undef LAUNCH_BRP
#undef LAUNCH_BRP2
        HIP_TRY(hipGetLastError());
        return GF_OK;
    }
    dim3 grid(A.tiles_x*A.tiles_y, n_out);
#define LAUNCH_BL(DEPTH_, BORDER_) do { if ((n_blend & (n_blend - 1)) == 0) hipLaunchKernelGGL((k_undistort_rotate_blend<DEPTH_, BORDER_, true>), grid, dim3(BLOCK), 0, (hipStream_t)stream, A, (const Xf*)xf_dev, lens, lk, frames_per_group, n_blend); \
                                         else hipLaunchKernelGGL((k_undistort_rotate_blend<DEPTH_, BORDER_, false>), grid, dim3(BLOCK), 0, (hipStream_t)stream, A, (const Xf*)xf_dev, lens, lk, frames_per_group, n_blend); } while (0)
    if (depth == GF_DEPTH_U8) { if (border == GF_BORDER_CONSTANT) LAUNCH_BL(GF_DEPTH_U8, GF_BORDER_CONSTANT); else LAUNCH_BL(GF_DEPTH_U8, GF_BORDER_REPLICATE); }
    else                      { if (border == GF_BORDER_CONSTANT) LAUNCH_BL(GF_DEPTH_U16, GF_BORDER_CONSTANT); else LAUNCH_BL(GF_DEPTH_U16, GF_BORDER_REPLICATE); }
#undef LAUNCH_BL
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_blend_accumulate(void* ctx, int depth, const void* frame, double* acc, double weight, int reset, size_t n, gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_blend_accumulate");
    if (depth != GF_DEPTH_U8 && depth != GF_DEPTH_U16) return fail(GF_ERR_UNSUPPORTED, "depth %s%lld not supported", "", depth);
    if (!frame || !acc || n == 0) return fail(GF_ERR_INVALID_ARG, "%s: null pointer or n == 0", "gf_blend_accumulate");
    if (((uintptr_t)acc & 7) || (depth == GF_DEPTH_U16 && ((uintptr_t)frame & 1))) return fail(GF_ERR_INVALID_ARG, "%s: misaligned buffer", "gf_blend_accumulate");
    DeviceGuard dg(((Ctx*)ctx)->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "libgfremap");
    const unsigned blocks = (unsigned)((n + BLOCK - 1)/BLOCK < 8192 ? (n + BLOCK - 1)/BLOCK : 8192);
    if (depth == GF_DEPTH_U8) hipLaunchKernelGGL(k_blend_acc<uint8_t>, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, (const uint8_t*)frame, acc, weight, reset, n);
    else                      hipLaunchKernelGGL(k_blend_acc<uint16_t>, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, (const uint16_t*)frame, acc, weight, reset, n);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_blend_finalize(void* ctx, int depth, const double* acc, void* out, size_t n, gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_blend_finalize");
    if (depth != GF_DEPTH_U8 && depth != GF_DEPTH_U16) return fail(GF_ERR_UNSUPPORTED, "depth %s%lld not supported", "", depth);
    if (!acc || !out || n == 0) return fail(GF_ERR_INVALID_ARG, "%s: null pointer or n == 0", "gf_blend_finalize");
    if (((uintptr_t)acc & 7) || (depth == GF_DEPTH_U16 && ((uintptr_t)out & 1))) return fail(GF_ERR_INVALID_ARG, "%s: misaligned buffer", "gf_blend_finalize");
    DeviceGuard dg(((Ctx*)ctx)->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "libgfremap");
    const unsigned blocks = (unsigned)((n + BLOCK - 1)/BLOCK < 8192 ? (n + BLOCK - 1)/BLOCK : 8192);
    if (depth == GF_DEPTH_U8) hipLaunchKernelGGL(k_blend_fin<uint8_t>, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, acc, (uint8_t*)out, n);
    else                      hipLaunchKernelGGL(k_blend_fin<uint16_t>, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, acc, (uint16_t*)out, n);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_fisheye_undistort_points(void* ctx, const double* pts, size_t n_pts, const double K_host[9], const double D_host[4],
                                       const double* PR, size_t n_sets, double* out, gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_fisheye_undistort_points");
    if (!pts || !K_host || !D_host || !PR || !out) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_fisheye_undistort_points");
    if (n_pts == 0 || n_sets == 0 || n_pts > (1u << 24) || n_sets > ((size_t)1 << 32))
        return fail(GF_ERR_INVALID_ARG, "gf_fisheye_undistort_points: bad counts %s(n_pts=%lld)", "", (long long)n_pts);
    if (((uintptr_t)pts | (uintptr_t)PR | (uintptr_t)out) & 7) return fail(GF_ERR_INVALID_ARG, "%s: misaligned buffer", "gf_fisheye_undistort_points");
    PtsK P;
    P.f[0] = K_host[0]; P.f[1] = K_host[4]; P.c[0] = K_host[2]; P.c[1] = K_host[5];
    for (int i = 0; i < 4; i++) P.k[i] = D_host[i];
    if (P.f[0] == 0 || P.f[1] == 0) return fail(GF_ERR_SINGULAR, "%s: zero focal length", "gf_fisheye_undistort_points");
    DeviceGuard dg(((Ctx*)ctx)->device);
    if (!dg.ok) return fail(GF_ERR_HIP, "%s: cannot select the context's device", "libgfremap");
    const size_t total = n_pts*n_sets;
    const size_t blocks = (total + BLOCK - 1)/BLOCK;
    if (blocks > 0x7fffffffull) return fail(GF_ERR_UNSUPPORTED, "%s: too many points", "gf_fisheye_undistort_points");
    hipLaunchKernelGGL(k_undistort_points, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)stream, pts, (unsigned)n_pts, P, PR, total, out);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

}  // extern "C"
