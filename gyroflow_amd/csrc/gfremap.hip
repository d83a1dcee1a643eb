// gfremap.hip -- HIP (gfx950 / CDNA4) kernels + C ABI of libgfremap.so.  See include/gfremap.h.
//
// Path replaced (reference = ElvinC/gyroflow, Python):
//   calibrate_video.py:330-384  FisheyeCalibrator.get_maps  -> cv2.fisheye.initUndistortRectifyMap
//   stabilizer.py:1423-1447     render-loop body            -> cv2.remap(INTER_LINEAR) + float64 blend
//   nonlinear_stretch.py:130    apply_stretch               -> cv2.remap(INTER_CUBIC)
//
// Layout: one thread produces PXT=4 horizontally adjacent output pixels (12 B of u8 BGR = one
// dwordx3 store; a wave64 writes two 384-B row segments = whole 128-B lines), a 256-thread
// workgroup covers a 128x8 output tile, blockIdx.z is the frame.  The fixed-point map
// (INTER_BITS=5) lives in registers only.  Source taps are gathered through L1/L2 with unaligned
// dword loads (6 useful bytes per tap pair); workgroups are dealt to XCDs in contiguous bands of
// tile rows so that each XCD's L2 sees one horizontal band of the source frame.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <mutex>

#include "../../include/gfremap.h"

namespace {

constexpr int INTER_BITS = 5;
constexpr int INTER_TAB_SIZE = 32;
constexpr int PXT = 4;            // pixels per thread (horizontal)
constexpr int TILE_W = 128;       // output tile width  (32 threads x 4 px)
constexpr int TILE_H = 8;         // output tile height
constexpr int BLOCK = 256;
constexpr int NXCD = 8;

thread_local char g_err[512] = "";

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

struct Ctx {
    int device;
    int16_t* cubic_tab;   // device, 1024*16 int16 (cv2 INTER_CUBIC Q15 table)
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
// Device helpers

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
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
    double f[2], c[2], k[4];
};

// Fixed-point source coordinate of output pixel (j, i): the per-pixel body of
// cv::fisheye::initUndistortRectifyMap with m1type == CV_16SC2.  rx/ry/rw are the row terms
// i*iR[.1] + iR[.2] (same two roundings as the reference loop's row start); the column term
// uses one fma instead of j sequential additions (differs by < 1e-12 px).

__device__ __forceinline__ void fisheye_uv(const Xf& xf, double rx, double ry, double rw, int j, double& u, double& v)
{
    const double dj = (double)j;
    const double _x = __fma_rn(dj, xf.iR[0], rx);
    const double _y = __fma_rn(dj, xf.iR[3], ry);
    const double _w = __fma_rn(dj, xf.iR[6], rw);
    if (_w <= 0) {
        u = (_x > 0) ? -INFINITY : INFINITY;
        v = (_y > 0) ? -INFINITY : INFINITY;
        return;
    }
    const double x = _x/_w, y = _y/_w;
    const double r = sqrt(x*x + y*y);
    const double theta = atan(r);
    const double t2 = theta*theta, t4 = t2*t2, t6 = t4*t2, t8 = t4*t4;
    const double theta_d = theta*(1 + xf.k[0]*t2 + xf.k[1]*t4 + xf.k[2]*t6 + xf.k[3]*t8);
    const double scale = (r == 0) ? 1.0 : theta_d/r;
    u = xf.f[0]*x*scale + xf.c[0];
    v = xf.f[1]*y*scale + xf.c[1];
}

__device__ __forceinline__ void row_terms(const Xf& xf, int i, double& rx, double& ry, double& rw)
{
    const double di = (double)i;
    rx = __dadd_rn(__dmul_rn(di, xf.iR[1]), xf.iR[2]);
    ry = __dadd_rn(__dmul_rn(di, xf.iR[4]), xf.iR[5]);
    rw = __dadd_rn(__dmul_rn(di, xf.iR[7]), xf.iR[8]);
}

// (sx, sy, a) exactly as the CV_16SC2/CV_16UC1 pair would carry them: wrapping (short) cast of the
// integer part, 10-bit fraction index a = fy*32 + fx.
__device__ __forceinline__ void q5_from_uv(double u, double v, int& sx, int& sy, int& a)
{
    const int iu = cv_round_d(u*INTER_TAB_SIZE);
    const int iv = cv_round_d(v*INTER_TAB_SIZE);
    sx = (int)(int16_t)(iu >> INTER_BITS);
    sy = (int)(int16_t)(iv >> INTER_BITS);
    a = (iv & (INTER_TAB_SIZE-1))*INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE-1));
}

struct SrcImg {
    const uint8_t* base;   // frame base (bytes)
    int w, h;
    long step;             // bytes
};

template <int DEPTH> struct PixT;
template <> struct PixT<GF_DEPTH_U8>  { typedef uint8_t  T; };
template <> struct PixT<GF_DEPTH_U16> { typedef uint16_t T; };

// One bilinear output pixel, cv::remapBilinear semantics (imgwarp.cpp).  Returns false when the
// pixel must be left untouched (BORDER_TRANSPARENT outside).  out[3] receives the channel values.
template <int DEPTH, int BORDER>
__device__ __forceinline__ bool bilinear_px(const SrcImg& s, int sx, int sy, int a, const int* cval, int* out)
{
    typedef typename PixT<DEPTH>::T T;
    const int fx = a & 31, fy = a >> 5;
    int p[4][3];                                    // taps: 00, 01, 10, 11
    const bool inlier = (unsigned)sx < (unsigned)(s.w - 1) && (unsigned)sy < (unsigned)(s.h - 1);
    if (inlier) {
        const uint8_t* q = s.base + (long)sy*s.step + (long)sx*(3*(int)sizeof(T));
        if (DEPTH == GF_DEPTH_U8) {
            // 6 bytes per row pair: two overlapping unaligned dwords (bytes 0..3 and 2..5) -- never
            // reads past the last byte of the frame.
            const uint32_t a0 = *(const u32_unaligned*)q,            a1 = *(const u32_unaligned*)(q + 2);
            const uint32_t b0 = *(const u32_unaligned*)(q + s.step), b1 = *(const u32_unaligned*)(q + s.step + 2);
            p[0][0] = a0 & 0xff; p[0][1] = (a0 >> 8) & 0xff;  p[0][2] = (a0 >> 16) & 0xff;
            p[1][0] = a0 >> 24;  p[1][1] = (a1 >> 16) & 0xff; p[1][2] = a1 >> 24;
            p[2][0] = b0 & 0xff; p[2][1] = (b0 >> 8) & 0xff;  p[2][2] = (b0 >> 16) & 0xff;
            p[3][0] = b0 >> 24;  p[3][1] = (b1 >> 16) & 0xff; p[3][2] = b1 >> 24;
        } else {
            U16x6 r0, r1;                                              // 12 bytes per row pair
            __builtin_memcpy(&r0, q, 12);
            __builtin_memcpy(&r1, q + s.step, 12);
            p[0][0] = r0.a & 0xffff; p[0][1] = r0.a >> 16; p[0][2] = r0.b & 0xffff;
            p[1][0] = r0.b >> 16;    p[1][1] = r0.c & 0xffff; p[1][2] = r0.c >> 16;
            p[2][0] = r1.a & 0xffff; p[2][1] = r1.a >> 16; p[2][2] = r1.b & 0xffff;
            p[3][0] = r1.b >> 16;    p[3][1] = r1.c & 0xffff; p[3][2] = r1.c >> 16;
        }
    } else {
        int x0 = sx, x1 = sx + 1, y0 = sy, y1 = sy + 1;
        bool vx0, vx1, vy0, vy1;
        if (BORDER == GF_BORDER_CONSTANT) {
            if (sx >= s.w || sx + 1 < 0 || sy >= s.h || sy + 1 < 0) {
                out[0] = cval[0]; out[1] = cval[1]; out[2] = cval[2];
                return true;
            }
            vx0 = (unsigned)x0 < (unsigned)s.w; vx1 = (unsigned)x1 < (unsigned)s.w;
            vy0 = (unsigned)y0 < (unsigned)s.h; vy1 = (unsigned)y1 < (unsigned)s.h;
        } else if (BORDER == GF_BORDER_TRANSPARENT) {
            return false;                           // (unsigned)sx >= w-1 || (unsigned)sy >= h-1
        } else {                                    // REPLICATE: clamp
            x0 = min(max(x0, 0), s.w - 1); x1 = min(max(x1, 0), s.w - 1);
            y0 = min(max(y0, 0), s.h - 1); y1 = min(max(y1, 0), s.h - 1);
            vx0 = vx1 = vy0 = vy1 = true;
        }
        const int xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
        const bool vs[4] = {vx0 && vy0, vx1 && vy0, vx0 && vy1, vx1 && vy1};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (vs[t]) {
                const T* q = (const T*)(s.base + (long)ys[t]*s.step) + xs[t]*3;
                p[t][0] = q[0]; p[t][1] = q[1]; p[t][2] = q[2];
            } else {
                p[t][0] = cval[0]; p[t][1] = cval[1]; p[t][2] = cval[2];
            }
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
    return true;
}

// Store PXT pixels (3 channels each) of one thread.  `keep` bit i set = pixel i untouched.
template <int DEPTH>
__device__ __forceinline__ void store_px(uint8_t* drow, int x0, int dw, const int (*px)[3], unsigned keep, bool vec_ok)
{
    typedef typename PixT<DEPTH>::T T;
    if (vec_ok && keep == 0 && x0 + PXT <= dw) {
        if (DEPTH == GF_DEPTH_U8) {
            uint32_t w0 = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
            uint32_t w1 = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
            uint32_t w2 = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
            uint32_t* d = (uint32_t*)(drow + (long)x0*3);
            struct alignas(4) W3 { uint32_t a, b, c; };
            *(W3*)d = W3{w0, w1, w2};
        } else {
            uint32_t w[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int e = i*2;          // element index in the 12-element run
                w[i] = (uint32_t)px[e/3][e%3] | ((uint32_t)px[(e+1)/3][(e+1)%3] << 16);
            }
            uint32_t* d = (uint32_t*)(drow + (long)x0*6);
            struct alignas(8) W2 { uint32_t a, b; };
            *(W2*)(d) = W2{w[0], w[1]}; *(W2*)(d + 2) = W2{w[2], w[3]}; *(W2*)(d + 4) = W2{w[4], w[5]};
        }
        return;
    }
    T* d = (T*)drow + (long)x0*3;
#pragma unroll
    for (int i = 0; i < PXT; i++)
        if (x0 + i < dw && !((keep >> i) & 1)) {
            d[i*3+0] = (T)px[i][0]; d[i*3+1] = (T)px[i][1]; d[i*3+2] = (T)px[i][2];
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
    const uint8_t* src; int sw, sh; long sstep, sframe;
    uint8_t* dst; int dw, dh; long dstep, dframe;
    int tiles_x, tiles_y;
    int cval[3];
    int dst_vec_ok;        // dst base/step/frame stride allow 4-byte (u8) / 8-byte (u16) vector stores
};

// K1: fused fisheye map + rotation + bilinear remap, batched over frames (blockIdx.y).
template <int DEPTH, int BORDER>
__global__ void __launch_bounds__(BLOCK) k_undistort_rotate(RemapArgs A, const Xf* __restrict__ xfs)
{
    int tx, ty;
    tile_of_block(blockIdx.x, A.tiles_x, A.tiles_y, tx, ty);
    const int frame = blockIdx.y;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x0 = tx*TILE_W + lx*PXT, y = ty*TILE_H + ly;
    if (y >= A.dh || x0 >= A.dw) return;
    const Xf& xf = xfs[frame];
    SrcImg s{A.src + (long)frame*A.sframe, A.sw, A.sh, A.sstep};
    double rx, ry, rw;
    row_terms(xf, y, rx, ry, rw);
    int px[PXT][3];
    unsigned keep = 0;
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        double u, v; int sx, sy, a;
        fisheye_uv(xf, rx, ry, rw, x0 + i, u, v);
        q5_from_uv(u, v, sx, sy, a);
        if (!bilinear_px<DEPTH, BORDER>(s, sx, sy, a, A.cval, px[i])) keep |= 1u << i;
    }
    uint8_t* drow = A.dst + (long)frame*A.dframe + (long)y*A.dstep;
    store_px<DEPTH>(drow, x0, A.dw, px, keep, A.dst_vec_ok);
}

// K1b: fused + hyperlapse blend: out = trunc(sum_b (1/N) * remap_b) in float64, reference order.
template <int DEPTH, int BORDER>
__global__ void __launch_bounds__(BLOCK) k_undistort_rotate_blend(RemapArgs A, const Xf* __restrict__ xfs,
                                                                   int frames_per_group, int n_blend)
{
    int tx, ty;
    tile_of_block(blockIdx.x, A.tiles_x, A.tiles_y, tx, ty);
    const int g = blockIdx.y;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x0 = tx*TILE_W + lx*PXT, y = ty*TILE_H + ly;
    if (y >= A.dh || x0 >= A.dw) return;
    const double wgt = 1.0/(double)n_blend;       // Python: 1/(self.hyperlapse_num_blended_frames)
    double acc[PXT][3];
#pragma unroll
    for (int i = 0; i < PXT; i++) acc[i][0] = acc[i][1] = acc[i][2] = 0.0;
    for (int b = 0; b < n_blend; b++) {
        const int frame = g*frames_per_group + b;
        const Xf& xf = xfs[frame];
        SrcImg s{A.src + (long)frame*A.sframe, A.sw, A.sh, A.sstep};
        double rx, ry, rw;
        row_terms(xf, y, rx, ry, rw);
#pragma unroll
        for (int i = 0; i < PXT; i++) {
            double u, v; int sx, sy, a; int o[3];
            fisheye_uv(xf, rx, ry, rw, x0 + i, u, v);
            q5_from_uv(u, v, sx, sy, a);
            bilinear_px<DEPTH, BORDER>(s, sx, sy, a, A.cval, o);
#pragma unroll
            for (int c = 0; c < 3; c++) acc[i][c] = __dadd_rn(acc[i][c], __dmul_rn(wgt, (double)o[c]));
        }
    }
    int px[PXT][3];
#pragma unroll
    for (int i = 0; i < PXT; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) px[i][c] = (int)acc[i][c];      // astype(uint): truncation
    uint8_t* drow = A.dst + (long)g*A.dframe + (long)y*A.dstep;
    store_px<DEPTH>(drow, x0, A.dw, px, 0u, A.dst_vec_ok);
}

// K4: remap from materialised CV_16SC2 + CV_16UC1 maps.
template <int DEPTH, int BORDER>
__global__ void __launch_bounds__(BLOCK) k_remap_linear(RemapArgs A, const uint8_t* __restrict__ map1, long m1step,
                                                         const uint8_t* __restrict__ map2, long m2step)
{
    int tx, ty;
    tile_of_block(blockIdx.x, A.tiles_x, A.tiles_y, tx, ty);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x0 = tx*TILE_W + lx*PXT, y = ty*TILE_H + ly;
    if (y >= A.dh || x0 >= A.dw) return;
    SrcImg s{A.src, A.sw, A.sh, A.sstep};
    const int16_t* m1 = (const int16_t*)(map1 + (long)y*m1step);
    const uint16_t* m2 = (const uint16_t*)(map2 + (long)y*m2step);
    int px[PXT][3];
    unsigned keep = 0;
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        const int x = min(x0 + i, A.dw - 1);
        const int sx = m1[x*2], sy = m1[x*2+1], a = m2[x] & 1023;
        if (!bilinear_px<DEPTH, BORDER>(s, sx, sy, a, A.cval, px[i])) keep |= 1u << i;
    }
    uint8_t* drow = A.dst + (long)y*A.dstep;
    store_px<DEPTH>(drow, x0, A.dw, px, keep, A.dst_vec_ok);
}

// K2: materialised maps.  One thread per output pixel, 64x4 tiles.
template <bool FIXED>
__global__ void __launch_bounds__(BLOCK) k_fisheye_maps(Xf xf, int w, int h, uint8_t* map1, long m1step, uint8_t* map2, long m2step)
{
    const int j = blockIdx.x*64 + (threadIdx.x & 63), i = blockIdx.y*4 + (threadIdx.x >> 6);
    if (j >= w || i >= h) return;
    double rx, ry, rw, u, v;
    row_terms(xf, i, rx, ry, rw);
    fisheye_uv(xf, rx, ry, rw, j, u, v);
    if (FIXED) {
        const int iu = cv_round_d(u*INTER_TAB_SIZE), iv = cv_round_d(v*INTER_TAB_SIZE);
        int16_t* m1 = (int16_t*)(map1 + (long)i*m1step);
        uint16_t* m2 = (uint16_t*)(map2 + (long)i*m2step);
        // one 4-byte store for the (x,y) pair
        *(uint32_t*)(m1 + 2*j) = (uint32_t)(uint16_t)(iu >> INTER_BITS) | ((uint32_t)(uint16_t)(iv >> INTER_BITS) << 16);
        m2[j] = (uint16_t)((iv & 31)*32 + (iu & 31));
    } else {
        ((float*)(map1 + (long)i*m1step))[j] = (float)u;
        ((float*)(map2 + (long)i*m2step))[j] = (float)v;
    }
}

// K3: convertMaps(32F,32F -> 16SC2) with the reference's float32 pre-scaling.
__global__ void __launch_bounds__(BLOCK) k_convert_maps(const uint8_t* mx, long mxstep, const uint8_t* my, long mystep,
                                                        float sxn, float sxd, float syn, float syd, int scale_on,
                                                        int w, int h, uint8_t* map1, long m1step, uint8_t* map2, long m2step)
{
    const int j = blockIdx.x*64 + (threadIdx.x & 63), i = blockIdx.y*4 + (threadIdx.x >> 6);
    if (j >= w || i >= h) return;
    float fx = ((const float*)(mx + (long)i*mxstep))[j], fy = ((const float*)(my + (long)i*mystep))[j];
    if (scale_on) {
        fx = __fdiv_rn(__fmul_rn(fx, sxn), sxd);
        fy = __fdiv_rn(__fmul_rn(fy, syn), syd);
    }
    const int ix = cv_round_f(__fmul_rn(fx, 32.f)), iy = cv_round_f(__fmul_rn(fy, 32.f));
    int16_t* m1 = (int16_t*)(map1 + (long)i*m1step);
    *(uint32_t*)(m1 + 2*j) = (uint32_t)(uint16_t)sat_s16(ix >> INTER_BITS) | ((uint32_t)(uint16_t)sat_s16(iy >> INTER_BITS) << 16);
    ((uint16_t*)(map2 + (long)i*m2step))[j] = (uint16_t)((iy & 31)*32 + (ix & 31));
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

__global__ void __launch_bounds__(BLOCK) k_remap_cubic_u8c3(const uint8_t* __restrict__ src, int sw, int sh, long sstep,
                                                             uint8_t* dst, int dw, int dh, long dstep,
                                                             const uint8_t* __restrict__ mx, long mxstep,
                                                             const uint8_t* __restrict__ my, long mystep,
                                                             const int16_t* __restrict__ tab, int border, int c0, int c1, int c2)
{
    const int dx = blockIdx.x*64 + (threadIdx.x & 63), dy = blockIdx.y*4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const float fxm = ((const float*)(mx + (long)dy*mxstep))[dx], fym = ((const float*)(my + (long)dy*mystep))[dx];
    const int isx = cv_round_f(__fmul_rn(fxm, 32.f)), isy = cv_round_f(__fmul_rn(fym, 32.f));
    const int a = (isy & 31)*32 + (isx & 31);
    const int sx = sat_s16(isx >> INTER_BITS) - 1, sy = sat_s16(isy >> INTER_BITS) - 1;
    const int16_t* w = tab + a*16;
    uint8_t* D = dst + (long)dy*dstep + dx*3;
    const int cval[3] = {c0, c1, c2};
    const unsigned width1 = max(sw - 3, 0), height1 = max(sh - 3, 0);
    if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        const uint8_t* S = src + (long)sy*sstep + sx*3;
        int sum[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // 12 bytes = 4 pixels x 3 channels: three overlapping-free unaligned dwords, all inside the row
            const uint8_t* q = S + (long)r*sstep;
            const uint32_t d0 = *(const u32_unaligned*)q, d1 = *(const u32_unaligned*)(q + 4), d2 = *(const u32_unaligned*)(q + 8);
            const int b[12] = {(int)(d0 & 0xff), (int)((d0 >> 8) & 0xff), (int)((d0 >> 16) & 0xff), (int)(d0 >> 24),
                               (int)(d1 & 0xff), (int)((d1 >> 8) & 0xff), (int)((d1 >> 16) & 0xff), (int)(d1 >> 24),
                               (int)(d2 & 0xff), (int)((d2 >> 8) & 0xff), (int)((d2 >> 16) & 0xff), (int)(d2 >> 24)};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int wt = w[r*4 + t];
                sum[0] += b[t*3]*wt; sum[1] += b[t*3+1]*wt; sum[2] += b[t*3+2]*wt;
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
            const uint8_t* S = src + (long)ys[i]*sstep + k;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (xs[q] >= 0) sum += ((int)S[xs[q]*3] - cv)*w[i*4 + q];
        }
        int v = (sum + (1 << 14)) >> 15;
        D[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
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

// ---------------------------------------------------------------------------------------------
// Host-side argument checking / launch helpers

int check_img(const char* what, const void* p, int w, int h, size_t step, int depth)
{
    const size_t esz = depth == GF_DEPTH_U8 ? 1 : 2;
    if (!p) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", what);
    if (w <= 0 || h <= 0 || w >= SHRT_MAX || h >= SHRT_MAX) return fail(GF_ERR_INVALID_ARG, "%s: bad dimensions (must be 1..32766), w=%lld", what, w);
    if (step < (size_t)w*3*esz) return fail(GF_ERR_INVALID_ARG, "%s: row step %lld smaller than a row", what, (long long)step);
    if (esz == 2 && ((step & 1) || ((uintptr_t)p & 1))) return fail(GF_ERR_INVALID_ARG, "%s: u16 image must be 2-byte aligned", what);
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
    A.src = (const uint8_t*)src; A.sw = sw; A.sh = sh; A.sstep = (long)sstep; A.sframe = (long)sframe;
    A.dst = (uint8_t*)dst; A.dw = dw; A.dh = dh; A.dstep = (long)dstep; A.dframe = (long)dframe;
    A.tiles_x = (dw + TILE_W - 1)/TILE_W; A.tiles_y = (dh + TILE_H - 1)/TILE_H;
    cval_from(bv, depth, A.cval);
    const size_t al = depth == GF_DEPTH_U8 ? 4 : 8;
    A.dst_vec_ok = (((uintptr_t)dst % al) == 0 && (dstep % al) == 0 && (dframe % al) == 0) ? 1 : 0;
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
    memcpy(&r, x, sizeof r);
    return r;
}

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

GF_API int gf_ctx_create(int device, void** ctx)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx pointer", "gf_ctx_create");
    *ctx = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(GF_ERR_INVALID_ARG, "gf_ctx_create: device %s%lld out of range", "", device);
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    std::call_once(g_cubic_once, build_cubic_host);
    Ctx* c = new (std::nothrow) Ctx{device, nullptr};
    if (!c) return fail(GF_ERR_HIP, "%s: out of host memory", "gf_ctx_create");
    hipError_t e = hipMalloc((void**)&c->cubic_tab, sizeof g_cubic_host);
    if (e == hipSuccess) e = hipMemcpy(c->cubic_tab, g_cubic_host, sizeof g_cubic_host, hipMemcpyHostToDevice);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (c->cubic_tab) (void)hipFree(c->cubic_tab);
        delete c;
        return fail(GF_ERR_HIP, "gf_ctx_create: %s (HIP error %lld)", "table upload failed", (long long)e);
    }
    *ctx = c;
    return GF_OK;
}

GF_API int gf_ctx_destroy(void* ctx)
{
    if (!ctx) return GF_OK;
    Ctx* c = (Ctx*)ctx;
    if (c->cubic_tab) (void)hipFree(c->cubic_tab);
    delete c;
    return GF_OK;
}

GF_API int gf_make_frame_xform(const double K[9], const double D[4], const double R[9], const double P[9], gf_frame_xform* out)
{
    if (!K || !D || !R || !P || !out) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_make_frame_xform");
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
    for (int i = 0; i < 4; i++) out->k[i] = D[i];
    return GF_OK;
}

static int maps_common(void* ctx, const gf_frame_xform* xf, int w, int h, void* m1, size_t s1, void* m2, size_t s2,
                       bool fixed, gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_fisheye_maps");
    if (!xf || !m1 || !m2) return fail(GF_ERR_INVALID_ARG, "%s: null pointer", "gf_fisheye_maps");
    if (w <= 0 || h <= 0 || w >= SHRT_MAX || h >= SHRT_MAX) return fail(GF_ERR_INVALID_ARG, "%s: bad size, w=%lld", "gf_fisheye_maps", w);
    const size_t min1 = (size_t)w*4, min2 = fixed ? (size_t)w*2 : (size_t)w*4;
    if (s1 < min1 || s2 < min2 || (s1 & 3) || (s2 & (fixed ? 1 : 3)) || ((uintptr_t)m1 & 3) || ((uintptr_t)m2 & (fixed ? 1 : 3)))
        return fail(GF_ERR_INVALID_ARG, "%s: map step/alignment invalid (step1=%lld)", "gf_fisheye_maps", (long long)s1);
    dim3 grid((w + 63)/64, (h + 3)/4);
    if (fixed) hipLaunchKernelGGL(k_fisheye_maps<true>, grid, dim3(BLOCK), 0, (hipStream_t)stream, to_xf(xf), w, h, (uint8_t*)m1, (long)s1, (uint8_t*)m2, (long)s2);
    else       hipLaunchKernelGGL(k_fisheye_maps<false>, grid, dim3(BLOCK), 0, (hipStream_t)stream, to_xf(xf), w, h, (uint8_t*)m1, (long)s1, (uint8_t*)m2, (long)s2);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_fisheye_maps_16sc2(void* ctx, const gf_frame_xform* xf, int w, int h, int16_t* map1, size_t s1, uint16_t* map2, size_t s2, gf_stream_t stream)
{
    return maps_common(ctx, xf, w, h, map1, s1, map2, s2, true, stream);
}

GF_API int gf_fisheye_maps_32f(void* ctx, const gf_frame_xform* xf, int w, int h, float* mx, size_t sx, float* my, size_t sy, gf_stream_t stream)
{
    return maps_common(ctx, xf, w, h, mx, sx, my, sy, false, stream);
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
    dim3 grid((w + 63)/64, (h + 3)/4);
    hipLaunchKernelGGL(k_convert_maps, grid, dim3(BLOCK), 0, (hipStream_t)stream, (const uint8_t*)mx, (long)mxs, (const uint8_t*)my, (long)mys,
                       sxn, sxd, syn, syd, scale_on, w, h, (uint8_t*)map1, (long)s1, (uint8_t*)map2, (long)s2);
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
    RemapArgs A = make_args(depth, src, sw, sh, sstep, 0, dst, dw, dh, dstep, 0, bv);
    dim3 grid(A.tiles_x*A.tiles_y, 1);
    DISPATCH_DEPTH_BORDER(k_remap_linear, depth, border, grid, (hipStream_t)stream, A, (const uint8_t*)map1, (long)s1, (const uint8_t*)map2, (long)s2);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

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
    dim3 grid((dw + 63)/64, (dh + 3)/4);
    hipLaunchKernelGGL(k_remap_cubic_u8c3, grid, dim3(BLOCK), 0, (hipStream_t)stream, src, sw, sh, (long)sstep, dst, dw, dh, (long)dstep,
                       (const uint8_t*)mx, (long)mxs, (const uint8_t*)my, (long)mys, ((Ctx*)ctx)->cubic_tab, border, cval[0], cval[1], cval[2]);
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

GF_API int gf_undistort_rotate_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                           void* dst, int dw, int dh, size_t dstep, size_t dframe,
                           const gf_frame_xform* xf_dev, int n_frames, int border, const double bv[4], gf_stream_t stream)
{
    int rc = fused_checks("gf_undistort_rotate_c3", ctx, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, xf_dev, n_frames, n_frames, border);
    if (rc) return rc;
    RemapArgs A = make_args(depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, bv);
    dim3 grid(A.tiles_x*A.tiles_y, n_frames);
    DISPATCH_DEPTH_BORDER(k_undistort_rotate, depth, border, grid, (hipStream_t)stream, A, (const Xf*)xf_dev);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_undistort_rotate_blend_c3(void* ctx, int depth, const void* src, int sw, int sh, size_t sstep, size_t sframe,
                                 void* dst, int dw, int dh, size_t dstep, size_t dframe,
                                 const gf_frame_xform* xf_dev, int n_out, int frames_per_group, int n_blend,
                                 int border, const double bv[4], gf_stream_t stream)
{
    if (n_blend < 1 || frames_per_group < n_blend) return fail(GF_ERR_INVALID_ARG, "%s: need 1 <= n_blend <= frames_per_group (n_blend=%lld)", "gf_undistort_rotate_blend_c3", n_blend);
    if (border == GF_BORDER_TRANSPARENT) return fail(GF_ERR_UNSUPPORTED, "%s: BORDER_TRANSPARENT depends on the previous output frame; use the unfused calls", "gf_undistort_rotate_blend_c3");
    const long long n_in = (long long)(n_out - 1)*frames_per_group + n_blend;
    int rc = fused_checks("gf_undistort_rotate_blend_c3", ctx, depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, xf_dev, n_in, n_out, border);
    if (rc) return rc;
    RemapArgs A = make_args(depth, src, sw, sh, sstep, sframe, dst, dw, dh, dstep, dframe, bv);
    dim3 grid(A.tiles_x*A.tiles_y, n_out);
    if (depth == GF_DEPTH_U8) {
        if (border == GF_BORDER_CONSTANT) hipLaunchKernelGGL((k_undistort_rotate_blend<GF_DEPTH_U8, GF_BORDER_CONSTANT>), grid, dim3(BLOCK), 0, (hipStream_t)stream, A, (const Xf*)xf_dev, frames_per_group, n_blend);
        else                              hipLaunchKernelGGL((k_undistort_rotate_blend<GF_DEPTH_U8, GF_BORDER_REPLICATE>), grid, dim3(BLOCK), 0, (hipStream_t)stream, A, (const Xf*)xf_dev, frames_per_group, n_blend);
    } else {
        if (border == GF_BORDER_CONSTANT) hipLaunchKernelGGL((k_undistort_rotate_blend<GF_DEPTH_U16, GF_BORDER_CONSTANT>), grid, dim3(BLOCK), 0, (hipStream_t)stream, A, (const Xf*)xf_dev, frames_per_group, n_blend);
        else                              hipLaunchKernelGGL((k_undistort_rotate_blend<GF_DEPTH_U16, GF_BORDER_REPLICATE>), grid, dim3(BLOCK), 0, (hipStream_t)stream, A, (const Xf*)xf_dev, frames_per_group, n_blend);
    }
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

GF_API int gf_blend_accumulate(void* ctx, int depth, const void* frame, double* acc, double weight, int reset, size_t n, gf_stream_t stream)
{
    if (!ctx) return fail(GF_ERR_INVALID_ARG, "%s: null ctx", "gf_blend_accumulate");
    if (depth != GF_DEPTH_U8 && depth != GF_DEPTH_U16) return fail(GF_ERR_UNSUPPORTED, "depth %s%lld not supported", "", depth);
    if (!frame || !acc || n == 0) return fail(GF_ERR_INVALID_ARG, "%s: null pointer or n == 0", "gf_blend_accumulate");
    if (((uintptr_t)acc & 7) || (depth == GF_DEPTH_U16 && ((uintptr_t)frame & 1))) return fail(GF_ERR_INVALID_ARG, "%s: misaligned buffer", "gf_blend_accumulate");
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
    const unsigned blocks = (unsigned)((n + BLOCK - 1)/BLOCK < 8192 ? (n + BLOCK - 1)/BLOCK : 8192);
    if (depth == GF_DEPTH_U8) hipLaunchKernelGGL(k_blend_fin<uint8_t>, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, acc, (uint8_t*)out, n);
    else                      hipLaunchKernelGGL(k_blend_fin<uint16_t>, dim3(blocks), dim3(BLOCK), 0, (hipStream_t)stream, acc, (uint16_t*)out, n);
    HIP_TRY(hipGetLastError());
    return GF_OK;
}

}  // extern "C"
