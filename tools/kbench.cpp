// tools/kbench.cpp -- standalone timing harness for libgfremap.so's fused kernel (no torch, no Python):
// allocates device frames with hipMalloc, builds per-frame transforms for a wobbling camera, calls the C ABI
// and times the launches with hipEvents on the launch stream.  Used for kernel A/B work; bench.py is the
// contract benchmark.
//
//   kbench [--in WxH] [--out WxH] [--frames N] [--iters K] [--depth 0|2] [--border 0|1|5] [--lib path] [--blend N] [--same-src]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#include "../include/gfremap.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)

typedef int (*ctx_create_t)(int, void**);
typedef int (*make_xf_t)(const double*, const double*, const double*, gf_frame_xform*);
typedef int (*fused_t)(void*, int, const void*, int, int, size_t, size_t, void*, int, int, size_t, size_t,
                       const double*, const gf_frame_xform*, int, int, const double*, gf_stream_t);
typedef int (*fused_blend_t)(void*, int, const void*, int, int, size_t, size_t, void*, int, int, size_t, size_t,
                             const double*, const gf_frame_xform*, int, int, int, int, const double*, gf_stream_t);
typedef const char* (*last_err_t)(void);

static void rot_zyx(double z, double y, double x, double* R)
{
    double cz = cos(z), sz = sin(z), cy = cos(y), sy = sin(y), cx = cos(x), sx = sin(x);
    double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Rz[i*3+k]*Ry[k*3+j]; T[i*3+j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += T[i*3+k]*Rx[k*3+j]; R[i*3+j] = s; }
}

int main(int argc, char** argv)
{
    int iw = 3840, ih = 2160, ow = 3840, oh = 2160, frames = 32, iters = 20, depth = 0, border = 0, blend = 0, same_src = 0;
    double amp = 3.6;   // degrees, typical correction angle of the real trace
    const char* libpath = "gyroflow_amd/libgfremap.so";
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--in") && i + 1 < argc) sscanf(argv[++i], "%dx%d", &iw, &ih);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) sscanf(argv[++i], "%dx%d", &ow, &oh);
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--depth") && i + 1 < argc) depth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--border") && i + 1 < argc) border = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--blend") && i + 1 < argc) blend = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--amp") && i + 1 < argc) amp = atof(argv[++i]);
        else if (!strcmp(argv[i], "--lib") && i + 1 < argc) libpath = argv[++i];
        else if (!strcmp(argv[i], "--same-src")) same_src = 1;      // every frame reads source frame 0 (frame stride 0): what the kernel costs when its source comes from L2 / the memory-side cache
        else { fprintf(stderr, "unknown arg %s\n", argv[i]); return 1; }
    }
    void* h = dlopen(libpath, RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    ctx_create_t ctx_create = (ctx_create_t)dlsym(h, "gf_ctx_create");
    make_xf_t make_xf = (make_xf_t)dlsym(h, "gf_make_frame_xform");
    fused_t fused = (fused_t)dlsym(h, "gf_undistort_rotate_c3");
    fused_blend_t fused_blend = (fused_blend_t)dlsym(h, "gf_undistort_rotate_blend_c3");
    last_err_t last_err = (last_err_t)dlsym(h, "gf_last_error");
    if (!ctx_create || !make_xf || !fused || !last_err) { fprintf(stderr, "missing symbols\n"); return 1; }

    const size_t esz = depth == 0 ? 1 : 2;
    const size_t sstep = (size_t)iw*3*esz, dstep = (size_t)ow*3*esz, sframe = sstep*ih, dframe = dstep*oh;
    void *src, *dst; gf_frame_xform* xf_dev;
    CK(hipMalloc(&src, sframe*frames)); CK(hipMalloc(&dst, dframe*frames)); CK(hipMalloc((void**)&xf_dev, sizeof(gf_frame_xform)*frames));
    {   // pseudo-random source content
        std::vector<uint8_t> hsrc(sframe);
        uint32_t s = 12345;
        for (int f = 0; f < std::min(frames, 8); f++) {
            for (size_t i = 0; i < sframe; i++) { s = s*1664525u + 1013904223u; hsrc[i] = (uint8_t)(s >> 24); }
            CK(hipMemcpy((char*)src + f*sframe, hsrc.data(), sframe, hipMemcpyHostToDevice));
        }
        for (int f = 8; f < frames; f++) CK(hipMemcpy((char*)src + f*sframe, (char*)src + (f % 8)*sframe, sframe, hipMemcpyDeviceToDevice));
    }
    // GoPro Hero7 4K wide-like lens, scaled to the input width
    const double sc = iw/3840.0;
    double K[9] = {1766.05*sc, 0, iw/2.0 + 7.3, 0, 1766.05*sc, ih/2.0 - 4.1, 0, 0, 1};
    double D[4] = {0.05764, -0.00234, 0.02008, -0.01086};
    double P[9] = {1766.05*sc*ow/iw, 0, ow/2.0, 0, 1766.05*sc*ow/iw, oh/2.0, 0, 0, 1};
    std::vector<gf_frame_xform> xf(frames);
    for (int f = 0; f < frames; f++) {
        double R[9];
        const double a = amp*M_PI/180;
        rot_zyx(a*sin(0.7*f + 0.3), a*sin(1.1*f + 1.0), a*cos(0.9*f), R);
        if (make_xf(K, R, P, &xf[f])) { fprintf(stderr, "make_xf: %s\n", last_err()); return 1; }
    }
    CK(hipMemcpy(xf_dev, xf.data(), sizeof(gf_frame_xform)*frames, hipMemcpyHostToDevice));
    void* ctx;
    if (ctx_create(0, &ctx)) { fprintf(stderr, "ctx: %s\n", last_err()); return 1; }
    hipStream_t st; CK(hipStreamCreate(&st));
    double bv[4] = {0, 0, 0, 0};
    const size_t sfr = same_src ? 0 : sframe;
    auto launch = [&]() {
        if (blend > 1) return fused_blend(ctx, depth, src, iw, ih, sstep, sfr, dst, ow, oh, dstep, dframe, D, xf_dev, frames/blend, blend, blend, border, bv, st);
        return fused(ctx, depth, src, iw, ih, sstep, sfr, dst, ow, oh, dstep, dframe, D, xf_dev, frames, border, bv, st);
    };
    for (int i = 0; i < 3; i++) if (launch()) { fprintf(stderr, "launch: %s\n", last_err()); return 1; }
    CK(hipStreamSynchronize(st));
    std::vector<hipEvent_t> e0(iters), e1(iters);
    for (int i = 0; i < iters; i++) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    for (int i = 0; i < iters; i++) { CK(hipEventRecord(e0[i], st)); launch(); CK(hipEventRecord(e1[i], st)); }
    CK(hipStreamSynchronize(st));
    std::vector<float> ms(iters);
    for (int i = 0; i < iters; i++) CK(hipEventElapsedTime(&ms[i], e0[i], e1[i]));
    std::sort(ms.begin(), ms.end());
    const double med = ms[iters/2], mn = ms[0];
    const double bytes = (double)(sframe + dframe)*frames - (blend > 1 ? (double)dframe*(frames - frames/blend) : 0.0);
    // checksum of the first output frame
    std::vector<uint8_t> hd(dframe);
    CK(hipMemcpy(hd.data(), dst, dframe, hipMemcpyDeviceToHost));
    uint64_t cs = 0; for (size_t i = 0; i < dframe; i += 97) cs = cs*31 + hd[i];
    typedef int (*stats_t)(unsigned long long*);
    if (stats_t rd = (stats_t)dlsym(h, "gf_debug_read_lb_stats")) {       // tools build with -DGF_LB -DGF_LB_STATS
        unsigned long long v[4];
        if (rd(v) == 0) printf("lb stats: thread-passes from LDS %llu, from the frame %llu; half tiles with a box %llu, without %llu\n", v[0], v[1], v[2], v[3]);
    }
    typedef int (*trace_t)(unsigned long long*);
    if (trace_t rd = (trace_t)dlsym(h, "gf_debug_read_wg_trace")) {        // tools build with -DGF_TRACE_WG: timeline of the last launch
        std::vector<unsigned long long> t(8192*3);
        if (rd(t.data()) == 0) {
            unsigned long long t0 = ~0ull, t1 = 0; int n = 0;
            for (int i = 0; i < 8192; i++) if (t[i*3+1]) { t0 = std::min(t0, t[i*3]); t1 = std::max(t1, t[i*3+1]); n++; }
            printf("wg trace: %d workgroups, launch span %.1f us (100 MHz clock)\n", n, (t1 - t0)/100.0);
            // busy workgroups over time, 40 bins
            const int NB = 40; std::vector<double> busy(NB, 0.0);
            const double span = (double)(t1 - t0), bw = span/NB;
            for (int i = 0; i < 8192; i++) if (t[i*3+1]) {
                const double a = (double)(t[i*3] - t0), b = (double)(t[i*3+1] - t0);
                for (int j = 0; j < NB; j++) { const double lo = j*bw, hi = lo + bw; busy[j] += std::max(0.0, std::min(b, hi) - std::max(a, lo))/bw; }
            }
            printf("resident workgroups per bin of %.1f us:", bw/100.0);
            for (int j = 0; j < NB; j++) printf(" %.0f", busy[j]);
            printf("\n");
            // lifetime statistics
            std::vector<double> life; for (int i = 0; i < 8192; i++) if (t[i*3+1]) life.push_back((t[i*3+1] - t[i*3])/100.0);
            std::sort(life.begin(), life.end());
            printf("workgroup lifetime us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f\n", life[0], life[life.size()/10], life[life.size()/2], life[life.size()*9/10], life.back());
            // per compute unit (XCC, HW_ID bits 8..15 = CU, SH, SE): mean lifetime of the workgroups it ran -- systematic or random?
            {
                std::vector<double> sum(8*256, 0.0); std::vector<int> cnt(8*256, 0);
                for (int i = 0; i < 8192; i++) if (t[i*3+1]) {
                    const int key = (int)((t[i*3+2] >> 32) & 7)*256 + (int)((t[i*3+2] >> 8) & 255);
                    sum[key] += (t[i*3+1] - t[i*3])/100.0; cnt[key]++;
                }
                std::vector<double> m; for (int q = 0; q < 8*256; q++) if (cnt[q]) m.push_back(sum[q]/cnt[q]);
                std::sort(m.begin(), m.end());
                if (!m.empty()) printf("per-CU mean lifetime us over %zu CUs: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f\n", m.size(), m[0], m[m.size()/10], m[m.size()/2], m[m.size()*9/10], m.back());
            }
            // first start and last end per XCC
            for (int x = 0; x < 8; x++) {
                unsigned long long a = ~0ull, b = 0, lastst = 0; int c = 0;
                for (int i = 0; i < 8192; i++) if (t[i*3+1] && (int)(t[i*3+2] >> 32 & 15) == x) { a = std::min(a, t[i*3]); b = std::max(b, t[i*3+1]); lastst = std::max(lastst, t[i*3]); c++; }
                if (c) printf("  xcc %d: %d wgs, first start %.1f us, last start %.1f us, last end %.1f us\n", x, c, (a - t0)/100.0, (lastst - t0)/100.0, (b - t0)/100.0);
            }
        }
    }
    if (same_src) printf("[same source frame] ");
    printf("%dx%d->%dx%d depth=%d border=%d frames=%d blend=%d: median %.3f ms (min %.3f) = %.2f us/frame, %.0f frames/s, %.0f GB/s algorithmic (%.1f%% of 8 TB/s) checksum %016llx\n",
           iw, ih, ow, oh, depth, border, frames, blend, med, mn, 1e3*med/frames, frames/(med*1e-3), bytes/(med*1e-3)/1e9,
           100*bytes/(med*1e-3)/8e12, (unsigned long long)cs);
    return 0;
}
