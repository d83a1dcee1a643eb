// tools/host_sanity.cpp -- exercises the library's HOST-ONLY entry points (no GPU call) in one executable so that the host
// code can run under AddressSanitizer / UBSan in the build container:
//   hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
//         gyroflow_amd/csrc/gfremap.hip gyroflow_amd/csrc/quat_host.cpp tools/host_sanity.cpp -o tools/_build/host_sanity
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/gfremap.h"

int main()
{
    int bad = 0;
    // lens table
    const double D[4] = {0.05764, -0.00234, 0.02008, -0.01086};
    std::vector<double> t(100000), S(t.size());
    for (size_t i = 0; i < t.size(); i++) t[i] = 254.9*(double)i/(double)t.size();
    if (gf_lens_scale_host(D, t.data(), t.size(), S.data())) { printf("lens_scale failed: %s\n", gf_last_error()); bad++; }
    for (size_t i = 1; i < t.size(); i++) {
        const double r = sqrt(t[i]), th = atan(r), th2 = th*th;
        const double want = th*(1 + th2*(D[0] + th2*(D[1] + th2*(D[2] + th2*D[3]))))/r;
        if (!(fabs(S[i] - want) <= 1e-13*fabs(want) + 1e-15)) { if (bad < 5) printf("lens mismatch at t=%g: %.17g vs %.17g\n", t[i], S[i], want); bad++; }
    }
    double out_nan[2]; const double tq[2] = {-1.0, 300.0};
    gf_lens_scale_host(D, tq, 2, out_nan);
    if (out_nan[1] == out_nan[1]) { printf("expected NaN outside the table\n"); bad++; }
    if (gf_lens_scale_host(nullptr, tq, 2, out_nan) != GF_ERR_INVALID_ARG) bad++;
    // frame transform
    const double K[9] = {1766, 0, 1920, 0, 1766, 1080, 0, 0, 1}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z[9] = {0};
    gf_frame_xform xf;
    if (gf_make_frame_xform(K, R, K, &xf)) bad++;
    if (fabs(xf.iR[0] - 1/1766.0) > 1e-18 || fabs(xf.iR[2] + 1920/1766.0) > 1e-15) { printf("iR wrong\n"); bad++; }
    if (gf_make_frame_xform(K, Z, K, &xf) != GF_ERR_SINGULAR) bad++;
    // cubic table
    std::vector<int16_t> tab(1024*16);
    if (gf_get_cubic_table(tab.data())) bad++;
    for (int a = 0; a < 1024; a++) { int s = 0; for (int k = 0; k < 16; k++) s += tab[a*16 + k]; if (s != 32768) { bad++; break; } }
    // quaternion recurrences
    const size_t n = 5000;
    std::vector<double> gyro(n*4), q(n*4);
    unsigned seed = 7;
    for (size_t i = 0; i < n; i++) {
        gyro[4*i] = 0.005*(double)i + (i > 2000 ? 0.3 : 0.0);
        for (int k = 1; k < 4; k++) { seed = seed*1664525u + 1013904223u; gyro[4*i + k] = ((double)(seed >> 8)/(1 << 24) - 0.5)*4; }
    }
    const double q0[4] = {1, 0.0001, 0.0001, 0.0001};
    if (gf_host_integrate_gyro(gyro.data(), n, q0, q.data())) bad++;
    std::vector<double> times(n); for (size_t i = 0; i < n; i++) times[i] = gyro[4*i];
    const long long m = gf_host_interp_orientations(times.data(), q.data(), n, -0.1, 1/60.0, nullptr, nullptr, 0);
    if (m <= 0) { printf("interp count %lld\n", m); bad++; }
    std::vector<double> ot((size_t)m), oq((size_t)m*4), sm((size_t)m*4), st((size_t)m*4);
    if (gf_host_interp_orientations(times.data(), q.data(), n, -0.1, 1/60.0, ot.data(), oq.data(), (size_t)m) != m) bad++;
    if (gf_host_interp_orientations(times.data(), q.data(), n, -0.1, 1/60.0, ot.data(), oq.data(), (size_t)m - 1) >= 0) bad++;   // capacity check
    if (gf_host_plain_slerp(oq.data(), (size_t)m, 0.08, sm.data())) bad++;
    if (gf_host_rot_between(sm.data(), oq.data(), (size_t)m, st.data())) bad++;
    for (long long i = 0; i < m; i++) {
        const double* s = &st[4*i];
        const double nn = s[0]*s[0] + s[1]*s[1] + s[2]*s[2] + s[3]*s[3];
        if (!(fabs(nn - 1) < 1e-9)) { bad++; break; }
    }
    printf("host_sanity: %s (%d problems), version %d\n", bad ? "FAILED" : "ok", bad, gf_version());
    return bad ? 1 : 0;
}
