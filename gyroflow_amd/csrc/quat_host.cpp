// quat_host.cpp -- HOST restatement of the per-sample Python loops that produce the per-frame stabilisation
// quaternions (the hot path's other per-frame input).  No GPU work: quaternion integration and smoothing are
// sequential recurrences and stay on the CPU (BASELINE.json north_star); this only removes the interpreter from them.
// Follows, operation by operation, /root/reference/gyro_integrator.py:102-183 (integrate_all without the accelerometer
// branch), :262-301 (get_interpolated_orientations), :400-429 (rate_to_quat), smoothing_algos.py:219-257 (PlainSlerp) and
// quaternion.py:10-20,108-131 (normalize, quaternion_multiply, rot_between, single_slerp).  Quaternions are (w,x,y,z).
// Built with -ffp-contract=off: every product and sum rounds as numpy's scalar arithmetic does.
#include <math.h>
#include <stddef.h>
#include <string.h>

#include "../../include/gfremap.h"

namespace {

inline void q_normalize(double* q)
{
    const double n = sqrt(((q[0]*q[0] + q[1]*q[1]) + q[2]*q[2]) + q[3]*q[3]);     // q / np.sqrt(q.dot(q))
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

inline void q_mul(const double* Q1, const double* Q2, double* o)                      // quaternion.py:14-20
{
    const double w0 = Q2[0], x0 = Q2[1], y0 = Q2[2], z0 = Q2[3];
    const double w1 = Q1[0], x1 = Q1[1], y1 = Q1[2], z1 = Q1[3];
    const double a = -x1*x0 - y1*y0 - z1*z0 + w1*w0;
    const double b = x1*w0 + y1*z0 - z1*y0 + w1*x0;
    const double c = -x1*z0 + y1*w0 + z1*x0 + w1*y0;
    const double d = x1*y0 - y1*x0 + z1*w0 + w1*z0;
    o[0] = a; o[1] = b; o[2] = c; o[3] = d;
}

inline void single_slerp(const double* v0, const double* v1in, double t, double* o)  // quaternion.py:108-131
{
    double v1[4] = {v1in[0], v1in[1], v1in[2], v1in[3]};
    double dot = ((v0[0]*v1[0] + v0[1]*v1[1]) + v0[2]*v1[2]) + v0[3]*v1[3];
    if (dot < 0.0) { for (int k = 0; k < 4; k++) v1[k] = -v1[k]; dot = -dot; }
    if (dot > 0.9995) {
        double r[4];
        for (int k = 0; k < 4; k++) r[k] = v0[k] + t*(v1[k] - v0[k]);
        const double n = sqrt(((r[0]*r[0] + r[1]*r[1]) + r[2]*r[2]) + r[3]*r[3]);
        for (int k = 0; k < 4; k++) o[k] = r[k]/n;
        return;
    }
    const double theta_0 = acos(dot), sin_theta_0 = sin(theta_0);
    const double theta = theta_0*t, sin_theta = sin(theta);
    const double s0 = cos(theta) - dot*sin_theta/sin_theta_0, s1 = sin_theta/sin_theta_0;
    for (int k = 0; k < 4; k++) o[k] = s0*v0[k] + s1*v1[k];
}

}  // namespace

extern "C" {

GF_API int gf_host_integrate_gyro(const double* gyro, size_t n, const double q0[4], double* out_q)
{
    if (!gyro || !q0 || !out_q || n == 0) return GF_ERR_INVALID_ARG;
    double q[4] = {q0[0], q0[1], q0[2], q0[3]};
    for (size_t i = 0; i < n; i++) {
        const double* g = gyro + 4*i;
        const double last_time = i > 0 ? g[-4] : g[0], next_time = i + 1 < n ? g[4] : g[0];
        const double dt = (next_time - last_time)/2;
        if (g[1] != 0 || g[2] != 0 || g[3] != 0) {                // np.any(omega)
            double ha[3] = {g[1]*dt*0.5, g[2]*dt*0.5, g[3]*dt*0.5};
            const double l = sqrt((ha[0]*ha[0] + ha[1]*ha[1]) + ha[2]*ha[2]);
            double dq[4] = {1, 0, 0, 0};
            if (l > 1.0e-12) {
                const double f = sin(l)/l;
                dq[0] = cos(l); dq[1] = ha[0]*f; dq[2] = ha[1]*f; dq[3] = ha[2]*f;
                q_normalize(dq);
            }
            double r[4];
            q_mul(q, dq, r);
            memcpy(q, r, sizeof q);
            q_normalize(q);
        }
        memcpy(out_q + 4*i, q, sizeof q);
    }
    return GF_OK;
}

GF_API long long gf_host_interp_orientations(const double* times, const double* quats, size_t n, double start, double interval,
                                             double* out_times, double* out_quats, size_t capacity)
{
    if (!times || !quats || n == 0 || !(interval > 0)) return GF_ERR_INVALID_ARG;
    size_t m = 0;
    double time = start;
    double tmp[4];
#define EMIT(qptr)                                                                         \
    do {                                                                                   \
        if (out_times && out_quats) {                                                      \
            if (m >= capacity) return GF_ERR_INVALID_ARG;                                  \
            out_times[m] = time; memcpy(out_quats + 4*m, (qptr), 4*sizeof(double));        \
        }                                                                                  \
        m++; time += interval;                                                             \
    } while (0)
    while (time < 0) EMIT(quats);
    while (times[0] >= time) EMIT(quats);
    for (size_t i = 0; i + 1 < n; i++) {
        while (times[i] <= time && time < times[i+1]) {
            const double weight = (time - times[i])/(times[i+1] - times[i]);
            single_slerp(quats + 4*i, quats + 4*(i+1), weight, tmp);
            EMIT(tmp);
        }
        if (time < times[i]) EMIT(quats + 4*i);          // continue even if gyro data is missing
    }
#undef EMIT
    return (long long)m;
}

GF_API int gf_host_plain_slerp(const double* quats, size_t n, double alpha, double* out)
{
    if (!quats || !out || n == 0) return GF_ERR_INVALID_ARG;
    double v[4];
    memcpy(v, quats, sizeof v);
    for (size_t i = 0; i < n; i++) {                      // forward pass into out
        double r[4];
        single_slerp(v, quats + 4*i, alpha, r);
        memcpy(v, r, sizeof v);
        memcpy(out + 4*i, v, sizeof v);
    }
    memcpy(v, out + 4*(n - 1), sizeof v);
    for (size_t i = n; i-- > 0;) {                        // reverse pass, in place (entry i is read before it is replaced)
        double r[4];
        single_slerp(v, out + 4*i, alpha, r);
        memcpy(v, r, sizeof v);
        memcpy(out + 4*i, v, sizeof v);
    }
    return GF_OK;
}

GF_API int gf_host_rot_between(const double* q1, const double* q2, size_t n, double* out)
{
    if (!q1 || !q2 || !out) return GF_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; i++) {
        const double inv[4] = {q1[4*i], -q1[4*i+1], -q1[4*i+2], -q1[4*i+3]};
        q_mul(inv, q2 + 4*i, out + 4*i);
    }
    return GF_OK;
}

}  // extern "C"
