"""GyroIntegrator: gyro samples -> orientations -> per-frame stabilisation quaternions, the producer of the hot path's
`stab_transform` input (/root/reference/gyro_integrator.py:12-327; call sites stabilizer.py:1608,1700-1703).

Same class and method names as the reference.  The per-sample Python loops (integrate_all :133-176, the resampling loop
:262-301, rot_between :323-325) run in libgfremap's host code (csrc/quat_host.cpp) with the same order of floating-point
operations; they are sequential recurrences and stay on the CPU.  Not mirrored: the accelerometer complementary filter
(use_acc=True, :149-160) and the legacy get_interpolated_stab_transform_old."""
import ctypes as C

import numpy as np

from . import _lib, smoothing_algos


def _dp(a):
    return a.ctypes.data_as(_lib._dp)


DEFAULT_ORIENTATION = (1, 0.0001, 0.0001, 0.0001)       # the reference's start quaternion when none is given (:78)


def _monotone_prefix(t):
    """Number of leading samples before the first timestamp that runs backwards (all of them if none does)."""
    back = np.flatnonzero(t[:-1] > t[1:])
    return t.shape[0] if back.size == 0 else int(back[0]) + 1


class GyroIntegrator:
    def __init__(self, gyro_data, time_scaling=1, gyro_scaling=1, zero_out_time=True, initial_orientation=None, acc_data=None,
                 acc_scaling=1):
        """Rows of gyro_data: t, gx, gy, gz (acc_data: t, ax, ay, az).  The log is taken in four steps, each a helper below:
        pair it with the accelerometer log, cut it at the first timestamp that runs backwards, scale it, move t = 0 to the
        first sample (reference :27-71)."""
        self.last_used_acc = False
        self.gyro = np.array(gyro_data, dtype=np.float64)
        self.acc, self.acc_available = self._pair_accelerometer(acc_data, time_scaling, acc_scaling)
        self._truncate(_monotone_prefix(self.gyro[:, 0]))
        self._scale(time_scaling, gyro_scaling)
        if zero_out_time:
            self._rebase_time()
        self.num_data_points = self.gyro.shape[0]
        self.gyro_sample_rate = self.num_data_points / (self.gyro[-1, 0] - self.gyro[0, 0])
        self.initial_orientation = np.array(DEFAULT_ORIENTATION if initial_orientation is None else initial_orientation, dtype=np.float64)
        self.orientation = self.initial_orientation.copy()
        self.orientation_list = self.time_list = None
        self.interp_times = self.interp_orientations = None
        self.smoothing_algo = None
        self.already_integrated = False

    def _pair_accelerometer(self, acc_data, time_scaling, acc_scaling):
        """-> (acc copy or None, usable).  A log of another length is refused with the reference's message.  A usable one has a
        side effect the reference has too (:40-41, pinned by tests/test_host.py::test_gyro_integrator_acc_scaling_quirk): the
        time and accelerometer scale factors land on the GYRO columns here, before the regular scaling step applies
        time_scaling a second time."""
        if acc_data is None:
            return None, False
        acc = np.array(acc_data, dtype=np.float64)
        if acc.shape[0] != self.gyro.shape[0]:
            print("Gyro and acceleration data don't line up")
            return None, False
        self.gyro[:, 0] *= time_scaling
        self.gyro[:, 1:4] *= acc_scaling
        return acc, True

    def _truncate(self, n_keep):
        if n_keep < self.gyro.shape[0]:
            print("Truncated bad gyro data")
            self.gyro = self.gyro[:n_keep, :]
            if self.acc_available:
                self.acc = self.acc[:n_keep, :]

    def _scale(self, time_scaling, gyro_scaling):
        self.gyro[:, 0] *= time_scaling
        self.gyro[:, 1:4] *= gyro_scaling

    def _rebase_time(self):
        self.gyro[:, 0] -= self.gyro[0, 0]
        if self.acc_available:
            self.acc[:, 0] -= self.acc[0, 0]

    def integrate_all(self, use_acc=False):
        if self.already_integrated and (use_acc == self.last_used_acc or not self.acc_available):
            return (self.time_list, self.orientation_list)
        if self.acc_available and use_acc:
            raise NotImplementedError("the accelerometer complementary filter is not mirrored")
        self.last_used_acc = use_acc
        g = np.ascontiguousarray(self.gyro[:, :4])
        out = np.empty((self.num_data_points, 4), np.float64)
        q0 = np.ascontiguousarray(self.initial_orientation, dtype=np.float64)
        _lib.check(_lib.load().gf_host_integrate_gyro(_dp(g), self.num_data_points, _dp(q0), _dp(out)))
        self.orientation_list = out
        self.orientation = out[-1].copy()
        self.time_list = self.gyro[:, 0].copy()
        self.already_integrated = True
        return (self.time_list, self.orientation_list)

    def get_orientations(self):
        if self.already_integrated:
            return (self.time_list, self.orientation_list)
        return None, None

    def set_smoothing_algo(self, algo):
        if not algo:
            algo = smoothing_algos.PlainSlerp()       # (the reference drops this default on the floor, :199-200)
        else:
            self.smoothing_algo = algo

    def get_smoothed_orientation(self):
        if not self.smoothing_algo:
            self.smoothing_algo = smoothing_algos.PlainSlerp()
        return self.smoothing_algo.get_smooth_orientations(self.time_list, self.orientation_list)

    def get_interpolated_orientations(self, start=0, interval=1 / 29.97):
        time_list, orientations = self.get_orientations()
        t = np.ascontiguousarray(time_list, dtype=np.float64)
        q = np.ascontiguousarray(orientations, dtype=np.float64)
        lib = _lib.load()
        n = lib.gf_host_interp_orientations(_dp(t), _dp(q), t.shape[0], float(start), float(interval), None, None, 0)
        if n < 0:
            raise _lib.GfError(int(n), "gf_host_interp_orientations: bad arguments")
        out_t = np.empty(n, np.float64)
        out_q = np.empty((n, 4), np.float64)
        if n:
            m = lib.gf_host_interp_orientations(_dp(t), _dp(q), t.shape[0], float(start), float(interval), _dp(out_t), _dp(out_q), n)
            assert m == n
        self.interp_times, self.interp_orientations = out_t, out_q
        return (self.interp_times, self.interp_orientations)

    def get_interpolated_stab_transform(self, start=0, interval=1 / 29.97):
        if self.smoothing_algo:
            if self.smoothing_algo.bypass_external_processing:
                raise NotImplementedError("smoothing algorithms that bypass orientation integration are not mirrored")
        else:
            self.smoothing_algo = smoothing_algos.PlainSlerp()
        time_list, interp_orientations = self.get_interpolated_orientations(start=start, interval=interval)
        _, smoothed = self.smoothing_algo.get_smooth_orientations(time_list, interp_orientations)
        smoothed = np.ascontiguousarray(smoothed, dtype=np.float64)
        stab = np.zeros(smoothed.shape)
        _lib.check(_lib.load().gf_host_rot_between(_dp(smoothed), _dp(interp_orientations), smoothed.shape[0], _dp(stab)))
        return time_list, stab
