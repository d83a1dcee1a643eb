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


class GyroIntegrator:
    def __init__(self, gyro_data, time_scaling=1, gyro_scaling=1, zero_out_time=True, initial_orientation=None, acc_data=None,
                 acc_scaling=1):
        self.gyro = np.array(gyro_data, dtype=np.float64)
        self.acc = None
        self.acc_available = False
        self.last_used_acc = False
        if acc_data is not None and self.gyro.shape[0] == np.asarray(acc_data).shape[0]:
            # the reference's constructor applies time_scaling / acc_scaling to the *gyro* columns here (:40-41); kept
            self.acc_available = True
            self.acc = np.array(acc_data, dtype=np.float64)
            self.gyro[:, 0] *= time_scaling
            self.gyro[:, 1:4] *= acc_scaling
        time_order_check = self.gyro[:-1, 0] > self.gyro[1:, 0]
        if np.any(time_order_check):
            print("Truncated bad gyro data")
            self.gyro = self.gyro[0:np.argmax(time_order_check) + 1, :]
            if self.acc_available:
                self.acc = self.acc[0:np.argmax(time_order_check) + 1, :]
        self.gyro[:, 0] *= time_scaling
        self.gyro[:, 1:4] *= gyro_scaling
        if zero_out_time:
            self.gyro[:, 0] -= self.gyro[0, 0]
            if self.acc_available:
                self.acc[:, 0] -= self.acc[0, 0]
        self.num_data_points = self.gyro.shape[0]
        self.gyro_sample_rate = self.num_data_points / (self.gyro[-1, 0] - self.gyro[0, 0])
        if initial_orientation is not None:
            self.initial_orientation = np.array(initial_orientation, dtype=np.float64)
        else:
            self.initial_orientation = np.array([1, 0.0001, 0.0001, 0.0001])
        self.orientation = np.copy(self.initial_orientation)
        self.orientation_list = None
        self.time_list = None
        self.already_integrated = False
        self.smoothing_algo = None
        self.interp_times = None
        self.interp_orientations = None

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
