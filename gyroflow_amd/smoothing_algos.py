"""The default smoothing algorithm of the reference (/root/reference/smoothing_algos.py:10-257, SmoothingAlgo base +
PlainSlerp).  Sequential recurrence: stays on the CPU; the per-sample loop runs in libgfremap's host code."""
import ctypes as C

import numpy as np

from . import _lib


class SmoothingAlgo:
    def __init__(self, name="Nothing"):
        self.name = name
        self.user_options = {}
        self.num_data_points = 0
        self.gyro_sample_rate = 1
        self.bypass_external_processing = False

    def add_user_option(self, name, default, minimum=None, maximum=None, **ui):
        self.user_options[name] = dict(value=default, default=default, min=minimum, max=maximum)

    def get_user_option_value(self, name):
        return self.user_options[name]["value"]

    def set_user_option(self, name, value):
        o = self.user_options[name]
        if o["min"] is not None:
            value = max(o["min"], value)
        if o["max"] is not None:
            value = min(o["max"], value)
        o["value"] = value

    def get_summary(self):
        return self.name + "," + ",".join("%s:%s" % (k, v["value"]) for k, v in self.user_options.items())

    def get_smooth_orientations(self, times, orientation_quats):
        self.num_data_points = times.shape[0]
        self.gyro_sample_rate = self.num_data_points / (times[-1] - times[0])
        return self.smooth_orientations_internal(times, orientation_quats)

    def smooth_orientations_internal(self, times, orientation_quats):
        return times, orientation_quats


class PlainSlerp(SmoothingAlgo):
    """Default symmetrical quaternion slerp without limits (smoothing_algos.py:208-257)."""

    def __init__(self):
        super().__init__("Plain 3D smoothing")
        self.add_user_option("smoothness", 0.2, 0, 30)

    def smooth_orientations_internal(self, times, orientation_list):
        alpha = 1
        smooth = self.get_user_option_value("smoothness")
        if smooth > 0:
            alpha = 1 - np.exp(-(1 / self.gyro_sample_rate) / smooth)
        q = np.ascontiguousarray(np.asarray(orientation_list, np.float64).reshape(-1, 4))
        out = np.empty_like(q)
        _lib.check(_lib.load().gf_host_plain_slerp(q.ctypes.data_as(_lib._dp), q.shape[0], float(alpha), out.ctypes.data_as(_lib._dp)))
        return times, out
