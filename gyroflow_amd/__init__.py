"""gyroflow_amd -- MI355X (gfx950) implementation of gyroflow's undistort + rotate remap hot path.

Python call surface kept from the reference (ElvinC/gyroflow): FisheyeCalibrator.get_maps, cv2-style remap,
NonlinearStretch, Stabilizer.renderfile / map_function / set_hyperlapse; all pixel work runs in libgfremap.so
(hand-written HIP, see gyroflow_amd/csrc/gfremap.hip and include/gfremap.h).  There is no CPU fallback.
"""
from . import _lib, cv                                   # noqa: F401
from .adaptive_zoom import AdaptiveZoom                  # noqa: F401
from .calibrate_video import FisheyeCalibrator           # noqa: F401
from .nonlinear_stretch import NonlinearStretch          # noqa: F401
from .stabilizer import Stabilizer, OnlyUndistort, ArraySource, ListSink, NullSink, synthetic_frames   # noqa: F401
from .frameio import RawVideoSource, RawVideoSink, PinnedRingSource         # noqa: F401
from .sharded import render_sharded, shard_layout        # noqa: F401  (also attaches Stabilizer.render_sharded)
from .gyro_integrator import GyroIntegrator              # noqa: F401
from .smoothing_algos import PlainSlerp                  # noqa: F401

__all__ = ["cv", "FisheyeCalibrator", "AdaptiveZoom", "NonlinearStretch", "Stabilizer", "OnlyUndistort", "ArraySource", "ListSink", "NullSink",
           "RawVideoSource", "RawVideoSink", "PinnedRingSource", "render_sharded", "shard_layout", "GyroIntegrator", "PlainSlerp",
           "synthetic_frames"]
