#!/usr/bin/env python3
"""Generate the committed golden fixtures in tests/golden/ from the reference's own Python.

Runs ONLY in the build container (needs /root/reference); its outputs (small .npz / .json data
files) are what travels to the GPU box.  The reference cannot be imported as-is (no cv2, tkinter,
PySide2 here -- ordinary ModuleNotFoundError), so inert stand-in modules are put in sys.modules
and `cv2.fisheye.initUndistortRectifyMap` / `cv2.convertMaps` RECORD their arguments: that is
exactly the data that crosses the reference's cv2 boundary (calibrate_video.py:376-383).  What
cv2 would have returned is NOT captured here -- nothing in the reference pins it ("parity
unpinned"); map/pixel expectations come from oracle/ and are stored separately by
make_oracle_vectors.py.

Fixtures written:
  getmaps_calls.npz     (K, D, R, P, size, m1type) for a list of get_maps() calls + the call kwargs
  stretch_maps.npz      NonlinearStretch.recompute_maps() float32 maps (small sizes in full,
                        4000x3000->3840x2160 as sampled rows/cols)
  quats_rollpitchyaw.npz  FakeData("rollpitchyaw") -> GyroIntegrator -> PlainSlerp -> 300 stab quats
  quats_gyroflow_sample.npz  stab_transform rows of plugins/blender/GX016015.MP4.gyroflow
  presets/*.json        the camera presets the BASELINE configs name (input data files)
"""
import json
import os
import shutil
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------- stand-in modules
calls = []


class _Any:
    """Inert stand-in: any attribute is another _Any, calling returns an _Any."""
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Any()

    def __call__(self, *a, **k):
        return _Any()

    def __add__(self, o):
        return 0
    __radd__ = __add__


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__getattr__ = lambda k: _Any()
    sys.modules[name] = m
    return m


def _rec_init(K, D, R, P, size, m1type):
    calls.append(dict(K=np.array(K, np.float64), D=np.array(D, np.float64).reshape(-1), R=np.array(R, np.float64),
                      P=np.array(P, np.float64), size=np.array(size, np.int64), m1type=int(m1type)))
    h, w = int(size[1]), int(size[0])
    if m1type == 11:
        return np.zeros((h, w, 2), np.int16), np.zeros((h, w), np.uint16)
    return np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)


def _rec_convert(m1, m2, t):
    calls[-1]["convert_in_dtype"] = str(m1.dtype)
    return np.zeros(m1.shape + (2,), np.int16), np.zeros(m1.shape, np.uint16)


fisheye = types.SimpleNamespace(initUndistortRectifyMap=_rec_init, CALIB_RECOMPUTE_EXTRINSIC=2, CALIB_CHECK_COND=4,
                                CALIB_FIX_SKEW=8)
_stub("cv2", fisheye=fisheye, convertMaps=_rec_convert, CV_16SC2=11, CV_32FC1=5, TERM_CRITERIA_EPS=2,
      TERM_CRITERIA_MAX_ITER=1, INTER_LINEAR=1, INTER_CUBIC=2, BORDER_CONSTANT=0, BORDER_REPLICATE=1,
      BORDER_TRANSPARENT=5)
_stub("tkinter", Tk=_Any())
_stub("tkinter.filedialog", askopenfilename=_Any())
_stub("PySide2")
for sub in ("QtCore", "QtWidgets", "QtGui"):
    setattr(sys.modules["PySide2"], sub, _stub("PySide2." + sub))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)

import calibrate_video            # noqa: E402  (reference)
import nonlinear_stretch          # noqa: E402  (reference)
import gyro_integrator            # noqa: E402  (reference)
import smoothing_algos            # noqa: E402  (reference)
from scipy.spatial.transform import Rotation  # noqa: E402

PRESETS = {
    "hero8_27k_43": "camera_presets/GoPro/GoPro_Hero8_Black_2_7K_4by3_60fps_wide.json",
    "hero7_4k_169": "camera_presets/GoPro/GoPro_Hero7_Black_4K_60_wide_16by9_1_120.json",
    "hero9_4k_43": "camera_presets/GoPro/GoPro_HERO9_Black_Wide_4K_4by3_Wide.json",
    "hero9_5k_169": "camera_presets/GoPro/GoPro_HERO9_Black_Wide_5K_16by9_Wide.json",
    "runcam5_stretched": "camera_presets/RunCam/Runcam_5_Orange_4K_30FPS_XV_16by9_stretched.json",
}


def copy_presets():
    os.makedirs(os.path.join(OUT, "presets"), exist_ok=True)
    for key, rel in PRESETS.items():
        shutil.copyfile(os.path.join(REF, rel), os.path.join(OUT, "presets", key + ".json"))


def unit(q):
    q = np.array(q, np.float64)
    return q / np.linalg.norm(q)


def gen_getmaps_calls():
    """Drive the reference's FisheyeCalibrator.get_maps and record what reaches cv2."""
    cases = [
        # (name, preset, kwargs)
        ("survey_capture", "hero8_27k_43", dict(fov_scale=0.8, new_img_dim=(2704, 2028), output_dim=(1920, 1080),
                                                update_new_K=False, quat=unit([0.9990482215818578, 0.02, -0.03, 0.025]),
                                                focalCenter=np.array([1352., 1014.]))),
        ("c1_1080p", "hero8_27k_43", dict(fov_scale=1.0, new_img_dim=(1920, 1080), output_dim=(1920, 1080),
                                          update_new_K=False, quat=unit([0.998, 0.03, 0.04, -0.02]), focalCenter=None)),
        ("c2_27k_to_1080p", "hero8_27k_43", dict(fov_scale=1.0, new_img_dim=(2704, 2028), output_dim=(1920, 1080),
                                                 update_new_K=False, quat=unit([0.9995, -0.01, 0.02, 0.022]),
                                                 focalCenter=np.array([1300., 1000.]))),
        ("c3_4k", "hero7_4k_169", dict(fov_scale=1.0, new_img_dim=(3840, 2160), output_dim=(3840, 2160),
                                       update_new_K=False, quat=unit([0.993, 0.08, -0.06, 0.05]), focalCenter=None)),
        ("c3_identity", "hero7_4k_169", dict(fov_scale=1.0, new_img_dim=(3840, 2160), output_dim=(3840, 2160),
                                             update_new_K=False, quat=None, focalCenter=None)),
        ("preview_720p", "hero7_4k_169", dict(fov_scale=0.9, new_img_dim=(3840, 2160), output_dim=(1280, 720),
                                              update_new_K=False, quat=unit([0.99, 0.1, 0.05, -0.07]), focalCenter=None,
                                              original_stretched=True)),
        ("c5_5k", "hero9_5k_169", dict(fov_scale=1.2, new_img_dim=(5312, 2988), output_dim=(5312, 2988),
                                       update_new_K=False, quat=unit([0.9992, 0.02, 0.03, 0.01]), focalCenter=None)),
        ("stretched_runcam", "runcam5_stretched", dict(fov_scale=1.0, new_img_dim=(3840, 2160), output_dim=(1920, 1080),
                                                        update_new_K=False, quat=unit([0.999, 0.02, -0.03, 0.01]),
                                                        focalCenter=None)),
        ("defaults_no_args", "hero8_27k_43", dict()),
    ]
    rec = {}
    meta = []
    for name, preset, kw in cases:
        cal = calibrate_video.FisheyeCalibrator()
        cal.load_calibration_json(os.path.join(REF, PRESETS[preset]))
        calls.clear()
        cal.get_maps(**kw)
        assert len(calls) == 1
        c = calls[0]
        for k in ("K", "D", "R", "P", "size"):
            rec[f"{name}.{k}"] = c[k]
        rec[f"{name}.m1type"] = np.array(c["m1type"])
        kwj = {k: (None if v is None else (np.asarray(v).tolist() if not isinstance(v, (bool, float, int)) else v))
               for k, v in kw.items()}
        meta.append(dict(name=name, preset=preset, kwargs=kwj, converted="convert_in_dtype" in c,
                         convert_in_dtype=c.get("convert_in_dtype")))
    np.savez_compressed(os.path.join(OUT, "getmaps_calls.npz"), **rec)
    with open(os.path.join(OUT, "getmaps_calls.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_stretch_maps():
    rec = {}
    for tag, in_size, out_size, safe, expo in [("s0", (400, 300), (384, 216), 0.0, 2), ("s1", (320, 240), (427, 240), 0.3, 2.5),
                                               ("s2", (12, 9), (16, 9), 0.0, 2), ("s3", (640, 480), (1280, 720), 0.999, 1),
                                               ("s4", (96, 72), (128, 72), 0.1, 3)]:
        ns = nonlinear_stretch.NonlinearStretch(in_size, out_size)
        ns.set_safe_area(safe); ns.set_expo(expo)
        ns.recompute_maps()
        rec[tag + ".map1"] = ns.map1; rec[tag + ".map2"] = ns.map2
        rec[tag + ".params"] = np.array([in_size[0], in_size[1], out_size[0], out_size[1], ns.safe_area, expo], np.float64)
    ns = nonlinear_stretch.NonlinearStretch((4000, 3000), (3840, 2160))
    ns.recompute_maps()
    rows = np.array([0, 1, 1079, 1080, 2159]); cols = np.arange(0, 3840, 97)
    rec["c4.rows"] = rows; rec["c4.cols"] = cols
    rec["c4.map1_rows"] = ns.map1[rows, :]; rec["c4.map2_cols"] = ns.map2[:, cols]
    rec["c4.map1_dtype_shape"] = np.array(ns.map1.shape)
    ns = nonlinear_stretch.NonlinearStretch((2880, 2160), (3840, 2160))
    ns.recompute_maps()
    rec["c4b.map1_row0"] = ns.map1[0, :]; rec["c4b.map2_col0"] = ns.map2[:, 0]
    np.savez_compressed(os.path.join(OUT, "stretch_maps.npz"), **rec)


def gen_quats():
    # gyrolog.FakeData("rollpitchyaw") (gyrolog.py:1136-1153), restated as data: gyrolog itself needs
    # orangebox/construct which are absent here.
    N = 1000
    gyro = np.zeros((N, 4)); gyro[:, 0] = np.arange(N) / 100
    gyro[200:300, 3] = 1; gyro[400:500, 1] = 1; gyro[600:700, 2] = 1
    init = Rotation.from_euler('zxy', [0, 0, np.pi / 2]).as_quat()[[3, 0, 1, 2]]   # stabilizer.py:122-123
    integ = gyro_integrator.GyroIntegrator(gyro, zero_out_time=False, initial_orientation=init)
    integ.integrate_all()
    integ.set_smoothing_algo(smoothing_algos.PlainSlerp())
    times, stab = integ.get_interpolated_stab_transform(start=0, interval=1 / 30)
    np.savez_compressed(os.path.join(OUT, "quats_rollpitchyaw.npz"), times=times, stab_transform=stab, gyro=gyro,
                        initial_orientation=init)

    with open(os.path.join(REF, "plugins/blender/GX016015.MP4.gyroflow")) as f:
        g = json.load(f)
    st = np.array(g["stab_transform"], np.float64)      # idx, t, fcorr, w, x, y, z  (stabilizer.py:1645-1652)
    cal = g["calibration_data"]
    np.savez_compressed(os.path.join(OUT, "quats_gyroflow_sample.npz"), stab_transform=st,
                        min_fcorr=np.array(g.get("min_fcorr", np.nan)), fps=np.array(g["video_info"]["fps"]),
                        K=np.array(cal["fisheye_params"]["camera_matrix"], np.float64),
                        D=np.array(cal["fisheye_params"]["distortion_coeffs"], np.float64),
                        calib_dimension=np.array([cal["calib_dimension"]["w"], cal["calib_dimension"]["h"]]),
                        video_dim=np.array([g["video_info"]["orig_w"], g["video_info"]["orig_h"]]),
                        raw_imu=np.array(g["raw_imu"], np.float64),                     # t, gx, gy, gz, ax, ay, az (5 decimals)
                        frame_orientation=np.array(g["frame_orientation"], np.float64))  # idx, t, w, x, y, z (full precision)
    print("gyroflow sample keys:", sorted(g.keys()))


if __name__ == "__main__":
    copy_presets()
    gen_getmaps_calls()
    gen_stretch_maps()
    gen_quats()
    print("wrote fixtures to", OUT)
