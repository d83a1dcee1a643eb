"""CPU tests of the host-side mirror against data captured from the reference's own Python
(tests/golden/*.npz, written by tests/golden/make_golden.py) and of the C ABI's surface."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, preset_path


def _cal(key):
    from gyroflow_amd import FisheyeCalibrator
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path(key))
    return cal


def test_frame_matrices_equal_reference_capture(golden_calls):
    """What our get_maps hands to the kernels == what the reference's get_maps hands to cv2 (bit for bit):
    scaled_K, D, R (quat -> matrix + axis flips), new_K, output size, map type."""
    z, meta = golden_calls
    assert len(meta) >= 9
    for m in meta:
        cal = _cal(m["preset"])
        kw = dict(m["kwargs"])
        quat = kw.get("quat")
        fc = kw.get("focalCenter")
        K, D, R, P, od = cal._frame_matrices(kw.get("fov_scale", 1.0),
                                             tuple(kw["output_dim"]) if kw.get("output_dim") else None,
                                             tuple(kw["new_img_dim"]) if kw.get("new_img_dim") else None,
                                             None if quat is None else np.array(quat),
                                             None if fc is None else np.array(fc),
                                             kw.get("original_stretched", True))
        n = m["name"]
        np.testing.assert_array_equal(K, z[n + ".K"])
        np.testing.assert_array_equal(np.asarray(D, np.float64).reshape(-1), z[n + ".D"])
        np.testing.assert_array_equal(R, z[n + ".R"])
        np.testing.assert_array_equal(P, z[n + ".P"])
        assert tuple(od) == tuple(int(v) for v in z[n + ".size"])
        stretched = cal.image_is_stretched()
        assert int(z[n + ".m1type"]) == (5 if stretched else 11)
        assert m["converted"] == stretched and (not stretched or m["convert_in_dtype"] == "float32")


def test_survey_known_values(golden_calls):
    """The values quoted in SURVEY.md 8(c) for the Hero8 capture."""
    cal = _cal("hero8_27k_43")
    q = np.array([0.9990482215818578, 0.02, -0.03, 0.025]); q /= np.linalg.norm(q)
    K, D, R, P, od = cal._frame_matrices(0.8, (1920, 1080), (2704, 2028), q, np.array([1352., 1014.]), True)
    assert P.tolist() == [[1490.6551163044169, 0.0, 960.0], [0.0, 1493.7196642070426, 540.0], [0.0, 0.0, 1.0]]
    assert R[0].tolist() == [0.9969500681630664, 0.0511512678970602, 0.058941576006925456]
    np.testing.assert_array_equal(K, cal.K)          # ratio 1 -> scaled_K == K with [2][2] = 1


def test_update_new_K_side_effect_and_reentrancy():
    cal = _cal("hero7_4k_169")
    K0 = cal.K.copy()
    a = cal._frame_matrices(1.3, (1280, 720), (3840, 2160), np.array([1, 0, 0, 0.]), None, True)
    b = cal._frame_matrices(1.3, (1280, 720), (3840, 2160), np.array([1, 0, 0, 0.]), None, True)
    np.testing.assert_array_equal(cal.K, K0)         # read-only use of K/D: safe from the preview thread
    np.testing.assert_array_equal(a[3], b[3])
    np.testing.assert_array_equal(a[2], np.eye(3))


def test_preset_loader_errors(tmp_path):
    from gyroflow_amd import FisheyeCalibrator
    with open(preset_path("hero8_27k_43")) as f:
        d = json.load(f)
    cal = FisheyeCalibrator()
    info = cal.load_calibration_data(d)
    assert info["width"] == 2704 and info["height"] == 2028 and cal.input_horizontal_stretch == 1
    assert tuple(cal.calib_dimension) == (2704, 2028) and cal.K.shape == (3, 3) and len(cal.D) == 4
    bad = dict(d); bad["use_opencv_fisheye"] = False
    with pytest.raises(Exception, match="Preset not for OpenCV fisheye lens model"):
        FisheyeCalibrator().load_calibration_data(bad)
    bad = dict(d); del bad["fisheye_params"]
    with pytest.raises(KeyError):
        FisheyeCalibrator().load_calibration_data(bad)
    cal = _cal("runcam5_stretched")
    assert cal.image_is_stretched() and cal.input_horizontal_stretch == 0.75
    assert cal.get_stretched_size_from_dimension((3840, 2160)) == (2880, 2160)


def test_preset_version_note_and_missing_fields(capsys):
    """calibrate_video.py:528-538: the version note is printed whether or not printinfo is set, a preset without
    "calibrator_version" raises KeyError, and both happen before K / D change."""
    from gyroflow_amd import FisheyeCalibrator
    from gyroflow_amd.calibrate_video import __version__
    with open(preset_path("hero8_27k_43")) as f:
        d = json.load(f)
    other = dict(d); other["calibrator_version"] = "0.0.1-test"
    FisheyeCalibrator().load_calibration_data(other)                       # printinfo=False
    out = capsys.readouterr().out
    assert out == "Note: Versions don't match. Calibrator: {}, Preset: 0.0.1-test. Should be fine though.\n".format(__version__)
    same = dict(d); same["calibrator_version"] = __version__
    FisheyeCalibrator().load_calibration_data(same)
    assert capsys.readouterr().out == ""
    FisheyeCalibrator().load_calibration_data(same, True)
    lines = capsys.readouterr().out.splitlines()
    assert lines[0] == "Preset name: " + d["name"] and lines[1] == "Note: " + d["note"] and lines[2].startswith("Made with %d frames" % d["num_images"])
    for key, flag in (("calibrator_version", False), ("note", True), ("date", True)):
        bad = dict(same); del bad[key]
        cal = FisheyeCalibrator()
        K0 = cal.K.copy()
        with pytest.raises(KeyError):
            cal.load_calibration_data(bad, flag)
        np.testing.assert_array_equal(cal.K, K0)                           # nothing but data_from_preset_file was touched
        assert cal.extra_cam_info is None and cal.data_from_preset_file
    capsys.readouterr()
    bad = dict(same); del bad["note"]
    assert FisheyeCalibrator().load_calibration_data(bad)["note"] is None      # only read under printinfo


def test_nonlinear_stretch_maps_equal_reference():
    """recompute_maps() == the reference's numpy arithmetic, every float32 bit (captured in stretch_maps.npz)."""
    from gyroflow_amd import NonlinearStretch
    z = np.load(os.path.join(GOLDEN, "stretch_maps.npz"))
    for tag in ("s0", "s1", "s2", "s3", "s4"):
        p = z[tag + ".params"]
        ns = NonlinearStretch((int(p[0]), int(p[1])), (int(p[2]), int(p[3])))
        ns.set_safe_area(p[4]); ns.set_expo(p[5] if p[5] != int(p[5]) else int(p[5]))
        ns.recompute_maps()
        assert ns.map1.dtype == np.float32 and ns.map1.shape == (int(p[3]), int(p[2]))
        np.testing.assert_array_equal(ns.map1, z[tag + ".map1"])
        np.testing.assert_array_equal(ns.map2, z[tag + ".map2"])
    ns = NonlinearStretch((4000, 3000), (3840, 2160))
    ns.recompute_maps()
    np.testing.assert_array_equal(ns.map1[z["c4.rows"], :], z["c4.map1_rows"])
    np.testing.assert_array_equal(ns.map2[:, z["c4.cols"]], z["c4.map2_cols"])
    # the values quoted in SURVEY.md 8(c)
    np.testing.assert_allclose(ns.map1[0, [0, 1, 1919, 1920, 3839]], [-0.5, 0.19462529, 1998.1113, 1999.5, 3998.8054], rtol=1e-7)
    np.testing.assert_allclose(ns.map2[[0, 1, 2159], 0], [-0.5, 0.8888889, 2998.111], rtol=1e-7)
    ns = NonlinearStretch()
    ns.set_safe_area(5)
    assert ns.safe_area == 0.999


def test_border_from_bg_color_and_hyperlapse():
    from gyroflow_amd import Stabilizer, cv
    assert Stabilizer.border_from_bg_color("REPLICATE") == (cv.BORDER_REPLICATE, 0)
    assert Stabilizer.border_from_bg_color("HISTORY") == (cv.BORDER_TRANSPARENT, 0)
    assert Stabilizer.border_from_bg_color("#000000") == (cv.BORDER_CONSTANT, [0, 0, 0])
    assert Stabilizer.border_from_bg_color("#336699") == (cv.BORDER_CONSTANT, [153, 102, 51])     # RGB -> BGR
    assert Stabilizer.border_from_bg_color("red") == (cv.BORDER_CONSTANT, [0, 0, 255])


def test_frame_shard_partitions():
    from gyroflow_amd.stabilizer import frame_shard
    for n, world, mult in [(3600, 1, 1), (3600, 8, 1), (3600, 7, 4), (10, 4, 4), (5, 8, 1), (1201, 3, 4)]:
        cover = []
        for r in range(world):
            lo, hi = frame_shard(n, r, world, mult)
            assert lo % mult == 0 and lo <= hi <= n
            cover += list(range(lo, hi))
        assert cover == list(range(n))


def test_abi_exports_every_declared_symbol():
    """libgfremap.so loads without a GPU and exports exactly what include/gfremap.h declares."""
    with open(os.path.join(ROOT, "include", "gfremap.h")) as f:
        hdr = f.read()
    declared = set(re.findall(r"GF_API\s+(?:const\s+char\*|int|long long)\s+(gf_\w+)\s*\(", hdr))
    assert len(declared) >= 15
    from gyroflow_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), "ctypes prototypes out of sync with the header"
    assert _lib.load().gf_version() == int(re.search(r"#define GF_VERSION (\d+)", hdr).group(1))
    assert ctypes.sizeof(_lib.FrameXform) == 13 * 8


def test_host_only_abi_calls(orc):
    """gf_make_frame_xform / gf_get_cubic_table do no GPU work: checked here against the oracle."""
    from gyroflow_amd import _lib, cv
    z = np.load(os.path.join(GOLDEN, "getmaps_calls.npz"))
    xf = _lib.make_frame_xform(z["c2_27k_to_1080p.K"], z["c2_27k_to_1080p.R"], z["c2_27k_to_1080p.P"])
    a = _lib.xform_to_array(xf)
    np.testing.assert_array_equal(a[:9].reshape(3, 3), orc.compute_iR(z["c2_27k_to_1080p.R"], z["c2_27k_to_1080p.P"]))
    K = z["c2_27k_to_1080p.K"]
    assert a[9:13].tolist() == [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    assert a.size == 13
    with pytest.raises(_lib.GfError) as e:
        _lib.make_frame_xform(np.eye(3), np.zeros((3, 3)), np.eye(3))
    assert e.value.code == -4 and "singular" in str(e.value)
    np.testing.assert_array_equal(cv.cubic_table(), orc.cubic_tab()[1])


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle."""
    pkg = os.path.join(ROOT, "gyroflow_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            with open(os.path.join(pkg, fn)) as f:
                src = f.read()
            assert "oracle" not in src.replace("oracle's", ""), fn
    with open(os.path.join(pkg, "csrc", "gfremap.hip")) as f:
        assert "cv_oracle" not in f.read()


def test_missing_library_fails_loudly(monkeypatch):
    from gyroflow_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgfremap.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_golden_quaternion_fixtures():
    """The reference-generated quaternion traces the benchmarks use."""
    q = np.load(os.path.join(GOLDEN, "quats_rollpitchyaw.npz"))["stab_transform"]
    assert q.shape == (300, 4)
    np.testing.assert_allclose(q[0], [0.999999999997, -1.023058603972e-10, -5.589539942008e-11, -2.454242589352e-06], atol=1e-12)
    np.testing.assert_allclose(q[150], [0.9988806643244, 0.04730000819578, -3.572948769755e-04, 1.974852067112e-06], atol=1e-12)
    g = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))
    st = g["stab_transform"]
    assert st.shape == (1622, 7) and float(g["min_fcorr"]) == 2.1857348445009372
    ang = 2 * np.degrees(np.arccos(np.clip(np.abs(st[:, 3] / np.linalg.norm(st[:, 3:7], axis=1)), 0, 1)))
    assert 3.0 < ang.mean() < 4.2 and 12.5 < ang.max() < 14.0          # mean 3.6 deg, max 13.4 deg


def test_lens_table_accuracy():
    """The kernels evaluate the lens model's radial scale S(t) = theta_d(atan sqrt t)/sqrt t from a 256-piece degree-6
    table (gfremap.hip).  Against an 80-bit evaluation of the formula, the distorted radius theta_d = sqrt(t) S(t) (the
    quantity that is multiplied by the focal length, ~2000 px) must agree to 3e-14 absolute on the whole table range
    t in [0, 255), for every committed preset's coefficients and for D = 0: < 1e-10 px on the map coordinates (the
    north_star tolerance is 1e-3 px).  S itself is checked to 2e-14 relative wherever |S| > 0.01 (real presets make
    theta_d cross zero beyond ~86 degrees, where a relative bound on S is meaningless)."""
    from gyroflow_amd import _lib
    rng = np.random.default_rng(0)
    t = np.concatenate([[0.0, 1e-300, 1e-12, 254.99999], rng.random(200000) * 4, 2.0 ** (rng.random(100000) * 8) - 1,
                        np.nextafter(2.0 ** np.arange(1, 8), 0) - 1, 2.0 ** np.arange(1, 8) - 1])
    tl = t.astype(np.longdouble)
    r = np.sqrt(tl); th = np.arctan(r)
    Ds = [np.zeros(4)]
    for key in ("hero8_27k_43", "hero7_4k_169", "hero9_4k_43", "hero9_5k_169", "runcam5_stretched"):
        Ds.append(np.asarray(_cal(key).D, np.float64).reshape(-1))
    for D in Ds:
        Dl = D.astype(np.longdouble)
        th2 = th * th
        want = np.where(tl > 0, th * (1 + th2 * (Dl[0] + th2 * (Dl[1] + th2 * (Dl[2] + th2 * Dl[3])))) / np.where(tl > 0, r, 1), 1)
        got = _lib.lens_scale_host(D, t)
        err_thd = np.abs((got.astype(np.longdouble) - want) * r)
        assert float(err_thd.max()) < 3e-14, (D, float(err_thd.max()))
        big = np.abs(want) > 0.01
        rel = np.abs((got.astype(np.longdouble) - want) / want)[big]
        assert float(rel.max()) < 2e-14, (D, float(rel.max()))
    assert np.isnan(_lib.lens_scale_host(Ds[1], [255.0, 1e6, np.inf, np.nan, -2.5])).all()      # outside: exact path


def test_adaptive_zoom_host_math_matches_per_frame_restatement(orc):
    """The batched findFcorr/findFov arithmetic (numpy, no GPU) against the per-frame restatement with scipy's interp1d,
    on polygons produced by the oracle's undistortPoints."""
    from oracle import ref_adaptive_zoom as raz
    from gyroflow_amd import FisheyeCalibrator
    from gyroflow_amd.adaptive_zoom import AdaptiveZoom
    z = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero9_4k_43"))
    az = AdaptiveZoom(cal)
    q = z["stab_transform"][::25, 3:7]
    polys = np.stack([raz.bounding_polygon(orc, cal.K, cal.D, cal.calib_dimension, qq) for qq in q])
    centers = np.tile([2000.0, 1500.0], (len(q), 1))
    for od in [(1920, 1080), (1080, 1920), (1440, 1080)]:
        got = az._find_fov_batch(centers, polys, od)
        want = np.array([raz.find_fov(c, p, od) for c, p in zip(centers, polys)])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
        f0, i0 = az.findFcorr(centers[3], polys[3], od)
        f1, i1 = raz.find_fcorr(centers[3], polys[3], od)
        assert f0 == f1 and i0 == i1
        assert az.findFov(centers[3], polys[3], od) == pytest.approx(want[3], rel=1e-12)
    a = np.arange(10.0)[::-1].copy()
    np.testing.assert_array_equal(az.min_rolling(a, 3), a[2:])
    np.testing.assert_array_equal(az.min_rolling(np.array([3.0, 1.0, 4.0, 1.0, 5.0, 9.0, 2.0, 6.0]), 3), [1, 1, 1, 1, 2, 2])
    # the boundary point list: 8 points per edge, clockwise from the top-left corner (adaptive_zoom.py:199-216)
    p = az._distorted_points(9)
    assert p.shape == (32, 2) and tuple(p[0]) == (0.0, 0.0) and tuple(p[8]) == (4000.0, 0.0) and tuple(p[16]) == (4000.0, 3000.0)
    assert tuple(p[24]) == (0.0, 3000.0) and tuple(p[31]) == (0.0, 375.0) and tuple(p[1]) == (500.0, 0.0)


def test_raw_video_adapters_roundtrip(tmp_path):
    """frameio.RawVideoSource / RawVideoSink: headerless bgr24 / bgr48le files, cv2.VideoCapture-like positions."""
    from gyroflow_amd import RawVideoSource, RawVideoSink, synthetic_frames
    for dtype in (np.uint8, np.uint16):
        frames = synthetic_frames(40, 24, n=5, dtype=dtype, seed=2)
        path = str(tmp_path / ("clip_%s.raw" % np.dtype(dtype).name))
        sink = RawVideoSink(path)
        for f in frames:
            sink.write(f)
        sink.close()
        assert sink.count == 5 and os.path.getsize(path) == 5 * 40 * 24 * 3 * np.dtype(dtype).itemsize
        src = RawVideoSource(path, 40, 24, dtype)
        assert src.count == 5
        src.set_pos(2)
        assert src.get_pos() == 2
        ok, f = src.read()
        assert ok and src.get_pos() == 3 and np.array_equal(f, frames[2])
        buf = np.zeros((4, 24, 40, 3), dtype)
        assert src.read_into(buf, 4) == 2 and np.array_equal(buf[1], frames[4])
        ok, f = src.read()
        assert not ok and f is None
        src.release()
    with pytest.raises(ValueError):
        RawVideoSource(path, 4000, 3000, np.uint16)


def test_quaternion_pipeline_reproduces_reference_capture():
    """GyroIntegrator + PlainSlerp (host code in libgfremap) against the trace the reference's own gyro_integrator /
    smoothing_algos produced for the FakeData('rollpitchyaw') gyro (tests/golden/make_golden.py)."""
    from gyroflow_amd import GyroIntegrator, PlainSlerp
    z = np.load(os.path.join(GOLDEN, "quats_rollpitchyaw.npz"))
    g = GyroIntegrator(z["gyro"], zero_out_time=False, initial_orientation=z["initial_orientation"])
    t_all, q_all = g.integrate_all()
    assert q_all.shape == (1000, 4) and np.allclose(np.linalg.norm(q_all, axis=1), 1, atol=1e-15)
    g.set_smoothing_algo(PlainSlerp())
    t, s = g.get_interpolated_stab_transform(start=0, interval=1 / 30)
    np.testing.assert_array_equal(t, z["times"])
    np.testing.assert_allclose(s, z["stab_transform"], rtol=0, atol=1e-13)


def test_quaternion_pipeline_reproduces_gyroflow_sample():
    """Real data: raw_imu of plugins/blender/GX016015.MP4.gyroflow -> frame_orientation (stored at full precision) and
    the stab_transform quaternions (stored to 5 decimals).  raw_imu itself is rounded to 5 decimals, hence ~1e-5."""
    from scipy.spatial.transform import Rotation
    from gyroflow_amd import GyroIntegrator
    z = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))
    fps = float(z["fps"])
    init = Rotation.from_euler('zxy', [0, 0, np.pi / 2]).as_quat()[[3, 0, 1, 2]]        # stabilizer.py:1690-1691
    g = GyroIntegrator(z["raw_imu"][:, [0, 1, 2, 3]], zero_out_time=False, initial_orientation=init)
    g.integrate_all()
    t, o = g.get_interpolated_orientations(start=0, interval=1 / fps)
    fo, st = z["frame_orientation"], z["stab_transform"]
    assert len(t) == len(fo) == 1622
    np.testing.assert_allclose(t, fo[:, 1], rtol=0, atol=1e-12)
    assert np.abs(o - fo[:, 2:]).max() < 2e-5
    _, s = g.get_interpolated_stab_transform(start=0, interval=1 / fps)
    assert np.abs(s - st[:, 3:7]).max() < 2e-5


def test_quaternion_host_code_matches_python_helpers():
    """csrc/quat_host.cpp against gyroflow_amd/quaternion.py (numpy, the reference's formulas) on random data,
    including the slerp branches (near-parallel, obtuse) and resampling with gaps / a negative start."""
    from gyroflow_amd import GyroIntegrator, PlainSlerp, quaternion as quat
    rng = np.random.default_rng(8)
    n = 400
    tt = np.cumsum(rng.uniform(0.002, 0.02, n)); tt[150:] += 0.4               # a gap in the gyro data
    gyro = np.column_stack([tt, rng.normal(0, 3.0, (n, 3))])
    gyro[20:30, 1:] = 0                                                        # samples without rotation
    g = GyroIntegrator(gyro, zero_out_time=False)
    _, q = g.integrate_all()
    ori = np.array([1, 0.0001, 0.0001, 0.0001])
    for i in range(n):
        last = gyro[i - 1, 0] if i > 0 else gyro[i, 0]
        nxt = gyro[i + 1, 0] if i < n - 1 else gyro[i, 0]
        if np.any(gyro[i, 1:]):
            ha = gyro[i, 1:] * ((nxt - last) / 2) * 0.5
            l = np.sqrt(ha.dot(ha))
            dq = quat.normalize(quat.quaternion(np.cos(l), *(ha * (np.sin(l) / l)))) if l > 1e-12 else quat.quaternion(1, 0, 0, 0)
            ori = quat.normalize(quat.quaternion_multiply(ori, dq))
        np.testing.assert_allclose(q[i], ori, rtol=0, atol=1e-14)
    t, o = g.get_interpolated_orientations(start=-0.05, interval=1 / 60)
    assert t[0] == -0.05 and np.all(o[t <= tt[0]] == q[0])
    k = np.searchsorted(tt, t[200], side="right") - 1
    w = (t[200] - tt[k]) / (tt[k + 1] - tt[k])
    np.testing.assert_allclose(o[200], quat.single_slerp(q[k], q[k + 1], w), rtol=0, atol=1e-14)
    ps = PlainSlerp(); ps.set_user_option("smoothness", 0.05)
    _, sm = ps.get_smooth_orientations(t, o)
    alpha = 1 - np.exp(-(1 / (len(t) / (t[-1] - t[0]))) / 0.05)
    v = o[0]; fwd = []
    for i in range(len(t)):
        v = quat.single_slerp(v, o[i], alpha); fwd.append(v)
    v = fwd[-1]; want = [None] * len(t)
    for i in range(len(t) - 1, -1, -1):
        v = quat.single_slerp(v, fwd[i], alpha); want[i] = v
    np.testing.assert_allclose(sm, np.array(want), rtol=0, atol=1e-13)
    far = quat.single_slerp([1, 0, 0, 0], [0, 1, 0, 0], 0.5)
    np.testing.assert_allclose(far, [0.70710678, 0.70710678, 0, 0], atol=1e-8)      # SURVEY.md 8c verified value


def test_rp_table_and_interpolation_error():
    """Constants of the RP coordinate engine (gfremap.hip: build_rp_table) and the accuracy claim behind its tile guard:
    M inverts the Vandermonde matrix of the 8 Chebyshev nodes, the row weights are the Lagrange basis at the 64 tile rows
    (partition of unity, exact on polynomials up to degree 7), and the tensor-product interpolant of a function whose
    nearest singularity is 8 half-tiles away (the guard's limit) is good to 1e-9 relative -- 1e-13 at 55 half-tiles,
    a 1080p frame's geometry."""
    from gyroflow_amd import _lib
    M, z, Ly = _lib.rp_table()
    assert np.allclose(np.sort(z), np.sort(np.cos(np.pi * (2 * np.arange(8) + 1) / 16)), atol=1e-16)
    V = z[:, None] ** np.arange(8)[None, :]
    assert np.abs(M @ V - np.eye(8)).max() < 1e-12
    eta = (np.arange(64) - 31.5) / 32
    assert np.abs(Ly.sum(1) - 1).max() < 1e-14
    for d in range(8):
        assert np.abs(Ly @ z ** d - eta ** d).max() < 1e-13
    # the kernel's pipeline in numpy: node values -> per node row coefficients (M) -> per tile row (Ly) -> Horner in z
    zz = (np.arange(64) - 31.5) / 32
    for dist, tol in ((8.0, 1e-9), (55.0, 1e-13)):
        def G(x, y):                      # analytic, singular at x^2 + y^2 = -dist^2 (x, y in half tiles): like 1/sqrt(q)
            return 1.0 / np.sqrt(1.0 + ((x + 0.3) ** 2 + (y - 0.2) ** 2) / dist ** 2)
        nodes = G(z[None, :], z[:, None])                                  # [node row m][node column a]
        C2 = nodes @ M.T                                                   # [m][power]
        C = Ly @ C2                                                        # [tile row][power]
        got = sum(C[:, k][:, None] * zz[None, :] ** k for k in range(8))
        want = G(zz[None, :], eta[:, None])
        assert np.abs(got / want - 1).max() < tol, (dist, np.abs(got / want - 1).max())


def test_renderfile_early_exits_need_no_gpu(capsys):
    """The reference's renderfile prints and returns for a bad file extension (stabilizer.py:1230-1233); the mirror does the same
    for that, for debug_text (cv2 drawing, not available) and for a sharded HISTORY render -- before any GPU work."""
    import inspect
    from gyroflow_amd import Stabilizer, ArraySource
    frames = [np.zeros((4, 4, 3), np.uint8)] * 4
    st = Stabilizer(ArraySource(frames), preset_path("hero8_27k_43"), np.tile([1.0, 0, 0, 0], (4, 1)), 10.0, 4, 4)
    assert st.renderfile(0, 1, "movie.xyz") is None
    assert "does not have a valid file extension" in capsys.readouterr().out
    assert st.renderfile(0, 1, "/nonexistent_dir/movie.mp4", debug_text=True) is None
    assert "debug_text" in capsys.readouterr().out
    assert st.renderfile(0, 1, "/nonexistent_dir/movie.mp4", bg_color="HISTORY", world=2) is None
    assert "does not shard" in capsys.readouterr().out
    names = list(inspect.signature(Stabilizer.renderfile).parameters)
    assert names[:19] == ["self", "starttime", "stoptime", "outpath", "out_size", "split_screen", "bitrate_mbits", "display_preview",
                          "scale", "vcodec", "vprofile", "pix_fmt", "debug_text", "custom_ffmpeg", "smoothingFocus", "fov_scale",
                          "bg_color", "audio", "viewer_thread"]
    d = inspect.signature(Stabilizer.renderfile).parameters
    assert d["outpath"].default == "Stabilized.mp4" and d["smoothingFocus"].default == 4.0 and d["split_screen"].default is False


def test_numa_pinning_helpers(tmp_path):
    """dist.pin_to_numa_node reads a node's cpulist from sysfs and intersects it with the current affinity mask; absent
    files or an empty intersection change nothing."""
    from gyroflow_amd import dist
    assert dist._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    before = os.sched_getaffinity(0)
    assert dist.pin_to_numa_node(None) is None
    assert dist.pin_to_numa_node(3, sysfs=str(tmp_path)) is None                 # no such node
    nd = tmp_path / "devices" / "system" / "node" / "node0"
    nd.mkdir(parents=True)
    (nd / "cpulist").write_text("100000-100003\n")
    assert dist.pin_to_numa_node(0, sysfs=str(tmp_path)) is None                 # empty intersection: unchanged
    assert os.sched_getaffinity(0) == before
    one = min(before)
    (nd / "cpulist").write_text("%d\n" % one)
    try:
        assert dist.pin_to_numa_node(0, sysfs=str(tmp_path)) == {one} and os.sched_getaffinity(0) == {one}
    finally:
        os.sched_setaffinity(0, before)


def test_library_has_no_forbidden_instruction():
    """The build refuses device code that contains v_ashr_pk_u8_i32 (gyroflow_amd/csrc/Makefile); check the library that ships."""
    import shutil
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not installed")
    from gyroflow_amd import _lib
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(_lib.LIB_PATH, os.path.join(d, "lib.so"))
        subprocess.check_call([objdump, "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL)
        cos = [f for f in os.listdir(d) if "amdgcn" in f]
        assert cos, "no gfx950 code object in the library"
        dis = subprocess.check_output([objdump, "-d", os.path.join(d, cos[0])]).decode()
    assert "v_dot4_u32_u8" in dis            # the disassembly is the device code of the fused kernel
    assert "v_ashr_pk_u8_i32" not in dis


def test_threaded_staging_copy_equals_plain_copy():
    """pipeline._host_copy (the decoder-output -> pinned-slot copy, row bands on a thread pool) for frames above and below its
    size threshold, both depths, a source with padded rows and a row count that does not divide by the thread count."""
    from gyroflow_amd import pipeline
    rng = np.random.default_rng(3)
    for shape, dt in (((1081, 1920, 3), np.uint8), ((37, 53, 3), np.uint8), ((1500, 1000, 3), np.uint16)):
        src = rng.integers(0, np.iinfo(dt).max, shape).astype(dt)
        dst = np.zeros(shape, dt)
        pipeline._host_copy(dst, src)
        np.testing.assert_array_equal(dst, src)
        padded = np.zeros((shape[0], shape[1] + 7, 3), dt)
        padded[:, :shape[1]] = src
        dst[...] = 0
        pipeline._host_copy(dst, padded[:, :shape[1]])
        np.testing.assert_array_equal(dst, src)


def test_gyro_integrator_acc_scaling_quirk(capsys):
    """The constructor's data preparation (/root/reference/gyro_integrator.py:27-71), case by case: with a usable accelerometer
    log the reference multiplies the GYRO time column by time_scaling and the gyro rates by acc_scaling (:40-41) before the
    regular scaling (:56-57) applies time_scaling again; a log of another length is refused with a message; the log is cut
    after the first timestamp that runs backwards; zero_out_time rebases both logs."""
    from gyroflow_amd import GyroIntegrator
    t = np.array([10.0, 10.5, 11.0, 11.5, 12.0])
    gyro = np.column_stack([t, np.arange(15.0).reshape(5, 3) + 1])
    acc = np.column_stack([t, np.ones((5, 3))])
    g = GyroIntegrator(gyro, time_scaling=2, gyro_scaling=3, acc_data=acc, acc_scaling=5, zero_out_time=False)
    np.testing.assert_array_equal(g.gyro[:, 0], t * 2 * 2)                       # time_scaling twice
    np.testing.assert_array_equal(g.gyro[:, 1:], gyro[:, 1:] * 5 * 3)            # acc_scaling lands on the gyro rates
    np.testing.assert_array_equal(g.acc, acc)                                    # and the accelerometer log is left as it came
    assert g.acc_available and g.gyro_sample_rate == 5 / (48.0 - 40.0)
    g = GyroIntegrator(gyro, time_scaling=2, gyro_scaling=3, zero_out_time=True)
    np.testing.assert_array_equal(g.gyro[:, 0], (t - 10.0) * 2)
    np.testing.assert_array_equal(g.gyro[:, 1:], gyro[:, 1:] * 3)
    assert g.acc is None and not g.acc_available
    np.testing.assert_array_equal(g.initial_orientation, [1, 0.0001, 0.0001, 0.0001])
    capsys.readouterr()
    g = GyroIntegrator(gyro, acc_data=acc[:4])
    assert capsys.readouterr().out == "Gyro and acceleration data don't line up\n" and not g.acc_available and g.acc is None
    bad = gyro.copy(); bad[3, 0] = 10.2                                          # runs backwards after sample 2
    g = GyroIntegrator(bad, acc_data=acc, zero_out_time=True)
    assert capsys.readouterr().out == "Truncated bad gyro data\n"
    assert g.num_data_points == 3 and g.acc.shape == (3, 4) and g.gyro[0, 0] == 0 and g.acc[0, 0] == 0 and g.gyro[2, 0] == 1.0
    g = GyroIntegrator(gyro, initial_orientation=[0, 1, 0, 0])
    np.testing.assert_array_equal(g.orientation, [0, 1, 0, 0])
    assert g.orientation is not g.initial_orientation and gyro[0, 0] == 10.0     # the caller's array is not modified
