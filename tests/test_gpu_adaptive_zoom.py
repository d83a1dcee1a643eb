"""AdaptiveZoom (adaptive_zoom.py:88-228) and cv2.fisheye.undistortPoints on the GPU against the per-frame restatement in
oracle/ref_adaptive_zoom.py (oracle undistortPoints + scipy interp1d).  float64 throughout; the only non-bit-exact
operation is tan() (device libm vs glibc), so the tolerance is 1e-9 px on points and 1e-10 relative on fov values."""
import os

import numpy as np
import pytest

from oracle import ref_adaptive_zoom as raz
from conftest import GOLDEN, preset_path

pytestmark = pytest.mark.gpu


def _sample():
    z = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))
    return z, z["stab_transform"]


def _sample_calibration(z):
    from gyroflow_amd import FisheyeCalibrator
    cal = FisheyeCalibrator()
    cal.load_calibration_data(dict(calib_dimension=dict(w=4000, h=3000), orig_dimension=dict(w=4000, h=3000), input_horizontal_stretch=1,
                                   use_opencv_fisheye=True, calibrator_version="0.3.0-beta", num_images=1, name="gyroflow sample", camera_brand="", camera_model="", lens_model="", camera_setting="", note="", calibrated_by="", date="",
                                   fisheye_params=dict(camera_matrix=z["K"].tolist(), distortion_coeffs=z["D"].tolist(), RMS_error=0.0)))
    return cal


def test_undistort_points_vs_oracle(gpu, orc):
    from gyroflow_amd import cv
    z, st = _sample()
    K, D = z["K"], z["D"]
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-500, 4500, (400, 2)), [[2000.0, 1500.0], [2000.0 + 1e-7, 1500.0], [0, 0], [4000, 3000]]])
    from scipy.spatial.transform import Rotation
    R = Rotation.from_euler("zyx", [3.0, -8.0, 5.0], degrees=True).as_matrix()
    got = cv.fisheye_undistortPoints(pts[None], K, D, R=R, P=K, device=gpu)
    assert got.shape == (1, len(pts), 2)
    want = orc.undistort_points(pts, K, D, R, K)
    np.testing.assert_allclose(got[0], want, rtol=0, atol=1e-9)
    # no R, no P: normalised coordinates
    got2 = cv.fisheye_undistortPoints(pts, K, D, device=gpu)
    want2 = orc.undistort_points(pts, K, D, np.eye(3), np.eye(3))
    np.testing.assert_allclose(got2, want2, rtol=0, atol=1e-12)


def test_undistort_points_not_converged_marker(gpu, orc):
    """A lens whose theta_d polynomial is not monotonic: points the Newton loop cannot invert come back as (-1e6,-1e6)
    in the oracle and on the GPU alike."""
    from gyroflow_amd import cv
    K = np.array([[300.0, 0, 320], [0, 300.0, 240], [0, 0, 1]])
    D = np.array([-0.9, 0.5, -0.3, 0.1])
    xs = np.linspace(-400, 1000, 57)
    pts = np.stack([xs, 0.7 * xs + 11], axis=1)
    got = cv.fisheye_undistortPoints(pts, K, D, R=np.eye(3), P=K, device=gpu)
    want = orc.undistort_points(pts, K, D, np.eye(3), K)
    bad_w = want[:, 0] == -1e6
    bad_g = got[:, 0] == -1e6
    assert bad_w.any() and not bad_w.all()
    np.testing.assert_array_equal(bad_g, bad_w)
    np.testing.assert_allclose(got[~bad_w], want[~bad_w], rtol=1e-9, atol=1e-7)


def test_bounding_polygons_batched(gpu, orc):
    from gyroflow_amd import AdaptiveZoom, FisheyeCalibrator
    z, st = _sample()
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero9_4k_43"))
    az = AdaptiveZoom(cal, device=gpu)
    q = st[::40, 3:7]
    polys = az.boundingPolygons(q)
    assert polys.shape == (len(q), 32, 2)
    for i in (0, 7, len(q) - 1):
        want = raz.bounding_polygon(orc, cal.K, cal.D, cal.calib_dimension, q[i])
        np.testing.assert_allclose(polys[i], want, rtol=0, atol=1e-9)
        np.testing.assert_allclose(az.boundingPolygon(q[i]), want, rtol=0, atol=1e-9)
    np.testing.assert_allclose(az.boundingPolygon(None), raz.bounding_polygon(orc, cal.K, cal.D, cal.calib_dimension, None), atol=1e-9)


@pytest.mark.parametrize("focus,win", [(2.0, (False, False)), (0.5, (100, 400)), (-1, (False, False)), (-2, (False, False)), (-1, (30, 90))])
def test_compute_matches_reference_restatement(gpu, orc, focus, win):
    from gyroflow_amd import AdaptiveZoom, FisheyeCalibrator
    z, st = _sample()
    cal = _sample_calibration(z)
    q = st[:600, 3:7]
    fps = float(z["fps"])
    az = AdaptiveZoom(cal, device=gpu)
    fov, centers = az.compute(q, (1920, 1080), fps, smoothingFocus=focus, tstart=win[0], tend=win[1])
    want, wc, _ = raz.compute(orc, z["K"], z["D"], (4000, 3000), q, (1920, 1080), fps, smoothingFocus=focus, tstart=win[0], tend=win[1])
    assert fov.shape == want.shape == (600,)
    np.testing.assert_allclose(fov, want, rtol=1e-10, atol=0)
    np.testing.assert_array_equal(centers, wc)


def test_compute_reproduces_gyroflow_sample_column(gpu):
    """Same known answer as tests/test_oracle.py::test_undistort_points_pinned_by_gyroflow_sample, through the product."""
    from gyroflow_amd import AdaptiveZoom, FisheyeCalibrator
    z, st = _sample()
    cal = _sample_calibration(z)
    fov, _ = AdaptiveZoom(cal, device=gpu).compute(st[:, 3:7], (1920, 1080), float(z["fps"]), smoothingFocus=2.0)
    assert np.abs(fov - st[:, 2]).max() < 1.5e-4
    assert abs(fov.min() - float(z["min_fcorr"])) < 5e-5


def test_renderfile_with_adaptive_zoom(gpu, orc):
    """renderfile with the reference's default smoothingFocus: per-frame fov = fov_scale*fcorr[n] feeds get_maps
    (stabilizer.py:1423-1427).  Expected frames from the oracle with fcorr from the per-frame restatement."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, synthetic_frames
    from test_gpu_stabilizer import _quats
    w, h = 338, 254
    n = 24
    frames = synthetic_frames(w, h, n=n, seed=11)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    stab = _quats(n + 3, seed=9)
    fps, start, stop, out_size = 10.0, 0.3, 2.2, (240, 136)
    st = Stabilizer(ArraySource(frames), cal, stab, fps, w, h, device=gpu)
    sink = ListSink()
    nw = st.renderfile(start, stop, sink, out_size=out_size, fov_scale=1.1, smoothingFocus=0.7, bg_color="#203040", batch=4)
    num_frames = int((stop - start) * fps); tstart = int(start * fps)
    fcorr, centers, _ = raz.compute(orc, cal.K, cal.D, cal.calib_dimension, stab, out_size, fps, smoothingFocus=0.7,
                                    tstart=tstart, tend=tstart + num_frames)
    assert nw == num_frames - 1 == len(sink.frames)
    nbad = 0
    for k, got in enumerate(sink.frames):
        fn = tstart + k
        K, D, R, P, od = cal._frame_matrices(1.1 * fcorr[fn], out_size, (w, h), stab[fn], centers[fn], True)
        m1, m2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2)
        want = orc.remap_linear(frames[fn], m1, m2, borderMode=0, borderValue=[64, 48, 32])
        nbad += int((got != want).sum())
    # fcorr agrees to ~1e-13 relative, so a map entry can flip only on an exact rounding tie
    assert nbad <= 6
