"""CPU tests of the oracle (oracle/cv_oracle.c): known answers derived by hand from the published OpenCV 4.5.2
algorithm, one test per documented quirk (Q1..Q6 in the C file), and the committed golden vectors.

PARITY UNPINNED: nothing in the reference's own files pins a map entry or an output pixel of cv2, and cv2 is not
importable here, so these tests pin the oracle to the algorithm as restated in SURVEY.md 8(c), not to a cv2 binary.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN


# ------------------------------------------------------------------ rounding primitives
def test_cvround_half_even_and_overflow(orc):
    assert [orc.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, -2.5001)] == [0, 2, 2, 0, -2, 2, -3]
    assert orc.cv_round(float("inf")) == -2**31 and orc.cv_round(float("nan")) == -2**31 and orc.cv_round(3e9) == -2**31
    assert orc.cv_round_f(0.5) == 0 and orc.cv_round_f(1.5) == 2 and orc.cv_round_f(1e20) == -2**31


# ------------------------------------------------------------------ interpolation tables
def test_linear_table_q4(orc):
    f, i = orc.linear_tab()
    assert i[0].tolist() == [32767, 0, 0, 1]                       # Q4: saturate + fix-up on the bottom-right tap
    fy, fx = np.divmod(np.arange(1024), 32)
    want = np.stack([(32 - fy) * (32 - fx), (32 - fy) * fx, fy * (32 - fx), fy * fx], 1) * 32
    np.testing.assert_array_equal(i[1:], want[1:])
    assert (i.sum(1) == 32768).all()
    np.testing.assert_array_equal(f, (want / 32768.0).astype(np.float32))


def test_linear_table_irregular_entry_is_invisible_in_u8():
    """The GPU kernel uses the closed form (32-fy)(32-fx).. for every entry; for a == 0 the table's {32767,0,0,1}
    yields the same byte for every (p00, p11) pair."""
    p00, p11 = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    assert (((p00 * 32767 + p11 + 16384) >> 15) == p00).all()


def test_cubic_table_q6(orc):
    f, i = orc.cubic_tab()
    assert (i.sum(1) == 32768).all()
    e0 = np.zeros(16, np.int16); e0[5] = 32767; e0[10] = 1          # fix-up searches rows/cols {2,3} only
    assert i[0].tolist() == e0.tolist()
    # 1-D coefficients at t = 0.5: (-0.09375, 0.59375, 0.59375, -0.09375); outer product * 32768 is exact
    c = np.array([-0.09375, 0.59375, 0.59375, -0.09375])
    np.testing.assert_array_equal(i[16 * 32 + 16].reshape(4, 4), np.rint(np.outer(c, c) * 32768).astype(np.int16))
    np.testing.assert_allclose(f.reshape(32, 32, 4, 4).sum((2, 3)), 1.0, atol=1e-6)


# ------------------------------------------------------------------ fisheye maps
def test_fisheye_known_answers(orc):
    """Equidistant model with D = 0: a ray at angle theta lands at c + f*theta.  Pixel (cx + f*tan(theta), cy) of the
    pinhole output therefore maps to u = cx + f*theta, v = cy."""
    th = 0.5
    f = 100.0 / np.tan(th)
    K = np.array([[f, 0, 200.0], [0, f, 150.0], [0, 0, 1]])
    mx, my = orc.fisheye_init_undistort_rectify_map(K, np.zeros(4), np.eye(3), K, (400, 300), orc.CV_32FC1)
    assert mx[150, 200] == 200.0 and my[150, 200] == 150.0             # Q3: r == 0 -> scale 1
    assert abs(mx[150, 300] - (200 + f * th)) < 1e-4 and my[150, 300] == 150.0
    assert abs(my[250, 200] - (150 + f * th)) < 1e-4
    m1, m2 = orc.fisheye_init_undistort_rectify_map(K, np.zeros(4), np.eye(3), K, (400, 300), orc.CV_16SC2)
    q = int(np.rint((200 + f * th) * 32))
    assert m1[150, 300].tolist() == [q >> 5, 150] and m2[150, 300] == (q & 31)
    # with k1: theta_d = theta (1 + k1 theta^2)
    mx2, _ = orc.fisheye_init_undistort_rectify_map(K, [0.1, 0, 0, 0], np.eye(3), K, (400, 300), orc.CV_32FC1)
    assert abs(mx2[150, 300] - (200 + f * th * (1 + 0.1 * th * th))) < 1e-4


def test_fisheye_fixed_equals_rounded_float(orc, golden_calls):
    """CV_16SC2 output == Q5 rounding of the f64 coordinates (checked through the float32 maps away from ties)."""
    z, _ = golden_calls
    c = "survey_capture"
    args = (z[c + ".K"], z[c + ".D"], z[c + ".R"], z[c + ".P"], (480, 270))
    m1, m2 = orc.fisheye_init_undistort_rectify_map(*args, orc.CV_16SC2)
    fx, fy = orc.fisheye_init_undistort_rectify_map(*args, orc.CV_32FC1)
    qx = (m1[..., 0].astype(np.int64) << 5) + (m2 & 31)
    qy = (m1[..., 1].astype(np.int64) << 5) + (m2 >> 5)
    assert np.abs(qx - fx.astype(np.float64) * 32).max() <= 0.5 + 1e-2
    assert np.abs(qy - fy.astype(np.float64) * 32).max() <= 0.5 + 1e-2


def test_fisheye_behind_camera_q2(orc):
    from scipy.spatial.transform import Rotation
    K = np.array([[300., 0, 160], [0, 300., 120], [0, 0, 1]])
    R = Rotation.from_euler("y", 180, degrees=True).as_matrix()
    m1, m2 = orc.fisheye_init_undistort_rectify_map(K, np.zeros(4), R, K, (64, 48), orc.CV_16SC2)
    assert (m1 == 0).all() and (m2 == 0).all()                          # cvRound(+-inf) = INT_MIN -> (short)(INT_MIN>>5) = 0
    m1b, _ = orc.fisheye_init_undistort_rectify_map(K, np.zeros(4), R, K, (64, 48), orc.CV_16SC2, wcheck=False)
    assert not (m1b == 0).all()                                         # without the branch the ray is mirrored instead


def test_inverse_and_row_start(orc):
    rng = np.random.default_rng(0)
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(rng.normal(0, 0.1, 3)).as_matrix()
    P = np.array([[1500., 0, 960], [0, 1490., 540], [0, 0, 1]])
    iR = orc.compute_iR(R, P)
    np.testing.assert_allclose(iR @ (P @ R), np.eye(3), atol=1e-12)
    with pytest.raises(ValueError):
        orc.compute_iR(np.zeros((3, 3)), P)


# ------------------------------------------------------------------ convertMaps
def test_convert_maps(orc):
    mx = np.array([[0.0, 0.015625, 0.046875, 1.5, -0.25, 2000.0, -2000.0, 1e9]], np.float32)
    my = np.array([[10.96875, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0]], np.float32)
    m1, m2 = orc.convert_maps(mx, my)
    # 0.015625*32 = 0.5 -> 0 (half-even); 0.046875*32 = 1.5 -> 2; -0.25*32 = -8 -> (-8>>5 = -1, frac 24)
    assert m1[0, :, 0].tolist() == [0, 0, 0, 1, -1, 2000, -2000, -32768]     # saturate_cast<short>, 1e9*32 -> INT_MIN
    assert (m2[0] & 31).tolist() == [0, 0, 2, 16, 24, 0, 0, 0]
    assert m1[0, 0, 1] == 10 and (m2[0, 0] >> 5) == 31
    mx = np.array([[1100.0, -1100.0]], np.float32)
    m1, _ = orc.convert_maps(mx * 32, mx * 32)
    assert m1[0, :, 0].tolist() == [32767, -32768]


# ------------------------------------------------------------------ bilinear remap
def _maps(dh, dw, x, y, fx=0, fy=0):
    m1 = np.empty((dh, dw, 2), np.int16); m1[..., 0] = x; m1[..., 1] = y
    m2 = np.full((dh, dw), fy * 32 + fx, np.uint16)
    return m1, m2


def test_remap_linear_identity_shift_and_half_pixel(orc):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:20, 0:30]
    out = orc.remap_linear(src, *_maps(20, 30, xx, yy))
    np.testing.assert_array_equal(out, src)                                  # identity, incl. last row/col (border path)
    out = orc.remap_linear(src, *_maps(20, 30, xx + 3, yy - 2), borderValue=(1, 2, 3))
    np.testing.assert_array_equal(out[2:, :-3], src[:-2, 3:])
    assert (out[:2] == [1, 2, 3]).all() and (out[:, -3:] == [1, 2, 3]).all()
    # half pixel in x: (a*16*32*32 + b*16*32*32 + 2^14) >> 15 = (a + b + 1) >> 1
    out = orc.remap_linear(src, *_maps(20, 30, xx, yy, fx=16))
    a = src[:, :-1].astype(int); b = src[:, 1:].astype(int)
    np.testing.assert_array_equal(out[:, :-1], (a + b + 1) >> 1)
    # last column blends with the constant border value 0
    np.testing.assert_array_equal(out[:, -1], (src[:, -1].astype(int) + 1) >> 1)


def test_remap_linear_borders(orc):
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (6, 7, 3), dtype=np.uint8)
    m1, m2 = _maps(1, 6, np.array([[-2, -1, 6, 7, 3, 3]]), np.array([[2, 2, 2, 2, -1, 5]]), fx=8, fy=8)
    prev = np.full((1, 6, 3), 77, np.uint8)
    c = orc.remap_linear(src, m1, m2, borderMode=orc.BORDER_CONSTANT, borderValue=(50, 60, 70))
    assert (c[0, 0] == [50, 60, 70]).all() and (c[0, 3] == [50, 60, 70]).all()     # fully outside -> cval
    s = src.astype(int); cv = np.array([50, 60, 70])
    w = np.array([24 * 24, 24 * 8, 8 * 24, 8 * 8]) * 32
    want1 = (cv * w[0] + s[2, 0] * w[1] + cv * w[2] + s[3, 0] * w[3] + 16384) >> 15  # sx = -1: left taps are cval
    assert (c[0, 1] == want1).all()
    r = orc.remap_linear(src, m1, m2, borderMode=orc.BORDER_REPLICATE)
    want = (s[2, 0] * (w[0] + w[1]) + s[3, 0] * (w[2] + w[3]) + 16384) >> 15         # clamped to column 0
    assert (r[0, 0] == want).all() and (r[0, 1] == want).all()
    want = (s[5, 3] * w[0] + s[5, 4] * w[1] + s[5, 3] * w[2] + s[5, 4] * w[3] + 16384) >> 15   # sy = 5: bottom row clamped
    assert (r[0, 5] == want).all()
    t = orc.remap_linear(src, m1, m2, dst=prev.copy(), borderMode=orc.BORDER_TRANSPARENT)
    assert (t == 77).all()                                                           # every pixel here has sx/sy outside [0, w-2]/[0, h-2]
    m1, m2 = _maps(1, 2, np.array([[5, 6]]), np.array([[4, 4]]))
    t = orc.remap_linear(src, m1, m2, dst=np.full((1, 2, 3), 77, np.uint8), borderMode=orc.BORDER_TRANSPARENT)
    assert (t[0, 0] == src[4, 5]).all() and (t[0, 1] == 77).all()                    # sx = w-2 is interpolated, w-1 is not
    with pytest.raises(ValueError):
        orc.remap_linear(src, m1, m2, borderMode=2)


def test_remap_linear_u16_float_path(orc):
    """u16 uses float32 weights, a left-to-right float sum and round-half-even (not the Q15 integer path)."""
    src = np.zeros((2, 2, 3), np.uint16)
    src[0, 0] = 65535; src[0, 1] = 1; src[1, 0] = 3; src[1, 1] = 2
    m1, m2 = _maps(1, 1, 0, 0, fx=1, fy=1)
    out = orc.remap_linear(src, m1, m2)
    w = np.float32([31 * 31, 31 * 1, 1 * 31, 1]) / np.float32(1024)
    t = np.float32(65535) * w[0]
    t = np.float32(t + np.float32(1) * w[1]); t = np.float32(t + np.float32(3) * w[2]); t = np.float32(t + np.float32(2) * w[3])
    assert out[0, 0, 0] == int(np.rint(t))
    # a tie: (1 + 2)/2 = 1.5 -> 2 ; (2 + 3)/2 = 2.5 -> 2 (half-even), the integer path would give 2 and 3
    src = np.array([[[1, 2, 0], [2, 3, 0]], [[0, 0, 0], [0, 0, 0]]], np.uint16)
    out = orc.remap_linear(src, *_maps(1, 1, 0, 0, fx=16, fy=0))
    assert out[0, 0, :2].tolist() == [2, 2]
    out = orc.remap_linear(src, *_maps(1, 1, 5, 5), borderValue=(70000, -5, 12.5))
    assert out[0, 0].tolist() == [65535, 0, 12]                                      # saturate_cast<ushort>(double)


def test_remap_threads_agree(orc):
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    m1 = rng.integers(-5, 90, (70, 90, 2)).astype(np.int16); m2 = rng.integers(0, 1024, (70, 90)).astype(np.uint16)
    a = orc.remap_linear(src, m1, m2, borderMode=1)
    b = orc.remap_linear(src, m1, m2, borderMode=1, nthreads=5)
    np.testing.assert_array_equal(a, b)


def test_fisheye_maps_threads_agree(orc):
    """Rows of initUndistortRectifyMap are independent: the row-striped variant (bench.py's "best CPU" leg) gives the same bits."""
    K = np.array([[420.0, 0, 318.5], [0, 421.0, 241.25], [0, 0, 1]]); D = [0.05, -0.01, 0.02, -0.01]
    a = np.deg2rad(4.0)
    R = np.array([[np.cos(a), -np.sin(a), 0.01], [np.sin(a), np.cos(a), -0.02], [0, 0, 1.0]])
    P = np.array([[400.0, 0, 320], [0, 400.0, 180], [0, 0, 1]])
    for mt in (orc.CV_16SC2, orc.CV_32FC1):
        s1, s2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, (333, 187), mt)
        t1, t2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, (333, 187), mt, nthreads=5)
        np.testing.assert_array_equal(s1, t1); np.testing.assert_array_equal(s2, t2)


# ------------------------------------------------------------------ bicubic remap
def test_remap_cubic_known_answers(orc):
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (12, 14, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:12, 0:14].astype(np.float32)
    out = orc.remap_cubic(src, xx, yy, borderMode=orc.BORDER_REPLICATE)
    np.testing.assert_array_equal(out, src)                       # integer coordinates: weight table entry 0 = {.., 32767, .., 1}
    # x + 0.5 along a row: 1-D kernel (-3, 19, 19, -3)/32 in Q15 (2-D entry = outer product with (0,1,0,0))
    out = orc.remap_cubic(src, xx + 0.5, yy, borderMode=orc.BORDER_REPLICATE)
    _, tab = orc.cubic_tab()
    wrow = tab[16].reshape(4, 4)[1].astype(int)
    assert wrow.tolist() == [-3072, 19456, 19456, -3072]
    s = src.astype(int)
    want = (s[5, 3] * wrow[0] + s[5, 4] * wrow[1] + s[5, 5] * wrow[2] + s[5, 6] * wrow[3] + 16384) >> 15
    np.testing.assert_array_equal(out[5, 4], np.clip(want, 0, 255))
    # constant border: fully outside -> cval, partially outside: cval*32768 + sum (S - cval)*w
    far = np.full((1, 1), -10, np.float32)
    assert orc.remap_cubic(src, far, far, borderValue=(4, 5, 6))[0, 0].tolist() == [4, 5, 6]


def test_remap_cubic_golden_stretch(orc):
    """nonlinear_stretch maps captured from the reference (tests/golden/stretch_maps.npz) drive the oracle without error and
    keep the centre column of a constant image constant (weights sum to 1)."""
    z = np.load(os.path.join(GOLDEN, "stretch_maps.npz"))
    src = np.full((300, 400, 3), 131, np.uint8)
    out = orc.remap_cubic(src, z["s0.map1"], z["s0.map2"])
    assert out.shape == (216, 384, 3)
    assert (out[10:-10, 20:-20] == 131).all()


# ------------------------------------------------------------------ undistortPoints / blend
def test_undistort_points_inverts_the_forward_model(orc):
    K = np.array([[1186., 0, 1352], [0, 1186., 1014], [0, 0, 1]]); D = np.array([0.04, 0.01, -0.02, 0.005])
    th = np.linspace(0.05, 1.2, 9); phi = np.linspace(0, 6, 9)
    thd = th * (1 + D[0] * th**2 + D[1] * th**4 + D[2] * th**6 + D[3] * th**8)
    pts = np.stack([K[0, 2] + K[0, 0] * thd * np.cos(phi), K[1, 2] + K[1, 1] * thd * np.sin(phi)], 1)
    und = orc.undistort_points(pts[None], K, D)[0]
    np.testing.assert_allclose(und, np.stack([np.tan(th) * np.cos(phi), np.tan(th) * np.sin(phi)], 1), atol=1e-7)
    und2 = orc.undistort_points(pts[None], K, D, R=np.eye(3), P=K)[0]
    np.testing.assert_allclose(und2, und * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]], atol=1e-4)


def test_estimate_new_camera_matrix_closed_forms(orc):
    """estimateNewCameraMatrixForUndistortRectify (calibrate_video.py:314) on lenses with a closed form -- parity unpinned, these
    pin the restatement to its own description: k = 0 makes undistortPoints the map r -> tan(r)."""
    w, h, f = 640, 480, 500.0
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]])
    P = orc.estimate_new_camera_matrix(K, np.zeros(4), (w, h), np.eye(3))
    fx, fy = w * 0.5 / np.tan(w / 2 / f), h * 0.5 / np.tan(h / 2 / f)          # edge mid-points land on the border
    assert fy > fx
    np.testing.assert_allclose(P, [[fy, 0, w / 2], [0, fy, h / 2], [0, 0, 1]], rtol=1e-12)       # balance 0: the larger focal length
    P1 = orc.estimate_new_camera_matrix(K, np.zeros(4), (w, h), np.eye(3), balance=1.0, fov_scale=2.0)
    np.testing.assert_allclose(P1, [[fx / 2, 0, w / 2], [0, fx / 2, h / 2], [0, 0, 1]], rtol=1e-12)
    P2 = orc.estimate_new_camera_matrix(K, np.zeros(4), (w, h), np.eye(3), balance=7.0, new_size=(320, 120))   # balance clamps to 1
    np.testing.assert_allclose(P2, [[fx * 0.5, 0, w / 4], [0, fx * 0.25, h / 8], [0, 0, 1]], rtol=1e-12)
    # aspect ratio fx/fy != 1: fy' = fx'/ratio, principal point stays at the centre for a centred lens
    Ka = np.array([[f, 0, w / 2], [0, f / 1.25, h / 2], [0, 0, 1]])
    Pa = orc.estimate_new_camera_matrix(Ka, np.zeros(4), (w, h), np.eye(3))
    np.testing.assert_allclose(Pa[1, 1], Pa[0, 0] / 1.25, rtol=1e-13)
    np.testing.assert_allclose(Pa[:2, 2], [w / 2, h / 2], rtol=1e-12)
    # odd sizes: the mid-points use integer w/2, h/2, so the centre of mass is off the principal point
    Ko = np.array([[f, 0, 320.5], [0, f, 240.5], [0, 0, 1]])
    Po = orc.estimate_new_camera_matrix(Ko, np.zeros(4), (641, 481), np.eye(3))
    pts = np.array([[[320, 0]], [[641, 240]], [[320, 481]], [[0, 240]]], np.float64)
    und = orc.undistort_points(pts, Ko, np.zeros(4)).reshape(4, 2)
    cn = und.mean(0)
    fo = max(641 * 0.5 / (cn[0] - und[:, 0].min()), 641 * 0.5 / (und[:, 0].max() - cn[0]),
             481 * 0.5 / (cn[1] - und[:, 1].min()), 481 * 0.5 / (und[:, 1].max() - cn[1]))
    np.testing.assert_allclose(Po, [[fo, 0, -cn[0] * fo + 320.5], [0, fo, -cn[1] * fo + 240.5], [0, 0, 1]], rtol=1e-12)


def test_blend_matches_numpy(orc):
    rng = np.random.default_rng(5)
    fr = [rng.integers(0, 65536, (5, 7, 3)).astype(np.uint16) for _ in range(3)]
    acc = np.full((5, 7, 3), 9.0); ref = np.full((5, 7, 3), 9.0)
    for i, f in enumerate(fr):
        orc.blend_accumulate(f, acc, 1 / 3, i == 0)
        if i == 0:
            ref = ref * 0.0
        ref += 1 / 3 * f.astype(np.float64)
    np.testing.assert_array_equal(acc, ref)
    np.testing.assert_array_equal(orc.blend_finalize(acc, np.uint16), ref.astype(np.uint16))


# ------------------------------------------------------------------ committed oracle vectors
def test_oracle_vectors_regression(orc, golden_calls):
    """tests/golden/oracle_vectors.npz (made by make_oracle_vectors.py) pins today's oracle output so that a change
    of compiler, libm or flags that alters a single map entry or pixel is caught."""
    path = os.path.join(GOLDEN, "oracle_vectors.npz")
    v = np.load(path)
    z, _ = golden_calls
    from gyroflow_amd.stabilizer import synthetic_frames
    for case in ("survey_capture", "c3_4k", "preview_720p"):
        K, D, R, P = z[case + ".K"], z[case + ".D"], z[case + ".R"], z[case + ".P"]
        size = tuple(int(s) for s in z[case + ".size"])
        m1, m2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, size, orc.CV_16SC2)
        ys, xs = v[case + ".ys"], v[case + ".xs"]
        np.testing.assert_array_equal(m1[np.ix_(ys, xs)], v[case + ".map1"])
        np.testing.assert_array_equal(m2[np.ix_(ys, xs)], v[case + ".map2"])
    frame = synthetic_frames(676, 507, n=1)[0]
    c = "survey_capture"
    s = 0.25
    K = z[c + ".K"].copy(); K[:2] *= s
    P = z[c + ".P"].copy(); P[:2] *= s
    m1, m2 = orc.fisheye_init_undistort_rectify_map(K, z[c + ".D"], z[c + ".R"], P, (480, 270), orc.CV_16SC2)
    for border in (0, 1):
        out = orc.remap_linear(frame, m1, m2, borderMode=border, borderValue=(10, 20, 30))
        np.testing.assert_array_equal(out[::9, ::11], v["remap_b%d" % border])


def test_undistort_points_pinned_by_gyroflow_sample(orc):
    """The one result of the reference's own that pins cv2 arithmetic on this path: the fcorr column and min_fcorr of
    plugins/blender/GX016015.MP4.gyroflow were produced by AdaptiveZoom.compute (smoothingFocus=2.0, out_size=(1920,1080):
    stabilizer.py:1574,1637-1640), i.e. by cv2.fisheye.undistortPoints on 32 boundary points per frame.  The file stores
    quaternions and fcorr rounded to 5 decimals, so the reproduction can only agree to ~1e-4."""
    from oracle import ref_adaptive_zoom as raz
    z = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))
    st = z["stab_transform"]
    fov, centers, _ = raz.compute(orc, z["K"], z["D"], tuple(int(v) for v in z["calib_dimension"]), st[:, 3:7], (1920, 1080),
                                  float(z["fps"]), smoothingFocus=2.0)
    assert fov.shape == (1622,)
    assert np.abs(fov - st[:, 2]).max() < 1.5e-4
    assert abs(fov.min() - float(z["min_fcorr"])) < 5e-5
    assert np.all(centers == [2000, 1500])
