"""ctypes/numpy front-end of the CPU oracle (oracle/cv_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see the header of cv_oracle.c): this is the
build's own restatement of the OpenCV 4.5.2 arithmetic behind the reference's
cv2.fisheye.initUndistortRectifyMap / cv2.convertMaps / cv2.remap calls
(/root/reference/calibrate_video.py:376-383, stabilizer.py:1442, nonlinear_stretch.py:130).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborc.so")

CV_32FC1 = 5
CV_16SC2 = 11
BORDER_CONSTANT = 0
BORDER_REPLICATE = 1
BORDER_TRANSPARENT = 5


def build(force=False):
    src = os.path.join(_HERE, "cv_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_cv_round_d.argtypes = [C.c_double]
        _lib.orc_cv_round_f.argtypes = [C.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _d9(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


def cv_round(v):
    return lib().orc_cv_round_d(float(v))


def cv_round_f(v):
    return lib().orc_cv_round_f(float(v))


def compute_iR(R, P):
    R = _d9(R); P = _d9(np.asarray(P, dtype=np.float64)[:, :3])
    out = np.empty(9, np.float64)
    rc = lib().orc_compute_iR(_p(R), _p(P), _p(out))
    if rc:
        raise ValueError("singular P*R")
    return out.reshape(3, 3)


def fisheye_init_undistort_rectify_map(K, D, R, P, size, m1type, iR=None, wcheck=True, nthreads=1):
    """cv2.fisheye.initUndistortRectifyMap(K, D, R, P, size, m1type) -> (map1, map2).
    nthreads > 1 stripes the rows over threads (same bits; OpenCV runs this loop on one thread)."""
    w, h = int(size[0]), int(size[1])
    K = _d9(K); D4 = np.zeros(4); Df = np.asarray(D, np.float64).reshape(-1); D4[:Df.size] = Df
    R = _d9(np.eye(3) if R is None else R); P = _d9(np.asarray(P, np.float64)[:, :3])
    if m1type == CV_16SC2:
        m1 = np.empty((h, w, 2), np.int16); m2 = np.empty((h, w), np.uint16)
    elif m1type == CV_32FC1:
        m1 = np.empty((h, w), np.float32); m2 = np.empty((h, w), np.float32)
    else:
        raise ValueError("m1type")
    iRp = _p(_d9(iR)) if iR is not None else None
    rc = lib().orc_fisheye_maps_mt(_p(K), _p(D4), _p(R), _p(P), iRp, w, h, m1type, 1 if wcheck else 0,
                                   _p(m1), C.c_size_t(m1.strides[0]), _p(m2), C.c_size_t(m2.strides[0]), int(nthreads))
    if rc:
        raise ValueError("orc_fisheye_maps rc=%d" % rc)
    return m1, m2


def convert_maps(mx, my):
    """cv2.convertMaps(mx_f32, my_f32, cv2.CV_16SC2) -> (map1 int16[H,W,2], map2 uint16[H,W])."""
    mx = np.ascontiguousarray(mx, np.float32); my = np.ascontiguousarray(my, np.float32)
    h, w = mx.shape
    m1 = np.empty((h, w, 2), np.int16); m2 = np.empty((h, w), np.uint16)
    lib().orc_convert_maps_32f_16sc2(_p(mx), C.c_size_t(mx.strides[0]), _p(my), C.c_size_t(my.strides[0]), w, h,
                                     _p(m1), C.c_size_t(m1.strides[0]), _p(m2), C.c_size_t(m2.strides[0]))
    return m1, m2


def _bv(borderValue):
    bv = np.zeros(4, np.float64)
    v = np.asarray(borderValue, np.float64).reshape(-1)
    if v.size == 1:
        bv[0] = v[0]      # cv2 Scalar(v) = (v,0,0,0)
    else:
        bv[:min(4, v.size)] = v[:4]
    return bv


def remap_linear(src, map1, map2, dst=None, borderMode=BORDER_CONSTANT, borderValue=0, nthreads=1):
    """cv2.remap(src, map1(16SC2), map2(16UC1), INTER_LINEAR, dst, borderMode, borderValue), 3 channels."""
    assert src.ndim == 3 and src.shape[2] == 3 and src.dtype in (np.uint8, np.uint16)
    src = np.ascontiguousarray(src)
    map1 = np.ascontiguousarray(map1, np.int16); map2 = np.ascontiguousarray(map2, np.uint16)
    dh, dw = map2.shape
    if dst is None or dst.shape != (dh, dw, 3) or dst.dtype != src.dtype or not dst.flags.c_contiguous:
        dst = np.zeros((dh, dw, 3), src.dtype)
    depth = 0 if src.dtype == np.uint8 else 2
    bv = _bv(borderValue)
    rc = lib().orc_remap_linear_c3(depth, _p(src), src.shape[1], src.shape[0], C.c_size_t(src.strides[0]),
                                   _p(dst), dw, dh, C.c_size_t(dst.strides[0]),
                                   _p(map1), C.c_size_t(map1.strides[0]), _p(map2), C.c_size_t(map2.strides[0]),
                                   int(borderMode), _p(bv), int(nthreads))
    if rc:
        raise ValueError("orc_remap_linear_c3 rc=%d" % rc)
    return dst


def remap_cubic(src, mx, my, dst=None, borderMode=BORDER_CONSTANT, borderValue=0, nthreads=1):
    """cv2.remap(src_u8c3, mx_f32, my_f32, INTER_CUBIC, ...)."""
    assert src.ndim == 3 and src.shape[2] == 3 and src.dtype == np.uint8
    src = np.ascontiguousarray(src)
    mx = np.ascontiguousarray(mx, np.float32); my = np.ascontiguousarray(my, np.float32)
    dh, dw = mx.shape
    if dst is None or dst.shape != (dh, dw, 3) or dst.dtype != src.dtype or not dst.flags.c_contiguous:
        dst = np.zeros((dh, dw, 3), src.dtype)
    bv = _bv(borderValue)
    rc = lib().orc_remap_cubic_u8c3_f32maps(_p(src), src.shape[1], src.shape[0], C.c_size_t(src.strides[0]),
                                            _p(dst), dw, dh, C.c_size_t(dst.strides[0]),
                                            _p(mx), C.c_size_t(mx.strides[0]), _p(my), C.c_size_t(my.strides[0]),
                                            int(borderMode), _p(bv), int(nthreads))
    if rc:
        raise ValueError("orc_remap_cubic rc=%d" % rc)
    return dst


def linear_tab():
    f = np.empty((1024, 4), np.float32); i = np.empty((1024, 4), np.int16)
    lib().orc_linear_tab(_p(f), _p(i))
    return f, i


def cubic_tab():
    f = np.empty((1024, 16), np.float32); i = np.empty((1024, 16), np.int16)
    lib().orc_cubic_tab(_p(f), _p(i))
    return f, i


def undistort_points(pts, K, D, R=None, P=None):
    """cv2.fisheye.undistortPoints(pts[N,1,2], K, D, R=R, P=P) -> same shape, f64."""
    pts = np.asarray(pts, np.float64)
    shp = pts.shape
    flat = np.ascontiguousarray(pts.reshape(-1, 2))
    out = np.empty_like(flat)
    K = _d9(K); D4 = np.zeros(4); Df = np.asarray(D, np.float64).reshape(-1); D4[:Df.size] = Df
    R = _d9(np.eye(3) if R is None else R); P = _d9(np.eye(3) if P is None else np.asarray(P, np.float64)[:, :3])
    lib().orc_undistort_points(_p(flat), flat.shape[0], _p(K), _p(D4), _p(R), _p(P), _p(out))
    return out.reshape(shp)


def estimate_new_camera_matrix(K, D, image_size, R=None, balance=0.0, new_size=None, fov_scale=1.0):
    """cv2.fisheye.estimateNewCameraMatrixForUndistortRectify (reference call: calibrate_video.py:314), restated from the OpenCV
    4.5 source as remembered -- PARITY UNPINNED (nothing in the reference tree holds an output of it).  Quirks restated: the four
    edge mid-points use integer w/2, h/2; cn[1] and the points' y are scaled by fx/fy; balance clamps to [0, 1]; fov_scale <= 0
    means 1.  Written independently of gyroflow_amd.cv's version (scalar loops, CPU undistort_points)."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    w, h = int(image_size[0]), int(image_size[1])
    balance = min(max(float(balance), 0.0), 1.0)
    pts = np.array([[[w // 2, 0]], [[w, h // 2]], [[w // 2, h]], [[0, h // 2]]], np.float64)
    und = undistort_points(pts, K, D, R=R).reshape(4, 2)
    sx = sy = 0.0
    for i in range(4):
        sx += und[i, 0]; sy += und[i, 1]
    cn0, cn1 = sx / 4.0, sy / 4.0
    ar = K[0, 0] / K[1, 1]
    cn1 *= ar
    minx = miny = float("inf"); maxx = maxy = -float("inf")
    for i in range(4):
        y = und[i, 1] * ar
        miny = min(miny, y); maxy = max(maxy, y); minx = min(minx, und[i, 0]); maxx = max(maxx, und[i, 0])
    f1 = w * 0.5 / (cn0 - minx); f2 = w * 0.5 / (maxx - cn0); f3 = h * 0.5 * ar / (cn1 - miny); f4 = h * 0.5 * ar / (maxy - cn1)
    fmin = min(f1, min(f2, min(f3, f4))); fmax = max(f1, max(f2, max(f3, f4)))
    f = balance * fmin + (1.0 - balance) * fmax
    f *= (1.0 / fov_scale) if fov_scale > 0 else 1.0
    nf0, nf1 = f, f
    nc0, nc1 = -cn0 * f + w * 0.5, -cn1 * f + (h * ar) * 0.5
    nf1 /= ar; nc1 /= ar
    if new_size is not None and int(new_size[0]) * int(new_size[1]) > 0:
        rx = new_size[0] / float(w); ry = new_size[1] / float(h)
        nf0 *= rx; nf1 *= ry; nc0 *= rx; nc1 *= ry
    return np.array([[nf0, 0.0, nc0], [0.0, nf1, nc1], [0.0, 0.0, 1.0]])


def blend_accumulate(frame, acc, weight, reset):
    frame = np.ascontiguousarray(frame)
    assert acc.dtype == np.float64 and acc.flags.c_contiguous and acc.size == frame.size
    lib().orc_blend_accumulate(0 if frame.dtype == np.uint8 else 2, _p(frame), _p(acc), C.c_double(weight),
                               int(bool(reset)), C.c_size_t(frame.size))
    return acc


def blend_finalize(acc, dtype):
    out = np.empty(acc.shape, dtype)
    lib().orc_blend_finalize(0 if np.dtype(dtype) == np.uint8 else 2, _p(acc), _p(out), C.c_size_t(acc.size))
    return out
