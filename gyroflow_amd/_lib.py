"""ctypes binding of libgfremap.so (include/gfremap.h).  No fallback: if the HIP library is missing or a
call fails, this raises -- the product path never routes through a CPU implementation."""
import ctypes as C
import os
import threading

import numpy as np
import torch  # noqa: F401  -- imported first so libgfremap.so binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GF_LIB_PATH") or os.path.join(_HERE, "libgfremap.so")     # GF_LIB_PATH: A/B builds of tools/ only

GF_OK = 0
GF_ERR_INVALID_ARG, GF_ERR_HIP, GF_ERR_UNSUPPORTED, GF_ERR_SINGULAR = -1, -2, -3, -4
DEPTH_U8, DEPTH_U16 = 0, 2
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_TRANSPARENT = 0, 1, 5


class GfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libgfremap error %d: %s" % (code, msg))
        self.code = code


class FrameXform(C.Structure):
    """gf_frame_xform: iR[9], f[2], c[2] (13 doubles)."""
    _fields_ = [("iR", C.c_double * 9), ("f", C.c_double * 2), ("c", C.c_double * 2)]


XFORM_DOUBLES = 13
assert C.sizeof(FrameXform) == XFORM_DOUBLES * 8

_vp, _sz, _i, _d, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_float
_dp = C.POINTER(C.c_double)

# name -> argtypes; every function returns int except the two noted in load()
SIGNATURES = {
    "gf_version": [],
    "gf_last_error": [],
    "gf_ctx_create": [_i, C.POINTER(_vp)],
    "gf_ctx_destroy": [_vp],
    "gf_ctx_set_kernel_mode": [_vp, _i],
    "gf_make_frame_xform": [_dp, _dp, _dp, C.POINTER(FrameXform)],
    "gf_make_frame_xforms": [_dp, _dp, _dp, _sz, _vp],
    "gf_fisheye_maps_16sc2": [_vp, _dp, C.POINTER(FrameXform), _i, _i, _vp, _sz, _vp, _sz, _vp],
    "gf_fisheye_maps_32f": [_vp, _dp, C.POINTER(FrameXform), _i, _i, _vp, _sz, _vp, _sz, _vp],
    "gf_convert_maps_32f_16sc2": [_vp, _vp, _sz, _vp, _sz, _f, _f, _f, _f, _i, _i, _vp, _sz, _vp, _sz, _vp],
    "gf_remap_linear_c3": [_vp, _i, _vp, _i, _i, _sz, _vp, _i, _i, _sz, _vp, _sz, _vp, _sz, _i, _dp, _vp],
    "gf_remap_cubic_u8c3_f32maps": [_vp, _vp, _i, _i, _sz, _vp, _i, _i, _sz, _vp, _sz, _vp, _sz, _i, _dp, _vp],
    "gf_superview_cubic_u8c3": [_vp, _vp, _i, _i, _sz, _sz, _vp, _i, _i, _sz, _sz, _vp, _vp, _i, _i, _dp, _vp],
    "gf_undistort_rotate_c3": [_vp, _i, _vp, _i, _i, _sz, _sz, _vp, _i, _i, _sz, _sz, _dp, _vp, _i, _i, _dp, _vp],
    "gf_rotate_c3": [_vp, _i, _vp, _i, _i, _sz, _sz, _vp, _sz, _sz, _i, _i, _vp],
    "gf_undistort_rotate_rot_c3": [_vp, _i, _vp, _i, _i, _sz, _sz, _vp, _i, _i, _sz, _sz, _dp, _vp, _i, _i, _i, _dp, _vp],
    "gf_undistort_rotate_stretched_c3": [_vp, _i, _vp, _i, _i, _sz, _sz, _vp, _i, _i, _sz, _sz, _dp, _vp, _i, _f, _f, _f, _f, _i, _dp, _vp],
    "gf_undistort_rotate_superview_u8c3": [_vp, _vp, _i, _i, _sz, _sz, _i, _i, _vp, _i, _i, _sz, _sz, _dp, _vp, _i, _vp, _vp, _i, _dp, _vp],
    "gf_undistort_rotate_c3_noalloc": [_i, _i, _vp, _i, _i, _sz, _sz, _vp, _i, _i, _sz, _sz, _dp, _vp, _i, _i, _dp, _vp],
    "gf_undistort_rotate_blend_c3": [_vp, _i, _vp, _i, _i, _sz, _sz, _vp, _i, _i, _sz, _sz, _dp, _vp, _i, _i, _i, _i, _dp, _vp],
    "gf_blend_accumulate": [_vp, _i, _vp, _vp, _d, _i, _sz, _vp],
    "gf_blend_finalize": [_vp, _i, _vp, _vp, _sz, _vp],
    "gf_fisheye_undistort_points": [_vp, _vp, _sz, _dp, _dp, _vp, _sz, _vp, _vp],
    "gf_host_integrate_gyro": [_dp, _sz, _dp, _dp],
    "gf_host_interp_orientations": [_dp, _dp, _sz, _d, _d, _dp, _dp, _sz],
    "gf_host_plain_slerp": [_dp, _sz, _d, _dp],
    "gf_host_rot_between": [_dp, _dp, _sz, _dp],
    "gf_get_cubic_table": [_vp],
    "gf_get_rp_table": [_dp],
    "gf_lens_scale_host": [_dp, _dp, _sz, _dp],
}

_lib = None
_lock = threading.Lock()
_ctxs = {}


def load():
    """Load libgfremap.so and declare every prototype of include/gfremap.h."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError("gyroflow_amd: %s not found -- build it with `make -C gyroflow_amd/csrc` "
                                      "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
                lib = C.CDLL(LIB_PATH)
                for name, args in SIGNATURES.items():
                    fn = getattr(lib, name)          # AttributeError if the symbol is not exported
                    fn.argtypes = args
                    fn.restype = C.c_char_p if name == "gf_last_error" else \
                        C.c_longlong if name == "gf_host_interp_orientations" else C.c_int
                _lib = lib
    return _lib


def check(rc):
    if rc != GF_OK:
        raise GfError(rc, (load().gf_last_error() or b"").decode("utf-8", "replace"))


def ctx(device_index):
    """Per-device opaque context (cached for the life of the process)."""
    lib = load()
    with _lock:
        c = _ctxs.get(device_index)
        if c is None:
            h = _vp()
            check(lib.gf_ctx_create(int(device_index), C.byref(h)))
            c = _ctxs[device_index] = h
    return c


KERNEL_DIRECT, KERNEL_RP, KERNEL_RL = 0, 1, 2
KERNEL_DEFAULT = KERNEL_RP


def set_kernel_mode(device_index, mode):
    """A/B switch of the fused kernel / CV_16SC2 map kernel of a device's context (gf_ctx_set_kernel_mode)."""
    check(load().gf_ctx_set_kernel_mode(ctx(device_index), int(mode)))


def border_value(v):
    arr = (C.c_double * 4)(0.0, 0.0, 0.0, 0.0)
    a = np.asarray(v, np.float64).reshape(-1)
    for k in range(min(4, a.size)):      # cv2 Scalar(v) = (v, 0, 0, 0)
        arr[k] = float(a[k])
    return arr


def distortion(D):
    """k1..k4 as a ctypes double[4] (missing coefficients are 0, as cv2 treats an empty D)."""
    arr = (C.c_double * 4)(0.0, 0.0, 0.0, 0.0)
    if D is not None:
        Df = np.asarray(D, np.float64).reshape(-1)
        for k in range(min(4, Df.size)):
            arr[k] = float(Df[k])
    return arr


def make_frame_xform(K, R, P):
    """Host: (K, R, P) as handed to cv2.fisheye.initUndistortRectifyMap -> FrameXform."""
    lib = load()
    Kc = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
    Rc = np.ascontiguousarray(np.asarray(np.eye(3) if R is None else R, np.float64).reshape(9))
    Pc = np.ascontiguousarray(np.asarray(P, np.float64)[:, :3].reshape(9))
    out = FrameXform()
    check(lib.gf_make_frame_xform(Kc.ctypes.data_as(_dp), Rc.ctypes.data_as(_dp), Pc.ctypes.data_as(_dp), C.byref(out)))
    return out


def make_frame_xforms(K, Rs, Ps):
    """Host: one lens matrix K, n rotations [n,3,3] and n new camera matrices [n,3,3] -> [n,13] float64 (rows = FrameXform)."""
    lib = load()
    Kc = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
    Rc = np.ascontiguousarray(np.asarray(Rs, np.float64).reshape(-1, 9))
    Pc = np.ascontiguousarray(np.asarray(Ps, np.float64).reshape(-1, 9))
    if Rc.shape != Pc.shape:
        raise ValueError("Rs and Ps must hold the same number of 3x3 matrices")
    out = np.empty((Rc.shape[0], XFORM_DOUBLES), np.float64)
    check(lib.gf_make_frame_xforms(Kc.ctypes.data_as(_dp), Rc.ctypes.data_as(_dp), Pc.ctypes.data_as(_dp), Rc.shape[0],
                                   out.ctypes.data_as(_vp)))
    return out


def xform_to_array(xf):
    return np.frombuffer(bytes(xf), dtype=np.float64).copy()


def rp_table():
    """Host: the RP engine's constants (M [8,8], nodes [8], row weights [64,8])."""
    out = np.empty(8 * 8 + 8 + 64 * 8, np.float64)
    check(load().gf_get_rp_table(out.ctypes.data_as(_dp)))
    return out[:64].reshape(8, 8), out[64:72], out[72:].reshape(64, 8)


def lens_scale_host(D, t):
    """Host: the kernels' table approximation of S(t) = theta_d(atan sqrt t)/sqrt t for an array of t."""
    t = np.ascontiguousarray(np.asarray(t, np.float64).reshape(-1))
    out = np.empty_like(t)
    check(load().gf_lens_scale_host(distortion(D), t.ctypes.data_as(_dp), t.size, out.ctypes.data_as(_dp)))
    return out
