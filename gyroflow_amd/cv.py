"""The slice of the cv2 call surface that gyroflow's hot path uses, running on MI355X through libgfremap.so.

Mirrors (names, argument meaning, return/ownership rules):
  cv2.fisheye.initUndistortRectifyMap   /root/reference/calibrate_video.py:322,376,383
  cv2.convertMaps                       /root/reference/calibrate_video.py:378
  cv2.remap                             /root/reference/stabilizer.py:1442, nonlinear_stretch.py:130, gyroflow.py:275,283
  cv2.fisheye.undistortPoints           /root/reference/adaptive_zoom.py:220
  cv2.rotate                            /root/reference/stabilizer.py:1353,1386

Arrays may be torch CUDA tensors (stay on the device, work is enqueued on torch's current stream) or numpy
arrays (uploaded, processed on cuda:<current>, downloaded: numpy in -> numpy out, like cv2).  There is no CPU
implementation behind these functions.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

CV_32FC1 = 5
CV_16SC2 = 11
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_TRANSPARENT = 0, 1, 5
ROTATE_90_CLOCKWISE, ROTATE_180, ROTATE_90_COUNTERCLOCKWISE = 0, 1, 2

_vp = C.c_void_p


def _device(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("gyroflow_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise ValueError("device must be a cuda device")
    return torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)


def _stream(dev):
    return _vp(torch.cuda.current_stream(dev).cuda_stream)


def _to_dev(a, dev):
    """-> (cuda tensor, came_from_numpy)"""
    if isinstance(a, torch.Tensor):
        if a.device.type != "cuda":
            return a.to(dev), False
        return a, False
    a = np.asarray(a)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev), True


def _ptr(t):
    return _vp(t.data_ptr())


def _depth(t):
    if t.dtype == torch.uint8:
        return _lib.DEPTH_U8
    if t.dtype == torch.uint16:
        return _lib.DEPTH_U16
    raise TypeError("only uint8 / uint16 images are supported, got %s" % t.dtype)


def _check_img(t, what):
    if t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("%s must be HxWx3, got %s" % (what, tuple(t.shape)))
    if t.stride(2) != 1 or t.stride(1) != 3:
        raise ValueError("%s must have packed pixels (row stride may be padded)" % what)


def fisheye_initUndistortRectifyMap(K, D, R, P, size, m1type, device=None):
    """cv2.fisheye.initUndistortRectifyMap(K, D, R, P, size, m1type) -> (map1, map2) as CUDA tensors.

    m1type CV_16SC2 -> int16[H,W,2] + uint16[H,W];  CV_32FC1 -> float32[H,W] x 2."""
    dev = _device(device)
    w, h = int(size[0]), int(size[1])
    xf = _lib.make_frame_xform(K, R, P)
    Dk = _lib.distortion(D)
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        if m1type == CV_16SC2:
            m1 = torch.empty((h, w, 2), dtype=torch.int16, device=dev)
            m2 = torch.empty((h, w), dtype=torch.uint16, device=dev)
            _lib.check(lib.gf_fisheye_maps_16sc2(ctx, Dk, C.byref(xf), w, h, _ptr(m1), w * 4, _ptr(m2), w * 2, _stream(dev)))
        elif m1type == CV_32FC1:
            m1 = torch.empty((h, w), dtype=torch.float32, device=dev)
            m2 = torch.empty((h, w), dtype=torch.float32, device=dev)
            _lib.check(lib.gf_fisheye_maps_32f(ctx, Dk, C.byref(xf), w, h, _ptr(m1), w * 4, _ptr(m2), w * 4, _stream(dev)))
        else:
            raise ValueError("m1type must be CV_16SC2 or CV_32FC1")
    return m1, m2


def convertMaps(map1, map2, dstmap1type=CV_16SC2, scale_x=None, scale_y=None):
    """cv2.convertMaps(map1_f32, map2_f32, cv2.CV_16SC2).

    scale_x=(num, den) / scale_y=(num, den) fold in the reference's float32 pre-scaling
    `map1 * orig_w / calib_w` (calibrate_video.py:378): multiply, round, divide, round -- in float32."""
    if dstmap1type != CV_16SC2:
        raise ValueError("only CV_16SC2 output is supported")
    dev = _device(map1.device if isinstance(map1, torch.Tensor) and map1.is_cuda else None)
    m1, np1 = _to_dev(map1, dev)
    m2, _ = _to_dev(map2, dev)
    if m1.dtype != torch.float32 or m2.dtype != torch.float32 or m1.shape != m2.shape or m1.dim() != 2:
        raise ValueError("maps must be two float32 HxW arrays")
    m1 = m1.contiguous(); m2 = m2.contiguous()
    h, w = m1.shape
    sx = (1.0, 1.0) if scale_x is None else scale_x
    sy = (1.0, 1.0) if scale_y is None else scale_y
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        o1 = torch.empty((h, w, 2), dtype=torch.int16, device=dev)
        o2 = torch.empty((h, w), dtype=torch.uint16, device=dev)
        _lib.check(lib.gf_convert_maps_32f_16sc2(ctx, _ptr(m1), w * 4, _ptr(m2), w * 4, float(sx[0]), float(sx[1]),
                                                 float(sy[0]), float(sy[1]), w, h, _ptr(o1), w * 4, _ptr(o2), w * 2,
                                                 _stream(dev)))
    if np1:
        return o1.cpu().numpy(), o2.cpu().numpy()
    return o1, o2


def remap(src, map1, map2, interpolation, dst=None, borderMode=BORDER_CONSTANT, borderValue=0):
    """cv2.remap for the two map/interpolation combinations the reference uses:
       INTER_LINEAR with (int16[H,W,2], uint16[H,W]) maps on uint8/uint16 HxWx3, and
       INTER_CUBIC  with (float32[H,W], float32[H,W]) maps on uint8 HxWx3.
    Returns dst: the given dst is written in place when its shape/dtype match (as cv2 does), otherwise a new
    zero-filled array is created."""
    src_is_t = isinstance(src, torch.Tensor)
    dev = _device(src.device if src_is_t and src.is_cuda else None)
    s, s_np = _to_dev(src, dev)
    _check_img(s, "src")
    m1, _ = _to_dev(map1, dev)
    m2, _ = _to_dev(map2, dev)
    dh, dw = int(m2.shape[0]), int(m2.shape[1])
    depth = _depth(s)

    d, d_np, dst_ok = None, False, False
    if dst is not None:
        shape_ok = tuple(dst.shape) == (dh, dw, 3)
        if isinstance(dst, torch.Tensor):
            dst_ok = shape_ok and dst.dtype == s.dtype and dst.is_cuda and dst.device == s.device and dst.stride(2) == 1 and dst.stride(1) == 3
            if dst_ok:
                d = dst
        else:
            dst_ok = shape_ok and dst.dtype == (np.uint8 if depth == 0 else np.uint16) and dst.flags.c_contiguous
            if dst_ok:
                d_np = True
                d = torch.from_numpy(dst).to(dev) if borderMode == BORDER_TRANSPARENT else \
                    torch.empty((dh, dw, 3), dtype=s.dtype, device=dev)
    if d is None:
        d = torch.zeros((dh, dw, 3), dtype=s.dtype, device=dev)
    esz = s.element_size()
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    bv = _lib.border_value(borderValue)
    with torch.cuda.device(dev):
        if interpolation == INTER_LINEAR:
            if m1.dtype != torch.int16 or m1.dim() != 3 or m1.shape[2] != 2 or m2.dtype != torch.uint16 or tuple(m1.shape[:2]) != (dh, dw):
                raise ValueError("INTER_LINEAR needs map1 int16[H,W,2] and map2 uint16[H,W]")
            m1 = m1.contiguous(); m2 = m2.contiguous()
            _lib.check(lib.gf_remap_linear_c3(ctx, depth, _ptr(s), s.shape[1], s.shape[0], s.stride(0) * esz,
                                              _ptr(d), dw, dh, d.stride(0) * esz, _ptr(m1), dw * 4, _ptr(m2), dw * 2,
                                              int(borderMode), bv, _stream(dev)))
        elif interpolation == INTER_CUBIC:
            if m1.dtype != torch.float32 or m2.dtype != torch.float32 or m1.shape != m2.shape or m1.dim() != 2:
                raise ValueError("INTER_CUBIC needs two float32 HxW maps")
            if depth != _lib.DEPTH_U8:
                raise TypeError("INTER_CUBIC is implemented for uint8 only")
            m1 = m1.contiguous(); m2 = m2.contiguous()
            _lib.check(lib.gf_remap_cubic_u8c3_f32maps(ctx, _ptr(s), s.shape[1], s.shape[0], s.stride(0), _ptr(d), dw, dh,
                                                       d.stride(0), _ptr(m1), dw * 4, _ptr(m2), dw * 4, int(borderMode), bv,
                                                       _stream(dev)))
        else:
            raise ValueError("interpolation must be INTER_LINEAR or INTER_CUBIC")
    if s_np or d_np:
        out = d.cpu().numpy()
        if d_np:
            dst[...] = out
            return dst
        return out
    return d


def xforms_to_device(xforms, device=None):
    """[N,13] float64 numpy (rows = gf_frame_xform) -> CUDA tensor for undistort_rotate()."""
    dev = _device(device)
    a = np.ascontiguousarray(np.asarray(xforms, np.float64).reshape(-1, _lib.XFORM_DOUBLES))
    return torch.from_numpy(a).to(dev)


def _frames4(t, what):
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.shape[3] != 3 or t.stride(3) != 1 or t.stride(2) != 3:
        raise ValueError("%s must be [N,]HxWx3 with packed pixels" % what)
    return t


def rotate(src, rotateCode, dst=None):
    """cv2.rotate(src, rotateCode) (stabilizer.py:1353,1386) for one frame [H,W,3] or a batch [N,H,W,3]; CUDA tensors stay on
    the device, numpy in -> numpy out."""
    if rotateCode not in (ROTATE_90_CLOCKWISE, ROTATE_180, ROTATE_90_COUNTERCLOCKWISE):
        raise ValueError("rotateCode must be ROTATE_90_CLOCKWISE, ROTATE_180 or ROTATE_90_COUNTERCLOCKWISE")
    dev = _device(src.device if isinstance(src, torch.Tensor) and src.is_cuda else None)
    t, was_np = _to_dev(src, dev)
    s = _frames4(t, "src")
    n, sh, sw = s.shape[0], s.shape[1], s.shape[2]
    dw, dh = (sw, sh) if rotateCode == ROTATE_180 else (sh, sw)
    if dst is None or was_np:
        d = torch.empty((n, dh, dw, 3), dtype=s.dtype, device=dev)
    else:
        d = _frames4(dst, "dst")
        if tuple(d.shape) != (n, dh, dw, 3) or d.dtype != s.dtype or d.device != dev:
            raise ValueError("dst has the wrong shape/dtype/device")
    esz = s.element_size()
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        _lib.check(lib.gf_rotate_c3(ctx, _depth(s), _ptr(s), sw, sh, s.stride(1) * esz, s.stride(0) * esz, _ptr(d), d.stride(1) * esz,
                                    d.stride(0) * esz, n, int(rotateCode), _stream(dev)))
    out = d if t.dim() == 4 else d[0]
    return out.cpu().numpy() if was_np else out


_cv_rotate = rotate          # undistort_rotate() has a parameter of that name


def undistort_rotate(src, D, xforms_dev, out_size, dst=None, borderMode=BORDER_CONSTANT, borderValue=0, map_scale=None, rotate=-1):
    """Fused get_maps + remap(INTER_LINEAR) for a batch of frames (stabilizer.py:1423-1443 per frame).

    src: CUDA uint8/uint16 [N,H,W,3] (or [H,W,3]; a stride-0 expand() over N reuses one input frame).
    D: the lens's distortion coefficients k1..k4 (host).  xforms_dev: CUDA float64 [N,13] from xforms_to_device().
    map_scale = (sx_num, sx_den, sy_num, sy_den): the float32 map rescale of a stretched preset (calibrate_video.py:375-380;
    FisheyeCalibrator.map_scale()), None for ordinary presets.
    rotate: a cv2.rotate code the reference would apply to every frame first (stabilizer.py:1385-1386), -1 for none; the
    rotation is folded into the kernel's tap addressing (RP kernel), or done by rotate() where that form does not exist.
    Returns dst [N,out_h,out_w,3]."""
    if not (isinstance(src, torch.Tensor) and src.is_cuda):
        raise TypeError("undistort_rotate works on CUDA tensors")
    dev = src.device
    s = _frames4(src, "src")
    n = s.shape[0]
    if rotate is not None and rotate != -1:
        if s.stride(0) == 0 and n > 1:
            s = s[:1].contiguous().expand(n, -1, -1, -1)
        rc = _lib.GF_ERR_UNSUPPORTED
        if map_scale is None:
            dw, dh = int(out_size[0]), int(out_size[1])
            d = dst
            if d is None:
                d = torch.zeros((n, dh, dw, 3), dtype=s.dtype, device=dev) if borderMode == BORDER_TRANSPARENT else \
                    torch.empty((n, dh, dw, 3), dtype=s.dtype, device=dev)
            d4 = _frames4(d, "dst")
            if tuple(d4.shape) != (n, dh, dw, 3) or d4.dtype != s.dtype or d4.device != dev:
                raise ValueError("dst has the wrong shape/dtype/device")
            if xforms_dev.dtype != torch.float64 or not xforms_dev.is_cuda or xforms_dev.numel() != n * _lib.XFORM_DOUBLES or not xforms_dev.is_contiguous():
                raise ValueError("xforms_dev must be a contiguous CUDA float64 [N,13] tensor with N = number of frames")
            esz = s.element_size()
            lib, ctx = _lib.load(), _lib.ctx(dev.index)
            with torch.cuda.device(dev):
                rc = lib.gf_undistort_rotate_rot_c3(ctx, _depth(s), _ptr(s), s.shape[2], s.shape[1], s.stride(1) * esz, s.stride(0) * esz,
                                                    _ptr(d4), dw, dh, d4.stride(1) * esz, d4.stride(0) * esz, _lib.distortion(D),
                                                    _ptr(xforms_dev), n, int(rotate), int(borderMode), _lib.border_value(borderValue), _stream(dev))
            if rc == _lib.GF_OK:
                return d4 if (dst is None or dst.dim() == 4) else d4[0]
            if rc != _lib.GF_ERR_UNSUPPORTED:
                _lib.check(rc)
        # no rotated form of this kernel (DIRECT / RL mode, stretched preset): rotate the batch, then the plain call
        return undistort_rotate(_cv_rotate(s, rotate), D, xforms_dev, out_size, dst=dst,
                                borderMode=borderMode, borderValue=borderValue, map_scale=map_scale)
    if xforms_dev.dtype != torch.float64 or not xforms_dev.is_cuda or xforms_dev.numel() != n * _lib.XFORM_DOUBLES or not xforms_dev.is_contiguous():
        raise ValueError("xforms_dev must be a contiguous CUDA float64 [N,13] tensor with N = number of frames")
    dw, dh = int(out_size[0]), int(out_size[1])
    if dst is None:
        d = torch.zeros((n, dh, dw, 3), dtype=s.dtype, device=dev) if borderMode == BORDER_TRANSPARENT else \
            torch.empty((n, dh, dw, 3), dtype=s.dtype, device=dev)
    else:
        d = _frames4(dst, "dst")
        if tuple(d.shape) != (n, dh, dw, 3) or d.dtype != s.dtype or d.device != dev:
            raise ValueError("dst has the wrong shape/dtype/device")
    esz = s.element_size()
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        if map_scale is None:
            _lib.check(lib.gf_undistort_rotate_c3(ctx, _depth(s), _ptr(s), s.shape[2], s.shape[1], s.stride(1) * esz, s.stride(0) * esz,
                                                  _ptr(d), dw, dh, d.stride(1) * esz, d.stride(0) * esz, _lib.distortion(D),
                                                  _ptr(xforms_dev), n, int(borderMode), _lib.border_value(borderValue), _stream(dev)))
        else:
            _lib.check(lib.gf_undistort_rotate_stretched_c3(ctx, _depth(s), _ptr(s), s.shape[2], s.shape[1], s.stride(1) * esz, s.stride(0) * esz,
                                                            _ptr(d), dw, dh, d.stride(1) * esz, d.stride(0) * esz, _lib.distortion(D),
                                                            _ptr(xforms_dev), n, float(map_scale[0]), float(map_scale[1]), float(map_scale[2]),
                                                            float(map_scale[3]), int(borderMode), _lib.border_value(borderValue), _stream(dev)))
    return d if (dst is None or dst.dim() == 4) else d[0]


def undistort_rotate_blend(src, D, xforms_dev, out_size, frames_per_group, n_blend, dst=None,
                           borderMode=BORDER_CONSTANT, borderValue=0):
    """Fused remap + hyperlapse blend (stabilizer.py:1431-1447,1464-1468): output g is the truncated float64 mean of
    the remaps of input frames g*frames_per_group .. +n_blend-1.  src [N_in,H,W,3], returns [n_out,out_h,out_w,3]."""
    if not (isinstance(src, torch.Tensor) and src.is_cuda):
        raise TypeError("undistort_rotate_blend works on CUDA tensors")
    dev = src.device
    s = _frames4(src, "src")
    n_in = s.shape[0]
    if n_blend < 1 or frames_per_group < n_blend or n_in < n_blend:
        raise ValueError("need 1 <= n_blend <= frames_per_group and at least n_blend input frames")
    n_out = (n_in - n_blend) // frames_per_group + 1
    if xforms_dev.dtype != torch.float64 or not xforms_dev.is_cuda or xforms_dev.numel() < ((n_out - 1) * frames_per_group + n_blend) * _lib.XFORM_DOUBLES:
        raise ValueError("xforms_dev must hold one transform per input frame")
    dw, dh = int(out_size[0]), int(out_size[1])
    d = torch.empty((n_out, dh, dw, 3), dtype=s.dtype, device=dev) if dst is None else _frames4(dst, "dst")
    if tuple(d.shape) != (n_out, dh, dw, 3) or d.dtype != s.dtype:
        raise ValueError("dst has the wrong shape/dtype")
    esz = s.element_size()
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        _lib.check(lib.gf_undistort_rotate_blend_c3(ctx, _depth(s), _ptr(s), s.shape[2], s.shape[1], s.stride(1) * esz, s.stride(0) * esz,
                                                    _ptr(d), dw, dh, d.stride(1) * esz, d.stride(0) * esz, _lib.distortion(D),
                                                    _ptr(xforms_dev), n_out, int(frames_per_group), int(n_blend), int(borderMode),
                                                    _lib.border_value(borderValue), _stream(dev)))
    return d


def superview_cubic(src, mapx_row, mapy_col, dst=None, borderMode=BORDER_CONSTANT, borderValue=0):
    """cv2.remap(img, map1, map2, cv2.INTER_CUBIC) for separable maps (map1[y, x] = mapx_row[x], map2[y, x] = mapy_col[y]),
    as NonlinearStretch builds them (nonlinear_stretch.py:58-100,130), batched: src CUDA uint8 [N,H,W,3] or [H,W,3];
    mapx_row / mapy_col CUDA float32 vectors of the output width / height."""
    if not (isinstance(src, torch.Tensor) and src.is_cuda):
        raise TypeError("superview_cubic works on CUDA tensors")
    dev = src.device
    s = _frames4(src, "src")
    if s.dtype != torch.uint8:
        raise TypeError("INTER_CUBIC is implemented for uint8 only")
    n = s.shape[0]
    if mapx_row.dtype != torch.float32 or mapy_col.dtype != torch.float32 or mapx_row.dim() != 1 or mapy_col.dim() != 1:
        raise ValueError("mapx_row / mapy_col must be float32 vectors")
    mx = mapx_row.to(dev).contiguous(); my = mapy_col.to(dev).contiguous()
    dw, dh = mx.numel(), my.numel()
    if dst is None:
        d = torch.zeros((n, dh, dw, 3), dtype=torch.uint8, device=dev)
    else:
        d = _frames4(dst, "dst")
        if tuple(d.shape) != (n, dh, dw, 3) or d.dtype != torch.uint8 or d.device != dev:
            raise ValueError("dst has the wrong shape/dtype/device")
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        _lib.check(lib.gf_superview_cubic_u8c3(ctx, _ptr(s), s.shape[2], s.shape[1], s.stride(1), s.stride(0), _ptr(d), dw, dh,
                                               d.stride(1), d.stride(0), _ptr(mx), _ptr(my), n, int(borderMode),
                                               _lib.border_value(borderValue), _stream(dev)))
    return d if (src.dim() == 4) else d[0]


def undistort_rotate_superview(src, D, xforms_dev, mid_size, mapx_row, mapy_col, dst=None, borderMode=BORDER_CONSTANT, borderValue=0,
                               fused=False, mid=None):
    """undistort_rotate() to an intermediate frame of mid_size followed by superview_cubic() (NonlinearStretch.apply_stretch on
    the remapped frame: stabilizer.py:1442 then nonlinear_stretch.py:130).
    src: CUDA uint8 [N,H,W,3]; mapx_row / mapy_col: float32 vectors of the output width / height (numpy, or tensors that
    are copied to the host once).  Returns dst [N, len(mapy_col), len(mapx_row), 3].

    Default: the two launches, the faster form on MI355X (35 vs 46 us per C4 frame: both stages are bound by instruction issue,
    so keeping the intermediate frame out of HBM buys nothing); `mid` may supply the [N, mid_h, mid_w, 3] intermediate buffer.
    fused=True: one launch whose intermediate frame lives in LDS -- 30 % less HBM traffic and no intermediate allocation, the
    same bytes out; maps that kernel cannot take (vertical scale != 1, non-monotone) still run as the two launches."""
    if not (isinstance(src, torch.Tensor) and src.is_cuda):
        raise TypeError("undistort_rotate_superview works on CUDA tensors")
    dev = src.device
    s = _frames4(src, "src")
    if s.dtype != torch.uint8:
        raise TypeError("the superview stretch (INTER_CUBIC) is implemented for uint8 only")
    n = s.shape[0]
    if xforms_dev.dtype != torch.float64 or not xforms_dev.is_cuda or xforms_dev.numel() != n * _lib.XFORM_DOUBLES or not xforms_dev.is_contiguous():
        raise ValueError("xforms_dev must be a contiguous CUDA float64 [N,13] tensor with N = number of frames")
    host = lambda m: np.ascontiguousarray(m.detach().cpu().numpy() if isinstance(m, torch.Tensor) else m, dtype=np.float32).reshape(-1)
    ondev = lambda m: m.reshape(-1) if (isinstance(m, torch.Tensor) and m.is_cuda) else torch.from_numpy(host(m)).to(dev)
    dw, dh = int(np.prod(mapx_row.shape)), int(np.prod(mapy_col.shape))
    mw, mh = int(mid_size[0]), int(mid_size[1])
    if dst is None:
        d = torch.empty((n, dh, dw, 3), dtype=torch.uint8, device=dev)
    else:
        d = _frames4(dst, "dst")
        if tuple(d.shape) != (n, dh, dw, 3) or d.dtype != torch.uint8 or d.device != dev:
            raise ValueError("dst has the wrong shape/dtype/device")
    def two_launches():
        m = mid
        if m is None or tuple(m.shape) != (n, mh, mw, 3) or m.dtype != torch.uint8 or m.device != dev:
            # BORDER_TRANSPARENT leaves uncovered pixels as they are: without a caller's buffer they are zeros, as in cv.remap
            m = (torch.zeros if borderMode == BORDER_TRANSPARENT else torch.empty)((n, mh, mw, 3), dtype=torch.uint8, device=dev)
        undistort_rotate(s, D, xforms_dev, (mw, mh), dst=m, borderMode=borderMode, borderValue=borderValue)
        superview_cubic(m, ondev(mapx_row), ondev(mapy_col), dst=d)        # maps already on the device are used where they lie

    if not fused:
        two_launches()
        return d if (src.dim() == 4) else d[0]
    mx, my = host(mapx_row), host(mapy_col)          # the fused entry takes the maps from the host (cached per context)
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        rc = lib.gf_undistort_rotate_superview_u8c3(ctx, _ptr(s), s.shape[2], s.shape[1], s.stride(1), s.stride(0), mw, mh,
                                                    _ptr(d), dw, dh, d.stride(1), d.stride(0), _lib.distortion(D), _ptr(xforms_dev), n,
                                                    mx.ctypes.data_as(_vp), my.ctypes.data_as(_vp), int(borderMode),
                                                    _lib.border_value(borderValue), _stream(dev))
        if rc == _lib.GF_ERR_UNSUPPORTED and borderMode != BORDER_TRANSPARENT:
            # maps outside the fused kernel's envelope: the same result from the two launches
            two_launches()
        else:
            _lib.check(rc)
    return d if (src.dim() == 4) else d[0]


def blend_accumulate(frame, acc, weight, reset):
    """acc (CUDA float64, same number of elements as frame) = (acc*0 if reset else acc) + weight*frame."""
    dev = frame.device
    f = frame.contiguous()
    if acc.dtype != torch.float64 or acc.numel() != f.numel() or not acc.is_contiguous():
        raise ValueError("acc must be a contiguous float64 tensor of frame's size")
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        _lib.check(lib.gf_blend_accumulate(ctx, _depth(f), _ptr(f), _ptr(acc), float(weight), int(bool(reset)), f.numel(), _stream(dev)))
    return acc


def blend_finalize(acc, dtype, out=None):
    """numpy-style truncating astype(uint8/uint16) of the float64 accumulator (into `out` when given)."""
    dev = acc.device
    if out is None:
        out = torch.empty(acc.shape, dtype=dtype, device=dev)
    elif out.dtype != dtype or out.numel() != acc.numel() or not out.is_contiguous() or out.device != dev:
        raise ValueError("out must be a contiguous tensor of acc's size and the requested dtype")
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        _lib.check(lib.gf_blend_finalize(ctx, _depth(out), _ptr(acc), _ptr(out), acc.numel(), _stream(dev)))
    return out


def fisheye_undistortPoints(distorted, K, D, R=None, P=None, device=None):
    """cv2.fisheye.undistortPoints(distorted, K, D, R=R, P=P) (adaptive_zoom.py:220).

    distorted: [1,N,2] or [N,2] float64 points.  R: None, one 3x3, or a batch [F,3,3] -- the batch form undistorts the
    same N points under F rotations in one launch (AdaptiveZoom.compute does this for every frame).  Returns a numpy
    array shaped like cv2's ([1,N,2] / [N,2]) for a single R, or [F,N,2] for a batch."""
    dev = _device(device)
    pts = np.asarray(distorted, np.float64)
    lead = pts.shape[:-2]
    pts2 = np.ascontiguousarray(pts.reshape(-1, 2))
    Kc = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
    Rm = np.eye(3)[None] if R is None else np.asarray(R, np.float64).reshape(-1, 3, 3)
    batch = R is not None and np.asarray(R).ndim == 3
    Pm = np.eye(3) if P is None else np.asarray(P, np.float64)[:3, :3]
    PR = np.empty((Rm.shape[0], 3, 3), np.float64)
    for i in range(3):                       # Matx33d product, k ascending like cv::Matx (RR = PP * RR)
        for j in range(3):
            PR[:, i, j] = (Pm[i, 0] * Rm[:, 0, j] + Pm[i, 1] * Rm[:, 1, j]) + Pm[i, 2] * Rm[:, 2, j]
    n_pts, n_sets = pts2.shape[0], PR.shape[0]
    if n_pts == 0:
        raise ValueError("no points")
    lib, ctx = _lib.load(), _lib.ctx(dev.index)
    with torch.cuda.device(dev):
        p_d = torch.from_numpy(pts2).to(dev)
        pr_d = torch.from_numpy(PR.reshape(-1, 9)).to(dev)
        out = torch.empty((n_sets, n_pts, 2), dtype=torch.float64, device=dev)
        _lib.check(lib.gf_fisheye_undistort_points(ctx, _ptr(p_d), n_pts, Kc.ctypes.data_as(_lib._dp), _lib.distortion(D),
                                                   _ptr(pr_d), n_sets, _ptr(out), _stream(dev)))
        res = out.cpu().numpy()
    return res if batch else res[0].reshape(lead + (n_pts, 2))


def fisheye_estimateNewCameraMatrixForUndistortRectify(K, D, image_size, R, P=None, balance=0.0, new_size=None, fov_scale=1.0,
                                                        device=None):
    """cv2.fisheye.estimateNewCameraMatrixForUndistortRectify(K, D, image_size, R, fov_scale=...) (calibrate_video.py:314).

    OpenCV 4.5 algorithm: the mid-points of the four image edges, (w/2, 0) (w, h/2) (w/2, h) (0, h/2) with *integer* w/2 and
    h/2, go through fisheye::undistortPoints (here: the GPU kernel behind fisheye_undistortPoints); the focal length is the
    one that maps the farthest (balance = 0) or nearest (balance = 1) of them onto the image border around their centre of
    mass, divided by fov_scale; the y axis is handled in units of the aspect ratio fx/fy.  Host float64 arithmetic in
    OpenCV's operation order.  Returns the 3x3 new camera matrix (float64)."""
    Km = np.asarray(K, np.float64).reshape(3, 3)
    w, h = int(image_size[0]), int(image_size[1])
    balance = min(max(float(balance), 0.0), 1.0)
    pts = np.array([[w // 2, 0], [w, h // 2], [w // 2, h], [0, h // 2]], np.float64)
    und = fisheye_undistortPoints(pts[None], Km, D, R=R, device=device)[0]
    cn = [((und[0, 0] + und[1, 0]) + und[2, 0] + und[3, 0]) / 4.0, ((und[0, 1] + und[1, 1]) + und[2, 1] + und[3, 1]) / 4.0]
    aspect_ratio = Km[0, 0] / Km[1, 1]
    cn[1] *= aspect_ratio                                     # convert to identity ratio
    py = und[:, 1] * aspect_ratio
    minx, maxx, miny, maxy = und[:, 0].min(), und[:, 0].max(), py.min(), py.max()
    f1 = w * 0.5 / (cn[0] - minx); f2 = w * 0.5 / (maxx - cn[0])
    f3 = h * 0.5 * aspect_ratio / (cn[1] - miny); f4 = h * 0.5 * aspect_ratio / (maxy - cn[1])
    fmin, fmax = min(f1, f2, f3, f4), max(f1, f2, f3, f4)
    f = balance * fmin + (1.0 - balance) * fmax
    f *= 1.0 / fov_scale if fov_scale > 0 else 1.0
    new_f = [f, f / aspect_ratio]                             # restore aspect ratio
    new_c = [-cn[0] * f + w * 0.5, (-cn[1] * f + (h * aspect_ratio) * 0.5) / aspect_ratio]
    if new_size is not None and int(new_size[0]) * int(new_size[1]) > 0:
        rx, ry = new_size[0] / float(w), new_size[1] / float(h)
        new_f = [new_f[0] * rx, new_f[1] * ry]; new_c = [new_c[0] * rx, new_c[1] * ry]
    return np.array([[new_f[0], 0.0, new_c[0]], [0.0, new_f[1], new_c[1]], [0.0, 0.0, 1.0]])


def cubic_table():
    """Host copy of the library's 1024x16 int16 bicubic table (no GPU needed)."""
    out = np.empty((1024, 16), np.int16)
    _lib.check(_lib.load().gf_get_cubic_table(out.ctypes.data_as(_vp)))
    return out
