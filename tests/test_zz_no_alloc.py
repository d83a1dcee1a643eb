"""gf_undistort_rotate_c3_noalloc (include/gfremap.h): the allocation-free variant of the fused entry point, called through
the C ABI, must write exactly the bytes gf_undistort_rotate_c3 writes -- on two calls in a row, the second with a smaller
input (fewer, smaller frames, another lens), since every call on a GPU reuses that GPU's static scratch array.  Collected
last on purpose (see the task notes: a failure here must not hide other tests)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import preset_path

pytestmark = pytest.mark.gpu


def _case(gpu, preset, w, h, ow, oh, n, dtype, seed):
    from gyroflow_amd import FisheyeCalibrator, synthetic_frames
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path(preset))
    rng = np.random.default_rng(seed)
    q = np.concatenate([np.ones((n, 1)), rng.normal(0, 0.03, (n, 3))], axis=1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    xf = np.ascontiguousarray(cal.frame_xforms(1.0, q, output_dim=(ow, oh), new_img_dim=(w, h)))
    ring = synthetic_frames(w, h, n=2, dtype=dtype, seed=seed)
    frames = torch.from_numpy(np.stack([ring[i % 2] for i in range(n)])).to(gpu)
    return cal, xf, frames


def _both(gpu, cal, xf, frames, ow, oh, border, bv):
    """-> (output of gf_undistort_rotate_c3, output of gf_undistort_rotate_c3_noalloc), both straight through the C ABI."""
    from gyroflow_amd import _lib, cv
    lib, ctx = _lib.load(), _lib.ctx(gpu.index or 0)
    n, h, w = frames.shape[:3]
    esz = frames.element_size()
    depth = _lib.DEPTH_U8 if frames.dtype == torch.uint8 else _lib.DEPTH_U16
    p = lambda t: C.c_void_p(t.data_ptr())
    prev = torch.full((n, oh, ow, 3), 77, dtype=frames.dtype, device=gpu)
    a, b = prev.clone(), prev.clone()
    D, B = _lib.distortion(cal.D), _lib.border_value(bv)
    st = C.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    xf_dev = cv.xforms_to_device(xf, gpu)
    _lib.check(lib.gf_undistort_rotate_c3(ctx, depth, p(frames), w, h, w * 3 * esz, h * w * 3 * esz, p(a), ow, oh, ow * 3 * esz, oh * ow * 3 * esz,
                                          D, p(xf_dev), n, border, B, st))
    _lib.check(lib.gf_undistort_rotate_c3_noalloc(gpu.index or 0, depth, p(frames), w, h, w * 3 * esz, h * w * 3 * esz, p(b), ow, oh, ow * 3 * esz,
                                                  oh * ow * 3 * esz, D, xf.ctypes.data_as(C.c_void_p), n, border, B, st))
    torch.cuda.synchronize(gpu)
    return a, b


def test_noalloc_equals_existing_entry_two_calls(gpu):
    # first call: 5 frames of 1920x1080 u8; second call: 2 smaller u16 frames of another lens, another border mode
    cal, xf, frames = _case(gpu, "hero8_27k_43", 1920, 1080, 1920, 1080, 5, np.uint8, 3)
    a, b = _both(gpu, cal, xf, frames, 1920, 1080, 0, (9, 8, 7))
    assert torch.equal(a, b)
    assert int((a != 77).sum()) > a.numel() // 2
    cal2, xf2, frames2 = _case(gpu, "hero7_4k_169", 960, 540, 800, 448, 2, np.uint16, 4)
    a2, b2 = _both(gpu, cal2, xf2, frames2, 800, 448, 1, 0)
    assert torch.equal(a2, b2)
    # and the first input again: nothing of the second call's tables may linger
    a3, b3 = _both(gpu, cal, xf, frames, 1920, 1080, 0, (9, 8, 7))
    assert torch.equal(a3, a) and torch.equal(b3, a)


def test_noalloc_call_leaves_device_memory_alone(gpu):
    """After the warm-up call on the device (which resolves the static array's address), a call changes neither the free device
    memory the driver reports (hipMemGetInfo) nor what torch's allocator holds."""
    from gyroflow_amd import _lib
    lib = _lib.load()
    cal, xf, frames = _case(gpu, "hero8_27k_43", 960, 540, 960, 540, 3, np.uint8, 6)
    dst = torch.zeros((3, 540, 960, 3), dtype=torch.uint8, device=gpu)
    p = lambda t: C.c_void_p(t.data_ptr())
    D, B = _lib.distortion(cal.D), _lib.border_value(0)
    st = C.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    call = lambda: lib.gf_undistort_rotate_c3_noalloc(gpu.index or 0, 0, p(frames), 960, 540, 960 * 3, 540 * 960 * 3, p(dst), 960, 540, 960 * 3,
                                                      540 * 960 * 3, D, xf.ctypes.data_as(C.c_void_p), 3, 0, B, st)
    assert call() == 0                       # warm-up
    torch.cuda.synchronize(gpu)
    free0, _ = torch.cuda.mem_get_info(gpu)
    held0, reserved0 = torch.cuda.memory_allocated(gpu), torch.cuda.memory_reserved(gpu)
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize(gpu)
    free1, _ = torch.cuda.mem_get_info(gpu)
    assert free1 == free0 and torch.cuda.memory_allocated(gpu) == held0 and torch.cuda.memory_reserved(gpu) == reserved0


def test_noalloc_rejects_oversize_and_bad_arguments(gpu):
    from gyroflow_amd import _lib
    lib = _lib.load()
    cal, xf, frames = _case(gpu, "hero8_27k_43", 64, 48, 64, 48, 1, np.uint8, 5)
    big = np.tile(xf, (65, 1))
    src = torch.zeros((1, 48, 64, 3), dtype=torch.uint8, device=gpu)
    dst = torch.zeros((65, 48, 64, 3), dtype=torch.uint8, device=gpu)
    p = lambda t: C.c_void_p(t.data_ptr())
    D, B = _lib.distortion(cal.D), _lib.border_value(0)
    call = lambda n, dev=0, x=big: lib.gf_undistort_rotate_c3_noalloc(dev, 0, p(src), 64, 48, 192, 0, p(dst), 64, 48, 192, 64 * 48 * 3, D,
                                                                      None if x is None else x.ctypes.data_as(C.c_void_p), n, 0, B, None)
    assert call(65) == -1 and b"GF_NOALLOC_MAX_FRAMES" in lib.gf_last_error()
    assert call(0) == -1 and call(1, x=None) == -1 and call(1, dev=9999) == -1
    assert call(64) == 0            # the maximum itself works (src stride 0: one input frame for all)
    torch.cuda.synchronize(gpu)
    assert int(dst[63].to(torch.int64).sum()) == int(dst[0].to(torch.int64).sum())
