"""cv2.rotate(frame, video_rotate_code) of the render loop (/root/reference/stabilizer.py:1385-1386; the code comes from the
constructor, :89,153, or from a Gyroflow data file, :1688) on the GPU: as a kernel of its own (gf_rotate_c3), and folded into the
fused kernel's tap addressing (gf_undistort_rotate_rot_c3).  Pure index work, so the bar is bit-exact: against numpy.rot90,
against rotate-then-remap on the GPU, and -- through renderfile -- against the reference loop on the CPU oracle fed with
numpy-rotated frames."""
import numpy as np
import pytest
import torch

from conftest import preset_path
from test_gpu_stabilizer import reference_loop, _quats

pytestmark = pytest.mark.gpu

# cv2.ROTATE_90_CLOCKWISE = 0, cv2.ROTATE_180 = 1, cv2.ROTATE_90_COUNTERCLOCKWISE = 2 as numpy.rot90 quarter turns (counter-clockwise)
ROT90_K = {0: 3, 1: 2, 2: 1}


def np_rotate(a, code):
    return np.ascontiguousarray(np.rot90(a, k=ROT90_K[code], axes=(-3, -2)))


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_rotate_equals_numpy_rot90(gpu, code, dtype):
    from gyroflow_amd import cv
    rng = np.random.default_rng(code)
    a = rng.integers(0, np.iinfo(dtype).max, size=(3, 37, 61, 3), endpoint=True).astype(dtype)
    got = cv.rotate(torch.from_numpy(a).to(gpu), code)
    np.testing.assert_array_equal(got.cpu().numpy(), np_rotate(a, code))
    np.testing.assert_array_equal(cv.rotate(a[0], code), np_rotate(a[0], code))           # numpy in -> numpy out, one frame
    # a padded row pitch on the source side
    pad = torch.zeros((3, 37, 70, 3), dtype=torch.from_numpy(a).dtype, device=gpu)
    pad[:, :, :61] = torch.from_numpy(a).to(gpu)
    np.testing.assert_array_equal(cv.rotate(pad[:, :, :61], code).cpu().numpy(), np_rotate(a, code))
    with pytest.raises(ValueError):
        cv.rotate(a, 3)


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("border", [0, 1, 5])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_fused_on_rotated_source_equals_rotate_then_fused(gpu, code, border, dtype):
    """Strong rotations and a wide field of view: interior tiles, every border case, frames whose width is not a multiple of
    four.  The rotated form must give the bytes of gf_rotate_c3 + gf_undistort_rotate_c3."""
    from gyroflow_amd import cv, FisheyeCalibrator, synthetic_frames
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(100 + 10*code + border)
    sw, sh = 963, 541
    n = 5
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero7_4k_169"))
    frames = np.stack(synthetic_frames(sw, sh, n=n, dtype=dtype, seed=code))
    rw, rh = (sw, sh) if code == 1 else (sh, sw)
    out_size = (1000, 520) if code == 1 else (560, 900)
    q = Rotation.from_euler("zyx", rng.normal(0, [6.0, 3.0, 3.0], (n, 3)), degrees=True).as_quat()[:, [3, 0, 1, 2]]
    fov = rng.uniform(0.8, 3.0, n)
    xf = cv.xforms_to_device(cal.frame_xforms(fov, q, output_dim=out_size, new_img_dim=(rw, rh)), gpu)
    src = torch.from_numpy(frames).to(gpu)
    bv = [17, 130, 250] if dtype == np.uint8 else [1700, 13000, 65000]
    seed = torch.full((n, out_size[1], out_size[0], 3), 99, dtype=src.dtype, device=gpu)      # BORDER_TRANSPARENT keeps these
    want = cv.undistort_rotate(cv.rotate(src, code), cal.D, xf, out_size, dst=seed.clone(), borderMode=border, borderValue=bv)
    got = cv.undistort_rotate(src, cal.D, xf, out_size, dst=seed.clone(), borderMode=border, borderValue=bv, rotate=code)
    assert torch.equal(got, want)
    if border == 5:
        print("BORDER_TRANSPARENT: %d output pixels left alone" % int((got == 99).all(dim=-1).sum()))
    # and against the rotated copy made by numpy, to keep gf_rotate_c3 out of the comparison
    want2 = cv.undistort_rotate(torch.from_numpy(np_rotate(frames, code)).to(gpu), cal.D, xf, out_size, dst=seed.clone(), borderMode=border, borderValue=bv)
    assert torch.equal(got, want2)


N_RANDOM = int(__import__("os").environ.get("GF_SWEEP_CASES", "36"))


@pytest.mark.parametrize("seed", range(N_RANDOM))
def test_rotated_fused_random_case(gpu, seed):
    """The sweep of tests/test_gpu_sweep.py (sizes from 1 pixel up, odd pitches, lenses, rotations past 90 degrees, every border,
    both depths) with the source stored turned: the rotated entry against numpy.rot90 + the plain entry, bit for bit."""
    from gyroflow_amd import cv, _lib
    from test_gpu_sweep import _case, _img
    c = _case(seed)
    rng, n, dtype, border = c["rng"], c["n"], c["dtype"], c["border"]
    code = int(rng.integers(0, 3))
    rw, rh, dw, dh = c["sw"], c["sh"], c["dw"], c["dh"]            # the frame the lens model is for: the turned one
    turned = np.stack([_img(rng, rh, rw, dtype) for _ in range(n)])
    stored = np.rot90(turned, k=-ROT90_K[code], axes=(-3, -2)).copy()                     # what the file holds: cv2.rotate undone
    np.testing.assert_array_equal(np_rotate(stored, code), turned)
    prev = np.stack([_img(rng, dh, dw, dtype) for _ in range(n)])
    bv = (7, 200, 33) if dtype == np.uint8 else (700, 65535, 12345)
    tdt = torch.uint8 if dtype == np.uint8 else torch.uint16
    sth, stw = stored.shape[1:3]
    big_s = torch.zeros((n, sth, stw + c["spad"], 3), dtype=tdt, device=gpu)               # padded, possibly odd, row pitch
    big_s[:, :, :stw] = torch.from_numpy(stored).to(gpu)
    xf = cv.xforms_to_device(np.stack([_lib.xform_to_array(_lib.make_frame_xform(c["K"], R, c["P"])) for R in c["Rs"]]), gpu)
    got = cv.undistort_rotate(big_s[:, :, :stw], c["D"], xf, (dw, dh), dst=torch.from_numpy(prev.copy()).to(gpu), borderMode=border,
                              borderValue=bv, rotate=code)
    want = cv.undistort_rotate(torch.from_numpy(turned).to(gpu), c["D"], xf, (dw, dh), dst=torch.from_numpy(prev.copy()).to(gpu),
                               borderMode=border, borderValue=bv)
    assert torch.equal(got, want), "case %d, rotate code %d" % (seed, code)


@pytest.mark.parametrize("kmode", ["direct", "rl"])
def test_rotated_source_in_the_other_kernel_modes(gpu, kmode):
    """DIRECT / RL have no rotated form: cv.undistort_rotate(rotate=...) rotates the batch first; same bytes as that mode's plain call."""
    from gyroflow_amd import cv, _lib, FisheyeCalibrator, synthetic_frames
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero7_4k_169"))
    frames = np.stack(synthetic_frames(640, 360, n=2))
    q = np.array([[0.999, 0.02, -0.03, 0.01], [0.998, -0.03, 0.02, 0.04]]); q /= np.linalg.norm(q, axis=1, keepdims=True)
    xf = cv.xforms_to_device(cal.frame_xforms(np.ones(2), q, output_dim=(400, 600), new_img_dim=(360, 640)), gpu)
    src = torch.from_numpy(frames).to(gpu)
    _lib.set_kernel_mode(gpu.index or 0, {"direct": _lib.KERNEL_DIRECT, "rl": _lib.KERNEL_RL}[kmode])
    try:
        got = cv.undistort_rotate(src, cal.D, xf, (400, 600), rotate=cv.ROTATE_90_CLOCKWISE)
        want = cv.undistort_rotate(torch.from_numpy(np_rotate(frames, 0)).to(gpu), cal.D, xf, (400, 600))
    finally:
        _lib.set_kernel_mode(gpu.index or 0, _lib.KERNEL_DEFAULT)
    assert torch.equal(got, want)


@pytest.mark.parametrize("code", [0, 1, 2])
@pytest.mark.parametrize("mode", [
    dict(bg="#000000", mult=1, nblend=1),
    dict(bg="#203040", mult=4, nblend=2, batch=6),
    dict(bg="REPLICATE", mult=1, nblend=1, dtype=np.uint16, batch=5),
    dict(bg="HISTORY", mult=1, nblend=1),
    dict(bg="#000000", mult=2, nblend=2, dtype=np.uint16),
])
def test_renderfile_with_video_rotation(gpu, orc, code, mode):
    """Stabilizer(video_rotation=code).renderfile on the decoded frames == renderfile on numpy-rotated frames (the un-rotated
    path) == the reference loop on the oracle fed with the numpy-rotated frames, bit for bit.  As in the reference (:164-165),
    orig_dimension stays the decoded size for the 90 degree codes."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, synthetic_frames
    dtype = mode.get("dtype", np.uint8)
    w, h = 338, 254
    n = 22
    frames = synthetic_frames(w, h, n=n, dtype=dtype, seed=11)
    rframes = [np_rotate(f, code) for f in frames]
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    stab = _quats(n + 5, seed=5)
    fps, start, stop = 10.0, 0.1, 2.0
    out_size = (240, 136) if code == 1 else (150, 230)
    kw = dict(out_size=out_size, fov_scale=1.2, smoothingFocus=-2, bg_color=mode["bg"], batch=mode.get("batch", 8), pipeline_depth=2)

    def render(fr, rot):
        st = Stabilizer(ArraySource(fr), cal, stab, fps, w, h, device=gpu, video_rotation=rot)
        st.set_hyperlapse(mode["mult"], mode["nblend"])
        sink = ListSink()
        assert st.renderfile(start, stop, sink, **kw) == len(sink.frames) > 0
        return sink.frames

    got = render(frames, code)
    plain = render(rframes, -1)
    want = reference_loop(orc, cal, rframes, set(), stab, fps, start, stop, out_size, 1.2, mode["bg"], mode["mult"], mode["nblend"], (w, h))
    assert len(got) == len(plain) == len(want)
    for a, b, c in zip(got, plain, want):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)


def test_gyroflow_file_rotation_code_is_applied(gpu, tmp_path):
    """import_gyroflow_file stores video_rotate_code (:1688); renderfile must honour it (round 2 dropped it silently)."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, synthetic_frames
    w, h = 200, 120
    frames = synthetic_frames(w, h, n=8)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    stab = _quats(12, seed=2)
    a = Stabilizer(ArraySource(frames), cal, stab, 10.0, w, h, device=gpu)
    a.video_rotate_code = 1                                   # what import_gyroflow_file assigns
    b = Stabilizer(ArraySource([np_rotate(f, 1) for f in frames]), cal, stab, 10.0, w, h, device=gpu)
    sa, sb = ListSink(), ListSink()
    kw = dict(out_size=(160, 96), smoothingFocus=-2)
    assert a.renderfile(0.0, 0.8, sa, **kw) == b.renderfile(0.0, 0.8, sb, **kw) > 0
    for x, y in zip(sa.frames, sb.frames):
        np.testing.assert_array_equal(x, y)
    a.video_rotate_code = 7
    assert a.renderfile(0.0, 0.8, ListSink(), **kw) is None    # prints and returns, like the reference's other early exits
