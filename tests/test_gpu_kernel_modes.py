"""Every variant of the fused kernels and of the CV_16SC2 map kernel (gf_ctx_set_kernel_mode): DIRECT = per-pixel lens
table, RP = coordinates from per-tile polynomials on 64 x 64 tiles (the default), RL = RP + source taps staged through LDS
by LDS-DMA.  Same bars as the default
variant: fixed-point maps identical to the oracle except < 2e-6 of entries by one Q5 step, fused output == maps + remap
bit for bit, pixels identical to the oracle chain wherever the maps agree.  Sizes are realistic on purpose: the polynomial
path needs an output focal length of at least 256 px (smaller frames take the per-pixel path, covered elsewhere)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, preset_path

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["direct", "rp", "rl"])
def mode(request, gpu):
    from gyroflow_amd import _lib
    _lib.set_kernel_mode(gpu.index or 0, {"direct": _lib.KERNEL_DIRECT, "rp": _lib.KERNEL_RP, "rl": _lib.KERNEL_RL}[request.param])
    yield request.param
    _lib.set_kernel_mode(gpu.index or 0, _lib.KERNEL_DEFAULT)


def _q5(m1, m2):
    return (m1.astype(np.int64) << 5) + np.stack([m2 & 31, m2 >> 5], -1)


def test_modes_4k_vs_oracle(gpu, orc, golden_calls, mode):
    """C3 at full size: maps and fused output against the oracle; fused == maps + remap on the GPU."""
    from gyroflow_amd import cv, _lib, synthetic_frames
    z, _ = golden_calls
    K, D, R, P = z["c3_4k.K"], z["c3_4k.D"], z["c3_4k.R"], z["c3_4k.P"]
    frame = synthetic_frames(3840, 2160, n=1)[0]
    o1, o2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, (3840, 2160), orc.CV_16SC2)
    g1, g2 = cv.fisheye_initUndistortRectifyMap(K, D, R, P, (3840, 2160), cv.CV_16SC2, device=gpu)
    diff = np.abs(_q5(g1.cpu().numpy(), g2.cpu().numpy()) - _q5(o1, o2))
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-6
    src = torch.from_numpy(frame).to(gpu)
    xf = cv.xforms_to_device(_lib.xform_to_array(_lib.make_frame_xform(K, R, P))[None], gpu)
    got = cv.undistort_rotate(src, D, xf, (3840, 2160))
    got = got.reshape(2160, 3840, 3)
    assert torch.equal(got, cv.remap(src, g1, g2, cv.INTER_LINEAR))
    want = orc.remap_linear(frame, o1, o2, nthreads=8)
    pd = np.abs(got.cpu().numpy().astype(np.int32) - want.astype(np.int32))
    print("C3 %s: %d map entries differ by one Q5 step; %d pixel values differ, max |diff| = %d LSB" % (mode, int((diff != 0).sum()), int((pd != 0).sum()), int(pd.max())))
    assert int((pd != 0).sum()) <= 24
    # a one-step Q5 flip moves a pixel by at most 1/32 of the local contrast: <= 8 LSB for u8
    assert int(pd.max()) <= 8


@pytest.mark.parametrize("border", [0, 1, 5])
def test_modes_borders_and_strong_rotation(gpu, orc, mode, border):
    """960x540 frames, rotations up to 20 degrees: large parts of the output look outside the frame (border paths), the
    source box of many tiles does not fit LDS or is clamped (RL falls back per tile / per pixel)."""
    from gyroflow_amd import cv, _lib
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(7 + border)
    sw, sh, dw, dh = 960, 540, 1000, 520
    K = np.array([[460., 0, 480], [0, 455., 270], [0, 0, 1]]); D = [0.04, 0.01, -0.02, 0.005]
    P = np.array([[430., 0, 500], [0, 430., 260], [0, 0, 1]])
    angles = [(0, 0, 0), (20, -3, 4), (-8, 12, -15), (2, 1, -1)]
    Rs = [Rotation.from_euler("zyx", a, degrees=True).as_matrix() for a in angles]
    frames = rng.integers(0, 256, (len(Rs), sh, sw, 3), dtype=np.uint8)
    prev = rng.integers(0, 256, (len(Rs), dh, dw, 3), dtype=np.uint8)
    xf = cv.xforms_to_device(np.stack([_lib.xform_to_array(_lib.make_frame_xform(K, R, P)) for R in Rs]), gpu)
    src = torch.from_numpy(frames).to(gpu)
    dst = torch.from_numpy(prev.copy()).to(gpu)
    fused = cv.undistort_rotate(src, D, xf, (dw, dh), dst=dst, borderMode=border, borderValue=(9, 99, 199)).cpu().numpy()
    for i, R in enumerate(Rs):
        g1, g2 = cv.fisheye_initUndistortRectifyMap(K, D, R, P, (dw, dh), cv.CV_16SC2, device=gpu)
        two = cv.remap(src[i], g1, g2, cv.INTER_LINEAR, dst=torch.from_numpy(prev[i].copy()).to(gpu), borderMode=border,
                       borderValue=(9, 99, 199)).cpu().numpy()
        np.testing.assert_array_equal(fused[i], two)
        o1, o2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, (dw, dh), orc.CV_16SC2)
        same = (g1.cpu().numpy() == o1).all(-1) & (g2.cpu().numpy() == o2)
        assert same.mean() > 1 - 1e-5
        want = orc.remap_linear(frames[i], o1, o2, dst=prev[i].copy(), borderMode=border, borderValue=(9, 99, 199))
        np.testing.assert_array_equal(fused[i][same], want[same])


def test_modes_batch_u8_u16_and_unaligned(gpu, mode):
    """C1-sized batches from the real gyro trace: u8 (aligned: LDS box in RL), u8 with a row pitch that is not a multiple of
    16 (RL runs as RP) and u16 (RP coordinates): fused == maps + remap, frame by frame."""
    from gyroflow_amd import FisheyeCalibrator, synthetic_frames, cv
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    w, h, ow, oh = 1920, 1080, 1920, 1080
    q = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))["stab_transform"][300:300 + 40 * 3:40, 3:7]
    xf = cal.frame_xforms(1.0, q, output_dim=(ow, oh), new_img_dim=(w, h))
    xf_dev = cv.xforms_to_device(xf, gpu)
    for dtype, pad in ((np.uint8, 0), (np.uint8, 3), (np.uint16, 0)):
        ring = synthetic_frames(w, h, n=3, dtype=dtype, seed=23)
        big = torch.zeros((3, h, w + pad, 3), dtype=torch.from_numpy(ring[0]).dtype, device=gpu)
        big[:, :, :w] = torch.from_numpy(np.stack(ring)).to(gpu)
        frames = big[:, :, :w]
        fused = cv.undistort_rotate(frames, cal.D, xf_dev, (ow, oh), borderMode=1)
        for i in range(3):
            m1, m2 = cal.get_maps(1.0, new_img_dim=(w, h), output_dim=(ow, oh), update_new_K=False, quat=q[i], device=gpu)
            assert torch.equal(fused[i], cv.remap(frames[i], m1, m2, cv.INTER_LINEAR, borderMode=1)), (dtype, pad, i)


@pytest.mark.parametrize("dtype,tdtype", [(np.uint8, torch.uint8), (np.uint16, torch.uint16)])
def test_modes_fused_blend_equals_separate_passes(gpu, mode, dtype, tdtype):
    """Hyperlapse: the fused remap + blend kernel equals remap, float64 accumulate and truncating cast done as separate passes,
    bit for bit, in every mode (1920x1080, 3 input frames per output frame, 2 output frames)."""
    from gyroflow_amd import FisheyeCalibrator, synthetic_frames, cv
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    w, h, ow, oh = 1920, 1080, 1920, 1080
    q = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))["stab_transform"][500:500 + 30 * 7:30, 3:7]
    xf_dev = cv.xforms_to_device(cal.frame_xforms(1.0, q, output_dim=(ow, oh), new_img_dim=(w, h)), gpu)
    ring = synthetic_frames(w, h, n=2, dtype=dtype, seed=31)
    frames = torch.from_numpy(np.stack([ring[i % 2] for i in range(7)])).to(gpu)
    fused = cv.undistort_rotate_blend(frames, cal.D, xf_dev, (ow, oh), 4, 3, borderMode=1)      # groups of 4 frames, 3 blended
    assert fused.shape == (2, oh, ow, 3)
    remapped = cv.undistort_rotate(frames, cal.D, xf_dev, (ow, oh), borderMode=1)
    for g in range(2):
        acc = torch.zeros((oh, ow, 3), dtype=torch.float64, device=gpu)
        for b in range(3):
            cv.blend_accumulate(remapped[4 * g + b], acc, 1 / 3, b == 0)
        assert torch.equal(fused[g], cv.blend_finalize(acc, tdtype)), (mode, g)


def test_modes_mixed_tiles_wide_angle(gpu, orc, mode):
    """A 65 degree yaw: part of the output looks behind the virtual camera (_w <= 0), part beyond the lens table, part
    is ordinary -- tiles of all three kinds (polynomial path, per-pixel table path, exact path) meet in one frame."""
    from gyroflow_amd import cv, _lib
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(99)
    sw, sh, dw, dh = 1280, 720, 1400, 800
    K = np.array([[600., 0, 640], [0, 600., 360], [0, 0, 1]]); D = [0.03, 0.02, -0.01, 0.002]
    P = np.array([[520., 0, 700], [0, 520., 400], [0, 0, 1]])
    for ang in ((0, 65, 0), (10, -50, 20)):
        R = Rotation.from_euler("zyx", ang, degrees=True).as_matrix()
        frame = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        o1, o2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, (dw, dh), orc.CV_16SC2)
        g1, g2 = cv.fisheye_initUndistortRectifyMap(K, D, R, P, (dw, dh), cv.CV_16SC2, device=gpu)
        diff = np.abs(_q5(g1.cpu().numpy(), g2.cpu().numpy()) - _q5(o1, o2))
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-5, (ang, int(diff.max()), float((diff != 0).mean()))
        src = torch.from_numpy(frame).to(gpu)
        xf = cv.xforms_to_device(_lib.xform_to_array(_lib.make_frame_xform(K, R, P))[None], gpu)
        got = cv.undistort_rotate(src, D, xf, (dw, dh), borderMode=0, borderValue=(1, 2, 3)).reshape(dh, dw, 3)
        assert torch.equal(got, cv.remap(src, g1, g2, cv.INTER_LINEAR, borderMode=0, borderValue=(1, 2, 3)))
        same = (g1.cpu().numpy() == o1).all(-1) & (g2.cpu().numpy() == o2)
        want = orc.remap_linear(frame, o1, o2, borderMode=0, borderValue=(1, 2, 3))
        np.testing.assert_array_equal(got.cpu().numpy()[same], want[same])


def test_set_kernel_mode_rejects_unknown_mode(gpu):
    from gyroflow_amd import _lib
    with pytest.raises(_lib.GfError) as e:
        _lib.set_kernel_mode(gpu.index or 0, 17)
    assert "unknown mode" in str(e.value)
    lib = _lib.load()
    assert lib.gf_ctx_set_kernel_mode(None, _lib.KERNEL_RP) != 0          # null ctx
