"""Size-independent properties at BASELINE.json's full frame sizes (the CPU oracle is too slow there):
  * the fused kernel equals materialised maps + remap on the GPU, bit for bit (both were checked against the oracle at
    small sizes in test_gpu_parity.py);
  * a batched call equals frame-by-frame calls (checksum of checksums), i.e. frames are independent -> shardable;
  * the fused hyperlapse blend equals remap + float64 accumulate + truncating cast done as separate passes."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, preset_path

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (preset, in WxH, out WxH, dtype, fov_scale)
    "C1": ("hero8_27k_43", (1920, 1080), (1920, 1080), np.uint8, 1.0),
    "C2": ("hero8_27k_43", (2704, 2028), (1920, 1080), np.uint8, 1.0),
    "C5": ("hero9_5k_169", (5312, 2988), (5312, 2988), np.uint16, 1.0),
}


def _setup(name, gpu, n):
    from gyroflow_amd import FisheyeCalibrator, synthetic_frames, cv
    preset, (w, h), (ow, oh), dtype, fov = CONFIGS[name]
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path(preset))
    q = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))["stab_transform"][300:300 + 40 * n:40, 3:7]
    ring = synthetic_frames(w, h, n=2, dtype=dtype, seed=17)
    frames = torch.from_numpy(np.stack([ring[i % 2] for i in range(n)])).to(gpu)
    xf = cal.frame_xforms(fov, q, output_dim=(ow, oh), new_img_dim=(w, h))
    return cal, q, frames, xf, (w, h), (ow, oh), fov


@pytest.mark.parametrize("name", ["C1", "C2", "C5"])
@pytest.mark.parametrize("border", [0, 1])
def test_fused_equals_maps_plus_remap_full_size(gpu, name, border):
    from gyroflow_amd import cv
    n = 3 if name != "C5" else 2
    cal, q, frames, xf, (w, h), (ow, oh), fov = _setup(name, gpu, n)
    fused = cv.undistort_rotate(frames, cal.D, cv.xforms_to_device(xf, gpu), (ow, oh), borderMode=border, borderValue=(5, 6, 7))
    for i in range(n):
        m1, m2 = cal.get_maps(fov, new_img_dim=(w, h), output_dim=(ow, oh), update_new_K=False, quat=q[i], device=gpu)
        two = cv.remap(frames[i], m1, m2, cv.INTER_LINEAR, borderMode=border, borderValue=(5, 6, 7))
        assert torch.equal(fused[i], two), "%s frame %d: fused kernel and maps+remap differ" % (name, i)
        # frame-by-frame call == its slice of the batched call (frames are independent)
        one = cv.undistort_rotate(frames[i:i + 1], cal.D, cv.xforms_to_device(xf[i:i + 1], gpu), (ow, oh), borderMode=border,
                                  borderValue=(5, 6, 7))
        assert torch.equal(one[0], fused[i])
    assert int(fused.to(torch.int64).sum()) > 0


def test_fused_blend_equals_separate_passes_c5(gpu):
    """C5: 5312x2988 u16, set_hyperlapse(4, 4): one output frame from four inputs."""
    from gyroflow_amd import cv
    cal, q, frames, xf, (w, h), (ow, oh), fov = _setup("C5", gpu, 4)
    xf_dev = cv.xforms_to_device(xf, gpu)
    fused = cv.undistort_rotate_blend(frames, cal.D, xf_dev, (ow, oh), 4, 4, borderMode=1)
    assert fused.shape == (1, oh, ow, 3)
    remapped = cv.undistort_rotate(frames, cal.D, xf_dev, (ow, oh), borderMode=1)
    acc = torch.zeros((oh, ow, 3), dtype=torch.float64, device=gpu)
    for b in range(4):
        cv.blend_accumulate(remapped[b], acc, 1 / 4, b == 0)
    want = cv.blend_finalize(acc, torch.uint16)
    assert torch.equal(fused[0], want)
    # torch float64 restatement of the numpy passes (stabilizer.py:1434,1447,1468)
    t = torch.zeros((oh, ow, 3), dtype=torch.float64, device=gpu)
    for b in range(4):
        t += (1 / 4) * remapped[b].to(torch.float64)
    assert torch.equal(want.to(torch.int32), t.to(torch.int32))


@pytest.mark.parametrize("border", [0, 1])
def test_stretched_preset_fused_equals_get_maps_plus_remap(gpu, border):
    """Stretched presets (calibrate_video.py:375-380) on the batched fused path: runcam5_stretched, 3840x2160 video ->
    1920x1080, per-frame rotations: gf_undistort_rotate_stretched_c3 == get_maps (float32 maps, numpy-style float32 rescale,
    convertMaps) + remap, bit for bit."""
    from gyroflow_amd import FisheyeCalibrator, synthetic_frames, cv
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("runcam5_stretched"))
    assert cal.image_is_stretched() and cal.map_scale() is not None
    w, h = (int(v) for v in cal.orig_dimension)
    ow, oh = 1920, 1080
    n = 3
    q = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))["stab_transform"][100:100 + 50 * n:50, 3:7]
    frames = torch.from_numpy(np.stack(synthetic_frames(w, h, n=n, seed=19))).to(gpu)
    fov = np.array([1.0, 1.3, 0.8])
    fc = np.array([[cal.calib_dimension[0] / 2 + 11 * i, cal.calib_dimension[1] / 2 - 7 * i] for i in range(n)], np.float64)
    xf = cal.frame_xforms(fov, q, output_dim=(ow, oh), new_img_dim=(w, h), focalCenters=fc)
    for i in range(n):
        np.testing.assert_array_equal(xf[i], cal.frame_xform(float(fov[i]), (ow, oh), (w, h), q[i], fc[i]))
    fused = cv.undistort_rotate(frames, cal.D, cv.xforms_to_device(xf, gpu), (ow, oh), borderMode=border, borderValue=(4, 5, 6),
                                map_scale=cal.map_scale())
    for i in range(n):
        m1, m2 = cal.get_maps(float(fov[i]), new_img_dim=(w, h), output_dim=(ow, oh), update_new_K=False, quat=q[i], focalCenter=fc[i], device=gpu)
        two = cv.remap(frames[i], m1, m2, cv.INTER_LINEAR, borderMode=border, borderValue=(4, 5, 6))
        assert torch.equal(fused[i], two), "frame %d" % i
