"""BASELINE config C4 (4K 4:3 -> undistort+rotate to 2880x2160 -> NonlinearStretch superview to 3840x2160) on the GPU:
  * the remap step at full size (hero9_4k_43, 4000x3000 -> 2880x2160): fused == maps + remap, bit for bit;
  * the stretch step at full size (2880x2160 -> 3840x2160, vertical scale exactly 1 -- the case gf_superview_cubic_u8c3's
    fast loop is written for): equals the general bicubic kernel fed the full float32 maps, bit for bit, for the three border
    modes, with and without a row pitch that is a multiple of 4; the maps are the reference's own
    (tests/golden/stretch_maps.npz c4b.*, captured from nonlinear_stretch.py:58-100);
  * the chain at a size the CPU oracle handles: remap -> uint8 -> bicubic stretch against oracle remap_linear -> remap_cubic;
  * the one-launch fused chain (gf_undistort_rotate_superview_u8c3) == the two launches, bit for bit."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, preset_path

pytestmark = pytest.mark.gpu


def _c4_maps():
    z = np.load(os.path.join(GOLDEN, "stretch_maps.npz"))
    return z["c4b.map1_row0"].copy(), z["c4b.map2_col0"].copy()


def _trace(n, start=200, step=37):
    return np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))["stab_transform"][start:start + step * n:step, 3:7]


@pytest.mark.parametrize("border", [0, 1])
def test_c4_remap_step_full_size(gpu, border):
    from gyroflow_amd import FisheyeCalibrator, synthetic_frames, cv
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero9_4k_43"))
    w, h, ow, oh = 4000, 3000, 2880, 2160
    q = _trace(2)
    frames = torch.from_numpy(np.stack(synthetic_frames(w, h, n=2, seed=41))).to(gpu)
    xf = cv.xforms_to_device(cal.frame_xforms(1.0, q, output_dim=(ow, oh), new_img_dim=(w, h)), gpu)
    fused = cv.undistort_rotate(frames, cal.D, xf, (ow, oh), borderMode=border, borderValue=(3, 2, 1))
    for i in range(2):
        m1, m2 = cal.get_maps(1.0, new_img_dim=(w, h), output_dim=(ow, oh), update_new_K=False, quat=q[i], device=gpu)
        assert torch.equal(fused[i], cv.remap(frames[i], m1, m2, cv.INTER_LINEAR, borderMode=border, borderValue=(3, 2, 1)))
    assert int(fused.to(torch.int64).sum()) > 0


@pytest.mark.parametrize("border", [0, 1, 5])
@pytest.mark.parametrize("pad", [0, 1])
def test_c4_stretch_step_full_size(gpu, border, pad):
    from gyroflow_amd import synthetic_frames, cv, NonlinearStretch
    mx, my = _c4_maps()
    ns = NonlinearStretch((2880, 2160), (3840, 2160)); ns.recompute_maps()
    np.testing.assert_array_equal(ns.map1[0], mx); np.testing.assert_array_equal(ns.map2[:, 0], my)      # the reference's own maps
    w, h = 2880, 2160
    ring = synthetic_frames(w, h, n=2, seed=43)
    big = torch.zeros((2, h, w + pad, 3), dtype=torch.uint8, device=gpu)          # pad = 1: row pitch 8643 bytes, not a multiple of 4
    big[:, :, :w] = torch.from_numpy(np.stack(ring)).to(gpu)
    src = big[:, :, :w]
    prev = torch.from_numpy(np.stack(synthetic_frames(3840, 2160, n=2, seed=44))).to(gpu)
    got = cv.superview_cubic(src, torch.from_numpy(mx).to(gpu), torch.from_numpy(my).to(gpu), dst=prev.clone(), borderMode=border,
                             borderValue=(9, 99, 199))
    m1 = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(mx[None, :], (2160, 3840)))).to(gpu)
    m2 = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(my[:, None], (2160, 3840)))).to(gpu)
    for i in range(2):
        want = cv.remap(src[i].contiguous(), m1, m2, cv.INTER_CUBIC, dst=prev[i].clone(), borderMode=border, borderValue=(9, 99, 199))
        assert torch.equal(got[i], want), "frame %d differs from the general bicubic kernel" % i


def _small_c4(gpu, n=2):
    from gyroflow_amd import FisheyeCalibrator, synthetic_frames, NonlinearStretch
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero9_4k_43"))
    w, h, ow, oh, fw = 500, 375, 360, 270, 480              # C4 divided by 8
    frames = np.stack(synthetic_frames(w, h, n=n, seed=45))
    q = _trace(n, start=400)
    ns = NonlinearStretch((ow, oh), (fw, oh)); ns.recompute_maps()
    return cal, frames, q, ns, (w, h), (ow, oh), (fw, oh)


def test_c4_chain_small_vs_oracle(gpu, orc):
    from gyroflow_amd import cv
    cal, frames, q, ns, (w, h), (ow, oh), fsz = _small_c4(gpu)
    xf = cv.xforms_to_device(cal.frame_xforms(1.0, q, output_dim=(ow, oh), new_img_dim=(w, h)), gpu)
    mid = cv.undistort_rotate(torch.from_numpy(frames).to(gpu), cal.D, xf, (ow, oh))
    out = ns.apply_stretch_batch(mid).cpu().numpy()
    for i in range(len(frames)):
        K, D, R, P, od = cal._frame_matrices(1.0, (ow, oh), (w, h), q[i], None, True)
        o1, o2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2)
        want_mid = orc.remap_linear(frames[i], o1, o2)
        np.testing.assert_array_equal(mid[i].cpu().numpy(), want_mid)
        np.testing.assert_array_equal(out[i], orc.remap_cubic(want_mid, ns.map1, ns.map2))


@pytest.mark.parametrize("border", [0, 1])
def test_c4_fused_equals_two_launches_small(gpu, orc, border):
    """The one-launch chain against the two launches and against the oracle chain, 500x375 -> 360x270 -> 480x270 (tiles of
    every kind: interior, all four edges, partial last tile row / column)."""
    from gyroflow_amd import cv
    cal, frames, q, ns, (w, h), (ow, oh), fsz = _small_c4(gpu, n=3)
    src = torch.from_numpy(frames).to(gpu)
    xf = cv.xforms_to_device(cal.frame_xforms(1.0, q, output_dim=(ow, oh), new_img_dim=(w, h)), gpu)
    mid = cv.undistort_rotate(src, cal.D, xf, (ow, oh), borderMode=border, borderValue=(7, 8, 9))
    two = ns.apply_stretch_batch(mid)
    one = cv.undistort_rotate_superview(src, cal.D, xf, (ow, oh), ns.map1[0], ns.map2[:, 0], borderMode=border, borderValue=(7, 8, 9), fused=True)
    assert one.shape == two.shape
    bad = (one != two).any(dim=-1)
    assert not bool(bad.any()), "fused chain differs from the two launches at %s" % (torch.nonzero(bad)[:5].tolist(),)
    K, D, R, P, od = cal._frame_matrices(1.0, (ow, oh), (w, h), q[1], None, True)
    o1, o2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2)
    want = orc.remap_cubic(orc.remap_linear(frames[1], o1, o2, borderMode=border, borderValue=(7, 8, 9)), ns.map1, ns.map2)
    np.testing.assert_array_equal(one[1].cpu().numpy(), want)


def test_c4_fused_equals_two_launches_full_size(gpu):
    """C4 itself: 4000x3000 -> 2880x2160 -> 3840x2160 with the reference's own stretch maps, two frames of the real trace; padded
    output rows (pitch not a multiple of 4: byte stores) and packed rows."""
    from gyroflow_amd import FisheyeCalibrator, synthetic_frames, cv
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero9_4k_43"))
    w, h, ow, oh = 4000, 3000, 2880, 2160
    mx, my = _c4_maps()
    q = _trace(2, start=700)
    frames = torch.from_numpy(np.stack(synthetic_frames(w, h, n=2, seed=47))).to(gpu)
    xf = cv.xforms_to_device(cal.frame_xforms(1.0, q, output_dim=(ow, oh), new_img_dim=(w, h)), gpu)
    mid = cv.undistort_rotate(frames, cal.D, xf, (ow, oh))
    two = cv.superview_cubic(mid, torch.from_numpy(mx).to(gpu), torch.from_numpy(my).to(gpu))
    one = cv.undistort_rotate_superview(frames, cal.D, xf, (ow, oh), mx, my, fused=True)
    assert torch.equal(one, cv.undistort_rotate_superview(frames, cal.D, xf, (ow, oh), mx, my))      # the default: two launches
    assert torch.equal(one, two)
    big = torch.full((2, 2160, 3841, 3), 55, dtype=torch.uint8, device=gpu)
    cv.undistort_rotate_superview(frames, cal.D, xf, (ow, oh), mx, my, dst=big[:, :, :3840], fused=True)
    assert torch.equal(big[:, :, :3840], two) and int((big[:, :, 3840] != 55).sum()) == 0


def test_c4_fused_other_maps_and_fallback(gpu):
    """Other stretch settings (safe area, exponent: other column ownership tables), a second call with other maps on the same
    context (the cached table is replaced), and maps outside the fused kernel's envelope (vertical scale != 1), which the
    Python call routes through the two launches."""
    from gyroflow_amd import cv, NonlinearStretch, _lib
    import ctypes as C
    cal, frames, q, ns0, (w, h), (ow, oh), fsz = _small_c4(gpu, n=2)
    src = torch.from_numpy(frames).to(gpu)
    xf = cv.xforms_to_device(cal.frame_xforms(1.0, q, output_dim=(ow, oh), new_img_dim=(w, h)), gpu)
    mid = cv.undistort_rotate(src, cal.D, xf, (ow, oh))
    for out_size, safe, expo in (((480, 270), 0.3, 1.7), ((640, 270), 0.0, 3.0), ((480, 270), 0.999, 2.0), ((400, 270), 0.0, 2.0), ((512, 288), 0.1, 2.0)):
        ns = NonlinearStretch((ow, oh), out_size, safe_area=0, expo=expo)
        ns.set_safe_area(safe)
        ns.recompute_maps()
        two = ns.apply_stretch_batch(mid)
        one = cv.undistort_rotate_superview(src, cal.D, xf, (ow, oh), ns.map1[0], ns.map2[:, 0], fused=True)
        assert torch.equal(one, two), (out_size, safe, expo)
    # the C ABI itself refuses what the kernel cannot do
    lib, ctx = _lib.load(), _lib.ctx(gpu.index or 0)
    ns = NonlinearStretch((ow, oh), (512, 288)); ns.recompute_maps()
    mx = np.ascontiguousarray(ns.map1[0]); my = np.ascontiguousarray(ns.map2[:, 0])
    out = torch.empty((2, 288, 512, 3), dtype=torch.uint8, device=gpu)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.gf_undistort_rotate_superview_u8c3(ctx, p(src), w, h, w * 3, w * h * 3, ow, oh, p(out), 512, 288, 512 * 3, 512 * 288 * 3,
                                                _lib.distortion(cal.D), p(xf), 2, mx.ctypes.data_as(C.c_void_p), my.ctypes.data_as(C.c_void_p), 0,
                                                _lib.border_value(0), None)
    assert rc == -3 and b"unit-step" in lib.gf_last_error()
