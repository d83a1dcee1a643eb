"""Seeded random sweep of the fused entry (gf_undistort_rotate_c3 through gyroflow_amd.cv) over the shapes the fixed cases
do not reach: source and output sizes from 1 pixel up, widths that are not multiples of 4 or of the 64-pixel tile, padded
rows whose pitch is odd, every border mode and both depths, lenses from mild to strong, rotations from a fraction of a
degree to past 90 degrees (rays behind the camera), several frames per call.

Per case three statements, the same ones tests/test_gpu_parity.py makes for its fixed cases:
  * fused == GPU maps + GPU remap, bit for bit;
  * GPU maps differ from the oracle's in at most one 1/32-px step at a vanishing fraction of entries (rounding ties);
  * where the maps agree, the fused pixels equal the oracle chain's.
"""
import os

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("GF_SWEEP_CASES", "36"))        # a one-off hunt can ask for thousands (seeds 0 .. N-1)


def _n(default):
    """seeds of the smaller sweeps: their default count, or GF_SWEEP_CASES when a hunt sets it"""
    return range(int(os.environ.get("GF_SWEEP_CASES", default)))


def _img(rng, h, w, dtype):
    if dtype == np.uint8:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    return rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)


def _case(seed):
    rng = np.random.default_rng(7000 + seed)
    small = seed % 4 == 0
    sw = int(rng.integers(1, 9)) if small else int(rng.integers(5, 200))
    sh = int(rng.integers(1, 9)) if small else int(rng.integers(5, 160))
    dw = int(rng.integers(1, 140)) if seed % 3 else int(rng.choice([1, 3, 63, 64, 65, 127, 129]))
    dh = int(rng.integers(1, 100)) if seed % 5 else int(rng.choice([1, 2, 63, 64, 65]))
    f = float(rng.uniform(0.35, 1.4)) * max(sw, sh, 8)
    K = np.array([[f, 0, sw / 2 + rng.uniform(-3, 3)], [0, f * rng.uniform(0.9, 1.1), sh / 2 + rng.uniform(-3, 3)], [0, 0, 1]])
    fo = f * float(rng.uniform(0.5, 1.6))
    P = np.array([[fo, 0, dw / 2 + rng.uniform(-2, 2)], [0, fo, dh / 2 + rng.uniform(-2, 2)], [0, 0, 1]])
    D = rng.uniform(-1, 1, 4) * np.array([0.3, 0.1, 0.05, 0.02]) * float(rng.choice([0.0, 0.2, 1.0]))
    n = int(rng.integers(1, 4))
    big = float(rng.choice([0.5, 5, 30, 100]))
    Rs = [Rotation.from_euler("zyx", rng.uniform(-1, 1, 3) * [big, big, big / 2], degrees=True).as_matrix() for _ in range(n)]
    return dict(rng=rng, sw=sw, sh=sh, dw=dw, dh=dh, K=K, P=P, D=D, Rs=Rs, n=n,
                dtype=[np.uint8, np.uint16][seed % 2], border=[0, 1, 5][seed % 3],
                spad=int(rng.integers(0, 4)), dpad=int(rng.integers(0, 4)))


@pytest.mark.parametrize("seed", range(N_CASES))
def test_fused_random_case(gpu, orc, seed):
    from gyroflow_amd import cv, _lib
    c = _case(seed)
    rng, sw, sh, dw, dh, n, dtype, border = c["rng"], c["sw"], c["sh"], c["dw"], c["dh"], c["n"], c["dtype"], c["border"]
    tdt = torch.uint8 if dtype == np.uint8 else torch.uint16
    frames = np.stack([_img(rng, sh, sw, dtype) for _ in range(n)])
    prev = np.stack([_img(rng, dh, dw, dtype) for _ in range(n)])
    bv = (7, 200, 33) if dtype == np.uint8 else (700, 65535, 12345)
    # padded rows: the row pitch of a u8 frame is then odd / not a multiple of 4
    big_s = torch.zeros((n, sh, sw + c["spad"], 3), dtype=tdt, device=gpu)
    big_s[:, :, :sw] = torch.from_numpy(frames).to(gpu)
    big_d = torch.zeros((n, dh, dw + c["dpad"], 3), dtype=tdt, device=gpu)
    big_d[:, :, :dw] = torch.from_numpy(prev).to(gpu)
    xf = cv.xforms_to_device(np.stack([_lib.xform_to_array(_lib.make_frame_xform(c["K"], R, c["P"])) for R in c["Rs"]]), gpu)
    fused = cv.undistort_rotate(big_s[:, :, :sw], c["D"], xf, (dw, dh), dst=big_d[:, :, :dw], borderMode=border, borderValue=bv)
    fused = fused.cpu().numpy()
    if c["dpad"]:
        assert int(big_d[:, :, dw:].to(torch.int64).sum()) == 0, "the kernel wrote into the row padding"
    n_entries = n_diff = 0
    for i in range(n):
        g1, g2 = cv.fisheye_initUndistortRectifyMap(c["K"], c["D"], c["Rs"][i], c["P"], (dw, dh), cv.CV_16SC2, device=gpu)
        two = cv.remap(torch.from_numpy(frames[i]).to(gpu), g1, g2, cv.INTER_LINEAR, dst=torch.from_numpy(prev[i].copy()).to(gpu),
                       borderMode=border, borderValue=bv).cpu().numpy()
        np.testing.assert_array_equal(fused[i], two, err_msg="fused != maps + remap, frame %d of case %d" % (i, seed))
        o1, o2 = orc.fisheye_init_undistort_rectify_map(c["K"], c["D"], c["Rs"][i], c["P"], (dw, dh), orc.CV_16SC2)
        g1, g2 = g1.cpu().numpy(), g2.cpu().numpy()
        gq = (g1.astype(np.int64) << 5) + np.stack([g2 & 31, g2 >> 5], -1)
        oq = (o1.astype(np.int64) << 5) + np.stack([o2 & 31, o2 >> 5], -1)
        same = (gq == oq).all(-1)
        # Where they differ it is by one step -- except entries OpenCV's own int16 wrap has folded (|coordinate| beyond 2^15 px
        # for rays near 90 degrees off axis), where one step before the wrap is one step after it too.
        assert np.abs(gq - oq).max() <= 1, "case %d: a map entry differs by more than one 1/32-px step" % seed
        n_entries += same.size
        n_diff += int((~same).sum())
        want = orc.remap_linear(frames[i], o1, o2, dst=prev[i].copy(), borderMode=border, borderValue=bv)
        np.testing.assert_array_equal(fused[i][same], want[same], err_msg="pixels differ where the maps agree, case %d" % seed)
    assert n_diff <= max(1, int(2e-4 * n_entries)), "case %d: %d of %d map entries differ" % (seed, n_diff, n_entries)


@pytest.mark.parametrize("mode", ["direct", "rl"])
@pytest.mark.parametrize("seed", [1, 2, 6, 9, 13, 22])
def test_fused_random_case_other_kernel_modes(gpu, orc, seed, mode):
    """The same statements in the two non-default kernel modes (the default, rp, is what test_fused_random_case runs)."""
    from gyroflow_amd import _lib
    kmode = {"direct": _lib.KERNEL_DIRECT, "rl": _lib.KERNEL_RL}[mode]
    _lib.set_kernel_mode(gpu.index or 0, kmode)
    try:
        test_fused_random_case(gpu, orc, seed)
    finally:
        _lib.set_kernel_mode(gpu.index or 0, _lib.KERNEL_DEFAULT)


def _sv_case(seed):
    rng = np.random.default_rng(9000 + seed)
    sw, sh = int(rng.integers(4, 420)), int(rng.integers(4, 150))
    dw = int(rng.choice([int(rng.integers(1, 700)), 255, 256, 257, 512, 513]))
    unit = seed % 2 == 0                                   # unit vertical step: the kernel's fast loop
    dh = int(rng.integers(1, 140)) if not unit else int(rng.integers(max(1, sh - 10), sh + 10))
    # horizontal map: monotone with a varying slope (what NonlinearStretch makes), or arbitrary (seed % 3 == 2)
    if seed % 3 == 2:
        mx = rng.uniform(-6, sw + 6, dw)
    else:
        t = np.linspace(-1, 1, dw) if dw > 1 else np.zeros(1)
        mx = (sw - 1) / 2 + (sw / 2 + rng.uniform(-3, 6)) * (t + rng.uniform(-0.3, 0.3) * t ** 3) / (1 + abs(rng.uniform(-0.3, 0.3)))
    if unit:
        my = np.arange(dh) + rng.integers(-4, 5) + float(rng.choice([0.0, -0.5, 0.25, rng.uniform(0, 1)]))
    else:
        my = np.linspace(rng.uniform(-5, 3), sh + rng.uniform(-3, 5), dh) if seed % 5 else rng.uniform(-6, sh + 6, dh)
    return dict(rng=rng, sw=sw, sh=sh, dw=dw, dh=dh, mx=mx.astype(np.float32), my=np.asarray(my, np.float32),
                border=[0, 1, 5][seed % 3], n=int(rng.integers(1, 4)), spad=int(rng.integers(0, 4)), dpad=int(rng.integers(0, 4)))


@pytest.mark.parametrize("seed", _n(30))
def test_superview_random_case(gpu, orc, seed):
    """gf_superview_cubic_u8c3 (separable maps, batched) == the oracle's INTER_CUBIC remap with the full 2-D maps, bit for bit:
    random sizes across the 256-column / 48-row strip boundaries, unit and other vertical steps, monotone and arbitrary
    column maps, all borders, padded rows."""
    from gyroflow_amd import cv
    c = _sv_case(seed)
    rng, sw, sh, dw, dh, n, border = c["rng"], c["sw"], c["sh"], c["dw"], c["dh"], c["n"], c["border"]
    frames = np.stack([_img(rng, sh, sw, np.uint8) for _ in range(n)])
    prev = np.stack([_img(rng, dh, dw, np.uint8) for _ in range(n)])
    bv = (9, 130, 250)
    big_s = torch.zeros((n, sh, sw + c["spad"], 3), dtype=torch.uint8, device=gpu)
    big_s[:, :, :sw] = torch.from_numpy(frames).to(gpu)
    big_d = torch.zeros((n, dh, dw + c["dpad"], 3), dtype=torch.uint8, device=gpu)
    big_d[:, :, :dw] = torch.from_numpy(prev).to(gpu)
    got = cv.superview_cubic(big_s[:, :, :sw], torch.from_numpy(c["mx"]).to(gpu), torch.from_numpy(c["my"]).to(gpu),
                             dst=big_d[:, :, :dw], borderMode=border, borderValue=bv).cpu().numpy()
    if c["dpad"]:
        assert int(big_d[:, :, dw:].to(torch.int64).sum()) == 0, "the kernel wrote into the row padding"
    m1 = np.ascontiguousarray(np.broadcast_to(c["mx"][None, :], (dh, dw)))
    m2 = np.ascontiguousarray(np.broadcast_to(c["my"][:, None], (dh, dw)))
    for i in range(n):
        want = orc.remap_cubic(frames[i], m1, m2, dst=prev[i].copy(), borderMode=border, borderValue=bv)
        np.testing.assert_array_equal(got[i], want, err_msg="frame %d of case %d" % (i, seed))


@pytest.mark.parametrize("seed", _n(12))
def test_fused_blend_random_case(gpu, seed):
    """gf_undistort_rotate_blend_c3 == the fused remap of every input frame followed by numpy's float64 mean in the reference's
    order and its truncating cast (stabilizer.py:1434-1468): random sizes, group shapes (multiplier >= blended frames), borders."""
    from gyroflow_amd import cv, _lib
    c = _case(100 + seed)
    rng, sw, sh, dw, dh, dtype = c["rng"], c["sw"], c["sh"], c["dw"], c["dh"], c["dtype"]
    border = [0, 1][seed % 2]                          # TRANSPARENT has no batched blend (frame n depends on output n-1)
    nb = int(rng.integers(1, 9 if seed % 4 else 12))   # > 8 takes the kernel without the RP engine
    mult = nb + int(rng.integers(0, 3))
    n_out = int(rng.integers(1, 4))
    n_in = (n_out - 1) * mult + nb
    frames = np.stack([_img(rng, sh, sw, dtype) for _ in range(n_in)])
    big = float(rng.choice([0.5, 5, 30]))
    Rs = [Rotation.from_euler("zyx", rng.uniform(-1, 1, 3) * [big, big, big / 2], degrees=True).as_matrix() for _ in range(n_in)]
    xf = cv.xforms_to_device(np.stack([_lib.xform_to_array(_lib.make_frame_xform(c["K"], R, c["P"])) for R in Rs]), gpu)
    src = torch.from_numpy(frames).to(gpu)
    bv = (7, 200, 33)
    remapped = cv.undistort_rotate(src, c["D"], xf, (dw, dh), borderMode=border, borderValue=bv).cpu().numpy()
    want = []
    for g in range(n_out):
        acc = np.zeros((dh, dw, 3), np.float64)
        for b in range(nb):
            acc += (1 / nb) * remapped[g * mult + b].astype(np.float64)
        want.append(acc.astype(dtype))
    got = cv.undistort_rotate_blend(src, c["D"], xf, (dw, dh), mult, nb, borderMode=border, borderValue=bv).cpu().numpy()
    np.testing.assert_array_equal(got, np.stack(want), err_msg="case %d: n_blend %d, multiplier %d" % (seed, nb, mult))


@pytest.mark.parametrize("seed", _n(10))
def test_fused_superview_random_case(gpu, seed):
    """gf_undistort_rotate_superview_u8c3 (or, outside its envelope, the fallback cv.undistort_rotate_superview takes) == the two
    launches: random source / intermediate / output sizes around the 64-pixel tile grid, random stretch settings."""
    from gyroflow_amd import cv, _lib, NonlinearStretch
    rng = np.random.default_rng(12000 + seed)
    sw, sh = int(rng.integers(80, 420)), int(rng.integers(60, 300))
    mw, mh = int(rng.integers(8, 330)), int(rng.integers(4, 200))
    ow = int(mw * rng.uniform(1.05, 1.6))
    oh = mh if seed % 4 else int(mh * rng.uniform(1.1, 1.3))      # seed % 4 == 0: vertical scale != 1 -> the fallback
    ns = NonlinearStretch((mw, mh), (ow, oh), safe_area=0, expo=float(rng.uniform(1.2, 3.0)))
    ns.set_safe_area(float(rng.choice([0.0, 0.2, 0.6])))
    ns.recompute_maps()
    f = float(rng.uniform(0.5, 1.2)) * max(sw, sh)
    K = np.array([[f, 0, sw / 2], [0, f, sh / 2], [0, 0, 1.0]])
    fo = float(rng.uniform(0.5, 1.3)) * max(mw, mh, 16)
    P = np.array([[fo, 0, mw / 2], [0, fo, mh / 2], [0, 0, 1.0]])
    D = rng.uniform(-1, 1, 4) * np.array([0.2, 0.05, 0.02, 0.01])
    n = int(rng.integers(1, 4))
    big = float(rng.choice([1, 8, 40]))
    Rs = [Rotation.from_euler("zyx", rng.uniform(-1, 1, 3) * [big, big, big / 2], degrees=True).as_matrix() for _ in range(n)]
    xf = cv.xforms_to_device(np.stack([_lib.xform_to_array(_lib.make_frame_xform(K, R, P)) for R in Rs]), gpu)
    src = torch.from_numpy(np.stack([_img(rng, sh, sw, np.uint8) for _ in range(n)])).to(gpu)
    border = [0, 1][seed % 2]
    mid = cv.undistort_rotate(src, D, xf, (mw, mh), borderMode=border, borderValue=(3, 60, 200))
    two = ns.apply_stretch_batch(mid)
    one = cv.undistort_rotate_superview(src, D, xf, (mw, mh), ns.map1[0], ns.map2[:, 0], borderMode=border, borderValue=(3, 60, 200), fused=True)
    # the border mode is the remap step's (stabilizer.py:1442); the stretch step keeps the reference's default, constant 0
    # (nonlinear_stretch.py:130) -- in the one-launch form too
    assert torch.equal(one, two), "case %d: %dx%d -> %dx%d -> %dx%d" % (seed, sw, sh, mw, mh, ow, oh)


@pytest.mark.parametrize("seed", _n(8))
def test_fused_stretched_random_case(gpu, orc, seed):
    """gf_undistort_rotate_stretched_c3 == float32 maps, numpy's float32 rescale, convertMaps, remap (calibrate_video.py:375-380
    followed by stabilizer.py:1442) on the GPU bit for bit, and the rescale + convertMaps step against the oracle."""
    from gyroflow_amd import cv, _lib
    c = _case(200 + seed)
    rng, sw, sh, dw, dh, n, dtype, border = c["rng"], c["sw"], c["sh"], c["dw"], c["dh"], c["n"], c["dtype"], c["border"]
    scale = (float(rng.integers(100, 4000)), float(rng.integers(100, 4000)), float(rng.integers(100, 3000)), float(rng.integers(100, 3000)))
    frames = np.stack([_img(rng, sh, sw, dtype) for _ in range(n)])
    prev = np.stack([_img(rng, dh, dw, dtype) for _ in range(n)])
    xf = cv.xforms_to_device(np.stack([_lib.xform_to_array(_lib.make_frame_xform(c["K"], R, c["P"])) for R in c["Rs"]]), gpu)
    bv = (5, 90, 250)
    got = cv.undistort_rotate(torch.from_numpy(frames).to(gpu), c["D"], xf, (dw, dh), dst=torch.from_numpy(prev.copy()).to(gpu),
                              borderMode=border, borderValue=bv, map_scale=scale).cpu().numpy()
    for i in range(n):
        f1, f2 = cv.fisheye_initUndistortRectifyMap(c["K"], c["D"], c["Rs"][i], c["P"], (dw, dh), cv.CV_32FC1, device=gpu)
        g1, g2 = cv.convertMaps(f1, f2, cv.CV_16SC2, scale_x=scale[:2], scale_y=scale[2:])
        want = cv.remap(torch.from_numpy(frames[i]).to(gpu), g1, g2, cv.INTER_LINEAR, dst=torch.from_numpy(prev[i].copy()).to(gpu),
                        borderMode=border, borderValue=bv).cpu().numpy()
        np.testing.assert_array_equal(got[i], want, err_msg="frame %d of case %d" % (i, seed))
        # the reference's own arithmetic: float32 numpy multiply, then divide (two roundings each), then convertMaps
        n1 = f1.cpu().numpy() * np.float32(scale[0]) / np.float32(scale[1])
        n2 = f2.cpu().numpy() * np.float32(scale[2]) / np.float32(scale[3])
        assert n1.dtype == np.float32
        o1, o2 = orc.convert_maps(n1, n2)
        np.testing.assert_array_equal(g1.cpu().numpy(), o1)
        np.testing.assert_array_equal(g2.cpu().numpy(), o2)
