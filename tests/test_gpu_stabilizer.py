"""Render-loop parity: gyroflow_amd.Stabilizer.renderfile against the reference loop's order of operations
(/root/reference/stabilizer.py:1378-1522) restated below on top of the CPU oracle.  Small frames, every mode."""
import numpy as np
import pytest
import torch

from conftest import preset_path

pytestmark = pytest.mark.gpu


def reference_loop(orc, cal, frames, fail_at, stab, fps, start, stop, out_size, fov_scale, bg_color, mult, nblend,
                   orig_dim):
    """The body of Stabilizer.renderfile (reference), with cv2 replaced by the oracle and fcorr == 1
    (smoothingFocus = -2, adaptive_zoom.py:152-153).  Returns the list of written frames."""
    from gyroflow_amd import Stabilizer
    borderMode, borderValue = Stabilizer.border_from_bg_color(bg_color)
    num_frames = int((stop - start) * fps)
    pos = int(start * fps)
    out = []
    frame_out = None
    frame_temp = np.zeros((out_size[1], out_size[0], 3), np.float64)
    nblend = min(mult, nblend)
    num_not_success = 0
    for i in range(1, num_frames):
        frame_num = pos
        success = pos < len(frames) and pos not in fail_at
        frame = frames[pos] if success else None
        if pos < len(frames):
            pos += 1
        if success:
            num_not_success = 0
        elif num_not_success >= 5:
            break
        else:
            num_not_success += 1
        if frame_num >= len(stab):
            break
        if not success:
            continue
        K, D, R, P, od = cal._frame_matrices(fov_scale * 1.0, out_size, orig_dim, stab[frame_num],
                                             np.array([cal.calib_dimension[0] / 2, cal.calib_dimension[1] / 2]), True)
        m1, m2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2)
        if (i - 1) % mult == 0 and nblend > 1:
            frame_temp = frame_temp * 0.0
        if (i - 1) % mult < nblend:
            frame_out = orc.remap_linear(frame, m1, m2, dst=frame_out, borderMode=borderMode, borderValue=borderValue)
            if nblend > 1:
                frame_temp += 1 / nblend * frame_out.astype(np.float64)
        if ((i - 1) - nblend + 1) % mult == 0:
            if nblend > 1:
                frame_out = frame_temp.astype(frame.dtype)
            out.append(frame_out.copy())
    return out


def _quats(n, seed=0):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    e = np.cumsum(rng.normal(0, 0.8, (n, 3)), axis=0)
    q = Rotation.from_euler("zyx", e, degrees=True).as_quat()
    return q[:, [3, 0, 1, 2]]


@pytest.mark.parametrize("mode", [
    dict(bg="#000000", mult=1, nblend=1),
    dict(bg="#336699", mult=1, nblend=1, batch=5),
    dict(bg="REPLICATE", mult=3, nblend=1),
    dict(bg="REPLICATE", mult=4, nblend=2, batch=6),
    dict(bg="#102030", mult=4, nblend=4, dtype=np.uint16),
    # bg_color="HISTORY" (BORDER_TRANSPARENT over the previous output, :1240-1241,1442): batched, the frames of a batch chained
    # on the GPU; batch boundaries inside and between hyperlapse groups, failed reads, 16-bit frames; and the per-frame form
    dict(bg="HISTORY", mult=1, nblend=1),
    dict(bg="HISTORY", mult=3, nblend=2),
    dict(bg="HISTORY", mult=1, nblend=1, batch=4, depth=3),
    dict(bg="HISTORY", mult=4, nblend=3, fail_at=(6, 11, 16, 17), batch=5),
    dict(bg="HISTORY", mult=2, nblend=2, dtype=np.uint16, batch=3),
    dict(bg="HISTORY", mult=2, nblend=1, pinned=True, batch=7),
    dict(bg="HISTORY", mult=3, nblend=2, sequential=True),
    dict(bg="#000000", mult=1, nblend=1, fail_at=(4, 9, 10)),
    dict(bg="#000000", mult=2, nblend=2, sequential=True, fail_at=(5,)),
    dict(bg="#000000", mult=1, nblend=1, short_stab=True),
    # failed reads inside hyperlapse blend groups on the batched path: the group falls back to the reference's sequence
    # (first frame of a group missing -> no reset, the previous group's accumulator carries over; middle / last frame missing)
    dict(bg="#000000", mult=2, nblend=2, fail_at=(5,), batch=4),
    dict(bg="#203040", mult=4, nblend=3, fail_at=(6, 11, 16, 17), batch=6),
    dict(bg="REPLICATE", mult=3, nblend=3, fail_at=(2, 3, 4, 13), batch=9),
    dict(bg="#000000", mult=3, nblend=2, fail_at=(8, 9, 10, 11, 12), pinned=True),
    dict(bg="#000000", mult=1, nblend=1, pinned=True, batch=7),
])
def test_renderfile_matches_reference_loop(gpu, orc, mode):
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, synthetic_frames
    dtype = mode.get("dtype", np.uint8)
    w, h = 338, 254                         # 2704x2028 / 8
    n = 30
    frames = synthetic_frames(w, h, n=n, dtype=dtype, seed=7)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    stab = _quats(21 if mode.get("short_stab") else n + 5, seed=3)
    fail_at = mode.get("fail_at", ())
    fps, start, stop = 10.0, 0.2, 2.9
    out_size = (240, 136)
    source = ArraySource(frames, fail_at=fail_at)
    if mode.get("pinned"):                  # frames arrive as pinned tensors (a decoder's own pinned ring): no staging copy
        from gyroflow_amd import PinnedRingSource

        class FailingPinned(PinnedRingSource):
            def read(self):
                p = self.pos
                ok, f = super().read()
                return (False, None) if p in fail_at else (ok, f)
        source = FailingPinned(frames, len(frames))
    st = Stabilizer(source, cal, stab, fps, w, h, device=gpu)
    st.set_hyperlapse(mode["mult"], mode["nblend"])
    sink = ListSink()
    nw = st.renderfile(start, stop, sink, out_size=out_size, fov_scale=1.3, smoothingFocus=-2, bg_color=mode["bg"],
                       batch=mode.get("batch", 8), sequential=mode.get("sequential", False), pipeline_depth=mode.get("depth", 2))
    want = reference_loop(orc, cal, frames, set(fail_at), stab, fps, start, stop, out_size, 1.3, mode["bg"], mode["mult"],
                          mode["nblend"], (w, h))
    assert nw == len(sink.frames) == len(want) and nw > 0
    for a, b in zip(sink.frames, want):
        np.testing.assert_array_equal(a, b)


def test_map_function_preview(gpu, orc):
    """map_function (stabilizer.py:1205-1215) + remap, the preview path of gyroflow.py:281-283."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, synthetic_frames, cv
    w, h = 480, 270
    frames = synthetic_frames(w, h, n=2)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero7_4k_169"))
    stab = _quats(4, seed=1)
    st = Stabilizer(ArraySource(frames), cal, stab, 30.0, w, h, device=gpu)
    m1, m2 = st.map_function(9, out_size=(320, 180))           # clamps to the last frame
    K, D, R, P, od = cal._frame_matrices(0.9, (320, 180), (w, h), stab[3], None, True)
    o1, o2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2)
    np.testing.assert_array_equal(m1.cpu().numpy(), o1)
    np.testing.assert_array_equal(m2.cpu().numpy(), o2)
    got = cv.remap(frames[1], m1, m2, cv.INTER_LINEAR)
    np.testing.assert_array_equal(got, orc.remap_linear(frames[1], o1, o2))


def test_only_undistort_renderfile(gpu, orc, tmp_path):
    """OnlyUndistort (stabilizer.py:1718-1844): static maps, BORDER_CONSTANT remap, centre crop; frames come from and go
    to raw-video files through the frameio adapters."""
    from gyroflow_amd import OnlyUndistort, FisheyeCalibrator, RawVideoSource, RawVideoSink, synthetic_frames
    w, h = 338, 254
    frames = synthetic_frames(w, h, n=14, seed=21)
    src_path, dst_path = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    s = RawVideoSink(src_path)
    for f in frames:
        s.write(f)
    s.close()
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    cal.calib_dimension = (w, h); cal.orig_dimension = (w, h)      # a preset for this frame size: maps come out w x h
    cal.K = cal.K.copy(); cal.K[:2] *= w / 2704
    ou = OnlyUndistort(RawVideoSource(src_path, w, h), cal, w, h, fps=10.0, fov_scale=1.5, device=gpu)
    out_size = (240, 136)
    n = ou.renderfile(0.3, 1.2, RawVideoSink(dst_path), out_size=out_size, batch=4)
    K, D, R, P, od = cal._frame_matrices(1.5, None, (w, h), None, None, True)
    assert od == (w, h)
    m1, m2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2)
    np.testing.assert_array_equal(ou.map1.cpu().numpy(), m1)
    np.testing.assert_array_equal(ou.map2.cpu().numpy(), m2)
    num_frames = int((1.2 - 0.3) * 10.0)
    assert n == num_frames
    got = RawVideoSource(dst_path, out_size[0], out_size[1])
    assert got.count == num_frames
    crop = (int((w - out_size[0]) / 2), int((h - out_size[1]) / 2))
    for k in range(num_frames):
        want = orc.remap_linear(frames[3 + k], m1, m2)[crop[1]:crop[1] + out_size[1], crop[0]:crop[0] + out_size[0]]
        ok, f = got.read()
        assert ok
        np.testing.assert_array_equal(f, want)
    with pytest.raises(NotImplementedError):
        ou.renderfile(0, 1, RawVideoSink(dst_path), out_size=out_size, scale=2)


def test_gyroflow_data_file_roundtrip(gpu, tmp_path):
    """import_gyroflow_file -> update_smoothing -> export_gyroflow_file (stabilizer.py:256-267,1574-1712) on a data file
    rebuilt from the fixture of plugins/blender/GX016015.MP4.gyroflow: the exported per-frame rows reproduce the ones the
    reference wrote (quaternions and fcorr to the file's 5 decimals, given that raw_imu is itself rounded)."""
    import json
    import os
    from conftest import GOLDEN
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, synthetic_frames
    z = np.load(os.path.join(GOLDEN, "quats_gyroflow_sample.npz"))
    cal0 = FisheyeCalibrator()
    cal0.load_calibration_json(preset_path("hero9_4k_43"))
    caldata = cal0.get_calibration_data()
    caldata["fisheye_params"]["camera_matrix"] = z["K"].tolist()
    caldata["fisheye_params"]["distortion_coeffs"] = z["D"].tolist()
    src_file = str(tmp_path / "in.gyroflow")
    with open(src_file, "w") as f:
        json.dump({"title": "Gyroflow data file", "calibration_data": caldata, "video_rotate_code": -1, "sensor_DOF": 6,
                   "raw_imu": z["raw_imu"].tolist()}, f)
    fps = float(z["fps"])
    st = Stabilizer(ArraySource(synthetic_frames(64, 48, n=2)), cal0, None, fps, 2704, 2028, device=gpu)
    assert not st.update_smoothing()
    assert st.import_gyroflow_file(src_file)
    assert np.array_equal(st.undistort.K, z["K"]) and st.acc_data.shape == (len(z["raw_imu"]), 4)
    assert st.update_smoothing()
    ref = z["stab_transform"]
    assert st.stab_transform.shape == (1622, 4) and np.abs(st.stab_transform - ref[:, 3:7]).max() < 2e-5
    out_file = str(tmp_path / "out.gyroflow")
    assert st.export_gyroflow_file(out_file, out_size=(1920, 1080), smoothingFocus=2.0, videofile="GX016015.MP4")
    with open(out_file) as f:
        g = json.load(f)
    rows = np.array(g["stab_transform"])
    assert rows.shape == ref.shape
    np.testing.assert_array_equal(rows[:, 0], ref[:, 0])
    assert np.abs(rows[:, 1] - ref[:, 1]).max() <= 1e-5
    assert np.abs(rows[:, 2] - ref[:, 2]).max() < 2e-4 and abs(g["min_fcorr"] - float(z["min_fcorr"])) < 1e-4
    assert np.abs(rows[:, 3:7] - ref[:, 3:7]).max() < 3e-5
    assert g["stab_summary"] == "Plain 3D smoothing,smoothness:0.2" and g["sensor_DOF"] == 6
    assert np.abs(np.array(g["frame_orientation"])[:, 2:] - z["frame_orientation"][:, 2:]).max() < 2e-5
    # the exported file loads again
    st2 = Stabilizer(ArraySource(synthetic_frames(64, 48, n=2)), cal0, None, fps, 2704, 2028, device=gpu)
    assert st2.import_gyroflow_file(out_file) and st2.update_smoothing()
    assert np.abs(st2.stab_transform - st.stab_transform).max() < 2e-5
    with open(str(tmp_path / "bad.gyroflow"), "w") as f:
        json.dump({"title": "x"}, f)
    assert st2.import_gyroflow_file(str(tmp_path / "bad.gyroflow")) is False


def test_renderfile_reference_call_surface(gpu, orc, tmp_path):
    """The reference's own call (gyroflow.py:2769-2773 -> stabilizer.py:1220-1222): every keyword it passes is accepted; a path
    receives the frames (raw video instead of the ffmpeg pipe); a bad extension, debug_text and a sharded HISTORY render print
    and return like the reference's early exits; a sink whose write fails stops the render without raising."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, RawVideoSource, synthetic_frames
    w, h, n = 338, 254, 20
    frames = synthetic_frames(w, h, n=n, seed=11)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    stab = _quats(n + 2, seed=5)
    st = Stabilizer(ArraySource(frames), cal, stab, 10.0, w, h, device=gpu)
    out_size = (240, 136)
    path = str(tmp_path / "Stabilized.mp4")

    class Viewer:
        map_function_enable = True
        frame = None
        update_once = False
    viewer = Viewer()
    nw = st.renderfile(0.0, 1.6, path, out_size=out_size, split_screen=False, bitrate_mbits=25, display_preview=True, scale=1,
                       vcodec="h264_vaapi", vprofile="high", pix_fmt="yuv420p", debug_text=False, custom_ffmpeg="", smoothingFocus=-2,
                       fov_scale=1.2, bg_color="#102030", audio=False, viewer_thread=viewer)
    sink = ListSink()
    assert st.renderfile(0.0, 1.6, sink, out_size, smoothingFocus=-2, fov_scale=1.2, bg_color="#102030") == nw == 15
    got = RawVideoSource(path, out_size[0], out_size[1])
    assert got.count == nw
    for f in sink.frames:
        np.testing.assert_array_equal(got.read()[1], f)
    assert viewer.update_once and np.array_equal(viewer.frame, sink.frames[-1]) and viewer.map_function_enable is True
    want = reference_loop(orc, cal, frames, set(), stab, 10.0, 0.0, 1.6, out_size, 1.2, "#102030", 1, 1, (w, h))
    for a, b in zip(sink.frames, want):
        np.testing.assert_array_equal(a, b)
    # early exits
    assert st.renderfile(0.0, 1.6, str(tmp_path / "x.txt"), out_size) is None
    assert st.renderfile(0.0, 1.6, ListSink(), out_size, debug_text=True) is None
    assert st.renderfile(0.0, 1.6, ListSink(), out_size, bg_color="HISTORY", world=2) is None

    class BadSink:
        n = 0

        def write(self, f):
            self.n += 1
            if self.n == 4:
                raise IOError("disk full")

        def close(self):
            self.closed = True
    bad = BadSink()
    assert st.renderfile(0.0, 1.6, bad, out_size, smoothingFocus=-2, batch=3) == 3 and bad.closed


def test_batched_render_4k_transforms_stay_with_their_batch(gpu):
    """Three batches of 3840x2160 frames with a different rotation per frame through the pinned pipeline: every output frame
    equals a single-frame call with its own transform.  (Each pipeline slot owns its batch's transforms; they used to live in
    a temporary that the next batch could overwrite before the kernel had run.)"""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, synthetic_frames, cv
    w, h, n = 3840, 2160, 13
    ring = synthetic_frames(w, h, n=2, seed=3)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero7_4k_169"))
    stab = _quats(n + 1, seed=9)
    st = Stabilizer(ArraySource(ring, count=n, ring=True), cal, stab, 30.0, w, h, device=gpu)
    sink = ListSink()
    nw = st.renderfile(0.0, (n + 0.5) / 30.0, sink, out_size=(w, h), smoothingFocus=-2, batch=4, pipeline_depth=2)
    assert nw == n - 1
    for k in range(nw):
        xf = cv.xforms_to_device(cal.frame_xforms(1.0, stab[k:k + 1], output_dim=(w, h), new_img_dim=(w, h)), gpu)
        one = cv.undistort_rotate(torch.from_numpy(ring[k % 2]).to(gpu), cal.D, xf, (w, h)).reshape(h, w, 3).cpu().numpy()
        assert np.array_equal(sink.frames[k], one), "frame %d was rendered with another frame's transform" % k


@pytest.mark.parametrize("mult,nblend", [(1, 1), (4, 2)])
def test_two_rank_render_equals_single_rank(gpu, mult, nblend):
    """renderfile(rank=0, world=2) + renderfile(rank=1, world=2) write the single-rank frames (frames are independent; shards
    are aligned to hyperlapse groups) -- both shards run on the one GPU of this box."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, synthetic_frames
    w, h, n = 338, 254, 40
    frames = synthetic_frames(w, h, n=n, seed=2)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    stab = _quats(n, seed=6)
    outs = []
    for rank, world in ((0, 1), (0, 2), (1, 2)):
        st = Stabilizer(ArraySource(frames), cal, stab, 10.0, w, h, device=gpu)
        st.set_hyperlapse(mult, nblend)
        sink = ListSink()
        st.renderfile(0.1, 3.8, sink, out_size=(240, 136), smoothingFocus=-2, bg_color="REPLICATE", batch=8, rank=rank, world=world)
        outs.append(sink.frames)
    single, r0, r1 = outs
    assert len(r0) > 0 and len(r1) > 0 and len(r0) + len(r1) == len(single)
    for a, b in zip(r0 + r1, single):
        np.testing.assert_array_equal(a, b)


def test_renderfile_stretched_preset_batched_equals_sequential(gpu):
    """A stretched preset renders through the batched fused path; the frames equal the per-frame sequence get_maps ->
    convertMaps -> remap (sequential=True)."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, synthetic_frames
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("runcam5_stretched"))
    w, h, n = 480, 270, 14                      # the 3840x2160 video of the preset, divided by 8
    frames = synthetic_frames(w, h, n=n, seed=13)
    stab = _quats(n + 2, seed=8)
    outs = []
    for seq in (False, True):
        st = Stabilizer(ArraySource(frames), cal, stab, 10.0, w, h, device=gpu)
        sink = ListSink()
        assert st.renderfile(0.0, 1.35, sink, out_size=(320, 180), smoothingFocus=-2, fov_scale=1.1, bg_color="REPLICATE", batch=5, sequential=seq) == 12
        outs.append(sink.frames)
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_aborted_render_does_not_leak_into_the_next(gpu):
    """A sink that fails on its k-th write stops the render (:1486-1491) with batches still in flight in the cached ring; the
    next render on the same Stabilizer must start clean -- same frames as a fresh Stabilizer, not the stale ones first."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, ListSink, synthetic_frames
    w, h, n = 338, 254, 40
    frames = synthetic_frames(w, h, n=n, seed=3)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    stab = _quats(n + 2, seed=9)
    kw = dict(out_size=(240, 136), smoothingFocus=-2, batch=4, pipeline_depth=3)

    class FailingSink(ListSink):
        def write(self, frame):
            if len(self.frames) == 5:
                raise IOError("disk full")
            super().write(frame)

    class InterruptingSource(ArraySource):
        def read(self):
            if self.pos == 17:
                raise KeyboardInterrupt
            return super().read()

    st = Stabilizer(ArraySource(frames), cal, stab, 10.0, w, h, device=gpu)
    bad = FailingSink()
    assert st.renderfile(0.0, 3.5, bad, **kw) == 5 and len(bad.frames) == 5
    ring = st._pipe
    again = ListSink()
    n2 = st.renderfile(1.0, 3.0, again, **kw)                  # another part of the clip, same ring
    assert st._pipe is ring
    fresh = ListSink()
    assert Stabilizer(ArraySource(frames), cal, stab, 10.0, w, h, device=gpu).renderfile(1.0, 3.0, fresh, **kw) == n2 == len(again.frames) > 0
    for a, b in zip(again.frames, fresh.frames):
        np.testing.assert_array_equal(a, b)
    # Ctrl-C out of the source: "terminating render" (:1520-1522), and again nothing stale afterwards
    st2 = Stabilizer(InterruptingSource(frames), cal, stab, 10.0, w, h, device=gpu)
    part = ListSink()
    st2.renderfile(0.0, 3.5, part, **kw)
    assert 0 < len(part.frames) < 17
    st2.cap = ArraySource(frames)
    again2 = ListSink()
    assert st2.renderfile(1.0, 3.0, again2, **kw) == n2
    for a, b in zip(again2.frames, fresh.frames):
        np.testing.assert_array_equal(a, b)


def test_default_ring_is_sized_by_the_clip_and_can_be_released(gpu):
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, NullSink, synthetic_frames
    w, h = 338, 254
    frames = synthetic_frames(w, h, n=8)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    st = Stabilizer(ArraySource(frames, count=100, ring=True), cal, _quats(120, seed=1), 10.0, w, h, device=gpu)
    assert st.default_batch(w * h * 3) == 256
    assert st.renderfile(0.0, 3.1, NullSink(), out_size=(240, 136), smoothingFocus=-2) == 30
    assert st._pipe.batch == 10 and st._pipe.depth == 3          # 30 frames over three slots, not 3 x 256
    small = st._pipe
    assert st.renderfile(0.0, 1.6, NullSink(), out_size=(240, 136), smoothingFocus=-2) == 15 and st._pipe is small     # a ring that is large enough is kept
    assert st.renderfile(0.0, 9.1, NullSink(), out_size=(240, 136), smoothingFocus=-2) == 90 and st._pipe.batch == 30
    st.release_pipeline()
    assert st._pipe is None
    assert st.renderfile(0.0, 1.6, NullSink(), out_size=(240, 136), smoothingFocus=-2) == 15 and st._pipe.batch == 5


@pytest.mark.parametrize("mult,nblend", [(1, 1), (4, 2)])
def test_render_sharded_writes_the_single_gpu_file(gpu, tmp_path, mult, nblend):
    """The whole-node entry: two fresh worker processes (both on cuda:0 here: oversubscribe) render their shards into their byte
    ranges of one raw file; the file equals the single-process render byte for byte.  A clip from a raw file (the workers map
    it themselves) and, for the short-data case, holes closed when a worker runs out of stabilisation data."""
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, RawVideoSource, synthetic_frames, render_sharded
    w, h, n = 338, 254, 45
    src_path = str(tmp_path / "clip.raw")
    np.stack(synthetic_frames(w, h, n=n, seed=21)).tofile(src_path)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(preset_path("hero8_27k_43"))
    stab = _quats(n + 3, seed=4)
    kw = dict(out_size=(240, 136), fov_scale=1.1, smoothingFocus=-2, bg_color="#102030", batch=6, pipeline_depth=2)
    one = str(tmp_path / "one.yuv"); two = str(tmp_path / "two.yuv")
    st = Stabilizer(RawVideoSource(src_path, w, h), cal, stab, 10.0, w, h, device=gpu)
    st.set_hyperlapse(mult, nblend)
    n1 = st.renderfile(0.2, 4.3, one, **kw)
    assert n1 > 0
    n2 = st.render_sharded(2, 0.2, 4.3, two, oversubscribe=True, **kw)
    assert n2 == n1
    a, b = np.fromfile(one, np.uint8), np.fromfile(two, np.uint8)
    assert a.size == n1 * 240 * 136 * 3 and np.array_equal(a, b)
    # stabilisation data that ends inside the second worker's shard: the file holds the frames that exist, in order
    short = stab[:30]
    st2 = Stabilizer(RawVideoSource(src_path, w, h), cal, short, 10.0, w, h, device=gpu)
    st2.set_hyperlapse(mult, nblend)
    m1 = st2.renderfile(0.2, 4.3, one, **kw)
    m2 = render_sharded(2, RawVideoSource(src_path, w, h), cal, short, 10.0, w, h, 0.2, 4.3, two, hyperlapse=(mult, nblend),
                        oversubscribe=True, **kw)
    assert 0 < m1 == m2 < n1 and np.array_equal(np.fromfile(one, np.uint8), np.fromfile(two, np.uint8))
    assert render_sharded(2, RawVideoSource(src_path, w, h), cal, stab, 10.0, w, h, 0.2, 4.3, two, oversubscribe=True, bg_color="HISTORY") is None
