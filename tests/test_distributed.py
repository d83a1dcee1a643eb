"""N>1 path on CPU: world_size-2 gloo processes shard the render loop's iterations with frame_shard(); the union of
the ranks' schedules equals the single-rank schedule (no exchange step), and the timing helpers reduce correctly."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT, preset_path


def _mk_stab(mult, nblend, n=47, fail_at=()):
    from gyroflow_amd import Stabilizer, ArraySource
    frames = [np.full((2, 2, 3), i, np.uint8) for i in range(n)]
    quats = np.tile(np.array([1.0, 0, 0, 0]), (n - 4, 1))
    st = Stabilizer(ArraySource(frames, fail_at=fail_at), preset_path("hero8_27k_43"), quats, 10.0, 2, 2)
    st.set_hyperlapse(mult, nblend)
    return st


def _schedule(st, start, stop, rank, world):
    from gyroflow_amd.stabilizer import frame_shard
    num_frames = int((stop - start) * st.fps)
    tstart = int(start * st.fps)
    lo, hi = frame_shard(max(num_frames - 1, 0), rank, world, st.hyperlapse_multiplier)
    return [(fn, int(fr[0, 0, 0]), ph, rm, em) for fn, fr, ph, rm, em in st.schedule(tstart + lo, hi - lo, first_iter=lo)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gyroflow_amd import dist as gdist
    r, lr, w = gdist.init_from_env()
    assert (r, w) == (rank, world)
    out = {}
    for mult, nblend in [(1, 1), (4, 2), (3, 3)]:
        out[(mult, nblend)] = _schedule(_mk_stab(mult, nblend), 0.3, 4.4, rank, world)
    gdist.barrier()
    mx = gdist.max_over_ranks(1.0 + rank)
    sm = gdist.sum_over_ranks(10.0 * (rank + 1))
    q.put((rank, out, mx, sm))
    gdist.barrier()
    gdist.shutdown()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[2] for r in res] == [2.0, 2.0] and [r[3] for r in res] == [30.0, 30.0]
    for key in res[0][1]:
        single = _schedule(_mk_stab(*key), 0.3, 4.4, 0, 1)
        assert res[0][1][key] + res[1][1][key] == single
        assert len(res[0][1][key]) > 0 and len(res[1][1][key]) > 0
        if key[0] > 1:
            assert res[1][1][key][0][2] == 0          # rank 1 starts on a hyperlapse group boundary


def test_schedule_matches_reference_indexing():
    """Frame-index quirks of stabilizer.py:1378-1468: loop starts at i=1, num_frames-1 iterations, frame_num is the
    position before the read, emission on the last blended frame of a group, failed reads and stab exhaustion."""
    st = _mk_stab(4, 2)
    s = _schedule(st, 0.3, 2.0, 0, 1)                     # tstart = 3, num_frames = 17 -> 16 iterations
    assert [e[0] for e in s] == list(range(3, 19))
    assert [e[2] for e in s[:5]] == [0, 1, 2, 3, 0]
    assert [e[0] for e in s if e[3]] == [3, 4, 7, 8, 11, 12, 15, 16]       # |0+1|4+5|... relative to the start
    assert [e[0] for e in s if e[4]] == [4, 8, 12, 16]
    st = _mk_stab(1, 1, fail_at=(5, 6))
    s = _schedule(st, 0.0, 1.0, 0, 1)
    assert [e[0] for e in s] == [0, 1, 2, 3, 4, 7, 8]     # 9 iterations, two failed reads skipped
    st = _mk_stab(1, 1, fail_at=tuple(range(10, 20)))
    assert [e[0] for e in _schedule(st, 0.0, 4.0, 0, 1)][-1] == 9        # sixth consecutive failure stops the render
    st = _mk_stab(1, 1)
    assert [e[0] for e in _schedule(st, 3.0, 4.7, 0, 1)][-1] == 42       # stab_transform has 43 entries


def test_shard_layout_matches_the_schedule():
    """render_sharded's file layout: every rank's first output frame and frame count, from arithmetic alone, equal what the
    ranks' schedules emit -- for plain renders, frame skipping and hyperlapse blends, 1 to 8 ranks."""
    from gyroflow_amd import shard_layout
    for mult, nblend in [(1, 1), (4, 2), (3, 3), (5, 1), (2, 7)]:
        for world in (1, 2, 3, 4, 8):
            st = _mk_stab(mult, nblend, n=200)
            lay = shard_layout(0.3, 17.9, st.fps, world, mult, nblend)
            first = 0
            for rank, (lo, hi, f0, n_out) in enumerate(lay):
                emits = [e for e in _schedule(st, 0.3, 17.9, rank, world) if e[4]]
                assert (f0, n_out) == (first, len(emits)), (mult, nblend, world, rank)
                first += n_out
            single = [e for e in _schedule(_mk_stab(mult, nblend, n=200), 0.3, 17.9, 0, 1) if e[4]]
            assert first == len(single)


def test_raw_video_source_survives_pickling(tmp_path):
    import pickle
    from gyroflow_amd import RawVideoSource
    p = str(tmp_path / "clip.raw")
    np.arange(5 * 4 * 6 * 3, dtype=np.uint8).tofile(p)
    s = RawVideoSource(p, 6, 4)
    s.set_pos(3)
    t = pickle.loads(pickle.dumps(s))
    assert (t.count, t.get_pos()) == (5, 3)
    np.testing.assert_array_equal(t.read()[1], s.read()[1])


def test_close_holes_compacts_the_common_file(tmp_path):
    """render_sharded's clean-up when workers wrote fewer frames than their slots hold: later shards move down, in order."""
    from gyroflow_amd.sharded import close_holes, RawVideoSlotSink
    fb = 7
    layout = [(0, 4, 0, 4), (4, 8, 4, 4), (8, 12, 8, 4)]
    path = str(tmp_path / "out.yuv")
    frame = lambda r, j: np.full(fb, 10 * r + j, np.uint8)
    for written in ([4, 4, 4], [4, 2, 0], [2, 0, 3], [0, 4, 1], [0, 0, 0]):
        with open(path, "wb") as f:
            f.truncate(12 * fb)
        for r, (lo, hi, first, n_out) in enumerate(layout):
            sink = RawVideoSlotSink(path, first, fb, n_out)
            for j in range(written[r]):
                sink.write(frame(r, j))
            sink.close()
        close_holes(path, layout, written, fb)
        want = np.concatenate([frame(r, j) for r in range(3) for j in range(written[r])] or [np.zeros(0, np.uint8)])
        np.testing.assert_array_equal(np.fromfile(path, np.uint8), want)
    sink = RawVideoSlotSink(path, 0, fb, 1)
    sink.write(frame(0, 0))
    import pytest
    with pytest.raises(IOError):
        sink.write(frame(0, 1))                      # past the slot
    with pytest.raises(IOError):
        RawVideoSlotSink(path, 0, fb, 4).write(np.zeros(fb + 1, np.uint8))      # not a frame of this file
