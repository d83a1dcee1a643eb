"""Whole-node render: one worker process per GPU, each rendering its contiguous shard of the clip into its own byte range of
ONE raw output file.

The reference renders a clip in one loop and writes one stream (/root/reference/stabilizer.py:1378-1522, out.write at :1487).
Frames are independent given their own (quaternion, fcorr, focal centre) (:1423-1443), so the loop's iterations split into
contiguous ranges (frame_shard, aligned to the hyperlapse multiplier so that no blend group straddles two workers) with no
exchange step between the workers -- no collective, xGMI idle.  Output frames have a fixed size, so every worker knows the byte
offset of its first frame in the common file: worker r writes frames [first_r, first_r + count_r) and the file reads exactly
like the single-GPU render.  bg_color="HISTORY" makes frame n depend on output n - 1 and does not shard.
"""
import multiprocessing

import numpy as np

from .stabilizer import Stabilizer, frame_shard


def shard_layout(starttime, stoptime, fps, world, mult=1, nblend=1):
    """[(lo, hi, first_out, n_out)] per rank: the loop iterations [lo, hi) rank r runs (k = i - 1 of the reference's loop
    counter) and the output frames it writes when every read succeeds -- one per iteration with k % mult == nblend - 1
    (:1464).  Pure host arithmetic."""
    num_frames = int((stoptime - starttime) * fps)
    n_iter = max(num_frames - 1, 0)
    nblend = min(mult, nblend)
    out, first = [], 0
    for r in range(world):
        lo, hi = frame_shard(n_iter, r, world, mult)
        emit_phase = nblend - 1
        n_out = 0 if hi <= lo else (hi - 1 - emit_phase) // mult - (lo - 1 - emit_phase) // mult
        out.append((lo, hi, first, n_out))
        first += n_out
    return out


class RawVideoSlotSink:
    """write(frame)/close() into a byte range of an existing raw-video file: frame j of this sink is frame first_frame + j of
    the file."""

    def __init__(self, path, first_frame, frame_bytes, max_frames):
        self._f = open(path, "r+b")
        self._f.seek(int(first_frame) * int(frame_bytes))
        self.frame_bytes, self.max_frames, self.count = int(frame_bytes), int(max_frames), 0

    def write(self, frame):
        buf = np.ascontiguousarray(frame)
        if buf.nbytes != self.frame_bytes or self.count >= self.max_frames:
            raise IOError("frame %d does not fit this worker's slot of the output file" % self.count)
        self._f.write(buf.tobytes())
        self.count += 1

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None


def _worker(rank, world, spec, queue):
    """One GPU's share: a fresh process (spawned before it has touched any device)."""
    written, err = 0, None
    try:
        import torch
        from . import dist as gdist
        n_dev = torch.cuda.device_count()
        if n_dev == 0:
            raise RuntimeError("no GPU visible to worker %d" % rank)
        dev_index = rank % n_dev
        gdist.pin_to_gpu(dev_index)                      # this worker's pinned ring and copy threads next to its GPU
        torch.cuda.set_device(dev_index)
        st = Stabilizer(spec["source"], spec["calibration"], spec["stab_transform"], spec["fps"], spec["width"], spec["height"],
                        device=torch.device("cuda", dev_index), video_rotation=spec["video_rotation"])
        st.set_hyperlapse(*spec["hyperlapse"])
        if spec["adaptive_zoom"] is not None:
            st.adaptive_zoom = spec["adaptive_zoom"]
        lo, hi, first, n_out = spec["layout"][rank]
        sink = RawVideoSlotSink(spec["outpath"], first, spec["frame_bytes"], n_out)
        written = st.renderfile(spec["starttime"], spec["stoptime"], sink, rank=rank, world=world, **spec["render_kw"]) or 0
    except BaseException as e:              # report, never hang the parent
        err = "%s: %s" % (type(e).__name__, e)
    queue.put((rank, int(written), err))


def render_sharded(n_gpus, source, calibration, stab_transform, fps, width, height, starttime, stoptime, outpath,
                   hyperlapse=(1, 1), video_rotation=-1, adaptive_zoom=None, frame_dtype=np.uint8, oversubscribe=False,
                   timeout=None, **render_kw):
    """Render [starttime, stoptime) on n_gpus GPUs of this node into the raw-video file `outpath` (bgr24 / bgr48le frames of
    out_size, in clip order: what Stabilizer.renderfile(outpath=...) writes on one GPU, byte for byte).

    source, calibration, stab_transform, fps, width, height, video_rotation: as for Stabilizer(); they are pickled to the
    workers, so `source` must survive that (RawVideoSource re-opens its file; ArraySource carries its frames).  hyperlapse =
    (multiplier, blended frames).  render_kw: renderfile's keywords (out_size, fov_scale, smoothingFocus, bg_color, batch, ...).
    n_gpus > visible GPUs is refused unless oversubscribe=True (workers then share devices: a rehearsal, not a speed-up).
    Returns the number of frames in the file.  Like renderfile, problems are printed, not raised: the return value is None
    when nothing was rendered."""
    import torch
    n_gpus = int(n_gpus)
    if n_gpus < 1:
        print("render_sharded: n_gpus must be >= 1")
        return
    n_dev = torch.cuda.device_count()            # does not initialise a device
    if n_dev == 0 or (n_gpus > n_dev and not oversubscribe):
        print("render_sharded: %d GPUs asked for, %d visible" % (n_gpus, n_dev))
        return
    if render_kw.get("bg_color") == "HISTORY":
        print('bg_color="HISTORY" makes every frame depend on the previous output: it does not shard')
        return
    for k in ("rank", "world", "outpath"):
        if k in render_kw:
            print("render_sharded: %s is set per worker, not by the caller" % k)
            return
    if not isinstance(outpath, str) or outpath.split(".")[-1].lower() not in Stabilizer.VALID_EXTENSIONS:
        print(f"{outpath} does not have a valid file extension")
        return
    out_size = render_kw.get("out_size", (1920, 1080))
    frame_bytes = int(out_size[0]) * int(out_size[1]) * 3 * np.dtype(frame_dtype).itemsize
    mult = int(hyperlapse[0]); nblend = min(mult, int(hyperlapse[1]))
    layout = shard_layout(starttime, stoptime, fps, n_gpus, mult, nblend)
    total = sum(n for _, _, _, n in layout)
    with open(outpath, "wb") as f:               # the whole file, so that every worker can seek into it
        f.truncate(total * frame_bytes)
    spec = dict(source=source, calibration=calibration, stab_transform=np.asarray(stab_transform, np.float64), fps=fps,
                width=width, height=height, video_rotation=video_rotation, hyperlapse=(mult, nblend), adaptive_zoom=adaptive_zoom,
                layout=layout, outpath=outpath, frame_bytes=frame_bytes, starttime=starttime, stoptime=stoptime, render_kw=render_kw)
    ctx = multiprocessing.get_context("spawn")   # fresh interpreters: nothing of this process's GPU state is inherited
    queue = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, n_gpus, spec, queue)) for r in range(n_gpus)]
    for p in procs:
        p.start()
    results = {}
    try:
        import queue as _q
        while len(results) < n_gpus:
            try:
                r, n, err = queue.get(timeout=1.0)
                results[r] = (n, err)
            except _q.Empty:
                dead = [i for i, p in enumerate(procs) if p.exitcode not in (None, 0) and i not in results]
                if dead:
                    for i in dead:
                        results[i] = (0, "worker exited with code %s" % procs[i].exitcode)
                if timeout is not None:
                    timeout -= 1.0
                    if timeout <= 0:
                        break
    finally:
        for p in procs:
            p.join(5 if len(results) == n_gpus else 0.1)
            if p.is_alive():
                p.terminate()
                p.join()
    written = [results.get(r, (0, "no result"))[0] for r in range(n_gpus)]
    failed = False
    for r in range(n_gpus):
        err = results.get(r, (0, "no result"))[1]
        if err:
            print("render_sharded: worker %d failed: %s" % (r, err))
            failed = True
    if failed:          # how far a failed worker got is unknown: the file is left as the workers wrote it, and not vouched for
        return
    close_holes(outpath, layout, written, frame_bytes)
    return sum(written)


def close_holes(outpath, layout, written, frame_bytes):
    """A worker that wrote fewer frames than its slot holds (stabilisation data ran out, reads failed) leaves a hole in the
    common file: move the later workers' frames down so that the file holds the frames that exist, in order -- what the
    single loop would have written -- and cut it there.  Frames only ever move towards the start, one at a time, so the
    copy is safe in place."""
    if list(written) == [n for _, _, _, n in layout]:
        return
    pos = 0
    with open(outpath, "r+b") as f:
        for (lo, hi, first, n_out), w in zip(layout, written):
            if w and first != pos:
                for j in range(w):
                    f.seek((first + j) * frame_bytes)
                    buf = f.read(frame_bytes)
                    f.seek((pos + j) * frame_bytes)
                    f.write(buf)
            pos += w
        f.truncate(pos * frame_bytes)


def _render_sharded_method(self, n_gpus, starttime, stoptime, outpath, frame_dtype=np.uint8, oversubscribe=False, **render_kw):
    """Stabilizer.render_sharded: this Stabilizer's clip, calibration, transforms, hyperlapse and rotation settings, on n_gpus GPUs."""
    return render_sharded(n_gpus, self.cap, self.undistort, self.stab_transform, self.fps, self.width, self.height, starttime,
                          stoptime, outpath, hyperlapse=(self.hyperlapse_multiplier, self.hyperlapse_num_blended_frames),
                          video_rotation=getattr(self, "video_rotate_code", -1), adaptive_zoom=self.adaptive_zoom,
                          frame_dtype=frame_dtype, oversubscribe=oversubscribe, **render_kw)


Stabilizer.render_sharded = _render_sharded_method
