"""Pinned, multi-buffered frame streaming (SURVEY.md kernel row K7).

The reference moves one frame in and one frame out of its loop per iteration through host memory
(/root/reference/stabilizer.py:1383 cap.read(), :1487 out.write()).  Here a batch of frames is staged in pinned
host memory, uploaded with hipMemcpyAsync on a copy stream, processed on a compute stream and downloaded on a
third stream, with `depth` slots in flight so PCIe transfers overlap the kernels.

Every slot owns all the memory a batch touches on either side of the bus -- frames in, frames out and the batch's
per-frame transforms -- so nothing a queued kernel reads can be recycled before the slot is retired.
"""
import numpy as np
import torch

from . import _lib


def _np_dtype(dt):
    return {torch.uint8: np.uint8, torch.uint16: np.uint16}[dt]


# The decoder-output -> pinned copy of a frame, in row bands on a few threads (numpy releases the GIL while it copies): one
# thread moves a 25 MB 4K frame at ~22 GB/s, i.e. 0.9 k frames/s, half of what the PCIe link behind it takes.
_COPY_THREADS = 0
_copy_pool = None


def _host_copy(dst, src):
    global _copy_pool, _COPY_THREADS
    src = np.asarray(src)
    if dst.nbytes < (4 << 20) or dst.shape != src.shape:
        dst[...] = src
        return
    if _copy_pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        _COPY_THREADS = max(1, min(8, n))
        _copy_pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="gf-stage")
    if _COPY_THREADS == 1:
        dst[...] = src
        return
    rows = dst.shape[0]
    step = -(-rows // _COPY_THREADS)
    futs = [_copy_pool.submit(np.copyto, dst[a:a + step], src[a:a + step]) for a in range(0, rows, step)]
    for f in futs:
        f.result()


class BatchPipeline:
    """depth-slot ring of (pinned in, device in, device out, pinned out, transforms) batch buffers on three HIP streams.

    Per batch:  slot = acquire()  ->  stage(slot, i, frame) for every frame  ->  commit(slot, n, launch, xf).
    stage() accepts a numpy frame (copied into the slot's pinned buffer: the one host copy a decoder's output costs) or a
    pinned torch tensor (uploaded straight from where it lies, e.g. a decoder's own pinned ring); either way the upload of
    frame i is queued at once, so it overlaps the staging of frame i + 1.  commit() uploads the [n,13] transforms `xf` into
    the slot's own device buffer and queues launch(dev_in[:n], dev_out, dev_xf[:n]) -> download.  `launch` runs with the
    compute stream current and returns m, the number of output frames it wrote into dev_out.  Finished batches are handed,
    in submission order, to `deliver(out_frames_numpy_view)`; the view is only valid during the call.
    submit(frames, launch, xf) does the three steps for a list of frames or one pinned [n,H,W,3] tensor."""

    def __init__(self, device, in_shape, out_shape, dtype, batch, deliver, depth=2, out_batch=None):
        self.dev = torch.device(device)
        self.batch = int(batch)
        self.out_batch = int(out_batch if out_batch is not None else batch)
        self.depth = int(depth)
        self.deliver = deliver
        ih, iw = in_shape
        oh, ow = out_shape
        with torch.cuda.device(self.dev):
            self.s_h2d = torch.cuda.Stream(self.dev)
            self.s_comp = torch.cuda.Stream(self.dev)
            self.s_d2h = torch.cuda.Stream(self.dev)
            self.slots = []
            for _ in range(self.depth):
                self.slots.append(dict(
                    hin=torch.empty((self.batch, ih, iw, 3), dtype=dtype).pin_memory(),
                    hout=torch.empty((self.out_batch, oh, ow, 3), dtype=dtype).pin_memory(),
                    din=torch.empty((self.batch, ih, iw, 3), dtype=dtype, device=self.dev),
                    dout=torch.zeros((self.out_batch, oh, ow, 3), dtype=dtype, device=self.dev),
                    hxf=torch.zeros((self.batch, _lib.XFORM_DOUBLES), dtype=torch.float64).pin_memory(),
                    dxf=torch.zeros((self.batch, _lib.XFORM_DOUBLES), dtype=torch.float64, device=self.dev),
                    ev_h2d=torch.cuda.Event(), ev_comp=torch.cuda.Event(), ev_d2h=torch.cuda.Event(),
                    pending=0, busy=False))
            # the allocations above (zero fills included) ran on the current stream: finish them before any side stream
            # touches the buffers
            torch.cuda.current_stream(self.dev).synchronize()
        for s in self.slots:
            s["hin_np"] = s["hin"].numpy()
            s["hout_np"] = s["hout"].numpy()
            s["hxf_np"] = s["hxf"].numpy()
        self.k = 0
        self._trace = None

    # ---- optional timeline of the three streams (tools/render_bench.py --overlap): HIP events around every batch's upload,
    # kernel and download; off by default (timing events cost a little on every record)
    def enable_trace(self):
        with torch.cuda.device(self.dev):
            self._trace = dict(base=torch.cuda.Event(enable_timing=True), rows=[])
            self._trace["base"].record(torch.cuda.current_stream(self.dev))

    def _tev(self, stream):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def trace_table(self):
        """After flush(): one row per batch, times in ms since enable_trace(): h2d/kernel/d2h start and end."""
        if self._trace is None:
            return []
        torch.cuda.synchronize(self.dev)
        base, out = self._trace["base"], []
        for r in self._trace["rows"]:
            if all(k in r for k in ("h0", "h1", "k0", "k1", "d0", "d1")):
                out.append({k: base.elapsed_time(r[k]) for k in ("h0", "h1", "k0", "k1", "d0", "d1")} | {"frames": r["n"]})
        return out

    def _retire(self, slot):
        if slot["busy"]:
            slot["ev_d2h"].synchronize()
            slot["busy"] = False
            if slot["pending"]:
                m, slot["pending"] = slot["pending"], 0
                self.deliver(slot["hout_np"][:m])

    def acquire(self):
        """The next slot of the ring, with its previous batch fully downloaded and delivered."""
        slot = self.slots[self.k % self.depth]
        self.k += 1
        self._retire(slot)
        return slot

    def stage(self, slot, i, frame):
        """Frame i of the slot's batch: host -> pinned (unless it already is) -> device, queued on the upload stream."""
        if i >= self.batch:
            raise ValueError("frame %d does not fit the pipeline's batch of %d" % (i, self.batch))
        if isinstance(frame, torch.Tensor):
            if not frame.is_pinned():
                raise ValueError("a tensor frame must live in pinned host memory")
            hsrc = frame
        else:
            _host_copy(slot["hin_np"][i], frame)     # the decoder-output -> pinned copy
            hsrc = slot["hin"][i]
        with torch.cuda.device(self.dev), torch.cuda.stream(self.s_h2d):
            if self._trace is not None and i == 0:
                slot["trow"] = {"h0": self._tev(self.s_h2d)}
            slot["din"][i].copy_(hsrc, non_blocking=True)

    def commit(self, slot, n, launch, xf=None):
        if n <= 0 or n > self.batch:
            raise ValueError("batch of %d frames does not fit the pipeline's %d" % (n, self.batch))
        with torch.cuda.device(self.dev):
            with torch.cuda.stream(self.s_h2d):
                dxf = None
                if xf is not None:
                    slot["hxf_np"][:n] = xf
                    slot["dxf"][:n].copy_(slot["hxf"][:n], non_blocking=True)
                    dxf = slot["dxf"][:n]
                slot["ev_h2d"].record(self.s_h2d)
                tr = slot.pop("trow", None) if self._trace is not None else None
                if tr is not None:
                    tr["h1"] = self._tev(self.s_h2d); tr["n"] = n
            with torch.cuda.stream(self.s_comp):
                self.s_comp.wait_event(slot["ev_h2d"])
                if tr is not None:
                    tr["k0"] = self._tev(self.s_comp)
                m = int(launch(slot["din"][:n], slot["dout"], dxf))
                slot["ev_comp"].record(self.s_comp)
                if tr is not None:
                    tr["k1"] = self._tev(self.s_comp)
            with torch.cuda.stream(self.s_d2h):
                self.s_d2h.wait_event(slot["ev_comp"])
                if tr is not None:
                    tr["d0"] = self._tev(self.s_d2h)
                if m:
                    slot["hout"][:m].copy_(slot["dout"][:m], non_blocking=True)
                slot["ev_d2h"].record(self.s_d2h)
                if tr is not None:
                    tr["d1"] = self._tev(self.s_d2h)
                    self._trace["rows"].append(tr)
            # the next upload into this slot's din / dxf must not overtake this kernel; the next kernel writing dout
            # must not overtake this download: both are ordered by acquire()'s host wait on ev_d2h.
        slot["pending"] = m
        slot["busy"] = True

    def submit(self, frames, launch, xf=None):
        slot = self.acquire()
        n = len(frames)
        if n == 0 or n > self.batch:
            raise ValueError("batch of %d frames does not fit the pipeline's %d" % (n, self.batch))
        if isinstance(frames, torch.Tensor):
            if not frames.is_pinned():
                raise ValueError("a tensor batch must live in pinned host memory")
            with torch.cuda.device(self.dev), torch.cuda.stream(self.s_h2d):
                if self._trace is not None:
                    slot["trow"] = {"h0": self._tev(self.s_h2d)}
                slot["din"][:n].copy_(frames, non_blocking=True)
        else:
            for i, f in enumerate(frames):
                self.stage(slot, i, f)
        self.commit(slot, n, launch, xf)

    def flush(self):
        for j in range(self.depth):
            self._retire(self.slots[(self.k + j) % self.depth])

    def abort(self):
        """Forget everything in flight: wait for the three streams (uploads into a slot that was never committed included),
        then mark every slot free without delivering its batch.  For a render that stops early; the ring stays usable."""
        with torch.cuda.device(self.dev):
            for st in (self.s_h2d, self.s_comp, self.s_d2h):
                st.synchronize()
        for slot in self.slots:
            slot["busy"] = False
            slot["pending"] = 0
            slot.pop("trow", None)
