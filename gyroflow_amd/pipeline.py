"""Pinned, double-buffered frame streaming (SURVEY.md kernel row K7).

The reference moves one frame in and one frame out of its loop per iteration through host memory
(/root/reference/stabilizer.py:1383 cap.read(), :1487 out.write()).  Here a batch of frames is staged in pinned
host memory, uploaded with hipMemcpyAsync on a copy stream, processed on a compute stream and downloaded on a
third stream, with `depth` slots in flight so PCIe transfers overlap the kernels.
"""
import numpy as np
import torch


def _np_dtype(dt):
    return {torch.uint8: np.uint8, torch.uint16: np.uint16}[dt]


class BatchPipeline:
    """depth-slot ring of (pinned in, device in, device out, pinned out) batch buffers on three HIP streams.

    submit(frames, launch) stages `frames` (list of HxWx3 numpy arrays, len <= batch; or an already pinned
    [n,H,W,3] torch tensor, e.g. a decoder's own output ring, which is uploaded without the staging copy) and enqueues
    upload -> launch(dev_in[:n], dev_out[:m]) -> download.  `launch` runs with the compute stream current and returns
    m, the number of output frames it wrote into dev_out.  Finished batches are handed, in submission order, to
    `deliver(out_frames_numpy_view)`; the view is only valid during the call."""

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
                    ev_h2d=torch.cuda.Event(), ev_comp=torch.cuda.Event(), ev_d2h=torch.cuda.Event(),
                    pending=0, busy=False))
        for s in self.slots:
            s["hin_np"] = s["hin"].numpy()
            s["hout_np"] = s["hout"].numpy()
        self.k = 0

    def _retire(self, slot):
        if slot["busy"]:
            slot["ev_d2h"].synchronize()
            slot["busy"] = False
            if slot["pending"]:
                self.deliver(slot["hout_np"][:slot["pending"]])
                slot["pending"] = 0

    def submit(self, frames, launch):
        slot = self.slots[self.k % self.depth]
        self.k += 1
        self._retire(slot)                       # the slot's previous batch must be fully downloaded and delivered
        n = len(frames)
        if n == 0 or n > self.batch:
            raise ValueError("batch of %d frames does not fit the pipeline's %d" % (n, self.batch))
        if isinstance(frames, torch.Tensor):
            if not frames.is_pinned():
                raise ValueError("a tensor batch must live in pinned host memory")
            hsrc = frames
        else:
            for i, f in enumerate(frames):
                slot["hin_np"][i] = f           # host copy into pinned memory (stands in for the decoder's output)
            hsrc = slot["hin"][:n]
        with torch.cuda.device(self.dev):
            with torch.cuda.stream(self.s_h2d):
                slot["din"][:n].copy_(hsrc, non_blocking=True)
                slot["ev_h2d"].record(self.s_h2d)
            with torch.cuda.stream(self.s_comp):
                self.s_comp.wait_event(slot["ev_h2d"])
                m = int(launch(slot["din"][:n], slot["dout"]))
                slot["ev_comp"].record(self.s_comp)
            with torch.cuda.stream(self.s_d2h):
                self.s_d2h.wait_event(slot["ev_comp"])
                if m:
                    slot["hout"][:m].copy_(slot["dout"][:m], non_blocking=True)
                slot["ev_d2h"].record(self.s_d2h)
            # the next upload into this slot's din must not overtake this kernel; the next kernel writing dout
            # must not overtake this download: both are ordered by _retire()'s host wait on ev_d2h.
        slot["pending"] = m
        slot["busy"] = True

    def flush(self):
        for j in range(self.depth):
            self._retire(self.slots[(self.k + j) % self.depth])
