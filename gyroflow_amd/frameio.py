"""Frame I/O adapters either side of the remap: the reference reads frames with
cv2.VideoCapture.read (/root/reference/stabilizer.py:1383) and writes them through vidgear's WriteGear to an ffmpeg pipe
(:1326,1487).  Decode and encode stay on the CPU and out of scope; what the GPU path needs from them is raw frames.
These adapters speak the same protocol as stabilizer.ArraySource / ListSink over headerless raw-video files (what
`ffmpeg -f rawvideo -pix_fmt bgr24|bgr48le` produces and consumes), memory-mapped so that a clip never has to fit in RAM."""
import numpy as np


class RawVideoSource:
    """Headerless BGR frames (uint8 bgr24 or uint16 bgr48le) in a file: read()/get_pos()/set_pos() like cv2.VideoCapture."""

    def __init__(self, path, width, height, dtype=np.uint8):
        self.path = path
        self.width, self.height = int(width), int(height)
        self.dtype = np.dtype(dtype)
        self.frame_bytes = self.width * self.height * 3 * self.dtype.itemsize
        self._open()

    def _open(self):
        path = self.path
        self._mm = np.memmap(path, dtype=self.dtype, mode="r")
        self.count = self._mm.size // (self.width * self.height * 3)
        if self.count == 0:
            raise ValueError("%s holds no complete %dx%d frame" % (path, self.width, self.height))
        self._frames = self._mm[:self.count * self.width * self.height * 3].reshape(self.count, self.height, self.width, 3)
        self.pos = 0

    # pickled as (path, geometry, position): a worker process maps the file itself (render_sharded)
    def __getstate__(self):
        return dict(path=self.path, width=self.width, height=self.height, dtype=self.dtype.str, pos=self.pos)

    def __setstate__(self, st):
        self.path, self.width, self.height, self.dtype = st["path"], st["width"], st["height"], np.dtype(st["dtype"])
        self.frame_bytes = self.width * self.height * 3 * self.dtype.itemsize
        self._open()
        self.pos = st["pos"]

    def get_pos(self):
        return self.pos

    def set_pos(self, n):
        self.pos = int(n)

    def read(self):
        if self.pos >= self.count:
            return False, None
        f = self._frames[self.pos]
        self.pos += 1
        return True, f

    def read_into(self, dst, n):
        """Copy up to n frames straight into `dst` (a numpy view of pinned memory, [>=n,H,W,3]); returns how many."""
        n = min(int(n), self.count - self.pos)
        if n > 0:
            dst[:n] = self._frames[self.pos:self.pos + n]
            self.pos += n
        return max(n, 0)

    def release(self):
        del self._frames
        del self._mm


class PinnedRingSource:
    """`count` frames cycling over a ring of frames held in pinned host memory: read() hands out pinned torch tensors, which
    BatchPipeline uploads from where they lie (no staging copy).  Stands in for a decoder that writes into its own pinned
    output ring; the ring is never rewritten, so a frame stays valid while its upload is in flight."""

    def __init__(self, frames, count):
        import torch
        self._ring = torch.from_numpy(np.ascontiguousarray(np.stack(frames))).pin_memory()
        self.count = int(count)
        self.pos = 0

    def get_pos(self):
        return self.pos

    def set_pos(self, n):
        self.pos = int(n)

    def read(self):
        if self.pos >= self.count:
            return False, None
        f = self._ring[self.pos % self._ring.shape[0]]
        self.pos += 1
        return True, f


class RawVideoSink:
    """Appends frames to a headerless raw-video file: write(frame)/close() like vidgear's WriteGear."""

    def __init__(self, path):
        self._f = open(path, "wb")
        self.count = 0

    def write(self, frame):
        self._f.write(np.ascontiguousarray(frame).tobytes())
        self.count += 1

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None
