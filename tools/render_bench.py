#!/usr/bin/env python3
"""End-to-end rate of Stabilizer.renderfile -- decode stand-in -> pinned -> H2D -> fused kernel -> D2H -> sink -- for a BASELINE
configuration (mode M2 of BASELINE.md: PCIe-inclusive; never bench.py's `value`).

  --source pinned : frames arrive as pinned tensors (PinnedRingSource: a decoder writing into its own pinned ring)
  --source raw    : frames come from a raw-video file through RawVideoSource (memory-mapped; one host copy into the pinned slot)
  --source array  : frames are pageable numpy arrays (ArraySource), copied into the pinned slot

Prints one JSON line."""
import argparse, json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gyroflow_amd import Stabilizer, FisheyeCalibrator, ArraySource, NullSink, PinnedRingSource, RawVideoSource, RawVideoSink, synthetic_frames

CONFIGS = {"c1": ("hero8_27k_43", 1920, 1080, 1920, 1080, np.uint8), "c2": ("hero8_27k_43", 2704, 2028, 1920, 1080, np.uint8),
           "c3": ("hero7_4k_169", 3840, 2160, 3840, 2160, np.uint8), "c5": ("hero9_5k_169", 5312, 2988, 5312, 2988, np.uint16)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=1200)
    ap.add_argument("--batch", type=int, default=0, help="0 = renderfile's default")
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--source", default="pinned", choices=["pinned", "raw", "array"])
    ap.add_argument("--hyperlapse", default="1,1")
    args = ap.parse_args()
    preset, w, h, ow, oh, dt = CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(os.path.join(ROOT, "tests", "golden", "presets", preset + ".json"))
    q = np.load(os.path.join(ROOT, "tests", "golden", "quats_gyroflow_sample.npz"))["stab_transform"][:, 3:7]
    q = np.tile(q, (args.frames // len(q) + 2, 1))[:args.frames + 8]
    ring = synthetic_frames(w, h, n=8, dtype=dt)
    fps = 60.0
    tmp = None
    if args.source == "pinned":
        src = PinnedRingSource(ring, args.frames + 8)
    elif args.source == "array":
        src = ArraySource(ring, count=args.frames + 8, ring=True)
    else:
        tmp = tempfile.NamedTemporaryFile(suffix=".raw", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        sink = RawVideoSink(tmp.name)
        for i in range(min(args.frames + 8, 96)):
            sink.write(ring[i % 8])
        sink.close()
        raw = RawVideoSource(tmp.name, w, h, dt)

        class Loop:          # cycle over the file's frames
            def __init__(self, n): self.pos, self.count = 0, n
            def get_pos(self): return self.pos
            def set_pos(self, p): self.pos = int(p)
            def read(self):
                if self.pos >= self.count: return False, None
                f = raw._frames[self.pos % raw.count]; self.pos += 1
                return True, f
        src = Loop(args.frames + 8)
    st = Stabilizer(src, cal, q, fps, w, h, device=dev)
    mult, nb = (int(v) for v in args.hyperlapse.split(","))
    st.set_hyperlapse(mult, nb)
    kw = dict(out_size=(ow, oh), smoothingFocus=-2, batch=args.batch or None, pipeline_depth=args.depth)
    st.renderfile(0.0, (min(args.frames, 200) + 1.5) / fps, NullSink(), **kw)          # warm-up: allocations, first-touch, lens table
    torch.cuda.synchronize()
    sink = NullSink()
    t0 = time.perf_counter()
    n = st.renderfile(0.0, (args.frames + 1.5) / fps, sink, **kw)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    n_in = args.frames
    esz = np.dtype(dt).itemsize
    print(json.dumps({"tool": "render_bench", "config": args.config, "source": args.source, "frames_in": n_in, "frames_out": n,
                      "seconds": dt_s, "frames_per_s": n_in / dt_s, "h2d_GBps": n_in * w * h * 3 * esz / dt_s / 1e9,
                      "d2h_GBps": n * ow * oh * 3 * esz / dt_s / 1e9,
                      "batch": args.batch or Stabilizer.default_batch(w * h * 3 * esz), "depth": args.depth,
                      "hyperlapse": [mult, nb], "checksum": sink.checksum}))


if __name__ == "__main__":
    main()
