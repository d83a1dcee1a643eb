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
    ap.add_argument("--bg", default="#000000", help='renderfile\'s bg_color: a colour, "REPLICATE", or "HISTORY" (frames chained on the GPU)')
    ap.add_argument("--sequential", action="store_true", help="the per-frame form of the loop (one blocking download per frame)")
    ap.add_argument("--rotate", type=int, default=-1, help="video_rotation: a cv2.rotate code folded into the kernel (-1 = none)")
    ap.add_argument("--overlap", action="store_true", help="HIP-event timeline of the three streams during the timed render (printed after the JSON line)")
    args = ap.parse_args()
    preset, w, h, ow, oh, dt = CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    line_out, sys.stdout = sys.stdout, sys.stderr        # stdout: the JSON line (and --overlap's table) only; the library's progress notes go to stderr
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
    st = Stabilizer(src, cal, q, fps, w, h, device=dev, video_rotation=args.rotate)
    if args.rotate in (0, 2):
        ow, oh = oh, ow              # remap the turned frame to a turned output
    mult, nb = (int(v) for v in args.hyperlapse.split(","))
    st.set_hyperlapse(mult, nb)
    kw = dict(out_size=(ow, oh), smoothingFocus=-2, batch=args.batch or None, pipeline_depth=args.depth, bg_color=args.bg, sequential=args.sequential)
    st.renderfile(0.0, (min(args.frames, 200) + 1.5) / fps, NullSink(), **kw)          # warm-up: allocations, first-touch, lens table
    torch.cuda.synchronize()
    sink = NullSink()
    if args.overlap and getattr(st, "_pipe", None) is not None:
        st._pipe.enable_trace()
    t0 = time.perf_counter()
    n = st.renderfile(0.0, (args.frames + 1.5) / fps, sink, **kw)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    n_in = args.frames
    esz = np.dtype(dt).itemsize
    sys.stdout = line_out
    print(json.dumps({"tool": "render_bench", "config": args.config, "source": args.source, "frames_in": n_in, "frames_out": n,
                      "seconds": dt_s, "frames_per_s": n_in / dt_s, "h2d_GBps": n_in * w * h * 3 * esz / dt_s / 1e9,
                      "d2h_GBps": n * ow * oh * 3 * esz / dt_s / 1e9,
                      "batch": args.batch or Stabilizer.default_batch(w * h * 3 * esz), "depth": args.depth,
                      "hyperlapse": [mult, nb], "bg_color": args.bg, "sequential": args.sequential, "rotate": args.rotate, "checksum": sink.checksum}))
    if args.overlap and getattr(st, "_pipe", None) is not None:
        overlap_report(st._pipe.trace_table())


def _inter(a, b):
    """Total length of the intersection of two lists of disjoint, sorted intervals."""
    i = j = 0
    tot = 0.0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def overlap_report(rows):
    """rows: BatchPipeline.trace_table().  The upload interval of a batch runs from its first frame's copy to its last (host staging
    of the frames in between included), so 'H2D' is the time the upload stream is owned by a batch, not pure copy time."""
    if not rows:
        print("# no trace rows")
        return
    H = [(r["h0"], r["h1"]) for r in rows]; Kk = [(r["k0"], r["k1"]) for r in rows]; Dd = [(r["d0"], r["d1"]) for r in rows]
    t0, t1 = min(h[0] for h in H), max(d[1] for d in Dd)
    wall = t1 - t0
    busy = lambda iv: sum(b - a for a, b in iv)
    print("# stream timeline from HIP events, %d batches, %.1f ms from the first upload to the last download" % (len(rows), wall))
    print("#   upload stream busy   %7.1f ms (%3.0f %% of that time)" % (busy(H), 100 * busy(H) / wall))
    print("#   kernel stream busy   %7.1f ms (%3.0f %%)" % (busy(Kk), 100 * busy(Kk) / wall))
    print("#   download stream busy %7.1f ms (%3.0f %%)" % (busy(Dd), 100 * busy(Dd) / wall))
    print("#   upload and download at the same time : %7.1f ms (%3.0f %% of the download time)" % (_inter(H, Dd), 100 * _inter(H, Dd) / max(busy(Dd), 1e-9)))
    print("#   kernel while an upload is in progress : %7.1f ms (%3.0f %% of the kernel time)" % (_inter(Kk, H), 100 * _inter(Kk, H) / max(busy(Kk), 1e-9)))
    print("#   kernel while a download is in progress: %7.1f ms (%3.0f %% of the kernel time)" % (_inter(Kk, Dd), 100 * _inter(Kk, Dd) / max(busy(Kk), 1e-9)))
    print("#   sum of the three busy times / wall    : %.2f (1.0 = nothing overlaps, 3.0 = everything does)" % ((busy(H) + busy(Kk) + busy(Dd)) / wall))
    print("#   first batches (ms): " + "; ".join("H %.1f-%.1f K %.1f-%.1f D %.1f-%.1f" % (r["h0"] - t0, r["h1"] - t0, r["k0"] - t0, r["k1"] - t0, r["d0"] - t0, r["d1"] - t0) for r in rows[:4]))


if __name__ == "__main__":
    main()
