#!/usr/bin/env python3
"""M2: end-to-end rate of the fused path with frames in pinned HOST memory -- pinned ring -> hipMemcpyAsync
H2D -> fused kernel -> hipMemcpyAsync D2H -> pinned, three streams, `depth` batches in flight (gyroflow_amd.pipeline).
PCIe-inclusive; this number is never bench.py's `value`."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gyroflow_amd import cv, FisheyeCalibrator, synthetic_frames
from gyroflow_amd.pipeline import BatchPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--batches", type=int, default=40)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--size", default="3840x2160")
args = ap.parse_args()
w, h = (int(v) for v in args.size.split("x"))
dev = torch.device("cuda", 0)
cal = FisheyeCalibrator(); cal.load_calibration_json(os.path.join(ROOT, "tests/golden/presets/hero7_4k_169.json"))
z = np.load(os.path.join(ROOT, "tests/golden/quats_gyroflow_sample.npz"))["stab_transform"][:, 3:7]
B = args.batch
ring = synthetic_frames(w, h, n=B)
hsrc = torch.from_numpy(np.stack(ring)).pin_memory()
count = [0]
def deliver(frames): count[0] += len(frames)
pipe = BatchPipeline(dev, (h, w), (h, w), torch.uint8, B, deliver, depth=args.depth)
xfs = [cal.frame_xforms(1.0, z[(k * B + np.arange(B)) % len(z)], output_dim=(w, h), new_img_dim=(w, h)) for k in range(args.batches + 3)]
def run(nb, k0):
    for k in range(nb):
        pipe.submit(hsrc, lambda din, dout, dxf: (cv.undistort_rotate(din, cal.D, dxf, (w, h), dst=dout[:B]), B)[1], xf=xfs[k0 + k])
    pipe.flush()
run(3, 0)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(args.batches, 3); torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = args.batches * B
print("M2 pipeline %dx%d u8: %d frames in %.3f s = %.0f frames/s (H2D+D2H %.1f GB/s each way), batch %d, depth %d" %
      (w, h, n, dt, n / dt, n * w * h * 3 / dt / 1e9, B, args.depth))
