#!/usr/bin/env python3
"""C4: 4000x3000 u8 -> undistort+rotate to 2880x2160 -> superview stretch to 3840x2160, two launches per batch
(the intermediate frame is needed: the reference rounds to uint8 between the two resampling steps)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gyroflow_amd import cv, FisheyeCalibrator, NonlinearStretch, synthetic_frames

dev = torch.device("cuda", 0)
B = 16
cal = FisheyeCalibrator(); cal.load_calibration_json(os.path.join(ROOT, "tests/golden/presets/hero9_4k_43.json"))
z = np.load(os.path.join(ROOT, "tests/golden/quats_gyroflow_sample.npz"))["stab_transform"][:, 3:7]
ring = synthetic_frames(4000, 3000, n=4)
src = torch.from_numpy(np.stack([ring[i % 4] for i in range(B)])).to(dev)
xf = cv.xforms_to_device(cal.frame_xforms(1.0, z[100:100 + B], output_dim=(2880, 2160), new_img_dim=(4000, 3000)), dev)
ns = NonlinearStretch((2880, 2160), (3840, 2160)); ns.recompute_maps()
mx = torch.from_numpy(ns.map1[0].copy()).to(dev); my = torch.from_numpy(ns.map2[:, 0].copy()).to(dev)
mid = torch.empty((B, 2160, 2880, 3), dtype=torch.uint8, device=dev)
out = torch.empty((B, 2160, 3840, 3), dtype=torch.uint8, device=dev)
def step():
    cv.undistort_rotate(src, cal.D, xf, (2880, 2160), dst=mid)
    cv.superview_cubic(mid, mx, my, dst=out)
for _ in range(3): step()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
t1 = t2 = 0.0
for _ in range(10):
    e[0].record(); cv.undistort_rotate(src, cal.D, xf, (2880, 2160), dst=mid); e[1].record(); cv.superview_cubic(mid, mx, my, dst=out); e[2].record()
    torch.cuda.synchronize(); t1 += e[0].elapsed_time(e[1]); t2 += e[1].elapsed_time(e[2])
t1 /= 10*B; t2 /= 10*B
alg = 36.0e6 + 2*18.6624e6 + 24.8832e6
print("C4 per frame: undistort+rotate %.1f us, superview (bicubic) %.1f us, total %.1f us = %.0f frames/s; algorithmic %.2f MB/frame -> %.0f GB/s (%.1f%% of 8 TB/s)" %
      (1e3*t1, 1e3*t2, 1e3*(t1+t2), 1e3/(t1+t2), alg/1e6, alg/((t1+t2)*1e-3)/1e9, 100*alg/((t1+t2)*1e-3)/8e12))
