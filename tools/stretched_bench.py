#!/usr/bin/env python3
"""Rate of the stretched-preset fused entry (gf_undistort_rotate_stretched_c3) at the preset's own size: 3840x2160 -> 1920x1080,
RunCam 5 Orange 4K XV (input_horizontal_stretch 0.75), 64 frames per launch, frames resident.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gyroflow_amd import FisheyeCalibrator, synthetic_frames, cv

dev = torch.device("cuda", 0)
cal = FisheyeCalibrator()
cal.load_calibration_json(os.path.join(ROOT, "tests", "golden", "presets", "runcam5_stretched.json"))
q = np.load(os.path.join(ROOT, "tests", "golden", "quats_gyroflow_sample.npz"))["stab_transform"][200:264, 3:7]
n = 64
ring = torch.from_numpy(np.stack(synthetic_frames(3840, 2160, n=8))).to(dev)
src = ring.repeat(n // 8, 1, 1, 1)
xf = cv.xforms_to_device(cal.frame_xforms(1.0, q, output_dim=(1920, 1080), new_img_dim=(3840, 2160)), dev)
dst = torch.empty((n, 1080, 1920, 3), dtype=torch.uint8, device=dev)
for _ in range(3):
    cv.undistort_rotate(src, cal.D, xf, (1920, 1080), dst=dst, map_scale=cal.map_scale())
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    cv.undistort_rotate(src, cal.D, xf, (1920, 1080), dst=dst, map_scale=cal.map_scale())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(json.dumps({"tool": "stretched_bench", "frames_per_launch": n, "ms_per_launch": dt * 1e3, "us_per_frame": dt * 1e6 / n,
                  "frames_per_s": n / dt, "checksum": int(dst[0, ::64, ::64].to(torch.int64).sum())}))
