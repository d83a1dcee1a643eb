"""The opt-in LDS-staged variant of the fused kernel (GF_USE_LDS_TILES=1, read when the library loads) must give the
same bytes as the default direct-gather kernel.  The switch is process-wide, so the check runs in a child process."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from gyroflow_amd import cv, FisheyeCalibrator, synthetic_frames
dev = torch.device("cuda", 0)
cal = FisheyeCalibrator(); cal.load_calibration_json(os.path.join(%(root)r, "tests/golden/presets/hero7_4k_169.json"))
z = np.load(os.path.join(%(root)r, "tests/golden/quats_gyroflow_sample.npz"))["stab_transform"]
h = hashlib.sha256()
for dtype, (w, h_), (ow, oh), border, fov in [(np.uint8, (960, 540), (960, 540), 0, 1.0), (np.uint8, (961, 541), (640, 362), 1, 1.4),
                                              (np.uint16, (672, 380), (672, 380), 0, 0.8), (np.uint8, (960, 540), (960, 540), 5, 0.6)]:
    # 16-byte aligned rows are required by the LDS variant; the odd-sized case exercises its fallback to the direct kernel
    frames = np.stack(synthetic_frames(w, h_, n=3, dtype=dtype, seed=3))
    q = z[200:800:200, 3:7]
    xf = cv.xforms_to_device(cal.frame_xforms(fov, q, output_dim=(ow, oh), new_img_dim=(w, h_)), dev)
    prev = torch.full((3, oh, ow, 3), 77, dtype=torch.from_numpy(frames).dtype, device=dev)
    out = cv.undistort_rotate(torch.from_numpy(frames).to(dev), cal.D, xf, (ow, oh), dst=prev, borderMode=border, borderValue=(1, 2, 3))
    h.update(out.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
"""


def _run(env_extra):
    env = dict(os.environ)
    env.pop("GF_USE_LDS_TILES", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]


def test_lds_tile_variant_matches_direct_kernel(gpu):
    assert _run({"GF_USE_LDS_TILES": "1"}) == _run({})
