#!/usr/bin/env python3
"""tools/twocall_bench.py [--config c3] [--frames 300]: the reference's own per-frame sequence through the drop-in API --
FisheyeCalibrator.get_maps(quat=...) then cv.remap(frame, map1, map2, INTER_LINEAR) (/root/reference/stabilizer.py:1423-1443) --
one frame per call, frames resident in HBM.  Prints one JSON line: frames/s of the sequence as Python drives it, and the
average duration of the two kernels (K2 k_fisheye_maps_rp, K4 k_remap_linear) from HIP events on the launch stream, with the
HBM fraction of each (K2 writes 6 B per output pixel; K4 reads the maps and the frame and writes the frame)."""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gyroflow_amd import cv, FisheyeCalibrator, synthetic_frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c5_noblend"])
    ap.add_argument("--frames", type=int, default=300)
    args = ap.parse_args()
    line_out, sys.stdout = sys.stdout, sys.stderr
    preset, iw, ih, ow, oh, dt, desc, _ = bench.CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(os.path.join(ROOT, "tests", "golden", "presets", preset + ".json"))
    quats = bench.load_quats()
    np_dt = np.uint8 if dt == "u8" else np.uint16
    ring = [torch.from_numpy(f).to(dev) for f in synthetic_frames(iw, ih, n=8, dtype=np_dt)]
    dst = torch.empty((oh, ow, 3), dtype=ring[0].dtype, device=dev)
    n = args.frames

    def one(i, ev=None):
        q = quats[i % len(quats)]
        if ev: ev[0].record()
        m1, m2 = cal.get_maps(1.0, output_dim=(ow, oh), new_img_dim=(iw, ih), quat=q, device=dev)
        if ev: ev[1].record()
        cv.remap(ring[i % 8], m1, m2, cv.INTER_LINEAR, dst=dst, borderMode=cv.BORDER_CONSTANT)
        if ev: ev[2].record()

    for i in range(20):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        one(i)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    for i in range(n):
        one(i, evs[i])
    torch.cuda.synchronize()
    k2 = float(np.median([e[0].elapsed_time(e[1]) for e in evs])) * 1e-3
    k4 = float(np.median([e[1].elapsed_time(e[2]) for e in evs])) * 1e-3
    esz = 1 if dt == "u8" else 2
    b2 = ow * oh * 6
    b4 = ow * oh * 6 + iw * ih * 3 * esz + ow * oh * 3 * esz
    print(json.dumps({"tool": "twocall_bench", "config": args.config, "frames": n, "frames_per_s": n / wall, "us_per_frame_wall": 1e6 * wall / n,
                      "get_maps": {"kernel": "k_fisheye_maps_rp", "us": 1e6 * k2, "algorithmic_bytes": b2, "GBps": b2 / k2 / 1e9, "frac": b2 / k2 / 1e9 / bench.HBM_PEAK_GBPS,
                                   "note": "event to event around the call: host work of the call that the GPU waits for is inside"},
                      "remap": {"kernel": "k_remap_linear", "us": 1e6 * k4, "algorithmic_bytes": b4, "GBps": b4 / k4 / 1e9, "frac": b4 / k4 / 1e9 / bench.HBM_PEAK_GBPS},
                      "checksum": int(dst[::64, ::64].to(torch.int64).sum())}), file=line_out, flush=True)


if __name__ == "__main__":
    main()
