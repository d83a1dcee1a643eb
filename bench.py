#!/usr/bin/env python3
"""bench.py -- throughput of the fused undistort+rotate remap hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A step = one launch of the fused kernel (gf_undistort_rotate_c3; for --config c4 followed by the superview launch,
gf_superview_cubic_u8c3) over one batch of B synthetic frames that are
already resident in HBM (mode M1 of BASELINE.md); every frame of every step uses its own rotation from the real
per-frame quaternion trace (tests/golden/quats_gyroflow_sample.npz).  Frames are independent, so with N ranks each
rank works through its own contiguous shard of the trace with no data-path collective (weak scaling: fixed work
per GPU); gloo carries only the start/stop barrier and the max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every timed launch on the stream
the kernel runs on; `cpu_baseline` times the CPU oracle (a port of the OpenCV arithmetic -- oracle/cv_oracle.c) on a
bounded sample of the same workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (preset, in_w, in_h, out_w, out_h, dtype, description)
    "c1": ("hero8_27k_43", 1920, 1080, 1920, 1080, "u8", "C1 1920x1080 u8 -> 1920x1080"),
    "c2": ("hero8_27k_43", 2704, 2028, 1920, 1080, "u8", "C2 2704x2028 u8 -> 1920x1080"),
    "c3": ("hero7_4k_169", 3840, 2160, 3840, 2160, "u8", "C3 3840x2160 4K u8 -> 3840x2160"),
    "c5": ("hero9_5k_169", 5312, 2988, 5312, 2988, "u16", "C5 5312x2988 u16 -> 5312x2988 (no blend)"),
    # C4: the fused remap to 2880x2160, then NonlinearStretch (bicubic) to 3840x2160 -- two launches per step
    "c4": ("hero9_4k_43", 4000, 3000, 2880, 2160, "u8", "C4 4000x3000 u8 -> 2880x2160 -> superview 3840x2160"),
}
STRETCH = {"c4": (3840, 2160)}
HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_frame(cfg):
    _, iw, ih, ow, oh, dt, _ = CONFIGS[cfg]
    esz = 1 if dt == "u8" else 2
    b = iw * ih * 3 * esz + ow * oh * 3 * esz         # source read once + destination written once; maps = 0
    if cfg in STRETCH:                                # the intermediate frame is read again, the stretched frame written
        b += ow * oh * 3 * esz + STRETCH[cfg][0] * STRETCH[cfg][1] * 3 * esz
    return b


def load_quats():
    z = np.load(os.path.join(ROOT, "tests", "golden", "quats_gyroflow_sample.npz"))
    q = z["stab_transform"][:, 3:7]
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def cpu_baseline(cfg, cal, quats, frames, n_frames, threads):
    """Reference-faithful CPU leg: fisheye map generation on ONE thread (cv::fisheye::initUndistortRectifyMap is not
    parallel) + bilinear remap striped over `threads` (cv::parallel_for_), via the CPU oracle."""
    from oracle import cv_oracle as orc
    _, iw, ih, ow, oh, dt, _ = CONFIGS[cfg]
    t_map = t_remap = 0.0
    dst = None
    ns = None
    if cfg in STRETCH:
        from gyroflow_amd import NonlinearStretch
        ns = NonlinearStretch((ow, oh), STRETCH[cfg]); ns.recompute_maps()
    for i in range(n_frames):
        K, D, R, P, od = cal._frame_matrices(1.0, (ow, oh), (iw, ih), quats[i % len(quats)], None, True)
        t0 = time.perf_counter()
        m1, m2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2)
        t1 = time.perf_counter()
        dst = orc.remap_linear(frames[i % len(frames)], m1, m2, dst=dst, nthreads=threads)
        if ns is not None:
            orc.remap_cubic(dst, ns.map1, ns.map2, nthreads=threads)
        t2 = time.perf_counter()
        t_map += t1 - t0
        t_remap += t2 - t1
    total = t_map + t_remap
    # SURVEY 8(d)'s second CPU figure, "best CPU": the map loop striped over all threads as well (same bits; OpenCV does
    # not do this), on a quarter of the sample.
    n_best = max(1, n_frames // 4)
    t_best = 0.0
    for i in range(n_best):
        K, D, R, P, od = cal._frame_matrices(1.0, (ow, oh), (iw, ih), quats[i % len(quats)], None, True)
        t0 = time.perf_counter()
        m1, m2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2, nthreads=threads)
        dst = orc.remap_linear(frames[i % len(frames)], m1, m2, dst=dst, nthreads=threads)
        if ns is not None:
            orc.remap_cubic(dst, ns.map1, ns.map2, nthreads=threads)
        t_best += time.perf_counter() - t0
    return {"value": n_frames / total, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frames of %s: map generation single-threaded (%.1f ms/frame, as OpenCV) + bilinear remap on %d "
                      "threads (%.1f ms/frame)" % (n_frames, CONFIGS[cfg][6], 1e3 * t_map / n_frames, threads,
                                                   1e3 * t_remap / n_frames),
            "best_cpu": {"value": n_best / t_best, "unit": "frames/s", "cores": threads,
                         "sample": "%d frames, map generation striped over %d threads too (not what OpenCV does)" % (n_best, threads)}}


def touched_source_bytes(cfg, cal, quats, idx):
    """SURVEY 8(d)'s touched-footprint figure: bytes of the unique source pixels that the bilinear taps of one output frame
    reference, from the CPU oracle's maps (the crop / zoom leaves part of every source frame unread); mean over frames `idx`."""
    from oracle import cv_oracle as orc
    _, iw, ih, ow, oh, dt, _ = CONFIGS[cfg]
    esz = 1 if dt == "u8" else 2
    tot = 0
    for i in idx:
        K, D, R, P, od = cal._frame_matrices(1.0, (ow, oh), (iw, ih), quats[i % len(quats)], None, True)
        m1, _ = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2, nthreads=min(16, os.cpu_count() or 1))
        sx = m1[..., 0].astype(np.int64).ravel(); sy = m1[..., 1].astype(np.int64).ravel()
        mask = np.zeros(ih * iw, np.bool_)
        for dy in (0, 1):
            for dx in (0, 1):
                x = sx + dx; y = sy + dy
                ok = (x >= 0) & (x < iw) & (y >= 0) & (y < ih)
                mask[(y[ok] * iw + x[ok])] = True
        tot += int(mask.sum()) * 3 * esz
    return tot / len(idx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    # 128 frames per launch: launch time fits 68 us + 19.1 us x frames (start-up and a tail of thinning workgroups, DESIGN.md
    # section 3), so larger launches run closer to the steady-state rate; 2 x 3.2 GB of HBM at 4K
    ap.add_argument("--batch", type=int, default=128, help="frames per step (per GPU)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames in the CPU baseline sample (0 = skip)")
    ap.add_argument("--border", default="constant", choices=["constant", "replicate"])
    args = ap.parse_args()

    import torch
    from gyroflow_amd import dist as gdist

    rank, local_rank, world = gdist.init_from_env()      # gloo: timing barrier only; no data-path collective
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    from gyroflow_amd import cv, FisheyeCalibrator, synthetic_frames
    from gyroflow_amd.stabilizer import frame_shard

    preset, iw, ih, ow, oh, dt, desc = CONFIGS[args.config]
    cal = FisheyeCalibrator()
    cal.load_calibration_json(os.path.join(ROOT, "tests", "golden", "presets", preset + ".json"))
    quats = load_quats()
    B, K, W = args.batch, args.steps, args.warmup

    # this rank's contiguous shard of the clip (BASELINE config: 3600 frames over N GPUs); the trace is tiled to length
    total_frames = max(3600, world * (K + W) * B)
    lo, hi = frame_shard(total_frames, rank, world)
    idx = (lo + np.arange((K + W) * B)) % len(quats)
    xf = cal.frame_xforms(1.0, quats[idx], output_dim=(ow, oh), new_img_dim=(iw, ih))
    xf_dev = cv.xforms_to_device(xf, dev).reshape(K + W, B, -1)

    np_dt = np.uint8 if dt == "u8" else np.uint16
    ring = synthetic_frames(iw, ih, n=8, dtype=np_dt)
    src = torch.empty((B, ih, iw, 3), dtype=torch.uint8 if dt == "u8" else torch.uint16, device=dev)
    for i in range(B):
        src[i].copy_(torch.from_numpy(ring[(i + rank) % len(ring)]))
    dst = torch.empty((B, oh, ow, 3), dtype=src.dtype, device=dev)
    border = cv.BORDER_CONSTANT if args.border == "constant" else cv.BORDER_REPLICATE

    if args.config in STRETCH:
        from gyroflow_amd import NonlinearStretch
        ns = NonlinearStretch((ow, oh), STRETCH[args.config]); ns.recompute_maps()
        mx = torch.from_numpy(ns.map1[0].copy()).to(dev); my = torch.from_numpy(ns.map2[:, 0].copy()).to(dev)
        out2 = torch.empty((B, STRETCH[args.config][1], STRETCH[args.config][0], 3), dtype=src.dtype, device=dev)

        def step(s):
            cv.undistort_rotate(src, cal.D, xf_dev[s], (ow, oh), dst=dst, borderMode=border)
            cv.superview_cubic(dst, mx, my, dst=out2)
    else:
        def step(s):
            cv.undistort_rotate(src, cal.D, xf_dev[s], (ow, oh), dst=dst, borderMode=border)

    for s in range(W):
        step(s)
    torch.cuda.synchronize(dev)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    gdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in range(K):
        ev0[s].record()
        step(W + s)
        ev1[s].record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    gdist.barrier()
    elapsed = gdist.max_over_ranks(elapsed)
    kernel_ms = gdist.max_over_ranks(float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)])))
    checksum = int(dst[0, ::64, ::64].to(torch.int64).sum())

    if rank == 0:
        bytes_per_launch = algorithmic_bytes_per_frame(args.config) * B
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            ent = tj.get("%s_b%d" % (args.config, B))
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
        out = {
            "metric": "4K frames/sec, fused undistort+rotate remap (frames resident in HBM)" if args.config == "c3"
                      else "frames/sec, fused undistort+rotate remap (frames resident in HBM)",
            "value": world * K * B / elapsed, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dt, "data": "synthetic",
            "config": {"workload": desc + ", batch %d frames/step/GPU, %s border, per-frame quaternions from the "
                                   "GX016015.MP4.gyroflow trace, preset %s" % (B, args.border, preset),
                       "frames_per_step_per_gpu": B, "sharding": "contiguous frame ranges, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": {"direct": "k_undistort_rotate", "rl": "k_undistort_rotate_rl"}.get(os.environ.get("GF_K1", "rp"), "k_undistort_rotate_rp")
                                   + (" + k_superview_cubic" if args.config in STRETCH else ""),
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "checksum": checksum,
        }
        if world == 1:
            n_cpu = args.cpu_frames
            if n_cpu < 0:
                n_cpu = {"c1": 400, "c2": 400, "c3": 96, "c4": 48, "c5": 36}[args.config]      # about 10 s of CPU work
            if n_cpu > 0:
                threads = min(16, os.cpu_count() or 1)
                out["cpu_baseline"] = cpu_baseline(args.config, cal, quats, ring, n_cpu, threads)
                # touched-footprint variant of the roofline figure (the honest lower one): unique source bytes + output bytes
                sample = [int(v) for v in idx[W * B:(W + 1) * B:max(1, B // 4)]]
                tb = touched_source_bytes(args.config, cal, quats, sample)
                _, _, _, ow_, oh_, dt_, _ = CONFIGS[args.config]
                t_bytes = (tb + ow_ * oh_ * 3 * (1 if dt_ == "u8" else 2)) * B
                if args.config not in STRETCH:
                    out["roofline"]["touched"] = {"bytes_per_launch": int(t_bytes), "achieved": t_bytes / (kernel_ms * 1e-3) / 1e9,
                                                  "frac": t_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                  "sample": "unique source pixels under the taps, CPU oracle maps of %d frames of the first timed launch" % len(sample)}
        print(json.dumps(out), flush=True)
    gdist.shutdown()


if __name__ == "__main__":
    main()
