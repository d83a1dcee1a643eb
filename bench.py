#!/usr/bin/env python3
"""bench.py -- throughput of the fused undistort+rotate remap hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--batch B] [--scaling weak|strong] [--mode m1|m2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

`--gpus N` without a launcher (no WORLD_SIZE in the environment) makes this process the launcher: before touching any device
it starts N fresh ranks of itself, one per GPU (gloo rendezvous on 127.0.0.1), and exits with their status; it refuses to run
with fewer GPUs than asked (exit 2) rather than print a line for another configuration.

Default (what the driver runs): mode M1 of BASELINE.md, weak scaling.  A step = one launch of the fused kernel
(gf_undistort_rotate_c3; c5: gf_undistort_rotate_blend_c3 with set_hyperlapse(4, 4); c4: the fused remap + superview
stretch) over one batch of B synthetic frames that are already resident in HBM; every frame of every step uses its own
rotation from the real per-frame quaternion trace (tests/golden/quats_gyroflow_sample.npz).  Frames are independent, so
with N ranks each rank works through its own contiguous shard of the trace with no data-path collective; gloo carries
only the start/stop barrier and the max-over-ranks time.

  --scaling strong : a step = one pass over BASELINE's fixed clip (C3: 3600 frames) sharded over the ranks with
                     frame_shard(); every rank renders its shard in launches of up to B frames.
  --mode m2        : frames start and end in pinned HOST memory; every step pushes one batch through the three-stream
                     BatchPipeline (H2D -> kernel -> D2H), so N ranks measure the PCIe / host-DRAM limit.  PCIe-inclusive
                     numbers are labelled as such and are never the default `value`.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every timed launch on the stream the
kernel runs on; `cpu_baseline` times the CPU oracle (a port of the OpenCV arithmetic -- oracle/cv_oracle.c) on a bounded
sample of the same workload (BASELINE.md section 2 protocol), rank 0 at N=1 only; `pcie_pipeline` is the end-to-end rate
of Stabilizer.renderfile from pinned host frames (N=1, reported beside `value`, never as it).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (preset, in_w, in_h, out_w, out_h, dtype, description, frames of BASELINE's clip)
    "c1": ("hero8_27k_43", 1920, 1080, 1920, 1080, "u8", "C1 1920x1080 u8 -> 1920x1080", 300),
    "c2": ("hero8_27k_43", 2704, 2028, 1920, 1080, "u8", "C2 2704x2028 u8 -> 1920x1080", 300),
    "c3": ("hero7_4k_169", 3840, 2160, 3840, 2160, "u8", "C3 3840x2160 4K u8 -> 3840x2160", 3600),
    # C5 as BASELINE states it: set_hyperlapse(4, 4) -- four remapped input frames are averaged in float64 into one output frame
    "c5": ("hero9_5k_169", 5312, 2988, 5312, 2988, "u16", "C5 5312x2988 u16 -> 5312x2988, fused 4-frame hyperlapse blend", 1200),
    "c5_noblend": ("hero9_5k_169", 5312, 2988, 5312, 2988, "u16", "5312x2988 u16 -> 5312x2988 (no blend)", 1200),
    # C4: remap to 2880x2160, then NonlinearStretch (bicubic) to 3840x2160.  c4 = the two launches (what cv.undistort_rotate_superview
    # does by default: the faster form, 35 vs 46 us per frame); c4_fused = the one-launch kernel whose intermediate frame stays in LDS
    "c4": ("hero9_4k_43", 4000, 3000, 2880, 2160, "u8", "C4 4000x3000 u8 -> 2880x2160 -> superview 3840x2160 (two launches)", 300),
    "c4_fused": ("hero9_4k_43", 4000, 3000, 2880, 2160, "u8", "C4 4000x3000 u8 -> 2880x2160 -> superview 3840x2160 (one fused launch)", 300),
}
STRETCH = {"c4": (3840, 2160), "c4_fused": (3840, 2160)}
BLEND = {"c5": 4}
HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_unit(cfg):
    """Bytes per OUTPUT frame: every source pixel read once, every output pixel written once, maps = 0."""
    _, iw, ih, ow, oh, dt, _, _ = CONFIGS[cfg]
    esz = 1 if dt == "u8" else 2
    src, dst = iw * ih * 3 * esz, ow * oh * 3 * esz
    if cfg in BLEND:
        return BLEND[cfg] * src + dst                 # n_blend inputs, one output; intermediates stay in registers
    if cfg == "c4_fused":
        return src + STRETCH[cfg][0] * STRETCH[cfg][1] * 3 * esz      # fused: the 2880x2160 intermediate never reaches HBM
    if cfg == "c4":
        return src + 2 * dst + STRETCH[cfg][0] * STRETCH[cfg][1] * 3 * esz
    return src + dst


def load_quats():
    z = np.load(os.path.join(ROOT, "tests", "golden", "quats_gyroflow_sample.npz"))
    q = z["stab_transform"][:, 3:7]
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, cal, quats, frames, n_units, threads, warm=10):
    """Reference-faithful CPU leg: per input frame, fisheye map generation on ONE thread
    (cv::fisheye::initUndistortRectifyMap is not parallel) + bilinear remap striped over `threads` (cv::parallel_for_);
    c4 adds the bicubic stretch, c5 the three numpy float64 blend passes of stabilizer.py:1434,1447,1468.  `warm` untimed
    frames, then n_units timed output frames; median / p10 / p90 of the per-unit times."""
    from oracle import cv_oracle as orc
    _, iw, ih, ow, oh, dt, _, _ = CONFIGS[cfg]
    nb = BLEND.get(cfg, 1)
    ns = None
    if cfg in STRETCH:
        from gyroflow_amd import NonlinearStretch
        ns = NonlinearStretch((ow, oh), STRETCH[cfg]); ns.recompute_maps()
    state = dict(dst=None, temp=np.zeros((oh, ow, 3), np.float64), i=0)

    def unit(map_threads):
        t_map = t_rest = 0.0
        for b in range(nb):
            i = state["i"]; state["i"] += 1
            K, D, R, P, od = cal._frame_matrices(1.0, (ow, oh), (iw, ih), quats[i % len(quats)], None, True)
            t0 = time.perf_counter()
            m1, m2 = orc.fisheye_init_undistort_rectify_map(K, D, R, P, od, orc.CV_16SC2, nthreads=map_threads)
            t1 = time.perf_counter()
            state["dst"] = orc.remap_linear(frames[i % len(frames)], m1, m2, dst=state["dst"], nthreads=threads)
            if ns is not None:
                orc.remap_cubic(state["dst"], ns.map1, ns.map2, nthreads=threads)
            if nb > 1:
                if b == 0:
                    state["temp"] = state["temp"] * 0.0
                state["temp"] += 1 / nb * state["dst"].astype(np.float64)
                if b == nb - 1:
                    state["temp"].astype(frames[0].dtype)
            t2 = time.perf_counter()
            t_map += t1 - t0; t_rest += t2 - t1
        return t_map, t_rest

    for _ in range(max(1, warm // nb)):
        unit(1)
    per = np.array([unit(1) for _ in range(n_units)])
    tot = per.sum(axis=1)
    n_best = max(3, n_units // 4)
    unit(threads)
    best = np.array([sum(unit(threads)) for _ in range(n_best)])
    return {"value": n_units / tot.sum(), "unit": "frames/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
            "ms_per_frame": {"median": 1e3 * float(np.median(tot)), "p10": 1e3 * float(np.percentile(tot, 10)), "p90": 1e3 * float(np.percentile(tot, 90))},
            "sample": "%d output frames of %s after %d warm-up frames: map generation single-threaded (%.1f ms/unit, as OpenCV) + "
                      "remap%s on %d threads (%.1f ms/unit)" % (n_units, CONFIGS[cfg][6], warm, 1e3 * per[:, 0].mean(),
                                                                 " + bicubic stretch" if ns is not None else " + float64 blend passes" if nb > 1 else "",
                                                                 threads, 1e3 * per[:, 1].mean()),
            "best_cpu": {"value": n_best / best.sum(), "unit": "frames/s", "cores": threads,
                         "sample": "%d output frames, map generation striped over %d threads too (not what OpenCV does)" % (n_best, threads)}}


def touched_source_bytes(cfg, cal, quats, idx):
    """SURVEY 8(d)'s touched-footprint figure: bytes of the unique source pixels that the bilinear taps of one output frame
    reference, from the CPU oracle's maps (the crop / zoom leaves part of every source frame unread); mean over frames `idx`."""
    from oracle import cv_oracle as orc
    _, iw, ih, ow, oh, dt, _, _ = CONFIGS[cfg]
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


def pcie_pipeline(cfg, dev, frames=640, batch=32, depth=3):
    """End-to-end Stabilizer.renderfile: pinned host frames -> H2D -> fused kernel -> D2H -> sink (mode M2, one GPU)."""
    import torch
    from gyroflow_amd import Stabilizer, FisheyeCalibrator, NullSink, PinnedRingSource, synthetic_frames
    preset, iw, ih, ow, oh, dt, _, _ = CONFIGS[cfg]
    cal = FisheyeCalibrator()
    cal.load_calibration_json(os.path.join(ROOT, "tests", "golden", "presets", preset + ".json"))
    q = load_quats()
    q = np.tile(q, (frames // len(q) + 2, 1))[:frames + 8]
    ring = synthetic_frames(iw, ih, n=8, dtype=np.uint8 if dt == "u8" else np.uint16)
    st = Stabilizer(PinnedRingSource(ring, frames + 8), cal, q, 60.0, iw, ih, device=dev)
    kw = dict(out_size=(ow, oh), smoothingFocus=-2, batch=batch, pipeline_depth=depth)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        st.renderfile(0.0, (min(frames, 4 * batch) + 1.5) / 60.0, NullSink(), **kw)      # warm-up: page-locks the ring, lens table
        torch.cuda.synchronize(dev)
        sink = NullSink()
        t0 = time.perf_counter()
        n = st.renderfile(0.0, (frames + 1.5) / 60.0, sink, **kw)
        torch.cuda.synchronize(dev)
        dt_s = time.perf_counter() - t0
    esz = 1 if dt == "u8" else 2
    return {"fps": frames / dt_s, "GBps_each_way": frames * iw * ih * 3 * esz / dt_s / 1e9, "batch": batch, "depth": depth,
            "frames": frames, "frames_out": n,
            "path": "Stabilizer.renderfile: PinnedRingSource -> BatchPipeline (3 streams) -> NullSink; PCIe-inclusive, never `value`"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    # 128 frames per launch: launch time fits 68 us + 19 us x frames (start-up and a tail of thinning workgroups, DESIGN.md
    # section 3), so larger launches run closer to the steady-state rate; 2 x 3.2 GB of HBM at 4K
    ap.add_argument("--batch", type=int, default=0, help="input frames per launch (per GPU); 0 = 128 (c5: 64)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="output frames in the CPU baseline sample (0 = skip)")
    ap.add_argument("--border", default="constant", choices=["constant", "replicate"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--mode", default="m1", choices=["m1", "m2"])
    ap.add_argument("--no-pcie", action="store_true", help="skip the pcie_pipeline leg")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than GPUs (ranks share devices: a rehearsal of the launch path, not a scaling number)")
    ap.add_argument("--dry-run", action="store_true", help="start the ranks, meet in the gloo group, print the rank map and exit (no GPU work)")
    args = ap.parse_args()

    from gyroflow_amd import dist as gdist

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # No launcher around us: this process becomes the launcher.  It starts one fresh rank per GPU BEFORE anything here
        # has touched a device (torch.cuda.device_count() does not initialise one) and never computes itself.
        import torch
        n_dev = 0 if args.dry_run else torch.cuda.device_count()
        if not args.dry_run and n_dev < args.gpus and not args.oversubscribe:
            print("bench.py: --gpus %d but %d GPU(s) visible; refusing to print a line for fewer GPUs than asked "
                  "(--oversubscribe rehearses the launch path with ranks sharing devices)" % (args.gpus, n_dev), file=sys.stderr)
            sys.exit(2)
        sys.exit(gdist.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        # a launcher's WORLD_SIZE and --gpus disagree: the line would carry an n_gpus nobody asked for (checked before the
        # rendezvous, which would wait for ranks that do not exist)
        if os.environ.get("RANK", "0") == "0":
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE %s ranks" % (args.gpus, os.environ.get("WORLD_SIZE")), file=sys.stderr)
        sys.exit(2)
    # stdout carries the ONE JSON line and nothing else: whatever is printed on the way -- the preset loader's version note
    # (unconditional, like the reference's), gloo's "[Gloo] Rank 0 is connected ..." from C++ -- goes to stderr, so file
    # descriptor 1 itself is pointed there and the line is written to a copy of the original
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    rank, local_rank, world = gdist.init_from_env()      # gloo: timing barrier only; no data-path collective
    if args.dry_run:
        gdist.barrier()
        ranks = gdist.sum_over_ranks(1.0)
        top = gdist.max_over_ranks(float(rank))
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_met": int(ranks), "max_rank": int(top)}), file=line_out, flush=True)
        gdist.shutdown()
        return

    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    n_dev = torch.cuda.device_count()
    if world > n_dev and not args.oversubscribe:
        if rank == 0:
            print("bench.py: %d ranks but %d GPU(s) visible (--oversubscribe lets ranks share devices)" % (world, n_dev), file=sys.stderr)
        gdist.shutdown()
        sys.exit(2)
    dev_index = local_rank % n_dev
    # one worker per GPU, on the CPUs next to it -- before torch.cuda / HIP start their own threads, so those inherit the mask
    pinned_cpus = gdist.pin_to_gpu(dev_index) if world > 1 or args.mode == "m2" else None
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from gyroflow_amd import cv, FisheyeCalibrator, synthetic_frames
    from gyroflow_amd.stabilizer import frame_shard

    cfg = args.config
    preset, iw, ih, ow, oh, dt, desc, clip_frames = CONFIGS[cfg]
    nb = BLEND.get(cfg, 1)
    cal = FisheyeCalibrator()
    cal.load_calibration_json(os.path.join(ROOT, "tests", "golden", "presets", preset + ".json"))
    quats = load_quats()
    B = args.batch or (64 if cfg.startswith("c5") else 128)
    B -= B % nb
    K, W = args.steps, args.warmup
    tdt = torch.uint8 if dt == "u8" else torch.uint16
    np_dt = np.uint8 if dt == "u8" else np.uint16
    border = cv.BORDER_CONSTANT if args.border == "constant" else cv.BORDER_REPLICATE

    ring = synthetic_frames(iw, ih, n=8, dtype=np_dt)
    src = torch.empty((B, ih, iw, 3), dtype=tdt, device=dev)
    for i in range(B):
        src[i].copy_(torch.from_numpy(ring[(i + rank) % len(ring)]))
    n_out_max = B // nb
    dst = torch.empty((n_out_max, oh, ow, 3), dtype=tdt, device=dev)
    if cfg in STRETCH:
        from gyroflow_amd import NonlinearStretch
        ns = NonlinearStretch((ow, oh), STRETCH[cfg]); ns.recompute_maps()
        mx = torch.from_numpy(ns.map1[0].copy()).to(dev); my = torch.from_numpy(ns.map2[:, 0].copy()).to(dev)
        out2 = torch.empty((B, STRETCH[cfg][1], STRETCH[cfg][0], 3), dtype=tdt, device=dev)

    def launch(n_in, xf_dev, s=src, d=dst):
        """one launch over n_in input frames (a multiple of n_blend); returns the output frame count"""
        if nb > 1:
            cv.undistort_rotate_blend(s[:n_in], cal.D, xf_dev, (ow, oh), nb, nb, dst=d[:n_in // nb], borderMode=border)
            return n_in // nb
        if cfg == "c4_fused":
            cv.undistort_rotate_superview(s[:n_in], cal.D, xf_dev, (ow, oh), mx, my, dst=out2[:n_in], borderMode=border, fused=True)
        elif cfg == "c4":
            cv.undistort_rotate_superview(s[:n_in], cal.D, xf_dev, (ow, oh), mx, my, dst=out2[:n_in], borderMode=border, mid=d[:n_in])
        else:
            cv.undistort_rotate(s[:n_in], cal.D, xf_dev, (ow, oh), dst=d[:n_in], borderMode=border)
        return n_in

    # ---- the work of this rank: a list of steps, each a list of launches (n_in, transforms)
    if args.scaling == "weak":
        # this rank's contiguous shard of a clip long enough for every rank (the trace is tiled to length)
        total_frames = max(clip_frames, world * (K + W) * B)
        lo, hi = frame_shard(total_frames, rank, world, nb)
        idx = (lo + np.arange((K + W) * B)) % len(quats)
        xf_all = cv.xforms_to_device(cal.frame_xforms(1.0, quats[idx], output_dim=(ow, oh), new_img_dim=(iw, ih)), dev).reshape(K + W, B, -1)
        steps = [[(B, xf_all[s])] for s in range(K + W)]
        units_per_step = world * (B // nb)
    else:
        # strong scaling: every step is one pass over BASELINE's clip, sharded by frame_shard (aligned to blend groups)
        lo, hi = frame_shard(clip_frames, rank, world, nb)
        idx = np.arange(lo, hi) % len(quats)
        xf_sh = cv.xforms_to_device(cal.frame_xforms(1.0, quats[idx], output_dim=(ow, oh), new_img_dim=(iw, ih)), dev)
        one = []
        for a in range(0, hi - lo, B):
            n_in = min(B, hi - lo - a); n_in -= n_in % nb
            if n_in:
                one.append((n_in, xf_sh[a:a + n_in]))
        steps = [one] * (K + W)
        units_per_step = sum((b - a) // nb for a, b in (frame_shard(clip_frames, r, world, nb) for r in range(world)))

    pipe = None
    if args.mode == "m2":
        from gyroflow_amd.pipeline import BatchPipeline
        delivered = [0]
        pipe = BatchPipeline(dev, (ih, iw), (oh, ow), tdt, B, lambda fr: delivered.__setitem__(0, delivered[0] + len(fr)), depth=3, out_batch=n_out_max)
        hsrc = torch.from_numpy(np.stack([ring[(i + rank) % len(ring)] for i in range(B)])).pin_memory()
        # the transforms of every launch as host arrays (the pipeline uploads them with the frames), outside the timed region
        cache = {}
        steps_host = [[(n, cache.setdefault(id(x), x.cpu().numpy())) for n, x in st] for st in steps]

    def run_step(s):
        if pipe is None:
            for n_in, xf in steps[s]:
                launch(n_in, xf)
        else:
            for n_in, xh in steps_host[s]:
                pipe.submit(hsrc[:n_in], lambda din, dout, dxf, n=n_in: launch(n, dxf, din, dout), xf=xh)

    for s in range(W):
        run_step(s)
    if pipe is not None:
        pipe.flush()
    torch.cuda.synchronize(dev)
    n_launch = sum(len(steps[W + s]) for s in range(K))
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    gdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in range(K):
        if pipe is None:
            ev0[s].record()
        run_step(W + s)
        if pipe is None:
            ev1[s].record()
    if pipe is not None:
        pipe.flush()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    gdist.barrier()
    elapsed = gdist.max_over_ranks(elapsed)
    kernel_ms = None
    if pipe is None:        # average launch duration from the HIP events on the launch stream (torch's current stream)
        kernel_ms = gdist.max_over_ranks(float(np.sum([a.elapsed_time(b) for a, b in zip(ev0, ev1)])) / max(n_launch, 1))
    check = out2 if cfg in STRETCH else dst
    checksum = int(check[0, ::64, ::64].to(torch.int64).sum())

    if rank == 0:
        unit_bytes = algorithmic_bytes_per_unit(cfg)
        out = {
            "metric": ("4K frames/sec, fused undistort+rotate remap" if cfg == "c3" else "frames/sec, fused undistort+rotate remap")
                      + (" (frames resident in HBM)" if pipe is None else " (PCIe-inclusive: pinned host -> GPU -> pinned host)"),
            "value": K * units_per_step / elapsed, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dt, "data": "synthetic", "mode": args.mode,
            "config": {"workload": desc + ", %d input frames/launch/GPU, %s border, per-frame quaternions from the "
                                   "GX016015.MP4.gyroflow trace, preset %s" % (B, args.border, preset),
                       "frames_per_step": units_per_step if args.scaling == "strong" else B // nb, "output_frames_per_step_all_gpus": units_per_step,
                       "sharding": "contiguous frame ranges (frame_shard), no collective",
                       "cpu_affinity": sorted(pinned_cpus) if pinned_cpus else None,     # every thread of rank 0 (dist.pin_to_gpu)
                       "gpus_visible": n_dev},
            "checksum": checksum,
        }
        if nb > 1:
            out["input_frames_per_s"] = out["value"] * nb
        if pipe is None:
            # one launch of the dominant kernel processes `units` output frames: average over the timed launches
            units_per_launch = sum(n // nb for st in steps[W:W + K] for n, _ in st) / max(n_launch, 1)
            bytes_per_launch = unit_bytes * units_per_launch
            achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
            kname = {"direct": "k_undistort_rotate", "rl": "k_undistort_rotate_rl"}.get(os.environ.get("GF_K1", "rp"), "k_undistort_rotate_rp")
            if nb > 1:
                kname = "k_undistort_rotate_blend_rp"
            if cfg == "c4_fused":
                kname = "k_undistort_rotate_superview"
            if cfg == "c4":
                kname += " + k_superview_cubic"
            traffic = tsrc = valu = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    ent = json.load(f).get("%s_b%d" % (cfg, B))
                if ent:      # measured in separate PMC passes of this command (tools/collect_r03_pmc.sh); stamped with the commit they were taken at
                    traffic, tsrc = ent["hbm_bytes_per_launch"], {"commit": ent.get("commit"), "source": ent.get("source")}
                    valu = ent.get("valu")
            # `achieved` / `peak` / `frac` are the HBM figures BASELINE's metric asks for.  They are NOT what limits these kernels: the
            # counters say VALU instruction issue (roofline.valu.frac = share of all SIMD cycles with a VALU instruction issuing),
            # with the vector-memory path second -- so `bound` names that unit, and the HBM fraction is reported beside it.
            out["roofline"] = {"bound": "valu" if valu else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_from": tsrc, "kernel": kname,
                               "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                               "algorithmic_bytes_per_output_frame": unit_bytes, "valu": valu}
        if world == 1 and pipe is None:
            n_cpu = args.cpu_frames
            if n_cpu < 0:      # >= 100 timed frames after 10 warm-up frames (about 10-30 s of CPU work)
                n_cpu = {"c1": 200, "c2": 200, "c3": 100, "c4": 100, "c4_fused": 100, "c5": 25, "c5_noblend": 100}[cfg]
            have_oracle = os.path.isdir(os.path.join(ROOT, "oracle"))
            if n_cpu > 0 and not have_oracle:       # the CPU legs are the oracle's; the GPU line does not depend on them
                out["cpu_baseline"] = {"error": "oracle/ is not present: no CPU baseline and no touched-footprint figure"}
            if n_cpu > 0 and have_oracle:
                threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
                out["cpu_baseline"] = cpu_baseline(cfg, cal, quats, ring, n_cpu, threads)
                if cfg not in STRETCH and nb == 1:
                    # touched-footprint variant of the roofline figure (the honest lower one): unique source bytes + output bytes
                    sample = [int(v) for v in idx[W * B:(W + 1) * B:max(1, B // 4)]]
                    tb = touched_source_bytes(cfg, cal, quats, sample)
                    t_bytes = (tb + ow * oh * 3 * (1 if dt == "u8" else 2)) * units_per_launch
                    out["roofline"]["touched"] = {"bytes_per_launch": int(t_bytes), "achieved": t_bytes / (kernel_ms * 1e-3) / 1e9,
                                                  "frac": t_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                  "sample": "unique source pixels under the taps, CPU oracle maps of %d frames of the first timed launch" % len(sample)}
            if not args.no_pcie and cfg in ("c1", "c2", "c3"):
                try:
                    out["pcie_pipeline"] = pcie_pipeline(cfg, dev)
                except Exception as e:          # an auxiliary measurement: report, do not lose the line
                    out["pcie_pipeline"] = {"error": repr(e)}
        print(json.dumps(out), file=line_out, flush=True)
    gdist.shutdown()


if __name__ == "__main__":
    main()
