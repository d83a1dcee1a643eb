#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of tools/render_bench.py: busy time of the H2D copies, the
D2H copies and the fused kernel inside the timed render, and how much of it overlaps (interval arithmetic on the trace's
start/end timestamps).  usage: overlap_summary.py <trace dir> [kernel substring]"""
import csv, glob, os, sys

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_undistort_rotate"


def load(pattern, pick):
    out = []
    for f in glob.glob(os.path.join(root, "**", pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                v = pick(r)
                if v:
                    out.append(v)
    return out


def span(r):
    return int(r["Start_Timestamp"]), int(r["End_Timestamp"])


kernels = load("*kernel_trace.csv", lambda r: span(r) if kern in r["Kernel_Name"] else None)
copies = load("*memory_copy_trace.csv", lambda r: (r["Direction"],) + span(r) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),))
big = [c for c in copies if c[3] >= 100000]               # frame copies only (>= 0.1 ms; transforms and tables are tiny)
h2d = [(a, b) for d, a, b, n in big if "HOST_TO_DEVICE" in d.upper() or d.upper() == "H2D"]
d2h = [(a, b) for d, a, b, n in big if "DEVICE_TO_HOST" in d.upper() or d.upper() == "D2H"]


def union(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def inter(x, y):
    """total length of the intersection of two interval unions"""
    return union(x) + union(y) - union(x + y)


# keep the second half of the run (the timed render; the first is render_bench's warm-up)
allv = kernels + h2d + d2h
t_mid = (min(a for a, _ in allv) + max(b for _, b in allv)) // 2
kernels, h2d, d2h = ([v for v in s if v[0] >= t_mid] for s in (kernels, h2d, d2h))
allv = kernels + h2d + d2h
wall = max(b for _, b in allv) - min(a for a, _ in allv)
ms = lambda ns: ns / 1e6
print("timed render: %.1f ms wall between first and last traced operation" % ms(wall))
print("  H2D frame copies : %5d copies, busy %.1f ms (%.0f%% of wall)" % (len(h2d), ms(union(h2d)), 100 * union(h2d) / wall))
print("  D2H frame copies : %5d copies, busy %.1f ms (%.0f%% of wall)%s" % (len(d2h), ms(union(d2h)), 100 * union(d2h) / wall,
      "" if d2h else "   [this rocprofv3 build reports no device-to-host records for copies into pinned memory]"))
print("  %s : %5d launches, busy %.1f ms (%.0f%% of wall)" % (kern, len(kernels), ms(union(kernels)), 100 * union(kernels) / wall))
print("  H2D and D2H at the same time       : %.1f ms (%.0f%% of the H2D time)" % (ms(inter(h2d, d2h)), 100 * inter(h2d, d2h) / max(union(h2d), 1)))
print("  kernel while a copy is in flight   : %.1f ms (%.0f%% of the kernel time)" % (ms(inter(kernels, h2d + d2h)), 100 * inter(kernels, h2d + d2h) / max(union(kernels), 1)))
print("  sum of the three busy times / wall : %.2f (1.0 = no overlap at all, 3.0 = everything concurrent)" % ((union(h2d) + union(d2h) + union(kernels)) / wall))
