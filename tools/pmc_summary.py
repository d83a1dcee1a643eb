#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_TAG (tools/pmc.sh): per-dispatch average of every counter for the fused kernel."""
import csv, glob, os, sys, collections
tag = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_undistort_rotate"
root = os.path.join("gpurun_out", "pmc_" + tag)
vals = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = (sum(v) / len(v), len(v))
for k, (v, n) in vals.items():
    print("%-34s %18.1f   (avg of %d dispatches)" % (k, v, n))
