#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_rotation.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r03/rot_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r03/rot_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/stall_pmc.sh base -- python3 bench.py --cpu-frames 0 --no-pcie --steps 3 --warmup 1 || exit 1
python3 tools/pmc_summary.py base k_undistort_rotate 2>/dev/null | head -0
for f in gpurun_out/stall_base/pass*/; do :; done
python3 - <<'PY'
import csv, glob, collections
vals = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/stall_base/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_undistort_rotate" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = (sum(v)/len(v), len(v))
with open("gpurun_out/r03/stall_base.txt", "w") as o:
    for k, (v, n) in vals.items():
        o.write("%-34s %18.1f   (avg of %d dispatches)\n" % (k, v, n))
print(open("gpurun_out/r03/stall_base.txt").read())
PY
