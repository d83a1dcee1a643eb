#!/bin/bash
# VALU / scalar / LDS instruction counts of the LB experiment against the shipped kernel (kbench, C3, 128 frames)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/pmc_lb; mkdir -p $O
P="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
GF_K1=rl timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $O/lb -- $R/tools/_build/kbench --lib $R/tools/_build/lib_lb_3.so --frames 128 --iters 3 > $O/lb.log 2>&1 || { echo lb failed; tail -3 $O/lb.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $O/rp -- $R/tools/_build/kbench --lib $R/gyroflow_amd/libgfremap.so --frames 128 --iters 3 > $O/rp.log 2>&1 || { echo rp failed; tail -3 $O/rp.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for tag in ("rp", "lb"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_undistort_rotate" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    px = 128*3840*2160
    print(tag, {k: round(sum(v)/len(v)/px*64, 2) for k, v in acc.items() if k.startswith("SQ_INSTS")}, "per pixel;",
          "VALU busy", round(sum(acc["SQ_ACTIVE_INST_VALU"])/len(acc["SQ_ACTIVE_INST_VALU"])*4/(1024*sum(acc["GRBM_GUI_ACTIVE"])/len(acc["GRBM_GUI_ACTIVE"])/8), 3))
PY
