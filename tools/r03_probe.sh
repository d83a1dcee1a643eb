#!/bin/bash
# round 3, first GPU call: LDS microbenchmark, available counters, PC-sampling probe on kbench
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_probe
mkdir -p $O
timeout -k 10 120 $R/tools/_build/ldsbench > $O/ldsbench.txt 2>&1 || { echo "ldsbench failed"; tail -5 $O/ldsbench.txt; exit 1; }
cat $O/ldsbench.txt
timeout -k 10 60 rocprofv3 -L > $O/avail.txt 2>&1 || echo "list-avail rc $?"
grep -c . $O/avail.txt
grep -i -A12 "pc.sampl" $O/avail.txt | head -40
cd /tmp
timeout -s KILL 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 --output-format csv -d $O/pcs -- $R/tools/_build/kbench --lib $R/gyroflow_amd/libgfremap.so --frames 64 --iters 3 > $O/pcs.log 2>&1
echo "pc sampling (stochastic) rc $?"; tail -5 $O/pcs.log; ls -R $O/pcs 2>/dev/null | head -20
