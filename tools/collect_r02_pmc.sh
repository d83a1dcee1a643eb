#!/bin/bash
# Round-2 evidence run (part B): rocprofv3 counter passes (tools/pmc.sh: one --pmc run per pass, no tracing) of bench.py.
set -u
for c in "$@"; do
  tools/pmc.sh r02_$c -- python3 bench.py --config $c --cpu-frames 0 --no-pcie --steps 3 --warmup 1 || exit 1
done
