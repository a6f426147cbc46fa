#!/bin/bash
# kernel stats of a short bench run (tuning aid).  usage: scripts/prof_stats.sh <tag> [bench args]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-stats}; shift || true
O=$R/gpurun_out/$T
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-train --steps 8 "$@" > $O/bench.json 2> $O/log.txt
find $O -name "*_kernel_trace.csv" -delete
echo done
