#!/bin/bash
# kernel stats of the training step.  usage: scripts/prof_train.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-train}
O=$R/gpurun_out/$T
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python3 $R/scripts/bench_train.py 4 > $O/bench_train.log 2> $O/log.txt
find $O -name "*_kernel_trace.csv" -delete
echo done
