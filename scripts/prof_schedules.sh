#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in alternate split; do
  O=$R/gpurun_out/prof_$mode
  mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-train --steps 12 --schedule $mode > $O/bench.json 2> $O/log.txt
  find $O -name "*_kernel_trace.csv" -delete
done
echo done
