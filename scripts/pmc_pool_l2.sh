#!/bin/bash
# L2 / memory-side counters of one pooling variant of scripts/bench_pool.py (separate passes: TCC has 4 slots, FETCH_SIZE 3, WRITE_SIZE 2)
# usage: scripts/pmc_pool_l2.sh <variant-word> <out-dir-under-gpurun_out> [kernel-substring]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-mfma64}
O=$R/gpurun_out/${2:-pmc_pool_l2}
K=${3:-pool}
mkdir -p $O
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/tcc -o tcc --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/write.log 2>&1
python3 $R/scripts/pmc_summarize.py $O "$K" > $O/summary.json
cat $O/summary.json
