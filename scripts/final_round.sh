#!/bin/bash
# Round-end measurements on the GPU box: the GPU test suite, bench.py on every BASELINE configuration, the default command's
# profile (kernel stats + PMC passes).  usage: scripts/final_round.sh <tag>  -> gpurun_out/<tag>/...
set -e
R=$GRAFT_REPO_ROOT
T=${1:-r05}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python bench.py > $O/bench_default_run.json 2> $O/bench_default_run.err
timeout -k 10 400 python bench.py --config V --no-cpu-baseline --no-train --api device > $O/bench_config_V.json 2> $O/bench_config_V.err
timeout -k 10 600 python bench.py --config M --no-cpu-baseline --no-train --api device > $O/bench_config_M.json 2> $O/bench_config_M.err
timeout -k 10 300 python bench.py --config P --no-cpu-baseline --no-train --api device > $O/bench_config_P.json 2> $O/bench_config_P.err
bash scripts/profile_bench.sh $T/prof > $O/prof.log 2>&1
echo final_round done
