#!/bin/bash
# A/B of scheduling / kernel switches on ONE box: bench.py (default scene, device API, no extras) under each environment, the
# sequence repeated so that box drift shows.  usage: scripts/ab_bench.sh <tag> <steps> "ENV1=.. ENV2=.." "ENV=.." ...   ("-" = default)
R=$GRAFT_REPO_ROOT
T=$1; K=$2; shift 2
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
for rep in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    [ "$e" = "-" ] && e=""
    env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-train --api device --steps $K > $O/ab_${i}_$rep.json 2> $O/ab_${i}_$rep.err || { tail -5 $O/ab_${i}_$rep.err; exit 1; }
    python - "$O/ab_${i}_$rep.json" "$e" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2] or 'default':60s} {d['value']:7.2f} scenes/s  {d['ms_per_step']:7.3f} ms  pooling {d['roofline']['avg_launch_ms']:.4f} ms  conv layer {d['roofline_conv']['avg_layer_ms']:.3f} ms", flush=True)
PY
  done
done
