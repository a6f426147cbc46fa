#!/bin/bash
# A/B of environment switches on the DROP-IN call's rate (bench.py's api_tuple object), one box, the sequence twice.
# usage: scripts/ab_api.sh <tag> "ENV=.." ...   ("-" = default)
R=$GRAFT_REPO_ROOT
T=$1; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
for rep in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    [ "$e" = "-" ] && e=""
    env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-train --steps 8 > $O/api_${i}_$rep.json 2> $O/api_${i}_$rep.err || { tail -5 $O/api_${i}_$rep.err; exit 1; }
    python - "$O/api_${i}_$rep.json" "$e" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
a = d["api_tuple"]
print(f"{sys.argv[2] or 'default':40s} device path {d['value']:6.2f} scenes/s; evaluate_scene(20-tuple) serial {a['serial']['ms_per_scene']:7.3f} ms, prefetched {a['prefetched']['ms_per_scene']:7.3f} ms ({a['steps']} steps)", flush=True)
PY
  done
done
