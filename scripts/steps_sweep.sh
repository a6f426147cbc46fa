#!/bin/bash
# does the headline depend on how long the timed region is?  bench.py (device API, no extras) at several --steps / --warmup, one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for rep in 1 2; do
 for sw in "6 2" "24 2" "24 8" "64 8" "6 2"; do
  set -- $sw
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-train --api device --steps $1 --warmup $2 > $O/s_$1_$2_$rep.json 2> $O/s_$1_$2_$rep.err || exit 1
  python - "$O/s_$1_$2_$rep.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"steps {d['steps']:3d} warmup {d['warmup']:2d}: {d['value']:7.2f} scenes/s  {d['ms_per_step']:7.3f} ms  pooling {d['roofline']['avg_launch_ms']:.4f} ms (frac {d['roofline']['frac']:.4f})  conv layer {d['roofline_conv']['avg_layer_ms']:.3f} ms", flush=True)
PY
 done
done
