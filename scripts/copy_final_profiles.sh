#!/bin/bash
# The judged copies of a scripts/final_round.sh run: gpurun_out/<tag>/... -> profiles/<prefix>_*  (gpurun_out/ is scratch).
# usage: scripts/copy_final_profiles.sh r5/final4 r05
set -e
F=gpurun_out/$1; P=profiles/$2
for c in default_run config_V config_M config_P; do cp $F/bench_$c.json ${P}_bench_$c.json; done
cp $F/prof/bench_under_rocprof.json ${P}_bench_under_rocprof.json
cp $F/prof/conv_pmc_summary.json ${P}_conv_pmc_summary.json
cp $F/prof/conv_pmc_per_layer.json ${P}_conv_pmc_per_layer.json
cp $F/prof/pool_pmc_summary.json ${P}_pool_pmc_summary.json
cp $F/prof/small_pmc_summary.json ${P}_small_pmc_summary.json
cp $F/prof/stats/p_kernel_stats.csv ${P}_kernel_stats.csv
python3 - "$F" "$2" <<'PY'
import json, sys
f, pre = sys.argv[1], sys.argv[2]
s = json.load(open(f"{f}/prof/pool_pmc_summary.json"))
d = s[[k for k in s if "cs_pool_kernelILb0" in k][0]]
p = json.load(open("profiles/pool_pmc.json"))
e = p["cs_pool_kernel"]
e.update(fetch_kib=d["FETCH_SIZE"]["mean"], write_kib=d["WRITE_SIZE"]["mean"], launches_averaged=d["FETCH_SIZE"]["n"],
         tcc_req=d["TCC_REQ_sum"]["mean"], tcc_hit=d["TCC_HIT_sum"]["mean"], tcc_miss=d["TCC_MISS_sum"]["mean"],
         commit=f"tree of scripts/final_round.sh {f.split('gpurun_out/')[-1]}",
         profile=f"{pre}_pool_pmc_summary.json (scripts/profile_bench.sh via scripts/final_round.sh: rocprofv3 --pmc passes over `bench.py --scenes 1 --streams 1`, whose one scene has Nv=133933)")
json.dump(p, open("profiles/pool_pmc.json", "w"), indent=1)
PY
echo copied
