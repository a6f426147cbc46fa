#!/usr/bin/env python3
"""Memory-side bytes of ONE 512->512 layer from a conv_pmc_summary.json (scripts/pmc_summarize.py): a layer is `chunks` launches
of conv_phase1_dma_kernel + as many of conv_phase2_kernel; 2 x FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE counts 64 B per 128-B
request on gfx950, MI355X_MICROARCH.md) per launch x launches per layer.  usage: pmc_conv_layer.py summary.json [chunks | bench.json of
the counted run: its roofline_conv.ceilings.chunks]   (17 with 8192-row chunks, 11 since the three-round chunks of round 5)"""
import json
import sys

d = json.load(open(sys.argv[1]))
chunks = 11
if len(sys.argv) > 2:
    a = sys.argv[2]
    chunks = int(a) if a.isdigit() else int(json.load(open(a))["roofline_conv"]["ceilings"]["chunks"])
out = {"chunks_per_layer": chunks}
tot = 0.0
for name, c in d.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    per = (2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024
    out[name] = {"fetch_kib_per_launch": c["FETCH_SIZE"]["mean"], "write_kib_per_launch": c["WRITE_SIZE"]["mean"],
                 "memory_side_bytes_per_launch": per, "memory_side_bytes_per_layer": per * chunks, "launches_seen": c["FETCH_SIZE"]["n"]}
    if "conv_phase1_dma" in name or "conv_phase2" in name:
        tot += per * chunks
out["memory_side_bytes_per_layer_phase1_plus_phase2"] = tot
print(json.dumps(out, indent=1))
