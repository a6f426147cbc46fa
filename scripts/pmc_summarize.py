#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch (+ launches).
usage: scripts/pmc_summarize.py <dir> [kernel-substring ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
wants = sys.argv[2:] if len(sys.argv) > 2 else ["pool"]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if any(w in k for w in wants):
            name = k
            for pre in ("void ", "(anonymous namespace)::"):
                if name.startswith(pre):
                    name = name[len(pre):]
            acc[name.split("(")[0][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
print(json.dumps(out, indent=1))
