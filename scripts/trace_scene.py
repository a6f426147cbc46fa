#!/usr/bin/env python3
"""Per-scene view of a rocprofv3 --kernel-trace of bench.py (timed region, T = 19): for a few scenes in the middle of the trace, the main
queue's busy time, every gap above 20 us with the kernels on either side, the sum of the small gaps, the span and busy time of the student's
convolution kernels, and when the other queues (the look-ahead) ran.  A scene = from one affinity kernel to the next.
usage: rocprofv3 --kernel-trace -d <dir> -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-train --api device --steps 10
       python scripts/trace_scene.py <dir> [first scene index, default 7] [scenes, default 3]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 7
count = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()


def short(n):
    for pre in ("void ", "(anonymous namespace)::"):
        if n.startswith(pre):
            n = n[len(pre):]
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
    return n.split("(")[0][:36]


aff = [i for i, r in enumerate(rows) if "affinity_cs_kernelILb0" in r[3]]
print(f"{len(rows)} kernels, {len(aff)} scenes in the trace")
for sidx in range(first, min(first + count, len(aff) - 1)):
    a0, a1 = rows[aff[sidx]][0], rows[aff[sidx + 1]][0]
    sel = [r for r in rows if a0 <= r[0] < a1]
    mainq = collections.Counter(r[2] for r in sel).most_common(1)[0][0]
    mk = [r for r in sel if r[2] == mainq]
    print(f"scene {sidx}: {(a1 - a0) / 1e6:.3f} ms; main queue {mainq}: {len(mk)} kernels, busy {sum(e - s for s, e, _, _ in mk) / 1e6:.3f} ms")
    for x, y in zip(mk[:-1], mk[1:]):
        g = y[0] - x[1]
        if g > 20000:
            print(f"    gap {g / 1e3:7.1f} us: {short(x[3])} -> {short(y[3])} at {(x[1] - a0) / 1e6:.2f} ms")
    small = sum(max(0, y[0] - x[1]) for x, y in zip(mk[:-1], mk[1:]) if y[0] - x[1] <= 20000)
    print(f"    sum of the gaps below 20 us: {small / 1e6:.3f} ms over {len(mk) - 1} kernel boundaries")
    convs = [r for r in mk if "conv_phase" in r[3]]
    if convs:
        print(f"    student: first convolution kernel at {(convs[0][0] - a0) / 1e6:.2f} ms, last ends at {(convs[-1][1] - a0) / 1e6:.2f}; "
              f"{len(convs)} kernels busy {sum(e - s for s, e, _, _ in convs) / 1e6:.3f} ms")
    oth = [r for r in sel if r[2] != mainq]
    if oth:
        print(f"    other queues (look-ahead): {len(oth)} kernels, busy {sum(e - s for s, e, _, _ in oth) / 1e6:.3f} ms, from "
              f"{(oth[0][0] - a0) / 1e6:.2f} to {(oth[-1][1] - a0) / 1e6:.2f} ms")
        top = collections.Counter()
        for s_, e_, _, n_ in oth:
            top[short(n_)] += e_ - s_
        print("    look-ahead kernels by time: " + ", ".join(f"{k} {v / 1e3:.0f} us" for k, v in top.most_common(12)))
