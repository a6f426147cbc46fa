#!/usr/bin/env python3
"""Timeline of the two-stream schedule from a rocprofv3 --kernel-trace of bench.py: for each of the last scenes, the main queue's
milestones (student start / end, affinity start, last pooling end, classify end) and every run of consecutive look-ahead-queue
kernels, all in ms relative to the student's first convolution.  usage: trace_timeline.py <dir> [--scenes 3]"""
import csv
import glob
import os
import sys
from collections import Counter

root = sys.argv[1]
nsc = int(sys.argv[sys.argv.index("--scenes") + 1]) if "--scenes" in sys.argv else 3
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()


def short(n):
    for pre in ("void ", "(anonymous namespace)::"):
        if n.startswith(pre):
            n = n[len(pre):]
    if n.startswith("_ZN12_GLOBAL__N_1"):
        n = n[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
    return n.split("(")[0].split("<")[0][:34]


qcount = Counter(q for _, _, q, n in rows if "conv_phase1" in n)
main_q = qcount.most_common(1)[0][0]
aff = [i for i, r in enumerate(rows) if "affinity_block_kernel" in r[3] and r[2] == main_q]
for k in range(len(aff) - nsc - 1, len(aff) - 1):
    a0, a1 = rows[aff[k]][0], rows[aff[k + 1]][0]                     # from one affinity start to the next
    seg = [r for r in rows if a0 <= r[0] < a1]
    conv = [r for r in seg if "conv_phase" in r[3]]
    t0 = conv[0][0]                                                  # the NEXT scene's student starts inside this window
    ms = lambda t: (t - t0) / 1e6
    pools = [r for r in seg if "cs_pool_kernel" in r[3]]
    print(f"--- window of {ms(a1) - ms(a0):.2f} ms: affinity(i) at {ms(a0):+.2f}, pooling(i) {ms(pools[0][0]):+.2f} .. {ms(pools[-1][1]):+.2f}, "
          f"student(i+1) 0.00 .. {ms(conv[-1][1]):+.2f}, affinity(i+1) at {ms(a1):+.2f}")
    side = [r for r in seg if r[2] != main_q]
    # runs of look-ahead kernels separated by more than 0.2 ms
    runs, cur = [], []
    for r in side:
        if cur and r[0] - cur[-1][1] > 200_000:
            runs.append(cur)
            cur = []
        cur.append(r)
    if cur:
        runs.append(cur)
    for run in runs:
        busy = sum(e - s for s, e, _, _ in run) / 1e6
        names = Counter(short(r[3]) for r in run)
        top = ", ".join(f"{n} x{c}" for n, c in names.most_common(4))
        print(f"    look-ahead {ms(run[0][0]):+8.2f} .. {ms(run[-1][1]):+8.2f} ms: {len(run):3d} kernels, busy {busy:6.2f} ms  [{short(run[0][3])} ... {short(run[-1][3])}]  {top}")
    big = sorted(side, key=lambda r: r[0] - r[1])[:6]
    for s, e, q, n in sorted(big):
        print(f"        {short(n):36s} {ms(s):+8.2f} ms, {(e - s) / 1e3:8.1f} us")
