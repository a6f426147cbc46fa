#!/usr/bin/env python3
"""Where is the GPU idle, per queue, in a rocprofv3 --kernel-trace of bench.py?  Reads *_kernel_trace.csv, keeps the last
`--scenes` scenes (a scene = from one affinity kernel to the next), and prints per queue: busy time, idle time, and the
largest idle gaps with the kernels on either side.  usage: trace_gaps.py <dir> [--scenes 4]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
nsc = int(sys.argv[sys.argv.index("--scenes") + 1]) if "--scenes" in sys.argv else 4
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
    return n.split("(")[0][:40]


marks = [s for s, e, q, n in rows if "affinity_block_kernel" in n or "affinity_cs_kernelILb0" in n]
if len(marks) < nsc + 2:
    sys.exit("too few scenes in the trace")
t0, t1 = marks[-nsc - 1], marks[-1]
sel = [r for r in rows if t0 <= r[0] < t1]
span = (t1 - t0) / 1e6
print(f"{nsc} scenes, {span / nsc:.3f} ms per scene ({len(sel)} kernels)")
byq = defaultdict(list)
for r in sel:
    byq[r[2]].append(r)
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in ks) / 1e6
    gaps = []
    for a, b in zip(ks[:-1], ks[1:]):
        g = b[0] - a[1]
        if g > 0:
            gaps.append((g, short(a[3]), short(b[3])))
    idle = sum(g for g, _, _ in gaps) / 1e6
    print(f"queue {q}: {len(ks)} kernels, busy {busy / nsc:.3f} ms per scene, idle between its kernels {idle / nsc:.3f} ms per scene")
    agg = defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        agg[(a, b)][0] += g
        agg[(a, b)][1] += 1
    for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"    {g / 1e3 / nsc:8.1f} us per scene in {c / nsc:6.1f} gaps (avg {g / c / 1e3:6.1f} us): {a} -> {b}")
# union busy time over all queues
ev = sorted([(s, 1) for s, e, _, _ in sel] + [(e, -1) for s, e, _, _ in sel])
depth, last, any_busy = 0, t0, 0
for t, d in ev:
    if depth > 0:
        any_busy += t - last
    depth += d
    last = t
print(f"GPU busy with at least one kernel: {any_busy / 1e6 / nsc:.3f} ms per scene; idle {span / nsc - any_busy / 1e6 / nsc:.3f} ms per scene")

# per scene: when does the look-ahead queue work, relative to the student's first convolution launch on the main queue?
main_q = max(byq, key=lambda q: len(byq[q]))
side_qs = [q for q in byq if q != main_q]
convs = [s for s, e, q, n in sel if q == main_q and "conv_phase1" in n]
l2 = [(s, e) for s, e, q, n in sel if q == main_q and "l2norm_rows" in n]
aff = [(s, e) for s, e, q, n in sel if q == main_q and "affinity_block" in n]
print("per scene (ms after the student's l2norm END): affinity start | look-ahead queue: first kernel start, last kernel end (relative to the PREVIOUS l2norm end)")
for i in range(1, len(l2)):
    t_l2 = l2[i][1]
    a = [s for s, e in aff if s >= t_l2]
    side = [(s, e, n) for s, e, q, n in sel if q in side_qs and l2[i - 1][1] <= s < t_l2]
    if not a or not side:
        continue
    print(f"  scene {i}: affinity starts {(a[0] - t_l2) / 1e6:6.3f} ms after l2norm; look-ahead kernels {len(side):4d}: first {(side[0][0] - l2[i - 1][1]) / 1e6:7.3f} ms, "
          f"last ends {(side[-1][1] - l2[i - 1][1]) / 1e6:7.3f} ms after the previous l2norm (this l2norm ends at {(t_l2 - l2[i - 1][1]) / 1e6:7.3f}); last look-ahead kernel: {short(side[-1][2])}")
