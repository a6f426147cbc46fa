#!/usr/bin/env python3
"""Per-kernel table of a `rocprofv3 --kernel-trace` run of scripts/bench_train.py from its rocpd database (the default output format of
ROCm 7.2's rocprofv3).  usage: train_kernel_table.py <results.db> <steps incl. the untimed first> [rows]"""
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end - start) / 1e6, avg(end - start) / 1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"kernel time {tot / steps:.2f} ms per step ({steps} steps)")
for r in rows[:top]:
    print(f"{r[2] / steps:8.3f} ms/step {r[1] / steps:7.1f} calls/step {r[3]:9.1f} us  {r[0][:110]}")
