#!/usr/bin/env python3
"""Register / LDS / spill figures of the kernels in a device assembly file (hipcc --cuda-device-only -S), one line each.
usage: kernel_resources.py file.s [name-substring ...]   (library kernels of rocPRIM are skipped unless named)"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2:]
    meta = txt[txt.index("amdhsa.kernels:"):]
    for blk in meta.split("\n  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if "rocprim" in name and not want:
            continue
        if want and not any(w in name for w in want):
            continue
        d = demangle(name)
        d = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0]
        print(f"{d:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} "
              f"lds {g('group_segment_fixed_size'):>6s} agpr {blk.split()[0]}")


if __name__ == "__main__":
    main()
