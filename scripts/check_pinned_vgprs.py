#!/usr/bin/env python3
"""Build-time guard for pool_mfma_persist_kernel: the kernel parks two asynchronous results in v200 and
v[202:203] across loop steps -- registers named in inline asm, which a clobber list does not reserve.  The guard reads the
kernel's ISA (hipcc -S) and fails when the COMPILER's own allocation comes near them: any vector register >= 192 other
than v200, v202, v203 in an instruction of that kernel.
usage: check_pinned_vgprs.py <file.s> <kernel-name-substring>"""
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2]
pinned = {200, 202, 203}
bad, seen, inside, n_kernels = set(), set(), False, 0
for line in text.splitlines():
    if re.match(r"^_Z\w*:", line):
        inside = want in line
        n_kernels += inside
        continue
    # a kernel's code ends at its .Lfunc_end label (NOT at the first s_endpgm: an early-exit block would end the scan early)
    if re.match(r"^\.Lfunc_end\d+:", line) or line.lstrip().startswith(".end_amdhsa_kernel"):
        inside = False
        continue
    if not inside:
        continue
    code = line.split(";")[0]
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
        seen.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", code):
        seen.add(int(a))
if not n_kernels:
    sys.exit(f"check_pinned_vgprs: no kernel matching '{want}' in {sys.argv[1]}")
# the descriptor must RESERVE the pinned registers: .amdhsa_next_free_vgpr >= 204 for every instantiation
nfv = [int(v) for name, v in re.findall(r"\.amdhsa_kernel (\S+)(?:.|\n)*?\.amdhsa_next_free_vgpr (\d+)", text) if want in name]
if len(nfv) != n_kernels or min(nfv) < 204:
    sys.exit(f"{want}: .amdhsa_next_free_vgpr = {nfv} for {n_kernels} instantiation(s): v200..v203 are not inside the allocation")
bad = {r for r in seen if r >= 192 and r not in pinned}
top = max((r for r in seen if r not in pinned), default=-1)
print(f"{want}: compiler-allocated VGPRs up to v{top} in {n_kernels} instantiation(s); pinned v200, v202, v203; next_free_vgpr {nfv}")
if bad:
    sys.exit(f"{want}: VGPRs {sorted(bad)} are allocated next to the pinned registers v200..v203 -- the asynchronous tile claim is no longer safe")
