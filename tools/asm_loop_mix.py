#!/usr/bin/env python3
"""Static instruction mix of the per-frame loop of one decode-kernel instantiation.
    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S ctcdecode_amd.hip -o kern.s
    python tools/asm_loop_mix.py kern.s ILb0ELb0ELi1E [loop.s]
"""
import collections
import re
import sys

L = open(sys.argv[1]).read().split("\n")
tag = sys.argv[2]
start = [i for i, l in enumerate(L) if re.match(r"^_Z\w*ctc_beam_decode_kernel" + tag + r"\w*:", l)][0]
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
K = L[start:end + 1]
lab = {}
for i, l in enumerate(K):
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        lab[m.group(1)] = i
lo = hi = 0
for i, l in enumerate(K):
    m = re.search(r"(s_cbranch\w*|s_branch)\s+(\.LBB[0-9_]+)", l)
    if m and m.group(2) in lab and lab[m.group(2)] < i and i - lab[m.group(2)] > hi - lo:
        lo, hi = lab[m.group(2)], i  # the longest backward branch closes the frame loop
loop = K[lo:hi + 1]
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(loop))
c = collections.Counter()
for l in loop:
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if m:
        c[m.group(1)] += 1
print("kernel lines", len(K), "frame loop", lo, "-", hi, "instructions", sum(c.values()))
for k, v in c.most_common(30):
    print("  %-28s %d" % (k, v))
