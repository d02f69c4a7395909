#!/usr/bin/env python3
"""Where do a kernel's code bytes come from?  Attributes the instructions of one kernel to source functions through the line table.

  hipcc <the library's flags> -gline-tables-only --cuda-device-only -DCTC_KERNEL_GROUP=<g> -c ctcdecode_amd/csrc/decode_kernels.hip -o g.co
  clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=g.co --output=g.elf
  llvm-objdump -d -l g.elf > g.s
  python tools/code_map.py g.s 'ILi0ELi1ELi0ELb0ELi1024ELi0ELb0E' [--lines beam_core.h:1240-1900:20]

(tools/code_map.sh does all of it.)  Inlined code is attributed to the function whose source line the instruction carries."""
import collections
import os
import re
import sys


def function_starts(path):
    starts = []
    pat = re.compile(r"^\s{0,4}(?:template\s*<[^>]*>\s*)?(?:CTC_HD|__device__|__global__|static|inline|constexpr)[^;]*\(")
    with open(path, encoding="utf-8") as f:
        lines = f.readlines()
    for i, l in enumerate(lines, 1):
        if pat.match(l) and not l.strip().startswith("//"):
            m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*\(", l.split("//")[0])
            # (the name before the first parenthesis that is not a keyword-ish macro)
            names = re.findall(r"([A-Za-z_][A-Za-z0-9_:]*)\s*\(", l.split("//")[0])
            names = [n for n in names if n not in ("CTC_HD", "__launch_bounds__", "if", "for", "while", "sizeof", "decltype")]
            if names:
                starts.append((i, names[0]))
    return starts


def main():
    asm, key = sys.argv[1], sys.argv[2]
    fine = None
    if "--lines" in sys.argv:
        f, rng, step = sys.argv[sys.argv.index("--lines") + 1].split(":")
        a, b = rng.split("-")
        fine = (f, int(a), int(b), int(step))
    per_line = collections.Counter()
    cur = None
    inside = False
    prev_addr = None
    prev_loc = None
    total = 0
    for l in open(asm, encoding="utf-8", errors="replace"):
        if l.startswith("; ") and l.rstrip().endswith("():"):
            inside = key in l
            prev_addr = None
            continue
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            inside = key in m.group(1)
            prev_addr = None
            continue
        if not inside:
            continue
        if l.startswith("; /"):
            mm = re.match(r"; (.*):(\d+)", l.strip())
            if mm:
                cur = (os.path.basename(mm.group(1)), int(mm.group(2)))
            continue
        mm = re.search(r"// ([0-9A-F]{12}):", l)
        if mm:
            addr = int(mm.group(1), 16)
            if prev_addr is not None:
                per_line[prev_loc] += addr - prev_addr
                total += addr - prev_addr
            prev_addr, prev_loc = addr, cur
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ctcdecode_amd", "csrc")
    starts = {}
    per_fn = collections.Counter()
    for (f, line), n in per_line.items():
        if f not in starts:
            p = os.path.join(csrc, f)
            starts[f] = function_starts(p) if os.path.exists(p) else []
        name = "?"
        for s, nm in starts[f]:
            if s <= line:
                name = nm
            else:
                break
        per_fn[(f, name)] += n
    print("kernel %s: %d code bytes attributed" % (key, total))
    for (f, name), n in per_fn.most_common(60):
        print("%8d  %5.1f%%  %s:%s" % (n, 100.0 * n / max(total, 1), f, name))
    if fine:
        f, a, b, step = fine
        print("--- %s lines %d-%d" % (f, a, b))
        for lo in range(a, b, step):
            n = sum(v for (ff, ln), v in per_line.items() if ff == f and lo <= ln < lo + step)
            if n:
                print("%8d  %s:%d-%d" % (n, f, lo, lo + step - 1))


if __name__ == "__main__":
    main()
