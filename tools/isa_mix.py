#!/usr/bin/env python
"""Static instruction mix per kernel of a gfx950 assembly file (hipcc -S --cuda-device-only): MFMA / VALU / SALU / LDS / VMEM /
waits / branches, whole kernel and per basic block above a size threshold (the tile loop's blocks are the large ones).
    python tools/isa_mix.py file.s [substring of the kernel name] [--blocks]"""
import collections
import re
import sys


def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")): return "vmem"
    return "other"


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    blocks = "--blocks" in sys.argv
    name, cur, per_block, label = None, None, None, None
    for line in open(path):
        m = re.match(r"^(\w+):\s+; @", line)
        if m:
            name = m.group(1)
            cur, per_block, label = collections.Counter(), collections.OrderedDict(), "entry"
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            if pat in name:
                print(name[:120])
                print("   total", dict(cur))
                if blocks:
                    for lb, c in per_block.items():
                        if sum(c.values()) >= 40:
                            print("   %-12s %4d" % (lb, sum(c.values())), dict(c))
            name = None
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            label = m.group(1)
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        k = klass(t.split()[0])
        cur[k] += 1
        per_block.setdefault(label, collections.Counter())[k] += 1


if __name__ == "__main__":
    main()
