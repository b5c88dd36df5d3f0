#!/usr/bin/env python
"""Static check of `hipcc -S --cuda-device-only` output: per kernel and per LOOP (a label reached again by a later branch), the
`s_waitcnt vmcnt(N)` values the COMPILER inserted and the ones that come from inline asm (between #ASMSTART / #ASMEND).

Why: hipcc's wait-count pass decides how much of a software prefetch is one.  A compiler `vmcnt(0)` inside a tile loop means every
outstanding load AND store is awaited there (round 5: conv1x1_stream waited for its own stores at the top of every step; conv1x1_deepk's
"two load groups in flight" were one; behind asm set-up waits -- invisible to the pass -- fire_dma and ConvDet got waits in front of
register-resident weights inside their tile loops).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Isqueezedet_amd/csrc [the file's flags in build.py] -x hip -S --cuda-device-only \
          squeezedet_amd/csrc/conv1x1.hip -o /tmp/conv1x1.s
    python tools/wait_scan.py /tmp/conv1x1.s [regex on the mangled kernel name]
tests/test_static_waits.py runs the same scan on the forward's kernels (no GPU needed)."""
import re
import sys


def scan_file(path, pattern=""):
    """-> [(kernel, [(loop label, instructions, mfma count, [compiler vmcnt], [asm vmcnt]), ...]), ...] for kernels matching `pattern`;
    only loops that contain MFMAs are listed."""
    out = []
    name, lines = None, []
    for l in open(path):
        m = re.match(r"^(\w+):\s+; @", l)
        if m:
            name, lines = m.group(1), []
            continue
        if name is None:
            continue
        if l.strip().startswith(".Lfunc_end"):
            if re.search(pattern, name):
                loops = _loops(lines)
                if loops:
                    out.append((name, loops))
            name = None
            continue
        lines.append(l)
    return out


def _loops(lines):
    inasm, tagged = False, []
    for l in lines:
        if "#ASMSTART" in l:
            inasm = True
        tagged.append((l, inasm))
        if "#ASMEND" in l:
            inasm = False
    labels = {}
    for i, (l, _) in enumerate(tagged):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    res = []
    for i, (l, _) in enumerate(tagged):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
        if not m:
            continue
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            body = tagged[labels[t]:i]
            nm = sum("v_mfma" in x for x, _ in body)
            if nm == 0:
                continue
            cw = [int(re.search(r"vmcnt\((\d+)\)", x).group(1)) for x, ia in body if "vmcnt" in x and not ia]
            aw = [int(re.search(r"vmcnt\((\d+)\)", x).group(1)) for x, ia in body if "vmcnt" in x and ia]
            res.append((t, i - labels[t], nm, cw, aw))
    return res


def _regs(tok):
    """'v[82:85]' -> {82..85}, 'v82' -> {82}; anything else -> empty"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def asm_load_violations(path, pattern=""):
    """Registers written by INLINE-ASM `global_load_dwordx4` (loads the compiler's wait-count pass cannot see: conv1x1_pipe's weight
    fragments) must not be mentioned by ANY instruction between the load and the first MFMA that reads them -- that MFMA sits behind
    the hand-counted wait (the wait's asm statement passes the registers through), anything earlier would read or clobber a register
    whose data is still in flight (a register-allocator copy at a loop edge, a spill).  -> [(kernel, loop label, instruction), ...];
    only loops that contain MFMAs are scanned, the body is walked twice (a load at the bottom is consumed at the top)."""
    bad = []
    name, lines = None, []
    for l in open(path):
        m = re.match(r"^(\w+):\s+; @", l)
        if m:
            name, lines = m.group(1), []
            continue
        if name is None:
            continue
        if l.strip().startswith(".Lfunc_end"):
            if re.search(pattern, name):
                bad += [(name, t, i) for t, i in _asm_load_check(lines)]
            name = None
            continue
        lines.append(l)
    return bad


def _asm_load_check(lines):
    inasm, tagged = False, []
    for l in lines:
        if "#ASMSTART" in l:
            inasm = True
        tagged.append((l, inasm))
        if "#ASMEND" in l:
            inasm = False
    labels = {}
    for i, (l, _) in enumerate(tagged):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, (l, _) in enumerate(tagged):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
        if not m:
            continue
        t = m.group(1) or m.group(2)
        if t not in labels or labels[t] >= i:
            continue
        body = tagged[labels[t]:i]
        if not any("v_mfma" in x for x, _ in body):
            continue
        flight = set()
        for x, ia in body + body:
            ins = x.split(";")[0].strip()
            if not ins or ins.startswith((".", "#")) or ins.endswith(":"):
                continue
            toks = [tk.strip() for tk in re.split(r"[\s,]+", ins) if tk.strip()]
            op, args = toks[0], toks[1:]
            mentioned = set()
            for a in args:
                mentioned |= _regs(a)
            if ia and op == "global_load_dwordx4":
                dst = _regs(args[0])
                if (mentioned - dst) & flight or dst & flight:
                    out.append((t, ins))
                flight |= dst
                continue
            if op.startswith("v_mfma"):
                srcs = set()
                for a in args[1:3]:
                    srcs |= _regs(a)
                if _regs(args[0]) & flight or (_regs(args[3]) if len(args) > 3 else set()) & flight:
                    out.append((t, ins))
                flight -= srcs
                continue
            if mentioned & flight:
                out.append((t, ins))
    return out


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, loops in scan_file(path, pat):
        print(name[:120])
        for t, n, nm, cw, aw in loops:
            print("   loop %s len %d mfma %d | compiler vmcnt: %s | asm vmcnt: %s" % (t, n, nm, " ".join(map(str, cw)) or "-", " ".join(map(str, aw[:12])) or "-"))


if __name__ == "__main__":
    main()
