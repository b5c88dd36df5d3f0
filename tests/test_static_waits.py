"""Static regression guard (no GPU): the wait counts hipcc puts into the tile loops of the forward's kernels and of the streaming 1x1
kernels.  Round 5 found `s_waitcnt vmcnt(0)` -- "wait for every outstanding load AND store" -- inside loops that were written as
software-pipelined: conv1x1_stream waited for its own stores at the top of every step, conv1x1_deepk's two load groups in flight were
one, fire_dma / ConvDet waited for register-resident weights in the middle of a tile (their set-up waits were inline asm, which the
compiler's wait-count pass cannot see).  The kernels are compiled here with the build's own flags (`hipcc -S --cuda-device-only`, ~40 s
in parallel) and scanned with tools/wait_scan.py: the loops that carry the MFMAs must hold COUNTED compiler waits only."""
import concurrent.futures as cf
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from squeezedet_amd import build as B  # noqa: E402
from tools import wait_scan as W  # noqa: E402

FILES = ["conv1x1.hip", "conv1x1k.hip", "fire3.hip", "convdet.hip", "chain.hip", "gemm1x1.hip", "conv3x3.hip"]
# the thresholds below were taken with this compiler; another version may schedule differently without anything being wrong
HIPCC_SEEN = "7.2"


def _hipcc_version():
    import shutil
    exe = B._hipcc()
    if not (os.path.isabs(exe) and os.path.exists(exe)) and shutil.which(exe) is None:
        return None
    r = subprocess.run([exe, "--version"], capture_output=True, text=True)
    import re
    m = re.search(r"HIP version:\s*(\d+\.\d+)", r.stdout)
    return m.group(1) if m else ""


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    ver = _hipcc_version()
    if ver is None:
        pytest.skip("no hipcc on this machine (the static wait-count guard needs the ROCm compiler)")
    if not ver.startswith(HIPCC_SEEN):
        pytest.skip("hipcc %s: the wait counts asserted here were taken with %s" % (ver, HIPCC_SEEN))
    d = tmp_path_factory.mktemp("isa")
    flags = {s: e for s, e in B.SOURCES}

    def comp(f):
        out = os.path.join(str(d), f.split(".")[0] + ".s")
        cmd = [B._hipcc()] + [c for c in B.COMMON if c != "-fPIC"] + flags[f] + ["-x", "hip", "-S", "--cuda-device-only", os.path.join(B.CSRC, f), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return f, out
    with cf.ThreadPoolExecutor(len(FILES)) as ex:
        return dict(ex.map(comp, FILES))


def _mfma_loops(path, pattern, min_mfma):
    loops = [(name, lp) for name, ls in W.scan_file(path, pattern) for lp in ls if lp[2] >= min_mfma]
    assert loops, "no loop with >= %d MFMAs in kernels matching %s" % (min_mfma, pattern)
    return loops


def test_conv1x1_stream_steps_never_wait_for_everything(asm):
    """float16 forms of the fire modules' stand-alone 1x1 convs: squeeze (<2,1,4>, <4,1,4>, <4,2,4>), expand (<1,4,4>): every compiler
    wait inside the two-step loop leaves the younger loads / stores in flight (the smallest count seen is 5)."""
    for name, (t, n, nm, cw, aw) in _mfma_loops(asm["conv1x1.hip"], r"conv1x1_streamIDF16_Li(1ELi4|2ELi1|4ELi1|4ELi2)ELi4ELb1", 8):
        assert cw and min(cw) >= 4, (name, t, cw)


def test_conv1x1_deepk_keeps_two_groups_in_flight(asm):
    """every fragment's wait leaves the other group's eight loads outstanding: vmcnt >= 8 throughout the block loop"""
    for name, (t, n, nm, cw, aw) in _mfma_loops(asm["conv1x1k.hip"], r"conv1x1_deepkIDF16_Li[1-6]ELi16E", 8):
        if n < 300:      # (the per-group MFMA sub-loops of partial unrolling carry no waits of their own)
            continue
        assert cw and min(cw) >= 8, (name, t, cw)


def test_fire_dma_tile_loops_hold_no_compiler_waits(asm):
    """the three forms without spills: hand-counted asm waits only (the fire5 + pool5 form reloads two spilled registers on its
    edge-tile path -- `scratch_load` + vmcnt(0) -- and is checked for nothing else)"""
    forms = {"ILb1ELb0ELi1ELi2ELi4ELi8ELi1ELi4": 0, "ILb1ELb1ELi1ELi2ELi2ELi8ELi2ELi4": 0, "ILb0ELb0ELi2ELi2ELi2ELi4ELi2ELi4": 0,
             "ILb0ELb1ELi2ELi1ELi1ELi8ELi3ELi4": 2}
    for form, allowed in forms.items():
        for name, (t, n, nm, cw, aw) in _mfma_loops(asm["fire3.hip"], "fire_dma" + form, 4):
            assert len(cw) <= allowed and all(c == 0 for c in cw), (name, t, cw)


def test_convdet_stage_loop_counts(asm):
    """the stage loop (360 MFMAs per trip): the explicit vmcnt(10) in front of the barrier, then counted waits for the weight steps --
    no vmcnt(0) behind the barrier (it was inherited from the tile loop's back edge while the tile's first wait was inline asm)"""
    for form in ("ILb1E", "ILb0E"):
        inner = [lp for _, lp in _mfma_loops(asm["convdet.hip"], "convdet_dma_kernel" + form, 300) if lp[1] < 2000]
        assert inner, form
        for t, n, nm, cw, aw in inner:
            assert cw and min(cw) >= 7 and aw == [10], (form, t, cw, aw)


def test_fire_chain_loops_are_hand_counted(asm):
    """the forward's five chain forms: no compiler wait inside any loop that carries MFMAs (every load there is an LDS-DMA)"""
    for form in ("ILi2ELi3ELb0ELi6ELi0", "ILi2ELi4ELb0ELi6ELi0", "ILi2ELi6ELb0ELi6ELi0", "ILi3ELi6ELb0ELi4ELi0", "ILi3ELi0ELb1ELi6ELi0"):
        for name, (t, n, nm, cw, aw) in _mfma_loops(asm["chain.hip"], "fire_chain" + form, 8):
            assert cw == [], (name, t, cw)


def test_conv1x1_pipe_loop_is_hand_counted_and_fragments_stay_untouched(asm):
    """conv1x1_pipe (gemm1x1.hip): every memory instruction of the K loop is inline asm, so (i) no compiler vmcnt wait may appear in a
    loop that carries MFMAs, (ii) the hand-counted waits are the ones the launcher's arithmetic expects -- (NS - 3)(Q + NTW) + NTW --
    and (iii) no instruction mentions a weight-fragment register between its asm load and the first MFMA that reads it (a register
    allocator copy at the loop edge would read data that has not arrived; the compiler cannot know)."""
    forms = {"IDF16_Li4ELi2ELi1ELi4ELb0E": (4, 1, 2), "IDF16_Li8ELi2ELi1ELi4ELb0E": (4, 2, 2), "IDF16_Li4ELi2ELi1ELi4ELb1E": (4, 1, 2),
             "IDF16_Li4ELi3ELi1ELi3ELb0E": (3, 1, 3), "IDF16_Li4ELi2ELi2ELi6ELb0E": (6, 2, 2), "IfLi4ELi2ELi1ELi4ELb1E": (4, 1, 2),
             "IDF16_Li4ELi5ELi1ELi4ELb0E": (4, 1, 5)}
    for form, (ns, q, ntw) in forms.items():
        loops = _mfma_loops(asm["gemm1x1.hip"], "conv1x1_pipe" + form, 8)
        for name, (t, n, nm, cw, aw) in loops:
            assert cw == [], (name, t, cw)
            assert aw and set(aw) == {(ns - 3) * (q + ntw) + ntw}, (name, t, aw)
    bad = W.asm_load_violations(asm["gemm1x1.hip"], "conv1x1_pipe")
    assert not bad, bad[:5]


def test_conv3x3_tile_forms_use_no_scratch(asm):
    """every float16 instantiation of conv3x3_tile (incl. the PAIR form): private_segment_fixed_size == 0.  A lambda the inliner leaves
    out of line takes the accumulators by reference, i.e. through scratch: round 6 shipped the PAIR form's epilogue that way for an
    hour -- results right, launches 9x slower, nothing but a per-launch table showed it."""
    import re
    s = open(asm["conv3x3.hip"]).read()
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*conv3x3_tileIDF16_\S*)(.*?)\.end_amdhsa_kernel", s, re.S):
        seen += 1
        priv = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
        assert priv == 0, (m.group(1), priv)
    assert seen >= 10, seen
