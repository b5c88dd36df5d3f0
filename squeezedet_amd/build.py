"""Builds squeezedet_amd/libsqdet_hip.so (the C-ABI library, include/sqdet.h) with hipcc
for gfx950.  hipcc cross-compiles without a GPU; the .so stays in-tree so it travels to the
GPU box with the repo snapshot.

    python -m squeezedet_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
# SQDET_BUILD_SUFFIX=_tl (with SQDET_EXTRA_DEFINES): an experiment build beside the default one -- its own object directory and
# libsqdet_hip_tl.so, loaded through SQDET_LIB (tools/ab_lib.sh, the timeline tools)
_SUFFIX = os.environ.get("SQDET_BUILD_SUFFIX", "")
OBJ = os.path.join(PKG, "csrc", "build" + _SUFFIX)
LIB = os.path.join(PKG, "libsqdet_hip%s.so" % _SUFFIX)

# (source, extra flags).  postproc.hip must not contract mul+add (bit-exact decode / IoU).
SOURCES = [
    ("common.cpp", []),
    # MFMA accumulators in VGPRs (unified gfx950 register file): hipcc's default AGPR form keeps a
    # second copy of the accumulators in VGPRs around the epilogue and halves occupancy.
    ("conv.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("conv3x3.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    # the ConvDet split-K kernel needs > 256 registers: accumulators in AGPRs (hipcc's default form), see convdet.hip
    # (-ffp-contract=off: its score epilogue shares postproc.h's float expressions with filter_fast.hip, bit for bit)
    ("convdet.hip", ["-ffp-contract=off"]),
    ("conv1x1.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("gemm1x1.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("conv1x1k.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("stem.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("stem2.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("stem3.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    # (-fno-honor-nans: without it every two-operand fmaxf on an MFMA result is preceded by a canonicalising v_max x, x, x:
    # 148 instead of 84 max instructions per pooled row)
    ("stem4.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"]),
    ("stem5.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"]),
    ("fire.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("fire2.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("fire3.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    # (-ffp-contract=off: its rider workgroups run filter_body.h's decode / IoU arithmetic, bit-exact by contract)
    ("chain.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-ffp-contract=off"]),
    ("pool.hip", []),
    ("bn.hip", ["-ffp-contract=off"]),
    ("preproc.hip", ["-ffp-contract=off"]),
    ("labels.hip", ["-ffp-contract=off"]),
    ("postproc.hip", ["-ffp-contract=off"]),
    ("filter_fast.hip", ["-ffp-contract=off"]),
    ("train.hip", ["-ffp-contract=off"]),
    ("wgrad.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    # (the calibration loops: hipcc's default AGPR form ROTATES the 16x16x32 loop's accumulators -- a[24:27] = mfma(.., a[22:25]) plus
    # v_accvgpr copies inside the loop -- so consecutive MFMAs depend on each other and the "bare MFMA loop" of rounds 3-4 read half
    # of what the box sustains: 1.07-1.29 PF/s where the clean loop reads ~1.7; found in round 5 with the counters)
    ("probe.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("net.cpp", []),
]
# (-Wno-inline-asm: the LDS-DMA helpers write M0 inside their asm and say so in the clobber list -- round-4 ADVICE; clang then warns
# once per inlined copy that M0 is a reserved register.  The clobber only makes the compiler re-materialise M0 for its own uses.)
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """SQDET_EXTRA_DEFINES="-DSQDET_CHAIN_DBG ..." adds experiment-only defines (tools/chainbench.py --dbg)."""
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "sqdet.h"))
    headers.append(os.path.abspath(__file__))
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + os.environ.get("SQDET_EXTRA_DEFINES", "").split() + ["-x", "hip", "-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print("[sqdet build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        if r.stderr.strip() and verbose:
            print(r.stderr[-4000:], file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
