"""Per-KERNEL counter summary of a whole bench.py step (the training configs: ~300 launches of ~40 kernels per step), from rocprofv3 PMC
passes of `python bench.py --config <c> --no-cpu-baseline --no-graph --steps 4 --warmup 2` collected SEPARATELY (MI355X_MICROARCH.md):

    python profiles/pmc_kernels.py <out.json> <config> <counter_collection.csv> [<counter_collection.csv> ...]

For every kernel name: launches, and per launch the mean FETCH_SIZE / WRITE_SIZE (KiB as reported), the corrected fabric bytes
2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes), SQ_INSTS_MFMA (matrix instructions,
all waves) and SQ_VALU_MFMA_BUSY_CYCLES.  `steps` = launches of the loss kernel (one per step); `hbm_bytes_per_step` = the sum over the
kernels of bytes x launches / steps.  The JSON carries the build fingerprint; bench.py attaches it to a line of the same build only."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    out, config, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, d in acc.items():
        if "calib_" in k or "at::native" in k or "rocclr" in k:
            continue
        n = max(len(v) for v in d.values())
        mean = lambda c: (sum(d[c]) / len(d[c])) if d.get(c) else None
        e = {"launches": n, "FETCH_SIZE_KB_raw": mean("FETCH_SIZE"), "WRITE_SIZE_KB": mean("WRITE_SIZE"), "SQ_INSTS_MFMA": mean("SQ_INSTS_MFMA"),
             "SQ_VALU_MFMA_BUSY_CYCLES": mean("SQ_VALU_MFMA_BUSY_CYCLES")}
        if e["FETCH_SIZE_KB_raw"] is not None and e["WRITE_SIZE_KB"] is not None:
            e["hbm_bytes_per_launch_corrected"] = int(2 * e["FETCH_SIZE_KB_raw"] * 1024 + e["WRITE_SIZE_KB"] * 1024)
        kernels[k[:120]] = {a: (round(b, 1) if isinstance(b, float) else b) for a, b in e.items()}
    steps = max([e["launches"] for k, e in kernels.items() if "loss_kernel" in k] or [0])
    per_step = None
    if steps:
        per_step = int(sum(e.get("hbm_bytes_per_launch_corrected", 0) * e["launches"] for e in kernels.values()) / steps)
    note = ("rocprofv3 --pmc <counter> (one pass per counter group) on `python bench.py --config %s --no-cpu-baseline --no-graph --steps 4 --warmup 2`; "
            "per-launch means over every dispatch of the kernel.  gfx950 correction: bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.  The counters sit "
            "on the L2's memory side: hits in the 256 MiB Infinity Cache are included (fabric traffic, an upper bound on DRAM traffic)." % config)
    json.dump({"note": note, "config": config, "build_fingerprint": bench.build_fingerprint(), "steps": steps, "hbm_bytes_per_step": per_step,
               "kernels": kernels}, open(out, "w"), indent=1)
    top = sorted(kernels.items(), key=lambda kv: -(kv[1].get("hbm_bytes_per_launch_corrected", 0) * kv[1]["launches"]))[:8]
    print("steps %d, fabric bytes per step %s" % (steps, per_step))
    for k, e in top:
        print("%-70s x%-5d %9.1f MB/launch  mfma %s" % (k[:70], e["launches"], e.get("hbm_bytes_per_launch_corrected", 0) / 1e6, e["SQ_INSTS_MFMA"]))


if __name__ == "__main__":
    main()
