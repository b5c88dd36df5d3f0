"""Builds the HBM-traffic summary bench.py attaches to its roofline object from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs of tools/pmc_forward.py, as MI355X_MICROARCH.md prescribes):

    python profiles/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <layer names, comma separated> [config]

tools/pmc_forward.py runs 5 forwards of the plan and nothing else after the parameters are packed, so the LAST
5 x L dispatches of the trace are the L launches of each forward, in plan order: launch i of a forward is layer i
(no kernel-name table to go stale).  Per layer: the mean over the last 3 forwards of FETCH_SIZE / WRITE_SIZE (KiB, as
rocprofv3 reports them) and the corrected byte count 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950: FETCH_SIZE
tallies 128-byte requests at 64 bytes -- MI355X_MICROARCH.md, HBM section; calibrated on the streaming 1x1 conv in
round 1).  The JSON carries the build fingerprint of the kernels it was taken on; bench.py refuses any other."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"], float(r["Counter_Value"])) for r in rows]


def main():
    import bench
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    layers = sys.argv[4].split(",")
    L = len(layers)
    assert len(fetch) >= 5 * L and len(write) >= 5 * L, (len(fetch), len(write), L)
    f_tail, w_tail = fetch[-3 * L:], write[-3 * L:]
    by_layer, kernels = {}, []
    for i, name in enumerate(layers):
        fk = [f_tail[r * L + i] for r in range(3)]
        wk = [w_tail[r * L + i] for r in range(3)]
        assert len({k for k, _ in fk}) == 1 and fk[0][0] == wk[0][0], "launch order differs between forwards / passes at layer %s" % name
        f = sum(v for _, v in fk) / 3.0
        w = sum(v for _, v in wk) / 3.0
        by_layer[name] = int(2 * f * 1024 + w * 1024)
        kernels.append({"layer": name, "kernel": fk[0][0][:120], "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                        "hbm_bytes_per_launch_corrected": by_layer[name]})
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python tools/pmc_forward.py`; per-launch means over 3 "
            "forwards, KiB as reported.  gfx950 correction: bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (FETCH_SIZE tallies 128-B requests "
            "at 64 B).  Counters sit on the L2's memory side: hits in the 256 MiB Infinity Cache are included, so 'traffic' is fabric "
            "traffic, an upper bound on DRAM traffic.")
    config = sys.argv[5] if len(sys.argv) > 5 else "sqdet_infer"
    note = note.replace("`python tools/pmc_forward.py`", "`PMC_CONFIG=%s python tools/pmc_forward.py`" % config)
    json.dump({"note": note, "config": config, "build_fingerprint": bench.build_fingerprint(), "kernels": kernels, "by_layer": by_layer},
              open(sys.argv[3], "w"), indent=1)
    for k in kernels:
        print("%-44s %10.1f MB  %s" % (k["layer"], k["hbm_bytes_per_launch_corrected"] / 1e6, k["kernel"][:60]))


if __name__ == "__main__":
    main()
