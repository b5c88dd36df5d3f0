"""Builds the HBM-traffic summary bench.py attaches to its roofline object from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs of the same command, as MI355X_MICROARCH.md prescribes):

    python profiles/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> "<command>"

Per kernel symbol (+ grid size) the mean FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) and the
corrected byte count  2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024  (gfx950: FETCH_SIZE counts 128-byte requests at
64 bytes -- MI355X_MICROARCH.md, HBM section; calibrated on the streaming 1x1 conv in round 1).  `by_layer` maps the
launches of sqdet_net_forward (SqueezeDet, batch 32, 375x1242, fp16) to those kernels; layers that share one
kernel instantiation AND grid (fire6/7, fire8/9, fire10/11) get the mean of the two."""
import collections
import csv
import json
import sys


def load(path, counter):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc.setdefault((r["Kernel_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


# layer of sqdet_net_forward -> (substring of the kernel symbol, rank among equal symbols by grid size, descending)
LAYERS = [
    ("conv1+pool1", "stem_strip", 0),
    ("fire2", "fire_streamIDF16_Li2ELi1ELi4E", 0), ("fire3+pool3", "fire_streamIDF16_Li4ELi1ELi4E", 0),
    ("fire3", "fire_streamIDF16_Li4ELi1ELi4E", 0), ("pool3", "maxpool3_kernel", 0),          # (plans without pool fusion)
    ("fire4", "fire_streamIDF16_Li4ELi2ELi8E", 0), ("fire5+pool5", "fire_streamIDF16_Li8ELi2ELi8E", 0),
    ("fire5", "fire_streamIDF16_Li8ELi2ELi8E", 0), ("pool5", "maxpool3_kernel", 1),
    ("fire6", "fire_fusedIDF16_Li3ELi3ELi8E", 0), ("fire7", "fire_fusedIDF16_Li3ELi3ELi8E", 0),
    ("fire8", "fire_fusedIDF16_Li4ELi4ELi8E", 0), ("fire9", "fire_fusedIDF16_Li4ELi4ELi8E", 0),
    ("fire10", "fire_fusedIDF16_Li6ELi3ELi8E", 0), ("fire11", "fire_fusedIDF16_Li6ELi3ELi8E", 0),
    ("conv12", "conv3x3_tileIDF16_Li8ELi5ELb1E", 0),
    ("interpret_output", "interpret_kernel", 0), ("filter_prediction", "filter_topn_fast", 0),
]


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    kernels = []
    for key in fetch:
        if "sqdet" not in key[0]:
            continue
        f, w = fetch[key], write.get(key, 0.0)
        kernels.append({"kernel": key[0], "grid_threads": key[1], "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                        "hbm_bytes_per_launch_corrected": int(2 * f * 1024 + w * 1024)})
    by_layer = {}
    for layer, sub, rank in LAYERS:
        cand = sorted([k for k in kernels if sub in k["kernel"]], key=lambda k: -k["grid_threads"])
        if rank < len(cand):
            by_layer[layer] = cand[rank]["hbm_bytes_per_launch_corrected"]
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `%s`; per-launch means in KiB as reported. "
            "gfx950 correction: bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (FETCH_SIZE tallies 128-B requests at 64 B). "
            "Kernels shared by two layers are averaged.  Counters sit on the L2's memory side: hits in the 256 MiB "
            "Infinity Cache are included, so 'traffic' is fabric traffic, an upper bound on DRAM traffic." % (sys.argv[4] if len(sys.argv) > 4 else ""))
    json.dump({"note": note, "kernels": kernels, "by_layer": by_layer}, open(sys.argv[3], "w"), indent=1)
    for k, v in by_layer.items():
        print("%-20s %8.1f MB" % (k, v / 1e6))


if __name__ == "__main__":
    main()
