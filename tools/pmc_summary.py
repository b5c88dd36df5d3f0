#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (name prefix filter) the mean of every counter.
    python tools/pmc_summary.py <counter_collection.csv> [substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
names = []
for r in rows:
    if flt not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:64], r["Grid_Size"], r["VGPR_Count"], r["LDS_Block_Size"])
    agg.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] not in names:
        names.append(r["Counter_Name"])
for (k, g, v, l), d in agg.items():
    print("%s grid=%s vgpr=%s lds=%s" % (k, g, v, l))
    for n in names:
        if d[n]:
            print("    %-28s %14.4g" % (n, sum(d[n]) / len(d[n])))
