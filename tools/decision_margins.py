#!/usr/bin/env python
"""End-to-end decision-margin report: image -> picks on the GPU vs the CPU oracle, 16 seeded images at
384x1248 and 375x1242, float32 and float16.    python tools/decision_margins.py [out.txt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import decision_margins as DM  # noqa: E402

out = []
for size in ((384, 1248), (375, 1242)):
    for dt in ("fp32", "fp16"):
        rows, summary = DM.run(size, dt, nimg=16, seed=40)
        out.append(DM.format_report(rows, summary))
text = "\n\n".join(out)
print(text)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        fh.write(text + "\n")
