#!/usr/bin/env python
"""The command the PMC passes profile: an inference config's plan (default the SqueezeDet configs[1] plan: batch 32, 375x1242, float16;
PMC_CONFIG=sqdetplus_infer: SqueezeDet+ at batch 8), 2 warm-up + 3 measured sqdet_net_forward calls on distinct input batches, nothing
else.  Run under
    rocprofv3 --pmc FETCH_SIZE -d <dir> -o fetch --output-format csv -- python tools/pmc_forward.py
    rocprofv3 --pmc WRITE_SIZE -d <dir> -o write --output-format csv -- python tools/pmc_forward.py
(separate passes, counters only) and feed the two counter_collection.csv files to profiles/pmc_traffic.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = bench.parse_args(["--config", os.environ.get("PMC_CONFIG", "sqdet_infer")])
model, mc, xs = bench.build_infer_model(args, 0)
plan = model._native_plan(args.batch)
preds = None
# the step bench.py times runs ConvDet's SCORE form (det_probs from its epilogue): profile THAT kernel variant
scores = torch.empty((args.batch, mc.ANCHORS), dtype=torch.float32, device=xs[0].device) if plan.scores_supported() else None
for i in range(5):
    preds = plan.forward(xs[i % len(xs)], preds, scores=scores)
torch.cuda.synchronize()
print("layers:", [n for n, _, _ in plan.layer_table()])
