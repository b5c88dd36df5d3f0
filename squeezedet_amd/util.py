"""Mirror of the reference's box utilities (reference src/utils/util.py).  `nms` and
`batch_iou` run on the GPU through sqdet_filter_prediction's NMS kernel; the two 4-value
format conversions are plain host arithmetic exactly as in the reference (they are applied to
the <=64 filtered boxes by eval.py:91, never to the 16848-row arrays)."""
import numpy as np
import torch

from . import ops


def bbox_transform(bbox):
    """[cx, cy, w, h] -> [xmin, ymin, xmax, ymax] (utils/util.py:167-179)."""
    cx, cy, w, h = bbox
    return [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]


def bbox_transform_inv(bbox):
    """[xmin, ymin, xmax, ymax] -> [cx, cy, w, h]; note the +1 (utils/util.py:181-196)."""
    xmin, ymin, xmax, ymax = bbox
    width = xmax - xmin + 1.0
    height = ymax - ymin + 1.0
    return [xmin + 0.5 * width, ymin + 0.5 * height, width, height]


def nms(boxes, probs, threshold, device="cuda:0"):
    """Non-Maximum supression (utils/util.py:56-76): returns the keep list in input order.
    Evaluated by the HIP filter kernel with a single class and every box a candidate."""
    n = len(probs)
    if n == 0:
        return []
    b = torch.as_tensor(np.ascontiguousarray(boxes, dtype=np.float32)).to(device).reshape(1, n, 4)
    p = torch.as_tensor(np.ascontiguousarray(probs, dtype=np.float32)).to(device).reshape(1, n)
    c = torch.zeros((1, n), dtype=torch.int64, device=device)
    # top_n = 0 selects the threshold branch; -inf threshold keeps every box as a candidate
    _, _, _, oi, cnt = ops.filter_prediction(b, p, c, 1, 0, float(threshold), float("-inf"), max_out=n)
    k = int(cnt[0].item())
    keep = [False] * n
    for i in oi[0, :k].cpu().numpy():
        keep[int(i)] = True
    return keep
