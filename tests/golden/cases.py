"""Seeded input generators shared by make_golden.py (which runs the reference
on them) and the tests (which run the oracle / the HIP path on them).
np.random.RandomState streams are stable across NumPy versions."""
import numpy as np

A = 16848  # SqueezeDet anchors (24*78*9)

FILTER_CASES = ["uniform0", "uniform1", "clustered2", "clustered3", "thresh_branch4", "oneclass5", "topn_all_suppressed6"]


def make_filter_case(name):
    """Returns boxes [A,4] f32 (cx,cy,w,h), probs [A] f32, cls [A] i64 and the
    mc overrides for the case."""
    seed = int(name[-1])
    rs = np.random.RandomState(seed)
    overrides = {}
    if name.startswith("uniform"):
        # cx in [0,1247], cy in [0,383], w in [1,300], h in [1,200]; probs U(0,1); cls randint(0,3)
        boxes = np.stack([rs.uniform(0, 1247, A), rs.uniform(0, 383, A), rs.uniform(1, 300, A), rs.uniform(1, 200, A)], 1).astype(np.float32)
        probs = rs.uniform(0, 1, A).astype(np.float32)
        cls = rs.randint(0, 3, A).astype(np.int64)
    elif name.startswith("clustered") or name.startswith("oneclass") or name.startswith("topn_all"):
        # detector-like: a few objects, many near-duplicate boxes around each -> NMS does real work
        nobj = 6 if not name.startswith("topn_all") else 1
        centers = np.stack([rs.uniform(100, 1100, nobj), rs.uniform(60, 320, nobj), rs.uniform(40, 250, nobj), rs.uniform(30, 150, nobj)], 1)
        which = rs.randint(0, nobj, A)
        jitter = rs.normal(0, 1, (A, 4)) * np.array([12., 8., 10., 8.])
        boxes = centers[which] + jitter
        boxes[:, 2:] = np.maximum(boxes[:, 2:], 1.0)
        boxes = boxes.astype(np.float32)
        probs = (rs.uniform(0, 1, A) ** 4).astype(np.float32)
        if name.startswith("oneclass"):
            cls = np.ones(A, np.int64)
        elif name.startswith("topn_all"):
            cls = np.zeros(A, np.int64)
        else:
            cls = (which % 3).astype(np.int64)
    elif name.startswith("thresh_branch"):
        # else-branch of nn_skeleton.py:716-720: TOP_N_DETECTION disabled, PROB_THRESH applied
        boxes = np.stack([rs.uniform(0, 1247, A), rs.uniform(0, 383, A), rs.uniform(1, 300, A), rs.uniform(1, 200, A)], 1).astype(np.float32)
        probs = rs.uniform(0, 1, A).astype(np.float32)
        cls = rs.randint(0, 3, A).astype(np.int64)
        overrides = {"TOP_N_DETECTION": 0, "PROB_THRESH": 0.985}
    else:
        raise ValueError(name)
    return boxes, probs, cls, overrides
