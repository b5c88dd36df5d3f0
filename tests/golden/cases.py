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


# ---- label assignment (dataset/imdb.py:195-239) ----
LABEL_CASES = ["squeezeDet_kitti0", "squeezeDet_contention1", "squeezeDet_nooverlap2", "res50_kitti3", "squeezeDetPlus_kitti4"]
LABEL_MAX_OBJECTS = 12


def _strict_pick(anchors, box, taken):
    """The anchor imdb.py:195-229 would give `box` when `taken` are gone -- or None when the pick hangs on a TIE
    (equal IoU / equal distance between two free anchors: np.argsort's order among equal keys is unspecified, so the
    reference's own answer depends on the NumPy build; such boxes are not used as golden cases).  Ties are structural
    here, not rare: a box lying inside several anchors of one shape (or containing several) has IoU = area ratio with
    all of them."""
    from oracle import sqdet_oracle as O
    ov = O.batch_iou(anchors, box)
    free = np.ones(len(anchors), bool)
    free[list(taken)] = False
    key = ov if ov[free].max() > 0 else -np.sum(np.square(box - anchors), axis=1)
    cand = np.where(free, key, -np.inf)
    a = int(np.argmax(cand))
    cand[a] = -np.inf
    return a if key[a] > cand.max() else None


def make_label_case(name):
    """Returns (config key, rois, orig_sizes): per image a list of [cx, cy, w, h, class] in ORIGINAL image coordinates
    (what dataset/kitti.py:50-78 stores in _rois) and the original (height, width).  Every box is drawn until its
    anchor pick (after scaling to the network input, imdb.py:186-190) is decided by a strict inequality."""
    from oracle import sqdet_oracle as O
    cfg = name.split("_")[0]
    mc = {"squeezeDet": O.kitti_squeezeDet_config, "squeezeDetPlus": O.kitti_squeezeDetPlus_config, "res50": O.kitti_res50_config}[cfg]()
    anchors = np.asarray(mc.ANCHOR_BOX)
    seed = int(name[-1])
    rs = np.random.RandomState(100 + seed)
    rois, sizes = [], []
    for i in range(6):
        h, w = [(375, 1242), (370, 1224), (374, 1238), (376, 1241)][rs.randint(0, 4)]
        sx, sy = mc.IMAGE_WIDTH / float(w), mc.IMAGE_HEIGHT / float(h)
        scale = np.array([sx, sy, sx, sy])
        taken, r = set(), []

        def add(box, cls):
            a = _strict_pick(anchors, np.asarray(box, np.float64) * scale, taken)
            if a is None:
                return False
            taken.add(a)
            r.append([float(box[0]), float(box[1]), float(box[2]), float(box[3]), int(cls)])
            return True

        def draw():
            bw = rs.uniform(18, 330)
            bh = bw * (rs.uniform(0.4, 0.7) if rs.uniform() < 0.6 else rs.uniform(1.8, 2.8))
            return [rs.uniform(0, w), rs.uniform(0, h), bw, min(bh, 300.0)]

        n = rs.randint(2, 9)
        if "nooverlap" in name:     # overlaps no anchor: nearest FREE anchor (imdb.py:222-229)
            assert add([-900.0 - 7.3 * i, -700.0 + 3.1 * i, 30.0, 20.0], 1)
        while len(r) < n:
            add(draw(), rs.randint(0, 3))
        if "contention" in name and i % 2 == 0:
            # the same box again (and a near-duplicate): they must take the next-best free anchors
            for rep in (list(r[0][:4]), [r[0][0] + 0.37, r[0][1] - 0.21, r[0][2] * 1.01, r[0][3] * 0.99], list(r[0][:4])):
                add(rep, r[0][4])
        if "nooverlap" in name:
            assert add([w + 800.0 + 2.7 * i, h + 500.0, 25.0 + i, 18.0], 2)
            if i == 1:
                add([-900.0 - 7.3 * i, -700.0 + 3.1 * i, 30.0, 20.0], 1)   # twice the same far box: second-nearest anchor
        rois.append(r[:LABEL_MAX_OBJECTS])
        sizes.append((h, w))
    return cfg, rois, sizes
