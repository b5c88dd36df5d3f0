"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN NumPy functions
(imported unchanged through oracle/ref_numpy_half.py).  Only runnable where
/root/reference is mounted; the outputs are committed so the GPU box (which has
no reference tree) can check against them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_imdb_half as ref_imdb  # noqa: E402
from oracle import ref_numpy_half as ref  # noqa: E402
from tests.golden import cases  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ns = ref.load()
    out = {}
    # ---- anchors (config/kitti_*_config.py:45-79) ----
    rows = [0, 1, 9, 701, 702, -1]
    for key, mod, fn in (("squeezeDet", ns.cfg_squeezeDet, "kitti_squeezeDet_config"),
                         ("squeezeDetPlus", ns.cfg_squeezeDetPlus, "kitti_squeezeDetPlus_config"),
                         ("res50", ns.cfg_res50, "kitti_res50_config")):
        mc = getattr(mod, fn)()
        ab = np.asarray(mc.ANCHOR_BOX)
        out["anchors_%s_shape" % key] = np.array(ab.shape)
        out["anchors_%s_rows" % key] = ab[rows]
        out["anchors_%s_sha256" % key] = np.array(sha(ab))
        out["anchors_%s_f32_sha256" % key] = np.array(sha(ab.astype(np.float32)))
    np.savez(os.path.join(HERE, "anchors.npz"), **out)

    # ---- util KATs (utils/util.py:32-76,167-196) ----
    out = {}
    b = np.array([[100, 100, 50, 50], [105, 100, 50, 50], [300, 100, 50, 50], [110, 100, 50, 50]], np.float32)
    out["iou_boxes"] = b
    out["iou_vs_row0"] = ns.util.batch_iou(b, b[0])
    out["nms_probs"] = np.array([.9, .8, .7, .6], np.float32)
    out["nms_keep"] = np.array(ns.util.nms(b, out["nms_probs"], 0.4))
    chain = np.array([[100, 100, 100, 100], [140, 100, 100, 100], [180, 100, 100, 100]], np.float32)
    out["chain_boxes"] = chain
    out["chain_probs"] = np.array([.9, .8, .7], np.float32)
    out["chain_iou01"] = ns.util.batch_iou(chain[1:], chain[0])
    out["chain_iou12"] = ns.util.batch_iou(chain[2:], chain[1])
    out["chain_keep"] = np.array(ns.util.nms(chain, out["chain_probs"], 0.4))
    out["bbox_transform_1234"] = np.array(ns.util.bbox_transform([1., 2., 3., 4.]))
    out["bbox_transform_inv_0023"] = np.array(ns.util.bbox_transform_inv([0., 0., 2., 3.]))
    rs = np.random.RandomState(7)
    bx = np.stack([rs.uniform(0, 1247, 300), rs.uniform(0, 383, 300), rs.uniform(1, 300, 300), rs.uniform(1, 200, 300)], 1).astype(np.float32)
    out["iou300_boxes_sha256"] = np.array(sha(bx))
    out["iou300_vs_row5"] = ns.util.batch_iou(bx, bx[5])
    np.savez(os.path.join(HERE, "util_kat.npz"), **out)

    # ---- filter_prediction (nn_skeleton.py:696-734) on seeded cases ----
    out = {}
    for name in cases.FILTER_CASES:
        mc = ns.cfg_squeezeDet.kitti_squeezeDet_config()
        boxes, probs, cls, overrides = cases.make_filter_case(name)
        for k, v in overrides.items():
            mc[k] = v
        if mc.TOP_N_DETECTION > 0:
            top = np.sort(probs)[::-1][:mc.TOP_N_DETECTION + 1]
            assert len(np.unique(top)) == len(top), "case %s has ties in the top-N+1 scores" % name
        fb, fp, fc = ref.filter_prediction(mc, boxes, probs, cls)
        out[name + "_in_sha256"] = np.array(sha(boxes) + sha(probs) + sha(cls))
        out[name + "_boxes"] = np.array(fb, np.float32).reshape(-1, 4)
        out[name + "_probs"] = np.array(fp, np.float32)
        out[name + "_cls"] = np.array(fc, np.int64)
        print(name, "->", len(fp), "detections; first probs", fp[:3])
    np.savez(os.path.join(HERE, "filter_prediction.npz"), **out)

    # ---- label assignment: imdb.read_batch (dataset/imdb.py:120-260) on seeded annotations ----
    out = {}
    cfgs = {"squeezeDet": (ns.cfg_squeezeDet, "kitti_squeezeDet_config"), "squeezeDetPlus": (ns.cfg_squeezeDetPlus, "kitti_squeezeDetPlus_config"),
            "res50": (ns.cfg_res50, "kitti_res50_config")}
    for name in cases.LABEL_CASES:
        cfg, rois, sizes = cases.make_label_case(name)
        mc = getattr(cfgs[cfg][0], cfgs[cfg][1])()
        labels, deltas, aidx, bboxes = ref_imdb.read_batch(mc, rois, sizes)
        # every pick must be decided by a STRICT inequality: np.argsort's order among equal keys is unspecified
        # (NumPy-version and CPU dependent), so a case with a tie at a decision point pins nothing
        anchors = np.asarray(mc.ANCHOR_BOX)
        for i in range(len(rois)):
            taken = set()
            for k, a in enumerate(aidx[i]):
                ov = ns.util.batch_iou(anchors, bboxes[i][k])
                free = np.ones(len(anchors), bool)
                free[list(taken)] = False
                if ov[a] > 0:
                    rest = free.copy(); rest[a] = False
                    assert ov[a] > ov[rest].max(), "%s image %d object %d: IoU tie at the pick" % (name, i, k)
                else:
                    assert ov[free].max() <= 0
                    dist = np.sum(np.square(bboxes[i][k] - anchors), axis=1)
                    rest = free.copy(); rest[a] = False
                    assert dist[a] < dist[rest].min(), "%s image %d object %d: distance tie at the pick" % (name, i, k)
                taken.add(int(a))
        B, M = len(rois), cases.LABEL_MAX_OBJECTS
        a = -np.ones((B, M), np.int64)
        d = np.zeros((B, M, 4), np.float64)
        bb = np.zeros((B, M, 4), np.float64)
        lb = -np.ones((B, M), np.int64)
        cnt = np.zeros(B, np.int32)
        for i in range(B):
            n = len(aidx[i])
            cnt[i] = n
            a[i, :n] = np.asarray(aidx[i], np.int64)
            d[i, :n] = np.asarray(deltas[i], np.float64)
            bb[i, :n] = np.asarray(bboxes[i], np.float64)
            lb[i, :n] = np.asarray(labels[i], np.int64)
            assert len(set(aidx[i])) == n
        out[name + "_aidx"], out[name + "_delta"], out[name + "_bbox"], out[name + "_label"], out[name + "_count"] = a, d, bb, lb, cnt
        print(name, "->", int(cnt.sum()), "objects; anchors of image 0:", a[0, :cnt[0]].tolist())
    np.savez(os.path.join(HERE, "labels.npz"), **out)


if __name__ == "__main__":
    main()
