"""Pins the oracle's NumPy half against vectors produced by the reference's
own functions (tests/golden/make_golden.py), and -- when /root/reference is
mounted -- against the live reference functions."""
import hashlib
import os

import numpy as np
import pytest

from oracle import ref_numpy_half as ref
from oracle import sqdet_oracle as O
from tests.golden import cases


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("key,fn", [("squeezeDet", O.kitti_squeezeDet_config),
                                    ("squeezeDetPlus", O.kitti_squeezeDetPlus_config),
                                    ("res50", O.kitti_res50_config)])
def test_anchors_match_reference(golden_dir, key, fn):
    g = np.load(os.path.join(golden_dir, "anchors.npz"))
    mc = fn()
    ab = np.asarray(mc.ANCHOR_BOX)
    assert ab.dtype == np.float64
    assert tuple(g["anchors_%s_shape" % key]) == ab.shape
    assert mc.ANCHORS == ab.shape[0]
    np.testing.assert_array_equal(ab[[0, 1, 9, 701, 702, -1]], g["anchors_%s_rows" % key])
    assert sha(ab) == str(g["anchors_%s_sha256" % key])
    assert sha(ab.astype(np.float32)) == str(g["anchors_%s_f32_sha256" % key])


def test_anchor_known_answers():
    # SURVEY.md 8c KATs (read off the reference run)
    ab = O.kitti_squeezeDet_config().ANCHOR_BOX
    assert ab.shape == (16848, 4)
    np.testing.assert_allclose(ab[0], [15.79746835, 15.36, 36, 37], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ab[1][2:], [366, 174])
    np.testing.assert_allclose(ab[9], [31.59493671, 15.36, 36, 37], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ab[701], [1232.20253165, 15.36, 72, 43], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ab[702], [15.79746835, 30.72, 36, 37], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ab[-1], [1232.20253165, 368.64, 72, 43], rtol=0, atol=1e-8)
    assert O.kitti_squeezeDetPlus_config().ANCHOR_BOX.shape == (15048, 4)
    # closed form: cx=(w+1)*IMG_W/(W+1), cy=(h+1)*IMG_H/(H+1), bit-for-bit
    h, w, k = 7, 33, 4
    a = (h * 78 + w) * 9 + k
    assert ab[a][0] == (w + 1) * 1248.0 / 79 and ab[a][1] == (h + 1) * 384.0 / 25


def test_util_kats(golden_dir):
    g = np.load(os.path.join(golden_dir, "util_kat.npz"))
    b = g["iou_boxes"]
    iou = O.batch_iou(b, b[0])
    assert iou.dtype == np.float32
    np.testing.assert_array_equal(iou, g["iou_vs_row0"])
    np.testing.assert_allclose(iou, [1.0, 0.8181818, 0.0, 0.6666667], rtol=1e-6)
    assert O.nms(b, g["nms_probs"], 0.4) == list(g["nms_keep"]) == [True, False, True, False]
    # the non-greedy chain case: greedy NMS would keep box 2
    c = g["chain_boxes"]
    np.testing.assert_array_equal(O.batch_iou(c[1:], c[0]), g["chain_iou01"])
    np.testing.assert_array_equal(O.batch_iou(c[2:], c[1]), g["chain_iou12"])
    assert O.nms(c, g["chain_probs"], 0.4) == list(g["chain_keep"]) == [True, False, False]
    np.testing.assert_array_equal(np.array(O.bbox_transform([1., 2., 3., 4.])), g["bbox_transform_1234"])
    np.testing.assert_array_equal(g["bbox_transform_1234"], [-0.5, 0, 2.5, 4])
    np.testing.assert_array_equal(np.array(O.bbox_transform_inv([0., 0., 2., 3.])), g["bbox_transform_inv_0023"])
    np.testing.assert_array_equal(g["bbox_transform_inv_0023"], [1.5, 2, 3, 4])
    rs = np.random.RandomState(7)
    bx = np.stack([rs.uniform(0, 1247, 300), rs.uniform(0, 383, 300), rs.uniform(1, 300, 300), rs.uniform(1, 200, 300)], 1).astype(np.float32)
    assert sha(bx) == str(g["iou300_boxes_sha256"])
    np.testing.assert_array_equal(O.batch_iou(bx, bx[5]), g["iou300_vs_row5"])


@pytest.mark.parametrize("name", cases.FILTER_CASES)
def test_filter_prediction_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "filter_prediction.npz"))
    boxes, probs, cls, overrides = cases.make_filter_case(name)
    assert sha(boxes) + sha(probs) + sha(cls) == str(g[name + "_in_sha256"])
    mc = O.kitti_squeezeDet_config()
    for k, v in overrides.items():
        mc[k] = v
    fb, fp, fc = O.filter_prediction(mc, boxes, probs, cls)
    np.testing.assert_array_equal(np.array(fb, np.float32).reshape(-1, 4), g[name + "_boxes"])
    np.testing.assert_array_equal(np.array(fp, np.float32), g[name + "_probs"])
    np.testing.assert_array_equal(np.array(fc, np.int64), g[name + "_cls"])


def test_filter_prediction_uniform0_known_answer(golden_dir):
    # 61 detections, first probs 0.99974638, 0.99969339, 0.99956030
    g = np.load(os.path.join(golden_dir, "filter_prediction.npz"))
    assert len(g["uniform0_probs"]) == 61
    np.testing.assert_allclose(g["uniform0_probs"][:3], [0.99974638, 0.99969339, 0.99956030], rtol=1e-7)


@pytest.mark.skipif(not ref.available(), reason="reference tree not mounted")
def test_oracle_vs_live_reference():
    ns = ref.load()
    for seed in range(20, 26):
        rs = np.random.RandomState(seed)
        n = 200
        bx = np.stack([rs.uniform(0, 600, n), rs.uniform(0, 383, n), rs.uniform(1, 300, n), rs.uniform(1, 200, n)], 1).astype(np.float32)
        pr = rs.uniform(0, 1, n).astype(np.float32)
        assert O.nms(bx, pr, 0.4) == ns.util.nms(bx, pr, 0.4)
        np.testing.assert_array_equal(O.batch_iou(bx, bx[3]), ns.util.batch_iou(bx, bx[3]))
    mc_ref = ns.cfg_squeezeDet.kitti_squeezeDet_config()
    mc = O.kitti_squeezeDet_config()
    for k in ("IMAGE_WIDTH", "IMAGE_HEIGHT", "CLASSES", "ANCHOR_PER_GRID", "ANCHORS", "EXP_THRESH", "TOP_N_DETECTION",
              "PROB_THRESH", "NMS_THRESH", "PLOT_PROB_THRESH", "BATCH_SIZE", "WEIGHT_DECAY", "LEARNING_RATE",
              "MAX_GRAD_NORM", "MOMENTUM", "LR_DECAY_FACTOR", "LOSS_COEF_BBOX", "LOSS_COEF_CONF_POS",
              "LOSS_COEF_CONF_NEG", "LOSS_COEF_CLASS", "EPSILON", "DECAY_STEPS"):
        assert mc[k] == mc_ref[k], k
    np.testing.assert_array_equal(mc.ANCHOR_BOX, mc_ref.ANCHOR_BOX)


_LABEL_CFG = {"squeezeDet": O.kitti_squeezeDet_config, "squeezeDetPlus": O.kitti_squeezeDetPlus_config, "res50": O.kitti_res50_config}


@pytest.mark.parametrize("name", cases.LABEL_CASES)
def test_label_assignment_matches_reference(golden_dir, name):
    """train_oracle.assign_anchors (the restatement the GPU label builder is checked against) reproduces, bit for bit,
    what the reference's own imdb.read_batch (dataset/imdb.py:120-260, run unchanged by make_golden.py) produced:
    anchor indices (incl. boxes competing for one anchor and boxes that overlap nothing) and the float64 deltas."""
    from oracle import train_oracle as TO
    g = np.load(os.path.join(golden_dir, "labels.npz"))
    mc = _LABEL_CFG[name.split("_")[0]]()
    aidx, delta, bbox, cnt = g[name + "_aidx"], g[name + "_delta"], g[name + "_bbox"], g[name + "_count"]
    assert cnt.sum() >= 12
    for b in range(len(cnt)):
        n = int(cnt[b])
        a, d = TO.assign_anchors(mc, bbox[b, :n])
        assert a == aidx[b, :n].tolist(), "%s image %d" % (name, b)
        assert np.array_equal(np.asarray(d, np.float64).reshape(n, 4), delta[b, :n]), "%s image %d deltas" % (name, b)
        assert (aidx[b, n:] == -1).all()
    if "contention" in name:      # identical boxes did get DIFFERENT anchors, in order of preference
        n0 = int(cnt[0])
        dup = [k for k in range(1, n0) if np.array_equal(bbox[0, k], bbox[0, 0])]
        assert dup and len(set(aidx[0, :n0].tolist())) == n0
    if "nooverlap" in name:       # the far boxes took the nearest free corner anchors
        A = mc.ANCHORS
        assert aidx[0, 0] == 0 and aidx[0, int(cnt[0]) - 1] >= A - 9


@pytest.mark.skipif(not ref.available(), reason="reference tree not mounted")
def test_label_golden_regenerates_from_live_reference(golden_dir):
    """The committed labels.npz is what the reference's read_batch returns today (scaled boxes included)."""
    from oracle import ref_imdb_half as ref_imdb
    ns = ref.load()
    g = np.load(os.path.join(golden_dir, "labels.npz"))
    for name in cases.LABEL_CASES[:2]:
        cfg, rois, sizes = cases.make_label_case(name)
        mc = ns.cfg_squeezeDet.kitti_squeezeDet_config()
        labels, deltas, aidx, bboxes = ref_imdb.read_batch(mc, rois, sizes)
        for b in range(len(rois)):
            n = len(aidx[b])
            assert [int(v) for v in aidx[b]] == g[name + "_aidx"][b, :n].tolist()
            assert np.array_equal(np.asarray(bboxes[b], np.float64), g[name + "_bbox"][b, :n])
            assert [int(v) for v in labels[b]] == g[name + "_label"][b, :n].tolist()
