"""KITTI evaluation plumbing (reference src/eval.py:69-101 + src/dataset/kitti.py:100-159): the
detection files squeezedet_amd/kitti_eval.py writes, judged by the REFERENCE'S OWN C++ evaluator built into
oracle/_ref/ (oracle/Makefile).  Host-only."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "evaluate_object")
CLASSES = ("car", "pedestrian", "cyclist")       # kitti.py:22


def _dataset(root, n_images, rs):
    """A tiny KITTI tree: training/label_2/<idx>.txt + ImageSets/val.txt; returns {idx: [(cls, x1,y1,x2,y2)]}."""
    os.makedirs(os.path.join(root, "training", "label_2"))
    os.makedirs(os.path.join(root, "ImageSets"))
    idxs, gts = ["%06d" % i for i in range(n_images)], {}
    for idx in idxs:
        rows = []
        for k in range(3):
            cls = CLASSES[k]
            x1, y1 = rs.uniform(10 + 300 * k, 200 + 300 * k), rs.uniform(50, 150)
            w, h = rs.uniform(60, 120), rs.uniform(60, 120)
            rows.append((cls, x1, y1, x1 + w, y1 + h))
        gts[idx] = rows
        with open(os.path.join(root, "training", "label_2", idx + ".txt"), "w") as f:
            for cls, x1, y1, x2, y2 in rows:
                f.write("%s 0.00 0 -1.50 %.2f %.2f %.2f %.2f 1.50 1.60 3.90 1.00 1.70 20.00 -1.50\n" % (cls.capitalize(), x1, y1, x2, y2))
    with open(os.path.join(root, "ImageSets", "val.txt"), "w") as f:
        f.write("\n".join(idxs) + "\n")
    return idxs, gts


def test_detection_file_format(tmp_path):
    from squeezedet_amd import kitti_eval as K
    ab = K.new_all_boxes(3, 1)
    K.add_detections(ab, 0, [[100.0, 50.0, 40.0, 20.0]], [0.87654], [1])
    K.write_detection_files(str(tmp_path), ["000007"], CLASSES, ab)
    line = open(tmp_path / "000007.txt").read()
    assert line == "pedestrian -1 -1 0.0 80.00 40.00 120.00 60.00 0.0 0.0 0.0 0.0 0.0 0.0 0.0 0.877\n"
    ab = K.new_all_boxes(3, 1)
    K.add_detections(ab, 0, [[100.0, 50.0, 40.0, 20.0]], [0.5], [0], scale=(2.0, 0.5))     # eval.py:83-84 rescale
    assert np.allclose(ab[0][0][0], [40.0, 80.0, 60.0, 120.0, 0.5])


@pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_ref/evaluate_object not built (needs /root/reference at build time)")
def test_reference_evaluator_reads_our_files(tmp_path):
    from squeezedet_amd import kitti_eval as K
    rs = np.random.RandomState(0)
    root = str(tmp_path / "KITTI")
    idxs, gts = _dataset(root, 48, rs)      # >= 41 objects per class: the evaluator samples 41 recall points
    perfect, half = K.new_all_boxes(3, len(idxs)), K.new_all_boxes(3, len(idxs))
    for i, idx in enumerate(idxs):
        for cls, x1, y1, x2, y2 in gts[idx]:
            box = [(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1]            # what filter_prediction would return
            sc = float(rs.uniform(0.5, 0.9))       # distinct scores: the evaluator samples recall through score thresholds
            K.add_detections(perfect, i, [box], [sc], [CLASSES.index(cls)])
            if i % 2 == 0:
                K.add_detections(half, i, [box], [sc], [CLASSES.index(cls)])
            else:                                                             # a false positive far away instead
                K.add_detections(half, i, [[box[0], box[1] + 200, box[2], box[3]]], [0.95], [CLASSES.index(cls)])
    aps, names = K.evaluate_detections(TOOL, root, "val", str(tmp_path / "eval_a"), 100, idxs, CLASSES, perfect)
    assert names[:3] == ["car_easy", "car_medium", "car_hard"] and len(aps) == 9
    assert all(abs(a - 1.0) < 1e-6 for a in aps), aps
    aps2, _ = K.evaluate_detections(TOOL, root, "val", str(tmp_path / "eval_b"), 100, idxs, CLASSES, half)
    assert all(0.0 < a < 0.75 for a in aps2), aps2
