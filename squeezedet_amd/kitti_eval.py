"""KITTI evaluation plumbing of the reference's eval path: turning filter_prediction output
into the `all_boxes` table (src/eval.py:69-91), writing the KITTI detection files
(src/dataset/kitti.py:100-127) and running the KITTI C++ evaluator / reading its AP files (:129-159).
Pure host code: the detector in front of it is the GPU hot path."""
import os
import subprocess

import numpy as np

from .util import bbox_transform


def new_all_boxes(num_classes, num_images):
    """all_boxes[cls][image] = list of [xmin, ymin, xmax, ymax, score] (eval.py:69-70)."""
    return [[[] for _ in range(num_images)] for _ in range(num_classes)]


def add_detections(all_boxes, image_index, det_bbox, score, det_class, scale=(1.0, 1.0)):
    """eval.py:81-91 for one image: det_bbox are (cx,cy,w,h) in network-input pixels as returned by
    model.filter_prediction; `scale` = (x_scale, y_scale) of imdb.read_image_batch (network / original size).
    The reference rescales det_boxes before filtering (:83-84); filtering is scale-free for a uniform
    per-axis scale except for the IoU, so callers that need bit-equal picks rescale first and pass scale=(1,1)."""
    for c, b, s in zip(det_class, det_bbox, score):
        b = np.asarray(b, dtype=np.float64).copy()
        b[0::2] /= scale[0]
        b[1::2] /= scale[1]
        all_boxes[int(c)][image_index].append(list(bbox_transform(b)) + [float(s)])
    return all_boxes


def write_detection_files(det_file_dir, image_idx, class_names, all_boxes):
    """kitti.py:111-127: one '<index>.txt' per image, one line per detection:
    '<cls> -1 -1 0.0 x1 y1 x2 y2 0.0 0.0 0.0 0.0 0.0 0.0 0.0 score' with %.2f coordinates and %.3f score."""
    os.makedirs(det_file_dir, exist_ok=True)
    for im_idx, index in enumerate(image_idx):
        with open(os.path.join(det_file_dir, index + ".txt"), "wt") as f:
            for cls_idx, cls in enumerate(class_names):
                for d in all_boxes[cls_idx][im_idx]:
                    f.write("{:s} -1 -1 0.0 {:.2f} {:.2f} {:.2f} {:.2f} 0.0 0.0 0.0 0.0 0.0 "
                            "0.0 0.0 {:.3f}\n".format(cls.lower(), d[0], d[1], d[2], d[3], d[4]))


def evaluate_detections(eval_tool, data_root_path, image_set, eval_dir, global_step, image_idx, class_names, all_boxes):
    """kitti.py:100-159: writes the detection files under eval_dir/detection_files_<step>/data, runs
    `<eval_tool> <data_root>/training <data_root>/ImageSets/<image_set>.txt <det dir> <N>` and returns
    (aps, names) = easy / medium / hard AP per class from the evaluator's stats_<cls>_ap.txt files."""
    det_file_dir = os.path.join(eval_dir, "detection_files_{:s}".format(str(global_step)), "data")
    write_detection_files(det_file_dir, image_idx, class_names, all_boxes)
    cmd = [eval_tool, os.path.join(data_root_path, "training"), os.path.join(data_root_path, "ImageSets", image_set + ".txt"),
           os.path.dirname(det_file_dir), str(len(image_idx))]
    print("Running: {}".format(" ".join(cmd)))
    subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    aps, names = [], []
    for cls in class_names:
        fn = os.path.join(os.path.dirname(det_file_dir), "stats_{:s}_ap.txt".format(cls))
        if os.path.exists(fn):
            with open(fn) as f:
                lines = f.readlines()
            assert len(lines) == 3, "Line number of {} should be 3".format(fn)
            aps.extend(float(line.split("=")[1].strip()) for line in lines)
        else:
            aps.extend([0.0, 0.0, 0.0])
        names.extend([cls + "_easy", cls + "_medium", cls + "_hard"])
    return aps, names
