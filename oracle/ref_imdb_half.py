"""Loader for the reference's label-assignment code, imported UNCHANGED: ``imdb.read_batch``
(/root/reference/src/dataset/imdb.py:120-260 -- every ground-truth box, in order, claims the free anchor of highest IoU,
or the nearest free anchor when nothing overlaps; then the (dx, dy, dw, dh) targets).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Only usable where ``/root/reference`` is mounted (the build
container): it is called by ``tests/golden/make_golden.py`` to produce the committed ``tests/golden/labels.npz`` and by
the ``not gpu`` cross-check; nothing reachable from ``-m gpu`` tests, ``smoke()`` or ``bench.py`` imports it.

The module parses under Python 3; what it cannot have here is OpenCV, and ``read_batch`` only uses ``cv2`` to fetch
the pixels (``cv2.imread``, ``cv2.resize``), which the label half never looks at.  So ``cv2`` is replaced by a stub
whose ``imread`` returns a zero image of the size registered for the path (the ORIGINAL image size enters the labels
through x_scale / y_scale, imdb.py:186-190) and whose ``resize`` returns a zero image of the requested size.
"""
import importlib.util
import os
import sys
import types

import numpy as np

from . import ref_numpy_half as ref

_imdb_mod = None
_sizes = {}   # path -> (h, w) of the "file" cv2.imread pretends to read


def _load():
    global _imdb_mod
    if _imdb_mod is not None:
        return _imdb_mod
    ns = ref.load()   # (also checks that the tree is there)
    sys.dont_write_bytecode = True
    saved_path = list(sys.path)
    saved = {k: sys.modules.get(k) for k in ("cv2", "utils", "utils.util", "dataset", "dataset.imdb")}
    try:
        cv2 = types.ModuleType("cv2")
        cv2.imread = lambda path: np.zeros(_sizes[path] + (3,), np.uint8)
        cv2.resize = lambda im, size: np.zeros((size[1], size[0], 3), np.float32)
        sys.modules["cv2"] = cv2
        # the reference's own utils.util (already imported unchanged by ref_numpy_half) serves `from utils.util import ...`
        pkg = types.ModuleType("utils")
        pkg.util = ns.util
        sys.modules["utils"] = pkg
        sys.modules["utils.util"] = ns.util
        spec = importlib.util.spec_from_file_location("sqdet_ref_imdb", os.path.join(ref.REFERENCE_ROOT, "src", "dataset", "imdb.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _imdb_mod = mod
        return mod
    finally:
        sys.path[:] = saved_path
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def read_batch(mc, rois, orig_sizes):
    """Runs the reference's ``imdb.read_batch(shuffle=False)`` on synthetic annotations.
    rois: per image a list of [cx, cy, w, h, class] in ORIGINAL image coordinates (what kitti._load_kitti_annotation
    stores in ``_rois``, dataset/kitti.py:50-78); orig_sizes: per image (height, width) of the original image.
    Returns the reference's (label_per_batch, delta_per_batch, aidx_per_batch, bbox_per_batch)."""
    mod = _load()
    n = len(rois)
    mc = type(mc)(mc)          # EasyDict copy: the caller's config is not modified
    mc.BATCH_SIZE = n
    mc.DATA_AUGMENTATION = False
    mc.DEBUG_MODE = False
    db = mod.imdb("synthetic", mc)
    db._image_idx = ["%06d" % i for i in range(n)]
    db._rois = {idx: [list(r) for r in rois[i]] for i, idx in enumerate(db._image_idx)}
    db._image_path_at = lambda idx: "synthetic/%s.png" % idx
    _sizes.clear()
    for i, idx in enumerate(db._image_idx):
        _sizes["synthetic/%s.png" % idx] = (int(orig_sizes[i][0]), int(orig_sizes[i][1]))
    db._cur_idx = -1           # non-shuffled branch: [_cur_idx + BATCH_SIZE >= len] wraps; start so the batch is images 0..n-1
    # imdb.py:146-151: with _cur_idx + B >= len the batch is idx[cur:] + idx[:cur + B - len]; cur = 0 gives all of them
    db._cur_idx = 0
    _, labels, deltas, aidx, bboxes = db.read_batch(shuffle=False)
    return labels, deltas, aidx, bboxes
