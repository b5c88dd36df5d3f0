"""Loader for the NumPy half of the reference, imported UNCHANGED.

Only usable where ``/root/reference`` is mounted (the build container); the GPU
box has no reference tree, so nothing reachable from ``-m gpu`` tests,
``smoke()`` or ``bench.py`` may call this.  It is used by
``tests/golden/make_golden.py`` (to generate committed fixtures) and by the
``not gpu`` tests that cross-check the restatement when the tree is present.

Recipe: the reference is Python 2 + TF 1.0; the NumPy
half imports under py3.10 with two stubs -- ``easydict.EasyDict`` and
``tensorflow.variable_scope`` -- and ``src/config`` ahead of ``src`` on
``sys.path`` so ``from config import base_model_config``
(src/config/kitti_squeezeDet_config.py:7) resolves to ``config.py``.
"""
import contextlib
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SQDET_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "config"))


class _EasyDict(dict):
    """Stand-in for easydict==1.6: attribute access on a dict."""
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


_loaded = None


def load():
    """Returns a namespace with the reference's own modules:
    ``.util`` (src/utils/util.py), ``.nn_skeleton`` (src/nn_skeleton.py),
    ``.cfg_squeezeDet`` / ``.cfg_squeezeDetPlus`` / ``.cfg_res50`` /
    ``.cfg_vgg16`` (src/config/kitti_*_config.py)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the reference mount is read-only
    saved_path = list(sys.path)
    saved_mods = {k: sys.modules.get(k) for k in ("easydict", "tensorflow", "config", "utils", "nn_skeleton")}
    try:
        ed = types.ModuleType("easydict")
        ed.EasyDict = _EasyDict
        sys.modules["easydict"] = ed
        tf = types.ModuleType("tensorflow")
        tf.variable_scope = lambda *a, **k: contextlib.nullcontext()
        sys.modules["tensorflow"] = tf
        for k in ("config", "utils", "nn_skeleton"):
            sys.modules.pop(k, None)
        sys.path.insert(0, os.path.join(REFERENCE_ROOT, "src"))
        sys.path.insert(0, os.path.join(REFERENCE_ROOT, "src", "config"))
        ns = types.SimpleNamespace()
        ns.cfg_squeezeDet = importlib.import_module("kitti_squeezeDet_config")
        ns.cfg_squeezeDetPlus = importlib.import_module("kitti_squeezeDetPlus_config")
        ns.cfg_res50 = importlib.import_module("kitti_res50_config")
        ns.cfg_vgg16 = importlib.import_module("kitti_vgg16_config")
        ns.util = importlib.import_module("utils.util")
        ns.nn_skeleton = importlib.import_module("nn_skeleton")
        _loaded = ns
        return ns
    finally:
        sys.path[:] = saved_path
        for k, v in saved_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def filter_prediction(mc, boxes, probs, cls_idx):
    """The reference's ModelSkeleton.filter_prediction (nn_skeleton.py:696-734),
    called unbound with a stand-in ``self`` carrying only ``mc``."""
    ns = load()
    return ns.nn_skeleton.ModelSkeleton.filter_prediction(
        types.SimpleNamespace(mc=mc), boxes, probs, cls_idx)
