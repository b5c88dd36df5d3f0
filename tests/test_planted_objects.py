"""CPU check of the planted-object generator (squeezedet_amd/synthetic.py) against the oracle: through the oracle's own conv / pool /
decode restatement the anchors that score above 0.5 are EXACTLY the planted (cell, shape) triples, at the nine separated levels, in
float32 and in float16-storage mode alike -- i.e. each object reaches exactly one cell of the final grid, nothing else fires, and the
detector channels are exact.  (The GPU side of the same images: tests/test_gpu_model.py
test_fp16_headline_path_picks_identical_planted_objects.)"""
import numpy as np
import pytest

from oracle import sqdet_oracle as O
from squeezedet_amd import synthetic as SY


@pytest.mark.parametrize("size", [(375, 1242), (384, 1248)], ids=["375x1242", "384x1248"])
@pytest.mark.parametrize("storage", ["fp32", "fp16"])
def test_planted_objects_fire_exactly_their_anchors(size, storage):
    omc = O.squeezeDet_config_for_input(*size)
    gw = omc.ANCHORS // omc.ANCHOR_PER_GRID // 24
    params = SY.planted_params(O.init_params("squeezeDet", seed=40, storage=storage), omc.ANCHOR_PER_GRID, omc.CLASSES)
    x, planted = SY.planted_images(omc, 2, seed=41)
    assert float(x.abs().max()) <= 144.0 and bool((x == x.round()).all())          # integer-valued: exact in float16
    preds, out, dets = O.detect("squeezeDet", omc, params, x, storage=storage)
    for b in range(2):
        want = sorted((cy * gw + cx) * 9 + k for (bb, cy, cx, c) in planted if bb == b for k in range(c, 9, 3))
        assert len(want) == 36
        p = out["det_probs"][b]
        got = sorted(np.nonzero(p > 0.5)[0].tolist())
        assert got == want
        assert len(set(np.round(p[got], 5))) == 9                                  # nine score levels
        assert float(np.max(np.delete(p, got))) < 0.003                             # the background: sigmoid(-6) * softmax level
        cls = out["det_class"][b][got]
        assert all(int(c) == (a % 9) % 3 for a, c in zip(got, cls))
        assert 8 <= len(dets[b][3]) <= 64


@pytest.mark.parametrize("storage", ["fp32", "fp16"])
def test_planted_objects_squeezedet_plus_geometry(storage):
    """The same generator on SqueezeDet+'s geometry (nets/squeezeDetPlus.py:40-61: 7x7/s2 VALID conv1, VALID pools, 22x76 grid,
    15048 anchors): 7x7 objects, detector threshold 6.0 -- exactly the planted (cell, shape) triples fire."""
    omc = O.kitti_squeezeDetPlus_config()
    gh, gw = SY._grid(omc, "squeezeDet+")
    assert (gh, gw) == (22, 76) and omc.ANCHORS == gh * gw * 9
    params = SY.planted_params(O.init_params("squeezeDet+", seed=40, storage=storage), omc.ANCHOR_PER_GRID, omc.CLASSES, arch="squeezeDet+")
    x, planted = SY.planted_images(omc, 2, seed=41, arch="squeezeDet+")
    assert float(x.abs().max()) <= 144.0 and bool((x == x.round()).all())
    preds, out, dets = O.detect("squeezeDet+", omc, params, x, storage=storage)
    for b in range(2):
        want = sorted((cy * gw + cx) * 9 + k for (bb, cy, cx, c) in planted if bb == b for k in range(c, 9, 3))
        assert len(want) == 36
        p = out["det_probs"][b]
        got = sorted(np.nonzero(p > 0.5)[0].tolist())
        assert got == want
        assert len(set(np.round(p[got], 5))) == 9
        assert float(np.max(np.delete(p, got))) < 0.003
        assert 8 <= len(dets[b][3]) <= 64
