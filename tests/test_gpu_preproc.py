"""GPU parity of sqdet_preprocess_bgr (demo.py:186-190 image preparation) against
oracle/preproc_oracle.py, and the demo script end to end on a synthetic image file."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import preproc_oracle as PO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEANS = [103.939, 116.779, 123.68]


@pytest.mark.parametrize("case", [((375, 1242), (384, 1248)), ((375, 1242), (375, 1242)), ((480, 640), (384, 1248)), ((37, 53), (96, 160))],
                         ids=["kitti->net", "identity", "vga->net", "small-up"])
def test_preprocess_matches_oracle(case):
    from squeezedet_amd import ops
    (hs, ws), (hd, wd) = case
    rs = np.random.RandomState(hs + wd)
    im = rs.randint(0, 256, size=(2, hs, ws, 3)).astype(np.uint8)
    ref = np.stack([PO.preprocess_bgr(im[i], hd, wd, MEANS) for i in range(2)])
    out = ops.preprocess_bgr(torch.from_numpy(im).to(DEV), hd, wd, MEANS, torch.float32).cpu().numpy()
    # same float32 operations in the same order (no FMA contraction): equal up to the last bit of the coordinate maths
    assert np.abs(out - ref).max() <= 2e-4
    assert (out == ref).mean() > 0.99
    h16 = ops.preprocess_bgr(torch.from_numpy(im).to(DEV), hd, wd, MEANS, torch.float16)
    assert h16.dtype == torch.float16 and np.abs(h16.float().cpu().numpy() - ref).max() <= 0.07   # fp16 rounding of +-150


def test_demo_script_end_to_end(tmp_path):
    """demo.py: file -> GPU preprocessing -> SqueezeDet (synthetic weights) -> filter_prediction -> annotated file."""
    from PIL import Image
    rs = np.random.RandomState(5)
    src = tmp_path / "img.png"
    Image.fromarray(rs.randint(0, 256, size=(375, 1242, 3)).astype(np.uint8)).save(src)
    out_dir = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo.py"), "--input_path", str(src), "--out_dir", str(out_dir)],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Image detection output saved to" in r.stdout
    with Image.open(out_dir / "out_img.png") as im:
        assert im.size == (1248, 384)
