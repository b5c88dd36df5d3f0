"""Pins oracle/preproc_oracle.py (the cv2.resize INTER_LINEAR + mean-subtract restatement)."""
import numpy as np

from oracle import preproc_oracle as PO


def test_identity_and_exact_cases():
    rs = np.random.RandomState(0)
    im = rs.randint(0, 256, size=(7, 9, 3)).astype(np.float32)
    assert np.array_equal(PO.resize_linear(im, 7, 9), im)                       # same size: sx = x, fx = 0
    const = np.full((5, 6, 3), 37.0, np.float32)
    assert np.array_equal(PO.resize_linear(const, 11, 13), np.full((11, 13, 3), 37.0, np.float32))
    # 2x magnification of a linear ramp: interior samples lie on the ramp at the half-pixel-centre positions
    ramp = np.tile(np.arange(8, dtype=np.float32)[None, :, None], (4, 1, 3))
    up = PO.resize_linear(ramp, 4, 16)
    want = np.clip((np.arange(16) + 0.5) / 2 - 0.5, 0, 7)
    assert np.allclose(up[0, :, 0], want, atol=1e-6)
    out = PO.preprocess_bgr(rs.randint(0, 256, size=(4, 5, 3)).astype(np.uint8), 4, 5, [103.939, 116.779, 123.68])
    assert out.dtype == np.float32 and out.shape == (4, 5, 3)


def test_matches_pil_bilinear_for_magnification():
    """KITTI 1242x375 -> the network's 1248x384 (demo.py:189) is a magnification, where PIL's BILINEAR (a
    triangle filter of support 1 at half-pixel centres) is the same function as cv2's INTER_LINEAR."""
    from PIL import Image
    rs = np.random.RandomState(1)
    src = rs.randint(0, 256, size=(75, 124)).astype(np.float32)
    got = PO.resize_linear(src[:, :, None], 96, 156)[:, :, 0]
    ref = np.asarray(Image.fromarray(src, mode="F").resize((156, 96), Image.BILINEAR), dtype=np.float32)
    assert np.abs(got - ref).max() < 5e-3          # values up to 255: float32 rounding-order differences only
