"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the image preparation of the
reference's callers -- src/demo.py:186-190 (`im.astype(np.float32)`, `cv2.resize(im, (W, H))`,
`im - mc.BGR_MEANS`) and src/dataset/imdb.py:101-118.  cv2 is not installable here (opencv-python==3.2.0.6 is
pinned in requirements.txt and absent), so cv2.resize's INTER_LINEAR float32 path is restated from its
published algorithm: half-pixel-centre source coordinates, borders clamped, horizontal pass then vertical
pass in float32.  Pinned in tests/test_oracle_preproc.py against PIL's bilinear resize (which coincides with it
for magnification) and against exact cases (identity, integer up-scaling of constant / linear ramps).
"""
import numpy as np


def _coords(n_dst, n_src):
    # cv::resize forms the source coordinate in double -- (float)((dx + 0.5) * scale - 0.5), scale = (double)src / dst --
    # and rounds it once to float32
    scale = float(n_src) / float(n_dst)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0
    hi = s >= n_src - 1
    s[hi], f[hi] = n_src - 1, 0
    s1 = np.minimum(s + 1, n_src - 1)
    return s, s1, f


def resize_linear(im_f32, dst_h, dst_w):
    """cv2.resize(im, (dst_w, dst_h)) for float32 HxWxC, INTER_LINEAR."""
    im = np.asarray(im_f32, dtype=np.float32)
    sy, sy1, fy = _coords(dst_h, im.shape[0])
    sx, sx1, fx = _coords(dst_w, im.shape[1])
    ax0 = (np.float32(1) - fx)[None, :, None]
    fxb = fx[None, :, None]
    h0 = im[sy][:, sx] * ax0 + im[sy][:, sx1] * fxb          # horizontal pass on the two source rows
    h1 = im[sy1][:, sx] * ax0 + im[sy1][:, sx1] * fxb
    ay0 = (np.float32(1) - fy)[:, None, None]
    return (h0 * ay0 + h1 * fy[:, None, None]).astype(np.float32)


def preprocess_bgr(im_u8, dst_h, dst_w, bgr_means):
    """demo.py:186-190 after imread: float32 cast, resize to the network input, subtract mc.BGR_MEANS."""
    im = np.asarray(im_u8).astype(np.float32)
    return (resize_linear(im, dst_h, dst_w) - np.asarray(bgr_means, dtype=np.float32).reshape(1, 1, 3)).astype(np.float32)
