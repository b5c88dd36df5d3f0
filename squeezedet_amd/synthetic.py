"""Seeded synthetic inputs for benchmarks and demos (there is no network for datasets or
checkpoints): KITTI-shaped images as demo.py:187-190 feeds them, and random-init weights of
the reference architecture (truncated normal like nn_skeleton.py:527-528 but
He-scaled so activations do not collapse and scores spread)."""
import math

import numpy as np
import torch


def synthetic_images(batch, img_h, img_w, seed=0, bgr_means=(103.939, 116.779, 123.68)):
    """float32 [B,H,W,3]: U{0..255} - BGR_MEANS (config/config.py:72)."""
    rng = np.random.RandomState(seed)
    im = rng.randint(0, 256, size=(batch, img_h, img_w, 3)).astype(np.float32)
    return torch.from_numpy((im - np.asarray(bgr_means)).astype(np.float32))


def synthetic_params(model, seed=0):
    """{name: float32 tensor} for every parameter of `model` (HWIO kernels, biases)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, p in model.params.items():
        shp = tuple(p.shape)
        if name.endswith("/kernels"):
            sigma = math.sqrt(2.0 / (shp[0] * shp[1] * shp[2]))
            if name.startswith("conv1/"):
                sigma /= 64.0   # inputs are pixel-scale (rms ~74): bring activations to O(1)
            if name.startswith("conv12"):
                sigma *= 2.0    # preds of std ~2: scores spread without saturating
            if name.startswith("conv5/"):
                # ResNet50+ConvDet: the residual stream reaches conv5 with std ~7 under these synthetic BN statistics
                # (measured); He x 2 gave preds of std 24 -- box deltas of +-50, exp() of them, a bbox loss of 6e4 and
                # float16 gradient overflow on EVERY step (all updates skipped).  He / 6 brings preds back to std ~2.
                sigma /= 6.0
            z = np.clip(rng.standard_normal(size=shp), -2.0, 2.0)
            out[name] = torch.from_numpy((z * sigma).astype(np.float32))
        elif name.endswith("/gamma") or name.endswith("/var"):
            # frozen batch-norm statistics / scales of _conv_bn_layer: positive, O(1); the last conv of a
            # residual branch is scaled down so the residual stream stays O(1) through the 13 blocks
            v = rng.uniform(0.5, 1.5, size=shp) * (0.3 if name.endswith("_branch2c/gamma") else 1.0)
            out[name] = torch.from_numpy(v.astype(np.float32))
        else:
            out[name] = torch.from_numpy(rng.uniform(-0.1, 0.1, size=shp).astype(np.float32))
    return out


# ---------------------------------------------------------------------------------------------------------------
# "Planted-object" detection head: random backbone weights give 16 848 anchor scores that form a near-continuum (the
# 64th and 65th differ by ~1e-4), so no float16 run can be compared pick-for-pick with another implementation.  This head
# makes the DISCRETE outputs of the path (which anchors enter the top-N, their order, classes, the NMS survivors)
# decidable, the way a trained detector's are: a handful of cells per image fire, everything else sits at one exact
# background level.
#   * three "detector" channels -- channels 0..2 of fire11's concat output -- get a bias shift so that only the top
#     ~0.135 % of cells stay positive after the ReLU (exact zeros elsewhere), and a x16 gain (a power of two: exact in
#     float16) so that the confidence gain below stays inside float16's range;
#   * conv12 is zero except: confidence logit of anchor shape k (channel K*C + k) = -6 + G * detector[k % C] at the
#     centre tap, G a power of two large enough that a firing cell saturates the sigmoid to exactly 1.0f; class logits
#     and box deltas are bias-only (class k % C wins, with a different margin per shape, so the nine shapes score at nine
#     separated levels; deltas are small float16-exact constants).
# A firing (cell, class c) plants the three anchor shapes c, c+3, c+6 of that cell (~20 anchors per image at z = 3);
# the background anchors tie exactly within a shape and are ranked by the repo's tie rule (higher anchor index first).
PLANT_TAIL = 1.35e-3          # fraction of (cell, detector channel) pairs that fire: z = 3 of a Gaussian
PLANT_CONF_BIAS = -6.0
PLANT_DET_SCALE = 16.0


def planted_stats(det_activations):
    """det_activations: [N,h,w,>=3] array of fire11's (post-ReLU) output under the ORIGINAL biases on a few calibration
    images.  Returns [(q_c, e_c)] per detector channel: q_c = the (1 - PLANT_TAIL) quantile (the bias shift), e_c = the
    distance to the (1 - PLANT_TAIL / 3) quantile (the scale of the exceedances, which sizes the confidence gain)."""
    a = np.asarray(det_activations, dtype=np.float64)
    stats = []
    for c in range(3):
        v = np.sort(a[..., c].ravel())
        n = len(v)
        q = v[min(n - 1, int(n * (1.0 - PLANT_TAIL)))]
        q2 = v[min(n - 1, int(n * (1.0 - PLANT_TAIL / 3.0)))]
        if not (q > 0 and q2 > q):
            raise ValueError("planted_stats: detector channel %d has no positive tail to calibrate on" % c)
        stats.append((float(np.float32(q)), float(np.float32(q2 - q))))
    return stats


def planted_head(params, stats, anchors_per_grid=9, classes=3):
    """params with the planted head installed (a new dict; untouched tensors are shared)."""
    K, C = int(anchors_per_grid), int(classes)
    p = dict(params)
    b11 = p["fire11/expand1x1/biases"].clone().float()
    w11 = p["fire11/expand1x1/kernels"].clone().float()
    w12 = torch.zeros_like(p["conv12/kernels"], dtype=torch.float32)
    b12 = torch.zeros(K * (C + 5), dtype=torch.float32)
    for c, (q, e) in enumerate(stats):
        b11[c] = (b11[c] - q) * PLANT_DET_SCALE
        w11[..., c] *= PLANT_DET_SCALE
        gain = 2.0 ** int(round(math.log2(23.0 / (0.02 * e * PLANT_DET_SCALE))))     # sigmoid(-6 + gain * y) == 1.0f once y > 2 % of e
        if not gain <= 32768.0:
            raise ValueError("planted_head: gain %g does not fit float16 (detector activations too small)" % gain)
        for k in range(c, K, C):
            w12[1, 1, c, K * C + k] = gain
    for k in range(K):
        b12[k * C + (k % C)] = 1.0 + 0.25 * k                          # softmax level of shape k: 0.58 .. 0.91
        b12[K * C + k] = PLANT_CONF_BIAS
        d = K * (C + 1) + 4 * k
        b12[d:d + 4] = torch.tensor([0.125 * ((k % 3) - 1), 0.0625 * ((k % 2) * 2 - 1), 0.25 * (k % 2), -0.125 * (k % 3)])
    p["fire11/expand1x1/biases"] = b11
    p["fire11/expand1x1/kernels"] = w11
    p["conv12/kernels"] = w12
    p["conv12/biases"] = b12
    return p
