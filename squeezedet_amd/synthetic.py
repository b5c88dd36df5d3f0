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
