"""Seeded synthetic inputs for benchmarks and demos (there is no network for datasets or
checkpoints): KITTI-shaped images as demo.py:187-190 feeds them, and random-init weights of
the reference architecture (truncated normal like nn_skeleton.py:527-528 but
He-scaled so activations do not collapse and scores spread)."""
import functools
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
# "Planted objects": a detection problem whose DISCRETE outputs (which anchors enter the top-N, their order, their classes,
# the NMS survivors) have real margins, the way a trained detector's have -- random weights give 16 848 anchor scores that
# form a near-continuum (the 64th and 65th differ by ~1e-4), so no float16 run can be compared pick-for-pick with another
# implementation.  (Round 3 thresholded three random feature channels of fire11: a threshold on a continuum always has
# cells within float16 noise of it, and 4 of 16 images flipped one.  Now the objects are in the IMAGE and the channels that
# carry them are exact.)
#   * images (planted_images): a low-amplitude integer background with a few 3x3 "objects" -- colour channel c at +144, the
#     others at -100 -- placed on conv1 windows chosen so that, through conv1's stride and the three 3x3/s2 max-pools, each
#     object reaches EXACTLY ONE cell of the final grid;
#   * three detector channels (planted_params): conv1 filter c sums colour c over its window with weight 2^-10 (every
#     product and partial sum is a multiple of 2^-10 below 2^14: exact in float32 in ANY summation order, so device and
#     oracle round the same value to float16); channel c of every squeeze1x1 / expand1x1 from fire2 to fire10 passes it on
#     with a single weight 1.0 and zero bias (exact), the max-pools are exact, and fire11/expand1x1 channel c is
#     relu(16 * (x - 0.875)): an object reads 1.265625 -> 6.25, the background (<= 0.14) and every partial window overlap
#     (<= 0.52) are exact zeros.  Every other channel of the backbone keeps its random weights (and also reads the detector
#     channels): the forward is the benchmark's, the decisions ride on three channels of it through EVERY launch;
#   * head: conv12 is zero except the confidence logit of anchor shape k = -6 + 4 * detector[k % 3] at the centre tap
#     (19 -> sigmoid == 1.0f exactly); class logits and box deltas are bias-only (class k % 3 wins with a different margin
#     per shape: nine separated score levels).  An object of colour c plants the anchor shapes c, c+3, c+6 of its cell; all
#     other anchors tie EXACTLY at their shape's background level and are ranked by the tie rule (higher index first);
#   * the object layout of an image is rejection-sampled so that no two planted boxes of a class have an IoU within
#     PLANT_IOU_MARGIN of NMS_THRESH (the boxes are a closed form of cell, shape and the constant deltas).
PLANT_W = 2.0 ** -10
PLANT_OBJ, PLANT_OFF, PLANT_BG = 144.0, -100.0, 16          # object colour / other colours / background amplitude (integers)
PLANT_DET_SCALE = 16.0
PLANT_CONF_BIAS, PLANT_CONF_GAIN = -6.0, 4.0
PLANT_IOU_MARGIN = 0.04
FIRES = ["fire%d" % i for i in range(2, 12)]
# Per architecture: conv1 (size, stride, padding), the padding of its three 3x3/s2 max-pools, the object's side (= conv1's window)
# and the detector threshold q of fire11/expand1x1 (relu(16 * (x - q))).
#   squeezeDet  (nets/squeezeDet.py:40-57):     3x3/s2 SAME, SAME pools; a 3x3 object reads 9*144*2^-10 = 1.265625, every partial
#                                               window overlap <= 0.52 -> q = 0.875;
#   squeezeDet+ (nets/squeezeDetPlus.py:40-61): 7x7/s2 VALID, VALID pools; a 7x7 object reads 49*144*2^-10 = 6.890625, the largest
#                                               partial overlap (a window two pixels off: 35 object pixels + 14 of background) <= 5.15
#                                               -> q = 6.0 (detector 14.25 on an object, exact zero elsewhere).
PLANT_GEOM = {"squeezeDet": dict(conv=(3, 2, "SAME"), pools=("SAME",) * 3, q=0.875),
              "squeezeDet+": dict(conv=(7, 2, "VALID"), pools=("VALID",) * 3, q=6.0)}
PLANT_Q = PLANT_GEOM["squeezeDet"]["q"]


def _geom(n, k=3, s=2, padding="SAME"):
    """TF geometry of one conv / pool dimension: (output size, pad before)."""
    if padding == "VALID":
        return (n - k) // s + 1, 0
    o = -(-n // s)
    return o, max((o - 1) * s + k - n, 0) // 2


def _same_geom(n, k=3, s=2):
    return _geom(n, k, s, "SAME")


@functools.lru_cache(maxsize=None)
def _single_cell_positions(n_img, arch="squeezeDet"):
    """Per final-grid index c (one image dimension of n_img pixels): the conv1 output positions whose value reaches final cell c
    and NO other through the three 3x3/s2 max-pools, and whose conv1 window lies inside the image.  Returns
    ({c: [positions]}, conv1 pad before, final size)."""
    g = PLANT_GEOM[arch]
    ck, cs, cpad = g["conv"]
    n1, p0 = _geom(n_img, ck, cs, cpad)
    sizes, pads = [n1], []
    for mode in g["pools"]:
        o, p = _geom(sizes[-1], 3, 2, mode)
        sizes.append(o)
        pads.append(p)
    out = {}
    for pos in range(n1):
        if not (cs * pos - p0 >= 0 and cs * pos - p0 + ck - 1 < n_img):
            continue
        reach = {pos}
        for lvl in range(3):
            nxt = set()
            for q in reach:
                for i in range(sizes[lvl + 1]):
                    if 2 * i - pads[lvl] <= q <= 2 * i - pads[lvl] + 2:
                        nxt.add(i)
            reach = nxt
        if len(reach) == 1:
            out.setdefault(next(iter(reach)), []).append(pos)
    return out, p0, sizes[-1]


def _grid(mc, arch="squeezeDet"):
    """(gh, gw) of the final map of `arch` on mc's image size."""
    return _single_cell_positions(int(mc.IMAGE_HEIGHT), arch)[2], _single_cell_positions(int(mc.IMAGE_WIDTH), arch)[2]


def _planted_boxes(mc, cell_y, cell_x, c, shapes=None, arch="squeezeDet"):
    """[3, 4] (cx, cy, w, h) of the anchors an object of colour c plants at a cell (or of the given shapes), as
    interpret_output decodes them (nn_skeleton.py:175-215: delta decode, clip to the image, back to centre form)."""
    K = mc.ANCHOR_PER_GRID
    gw = _grid(mc, arch)[1]
    assert len(mc.ANCHOR_BOX) == K * gw * _grid(mc, arch)[0], "mc.ANCHOR_BOX does not belong to the %s grid" % arch
    out = []
    for k in (shapes if shapes is not None else range(c, K, 3)):
        ax, ay, aw, ah = [float(v) for v in mc.ANCHOR_BOX[(cell_y * gw + cell_x) * K + k]]
        dx, dy, dw, dh = planted_deltas(k)
        cx, cy, w, h = ax + dx * aw, ay + dy * ah, aw * math.exp(dw), ah * math.exp(dh)
        xmin = min(max(0.0, cx - w / 2), mc.IMAGE_WIDTH - 1.0)
        ymin = min(max(0.0, cy - h / 2), mc.IMAGE_HEIGHT - 1.0)
        xmax = max(min(mc.IMAGE_WIDTH - 1.0, cx + w / 2), 0.0)
        ymax = max(min(mc.IMAGE_HEIGHT - 1.0, cy + h / 2), 0.0)
        ww, hh = xmax - xmin + 1.0, ymax - ymin + 1.0
        out.append([xmin + 0.5 * ww, ymin + 0.5 * hh, ww, hh])
    return np.asarray(out, np.float64)


def _iou(a, b):
    lr = min(a[0] + a[2] / 2, b[0] + b[2] / 2) - max(a[0] - a[2] / 2, b[0] - b[2] / 2)
    tb = min(a[1] + a[3] / 2, b[1] + b[3] / 2) - max(a[1] - a[3] / 2, b[1] - b[3] / 2)
    if lr <= 0 or tb <= 0:
        return 0.0
    inter = lr * tb
    return inter / (a[2] * a[3] + b[2] * b[3] - inter)


def planted_deltas(k):
    """The constant (float16-exact) box deltas of anchor shape k.  Shape 8 has the highest background score level: its
    highest-index anchors fill the top-N behind the planted ones, and ITS deltas are chosen so that those neighbouring boxes keep
    every mutual IoU (and the IoU with the other two shapes of an object of colour 2) >= 0.07 away from NMS_THRESH at both benchmark sizes."""
    if k == 8:
        return (-0.125, -0.125, -0.25, -0.375)
    return (0.125 * ((k % 3) - 1), 0.0625 * ((k % 2) * 2 - 1), 0.25 * (k % 2), -0.125 * (k % 3))


def planted_images(mc, batch, seed=0, objects_per_colour=4, arch="squeezeDet"):
    """float32 [B,H,W,3] integer-valued images (exact in float16) with `objects_per_colour` objects of each of the three
    colours per image, and the list of planted (image, cell_y, cell_x, colour).  arch: whose conv1 / pool geometry the objects
    are placed for (PLANT_GEOM); an object is one conv1 window wide."""
    rng = np.random.RandomState(seed)
    H, W = int(mc.IMAGE_HEIGHT), int(mc.IMAGE_WIDTH)
    side, stride = PLANT_GEOM[arch]["conv"][:2]
    rows, py0, gh = _single_cell_positions(H, arch)
    cols, px0, gw = _single_cell_positions(W, arch)
    x = rng.randint(-PLANT_BG, PLANT_BG + 1, size=(batch, H, W, 3)).astype(np.float32)
    planted = []
    # the top-N is filled, behind the 9 * objects_per_colour planted anchors, with the highest-index anchors of shape 8 (the
    # highest background level; class 2): their boxes take part in class 2's NMS and in its IoU margins
    K = mc.ANCHOR_PER_GRID
    nfill = max(0, mc.TOP_N_DETECTION - 3 * 3 * objects_per_colour)
    fill = [_planted_boxes(mc, cell // gw, cell % gw, 2, shapes=[K - 1], arch=arch)[0] for cell in range(gh * gw - 1, gh * gw - 1 - nfill, -1)]
    for b in range(batch):
        for attempt in range(400):
            cells, boxes, ok = [], {0: [], 1: [], 2: list(fill)}, True
            for c in range(3):
                for _ in range(objects_per_colour):
                    cy, cx = sorted(rows)[rng.randint(len(rows))], sorted(cols)[rng.randint(len(cols))]
                    if any(abs(cy - oy) < 2 and abs(cx - ox) < 2 for oy, ox, _ in cells):
                        ok = False              # (two objects never share or touch a cell: no window sees two of them)
                    cells.append((cy, cx, c))
                    boxes[c].extend(_planted_boxes(mc, cy, cx, c, arch=arch))
            for c in range(3):
                bl = boxes[c]
                for i in range(len(bl)):
                    for j in range(i + 1, len(bl)):
                        if abs(_iou(bl[i], bl[j]) - mc.NMS_THRESH) < PLANT_IOU_MARGIN:
                            ok = False
            if ok:
                break
        else:
            raise RuntimeError("planted_images: no object layout with IoU margins found for image %d" % b)
        for cy, cx, c in cells:
            y1, x1 = rows[cy][rng.randint(len(rows[cy]))], cols[cx][rng.randint(len(cols[cx]))]
            iy, ix = stride * y1 - py0, stride * x1 - px0
            x[b, iy:iy + side, ix:ix + side, :] = PLANT_OFF
            x[b, iy:iy + side, ix:ix + side, c] = PLANT_OBJ
            planted.append((b, cy, cx, c))
    return torch.from_numpy(x), planted


def planted_params(params, anchors_per_grid=9, classes=3, arch="squeezeDet"):
    """params with the three detector channels and the planted head installed (a new dict; untouched tensors are shared)."""
    K, C = int(anchors_per_grid), int(classes)
    q = PLANT_GEOM[arch]["q"]
    assert C == 3, "three colours <-> three classes"
    p = dict(params)

    def upd(name, fn):
        t = p[name].clone().float()
        fn(t)
        p[name] = t
    def conv1_k(t):
        t[:, :, :, :3] = 0.0
        for c in range(3):
            t[:, :, c, c] = PLANT_W
    upd("conv1/kernels", conv1_k)
    upd("conv1/biases", lambda t: t[:3].zero_())
    def passthrough(t):                       # [1,1,cin,cout]: out channel c reads in channel c only
        t[:, :, :, :3] = 0.0
        for c in range(3):
            t[0, 0, c, c] = 1.0
    for f in FIRES:
        upd(f + "/squeeze1x1/kernels", passthrough)
        upd(f + "/squeeze1x1/biases", lambda t: t[:3].zero_())
        if f != "fire11":
            upd(f + "/expand1x1/kernels", passthrough)
            upd(f + "/expand1x1/biases", lambda t: t[:3].zero_())
    def det_k(t):
        t[:, :, :, :3] = 0.0
        for c in range(3):
            t[0, 0, c, c] = PLANT_DET_SCALE
    upd("fire11/expand1x1/kernels", det_k)
    upd("fire11/expand1x1/biases", lambda t: t[:3].fill_(-q * PLANT_DET_SCALE))
    w12 = torch.zeros_like(p["conv12/kernels"], dtype=torch.float32)
    b12 = torch.zeros(K * (C + 5), dtype=torch.float32)
    for k in range(K):
        w12[1, 1, k % C, K * C + k] = PLANT_CONF_GAIN
        b12[k * C + (k % C)] = 1.0 + 0.25 * k                          # softmax level of shape k: 0.58 .. 0.91
        b12[K * C + k] = PLANT_CONF_BIAS
        d = K * (C + 1) + 4 * k
        b12[d:d + 4] = torch.tensor(planted_deltas(k))
    p["conv12/kernels"] = w12
    p["conv12/biases"] = b12
    return p
