"""Self-checks of the restated TF-graph half (parity unpinned against TF
itself -- see oracle/__init__.py): TF SAME-padding table from SURVEY.md 8a,
conv/pool against naive NumPy loops, interpret_output against a scalar
per-anchor transcription of SURVEY.md 9.9."""
import math

import numpy as np
import torch

from oracle import sqdet_oracle as O


def test_same_pads_table():
    # (in, k, s) -> (before, after); SURVEY.md layer tables
    assert O.same_pads(384, 3, 2) == (0, 1) and O.same_pads(1248, 3, 2) == (0, 1)     # conv1 @384x1248
    assert O.same_pads(375, 3, 2) == (1, 1) and O.same_pads(1242, 3, 2) == (0, 1)     # conv1 @375x1242
    assert O.same_pads(188, 3, 2) == (0, 1) and O.same_pads(621, 3, 2) == (1, 1)      # pool1 @188x621
    assert O.same_pads(94, 3, 2) == (0, 1) and O.same_pads(311, 3, 2) == (1, 1)       # pool3
    assert O.same_pads(47, 3, 2) == (1, 1) and O.same_pads(156, 3, 2) == (0, 1)       # pool5
    assert O.same_pads(24, 3, 1) == (1, 1)
    assert O.squeezedet_grid(384, 1248) == (24, 78) and O.squeezedet_grid(375, 1242) == (24, 78)
    assert O.out_size(375, 7, 2, "VALID") == 185 and O.out_size(1242, 7, 2, "VALID") == 618


def _naive_conv(x, w, b, s, padding, relu):
    N, H, W, C = x.shape
    k = w.shape[0]
    F_ = w.shape[3]
    if padding == "SAME":
        pt, pb = O.same_pads(H, k, s)
        pl, pr = O.same_pads(W, k, s)
    else:
        pt = pb = pl = pr = 0
    xp = np.zeros((N, H + pt + pb, W + pl + pr, C), np.float64)
    xp[:, pt:pt + H, pl:pl + W] = x
    Ho, Wo = O.out_size(H, k, s, padding), O.out_size(W, k, s, padding)
    y = np.zeros((N, Ho, Wo, F_), np.float64)
    for oy in range(Ho):
        for ox in range(Wo):
            patch = xp[:, oy * s:oy * s + k, ox * s:ox * s + k, :]
            y[:, oy, ox, :] = np.tensordot(patch, w, axes=([1, 2, 3], [0, 1, 2]))
    y += b
    return np.maximum(y, 0) if relu else y


def test_conv_layer_vs_naive():
    rs = np.random.RandomState(0)
    for (H, W, C, F_, k, s, pad, relu) in [(9, 11, 3, 8, 3, 2, "SAME", True), (10, 12, 3, 8, 3, 2, "SAME", True),
                                            (7, 9, 8, 5, 3, 1, "SAME", False), (7, 9, 8, 16, 1, 1, "SAME", True),
                                            (15, 17, 3, 6, 7, 2, "VALID", True)]:
        x = rs.randn(2, H, W, C).astype(np.float32)
        w = rs.randn(k, k, C, F_).astype(np.float32)
        b = rs.randn(F_).astype(np.float32)
        y = O.conv_layer(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s, pad, relu).numpy()
        np.testing.assert_allclose(y, _naive_conv(x, w, b, s, pad, relu), rtol=1e-4, atol=1e-4)


def test_pool_same_ignores_padding():
    x = -np.ones((1, 5, 6, 2), np.float32) * 3  # all negative: zero padding would win, -inf must not
    y = O.pooling_layer(torch.from_numpy(x), 3, 2, "SAME").numpy()
    assert y.shape == (1, 3, 3, 2) and (y == -3).all()
    rs = np.random.RandomState(1)
    x = rs.randn(2, 7, 10, 3).astype(np.float32)
    y = O.pooling_layer(torch.from_numpy(x), 3, 2, "SAME").numpy()
    pt, _ = O.same_pads(7, 3, 2)
    pl, _ = O.same_pads(10, 3, 2)
    for oy in range(y.shape[1]):
        for ox in range(y.shape[2]):
            ys = slice(max(oy * 2 - pt, 0), min(oy * 2 - pt + 3, 7))
            xs = slice(max(ox * 2 - pl, 0), min(ox * 2 - pl + 3, 10))
            np.testing.assert_array_equal(y[:, oy, ox], x[:, ys, xs].max(axis=(1, 2)))
    yv = O.pooling_layer(torch.from_numpy(x), 3, 2, "VALID").numpy()
    assert yv.shape == (2, 3, 4, 3)


def test_interpret_output_vs_scalar_transcription():
    mc = O.kitti_squeezeDet_config()
    rs = np.random.RandomState(3)
    preds = (rs.randn(1, 24, 78, 72) * 1.5).astype(np.float32)
    out = O.interpret_output(preds, mc)
    assert out["det_boxes"].shape == (1, 16848, 4) and out["det_boxes"].dtype == np.float32
    assert out["det_probs"].dtype == np.float32 and out["det_class"].dtype == np.int64
    f = np.float32
    anchors = mc.ANCHOR_BOX.astype(np.float32)
    for a in rs.randint(0, 16848, 200):
        hw, k = divmod(a, 9)
        h, w = divmod(hw, 78)
        p = preds[0, h, w]
        lg = p[3 * k:3 * k + 3]
        e = np.exp(lg - lg.max()).astype(f)
        cls = e * (f(1) / ((e[0] + e[1]) + e[2]))
        conf = f(1) / (f(1) + np.exp(-p[27 + k]).astype(f))
        dx, dy, dw, dh = p[36 + 4 * k:36 + 4 * k + 4]
        ax, ay, aw, ah = anchors[a]
        sexp = lambda v: f(np.exp(1.0)) * (v - f(1) + f(1)) if v > 1 else np.exp(v).astype(f)
        cx, cy = ax + dx * aw, ay + dy * ah
        bw, bh = aw * sexp(dw), ah * sexp(dh)
        xmin, ymin, xmax, ymax = cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2
        xmin = min(max(f(0), xmin), f(1247)); ymin = min(max(f(0), ymin), f(383))
        xmax = max(min(f(1247), xmax), f(0)); ymax = max(min(f(383), ymax), f(0))
        w2, h2 = xmax - xmin + f(1), ymax - ymin + f(1)
        np.testing.assert_array_equal(out["det_boxes"][0, a], np.array([xmin + f(0.5) * w2, ymin + f(0.5) * h2, w2, h2], f))
        pr = cls * conf
        assert out["det_probs"][0, a] == pr.max() and out["det_class"][0, a] == pr.argmax()
    # every box inside the image, w/h >= 1 (bbox_transform_inv adds +1)
    b = out["det_boxes"][0]
    assert (b[:, 2] >= 1).all() and (b[:, 3] >= 1).all()


def test_forward_shapes_and_storage_modes():
    for arch, (ih, iw), grid in (("squeezeDet", (384, 1248), (24, 78)), ("squeezeDet", (375, 1242), (24, 78)),
                                  ("squeezeDet+", (375, 1242), (22, 76))):
        if arch == "squeezeDet+":
            ih, iw = 121, 187  # keep the CPU suite fast: only shapes matter here
            grid = None
        p = O.init_params(arch, seed=0)
        x = O.synthetic_images(1, ih, iw, seed=0)
        col = {}
        y = O.forward(arch, p, x, collect=col)
        if grid:
            assert tuple(y.shape) == (1,) + grid + (72,)
        assert y.shape[-1] == 72 and torch.isfinite(y).all()
    n_params = sum(int(np.prod(s)) for s in O.param_shapes("squeezeDet").values())
    assert n_params == 2082120  # BASELINE.md derived parameter count
    assert sum(int(np.prod(s)) for s in O.param_shapes("squeezeDet+").values()) == 7021640


def test_float16_storage_restatement_rounds_values_not_gradients():
    """train_oracle._q (the mixed-precision restatement): the forward value is the float16-rounded one, the gradient
    passes straight through (weight gradients stay float32 on the device as well); an `override` pins the forward
    value of a stored activation without touching the gradient path."""
    import torch
    from oracle import train_oracle as TO
    t = torch.tensor([0.1, 1.0 + 2.0 ** -12, -3.3333, 70000.0], requires_grad=True)
    q = TO._q(t, "fp16")
    assert torch.equal(q.detach(), t.detach().half().float())
    (q * torch.tensor([1.0, 2.0, 3.0, 0.0])).sum().backward()
    assert torch.equal(t.grad, torch.tensor([1.0, 2.0, 3.0, 0.0]))
    assert TO._q(t, "fp32") is t
    # override: a two-layer toy through forward_train's own helper semantics
    x = torch.randn(1, 4, 4, 8)
    w = (torch.randn(1, 1, 8, 8) * 0.3).requires_grad_(True)
    b = torch.zeros(8)
    y = TO._conv(x, w, b, 1, "SAME", True, "fp16")
    pinned = (y.detach() + 0.25) * (y.detach() > 0)                      # some other forward value, same ReLU pattern
    y2 = y + (pinned - y).detach()
    (g_free,) = torch.autograd.grad(y.sum(), w, retain_graph=True)
    (g_pin,) = torch.autograd.grad(y2.sum(), w)
    assert torch.equal(y2.detach(), pinned) and torch.equal(g_free, g_pin)
