"""GPU parity tests of the TRAINING path (float32): every backward / loss / optimizer kernel against
PyTorch-CPU autograd on the restated graph (oracle/train_oracle.py; parity unpinned against TF --
see that module's header), then one whole training step.

Tolerances: gradients are sums of up to ~1e5 float32 products evaluated in a different order than
the CPU's, so they are compared at 2e-4 relative to the tensor's max (north_star: 1e-3 rel)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sqdet_oracle as O
from oracle import train_oracle as TO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from squeezedet_amd import ops
    return ops


def _close(got, ref, rel=2e-4, what=""):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().float().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, what
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(got - ref).max()
    assert err <= rel * scale + 1e-7, "%s: max err %g vs scale %g" % (what, err, scale)


def _conv_ref(x, w, stride=1):
    k = w.shape[0]
    pad = (k - 1) // 2
    return F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, stride=stride, padding=pad).permute(0, 2, 3, 1)


BWD_CASES = [("e1_16_64", 2, 23, 31, 16, 64, 1), ("e3_16_64", 2, 23, 31, 16, 64, 3), ("sq_128_32", 1, 12, 39, 128, 32, 1),
             ("e3_48_192", 1, 12, 20, 48, 192, 3), ("sq_768_96", 1, 9, 14, 768, 96, 1), ("e3_96_384", 1, 9, 14, 96, 384, 3),
             ("convdet_768_72", 2, 7, 13, 768, 72, 3), ("e1_32_128_large", 2, 47, 63, 32, 128, 1),
             # backward-filter tile variants: 9 taps per workgroup (>= 100k pixels, Cin 16 / 32), Cout 16 / 48 / 96
             # squeezes, Cin 48 / 64 / 96 expands, ResNet-style 64->64 3x3 and 256->64 1x1, ragged Cin / Cout tails
             ("e3_16_64_9tap", 2, 190, 270, 16, 64, 3), ("e3_32_128_9tap", 1, 250, 401, 32, 128, 3),
             ("sq_96_16", 1, 21, 37, 96, 16, 1), ("sq_256_48", 1, 12, 39, 256, 48, 1), ("sq_512_96", 1, 9, 17, 512, 96, 1),
             ("e1_48_192", 1, 12, 20, 48, 192, 1), ("e3_64_256", 1, 11, 19, 64, 256, 3), ("e1_96_384", 1, 9, 14, 96, 384, 1),
             ("res_64_64", 1, 17, 33, 64, 64, 3), ("res_256_64", 1, 17, 33, 256, 64, 1), ("ragged_20_36", 1, 6, 18, 20, 36, 3),
             ("ragged_40_24", 2, 5, 16, 40, 24, 1),
             # SqueezeDet+ late modules (22x76 map, squeeze 384, expand 256) and its 96 / 192 / 288-channel squeezes
             ("plus_e1_384_256", 1, 22, 76, 384, 256, 1), ("plus_e3_384_256", 1, 22, 76, 384, 256, 3),
             ("plus_sq_512_384", 1, 22, 76, 512, 384, 1), ("plus_e3_192_128", 1, 23, 39, 192, 128, 3),
             ("plus_sq_256_288", 1, 23, 39, 256, 288, 1),
             # ResNet50 res4 at the training batch (8 x 24 x 78): 8 - 16 slabs of 0.13 - 0.59 M elements, the slab reduction's
             # four-elements-per-thread form (wgrad.hip slab_reduce_body)
             ("res4_2a_1024_256", 8, 24, 78, 1024, 256, 1), ("res4_2b_256_256", 8, 24, 78, 256, 256, 3),
             ("res4_512_256_16slabs", 8, 24, 78, 512, 256, 1)]


@pytest.mark.parametrize("case", BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_conv_backward_data_and_filter(case):
    ops = _ops()
    name, N, H, W, cin, cout, k = case
    rs = np.random.RandomState(len(name) * 7 + cin)
    x = torch.from_numpy(rs.randn(N, H, W, cin).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((rs.randn(k, k, cin, cout) * (2.0 / (k * k * cin)) ** 0.5).astype(np.float32)).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    dy = torch.from_numpy(rs.randn(N, H, W, cout).astype(np.float32))
    y = _conv_ref(x, w) + b
    y.backward(dy)
    dxg = ops.conv2d_bwd_data(dy.to(DEV), ops.PackedConvBwd(w.detach().to(DEV)))
    wd = 1e-4
    dwg, dbg = ops.conv2d_bwd_filter(x.detach().to(DEV), dy.to(DEV), k, cin, cout, w_for_decay=w.detach().to(DEV), weight_decay=wd)
    torch.cuda.synchronize()
    _close(dxg, x.grad, what=name + " dx")
    _close(dwg, w.grad + wd * w.detach(), what=name + " dW")
    _close(dbg, b.grad, what=name + " dbias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_wgrad_plan_one_reduction_for_many_convs(dtype):
    """ops.WgradPlan (sqdet_conv2d_nhwc_bwd_filter_partial per conv + ONE sqdet_slab_reduce_many) against the per-conv
    two-launch sqdet_conv2d_nhwc_bwd_filter: bitwise, with and without bias / weight decay, channel-sliced dy, twice."""
    ops = _ops()
    ev = 8 if dtype == torch.float16 else 4
    cases = [c for c in BWD_CASES if c[4] % ev == 0 and c[5] % ev == 0 and c[0] in (
        "e3_16_64", "convdet_768_72", "e3_32_128_9tap", "sq_512_96", "ragged_20_36", "ragged_40_24", "plus_e3_384_256", "e1_48_192")]
    assert len(cases) >= 6
    rs = np.random.RandomState(5)
    items, data, want = [], {}, {}
    for i, (name, N, H, W, cin, cout, k) in enumerate(cases):
        x = torch.from_numpy(rs.randn(N, H, W, cin).astype(np.float32)).to(DEV, dtype)
        off = ev * (i % 3)
        dy = torch.from_numpy(rs.randn(N, H, W, cout + off + ev).astype(np.float32)).to(DEV, dtype)     # the conv reads a channel slice
        w = torch.from_numpy(rs.randn(k, k, cin, cout).astype(np.float32)).to(DEV) if i % 2 else None
        dw = torch.empty(k, k, cin, cout, dtype=torch.float32, device=DEV)
        db = torch.empty(cout, dtype=torch.float32, device=DEV) if i % 3 else None
        items.append((name, (N, H, W, cin, cout, k), dw, db, w, 1e-3 * i))
        data[name] = (x, dy, off)
        want[name] = ops.conv2d_bwd_filter(x, dy, k, cin, cout, w_for_decay=w, weight_decay=1e-3 * i, dy_coffset=off,
                                           want_bias=db is not None, grad_scale=0.25)
    plan = ops.WgradPlan(items)
    for _ in range(2):
        for name, _, dw, db, _, _ in items:
            dw.fill_(float("nan"))
            x, dy, off = data[name]
            plan.partial(name, x, dy, dy_coffset=off)
        plan.reduce(0.25)
        torch.cuda.synchronize()
        for name, _, dw, db, _, _ in items:
            assert torch.equal(dw, want[name][0]), name
            assert db is None or torch.equal(db, want[name][1]), name


@pytest.mark.parametrize("case", ["res4_2a_1024_256", "res4_2b_256_256", "res4_512_256_16slabs", "plus_e3_384_256", "convdet_768_72"])
def test_slab_reduction_wide_form_has_the_four_lane_forms_bits(case):
    """The slab reduction's four-elements-per-thread form (large gradients, <= 16 slabs) against the four-lanes-per-element
    form ("dbg" 52) on the same partial sums: bitwise, with weight decay, a bias gradient, a gradient scale, and a dW that is
    only 4-byte aligned (a trainer's flat gradient view); the same through WgradPlan's one-launch reduction."""
    ops = _ops()
    name, N, H, W, cin, cout, k = next(c for c in BWD_CASES if c[0] == case)
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.randn(N, H, W, cin).astype(np.float32)).to(DEV, torch.float16)
    dy = torch.from_numpy(rs.randn(N, H, W, cout).astype(np.float32)).to(DEV, torch.float16)
    w = torch.from_numpy(rs.randn(k, k, cin, cout).astype(np.float32)).to(DEV)
    cnt = k * k * cin * cout

    def run():
        flat = torch.full((cnt + 8,), float("nan"), dtype=torch.float32, device=DEV)
        dw_odd = flat[1:1 + cnt].view(k, k, cin, cout)             # 4-byte aligned only
        a = ops.conv2d_bwd_filter(x, dy, k, cin, cout, w_for_decay=w, weight_decay=1e-3, grad_scale=0.125)
        b = ops.conv2d_bwd_filter(x, dy, k, cin, cout, w_for_decay=w, weight_decay=1e-3, grad_scale=0.125, dw=dw_odd)
        dwp, dbp = torch.empty_like(a[0]), torch.empty_like(a[1])
        plan = ops.WgradPlan([(name, (N, H, W, cin, cout, k), dwp, dbp, w, 1e-3)])
        plan.partial(name, x, dy)
        plan.reduce(0.125)
        torch.cuda.synchronize()
        assert bool(torch.isnan(flat[0])) and bool(torch.isnan(flat[1 + cnt:]).all())
        return a[0], a[1], b[0].clone(), b[1], dwp, dbp
    new = run()
    ops.set_option("dbg", 52)
    try:
        old = run()
    finally:
        ops.set_option("dbg", 0)
    for got, ref in zip(new, old):
        assert torch.isfinite(ref).all() and torch.equal(got, ref)
    assert torch.equal(new[0], new[2]) and torch.equal(new[0], new[4]) and torch.equal(new[1], new[5])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_maxpool_window_index_forward_and_backward(dtype):
    """sqdet_maxpool_nhwc_fwd_idx / sqdet_maxpool_nhwc_bwd_idx: y bitwise sqdet_maxpool_nhwc_fwd's, the index names the FIRST
    maximum (ties included: quantised inputs), and dx -- with and without the fused ReLU backward -- bitwise the x-searching
    kernels' (sqdet_maxpool_nhwc_bwd / _bwd_relu)."""
    ops = _ops()
    rs = np.random.RandomState(17)
    ev = 8 if dtype == torch.float16 else 4
    for (N, H, W, C, pad) in ((2, 47, 156, 4 * ev, "SAME"), (1, 94, 311, ev, "SAME"), (2, 20, 31, 2 * ev, "VALID"), (1, 7, 9, ev, "SAME")):
        x = torch.from_numpy(np.maximum(np.round(rs.randn(N, H, W, C) * 2) / 2, 0).astype(np.float32)).to(DEV, dtype)   # many ties, ReLU-like
        y0 = ops.maxpool_nhwc(x, 3, 2, pad)
        y, idx = ops.maxpool_nhwc_idx(x, 3, 2, pad)
        assert torch.equal(y, y0)
        # the index against a direct scan on the host
        xc, ic = x.float().cpu().numpy(), idx.cpu().numpy()
        Ho, Wo = y.shape[1:3]
        pt = max((Ho - 1) * 2 + 3 - H, 0) // 2 if pad == "SAME" else 0
        pl = max((Wo - 1) * 2 + 3 - W, 0) // 2 if pad == "SAME" else 0
        for (n, oy, ox) in [(0, 0, 0), (N - 1, Ho - 1, Wo - 1), (0, Ho // 2, Wo // 3), (N - 1, 1, Wo - 1), (0, Ho - 1, 0)]:
            best, pos = np.full(C, -np.inf, np.float32), np.full(C, 255, np.int64)
            for t in range(9):
                iy, ix = 2 * oy - pt + t // 3, 2 * ox - pl + t % 3
                if 0 <= iy < H and 0 <= ix < W:
                    win = xc[n, iy, ix] > best
                    best, pos = np.where(win, xc[n, iy, ix], best), np.where(win, t, pos)
            np.testing.assert_array_equal(ic[n, oy, ox], pos)
        dy = torch.from_numpy(rs.randn(*y.shape).astype(np.float32)).to(DEV, dtype)
        for relu in (False, True):
            want = ops.maxpool_bwd(x, dy, 3, 2, pad, relu=relu)
            got = ops.maxpool_bwd_idx(idx, y, dy, (H, W), 3, 2, pad, relu=relu)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (N, H, W, C, pad, relu)
    # a window whose valid cells are ALL -inf still names a cell (the first valid one), and its gradient lands there as in the
    # x-searching kernel (round-3 ADVICE: the index used to stay 255 and the gradient went nowhere)
    x = torch.from_numpy(rs.randn(1, 9, 11, ev).astype(np.float32)).to(DEV, dtype)
    x[0, 0:4, 0:5, :] = float("-inf")
    y, idx = ops.maxpool_nhwc_idx(x, 3, 2, "SAME")
    assert torch.equal(y, ops.maxpool_nhwc(x, 3, 2, "SAME")) and bool(torch.isinf(y[0, 0, 0]).all())
    assert int(idx.max()) < 9, "a window index was left unset"
    pt, pl = max((y.shape[1] - 1) * 2 + 3 - 9, 0) // 2, max((y.shape[2] - 1) * 2 + 3 - 11, 0) // 2
    first = next(t for t in range(9) if 0 <= 0 - pt + t // 3 and 0 <= 0 - pl + t % 3)
    assert bool((idx[0, 0, 0] == first).all())
    dy = torch.from_numpy(rs.randn(*y.shape).astype(np.float32)).to(DEV, dtype)
    assert torch.equal(ops.maxpool_bwd_idx(idx, y, dy, (9, 11), 3, 2, "SAME", relu=False), ops.maxpool_bwd(x, dy, 3, 2, "SAME", relu=False))
    # NaN cells (round-4 ADVICE): a NaN never wins the comparison wherever it sits in the window -- first valid cell or not -- so the
    # training forward's pooled tensor stays BITWISE the inference pool's, the index names the first maximum among the other cells
    # (the first valid cell when there is none), and both backward kernels route the gradient alike
    xn = torch.from_numpy(np.round(rs.randn(2, 13, 17, ev) * 2).astype(np.float32) / 2 + 0.0).to(DEV, dtype)   # (+ 0.0: no negative zeros -- the comparison below is on BITS)
    xn[0, 0, 0, :] = float("nan")            # the FIRST valid cell of window (0, 0)
    xn[0, 5, 6, :] = float("nan")            # an inner cell of several windows
    xn[1, 3:6, 3:6, :] = float("nan")        # a whole window of NaN (output cell (2, 2): -inf, the first cell named)
    yn0 = ops.maxpool_nhwc(xn, 3, 2, "SAME")
    yn, idxn = ops.maxpool_nhwc_idx(xn, 3, 2, "SAME")
    torch.cuda.synchronize()
    assert torch.equal(yn.view(torch.int16 if dtype == torch.float16 else torch.int32), yn0.view(torch.int16 if dtype == torch.float16 else torch.int32))
    assert not bool(torch.isnan(yn).any()) and int(idxn.max()) < 9
    dyn = torch.from_numpy(rs.randn(*yn.shape).astype(np.float32)).to(DEV, dtype)
    assert torch.equal(ops.maxpool_bwd_idx(idxn, yn, dyn, (13, 17), 3, 2, "SAME", relu=False), ops.maxpool_bwd(xn, dyn, 3, 2, "SAME", relu=False))


def test_fire_backward_channel_slices_and_accumulate():
    """A fire module's backward: dY is the concat gradient; expand1x1 / expand3x3 read its two channel
    ranges, and the squeeze tensor's gradient is the SUM of their backward-data results."""
    ops = _ops()
    rs = np.random.RandomState(2)
    N, H, W, s, e = 2, 13, 21, 32, 128
    S = torch.from_numpy(np.maximum(rs.randn(N, H, W, s), 0).astype(np.float32)).requires_grad_(True)
    w1 = torch.from_numpy((rs.randn(1, 1, s, e) * 0.2).astype(np.float32)).requires_grad_(True)
    w3 = torch.from_numpy((rs.randn(3, 3, s, e) * 0.08).astype(np.float32)).requires_grad_(True)
    dY = torch.from_numpy(rs.randn(N, H, W, 2 * e).astype(np.float32))
    Y = torch.cat([_conv_ref(S, w1), _conv_ref(S, w3)], dim=3)
    Y.backward(dY)
    dYd, Sd = dY.to(DEV), S.detach().to(DEV)
    dS = ops.conv2d_bwd_data(dYd, ops.PackedConvBwd(w1.detach().to(DEV)), dy_coffset=0)
    ops.conv2d_bwd_data(dYd, ops.PackedConvBwd(w3.detach().to(DEV)), dx=dS, dy_coffset=e, accumulate=True)
    dw1, _ = ops.conv2d_bwd_filter(Sd, dYd, 1, s, e, dy_coffset=0)
    dw3, _ = ops.conv2d_bwd_filter(Sd, dYd, 3, s, e, dy_coffset=e)
    torch.cuda.synchronize()
    _close(dS, S.grad, what="dS")
    _close(dw1, w1.grad, what="dW1")
    _close(dw3, w3.grad, what="dW3")


def test_relu_dropout_maxpool_backward():
    ops = _ops()
    rs = np.random.RandomState(3)
    y = torch.from_numpy(np.maximum(rs.randn(2, 9, 11, 16), 0).astype(np.float32))
    dy = torch.from_numpy(rs.randn(2, 9, 11, 16).astype(np.float32))
    got = ops.relu_bwd(y.to(DEV), dy.to(DEV).clone())
    np.testing.assert_array_equal(got.cpu().numpy(), (dy * (y > 0)).numpy())
    m = torch.from_numpy((rs.uniform(size=(2, 9, 11, 16)) < 0.5).astype(np.float32))
    np.testing.assert_array_equal(ops.scale_mask(dy.to(DEV), m.to(DEV), 2.0).cpu().numpy(), (dy * m * 2.0).numpy())
    for (H, W, pad) in ((47, 156, "SAME"), (94, 311, "SAME"), (20, 31, "VALID")):
        x = torch.from_numpy(rs.randn(1, H, W, 8).astype(np.float32)).requires_grad_(True)   # distinct values: unique argmax
        yp = O.pooling_layer(x, 3, 2, pad)
        g = torch.from_numpy(rs.randn(*yp.shape).astype(np.float32))
        yp.backward(g)
        dx = ops.maxpool_bwd(x.detach().to(DEV), g.to(DEV), 3, 2, pad)
        torch.cuda.synchronize()
        np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy(), rtol=1e-6, atol=1e-6)
    # other window shapes run the generic gather kernel
    for (H, W, size, stride, pad) in ((21, 30, 2, 2, "VALID"), (13, 17, 3, 1, "SAME")):
        x = torch.from_numpy(rs.randn(2, H, W, 8).astype(np.float32)).requires_grad_(True)
        yp = O.pooling_layer(x, size, stride, pad)
        g = torch.from_numpy(rs.randn(*yp.shape).astype(np.float32))
        yp.backward(g)
        dx = ops.maxpool_bwd(x.detach().to(DEV), g.to(DEV), size, stride, pad)
        torch.cuda.synchronize()
        np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_relu_backward_fused_into_the_producing_kernel(dtype):
    """sqdet_conv2d_nhwc_bwd_data_relu / sqdet_maxpool_nhwc_bwd_relu: the ReLU backward of the layer below taken in the
    epilogue of the kernel that produces the gradient (incl. after an accumulation) == the separate relu_bwd pass, bitwise;
    every kernel family that serves a backward-data conv (generic, 1x1 tile, 3x3 tile; sliced dY, accumulate)."""
    ops = _ops()
    rs = np.random.RandomState(12)
    for (N, H, W, cin, cout, k) in ((2, 13, 21, 32, 128, 1), (1, 24, 78, 96, 384, 3), (2, 9, 11, 16, 64, 3), (1, 17, 19, 48, 200, 1)):
        dY = torch.from_numpy(rs.randn(N, H, W, 2 * cout).astype(np.float32)).to(DEV, dtype)
        r = torch.from_numpy(np.maximum(rs.randn(N, H, W, cin), 0).astype(np.float32)).to(DEV, dtype)   # ~half zeros
        pw = ops.PackedConvBwd(torch.from_numpy((rs.randn(k, k, cin, cout) * 0.1).astype(np.float32)).to(DEV), dtype)
        p1 = ops.PackedConvBwd(torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.1).astype(np.float32)).to(DEV), dtype)
        # plain
        a = ops.relu_bwd(r, ops.conv2d_bwd_data(dY, pw, dy_coffset=cout))
        b = ops.conv2d_bwd_data(dY, pw, dy_coffset=cout, relu_of=r)
        assert torch.equal(a, b), (cin, cout, k)
        # accumulate: mask applies to the SUM
        a = ops.conv2d_bwd_data(dY, p1, dy_coffset=0)
        ops.conv2d_bwd_data(dY, pw, dx=a, dy_coffset=cout, accumulate=True)
        a = ops.relu_bwd(r, a)
        b = ops.conv2d_bwd_data(dY, p1, dy_coffset=0)
        ops.conv2d_bwd_data(dY, pw, dx=b, dy_coffset=cout, accumulate=True, relu_of=r)
        assert torch.equal(a, b), (cin, cout, k, "accumulate")
    for (H, W, size, stride, pad) in ((47, 156, 3, 2, "SAME"), (20, 31, 3, 2, "VALID"), (13, 17, 3, 1, "SAME")):
        x = torch.from_numpy(np.maximum(rs.randn(2, H, W, 16), 0).astype(np.float32)).to(DEV, dtype)
        yp = ops.maxpool_nhwc(x, size, stride, pad)
        g = torch.from_numpy(rs.randn(*yp.shape).astype(np.float32)).to(DEV, dtype)
        a = ops.relu_bwd(x, ops.maxpool_bwd(x, g, size, stride, pad))
        b = ops.maxpool_bwd(x, g, size, stride, pad, relu=True)
        assert torch.equal(a, b), (H, W, size, stride, pad)
    torch.cuda.synchronize()


def test_float16_elementwise_backward_ops():
    """The activation-side backward kernels on float16 tensors (mixed-precision training): exact against the same
    arithmetic on the float16-rounded values."""
    ops = _ops()
    rs = np.random.RandomState(13)
    y = torch.from_numpy(np.maximum(rs.randn(2, 9, 11, 16), 0).astype(np.float32)).half()
    dy = torch.from_numpy(rs.randn(2, 9, 11, 16).astype(np.float32)).half()
    got = ops.relu_bwd(y.to(DEV), dy.to(DEV).clone())
    assert got.dtype == torch.float16
    np.testing.assert_array_equal(got.cpu().numpy(), (dy * (y > 0)).numpy())
    m = torch.from_numpy((rs.uniform(size=(2, 9, 11, 16)) < 0.5).astype(np.float32)).half()
    np.testing.assert_array_equal(ops.scale_mask(dy.to(DEV), m.to(DEV), 2.0).cpu().numpy(), (dy.float() * m.float() * 2.0).half().numpy())
    # dropout backward + the ReLU backward below it in one pass (sqdet_scale_mask_relu) = the two launches
    for dt in (torch.float16, torch.float32):
        a, b, c = dy.to(DEV, dt), m.to(DEV, dt), y.to(DEV, dt)
        assert torch.equal(ops.scale_mask(a, b, 2.0, relu_of=c), ops.relu_bwd(c, ops.scale_mask(a, b, 2.0)))
    for (H, W, pad) in ((47, 156, "SAME"), (20, 31, "VALID")):
        # distinct float16 values per channel so the argmax is unique
        bits = np.stack([rs.permutation(H * W) for _ in range(8)], -1).reshape(1, H, W, 8) + 0x2000
        vals = bits.astype(np.uint16).view(np.float16).astype(np.float32)     # consecutive float16 bit patterns: all distinct
        x = torch.from_numpy(vals).requires_grad_(True)
        yp = O.pooling_layer(x, 3, 2, pad)
        g = torch.from_numpy(rs.randn(*yp.shape).astype(np.float32)).half().float()
        yp.backward(g)
        dx = ops.maxpool_bwd(x.detach().half().to(DEV), g.half().to(DEV), 3, 2, pad)
        torch.cuda.synchronize()
        assert dx.dtype == torch.float16
        np.testing.assert_allclose(dx.float().cpu().numpy(), x.grad.numpy(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("case", BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_conv_backward_float16(case):
    """Backward-data and backward-filter on float16 activations / activation gradients (float32 dW, dbias): against
    float32 autograd on the SAME float16-rounded inputs; the MFMA accumulates in float32, so only the summation
    order differs (dx additionally rounds to float16)."""
    ops = _ops()
    name, N, H, W, cin, cout, k = case
    if cin % 8 or cout % 8:
        pytest.skip("float16 rows are 16-byte vectors: channel counts must be multiples of 8")
    rs = np.random.RandomState(len(name) * 11 + cout)
    x = torch.from_numpy(rs.randn(N, H, W, cin).astype(np.float32)).half().float().requires_grad_(True)
    w = torch.from_numpy((rs.randn(k, k, cin, cout) * (2.0 / (k * k * cin)) ** 0.5).astype(np.float32)).half().float().requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    dy = torch.from_numpy(rs.randn(N, H, W, cout).astype(np.float32)).half().float()
    y = _conv_ref(x, w) + b
    y.backward(dy)
    dxg = ops.conv2d_bwd_data(dy.half().to(DEV), ops.PackedConvBwd(w.detach().to(DEV), torch.float16))
    dwg, dbg = ops.conv2d_bwd_filter(x.detach().half().to(DEV), dy.half().to(DEV), k, cin, cout, grad_scale=0.25)
    torch.cuda.synchronize()
    assert dxg.dtype == torch.float16 and dwg.dtype == torch.float32 and dbg.dtype == torch.float32
    _close(dxg, x.grad, rel=2e-3, what=name + " dx (f16)")
    _close(dwg, 0.25 * w.grad, rel=1e-4, what=name + " dW (f16 in, f32 out)")
    _close(dbg, 0.25 * b.grad, rel=1e-4, what=name + " dbias (f16 in, f32 out)")


def test_optimizer_skips_overflowed_step():
    """found_inf: a non-finite gradient norm anywhere skips the whole update (loss-scaled float16 training)."""
    ops = _ops()
    offs, cnts = [0, 128], [100, 64]
    opt = ops.MomentumOptimizer(offs, cnts, [1e-4, 0.0], DEV)
    p = torch.randn(192, device=DEV)
    g = torch.randn(192, device=DEV)
    acc = torch.zeros(192, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    p0 = p.clone()
    g2 = g.clone()
    g2[130] = float("inf")
    opt.step(p, g2, acc, 0.01, 0.9, 1.0, 1.0, found_inf=flag)
    assert int(flag.item()) == 1 and torch.equal(p, p0) and float(acc.abs().sum()) == 0.0
    opt.step(p, g.clone(), acc, 0.01, 0.9, 1.0, 1.0, found_inf=flag)
    assert int(flag.item()) == 0 and not torch.equal(p, p0)


def test_loss_forward_backward_vs_oracle():
    ops = _ops()
    mc = O.squeezeDet_config_for_input(128, 256)
    B = 3
    rs = np.random.RandomState(5)
    gh, gw = O.squeezedet_grid(128, 256)
    preds = torch.from_numpy((rs.randn(B, gh, gw, 72) * 1.2).astype(np.float32)).requires_grad_(True)
    mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=6)
    parts = TO.loss_graph(mc, preds, mask, delta, box, labels)
    total = parts["class_loss"] + parts["conf_loss"] + parts["bbox_loss"]
    (dref,) = torch.autograd.grad(total, preds)
    anchors = torch.from_numpy(mc.ANCHOR_BOX.astype(np.float32)).to(DEV)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dp, ious, losses = ops.loss_fwd_bwd(preds.detach().to(DEV), anchors, t(mask.reshape(B, -1)), t(delta), t(box), t(labels), mc,
                                        float(mask.sum()))
    torch.cuda.synchronize()
    ls = losses.cpu().numpy()
    np.testing.assert_allclose(ls, [float(parts["class_loss"]), float(parts["conf_loss"]), float(parts["bbox_loss"])], rtol=2e-5)
    np.testing.assert_allclose(ious.cpu().numpy(), parts["ious"].numpy(), rtol=1e-5, atol=1e-6)
    _close(dp, dref, rel=5e-5, what="dpreds")


def test_mixed_precision_loss_one_launch_equals_convert_loss_convert():
    """sqdet_loss_fwd_bwd_mixed (float16 preds in; float32 dpreds and the loss-scaled float16 gradient out) against
    convert_scale -> sqdet_loss_fwd_bwd_dev -> convert_scale: bitwise, num_objects from the host or from the device."""
    ops = _ops()
    mc = O.squeezeDet_config_for_input(128, 256)
    B = 3
    rs = np.random.RandomState(15)
    gh, gw = O.squeezedet_grid(128, 256)
    preds = torch.from_numpy((rs.randn(B, gh, gw, 72) * 1.2).astype(np.float32)).to(DEV, torch.float16)
    mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=16)
    anchors = torch.from_numpy(mc.ANCHOR_BOX.astype(np.float32)).to(DEV)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    args = (anchors, t(mask.reshape(B, -1)), t(delta), t(box), t(labels), mc)
    for nobj in (float(mask.sum()), torch.full((1,), float(mask.sum()), device=DEV)):
        dp, ious, losses = ops.loss_fwd_bwd(ops.convert_scale(preds, torch.float32), *args, nobj, global_batch=2 * B)
        want = ops.convert_scale(dp, torch.float16, 512.0)
        g16, dp2, ious2, losses2 = ops.loss_fwd_bwd_mixed(preds, *args, nobj, 512.0, global_batch=2 * B)
        torch.cuda.synchronize()
        assert g16.dtype == torch.float16 and torch.equal(g16, want)
        assert torch.equal(dp2, dp) and torch.equal(ious2, ious) and torch.equal(losses2, losses)


def test_loss_two_replica_shares_sum_to_the_full_batch_graph():
    """global_num_objects mode (SURVEY.md 8e option b) emulated in one process: two replicas of batch 2, each handed the
    GLOBAL num_objects and global_batch = 4, produce the dpreds rows and (summed) losses of ONE graph of batch 4 -- checked
    against the oracle's autograd at batch 4 (nn_skeleton.py:180,297-321: class / bbox divide by num_objects, the confidence
    term also by the batch).  With the local batch as the divisor the confidence gradient came out world x too large."""
    ops = _ops()
    mc = O.squeezeDet_config_for_input(128, 256)
    B = 4
    rs = np.random.RandomState(15)
    gh, gw = O.squeezedet_grid(128, 256)
    preds = torch.from_numpy((rs.randn(B, gh, gw, 72) * 1.2).astype(np.float32)).requires_grad_(True)
    mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=16)
    parts = TO.loss_graph(mc, preds, mask, delta, box, labels)
    total = parts["class_loss"] + parts["conf_loss"] + parts["bbox_loss"]
    (dref,) = torch.autograd.grad(total, preds)
    anchors = torch.from_numpy(mc.ANCHOR_BOX.astype(np.float32)).to(DEV)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    nobj = float(mask.sum())
    full = ops.loss_fwd_bwd(preds.detach().to(DEV), anchors, t(mask.reshape(B, -1)), t(delta), t(box), t(labels), mc, nobj)
    halves = [ops.loss_fwd_bwd(preds.detach()[i:i + 2].to(DEV).contiguous(), anchors, t(mask[i:i + 2].reshape(2, -1)), t(delta[i:i + 2]),
                               t(box[i:i + 2]), t(labels[i:i + 2]), mc, nobj, global_batch=B) for i in (0, 2)]
    torch.cuda.synchronize()
    dp = torch.cat([h[0] for h in halves], 0)
    assert torch.equal(dp, full[0])                      # per-anchor arithmetic is identical: bitwise
    _close(dp, dref, rel=5e-5, what="dpreds of the two shares vs the batch-4 graph")
    ls = (halves[0][2] + halves[1][2]).cpu().numpy()
    np.testing.assert_allclose(ls, [float(parts["class_loss"]), float(parts["conf_loss"]), float(parts["bbox_loss"])], rtol=2e-5)
    # and the default (replica-mean) call of a half is the reference at batch 2: a different, self-consistent normalisation
    own = ops.loss_fwd_bwd(preds.detach()[:2].to(DEV).contiguous(), anchors, t(mask[:2].reshape(2, -1)), t(delta[:2]), t(box[:2]),
                           t(labels[:2]), mc, float(mask[:2].sum()))
    p2 = preds.detach()[:2].clone().requires_grad_(True)
    parts2 = TO.loss_graph(mc, p2, mask[:2], delta[:2], box[:2], labels[:2])
    (d2,) = torch.autograd.grad(parts2["class_loss"] + parts2["conf_loss"] + parts2["bbox_loss"], p2)
    _close(own[0], d2, rel=5e-5, what="replica-mean dpreds")


def _trainer(img=(128, 256), batch=2, seed=0, dtype=torch.float32, **kw):
    import squeezedet_amd as S
    from squeezedet_amd import nets
    from squeezedet_amd.train import SqueezeDetTrainer
    mc = S.kitti_squeezeDet_config_for_input(*img)
    mc.LOAD_PRETRAINED_MODEL = False
    mc.IS_TRAINING = True
    mc.BATCH_SIZE = batch
    m = nets.SqueezeDet(mc, gpu_id="0", dtype=dtype)
    params = O.init_params("squeezeDet", seed=seed)
    m.load_params(params)
    return SqueezeDetTrainer(m, **kw), mc, params


def test_full_training_step_vs_oracle():
    """forward(train) -> loss -> backward -> clipped Momentum update: every gradient and every updated
    variable against PyTorch-CPU autograd + the restated train graph (nn_skeleton.py:285-361)."""
    tr, mc, params = _trainer()
    omc = O.squeezeDet_config_for_input(128, 256)
    omc.IS_TRAINING = True
    B = 2
    x = O.synthetic_images(B, 128, 256, seed=11)
    mask, delta, box, labels = TO.synthetic_labels(omc, B, seed=12)
    gh, gw = O.squeezedet_grid(128, 256)
    dm = torch.from_numpy((np.random.RandomState(13).uniform(size=(B, gh, gw, 768)) < 0.5).astype(np.float32))
    ref = TO.loss_and_grads("squeezeDet", omc, params, x, dm, mask, delta, box, labels)
    out = tr.step(x, mask, delta, box, labels, dropout_mask=dm, apply_update=False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(out["class_loss"]), ref["class_loss"], rtol=2e-4)
    np.testing.assert_allclose(float(out["conf_loss"]), ref["conf_loss"], rtol=2e-4)
    np.testing.assert_allclose(float(out["bbox_loss"]), ref["bbox_loss"], rtol=2e-4)
    _close(out["preds"], ref["preds"], rel=1e-4, what="preds (training forward, dropout on)")
    _close(out["dpreds"], ref["dpreds"], rel=2e-4, what="dpreds")
    assert "conv1/kernels" not in tr.gview                        # frozen (nets/squeezeDet.py:40-42)
    worst = 0.0
    for name, gref in ref["grads"].items():
        wdg = omc.WEIGHT_DECAY * params[name] if name.endswith("/kernels") else 0.0   # added by the optimizer kernel
        got = tr.gview[name].cpu() + wdg
        scale = float(gref.abs().max())
        err = float((got - gref).abs().max())
        worst = max(worst, err / max(scale, 1e-12))
        assert err <= 1e-3 * scale + 1e-7, "%s: grad err %g vs scale %g" % (name, err, scale)   # north_star 1e-3 rel
    # the update itself
    mom = {k: torch.zeros_like(v) for k, v in params.items()}
    p_ref, m_ref = TO.apply_gradients(omc, params, mom, ref["grads"], step=0)
    tr.opt.step(tr.flat_params, tr.flat_grads, tr.flat_accum, tr.learning_rate(), omc.MOMENTUM, omc.MAX_GRAD_NORM, 1.0)
    torch.cuda.synchronize()
    for name in ref["grads"]:
        _close(tr.view[name], p_ref[name], rel=2e-5, what=name + " after update")
    _close(tr.model.params["conv1/kernels"], params["conv1/kernels"], rel=0, what="frozen conv1")


def test_mixed_precision_training_step_vs_oracle():
    """float16 activations / activation gradients, float32 master weights and weight gradients, loss scale 256:
    the same step against the oracle restated with float16 storage (oracle/train_oracle.py _q: losses 0.2 %, every
    gradient tensor within 2 % of its largest element) and, for direction, against the float32 oracle.  The flat
    gradients come out true-scale (the loss scale is divided out by the backward-filter kernels)."""
    tr, mc, params = _trainer(dtype=torch.float16, loss_scale=256.0)
    omc = O.squeezeDet_config_for_input(128, 256)
    omc.IS_TRAINING = True
    B = 2
    x = O.synthetic_images(B, 128, 256, seed=11)
    mask, delta, box, labels = TO.synthetic_labels(omc, B, seed=12)
    gh, gw = O.squeezedet_grid(128, 256)
    dm = torch.from_numpy((np.random.RandomState(13).uniform(size=(B, gh, gw, 768)) < 0.5).astype(np.float32))
    ref16 = TO.loss_and_grads("squeezeDet", omc, params, x, dm, mask, delta, box, labels, storage="fp16")
    ref32 = TO.loss_and_grads("squeezeDet", omc, params, x, dm, mask, delta, box, labels)
    out = tr.step(x, mask, delta, box, labels, dropout_mask=dm, apply_update=False, keep_activations=True)
    torch.cuda.synchronize()
    assert out["preds"].dtype == torch.float16 and tr.flat_grads.dtype == torch.float32
    # forward: against the oracle restated with float16 storage, and (looser) the float32 oracle
    for k in ("class_loss", "conf_loss", "bbox_loss"):
        np.testing.assert_allclose(float(out[k]), ref16[k], rtol=3e-3)
        np.testing.assert_allclose(float(out[k]), ref32[k], rtol=1e-2)
    _close(out["preds"], ref16["preds"], rel=3e-3, what="preds (float16 training forward)")
    # backward: float16 storage turns 1-ulp forward differences into different ReLU / max-pool decisions for ~1e-3 of
    # the elements, which moves a gradient by ~sqrt(1e-3) however exact the backward kernels are.  So the backward is
    # checked with the oracle's forward VALUES pinned to the activations the device kept (same decisions, the
    # oracle's own float32 backward arithmetic): every gradient within 1 % of its largest element ...
    acts = {k: v.float().cpu() for k, v in out["activations"].items()}
    ref = TO.loss_and_grads("squeezeDet", omc, params, x, dm, mask, delta, box, labels, storage="fp16", override=acts)
    for name, gref in ref["grads"].items():
        wdg = omc.WEIGHT_DECAY * params[name] if name.endswith("/kernels") else 0.0
        got = tr.gview[name].cpu() + wdg
        scale = float(gref.abs().max())
        err = float((got - gref).abs().max())
        assert err <= 1e-2 * scale + 1e-7, "%s: grad err %g vs scale %g" % (name, err, scale)
        # ... and against the free-running float32 oracle the direction of every gradient agrees
        g32 = ref32["grads"][name]
        cos = float((got * g32).sum() / (got.norm() * g32.norm() + 1e-30))
        assert cos >= 0.995, "%s: cos %g vs the float32 oracle" % (name, cos)
    # the master weights stay float32 and the update is the float32 optimizer kernel
    assert tr.flat_params.dtype == torch.float32 and tr.model.params["conv12/kernels"].dtype == torch.float32


def test_mixed_precision_overflow_skips_and_rescales():
    """A loss scale that overflows float16 gradients: the step is skipped (weights, momentum, global_step untouched),
    the scale halves until the gradients are finite, and training then proceeds."""
    tr, mc, params = _trainer(seed=3, dtype=torch.float16, loss_scale=2.0 ** 40, growth_interval=4)
    omc = O.squeezeDet_config_for_input(128, 256)
    x = O.synthetic_images(2, 128, 256, seed=21)
    mask, delta, box, labels = TO.synthetic_labels(omc, 2, seed=22)
    w0 = tr.flat_params.clone()
    tr.step(x, mask, delta, box, labels)
    assert tr.skipped_steps == 1 and tr.global_step == 0 and tr.loss_scale == 2.0 ** 39
    assert torch.equal(tr.flat_params, w0) and float(tr.flat_accum.abs().sum()) == 0.0
    hist = []
    for _ in range(60):
        o = tr.step(x, mask, delta, box, labels)
        if tr.global_step:
            hist.append(float(o["class_loss"]) + float(o["conf_loss"]) + float(o["bbox_loss"]))
        if tr.global_step >= 12:
            break
    assert tr.global_step >= 12 and tr.skipped_steps > 1 and tr.loss_scale <= 65536.0
    assert np.isfinite(hist).all() and min(hist[-3:]) < hist[0], hist


def test_training_reduces_loss_on_fixed_batch():
    """A few steps on one fixed batch: the total loss must go down (sanity of signs / learning rate path)."""
    tr, mc, params = _trainer(seed=3)
    omc = O.squeezeDet_config_for_input(128, 256)
    x = O.synthetic_images(2, 128, 256, seed=21)
    mask, delta, box, labels = TO.synthetic_labels(omc, 2, seed=22)
    hist = []
    for _ in range(12):
        o = tr.step(x, mask, delta, box, labels)
        hist.append(float(o["class_loss"]) + float(o["conf_loss"]) + float(o["bbox_loss"]))
    assert tr.global_step == 12 and np.isfinite(hist).all()
    assert min(hist[-3:]) < hist[0], hist


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_side_stream_weight_gradients_equal_the_serial_order(dtype):
    """overlap_wgrad=True issues the weight-gradient launches on a second stream beside the backward-data chain, and from the
    second step on their slab reductions are ONE launch: three steps from the same start must leave bit-identical gradients
    and variables to the single-stream, reduction-per-conv order."""
    omc = O.squeezeDet_config_for_input(128, 256)
    x = O.synthetic_images(2, 128, 256, seed=41)
    mask, delta, box, labels = TO.synthetic_labels(omc, 2, seed=42)
    res = []
    for overlap, plan in ((False, False), (True, True), (True, False)):
        tr, mc, params = _trainer(seed=5, dtype=dtype, overlap_wgrad=overlap)
        tr.seed, tr.plan_wgrads = 77, plan        # plan: one slab reduction per step (ops.WgradPlan) from the second step on
        for _ in range(3):
            tr.step(x, mask, delta, box, labels)
        torch.cuda.synchronize()
        assert bool(tr._wplans) == plan
        res.append((tr.flat_grads.clone(), tr.flat_params.clone()))
    for r in res[1:]:
        assert torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1])


def test_momentum_clip_optimizer_vs_oracle():
    ops = _ops()
    mc = O.kitti_squeezeDet_config()
    rs = np.random.RandomState(8)
    shapes = {"a/kernels": (3, 3, 16, 64), "a/biases": (64,), "b/kernels": (1, 1, 64, 16), "b/biases": (16,), "c/kernels": (3, 3, 96, 384)}
    params = {k: torch.from_numpy(rs.randn(*s).astype(np.float32) * 0.1) for k, s in shapes.items()}
    grads = {k: torch.from_numpy(rs.randn(*s).astype(np.float32) * (5.0 if "a/" in k else 0.01)) for k, s in shapes.items()}
    mom = {k: torch.from_numpy(rs.randn(*s).astype(np.float32) * 0.01) for k, s in shapes.items()}
    # oracle: weight decay is part of the gradient (TF differentiates the total loss), then clip, then momentum
    g_ref = {k: g + (mc.WEIGHT_DECAY * params[k] if k.endswith("kernels") else 0) for k, g in grads.items()}
    p_ref, m_ref = TO.apply_gradients(mc, params, mom, g_ref, step=20000)
    offs, cnts, decs, o = [], [], [], 0
    for k, s in shapes.items():
        offs.append(o); cnts.append(int(np.prod(s))); decs.append(mc.WEIGHT_DECAY if k.endswith("kernels") else 0.0)
        o += (cnts[-1] + 63) // 64 * 64
    flat = lambda d: torch.cat([F.pad(d[k].reshape(-1), (0, (c + 63) // 64 * 64 - c)) for k, c in zip(shapes, cnts)]).to(DEV)
    P, G, M = flat(params), flat(grads), flat(mom)
    opt = ops.MomentumOptimizer(offs, cnts, decs, DEV)
    opt.step(P, G, M, TO.learning_rate(mc, 20000), mc.MOMENTUM, mc.MAX_GRAD_NORM)
    torch.cuda.synchronize()
    assert TO.learning_rate(mc, 20000) == 0.01 * 0.25
    for k, off, c in zip(shapes, offs, cnts):
        _close(P[off:off + c].reshape(shapes[k]), p_ref[k], rel=1e-6, what=k + " param")
        _close(M[off:off + c].reshape(shapes[k]), m_ref[k], rel=1e-6, what=k + " momentum")


@pytest.mark.parametrize("cfg", ["squeezeDet", "res50"])
def test_build_labels_vs_oracle(cfg):
    """sqdet_build_labels (anchor assignment + dense placeholders, imdb.py:195-239 + train.py:163-224) against the
    restated per-image Python: anchor indices bit-exact, incl. boxes competing for one anchor (identical boxes),
    a box that overlaps nothing (nearest free anchor) and empty images."""
    ops = _ops()
    mc = O.kitti_squeezeDet_config() if cfg == "squeezeDet" else O.kitti_res50_config()
    rs = np.random.RandomState(77)
    B, M, A, C = 6, 12, mc.ANCHORS, mc.CLASSES
    gt = np.zeros((B, M, 4), np.float64)
    cls = rs.randint(0, C, size=(B, M)).astype(np.int32)
    cnt = np.array([8, 12, 0, 5, 3, 1], np.int32)
    for b in range(B):
        n = cnt[b]
        gt[b, :n] = np.stack([rs.uniform(0, mc.IMAGE_WIDTH, n), rs.uniform(0, mc.IMAGE_HEIGHT, n), rs.uniform(10, 400, n), rs.uniform(10, 250, n)], 1)
    gt[1, 1] = gt[1, 0]; gt[1, 2] = gt[1, 0]                  # three identical boxes: 2nd / 3rd take the next-best anchors
    gt[3, 2] = [-900.0, -700.0, 30.0, 20.0]                   # overlaps no anchor: nearest free anchor (imdb.py:222-229)
    gt[4, 0] = [mc.ANCHOR_BOX[5000][0], mc.ANCHOR_BOX[5000][1], mc.ANCHOR_BOX[5000][2], mc.ANCHOR_BOX[5000][3]]   # IoU exactly 1
    mask, delta, box, lab, aidx = ops.build_labels(mc.ANCHOR_BOX, gt, cls, cnt, C, device=DEV)
    torch.cuda.synchronize()
    r_mask, r_delta = np.zeros((B, A), np.float32), np.zeros((B, A, 4), np.float32)
    r_box, r_lab = np.zeros((B, A, 4), np.float32), np.zeros((B, A, C), np.float32)
    for b in range(B):
        aidxs, deltas = TO.assign_anchors(mc, gt[b, :cnt[b]])
        assert len(set(aidxs)) == len(aidxs)
        assert aidx[b, :cnt[b]].cpu().tolist() == aidxs, "image %d" % b
        assert (aidx[b, cnt[b]:] == -1).all()
        for j, a in enumerate(aidxs):
            r_mask[b, a] = 1
            r_delta[b, a] = np.asarray(deltas[j], np.float64).astype(np.float32)
            r_box[b, a] = gt[b, j].astype(np.float32)
            r_lab[b, a, cls[b, j]] = 1
    assert aidx[4, 0].item() == 5000
    assert np.array_equal(mask.cpu().numpy(), r_mask) and np.array_equal(lab.cpu().numpy(), r_lab)
    assert np.array_equal(box.cpu().numpy(), r_box)
    np.testing.assert_allclose(delta.cpu().numpy(), r_delta, rtol=1e-6, atol=1e-7)


def test_build_labels_crowded_images_vs_oracle():
    """Many boxes crowding the same anchors: the per-box candidates of the parallel first pass collide again and again and
    the in-order resolution (labels_resolve_kernel) falls back to the sweep over the free anchors -- same picks as the
    sequential reference loop (imdb.py:195-239)."""
    ops = _ops()
    mc = O.kitti_squeezeDet_config()
    rs = np.random.RandomState(5)
    B, M = 4, 32
    gt = np.zeros((B, M, 4), np.float64)
    cnt = np.array([32, 32, 17, 32], np.int32)
    for b in range(B):
        c = np.array([rs.uniform(200, 1000), rs.uniform(100, 280), rs.uniform(40, 200), rs.uniform(40, 150)])
        gt[b] = c + rs.uniform(-3.0, 3.0, (M, 4)) * (1.0 if b else 0.0)      # image 0: 32 IDENTICAL boxes
    gt[3, :, 0] -= 5000.0                                                      # image 3: all far outside (distance fallback, crowded)
    cls = rs.randint(0, mc.CLASSES, size=(B, M)).astype(np.int32)
    mask, delta, box, lab, aidx = ops.build_labels(mc.ANCHOR_BOX, gt, cls, cnt, mc.CLASSES, device=DEV)
    torch.cuda.synchronize()
    for b in range(B):
        aidxs, _ = TO.assign_anchors(mc, gt[b, :cnt[b]])
        assert aidx[b, :cnt[b]].cpu().tolist() == aidxs, "image %d" % b
        assert int(mask[b].sum().item()) == cnt[b]


def test_training_step_from_gpu_built_labels():
    """The trainer accepts the device tensors of build_labels directly (no host round trip of the labels)."""
    ops = _ops()
    tr, mc, params = _trainer()
    omc = O.squeezeDet_config_for_input(128, 256)
    rs = np.random.RandomState(3)
    gt = np.stack([rs.uniform(0, 256, (2, 4)), rs.uniform(0, 128, (2, 4)), rs.uniform(20, 120, (2, 4)), rs.uniform(20, 90, (2, 4))], 2)
    cls, cnt = rs.randint(0, 3, (2, 4)).astype(np.int32), np.array([4, 2], np.int32)
    mask, delta, box, lab, _ = ops.build_labels(omc.ANCHOR_BOX, gt, cls, cnt, 3, device=DEV)
    x = O.synthetic_images(2, 128, 256, seed=9)
    out = tr.step(x, mask, delta, box, lab)
    assert float(out["num_objects"]) == 6.0 and np.isfinite(float(out["bbox_loss"]))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_graphed_step_equals_the_eager_step(dtype):
    """train.GraphedStep (label build on the side stream, forward, loss, backward captured as one hipGraph; all-reduce + update
    eager) against trainer.step on device-built labels: four steps from the same start -- dropout masks from the same
    counter stream -- leave bit-identical variables, momentum and reported losses."""
    from squeezedet_amd.train import GraphedStep
    ops = _ops()
    omc = O.squeezeDet_config_for_input(128, 256)
    rs = np.random.RandomState(31)
    B, M = 2, 5
    gt = torch.from_numpy(np.stack([rs.uniform(0, 256, (B, M)), rs.uniform(0, 128, (B, M)), rs.uniform(20, 120, (B, M)),
                                    rs.uniform(20, 90, (B, M))], 2)).to(DEV)
    cls = torch.from_numpy(rs.randint(0, 3, (B, M)).astype(np.int32)).to(DEV)
    cnt = torch.from_numpy(np.array([5, 3], np.int32)).to(DEV)
    xs = [O.synthetic_images(B, 128, 256, seed=60 + i).to(DEV, dtype) for i in range(4)]
    res = []
    for graphed in (False, True):
        tr, mc, params = _trainer(seed=9, dtype=dtype)
        tr.seed = 1234
        gs = GraphedStep(tr, torch.from_numpy(omc.ANCHOR_BOX).to(DEV), 3) if graphed else None
        losses = []
        for x in xs:
            if graphed:
                out = gs.step(x, gt, cls, cnt)
            else:
                out = tr.step(x, *ops.build_labels(omc.ANCHOR_BOX, gt, cls, cnt, 3, device=DEV)[:4])
            losses.append([float(out[k]) for k in ("class_loss", "conf_loss", "bbox_loss")])
        torch.cuda.synchronize()
        res.append((tr.flat_params.clone(), tr.flat_accum.clone(), losses))
    assert res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_squeezedet_plus_training_step_vs_oracle():
    """The same trainer on SqueezeDet+ (nets/squeezeDetPlus.py:30-79: 7x7/s2 VALID conv1 -- frozen --, VALID pools,
    wider fire modules; train.py --net squeezeDet+) at its full 1242x375 size, batch 1: losses and every gradient
    against the oracle's autograd."""
    import squeezedet_amd as S
    from squeezedet_amd import nets
    from squeezedet_amd.train import SqueezeDetTrainer
    mc = S.kitti_squeezeDetPlus_config()
    mc.LOAD_PRETRAINED_MODEL = False
    mc.IS_TRAINING = True
    mc.BATCH_SIZE = 1
    m = nets.SqueezeDetPlus(mc, gpu_id="0", dtype=torch.float32)
    params = O.init_params("squeezeDet+", seed=1)
    m.load_params(params)
    tr = SqueezeDetTrainer(m)
    omc = O.kitti_squeezeDetPlus_config()
    omc.IS_TRAINING = True
    x = O.synthetic_images(1, mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, seed=51)
    mask, delta, box, labels = TO.synthetic_labels(omc, 1, seed=52)
    dshape = tuple(tr.model.preds.inputs[0].get_shape())        # conv12's input = the dropout's output [1,22,76,512]
    dm = torch.from_numpy((np.random.RandomState(53).uniform(size=dshape) < 0.5).astype(np.float32))
    ref = TO.loss_and_grads("squeezeDet+", omc, params, x, dm, mask, delta, box, labels)
    out = tr.step(x, mask, delta, box, labels, dropout_mask=dm, apply_update=False, keep_activations=True)
    torch.cuda.synchronize()
    for k in ("class_loss", "conf_loss", "bbox_loss"):
        np.testing.assert_allclose(float(out[k]), ref[k], rtol=5e-4)
    _close(out["preds"], ref["preds"], rel=1e-3, what="preds (SqueezeDet+ training forward)")
    assert "conv1/kernels" not in tr.gview
    # gradients: with the oracle's forward values pinned to the device's activations (same ReLU / max-pool decisions;
    # at this depth and width even float32 summation-order differences flip a few near-zero ReLUs, and ONE flipped
    # element with a large gradient moved a filter gradient by 3e-3 of its maximum in the free-running comparison)
    acts = {k: v.float().cpu() for k, v in out["activations"].items()}
    ref = TO.loss_and_grads("squeezeDet+", omc, params, x, dm, mask, delta, box, labels, override=acts)
    for name, gref in ref["grads"].items():
        wdg = omc.WEIGHT_DECAY * params[name] if name.endswith("/kernels") else 0.0
        got = tr.gview[name].cpu() + wdg
        scale = float(gref.abs().max())
        err = float((got - gref).abs().max())
        assert err <= 1e-3 * scale + 1e-7, "%s: grad err %g vs scale %g" % (name, err, scale)


def test_squeezedet_full_size_training_step_vs_oracle():
    """BASELINE configs[2]'s kernel selection (bench.py sqdet_train_fp32: 384x1248, batch 20): at full size the training
    forward / backward run other kernels than at the 128x256 of test_full_training_step_vs_oracle -- conv1x1_deepk on every
    >= 8 k-pixel 1x1 (batch 5 puts the 24x78 maps above that too), fire_stream's keep forms, many-slab weight gradients, the
    index pools on 192x624 maps.  One float32 step at 5x384x1248: losses, preds and every gradient against the oracle's
    autograd (forward values pinned to the device's activations for the gradients, as in the SqueezeDet+ test below)."""
    B, size = 5, (384, 1248)
    tr, mc, params = _trainer(size, B)
    omc = O.squeezeDet_config_for_input(*size)
    omc.IS_TRAINING = True
    x = O.synthetic_images(B, size[0], size[1], seed=61)
    mask, delta, box, labels = TO.synthetic_labels(omc, B, seed=62)
    gh, gw = O.squeezedet_grid(*size)
    dm = torch.from_numpy((np.random.RandomState(63).uniform(size=(B, gh, gw, 768)) < 0.5).astype(np.float32))
    ref = TO.loss_and_grads("squeezeDet", omc, params, x, dm, mask, delta, box, labels)
    out = tr.step(x, mask, delta, box, labels, dropout_mask=dm, apply_update=False, keep_activations=True)
    torch.cuda.synchronize()
    for k in ("class_loss", "conf_loss", "bbox_loss"):
        np.testing.assert_allclose(float(out[k]), ref[k], rtol=5e-4)
    _close(out["preds"], ref["preds"], rel=1e-3, what="preds (full-size training forward)")
    _close(out["dpreds"], ref["dpreds"], rel=1e-3, what="dpreds")
    acts = {k: v.float().cpu() for k, v in out["activations"].items()}
    ref = TO.loss_and_grads("squeezeDet", omc, params, x, dm, mask, delta, box, labels, override=acts)
    assert set(ref["grads"]) == set(tr.gview)
    for name, gref in ref["grads"].items():
        wdg = omc.WEIGHT_DECAY * params[name] if name.endswith("/kernels") else 0.0
        got = tr.gview[name].cpu() + wdg
        scale = float(gref.abs().max())
        err = float((got - gref).abs().max())
        assert err <= 1e-3 * scale + 1e-7, "%s: grad err %g vs scale %g" % (name, err, scale)   # north_star 1e-3 rel


@pytest.mark.parametrize("name", __import__("tests.golden.cases", fromlist=["x"]).LABEL_CASES)
def test_build_labels_vs_reference_golden(name):
    """sqdet_build_labels against what the REFERENCE's own imdb.read_batch returned (dataset/imdb.py:120-260, run
    unchanged by tests/golden/make_golden.py on seeded annotations; tests/golden/labels.npz): anchor indices
    bit-exact -- boxes competing for one anchor, boxes that overlap nothing, three anchor sets -- and the float32
    deltas / boxes equal to the float64 golden values rounded once (1e-6)."""
    from tests.golden import cases
    ops = _ops()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "labels.npz"))
    cfg = name.split("_")[0]
    mc = {"squeezeDet": O.kitti_squeezeDet_config, "squeezeDetPlus": O.kitti_squeezeDetPlus_config, "res50": O.kitti_res50_config}[cfg]()
    aidx, delta, bbox, label, cnt = [g[name + k] for k in ("_aidx", "_delta", "_bbox", "_label", "_count")]
    B, M = aidx.shape
    assert M == cases.LABEL_MAX_OBJECTS
    cls = np.where(label >= 0, label, 0).astype(np.int32)
    mask, d_delta, d_box, lab, d_aidx = ops.build_labels(mc.ANCHOR_BOX, bbox.copy(), cls, cnt.astype(np.int32), mc.CLASSES, device=DEV)
    torch.cuda.synchronize()
    assert np.array_equal(d_aidx.cpu().numpy().astype(np.int64), aidx), "anchor picks differ from the reference's"
    dm, dd, db, dl = mask.cpu().numpy(), d_delta.cpu().numpy(), d_box.cpu().numpy(), lab.cpu().numpy()
    assert dm.sum() == cnt.sum()
    for b in range(B):
        for k in range(int(cnt[b])):
            a = int(aidx[b, k])
            assert dm[b, a] == 1 and dl[b, a, int(label[b, k])] == 1 and dl[b, a].sum() == 1
            np.testing.assert_allclose(dd[b, a], delta[b, k], rtol=1e-6, atol=1e-7)
            assert np.array_equal(db[b, a], bbox[b, k].astype(np.float32))


def test_pack_many_equals_the_per_kernel_packers():
    """sqdet_conv_pack_many (ops.PackPlan: every trainable kernel of a step packed by one launch, forward and backward-data
    orders) writes exactly the bytes of sqdet_conv_pack_weights / sqdet_conv_pack_weights_bwd_data, float16 and float32,
    for SqueezeDet's kernel shapes incl. ragged channel counts -- and re-packs in place after the weights changed."""
    import collections
    ops = _ops()
    rs = np.random.RandomState(9)
    shapes = [(1, 64, 16), (1, 16, 64), (3, 16, 64), (1, 256, 48), (3, 48, 192), (1, 768, 96), (3, 96, 384), (3, 768, 72), (3, 24, 40)]
    for tdt in (torch.float16, torch.float32):
        ws = collections.OrderedDict(("k%d" % i, torch.from_numpy(rs.randn(k, k, ci, co).astype(np.float32)).to(DEV)) for i, (k, ci, co) in enumerate(shapes))
        plan = ops.PackPlan(ws, tdt, bwd_names=set(ws))
        for rep in range(2):
            plan.run()
            torch.cuda.synchronize()
            for n, w in ws.items():
                assert torch.equal(plan.fwd[n].data, ops.pack_conv_weights(w, tdt).data), (n, tdt)
                assert torch.equal(plan.bwd[n].data, ops.PackedConvBwd(w, tdt).data), (n, tdt)
            for w in ws.values():
                w.mul_(-0.5)                       # in place: the plan reads the same storage
