"""GPU parity tests of the ResNet50+ConvDet path (BASELINE.json config 5):
the BN fold kernel, the residual-add conv epilogue, the builder graph and the native plan
(SQDET_ARCH_RESNET50) against oracle/resnet_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import resnet_oracle as R
from oracle import sqdet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, ref, dtype, what):
    got = got.float().cpu().numpy()
    ref = ref.numpy() if isinstance(ref, torch.Tensor) else ref
    assert got.shape == ref.shape, what
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    tol = 1e-3 * scale + 1e-5 if dtype == torch.float32 else 1e-2 * scale + 1e-3   # north_star: 1e-3 rel in fp32
    assert err <= tol, "%s: max err %g vs scale %g" % (what, err, scale)


@pytest.mark.parametrize("with_bias", [False, True])
def test_fold_batchnorm_kernel(with_bias):
    from squeezedet_amd import ops
    g = torch.Generator().manual_seed(7)
    k, cin, cout = 3, 24, 40
    w = torch.randn(k, k, cin, cout, generator=g)
    cb = torch.randn(cout, generator=g) if with_bias else None
    gamma, var = torch.rand(cout, generator=g) + 0.5, torch.rand(cout, generator=g) + 0.5
    beta, mean = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    wf, bf = ops.fold_batchnorm(w.to(DEV), cb.to(DEV) if with_bias else None, gamma.to(DEV), beta.to(DEV), mean.to(DEV),
                                var.to(DEV), R.BN_EPS)
    rw, rb = R.fold_batchnorm(w, cb, gamma, beta, mean, var)
    # same float32 operations in the same order; the device sqrt/divide may differ by an ulp
    assert torch.allclose(wf.cpu(), rw, rtol=5e-7, atol=0)
    assert torch.allclose(bf.cpu(), rb, rtol=5e-7, atol=1e-7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("shape", [(1, 13, 17, 64, 256, 1), (2, 9, 11, 256, 1024, 1), (1, 8, 12, 32, 48, 3),
                                   (1, 24, 78, 512, 2048, 1), (3, 7, 5, 128, 512, 1)],
                         ids=["64->256", "256->1024", "3x3", "512->2048", "128->512-ragged"])
def test_conv_residual_add_epilogue(dtype, shape):
    """sqdet_conv2d_add_nhwc_fwd: y = relu(conv(x) + b + y), the residual add of resnet50_convDet.py:55."""
    from squeezedet_amd import ops
    n, h, w, cin, cout, k = shape
    st = "fp16" if dtype == torch.float16 else "fp32"
    g = torch.Generator().manual_seed(11)
    x = O._round_storage(torch.randn(n, h, w, cin, generator=g), st)
    sc = O._round_storage(torch.randn(n, h, w, cout, generator=g), st)
    wt = O._round_storage(torch.randn(k, k, cin, cout, generator=g) / (k * cin ** 0.5), st)
    b = torch.randn(cout, generator=g) * 0.1
    ref = O._round_storage(torch.relu(O.conv_layer(x, wt, b, 1, "SAME", False, "fp32") + sc), st)
    out = sc.to(DEV, dtype).contiguous()
    y = ops.conv2d_nhwc(x.to(DEV, dtype), ops.pack_conv_weights(wt.to(DEV), dtype), b.to(DEV), 1, "SAME", True, out=out,
                        accumulate=True)
    assert y.data_ptr() == out.data_ptr()
    _close(y, ref, dtype, "conv+add+relu")
    # sqdet_conv2d_res_nhwc_fwd: the shortcut in a tensor of its own, left untouched -- bitwise the in-place form's result (1x1: the
    # pipelined GEMM tile reads the residual tile by LDS-DMA; 3x3: copy + add)
    res = sc.to(DEV, dtype).contiguous()
    keep = res.clone()
    y2 = ops.conv2d_nhwc(x.to(DEV, dtype), ops.pack_conv_weights(wt.to(DEV), dtype), b.to(DEV), 1, "SAME", True, residual=res)
    torch.cuda.synchronize()
    assert y2.data_ptr() != res.data_ptr() and torch.equal(res, keep), "the residual tensor was modified"
    assert torch.equal(y2, y), "residual form differs from the in-place form"


def _model(dtype, batch, size, seed=0):
    import squeezedet_amd as S
    from squeezedet_amd import nets
    mc = S.kitti_res50_config_for_input(*size)
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = batch
    m = nets.ResNet50ConvDet(mc, gpu_id="0", dtype=dtype)
    params = R.init_params(seed=seed)
    m.load_params(params)
    return m, mc, params


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_resnet50_block_by_block_vs_oracle(dtype):
    """Builder-graph path: conv1, pool1 and every residual block output against the oracle's."""
    size = (135, 200)
    st = "fp16" if dtype == torch.float16 else "fp32"
    m, mc, params = _model(dtype, 2, size)
    x = O.synthetic_images(2, size[0], size[1], seed=1, storage=st)
    col = {}
    R.forward(params, x, st, collect=col)
    nodes, stack, seen = {}, [m.preds], set()
    while stack:
        nd = stack.pop()
        if nd in seen:
            continue
        seen.add(nd)
        if nd.name in col and nd.op in ("conv_bn", "pool", "add_relu", "conv"):
            nodes[nd.name] = nd
        stack.extend(nd.inputs)
    assert set(nodes) == set(col)
    names = list(col)
    outs = m.run([nodes[n] for n in names], {m.image_input: x}, use_plan=False)
    for n, got in zip(names, outs):
        _close(got, col[n], dtype, n)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_resnet50_plan_equals_graph_and_oracle(dtype):
    """Native plan (SQDET_ARCH_RESNET50: lazy BN fold + in-place residual epilogue) == builder graph bit
    for bit, both within tolerance of the oracle; then detections through interpret_output."""
    size = (135, 200)
    st = "fp16" if dtype == torch.float16 else "fp32"
    m, mc, params = _model(dtype, 2, size, seed=2)
    x = O.synthetic_images(2, size[0], size[1], seed=4, storage=st)
    ref = R.forward(params, x, st)
    (pg,) = m.run([m.preds], {m.image_input: x}, use_plan=False)
    (pp,) = m.run([m.preds], {m.image_input: x}, use_plan=True)
    assert torch.equal(pg, pp)
    _close(pp, ref, dtype, "preds")
    # a parameter update re-folds: scale one gamma, plan and graph must follow
    name = "conv4_x/res4b/res4b_branch2/res4b_branch2b/gamma"
    params2 = dict(params)
    params2[name] = params[name] * 1.5
    m.load_params({name: params2[name]})
    ref2 = R.forward(params2, x, st)
    (pp2,) = m.run([m.preds], {m.image_input: x}, use_plan=True)
    assert not torch.equal(pp2, pp)
    _close(pp2, ref2, dtype, "preds after gamma update")
    # decode + filter run on the plan's preds like every other net
    boxes, probs, cls = m.detect(x)
    r = O.interpret_output(pp2.float().cpu().numpy(), mc)
    assert np.allclose(boxes.cpu().numpy(), r["det_boxes"], atol=1e-3)
    assert np.allclose(probs.cpu().numpy(), r["det_probs"], atol=1e-5)
    assert (cls.cpu().numpy() == r["det_class"]).mean() > 0.999      # equal up to exact class-score ties


def test_resnet50_full_size_properties():
    """BASELINE.json config 5 shape (375x1242, batch 8, fp16) on the plan: finite, deterministic,
    per-image independence (image i alone gives the same preds as inside the batch)."""
    m, mc, params = _model(torch.float16, 8, (375, 1242))
    x = O.synthetic_images(8, 375, 1242, seed=9, storage="fp16")
    (p1,) = m.run([m.preds], {m.image_input: x})
    (p2,) = m.run([m.preds], {m.image_input: x})
    assert tuple(p1.shape) == (8, 24, 78, 72)
    assert torch.isfinite(p1.float()).all()
    assert torch.equal(p1, p2)
    assert float(p1.float().abs().max()) > 0.1
    m1, _, _ = _model(torch.float16, 1, (375, 1242))
    (q,) = m1.run([m1.preds], {m1.image_input: x[5:6]})
    assert torch.equal(q[0], p1[5])


# ------------------------------------------------------------------ training (BASELINE.json config 5)
def test_fold_batchnorm_backward_and_subsample():
    from squeezedet_amd import ops
    g = torch.Generator().manual_seed(17)
    k, cin, cout = 3, 20, 72
    w = torch.randn(k, k, cin, cout, generator=g).requires_grad_(True)
    gamma = (torch.rand(cout, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(cout, generator=g).requires_grad_(True)
    mean, var = torch.randn(cout, generator=g), torch.rand(cout, generator=g) + 0.5
    dwf, dbf = torch.randn(k, k, cin, cout, generator=g), torch.randn(cout, generator=g)
    wf, bf = R.fold_batchnorm(w, None, gamma, beta, mean, var)
    ((wf * dwf).sum() + (bf * dbf).sum()).backward()
    dw, dg, db = ops.fold_batchnorm_bwd(w.detach().to(DEV), dwf.to(DEV), dbf.to(DEV), None, gamma.detach().to(DEV), mean.to(DEV),
                                        var.to(DEV), R.BN_EPS)
    assert torch.allclose(dw.cpu(), w.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(dg.cpu(), gamma.grad, rtol=1e-4, atol=1e-4)
    assert torch.equal(db.cpu(), beta.grad)
    x = torch.randn(2, 7, 9, 16, generator=g)
    for dt in (torch.float32, torch.float16):
        assert torch.equal(ops.subsample_nhwc(x.to(DEV, dt), 2).cpu(), x.to(dt)[:, ::2, ::2, :])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_pack_plan_folds_batchnorm_and_fold_backward_plan(dtype):
    """ops.PackPlan with bn (sqdet_conv_pack_many_prepare_bn): the packed kernels and folded biases of many _conv_bn_layer convs
    in one launch, bitwise fold_batchnorm + the per-kernel packers; ops.FoldBwdPlan (sqdet_fold_batchnorm_bwd_many): bitwise
    fold_batchnorm_bwd per conv."""
    from squeezedet_amd import ops
    g = torch.Generator().manual_seed(23)
    shapes = [("a", 1, 64, 256), ("b", 3, 64, 64), ("c", 1, 256, 64), ("d", 3, 24, 72), ("plain", 3, 32, 72)]
    W, bn, fold_items, want = {}, {}, [], {}
    for name, k, cin, cout in shapes:
        W[name] = torch.randn(k, k, cin, cout, generator=g).to(DEV)
        if name != "plain":
            gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(DEV), torch.randn(cout, generator=g).to(DEV)
            mean, var = torch.randn(cout, generator=g).to(DEV), (torch.rand(cout, generator=g) + 0.5).to(DEV)
            bn[name] = (gamma, beta, mean, var, None)
            dwf, dbf = torch.randn(k, k, cin, cout, generator=g).to(DEV), torch.randn(cout, generator=g).to(DEV)
            outs = [torch.empty_like(W[name]), torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)]
            fold_items.append((W[name], dwf, dbf, None, gamma, mean, var) + tuple(outs))
            want[name] = ops.fold_batchnorm_bwd(W[name], dwf, dbf, None, gamma, mean, var, R.BN_EPS)
    plan = ops.PackPlan(W, dtype, bwd_names=("a", "b", "c", "plain"), bn=bn, eps=R.BN_EPS)
    fplan = ops.FoldBwdPlan(fold_items, R.BN_EPS)
    for _ in range(2):
        plan.run()
        fplan.run()
        torch.cuda.synchronize()
        for name, k, cin, cout in shapes:
            if name in bn:
                wf, bf = ops.fold_batchnorm(W[name], None, bn[name][0], bn[name][1], bn[name][2], bn[name][3], R.BN_EPS)
                assert torch.equal(plan.bias[name], bf), name
            else:
                wf = W[name]
            assert torch.equal(plan.fwd[name].data, ops.pack_conv_weights(wf, dtype).data), name
            if name in plan.bwd:
                assert torch.equal(plan.bwd[name].data, ops.PackedConvBwd(wf, dtype).data), name
        for it in fold_items:
            name = [n for n in W if W[n] is it[0]][0]
            for got, ref in zip(it[7:], want[name]):
                assert torch.equal(got, ref), name


def _trainer(size=(96, 160), batch=2, seed=0, dtype=torch.float32, **kw):
    import squeezedet_amd as S
    from squeezedet_amd import nets
    from squeezedet_amd.train import ResNet50ConvDetTrainer
    mc = S.kitti_res50_config_for_input(*size)
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = batch
    mc.IS_TRAINING = True
    m = nets.ResNet50ConvDet(mc, gpu_id="0", dtype=dtype)
    params = R.init_params(seed=seed)
    m.load_params(params)
    return ResNet50ConvDetTrainer(m, **kw), mc, params


def test_resnet50_training_step_vs_oracle():
    """forward(train) -> loss -> backward -> clipped Momentum on the trainable res4* + conv5 variables (kernels,
    gamma, beta; nets/resnet50_convDet.py:94-132) against PyTorch-CPU autograd of the UNFOLDED graph."""
    from oracle import train_oracle as TO
    size, B = (96, 160), 2
    tr, mc, params = _trainer(size, B)
    x = O.synthetic_images(B, size[0], size[1], seed=31)
    mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=32)
    gh, gw = tr.model.preds.get_shape()[1:3]
    dm = torch.from_numpy((np.random.RandomState(33).uniform(size=(B, gh, gw, 1024)) < 0.5).astype(np.float32))
    ref = R.loss_and_grads(mc, params, x, dm, mask, delta, box, labels)
    out = tr.step(x, mask, delta, box, labels, dropout_mask=dm, apply_update=False)
    torch.cuda.synchronize()
    for k in ("class_loss", "conf_loss", "bbox_loss"):
        np.testing.assert_allclose(float(out[k]), ref[k], rtol=5e-4)
    _close(out["preds"], ref["preds"], torch.float32, "preds (training forward)")
    assert set(tr.names) == set(ref["grads"])
    assert "conv3_x/res3d/res3d_branch2/res3d_branch2c/kernels" not in tr.gview and "conv4_x/res4a/res4a_branch1/mean" not in tr.gview
    for name, gref in ref["grads"].items():
        wdg = mc.WEIGHT_DECAY * params[name] if name.endswith("/kernels") else 0.0   # added by the optimizer kernel
        got = tr.gview[name].cpu() + wdg
        scale = float(gref.abs().max())
        err = float((got - gref).abs().max())
        assert err <= 2e-3 * scale + 1e-7, "%s: grad err %g vs scale %g" % (name, err, scale)
    mom = {k: torch.zeros_like(v) for k, v in params.items()}
    p_ref, _ = TO.apply_gradients(mc, params, mom, ref["grads"], step=0)
    tr.opt.step(tr.flat_params, tr.flat_grads, tr.flat_accum, tr.learning_rate(), mc.MOMENTUM, mc.MAX_GRAD_NORM, 1.0)
    torch.cuda.synchronize()
    for name in ref["grads"]:
        got, want = tr.view[name].cpu(), p_ref[name]
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-7, name


def test_resnet50_mixed_precision_training_step_vs_oracle():
    """BASELINE.json configs[4] (ResNet50+ConvDet float16 training): float16 activations and activation gradients,
    float32 master weights / weight gradients / optimizer, against the float32 oracle of the unfolded graph.  The
    synthetic weights make very large activation gradients, so the loss scale is searched downwards the way the
    trainer's overflow handling does (a scale whose gradients are all finite)."""
    from oracle import train_oracle as TO
    size, B = (96, 160), 2
    tr, mc, params = _trainer(size, B, dtype=torch.float16, loss_scale=256.0)
    x = O.synthetic_images(B, size[0], size[1], seed=31)
    mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=32)
    gh, gw = tr.model.preds.get_shape()[1:3]
    dm = torch.from_numpy((np.random.RandomState(33).uniform(size=(B, gh, gw, 1024)) < 0.5).astype(np.float32))
    ref = R.loss_and_grads(mc, params, x, dm, mask, delta, box, labels, storage="fp16")   # float16-storage restatement
    ref32 = R.loss_and_grads(mc, params, x, dm, mask, delta, box, labels)
    for _ in range(24):
        out = tr.step(x, mask, delta, box, labels, dropout_mask=dm, apply_update=False, keep_activations=True)
        if bool(torch.isfinite(tr.flat_grads).all()):
            break
        tr.loss_scale /= 4.0
    torch.cuda.synchronize()
    assert out["preds"].dtype == torch.float16 and bool(torch.isfinite(tr.flat_grads).all()), tr.loss_scale
    for k in ("class_loss", "conf_loss", "bbox_loss"):
        np.testing.assert_allclose(float(out[k]), ref[k], rtol=2e-2)
        np.testing.assert_allclose(float(out[k]), ref32[k], rtol=3e-2)
    _close(out["preds"], ref["preds"], torch.float16, "preds (float16 training forward)")
    # backward: with the oracle's forward values pinned to the activations the device kept (same ReLU decisions, same
    # preds -- these synthetic weights give |preds| up to ~50, so d(loss)/d(preds) is very sensitive to them), every
    # gradient within 1 % of its largest element; see tests/test_gpu_train.py for the reasoning
    acts = {k: v.float().cpu() for k, v in out["activations"].items()}
    pinned = R.loss_and_grads(mc, params, x, dm, mask, delta, box, labels, storage="fp16", override=acts)
    for name, gref in pinned["grads"].items():
        wdg = mc.WEIGHT_DECAY * params[name] if name.endswith("/kernels") else 0.0
        got = tr.gview[name].cpu() + wdg
        scale = float(gref.abs().max())
        err = float((got - gref).abs().max())
        assert err <= 1e-2 * scale + 1e-7, "%s: grad err %g vs scale %g" % (name, err, scale)
        g32 = ref32["grads"][name]
        cos = float((got * g32).sum() / (got.norm() * g32.norm() + 1e-30))
        assert cos >= 0.99, "%s: cos %g vs the float32 oracle" % (name, cos)
    # a few real steps (dynamic loss scale; overflowed steps are skipped, not applied)
    hist = []
    for i in range(10):
        o = tr.step(x, mask, delta, box, labels, dropout_mask=dm)
        hist.append(float(o["class_loss"]) + float(o["conf_loss"]) + float(o["bbox_loss"]))
    # (these synthetic weights produce activation gradients of ~1e5: several of the ten steps overflow float16 at the
    # scale found above, are skipped on the device and halve the scale -- how many depends on summation order)
    assert tr.global_step + tr.skipped_steps == 10 and tr.global_step >= 3 and np.isfinite(hist).all()
    assert min(hist[1:]) < hist[0], hist


def test_resnet50_full_size_mixed_precision_training_step_vs_oracle():
    """BASELINE configs[4]'s kernel selection (bench.py res50_train_fp16: 375x1242, batch 8): one mixed-precision step at
    5x375x1242 -- the 47x155 and 24x78 maps are then above the 8 k-pixel threshold of the streaming 1x1 as at batch 8, the 7x7
    stem, the 93x310 pools and the many-slab weight gradients run at their benchmark shapes -- losses and preds against the
    float16-storage oracle, every gradient against the oracle's backward with the forward values pinned to the device's
    activations (tests/test_gpu_train.py has the reasoning)."""
    from oracle import train_oracle as TO
    size, B = (375, 1242), 5
    tr, mc, params = _trainer(size, B, dtype=torch.float16, loss_scale=256.0)
    x = O.synthetic_images(B, size[0], size[1], seed=71)
    mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=72)
    gh, gw = tr.model.preds.get_shape()[1:3]
    assert (gh, gw) == (24, 78)
    dm = torch.from_numpy((np.random.RandomState(73).uniform(size=(B, gh, gw, 1024)) < 0.5).astype(np.float32))
    ref = R.loss_and_grads(mc, params, x, dm, mask, delta, box, labels, storage="fp16")
    for _ in range(24):
        out = tr.step(x, mask, delta, box, labels, dropout_mask=dm, apply_update=False, keep_activations=True)
        if bool(torch.isfinite(tr.flat_grads).all()):
            break
        tr.loss_scale /= 4.0
    torch.cuda.synchronize()
    assert out["preds"].dtype == torch.float16 and bool(torch.isfinite(tr.flat_grads).all()), tr.loss_scale
    for k in ("class_loss", "conf_loss", "bbox_loss"):
        np.testing.assert_allclose(float(out[k]), ref[k], rtol=2e-2)
    _close(out["preds"], ref["preds"], torch.float16, "preds (full-size float16 training forward)")
    acts = {k: v.float().cpu() for k, v in out["activations"].items()}
    pinned = R.loss_and_grads(mc, params, x, dm, mask, delta, box, labels, storage="fp16", override=acts)
    assert set(pinned["grads"]) == set(tr.names)
    for name, gref in pinned["grads"].items():
        wdg = mc.WEIGHT_DECAY * params[name] if name.endswith("/kernels") else 0.0
        got = tr.gview[name].cpu() + wdg
        scale = float(gref.abs().max())
        err = float((got - gref).abs().max())
        assert err <= 1e-2 * scale + 1e-7, "%s: grad err %g vs scale %g" % (name, err, scale)


def test_resnet50_training_reduces_loss():
    from oracle import train_oracle as TO
    tr, mc, params = _trainer(seed=5)
    x = O.synthetic_images(2, 96, 160, seed=41)
    mask, delta, box, labels = TO.synthetic_labels(mc, 2, seed=42)
    gh, gw = tr.model.preds.get_shape()[1:3]
    dm = torch.from_numpy((np.random.RandomState(43).uniform(size=(2, gh, gw, 1024)) < 0.5).astype(np.float32))   # fixed dropout mask
    hist = []
    for _ in range(10):
        o = tr.step(x, mask, delta, box, labels, dropout_mask=dm)
        hist.append(float(o["class_loss"]) + float(o["conf_loss"]) + float(o["bbox_loss"]))
    assert tr.global_step == 10 and np.isfinite(hist).all()
    assert min(hist[1:]) < hist[0], hist
    # the inference path sees the updated variables (plan re-folds lazily)
    tr.model.keep_prob = 1.0
    (p,) = tr.model.run([tr.model.preds], {tr.model.image_input: x})
    assert torch.isfinite(p).all()


def test_resnet50_side_stream_weight_gradients_equal_the_serial_order():
    """Weight-gradient (+ folded-BN gradient) launches on a second stream: same bits as the single-stream order."""
    from oracle import train_oracle as TO
    size, B = (96, 160), 2
    res = []
    for overlap in (False, True):
        tr, mc, params = _trainer(size, B, seed=2, overlap_wgrad=overlap)
        tr.seed = 5
        x = O.synthetic_images(B, size[0], size[1], seed=51)
        mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=52)
        for _ in range(2):
            tr.step(x, mask, delta, box, labels)
        torch.cuda.synchronize()
        res.append((tr.flat_grads.clone(), tr.flat_params.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_resnet50_relu_backward_rides_in_the_last_contribution(dtype):
    """fuse_relu_bwd (default): the launch that delivers the last contribution to a ReLU output's gradient -- the backward-data
    conv of its reader, accumulating at the residual junctions, or the dropout backward under conv5 -- also applies the ReLU
    mask; 18 relu_bwd launches per step fewer.  Same bits as the separate launches, and no relu_bwd launch is left."""
    from oracle import train_oracle as TO
    from squeezedet_amd import ops
    size, B = (96, 160), 2
    res, calls = [], []
    real = ops.relu_bwd
    for fuse in (False, True):
        tr, mc, params = _trainer(size, B, seed=3, dtype=dtype, **({"loss_scale": 1.0} if dtype == torch.float16 else {}))
        tr.fuse_relu_bwd = fuse
        tr.seed = 7
        x = O.synthetic_images(B, size[0], size[1], seed=61)
        mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=62)
        n = [0]

        def counted(*a, **k):
            n[0] += 1
            return real(*a, **k)
        ops.relu_bwd = counted
        try:
            for _ in range(2):
                tr.step(x, mask, delta, box, labels)
        finally:
            ops.relu_bwd = real
        torch.cuda.synchronize()
        calls.append(n[0])
        res.append((tr.flat_grads.clone(), tr.flat_params.clone()))
    assert calls[0] == 2 * 18 and calls[1] == 0, calls
    assert dtype == torch.float16 or torch.isfinite(res[0][0]).all()
    bits = lambda t_: t_.view(torch.int32)       # (bit patterns: an overflowed float16 step must overflow the same way)
    assert torch.equal(bits(res[0][0]), bits(res[1][0])) and torch.equal(bits(res[0][1]), bits(res[1][1]))
