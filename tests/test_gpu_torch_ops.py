"""The PyTorch custom-op surface (squeezedet_amd/torch_ops.py: torch.ops.sqdet.*): every op is called and compared with the
ctypes path it wraps, torch.library.opcheck validates schema / fake implementation / autograd registration / aot dispatch,
and loss.backward() THROUGH the ops (conv2d, maxpool, fire: SqueezeDet's whole forward graph written with them, the way the
reference writes it with tf.nn.conv2d / max_pool, nn_skeleton.py:539-547,580, nets/squeezeDet.py:30-106) gives the oracle's
gradients (oracle/train_oracle.loss_and_grads)."""
import numpy as np
import pytest
import torch

from oracle import sqdet_oracle as O
from oracle import train_oracle as TO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from squeezedet_amd import ops
    return ops


def _rand(shape, seed, scale=1.0, relu=False):
    a = np.random.RandomState(seed).randn(*shape).astype(np.float32) * scale
    return torch.from_numpy(np.maximum(a, 0) if relu else a).to(DEV)


def test_every_custom_op_matches_the_ctypes_path():
    ops = _ops()
    S = torch.ops.sqdet
    mc = O.kitti_squeezeDet_config()
    x = _rand((2, 24, 40, 16), 1, relu=True)
    w, b = _rand((3, 3, 16, 32), 2, 0.1), _rand((32,), 3, 0.1)
    want = ops.conv2d_nhwc(x, ops.pack_conv_weights(w, torch.float32), b, 1, "SAME", True)
    assert torch.equal(S.conv2d(x, w, b, 1, True, True), want)
    pk = ops.pack_conv_weights(w, torch.float32)
    assert torch.equal(S.conv2d_nhwc(x, pk.data, b, 3, 32, 1, True, True), want)
    assert torch.equal(S.maxpool(x, 3, 2, True), ops.maxpool_nhwc(x, 3, 2, "SAME"))
    assert torch.equal(S.maxpool_nhwc(x, 3, 2, False), ops.maxpool_nhwc(x, 3, 2, "VALID"))
    ws, bs = _rand((1, 1, 16, 8), 4, 0.2), _rand((8,), 5, 0.1)
    w1, b1, w3, b3 = _rand((1, 1, 8, 64), 6, 0.2), _rand((64,), 7, 0.1), _rand((3, 3, 8, 64), 8, 0.1), _rand((64,), 9, 0.1)
    y, sq = S.fire(x, ws, bs, w1, b1, w3, b3)
    p = lambda t: ops.pack_conv_weights(t, torch.float32)
    sq_w = ops.conv2d_nhwc(x, p(ws), bs, 1, "SAME", True)
    y_w = torch.cat([ops.conv2d_nhwc(sq_w, p(w1), b1, 1, "SAME", True), ops.conv2d_nhwc(sq_w, p(w3), b3, 1, "SAME", True)], 3)
    assert torch.equal(sq, sq_w) and torch.equal(y, y_w)
    preds = _rand((2, 24, 78, 72), 10, 1.5)
    anchors = torch.from_numpy(mc.ANCHOR_BOX.astype(np.float32)).to(DEV)
    got = S.interpret_output(preds, anchors, 3, 9, float(mc.IMAGE_WIDTH), float(mc.IMAGE_HEIGHT), float(mc.EXP_THRESH))
    ref = ops.interpret_output(preds, anchors, 3, 9, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH)[:3]
    assert all(torch.equal(a, b_) for a, b_ in zip(got, ref))
    f1 = S.filter_prediction(got[0], got[1], got[2], 3, mc.TOP_N_DETECTION, float(mc.NMS_THRESH), float(mc.PROB_THRESH), 64)
    f2 = ops.filter_prediction(ref[0], ref[1], ref[2], 3, mc.TOP_N_DETECTION, mc.NMS_THRESH, mc.PROB_THRESH, 64)
    f3 = S.detect_filter(preds, anchors, 3, 9, float(mc.IMAGE_WIDTH), float(mc.IMAGE_HEIGHT), float(mc.EXP_THRESH), mc.TOP_N_DETECTION,
                         float(mc.NMS_THRESH))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b_) for a, b_ in zip(f1, f2)) and all(torch.equal(a, b_) for a, b_ in zip(f3, f2))


def test_net_forward_op():
    import squeezedet_amd as S_
    from squeezedet_amd import nets, torch_ops
    mc = S_.kitti_squeezeDet_config_for_input(128, 256)
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = 2
    m = nets.SqueezeDet(mc, gpu_id="0", dtype=torch.float16)
    m.load_params(O.init_params("squeezeDet", seed=1, storage="fp16"))
    plan = m._native_plan(2)
    pid = torch_ops.register_plan(plan)
    x = O.synthetic_images(2, 128, 256, seed=2, storage="fp16").to(DEV, torch.float16)
    assert torch.equal(torch.ops.sqdet.net_forward(x, pid), plan.forward(x))
    torch.library.opcheck(torch.ops.sqdet.net_forward, (x, pid), test_utils=("test_schema", "test_faketensor"))


def test_opcheck_differentiable_ops():
    """schema, fake tensors, autograd registration and aot dispatch (the backward passes are custom ops too)."""
    x = _rand((1, 9, 14, 16), 11, relu=True).requires_grad_(True)
    w, b = _rand((3, 3, 16, 32), 12, 0.1).requires_grad_(True), _rand((32,), 13, 0.1).requires_grad_(True)
    torch.library.opcheck(torch.ops.sqdet.conv2d, (x, w, b, 1, True, True))
    torch.library.opcheck(torch.ops.sqdet.maxpool, (x, 3, 2, True))
    ws, bs = _rand((1, 1, 16, 8), 14, 0.2).requires_grad_(True), _rand((8,), 15, 0.1).requires_grad_(True)
    w1, b1 = _rand((1, 1, 8, 64), 16, 0.2).requires_grad_(True), _rand((64,), 17, 0.1).requires_grad_(True)
    w3, b3 = _rand((3, 3, 8, 64), 18, 0.1).requires_grad_(True), _rand((64,), 19, 0.1).requires_grad_(True)
    torch.library.opcheck(torch.ops.sqdet.fire, (x, ws, bs, w1, b1, w3, b3))
    preds = _rand((1, 6, 9, 72), 20, 1.5)
    anchors = torch.abs(_rand((6 * 9 * 9, 4), 21, 40.0)) + 10
    for op, args in ((torch.ops.sqdet.interpret_output, (preds, anchors, 3, 9, 1248.0, 384.0, 1.0)),
                     (torch.ops.sqdet.detect_filter, (preds, anchors, 3, 9, 1248.0, 384.0, 1.0, 64, 0.4))):
        torch.library.opcheck(op, args, test_utils=("test_schema", "test_faketensor"))


def test_backward_through_the_ops_gives_the_oracles_gradients():
    """SqueezeDet's forward graph written with torch.ops.sqdet.conv2d / maxpool / fire (float32, 128 x 256, batch 2), the loss
    gradient from the loss kernel (sqdet_loss_fwd_bwd) pushed through preds.backward(): every trainable variable's gradient
    against oracle/train_oracle.loss_and_grads (autograd on the restated train graph, weight decay added here as the
    reference's 'losses' collection does, nn_skeleton.py:66-69)."""
    ops = _ops()
    S = torch.ops.sqdet
    mc = O.squeezeDet_config_for_input(128, 256)
    mc.IS_TRAINING = False                                       # keep_prob 1.0: no dropout mask to share
    B = 2
    params = O.init_params("squeezeDet", seed=5)
    x = O.synthetic_images(B, 128, 256, seed=6)
    mask, delta, box, labels = TO.synthetic_labels(mc, B, seed=7)
    gh, gw = O.squeezedet_grid(128, 256)
    ref = TO.loss_and_grads("squeezeDet", mc, params, x, torch.ones((B, gh, gw, 768)), mask, delta, box, labels)
    P = {k: v.to(DEV).requires_grad_(not k.startswith("conv1/")) for k, v in params.items()}
    t = x.to(DEV)
    for kind, name, a in O.layer_specs("squeezeDet"):
        if kind == "conv":
            t = S.conv2d(t, P[name + "/kernels"], P[name + "/biases"], a["stride"], a["padding"] == "SAME", a["relu"])
        elif kind == "pool":
            t = S.maxpool(t, a["size"], a["stride"], a["padding"] == "SAME")
        else:
            t = S.fire(t, P[name + "/squeeze1x1/kernels"], P[name + "/squeeze1x1/biases"], P[name + "/expand1x1/kernels"],
                       P[name + "/expand1x1/biases"], P[name + "/expand3x3/kernels"], P[name + "/expand3x3/biases"])[0]
    preds = t
    np.testing.assert_allclose(preds.detach().cpu().numpy(), ref["preds"].numpy(), rtol=1e-3, atol=1e-4 * float(ref["preds"].abs().max()))
    anchors = torch.from_numpy(mc.ANCHOR_BOX.astype(np.float32)).to(DEV)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dpreds, _, losses = ops.loss_fwd_bwd(preds.detach(), anchors, dv(mask.reshape(B, -1)), dv(delta), dv(box), dv(labels), mc, float(mask.sum()))
    preds.backward(dpreds)
    torch.cuda.synchronize()
    for name, g in ref["grads"].items():
        got = P[name].grad.cpu()
        if name.endswith("/kernels"):
            got = got + mc.WEIGHT_DECAY * params[name]
        scale = float(g.abs().max())
        err = float((got - g).abs().max())
        assert err <= 2e-3 * scale + 1e-7, "%s: max err %g vs scale %g" % (name, err, scale)
    assert P["conv1/kernels"].grad is None
