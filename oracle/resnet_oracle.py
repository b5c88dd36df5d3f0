"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the ResNet50+ConvDet forward
graph of the reference -- src/nets/resnet50_convDet.py:31-169 on top of
ModelSkeleton._conv_bn_layer (src/nn_skeleton.py:374-468).  The reference's TF-0.x graph cannot be
executed here (no TensorFlow), so this half is "parity unpinned" like the rest of the TF half; it is
anchored on the TF op semantics of the call sites cited below and on an independent float64 check
(tests/test_oracle_resnet.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch

from . import sqdet_oracle as so

# (stage scope, block names, branch2a/2b filters, output filters) -- resnet50_convDet.py:47-118
STAGES = [("conv2_x", ["2a", "2b", "2c"], 64, 256),
          ("conv3_x", ["3a", "3b", "3c", "3d"], 128, 512),
          ("conv4_x", ["4a", "4b", "4c", "4d", "4e", "4f"], 256, 1024)]
BN_EPS = 1e-5   # config/config.py:131


def conv_bn_specs():
    """[(scoped variable name, cin, cout, k, stride, relu, with_bias)] in variable-creation order, plus
    block wiring: resnet50_convDet.py:41-118 and _res_branch :134-169 (stride 2 sits on branch1 and
    branch2a of res3a / res4a)."""
    specs = [("conv1", 3, 64, 7, 2, True, True)]
    c = 64
    for scope, blocks, in_f, out_f in STAGES:
        for i, n in enumerate(blocks):
            blk = "%s/res%s/" % (scope, n)
            stride = 2 if (i == 0 and scope != "conv2_x") else 1
            if i == 0:
                specs.append((blk + "res%s_branch1" % n, c, out_f, 1, stride, False, False))
            b2 = blk + "res%s_branch2/res%s" % (n, n)
            specs.append((b2 + "_branch2a", c, in_f, 1, stride, True, False))
            specs.append((b2 + "_branch2b", in_f, in_f, 3, 1, True, False))
            specs.append((b2 + "_branch2c", in_f, out_f, 1, 1, False, False))
            c = out_f
    return specs


def param_shapes(num_output=72):
    shapes = {}
    for name, cin, cout, k, stride, relu, with_bias in conv_bn_specs():
        shapes[name + "/kernels"] = (k, k, cin, cout)
        if with_bias:
            shapes[name + "/biases"] = (cout,)
        for v in ("gamma", "beta", "mean", "var"):
            shapes[name + "/" + v] = (cout,)
    shapes["conv5/kernels"] = (3, 3, 1024, num_output)
    shapes["conv5/biases"] = (num_output,)
    return shapes


def init_params(seed=0, storage="fp32", num_output=72):
    """Seeded synthetic parameters: He-scaled truncated-normal kernels (conv1 scaled for pixel-range
    inputs, as in sqdet_oracle.init_params) and NON-trivial BN statistics so the fold is exercised:
    gamma ~ U(0.5, 1.5) (x0.3 on branch2c so the residual stream stays O(1) over 13 blocks),
    beta ~ U(-0.2, 0.2), mean ~ U(-0.2, 0.2), var ~ U(0.5, 1.5)."""
    rng = np.random.RandomState(seed)
    params = {}
    for name, shp in param_shapes(num_output).items():
        layer, var = name.rsplit("/", 1)
        if var == "kernels":
            fan_in = shp[0] * shp[1] * shp[2]
            sigma = math.sqrt(2.0 / fan_in)
            if layer == "conv1":
                sigma /= 64.0
            if layer == "conv5":
                sigma *= 2.0
            z = rng.standard_normal(size=shp)
            bad = np.abs(z) > 2.0
            while bad.any():
                z[bad] = rng.standard_normal(size=int(bad.sum()))
                bad = np.abs(z) > 2.0
            v = z * sigma
        elif var == "gamma":
            v = rng.uniform(0.5, 1.5, size=shp) * (0.3 if layer.endswith("_branch2c") else 1.0)
        elif var == "var":
            v = rng.uniform(0.5, 1.5, size=shp)
        elif var in ("beta", "mean"):
            v = rng.uniform(-0.2, 0.2, size=shp)
        else:   # biases
            v = rng.uniform(-0.1, 0.1, size=shp)
        params[name] = torch.from_numpy(v.astype(np.float32))
    return params


def fold_batchnorm(w, conv_bias, gamma, beta, mean, var, eps=BN_EPS):
    """tf.nn.batch_normalization (nn_skeleton.py:446-449): inv = rsqrt(var + eps) * gamma;
    y = x * inv + (beta - mean * inv), x = conv [+ biases] -> the same conv with kernel W * inv and bias
    (biases - mean) * inv + beta.  float32."""
    inv = gamma / torch.sqrt(var + eps)
    cb = conv_bias if conv_bias is not None else torch.zeros_like(mean)
    return w * inv.view(1, 1, 1, -1), (cb - mean) * inv + beta


def conv_bn_layer(x, params, name, stride, relu, with_bias, storage="fp32", folded=None, round_out=True):
    """ModelSkeleton._conv_bn_layer, nn_skeleton.py:374-468: conv2d(SAME) [+ bias_add] ->
    batch_normalization(frozen mean/var) -> [relu].
    storage='fp32': op for op as the reference (conv, bias, x*inv + (beta - mean*inv)).
    storage='fp16': the MI355X fp16 storage model -- the FOLDED kernel is what is rounded to fp16
    (operands fp16, fp32 accumulate, fp32 folded bias, output rounded to fp16)."""
    P = params
    w, g, b, m, v = (P[name + "/" + s] for s in ("kernels", "gamma", "beta", "mean", "var"))
    cb = P[name + "/biases"] if with_bias else None
    if storage == "fp16" or folded:
        wf, bf = fold_batchnorm(w, cb, g, b, m, v)
        wf = so._round_storage(wf, storage)
        return so.conv_layer(x, wf, bf, stride, "SAME", relu, storage if round_out else "fp32")
    y = so.conv_layer(x, w, cb if cb is not None else torch.zeros_like(m), stride, "SAME", False, "fp32")
    inv = torch.rsqrt(v + BN_EPS) * g
    y = y * inv + (b - m * inv)
    return torch.relu(y) if relu else y


def forward(params, x, storage="fp32", collect=None, folded=None):
    """ResNet50ConvDet._add_forward_graph, resnet50_convDet.py:31-132.  x: [N,H,W,3] float32 (BGR,
    mean-subtracted) -> preds [N,gh,gw,num_output].  In fp16 storage the residual sum is formed from
    the fp32 branch2c result and the fp16-stored shortcut, then rounded once (the fused epilogue)."""
    def note(name, t):
        if collect is not None:
            collect[name] = t
        return t

    t = note("conv1", conv_bn_layer(x, params, "conv1", 2, True, True, storage, folded))
    t = note("pool1", so.pooling_layer(t, 3, 2, "VALID"))
    for scope, blocks, in_f, out_f in STAGES:
        for i, n in enumerate(blocks):
            blk = "%s/res%s/" % (scope, n)
            stride = 2 if (i == 0 and scope != "conv2_x") else 1
            shortcut = t
            if i == 0:
                shortcut = conv_bn_layer(t, params, blk + "res%s_branch1" % n, stride, False, False, storage, folded)
            b2 = blk + "res%s_branch2/res%s" % (n, n)
            u = conv_bn_layer(t, params, b2 + "_branch2a", stride, True, False, storage, folded)
            u = conv_bn_layer(u, params, b2 + "_branch2b", 1, True, False, storage, folded)
            u = conv_bn_layer(u, params, b2 + "_branch2c", 1, False, False, storage, folded, round_out=False)
            t = note("res" + n, so._round_storage(torch.relu(shortcut + u), storage))   # :55
    # drop4: keep_prob = 1.0 at inference (nn_skeleton.py:78)
    w5 = so._round_storage(params["conv5/kernels"], storage)
    return note("conv5", so.conv_layer(t, w5, params["conv5/biases"], 1, "SAME", False, storage))


# --------------------------------------------------------------------------
# training graph (differentiable, float32, BN UNFOLDED): resnet50_convDet.py:31-132 with IS_TRAINING,
# loss / optimizer from oracle/train_oracle.py (nn_skeleton.py:285-361)
# --------------------------------------------------------------------------
def trainable_names(params):
    """conv1..res3d frozen (resnet50_convDet.py:41-92, freeze=True); res4* kernels/gamma/beta and conv5 train;
    mean / var are never trainable (nn_skeleton.py:437-438)."""
    out = []
    for n in params:
        leaf = n.rsplit("/", 1)[1]
        if n.startswith("conv5/") or (n.startswith("conv4_x/") and leaf in ("kernels", "gamma", "beta")):
            out.append(n)
    return out


def forward_train(params, x, dropout_mask, keep_prob=0.5, storage="fp32", override=None):
    """storage="fp16" restates the device's mixed-precision step (train_oracle._q): the batch norm is folded into
    the kernel in float32, the FOLDED kernel and every stored activation are rounded to float16, the folded bias,
    the accumulation and the residual add stay float32 (the add is the conv's epilogue on the device)."""
    from . import train_oracle as TO
    q = lambda t: TO._q(t, storage)

    def ov(name, t):
        """override (train_oracle.forward_train): pin the forward value of a stored activation -- names: the
        _conv_bn_layer's parameter prefix, "res<block>" (a block's output), "drop4", "conv5"."""
        if override is not None and name in override:
            v = override[name].to(torch.float32)
            assert v.shape == t.shape, name
            return t + (v - t).detach()
        return t

    def cbn(t, name, stride, relu, with_bias, rounded=True):
        return ov(name, cbn_(t, name, stride, relu, with_bias, rounded)) if rounded else cbn_(t, name, stride, relu, with_bias, rounded)

    def cbn_(t, name, stride, relu, with_bias, rounded=True):
        P = params
        inv = torch.rsqrt(P[name + "/var"] + BN_EPS) * P[name + "/gamma"]
        if storage == "fp16":
            bias = P[name + "/beta"] - P[name + "/mean"] * inv + (P[name + "/biases"] * inv if with_bias else 0.0)
            y = TO._conv(t, P[name + "/kernels"] * inv.view(1, 1, 1, -1), bias, stride, "SAME", False, storage if rounded else "fp16w")
            y = torch.relu(y) if relu else y
            return q(y) if rounded else y
        y = TO._conv(t, P[name + "/kernels"], P[name + "/biases"] if with_bias else torch.zeros_like(P[name + "/mean"]), stride, "SAME", False)
        y = y * inv + (P[name + "/beta"] - P[name + "/mean"] * inv)
        return torch.relu(y) if relu else y

    t = cbn(q(x), "conv1", 2, True, True)
    t = so.pooling_layer(t, 3, 2, "VALID")
    for scope, blocks, in_f, out_f in STAGES:
        for i, n in enumerate(blocks):
            blk = "%s/res%s/" % (scope, n)
            stride = 2 if (i == 0 and scope != "conv2_x") else 1
            shortcut = cbn(t, blk + "res%s_branch1" % n, stride, False, False) if i == 0 else t
            b2 = blk + "res%s_branch2/res%s" % (n, n)
            u = cbn(t, b2 + "_branch2a", stride, True, False)
            u = cbn(u, b2 + "_branch2b", 1, True, False)
            u = cbn(u, b2 + "_branch2c", 1, False, False, rounded=False)
            t = ov("res" + n, q(torch.relu(shortcut + u)))
    t = ov("drop4", q(t * dropout_mask / keep_prob))       # drop4 (resnet50_convDet.py:126)
    return ov("conv5", TO._conv(t, params["conv5/kernels"], params["conv5/biases"], 1, "SAME", False, storage))


def loss_and_grads(mc, params, x, dropout_mask, input_mask, box_delta_input, box_input, labels, storage="fp32", override=None):
    """Total loss (incl. weight decay on the trainable kernels) and its gradient w.r.t. every trainable variable."""
    from . import train_oracle as TO
    names = trainable_names(params)
    p = {k: v.clone().requires_grad_(k in names) for k, v in params.items()}
    preds = forward_train(p, x, dropout_mask, 0.5, storage, override)
    parts = TO.loss_graph(mc, preds, input_mask, box_delta_input, box_input, labels)
    wd = sum(mc.WEIGHT_DECAY * (p[k] ** 2).sum() / 2 for k in names if k.endswith("/kernels"))
    main = parts["class_loss"] + parts["conf_loss"] + parts["bbox_loss"]
    grads = torch.autograd.grad(main + wd, [p[k] for k in names], retain_graph=True)
    dpreds = torch.autograd.grad(main, preds)[0]
    return dict(class_loss=float(parts["class_loss"].detach()), conf_loss=float(parts["conf_loss"].detach()),
                bbox_loss=float(parts["bbox_loss"].detach()), grads=dict(zip(names, [g.detach() for g in grads])),
                preds=preds.detach(), dpreds=dpreds.detach())


def forward_float64(params, x):
    """Independent check of the float32 restatement: the same graph in float64, BN unfolded."""
    P = {k: v.double() for k, v in params.items()}
    return forward(P, x.double(), "fp32")
