"""CPU oracle of the TRAINING path (TEST INFRASTRUCTURE -- see oracle/__init__.py): the loss graph
(nn_skeleton.py:285-327), the train graph (nn_skeleton.py:329-361) and the dense-label builder
(dataset/imdb.py:195-239 + train.py:205-222), restated with PyTorch-CPU float32 autograd.

Parity status: the label assignment (assign_anchors) is **pinned**: the reference's own imdb.read_batch is imported
unchanged (oracle/ref_imdb_half.py, cv2 stubbed -- the label half never looks at pixels) and run on seeded annotations;
tests/golden/labels.npz holds its anchor indices / deltas and tests/test_oracle_golden.py checks assign_anchors against
them bit for bit.  (Decision points with TIES -- equal IoU between two free anchors, structural for boxes lying inside
or containing several anchors of one shape -- are resolved by np.argsort's unspecified order in the reference, i.e. by
the NumPy build; the golden cases are drawn tie-free, and ours resolve ties towards the higher anchor index.)
The loss / optimizer half is **unpinned** -- that arithmetic lives in tensorflow-gpu==1.0.0 (softmax,
sigmoid, log, MomentumOptimizer, clip_by_norm, exponential_decay), which cannot run here; the
reference ships no training vectors.  Restated from the call sites + TF's documented semantics:
  * tf.nn.l2_loss(v) = sum(v**2)/2, so the weight-decay gradient is wd*v (kernels of trainable
    layers only: nn_skeleton.py:66-69; biases are created without decay, :535);
  * MomentumOptimizer: accum = momentum*accum + grad ; var -= lr*accum;
  * tf.clip_by_norm(g, c) = g*c/max(||g||_2, c), applied PER VARIABLE (nn_skeleton.py:347-349);
  * exponential_decay(staircase=True): lr = LEARNING_RATE * LR_DECAY_FACTOR**floor(step/DECAY_STEPS);
  * self.ious is produced through Variable.assign (nn_skeleton.py:263-268): no gradient flows into it;
  * tf.nn.dropout(x, keep_prob) = x * floor(keep_prob + U) / keep_prob (the mask is an INPUT here).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import sqdet_oracle as O


# --------------------------------------------------------------------------
# dense labels (dataset/imdb.py:195-239, train.py:163-224, utils/util.py:139-158)
# --------------------------------------------------------------------------
def assign_anchors(mc, gt_boxes):
    """imdb.py:195-239 for one image: gt_boxes [n,4] (cx,cy,w,h) -> (anchor idx list, delta list).
    Each GT takes the free anchor of highest IoU; if every overlap is 0, the nearest free anchor."""
    anchor = np.asarray(mc.ANCHOR_BOX)
    aidx_set = set()
    aidxs, deltas = [], []
    for i in range(len(gt_boxes)):
        overlaps = O.batch_iou(anchor, gt_boxes[i])
        aidx = len(anchor)
        for ov_idx in np.argsort(overlaps, kind="stable")[::-1]:
            if overlaps[ov_idx] <= 0:
                break
            if ov_idx not in aidx_set:
                aidx_set.add(ov_idx)
                aidx = ov_idx
                break
        if aidx == len(anchor):
            dist = np.sum(np.square(gt_boxes[i] - anchor), axis=1)
            for dist_idx in np.argsort(dist, kind="stable"):
                if dist_idx not in aidx_set:
                    aidx_set.add(dist_idx)
                    aidx = dist_idx
                    break
        cx, cy, w, h = gt_boxes[i]
        deltas.append([(cx - anchor[aidx][0]) / anchor[aidx][2], (cy - anchor[aidx][1]) / anchor[aidx][3],
                       np.log(w / anchor[aidx][2]), np.log(h / anchor[aidx][3])])
        aidxs.append(int(aidx))
    return aidxs, deltas


def synthetic_labels(mc, batch, seed=0):
    """per image n~U{1..8} GT boxes, w in [20,300], h in [20,200], centre
    uniform in the image, class U{0..C-1}; dense tensors per train.py:205-222 (duplicates of the
    same (image, anchor) are discarded, train.py:178-190)."""
    rs = np.random.RandomState(seed)
    A, C = mc.ANCHORS, mc.CLASSES
    mask = np.zeros((batch, A, 1), np.float32)
    delta = np.zeros((batch, A, 4), np.float32)
    box = np.zeros((batch, A, 4), np.float32)
    labels = np.zeros((batch, A, C), np.float32)
    for b in range(batch):
        n = rs.randint(1, 9)
        gt = np.stack([rs.uniform(0, mc.IMAGE_WIDTH, n), rs.uniform(0, mc.IMAGE_HEIGHT, n),
                       rs.uniform(20, min(300, mc.IMAGE_WIDTH), n), rs.uniform(20, min(200, mc.IMAGE_HEIGHT), n)], 1)
        cls = rs.randint(0, C, n)
        aidxs, deltas = assign_anchors(mc, gt)
        for j, a in enumerate(aidxs):
            if mask[b, a, 0] == 0:
                mask[b, a, 0] = 1.0
                delta[b, a] = deltas[j]
                box[b, a] = gt[j]
                labels[b, a, cls[j]] = 1.0
    return mask, delta, box, labels


# --------------------------------------------------------------------------
# differentiable forward graph (training mode)
# --------------------------------------------------------------------------
def _q(t, storage):
    """storage == "fp16": the value rounded to float16 (what a mixed-precision kernel stores), with a straight-through
    gradient -- so ReLU / max-pool decisions are made on the float16 values the device sees while the weight
    gradients stay float32, as in the device's mixed-precision step."""
    if storage != "fp16":
        return t
    return t + (t.to(torch.float16).to(torch.float32) - t).detach()


def _conv(x, w_hwio, b, stride, padding, relu, storage="fp32"):
    w_hwio = _q(w_hwio, "fp16" if storage == "fp16w" else storage)   # "fp16w": float16 kernel, float32 result
    k = w_hwio.shape[0]
    H, W = x.shape[1], x.shape[2]
    xn = x.permute(0, 3, 1, 2)
    if padding == "SAME":
        pt, pb = O.same_pads(H, k, stride)
        pl, pr = O.same_pads(W, k, stride)
        xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, w_hwio.permute(3, 2, 0, 1), None, stride=stride) + b.view(1, -1, 1, 1)
    if relu:
        y = torch.relu(y)
    return _q(y.permute(0, 2, 3, 1), storage)


def forward_train(arch, params, x, dropout_mask, keep_prob=0.5, storage="fp32", override=None):
    """_add_forward_graph with IS_TRAINING=True: dropout (nets/squeezeDet.py:74) active.  storage="fp16" restates
    the device's mixed-precision step: input, kernels and every stored activation rounded to float16 (float32
    accumulation and biases), see _q.  override: {name: tensor} -- the forward VALUE of that stored activation is
    replaced by the given tensor (straight-through, like _q), e.g. by the activations a device run kept: every ReLU /
    max-pool decision of the backward pass is then made on exactly the device's values, which isolates the backward
    arithmetic from the 1-ulp forward differences that float16 storage amplifies from layer to layer.  Names: conv /
    pool layer names, "<fire>/squeeze1x1", "<fire>" (the concat), "drop" (conv12's input)."""
    def ov(name, t):
        if override is not None and name in override:
            v = override[name].to(torch.float32)
            assert v.shape == t.shape, name
            return t + (v - t).detach()
        return t
    t = _q(x, storage)
    specs = O.layer_specs(arch)
    for kind, name, a in specs:
        if name == "conv12":
            t = ov("drop", _q(t * dropout_mask / keep_prob, storage))
        if kind == "conv":
            t = ov(name, _conv(t, params[name + "/kernels"], params[name + "/biases"], a["stride"], a["padding"], a["relu"], storage))
        elif kind == "pool":
            t = ov(name, O.pooling_layer(t, a["size"], a["stride"], a["padding"]))
        else:
            sq = ov(name + "/squeeze1x1", _conv(t, params[name + "/squeeze1x1/kernels"], params[name + "/squeeze1x1/biases"], 1, "SAME", True, storage))
            e1 = _conv(sq, params[name + "/expand1x1/kernels"], params[name + "/expand1x1/biases"], 1, "SAME", True, storage)
            e3 = _conv(sq, params[name + "/expand3x3/kernels"], params[name + "/expand3x3/biases"], 1, "SAME", True, storage)
            t = ov(name, torch.cat([e1, e3], dim=3))
    return t


def loss_graph(mc, preds, input_mask, box_delta_input, box_input, labels, num_objects=None, global_batch=None):
    """_add_interpretation_graph + _add_loss_graph (nn_skeleton.py:142-327) on a preds tensor
    [B,gh,gw,K*(C+5)] (torch, may require grad).  Returns dict of scalar losses (without weight
    decay) and the detached ious.
    num_objects / global_batch (not in the reference, which is single-device): evaluate these B samples as a SHARE of one
    graph of `global_batch` samples holding `num_objects` objects -- the shares of a partition of the batch then sum to
    exactly the reference's losses at the full batch (:180 num_objects is a batch total; :304-312 reduce_mean divides
    the confidence term by the batch)."""
    B = preds.shape[0]
    K, C, A = mc.ANCHOR_PER_GRID, mc.CLASSES, mc.ANCHORS
    eps = mc.EPSILON
    ncp = K * C
    pcp = torch.softmax(preds[..., :ncp].reshape(-1, C), dim=1).reshape(B, A, C)
    conf = torch.sigmoid(preds[..., ncp:ncp + K].reshape(B, A))
    delta = preds[..., ncp + K:].reshape(B, A, 4)
    mask = torch.as_tensor(input_mask, dtype=torch.float32)
    dl_in = torch.as_tensor(box_delta_input, dtype=torch.float32)
    bx_in = torch.as_tensor(box_input, dtype=torch.float32)
    lab = torch.as_tensor(labels, dtype=torch.float32)
    num_objects = mask.sum() if num_objects is None else torch.as_tensor(num_objects, dtype=torch.float32)
    # ious (nn_skeleton.py:240-269), no gradient (Variable.assign)
    with torch.no_grad():
        out = O.interpret_output(preds.detach().numpy(), mc)
        db = torch.from_numpy(out["det_boxes"])
        b1 = [db[..., 0] - db[..., 2] / 2, db[..., 1] - db[..., 3] / 2, db[..., 0] + db[..., 2] / 2, db[..., 1] + db[..., 3] / 2]
        b2 = [bx_in[..., 0] - bx_in[..., 2] / 2, bx_in[..., 1] - bx_in[..., 3] / 2,
              bx_in[..., 0] + bx_in[..., 2] / 2, bx_in[..., 1] + bx_in[..., 3] / 2]
        w = torch.clamp(torch.minimum(b1[2], b2[2]) - torch.maximum(b1[0], b2[0]), min=0.0)
        h = torch.clamp(torch.minimum(b1[3], b2[3]) - torch.maximum(b1[1], b2[1]), min=0.0)
        inter = w * h
        union = (b1[2] - b1[0]) * (b1[3] - b1[1]) + (b2[2] - b2[0]) * (b2[3] - b2[1]) - inter
        ious = inter / (union + eps) * mask.reshape(B, A)
    # :292-299
    class_loss = ((lab * (-torch.log(pcp + eps)) + (1 - lab) * (-torch.log(1 - pcp + eps))) * mask * mc.LOSS_COEF_CLASS).sum() / num_objects
    # :304-312
    m2 = mask.reshape(B, A)
    conf_loss = (((ious - conf) ** 2) * (m2 * mc.LOSS_COEF_CONF_POS / num_objects
                                          + (1 - m2) * mc.LOSS_COEF_CONF_NEG / (A - num_objects))).sum(dim=1).sum() / float(global_batch or B)
    # :317-323
    bbox_loss = (mc.LOSS_COEF_BBOX * (mask * (delta - dl_in)) ** 2).sum() / num_objects
    return dict(class_loss=class_loss, conf_loss=conf_loss, bbox_loss=bbox_loss, ious=ious, num_objects=num_objects,
                pred_class_probs=pcp, pred_conf=conf)


def trainable_names(arch, params):
    """conv1 is frozen (nets/squeezeDet.py:40-42); everything else trains."""
    return [n for n in params if not n.startswith("conv1/")]


def loss_and_grads(arch, mc, params, x, dropout_mask, input_mask, box_delta_input, box_input, labels, storage="fp32",
                   override=None):
    """Total loss (incl. weight decay) and its gradient w.r.t. every trainable variable."""
    p = {k: v.clone().requires_grad_(k in trainable_names(arch, params)) for k, v in params.items()}
    preds = forward_train(arch, p, x, dropout_mask, 0.5 if mc.IS_TRAINING else 1.0, storage, override)
    parts = loss_graph(mc, preds, input_mask, box_delta_input, box_input, labels)
    wd = sum(mc.WEIGHT_DECAY * (p[k] ** 2).sum() / 2 for k in trainable_names(arch, params) if k.endswith("/kernels"))
    loss = parts["class_loss"] + parts["conf_loss"] + parts["bbox_loss"] + wd
    names = trainable_names(arch, params)
    grads = torch.autograd.grad(loss, [p[k] for k in names], retain_graph=True)
    dpreds = torch.autograd.grad(parts["class_loss"] + parts["conf_loss"] + parts["bbox_loss"], preds)[0]
    return dict(loss=float(loss.detach()), class_loss=float(parts["class_loss"].detach()),
                conf_loss=float(parts["conf_loss"].detach()), bbox_loss=float(parts["bbox_loss"].detach()), grads=dict(zip(names, [g.detach() for g in grads])),
                preds=preds.detach(), dpreds=dpreds.detach(), ious=parts["ious"])


def learning_rate(mc, step):
    return mc.LEARNING_RATE * mc.LR_DECAY_FACTOR ** (step // mc.DECAY_STEPS)


def apply_gradients(mc, params, momenta, grads, step):
    """_add_train_graph (nn_skeleton.py:329-361): per-variable clip_by_norm, then Momentum."""
    lr = learning_rate(mc, step)
    new_p, new_m = dict(params), dict(momenta)
    for k, g in grads.items():
        norm = float(torch.sqrt((g.double() ** 2).sum()))
        gc = g * (mc.MAX_GRAD_NORM / max(norm, mc.MAX_GRAD_NORM))
        acc = mc.MOMENTUM * momenta[k] + gc
        new_m[k] = acc
        new_p[k] = params[k] - lr * acc
    return new_p, new_m
