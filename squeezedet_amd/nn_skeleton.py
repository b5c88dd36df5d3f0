"""Host-side mirror of the reference's ModelSkeleton (reference src/nn_skeleton.py): the same
builder methods (`_conv_layer`, `_pooling_layer`, `_fire_layer` in the nets), the same model
attributes (`image_input, preds, pred_class_probs, pred_conf, pred_box_delta, det_boxes,
det_probs, det_class, model_params`) and `filter_prediction`, but every op is a HIP kernel of
libsqdet_hip.so.  Like TF 1.0 it is define-then-run: builders record symbolic nodes, and
`Session.run(fetches, feed_dict)` (the call shape of demo.py:193-195 / eval.py:75-77)
executes them on the GPU.  There is no CPU execution path.
"""
import collections
import contextlib
import os

import numpy as np
import torch

from . import ops
from ._lib import SqdetError


class Node:
    """A symbolic tensor of the model graph (what a tf.Tensor is to the reference)."""

    def __init__(self, model, op, inputs=(), shape=None, name=None, **attrs):
        self.model, self.op, self.inputs, self.shape, self.name, self.attrs = model, op, tuple(inputs), shape, name, attrs
        self.consumers = 0
        self.readers = []
        for i in self.inputs:
            i.consumers += 1
            i.readers.append(self)

    def get_shape(self):
        return tuple(self.shape)

    def __repr__(self):
        return "<Node %s %s %s>" % (self.op, self.name, self.shape)

    # hashable by identity so nodes can key feed_dict like TF tensors do
    __hash__ = object.__hash__


class Session:
    """Stand-in for tf.Session for demo.py / eval.py shaped callers: `run(fetches, feed_dict)`
    returns fresh NumPy arrays owned by the caller (eval.py:83-84 mutates them in place)."""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        single = isinstance(fetches, Node)
        fl = [fetches] if single else list(fetches)
        outs = fl[0].model.run(fl, feed_dict or {}, as_numpy=True)
        return outs[0] if single else outs


def _truncated_normal(shape, stddev, generator, device):
    """tf.truncated_normal_initializer: N(0, stddev) re-drawn outside 2 sigma (nn_skeleton.py:527-528)."""
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=stddev, a=-2 * stddev, b=2 * stddev, generator=generator)
    return t.to(device)


def _out_size(n, k, s, padding):
    return -(-n // s) if padding.upper() == "SAME" else (n - k) // s + 1


class ModelSkeleton:
    """Base class of NN detection models (nn_skeleton.py:72-135)."""

    # set by subclasses whose _add_forward_graph has a native plan (sqdet_net_*)
    NATIVE_ARCH = None

    def __init__(self, mc, gpu_id=0, dtype=torch.float32, seed=0):
        self.mc = mc
        # FLAGS.gpu arrives as a string (demo.py:174).  Without a HIP device the graph can still be
        # BUILT (shapes, parameter table, analytical counters) but nothing can run: no CPU path.
        self.has_device = torch.cuda.is_available()
        self.device = torch.device("cuda", int(gpu_id)) if self.has_device else torch.device("cpu")
        self.dtype = dtype
        # nn_skeleton.py:78
        self.keep_prob = 0.5 if mc.IS_TRAINING else 1.0
        self._gen = torch.Generator().manual_seed(seed)
        self._seed = int(seed)
        # nn_skeleton.py:81-84,121: [BATCH, H, W, 3] float32 BGR mean-subtracted NHWC
        self.ph_image_input = Node(self, "placeholder", shape=(mc.BATCH_SIZE, mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, 3), name="image_input")
        self.image_input = self.ph_image_input
        self.params = collections.OrderedDict()   # '<layer>/kernels' (HWIO f32) / '<layer>/biases'
        self.trainable = {}
        self.model_params = []                    # nn_skeleton.py:128
        self.model_size_counter = []
        self.flop_counter = []
        self.activation_counter = [("input", mc.IMAGE_WIDTH * mc.IMAGE_HEIGHT * 3)]
        self._packed = {}
        self._scope = []                          # tf.variable_scope stack
        # native plans: [0] = the model's plan; [1] = the second serving lane's (detect_filter_pipelined, two batches in flight).
        # A plan is stale when its parameter version is behind the model's (`_plan_stale = True` bumps the model's)
        self._plans, self._plan_ver, self._param_version = {}, {}, 0
        self._plan_stale = True
        self._anchors_f32 = None
        self.caffemodel_weight = None
        # serving lanes of detect_filter_pipelined(defer=True): BATCHES IN FLIGHT (see its docstring: the completion contract of a
        # deferred call depends on it).  None = SQDET_SERVE_LANES from the environment, default 2; the `lanes` argument overrides.
        self.serve_lanes = None
        self._lanes, self._lane_next, self._lanes_checked, self._lane_check = None, 0, False, None
        # measurement hook (bench.py's latency_ms_per_batch): a list -> every detect_filter_pipelined call appends
        # (lane, (event before the call's device work, event behind it)) on the stream the call runs on
        self._latency_probe = None

    # ------------------------------------------------------------------ builders
    def _add_forward_graph(self):
        """NN architecture specification."""
        raise NotImplementedError

    def _new_param(self, name, value, trainable):
        self.params[name] = value.to(self.device, torch.float32).contiguous()
        self.trainable[name] = trainable
        self.model_params.append(self.params[name])

    def _conv_layer(self, layer_name, inputs, filters, size, stride, padding="SAME", freeze=False, xavier=False,
                    relu=True, stddev=0.001):
        """Convolutional layer constructor (nn_skeleton.py:471-563): kernel '<layer>/kernels'
        [size,size,Cin,filters] HWIO, '<layer>/biases' [filters]; conv2d -> bias_add -> relu."""
        mc = self.mc
        channels = int(inputs.get_shape()[3])
        use_pretrained_param = False
        if mc.LOAD_PRETRAINED_MODEL:
            cw = self.caffemodel_weight
            if layer_name in cw:
                kernel_val = np.transpose(cw[layer_name][0], [2, 3, 1, 0])  # OIHW -> HWIO (:496)
                bias_val = cw[layer_name][1]
                if kernel_val.shape == (size, size, channels, filters) and bias_val.shape == (filters,):
                    use_pretrained_param = True
                else:
                    print("Shape of the pretrained parameter of {} does not match, "
                          "use randomly initialized parameter".format(layer_name))
            else:
                print("Cannot find {} in the pretrained model. Use randomly initialized parameters".format(layer_name))
        if use_pretrained_param:
            kernel = torch.from_numpy(np.ascontiguousarray(kernel_val, dtype=np.float32))
            biases = torch.from_numpy(np.ascontiguousarray(bias_val, dtype=np.float32))
        elif xavier:
            fan_in, fan_out = size * size * channels, size * size * filters
            lim = (6.0 / (fan_in + fan_out)) ** 0.5
            kernel = (torch.rand((size, size, channels, filters), generator=self._gen) * 2 - 1) * lim
            biases = torch.zeros(filters)
        else:
            kernel = _truncated_normal((size, size, channels, filters), stddev, self._gen, "cpu")
            biases = torch.zeros(filters)
        self._new_param(layer_name + "/kernels", kernel, not freeze)
        self._new_param(layer_name + "/biases", biases, not freeze)

        n, h, w, _ = inputs.get_shape()
        out_shape = (n, _out_size(h, size, stride, padding), _out_size(w, size, stride, padding), filters)
        out = Node(self, "conv", [inputs], out_shape, layer_name, size=size, stride=stride, padding=padding, relu=relu)
        # nn_skeleton.py:549-561 analytical counters
        self.model_size_counter.append((layer_name, (1 + size * size * channels) * filters))
        num_flops = (1 + 2 * channels * size * size) * filters * out_shape[1] * out_shape[2]
        if relu:
            num_flops += 2 * filters * out_shape[1] * out_shape[2]
        self.flop_counter.append((layer_name, num_flops))
        self.activation_counter.append((layer_name, out_shape[1] * out_shape[2] * out_shape[3]))
        return out

    @contextlib.contextmanager
    def variable_scope(self, name):
        """tf.variable_scope: prefixes the names of the variables created inside (resnet50_convDet.py:47-49)."""
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def _conv_bn_layer(self, inputs, conv_param_name, bn_param_name, scale_param_name, filters, size, stride,
                       padding="SAME", freeze=False, relu=True, conv_with_bias=False, stddev=0.001):
        """Convolution + BatchNorm + [relu] layer (nn_skeleton.py:374-468).  Batch mean and var are
        constants; variables '<scope>/<conv_param_name>/kernels' [, 'biases'], 'gamma', 'beta', 'mean',
        'var'.  Executed as ONE conv: the frozen BN is folded into kernel and bias
        (sqdet_fold_batchnorm)."""
        mc = self.mc
        channels = int(inputs.get_shape()[3])
        if mc.LOAD_PRETRAINED_MODEL:
            cw = self.caffemodel_weight
            kernel = torch.from_numpy(np.ascontiguousarray(np.transpose(cw[conv_param_name][0], [2, 3, 1, 0]), dtype=np.float32))
            bias = torch.from_numpy(np.ascontiguousarray(cw[conv_param_name][1], dtype=np.float32)) if conv_with_bias else None
            vec = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(-1)
            mean, var = vec(cw[bn_param_name][0]), vec(cw[bn_param_name][1])
            gamma, beta = vec(cw[scale_param_name][0]), vec(cw[scale_param_name][1])
        else:
            kernel = _truncated_normal((size, size, channels, filters), stddev, self._gen, "cpu")
            bias = torch.zeros(filters) if conv_with_bias else None
            mean, var, gamma, beta = torch.zeros(filters), torch.ones(filters), torch.ones(filters), torch.zeros(filters)
        if tuple(kernel.shape) != (size, size, channels, filters):
            raise SqdetError("%s: pretrained kernel shape %s != %s" % (conv_param_name, tuple(kernel.shape), (size, size, channels, filters)))
        name = "/".join(self._scope + [conv_param_name])
        self._new_param(name + "/kernels", kernel, not freeze)
        if conv_with_bias:
            self._new_param(name + "/biases", bias, not freeze)
        self._new_param(name + "/gamma", gamma, not freeze)
        self._new_param(name + "/beta", beta, not freeze)
        self._new_param(name + "/mean", mean, False)
        self._new_param(name + "/var", var, False)
        n, h, w, _ = inputs.get_shape()
        out_shape = (n, _out_size(h, size, stride, padding), _out_size(w, size, stride, padding), filters)
        out = Node(self, "conv_bn", [inputs], out_shape, name, size=size, stride=stride, padding=padding, relu=relu,
                   with_bias=conv_with_bias)
        self.model_size_counter.append((conv_param_name, (1 + size * size * channels) * filters))
        num_flops = (1 + 2 * channels * size * size) * filters * out_shape[1] * out_shape[2]
        if relu:
            num_flops += 2 * filters * out_shape[1] * out_shape[2]
        self.flop_counter.append((conv_param_name, num_flops))
        self.activation_counter.append((conv_param_name, out_shape[1] * out_shape[2] * out_shape[3]))
        return out

    def _add_relu(self, shortcut, branch, name=None):
        """tf.nn.relu(shortcut + branch, 'relu') (resnet50_convDet.py:55)."""
        assert shortcut.get_shape() == branch.get_shape()
        return Node(self, "add_relu", [shortcut, branch], shortcut.get_shape(), name)

    def _pooling_layer(self, layer_name, inputs, size, stride, padding="SAME"):
        """Pooling layer constructor (nn_skeleton.py:565-586)."""
        n, h, w, c = inputs.get_shape()
        out_shape = (n, _out_size(h, size, stride, padding), _out_size(w, size, stride, padding), c)
        out = Node(self, "pool", [inputs], out_shape, layer_name, size=size, stride=stride, padding=padding)
        self.activation_counter.append((layer_name, int(np.prod(out_shape[1:]))))
        return out

    def _concat(self, values, axis, name=None):
        """tf.concat on the channel axis (nets/squeezeDet.py:106)."""
        assert axis == 3
        s = values[0].get_shape()
        return Node(self, "concat", values, (s[0], s[1], s[2], sum(v.get_shape()[3] for v in values)), name)

    def _dropout(self, inputs, keep_prob, name=None):
        """tf.nn.dropout (nets/squeezeDet.py:74); identity at inference (keep_prob == 1.0)."""
        if keep_prob == 1.0:
            return inputs
        # training graph (mc.IS_TRAINING): x * floor(keep_prob + U) / keep_prob, float32 only
        return Node(self, "dropout", [inputs], inputs.get_shape(), name, keep_prob=keep_prob)

    # ------------------------------------------------------------------ interpretation
    def _add_interpretation_graph(self):
        """Interpret NN output (nn_skeleton.py:142-283)."""
        mc = self.mc
        n, gh, gw, ch = self.preds.get_shape()
        assert ch == mc.ANCHOR_PER_GRID * (mc.CLASSES + 1 + 4)
        assert gh * gw * mc.ANCHOR_PER_GRID == mc.ANCHORS, "grid %dx%d does not match mc.ANCHORS" % (gh, gw)
        interp = Node(self, "interpret", [self.preds], None, "interpret_output")
        B, A = mc.BATCH_SIZE, mc.ANCHORS
        mk = lambda i, shape, nm: Node(self, "interpret_out", [interp], shape, nm, index=i)
        self.det_boxes = mk(0, (B, A, 4), "bbox")
        self.det_probs = mk(1, (B, A), "score")
        self.det_class = mk(2, (B, A), "class_idx")
        self.pred_class_probs = mk(3, (B, A, mc.CLASSES), "pred_class_probs")
        self.pred_conf = mk(4, (B, A), "pred_confidence_score")
        self.pred_box_delta = Node(self, "box_delta", [self.preds], (B, A, 4), "bbox_delta")

    def anchors_f32(self):
        """float32(mc.ANCHOR_BOX) on the device -- cast first, then compute (nn_skeleton.py:187-201)."""
        if self._anchors_f32 is None:
            self._anchors_f32 = torch.from_numpy(np.asarray(self.mc.ANCHOR_BOX).astype(np.float32)).to(self.device)
        return self._anchors_f32

    # ------------------------------------------------------------------ parameters
    def load_params(self, values):
        """values: {name: array/tensor} with the reference's names and layouts ('<layer>/kernels'
        HWIO, '<layer>/biases')."""
        for name, v in values.items():
            if name not in self.params:
                raise SqdetError("unknown parameter %r" % name)
            t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(self.device, torch.float32)
            if tuple(t.shape) != tuple(self.params[name].shape):
                raise SqdetError("parameter %r: shape %s != %s" % (name, tuple(t.shape), tuple(self.params[name].shape)))
            self.params[name].copy_(t)
        self._packed.clear()
        self._plan_stale = True

    def _packed_conv(self, name):
        if name not in self._packed:
            self._packed[name] = ops.pack_conv_weights(self.params[name + "/kernels"], self.dtype)
        return self._packed[name]

    def _folded_conv(self, name, with_bias):
        """(PackedConv, folded bias) of a _conv_bn_layer: BN folded by sqdet_fold_batchnorm, then packed."""
        if name not in self._packed:
            P = self.params
            wf, bf = ops.fold_batchnorm(P[name + "/kernels"], P[name + "/biases"] if with_bias else None, P[name + "/gamma"],
                                        P[name + "/beta"], P[name + "/mean"], P[name + "/var"], self.mc.BATCH_NORM_EPSILON)
            self._packed[name] = (ops.pack_conv_weights(wf, self.dtype), bf)
        return self._packed[name]

    # ------------------------------------------------------------------ execution
    @property
    def _plan_stale(self):
        return self._plan_ver.get(0) != self._param_version

    @_plan_stale.setter
    def _plan_stale(self, stale):
        if stale:
            self._param_version += 1            # every plan (both serving lanes) re-reads the variables at its next use
        else:
            self._plan_ver[0] = self._param_version

    @property
    def _plan(self):
        return self._plans.get(0)

    def _native_plan(self, batch, which=0):
        plan = self._plans.get(which)
        if plan is None or plan.batch != batch:
            mc = self.mc
            plan = ops.NetPlan(self.NATIVE_ARCH, self.dtype, batch, mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, mc.CLASSES,
                               mc.ANCHOR_PER_GRID, self.device)
            plan.set_bn_epsilon(mc.BATCH_NORM_EPSILON)
            self._plans[which] = plan
            self._plan_ver[which] = None
        if self._plan_ver.get(which) != self._param_version:
            specs = dict(plan.param_specs())
            if set(specs) != set(self.params):
                raise SqdetError("native plan parameters do not match the python graph")
            for name, t in self.params.items():
                plan.set_param(name, t)
            self._plan_ver[which] = self._param_version
        return plan

    def _to_input(self, value):
        x = value if isinstance(value, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(value, dtype=np.float32)))
        if x.is_cuda and x.dtype != self.dtype and x.dtype in (torch.float32, torch.float16) and x.numel() % 4 == 0:
            x = ops.convert_scale(x.contiguous(), self.dtype)      # (device float32 -> float16: the HIP cast kernel)
        x = x.to(self.device, self.dtype).contiguous()
        mc = self.mc
        if x.dim() != 4 or tuple(x.shape[1:]) != (mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, 3):
            raise SqdetError("image_input must be [B,%d,%d,3] NHWC, got %s" % (mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, tuple(x.shape)))
        return x

    def _eval(self, node, env, use_plan):
        if node in env:
            return env[node]
        mc = self.mc
        if node.op == "placeholder":
            raise SqdetError("placeholder %s was not fed" % node.name)
        if use_plan and node is self.preds and self.NATIVE_ARCH is not None and self.keep_prob == 1.0:
            x = self._eval(self.image_input, env, use_plan)
            v = self._native_plan(int(x.shape[0])).forward(x)
        elif node.op == "conv":
            x = self._eval(node.inputs[0], env, use_plan)
            v = ops.conv2d_nhwc(x, self._packed_conv(node.name), self.params[node.name + "/biases"], node.attrs["stride"],
                                node.attrs["padding"], node.attrs["relu"])
        elif node.op == "conv_bn":
            x = self._eval(node.inputs[0], env, use_plan)
            pk, bf = self._folded_conv(node.name, node.attrs["with_bias"])
            v = ops.conv2d_nhwc(x, pk, bf, node.attrs["stride"], node.attrs["padding"], node.attrs["relu"])
        elif node.op == "add_relu":
            shortcut, branch = node.inputs
            if branch.op == "conv_bn" and not branch.attrs["relu"] and branch.consumers == 1:
                # branch2c adds into the shortcut in its own epilogue: relu(conv + b + shortcut).  The
                # shortcut tensor is updated in place when this add is its last reader (branch1 output,
                # or a block input whose only other reader -- branch2a -- has already run).
                bx = self._eval(branch.inputs[0], env, use_plan)
                sv = self._eval(shortcut, env, use_plan)
                in_place = (shortcut not in self._fetching and shortcut.op != "placeholder" and
                            all(r in env for r in shortcut.readers if r is not node))
                pk, bf = self._folded_conv(branch.name, branch.attrs["with_bias"])
                if in_place:
                    v = ops.conv2d_nhwc(bx, pk, bf, branch.attrs["stride"], branch.attrs["padding"], True, out=sv, accumulate=True)
                else:       # the shortcut has other readers: read as a residual tensor, not copied
                    v = ops.conv2d_nhwc(bx, pk, bf, branch.attrs["stride"], branch.attrs["padding"], True, residual=sv)
            else:
                a = self._eval(shortcut, env, use_plan)
                b = self._eval(branch, env, use_plan)
                v = ops.add_relu(a, b)
        elif node.op == "pool":
            src = node.inputs[0]
            fused = None
            if (node.attrs["size"] == 3 and node.attrs["stride"] == 2 and src.op in ("conv", "conv_bn") and src.consumers == 1 and src not in env
                    and src not in self._fetching and src.attrs["stride"] == 2 and src.attrs["relu"] and int(src.inputs[0].shape[3]) == 3
                    and ops.stem_supported(int(src.shape[3]), src.attrs["size"])):
                # conv1 + pool1 of the node-by-node evaluation (the frozen prefix of the trainers, fetches of inner tensors) as the fused
                # stem launch, like the native plan's (conv activations never reach HBM); shapes no stem kernel takes: conv, then pool
                xin = self._eval(src.inputs[0], env, use_plan)
                if src.op == "conv":
                    pk, bf = self._packed_conv(src.name), self.params[src.name + "/biases"]
                else:
                    pk, bf = self._folded_conv(src.name, src.attrs["with_bias"])
                try:
                    fused = ops.stem_conv_pool(xin, pk, bf, src.attrs["padding"], node.attrs["padding"])
                except SqdetError:
                    fused = None
            if fused is not None:
                v = fused
            else:
                x = self._eval(src, env, use_plan)
                v = ops.maxpool_nhwc(x, node.attrs["size"], node.attrs["stride"], node.attrs["padding"])
        elif node.op == "concat":
            if all(i.op == "conv" and i.consumers == 1 for i in node.inputs):
                # the producing convs write their channel range of the concat tensor directly
                xs = [self._eval(i.inputs[0], env, use_plan) for i in node.inputs]
                b = int(xs[0].shape[0])
                v = torch.empty((b,) + tuple(node.shape[1:]), dtype=self.dtype, device=self.device)
                off = 0
                for i, x in zip(node.inputs, xs):
                    ops.conv2d_nhwc(x, self._packed_conv(i.name), self.params[i.name + "/biases"], i.attrs["stride"],
                                    i.attrs["padding"], i.attrs["relu"], out=v, out_coffset=off)
                    off += i.shape[3]
            else:
                xs = [self._eval(i, env, use_plan) for i in node.inputs]
                v = torch.empty(tuple(xs[0].shape[:3]) + (node.shape[3],), dtype=self.dtype, device=self.device)
                off = 0
                for x in xs:
                    ops.copy_channels(x.contiguous(), v, off)
                    off += int(x.shape[3])
        elif node.op == "dropout":
            x = self._eval(node.inputs[0], env, use_plan)
            kp = node.attrs["keep_prob"]
            self._dropout_calls = getattr(self, "_dropout_calls", 0) + 1
            dmask = ops.dropout_mask(tuple(x.shape), kp, (self._seed << 32) + self._dropout_calls, x.dtype, x.device)
            v = ops.scale_mask(x.contiguous(), dmask, 1.0 / kp)
        elif node.op == "interpret":
            preds = self._eval(node.inputs[0], env, use_plan)
            v = ops.interpret_output(preds, self.anchors_f32(), mc.CLASSES, mc.ANCHOR_PER_GRID, mc.IMAGE_WIDTH,
                                     mc.IMAGE_HEIGHT, mc.EXP_THRESH, with_class_probs=True)
        elif node.op == "interpret_out":
            v = self._eval(node.inputs[0], env, use_plan)[node.attrs["index"]]
        elif node.op == "box_delta":
            preds = self._eval(node.inputs[0], env, use_plan)
            k = mc.ANCHOR_PER_GRID * (mc.CLASSES + 1)
            v = preds[..., k:].reshape(preds.shape[0], -1, 4)
        else:
            raise SqdetError("unknown op %s" % node.op)
        env[node] = v
        return v

    def run(self, fetches, feed_dict, as_numpy=False, use_plan=True):
        """Evaluates graph nodes.  feed_dict: {model.image_input: array or device tensor}."""
        if not self.has_device:
            raise SqdetError("squeezedet_amd needs a HIP device to run: there is no CPU path")
        with torch.cuda.device(self.device):     # kernels go to the stream of the MODEL's device (gpu_id), whatever is current
            return self._run(fetches, feed_dict, as_numpy, use_plan)

    def _run(self, fetches, feed_dict, as_numpy, use_plan):
        env = {}
        self._fetching = set(fetches)
        for k, v in feed_dict.items():
            if k is not self.image_input and k is not self.ph_image_input:
                raise SqdetError("only image_input can be fed")
            env[self.image_input] = self._to_input(v)
        outs = [self._eval(f, env, use_plan) for f in fetches]
        if as_numpy:
            torch.cuda.current_stream().synchronize()
            outs = [o.float().cpu().numpy() if o.dtype == torch.float16 else o.cpu().numpy() for o in outs]
        return outs

    def detect(self, images, use_plan=True):
        """Device-resident hot path: images [B,H,W,3] -> (det_boxes [B,A,4] f32, det_probs [B,A] f32,
        det_class [B,A] i64) device tensors (nothing is copied to the host)."""
        return tuple(self.run([self.det_boxes, self.det_probs, self.det_class], {self.image_input: images},
                              use_plan=use_plan))

    def detect_filter_pipelined(self, images, to_host=False, defer=False, lanes=None):
        """One step of the serving loop as a two-stage pipeline: the network forward runs on the caller's stream,
        interpret_output + filter_prediction (a few dozen microseconds of latency-bound work on 32 workgroups)
        run on a side HIP stream behind an event, so the NEXT batch's forward starts while this batch's boxes
        are being decoded and suppressed.  Returns filter_prediction_batch's tuple; the tensors are complete once
        flush_pipeline() has been called and the caller's stream (or the device) is synchronised.

        Models with a native plan run it on TWO static sets of buffers (preds, det_*, outputs) used alternately, with
        explicit events in both directions -- no allocation per step.  (Per-step torch allocations were the first
        version: preds had to be record_stream'ed for the side stream, so the caching allocator could not reuse a
        block until its event had completed; a host running a hundred steps ahead then asked for a hundred preds
        buffers, i.e. hipMalloc inside the serving loop -- the same binary measured 0.77 or 1.0-1.2 ms per step from
        one run to the next.)  The returned tensors are those of the slot: valid until the second-next call of the lane.
        to_host=True: the filtered rows (<= TOP_N per image: boxes, probs, classes, anchor indices, counts) are also copied
        to the slot's PINNED host buffers on the side stream -- what sess.run + filter_prediction hand the reference's
        caller -- and those host tensors are returned.

        defer=True (plans with the score epilogue and fire_chain launches: float16 SqueezeDet): the decode + filter of this
        call is carried out BY THE NEXT CALL's forward (or by flush_pipeline()): it is handed to the plan as a post job
        (sqdet_net_set_post_job) and runs in rider workgroups of that forward's fire_chain launches, one image per
        otherwise idle CU, writing the rows straight into the slot's pinned host buffer -- no side stream, no events, no
        extra launch.  Every launch of the forward fills the chip exactly once (persistent kernels with a static share of
        tiles per workgroup), so side work on another stream costs a whole round of whatever it lands beside, and every
        event ordering the two streams drains the forward's queue: measured 35 us per 0.49 ms step wherever the filter
        launch was placed (side stream, same stream, with or without the score kernel, beside the stem or beside the
        fire_chain launches) -- whereas the six fire_chain launches occupy 240 of the 256 CUs at batch 32.
        (SQDET_POST_DEFER=signal: the previous form -- the side stream's launch gated on a mid-forward event.)

        lanes (deferred calls on native plans; None = the model attribute `serve_lanes`, whose default is 2, or the environment's
        SQDET_SERVE_LANES): the number of BATCHES IN FLIGHT.  Consecutive calls alternate between `lanes` serving lanes -- each its
        own plan (workspace), HIP stream and pipeline slots, nothing ordering the lanes against each other -- so one lane's launch
        ramps and tails are filled by the other lanes' launches (throughput +20 % at batch 32 with two; three pay at batch 1).
        THE COMPLETION CONTRACT DEPENDS ON IT: the rows a deferred call returns are complete after the next call OF ITS LANE, i.e.
        after `lanes` further calls -- or after flush_pipeline() -- plus a synchronisation of the caller's stream behind that
        call; lanes=1 is the single-stream behaviour (complete after the NEXT call).  The result latency of a steady serving
        loop is therefore `lanes` step times (bench.py reports it as latency_ms_per_batch)."""
        with torch.cuda.device(self.device):
            lane_set = self._serving_lanes(defer, lanes)
            probe = self._latency_probe
            if lane_set is None:
                if probe is None:
                    return self._detect_filter_pipelined(images, to_host, defer)
                evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                evs[0].record()
                out = self._detect_filter_pipelined(images, to_host, defer)
                evs[1].record()
                probe.append((0, evs))
                return out
            # A lane starts behind the caller's stream (the input may have been produced there).
            if not self._lanes_checked:
                self.warm_up_lanes(images, lanes=len(lane_set))
            lane = lane_set[self._lane_next % len(lane_set)]
            self._lane_next = (self._lane_next + 1) % len(lane_set)
            cur = torch.cuda.current_stream()
            lane["in_ev"].record(cur)
            with torch.cuda.stream(lane["stream"]), self._lane_state(lane):
                lane["stream"].wait_event(lane["in_ev"])
                if probe is not None:
                    evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    evs[0].record()
                out = self._detect_filter_pipelined(images, to_host, defer)
                if probe is not None:
                    evs[1].record()
                    probe.append((lane["which"], evs))
                if isinstance(images, torch.Tensor) and images.is_cuda:
                    images.record_stream(lane["stream"])
            return out

    def warm_up_lanes(self, images, lanes=None):
        """Builds the serving lanes' plans for this batch and makes sure their HIP streams really run CONCURRENTLY; called by the
        first deferred detect_filter_pipelined of a lane set (a caller that must not pay ~20-40 ms inside its first serving call,
        or that captures streams, calls it ahead of time).  Which hardware queue a HIP stream lands on is the runtime's business,
        and two streams that share one serialise -- measured on this stack: of ten streams of torch's pool, the pairs containing
        one particular stream gave 0.468 ms per forward (= one stream) where every other pair gave 0.371.  For every lane k >= 1:
        16 forwards alternating between lane 0's stream and lane k's are timed with HIP events against 16 on lane 0's stream
        alone (median of three repetitions each, plans built and warmed on the lane streams first); a pair that gains less than
        6 % has lane k's stream replaced (up to four candidates, the best kept).  Result: self._lane_check."""
        with torch.cuda.device(self.device):
            lane_set = self._serving_lanes(True, lanes)
            self._lanes_checked = True
            if lane_set is None or os.environ.get("SQDET_LANE_CHECK", "1") == "0":
                return None
            x = self._to_input(images)
            B = int(x.shape[0])
            cur = torch.cuda.current_stream()
            plans, pre = [], []
            for lane in lane_set:
                lane["stream"].wait_stream(cur)
                with torch.cuda.stream(lane["stream"]):
                    plans.append(self._native_plan(B, lane["which"]))
                    pre.append(torch.empty((B, plans[0].gh, plans[0].gw, plans[0].out_ch), dtype=self.dtype, device=self.device))
            NF, REPS = 16, 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(k, sb):
                """ms per forward: NF forwards alternating between lane 0 (plan 0, its stream) and lane k's plan on stream sb"""
                sa = lane_set[0]["stream"]
                ts = []
                for rep in range(REPS + 1):               # (first repetition: warm-up of streams / plans)
                    torch.cuda.synchronize(self.device)
                    e0.record(sa)
                    if sb is not sa:
                        sb.wait_event(e0)
                    for i in range(NF):
                        j, st = (0, sa) if i % 2 == 0 else (k, sb)
                        with torch.cuda.stream(st):
                            plans[j].forward(x, pre[j])
                    if sb is not sa:
                        sa.wait_stream(sb)
                    e1.record(sa)
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / NF)
                return float(np.median(ts[1:]))

            single = timed(0, lane_set[0]["stream"])
            report = dict(single_ms=single, forwards_per_sample=NF, repetitions=REPS, pairs=[])
            for k in range(1, len(lane_set)):
                best, best_t, cand = None, None, lane_set[k]["stream"]
                for attempt in range(4):
                    t = timed(k, cand)
                    if best_t is None or t < best_t:
                        best, best_t = cand, t
                    if t < 0.94 * single:
                        break
                    cand = torch.cuda.Stream(device=self.device)
                lane_set[k]["stream"] = best
                report["pairs"].append(dict(lane=k, pair_ms=best_t, attempts=attempt + 1))
            report["pair_ms"] = max(p["pair_ms"] for p in report["pairs"])
            report["attempts"] = max(p["attempts"] for p in report["pairs"])
            cur.wait_stream(lane_set[0]["stream"])
            self._lane_check = report
            return report

    @contextlib.contextmanager
    def _lane_state(self, lane):
        """The pipeline state (_pipe, post_stream, _post_event, which plan) of `lane` installed for the duration of a call."""
        saved = (getattr(self, "_pipe", None), getattr(self, "post_stream", None), getattr(self, "_post_event", None))
        self._pipe, self.post_stream, self._post_event = lane["pipe"], lane["post_stream"], lane["post_event"]
        self._lane_plan = lane["which"]
        try:
            yield
        finally:
            lane["pipe"], lane["post_stream"], lane["post_event"] = self._pipe, self.post_stream, self._post_event
            self._lane_plan = 0
            self._pipe, self.post_stream, self._post_event = saved

    def _serving_lanes(self, defer, lanes=None):
        """The serving lanes of detect_filter_pipelined (None: single-lane operation).  Used for deferred steps on native plans.
        Count: the `lanes` argument, else the attribute serve_lanes (None = SQDET_SERVE_LANES from the environment, default 2).
        At batch 32 three lanes are no better than two (0.396 against 0.390 ms per step); at batch 1, where a forward leaves most of
        the chip idle, three give 11.1-11.8 k img/s against 8.3 k with two and four fall back to 8.4 k (bench.py's sqdet_sample_b1
        config asks for three); SqueezeDet+ at batch 8 loses 11 % with three (1.147 against 1.018 ms).  A change of the count
        flushes the old lanes first; the new set's streams are checked again at its first use (warm_up_lanes)."""
        n = lanes if lanes is not None else self.serve_lanes
        if n is None:
            n = int(os.environ.get("SQDET_SERVE_LANES", "2"))
        n = int(n)
        if n < 1:
            raise SqdetError("detect_filter_pipelined: lanes must be >= 1, got %r" % (n,))
        if not defer or self.NATIVE_ARCH is None or n < 2:
            if self._lanes is not None and defer and any(l["pipe"] is not None and l["pipe"].get("pending") is not None for l in self._lanes):
                # the lane set holds pending rows of earlier multi-lane calls: carried out ONCE, ahead of the first single-lane call (the
                # set is kept).  The single-lane pipe's own pending job is left alone -- it rides in this call's forward as ever.
                with torch.cuda.device(self.device):
                    self._flush_lane_set(torch.cuda.current_stream())
            return None
        if self._lanes is None or len(self._lanes) != n:
            if self._lanes is not None:
                self.flush_pipeline()
            self._lanes = [dict(which=k, stream=torch.cuda.Stream(device=self.device), in_ev=torch.cuda.Event(), pipe=None,
                                post_stream=None, post_event=None) for k in range(n)]
            self._lane_next = 0
            self._lanes_checked = False
            self._lane_check = None
        return self._lanes

    def flush_pipeline(self):
        """Enqueues the side work of the last call(s) now (nothing to overlap it with) and makes the CALLER's stream wait for all of
        it: every serving lane's stream AND every post-processing side stream (a lane's own, and the single-lane one).  After
        flush_pipeline() a synchronisation of the caller's stream alone (torch.cuda.current_stream().synchronize()) is enough to
        read every returned row, device or pinned host."""
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            self._flush_lane_set(cur)
            self._flush_pipe(getattr(self, "_pipe", None))
            if getattr(self, "post_stream", None) is not None:
                cur.wait_stream(self.post_stream)

    def _flush_lane_set(self, cur):
        """The serving lanes' half of flush_pipeline: every lane's pending side work enqueued on its own stream, `cur` waits for it."""
        if self._lanes is None:
            return
        for lane in self._lanes:
            with torch.cuda.stream(lane["stream"]), self._lane_state(lane):
                self._flush_pipe(self._pipe)
            cur.wait_stream(lane["stream"])
            if lane["post_stream"] is not None:
                cur.wait_stream(lane["post_stream"])

    def _flush_pipe(self, pipe):
        if pipe is not None and pipe.get("pending") is not None:
            s = pipe["pending"]
            if s.get("ride"):       # same stream as the forward: stream order is all the synchronisation there is
                cur = torch.cuda.current_stream()
                s["fwd_done"].record(cur)
                self._enqueue_post(s, None, stream=cur)
            else:
                self._enqueue_post(s, None)
            pipe["pending"] = None

    def _enqueue_post(self, s, gate, stream=None):
        """Decode + filter + row copy of slot s on the side stream, behind its forward (and `gate`, an event of a later forward)."""
        mc = self.mc
        pstream = stream if stream is not None else (torch.cuda.current_stream() if os.environ.get("SQDET_POST_INLINE") == "1" else self.post_stream)
        with torch.cuda.stream(pstream):
            pstream.wait_event(s["fwd_done"])
            if gate is not None:
                pstream.wait_event(gate)
            if s["fused_post"]:
                # decode + top-N + NMS in one call (score kernel unless scored + filter kernel): boxes / classes are decoded for the selected anchors only
                ops.detect_filter(s["preds"], self.anchors_f32(), mc.CLASSES, mc.ANCHOR_PER_GRID, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT,
                                  mc.EXP_THRESH, mc.TOP_N_DETECTION, mc.NMS_THRESH, scratch=s["det"][1], out=s["out"],
                                  scores_ready=s["scored"], max_workgroups=s["post_wgs"] if gate is not None else 0)
            else:
                ops.interpret_output(s["preds"], self.anchors_f32(), mc.CLASSES, mc.ANCHOR_PER_GRID, mc.IMAGE_WIDTH,
                                     mc.IMAGE_HEIGHT, mc.EXP_THRESH, out=s["det"])
                ops.filter_prediction(s["det"][0], s["det"][1], s["det"][2], mc.CLASSES, mc.TOP_N_DETECTION, mc.NMS_THRESH,
                                      mc.PROB_THRESH, out=s["out"])
            if s["to_host"]:
                if s["host"] is None:
                    s["host_flat"] = torch.empty(s["flat"].shape, dtype=torch.uint8).pin_memory()
                    s["host"] = self._out_views(s["host_flat"])
                ops.copy_to_pinned_host(s["flat"], s["host_flat"])      # all five outputs in one launch (never blocks the host)
            s["post_done"].record(pstream)

    def _detect_filter_pipelined(self, images, to_host, defer=False):
        mc = self.mc
        if getattr(self, "post_stream", None) is None:
            # high priority: the two small post-processing kernels are dispatched as soon as CUs free up at a kernel
            # boundary of the forward instead of waiting for its queue to drain
            self.post_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("SQDET_POST_PRIORITY", "-1")))
            self._post_event = torch.cuda.Event()
            self._pipe = None
        cur = torch.cuda.current_stream()
        if self.NATIVE_ARCH is not None:
            x = self._to_input(images)
            B = int(x.shape[0])
            plan = self._native_plan(B, getattr(self, "_lane_plan", 0))
            if self._pipe is None or self._pipe["batch"] != B:
                self.flush_pipeline()
                A = mc.ANCHORS
                M = mc.TOP_N_DETECTION if 0 < mc.TOP_N_DETECTION < A else min(A, 1024)
                f32, dev = torch.float32, self.device
                def out_views(flat):
                    """(boxes [B,M,4] f32, probs [B,M] f32, cls [B,M] i32, anchor index [B,M] i32, count [B] i32) as views of ONE
                    byte buffer, so the filtered rows leave the device in a single copy"""
                    o, views = 0, []
                    for shape, dt in (((B, M, 4), f32), ((B, M), f32), ((B, M), torch.int32), ((B, M), torch.int32), ((B,), torch.int32)):
                        nb = int(np.prod(shape)) * 4
                        views.append(flat[o:o + nb].view(dt).view(shape))
                        o += (nb + 255) // 256 * 256
                    return tuple(views)
                out_bytes = sum((int(np.prod(sh)) * 4 + 255) // 256 * 256 for sh in ((B, M, 4), (B, M), (B, M), (B, M), (B,)))

                def mk():
                    flat = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
                    sig = torch.cuda.Event()
                    sig.record(cur)                    # (creates the handle sqdet_net_set_signal is given)
                    return dict(preds=torch.empty((B, plan.gh, plan.gw, plan.out_ch), dtype=self.dtype, device=dev),
                                det=(torch.empty((B, A, 4), dtype=f32, device=dev), torch.empty((B, A), dtype=f32, device=dev),
                                     torch.empty((B, A), dtype=torch.int64, device=dev)),
                                flat=flat, out=out_views(flat), host_flat=None, host=None,
                                fwd_done=torch.cuda.Event(), post_done=torch.cuda.Event(), sig=sig, used=False)
                self._out_views = out_views
                self._pipe = dict(batch=B, slots=[mk(), mk()], k=0, pending=None)
            pipe = self._pipe
            s = pipe["slots"][pipe["k"] & 1]
            pipe["k"] += 1
            if s["used"]:
                cur.wait_event(s["post_done"])          # the side stream has finished reading this slot's preds
            fused_post = ops.detect_filter_supported(mc.ANCHORS, mc.TOP_N_DETECTION) and os.environ.get("SQDET_SPLIT_POST") != "1"
            # the score half of interpret_output rides in the ConvDet launch's epilogue where the plan has it (float16
            # SqueezeDet head): what is left for the side stream is ONE filter launch + the row copy
            # (SQDET_SCORE_EPILOGUE=0: the stand-alone score kernel on the side stream, for A/B)
            scored = fused_post and plan.scores_supported() and os.environ.get("SQDET_SCORE_EPILOGUE") != "0"
            mode = os.environ.get("SQDET_POST_DEFER", "ride")
            ride = bool(defer and scored and mode == "ride" and plan.rider_capacity() >= B)
            ov = plan.overlap_layer() if (defer and scored and mode == "signal") else -1
            deferred = ov >= 0
            s.update(fused_post=fused_post, scored=scored, to_host=to_host, ride=ride,
                     post_wgs=int(os.environ.get("SQDET_POST_WGS", "16")) if B > 16 else 0)
            if to_host and s["host"] is None:
                s["host_flat"] = torch.empty(s["flat"].shape, dtype=torch.uint8).pin_memory()
                s["host"] = self._out_views(s["host_flat"])
            prev = pipe["pending"]
            if ride:
                # everything on the caller's stream: the previous call's decode + filter rides in this forward's fire_chain
                # launches and writes its rows where the caller reads them (no post_done wait above either: a slot's preds /
                # scores are next overwritten by the ConvDet launch of the second-next forward, behind its riders in stream order)
                if prev is not None:
                    if prev.get("ride"):
                        plan.set_post_job(prev["preds"], prev["det"][1], self.anchors_f32(), prev["host"] if prev["to_host"] else prev["out"],
                                          mc.CLASSES, mc.ANCHOR_PER_GRID, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH,
                                          mc.TOP_N_DETECTION, mc.NMS_THRESH)
                    else:
                        self._enqueue_post(prev, None)
                plan.set_signal(-1, None)
                plan.forward(x, s["preds"], scores=s["det"][1])
                pipe["pending"] = s
                s["used"] = False                       # (no side-stream reader to wait for)
                return s["host"] if to_host else s["out"]
            plan.set_signal(ov, s["sig"] if (deferred and prev is not None) else None)
            plan.forward(x, s["preds"], scores=s["det"][1] if scored else None)
            s["fwd_done"].record(cur)
            if prev is not None:                        # the previous call's side work: beside THIS forward's fire_chain launches
                if prev.get("ride"):
                    self._enqueue_post(prev, None, stream=cur)
                else:
                    self._enqueue_post(prev, s["sig"] if deferred else None)
                pipe["pending"] = None
            if deferred:
                pipe["pending"] = s
            else:
                self._enqueue_post(s, None)
            s["used"] = True
            return s["host"] if to_host else s["out"]
        (preds,) = self.run([self.preds], {self.image_input: images})
        self._post_event.record(cur)
        with torch.cuda.stream(self.post_stream):
            self.post_stream.wait_event(self._post_event)
            boxes, probs, cls = ops.interpret_output(preds, self.anchors_f32(), mc.CLASSES, mc.ANCHOR_PER_GRID, mc.IMAGE_WIDTH,
                                                     mc.IMAGE_HEIGHT, mc.EXP_THRESH)[:3]
            out = self.filter_prediction_batch(boxes, probs, cls)
            preds.record_stream(self.post_stream)      # the allocator must not hand preds' memory out before the side stream is done
            if to_host:
                out = tuple(t_.to("cpu", non_blocking=True) for t_ in out)
        return out

    # ------------------------------------------------------------------ filter_prediction
    def filter_prediction_batch(self, det_boxes, det_probs, det_class, max_out=None):
        """Batched, device-resident filter_prediction: returns (boxes [B,M,4], probs [B,M], cls [B,M] i32,
        anchor_index [B,M] i32, count [B] i32) device tensors."""
        mc = self.mc
        return ops.filter_prediction(det_boxes, det_probs, det_class, mc.CLASSES, mc.TOP_N_DETECTION, mc.NMS_THRESH,
                                     mc.PROB_THRESH, max_out)

    def filter_prediction(self, boxes, probs, cls_idx):
        """Filter bounding box predictions with probability threshold and non-maximum
        supression (nn_skeleton.py:696-734).  Same arguments and return value as the reference:
          boxes: array of [cx, cy, w, h]; probs: array of probabilities; cls_idx: array of class indices
          -> (final_boxes, final_probs, final_cls_idx) Python lists, ordered by class.
        Runs on the GPU (top-N select + per-class NMS kernel)."""
        if not self.has_device:
            raise SqdetError("squeezedet_amd needs a HIP device to run: there is no CPU path")
        b = torch.as_tensor(np.ascontiguousarray(boxes, dtype=np.float32)).to(self.device).reshape(1, -1, 4)
        p = torch.as_tensor(np.ascontiguousarray(probs, dtype=np.float32)).to(self.device).reshape(1, -1)
        c = torch.as_tensor(np.ascontiguousarray(cls_idx, dtype=np.int64)).to(self.device).reshape(1, -1)
        ob, op, oc, oi, cnt = self.filter_prediction_batch(b, p, c)
        n = int(cnt[0].item())
        if n < 0:
            ob, op, oc, oi, cnt = self.filter_prediction_batch(b, p, c, max_out=-n)
            n = int(cnt[0].item())
        ob, op, oc = ob[0, :n].cpu().numpy(), op[0, :n].cpu().numpy(), oc[0, :n].cpu().numpy()
        return [ob[i] for i in range(n)], [op[i] for i in range(n)], [int(oc[i]) for i in range(n)]
