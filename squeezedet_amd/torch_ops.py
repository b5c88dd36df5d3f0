"""torch.ops.sqdet.*: the C-ABI entry points as PyTorch custom ops (north_star: "a thin C-ABI .so exposed through
PyTorch-ROCm custom ops"), with fake (meta) implementations for tracing and autograd formulas on the backward KERNELS of
libsqdet_hip.so -- never on PyTorch compute ops.  Call shapes follow the reference's builders:

  sqdet::conv2d(x, w_hwio, bias, stride, same, relu)          ModelSkeleton._conv_layer   nn_skeleton.py:471-563 (539-547)
  sqdet::maxpool(x, size, stride, same)                       ModelSkeleton._pooling_layer :565-586 (580)
  sqdet::fire(x, ws, bs, w1, b1, w3, b3)                      SqueezeDet._fire_layer      nets/squeezeDet.py:81-106
  sqdet::conv2d_nhwc / maxpool_nhwc                           the same two on PRE-PACKED kernels (inference: no pack per call)
  sqdet::interpret_output / filter_prediction / detect_filter _add_interpretation_graph :142-283, filter_prediction :696-734
  sqdet::net_forward(x, plan_id)                              _add_forward_graph as one native plan (register_plan() -> id)

conv2d / maxpool / fire are differentiable (stride-1 SAME convs, like every trainable conv of the reference's nets); their
backward passes are custom ops themselves (sqdet::conv2d_bwd, maxpool_bwd, fire_bwd), so torch.library.opcheck's
aot-dispatch test traces through them.  Kernels arrive HWIO float32 (the reference's variables) and are packed per call."""
import torch

from . import ops
from ._lib import SqdetError


def _pad(same):
    return "SAME" if same else "VALID"


_PLANS = {}


def register_plan(plan):
    """A NetPlan -> the integer handle sqdet::net_forward takes (custom ops carry tensors and scalars only)."""
    pid = len(_PLANS) + 1
    _PLANS[pid] = plan
    return pid


def _register():
    from torch.library import custom_op

    # ------------------------------------------------------------------ inference forms on pre-packed kernels
    @custom_op("sqdet::maxpool_nhwc", mutates_args=())
    def _maxpool_nhwc(x: torch.Tensor, size: int, stride: int, same: bool) -> torch.Tensor:
        return ops.maxpool_nhwc(x, size, stride, _pad(same))

    @_maxpool_nhwc.register_fake
    def _(x, size, stride, same):
        p = _pad(same)
        return x.new_empty((x.shape[0], ops._out_size(x.shape[1], size, stride, p), ops._out_size(x.shape[2], size, stride, p), x.shape[3]))

    @custom_op("sqdet::conv2d_nhwc", mutates_args=())
    def _conv_nhwc(x: torch.Tensor, packed: torch.Tensor, bias: torch.Tensor, k: int, cout: int, stride: int, same: bool,
                   relu: bool) -> torch.Tensor:
        pc = ops.PackedConv.__new__(ops.PackedConv)
        pc.k, pc.cin, pc.cout, pc.dtype, pc.data = k, int(x.shape[3]), cout, x.dtype, packed
        return ops.conv2d_nhwc(x, pc, bias, stride, _pad(same), relu)

    @_conv_nhwc.register_fake
    def _(x, packed, bias, k, cout, stride, same, relu):
        p = _pad(same)
        return x.new_empty((x.shape[0], ops._out_size(x.shape[1], k, stride, p), ops._out_size(x.shape[2], k, stride, p), cout))

    # ------------------------------------------------------------------ differentiable builders
    @custom_op("sqdet::conv2d", mutates_args=())
    def _conv(x: torch.Tensor, w_hwio: torch.Tensor, bias: torch.Tensor, stride: int, same: bool, relu: bool) -> torch.Tensor:
        return ops.conv2d_nhwc(x, ops.pack_conv_weights(w_hwio, x.dtype), bias, stride, _pad(same), relu)

    @_conv.register_fake
    def _(x, w_hwio, bias, stride, same, relu):
        p, k = _pad(same), w_hwio.shape[0]
        return x.new_empty((x.shape[0], ops._out_size(x.shape[1], k, stride, p), ops._out_size(x.shape[2], k, stride, p), w_hwio.shape[3]))

    @custom_op("sqdet::conv2d_bwd", mutates_args=())
    def _conv_bwd(x: torch.Tensor, w_hwio: torch.Tensor, y: torch.Tensor, gy: torch.Tensor, relu: bool,
                  need_dx: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        k, _, cin, cout = [int(v) for v in w_hwio.shape]
        g = gy.contiguous().clone()
        if relu:
            ops.relu_bwd(y, g)                                            # g *= (y > 0)
        dw, db = ops.conv2d_bwd_filter(x, g, k, cin, cout)
        dx = ops.conv2d_bwd_data(g, ops.PackedConvBwd(w_hwio, x.dtype)) if need_dx else torch.zeros_like(x)
        return dx, dw, db

    @_conv_bwd.register_fake
    def _(x, w_hwio, y, gy, relu, need_dx):
        return torch.empty_like(x), torch.empty_like(w_hwio, dtype=torch.float32), w_hwio.new_empty((w_hwio.shape[3],), dtype=torch.float32)

    def _conv_setup(ctx, inputs, output):
        x, w, b, stride, same, relu = inputs
        if stride != 1 or not same:
            if any(isinstance(t, torch.Tensor) and t.requires_grad for t in (x, w, b)):
                # fail where the graph is BUILT, not at backward time
                raise SqdetError("sqdet::conv2d: only stride-1 SAME convs have a backward kernel (the reference's conv1 is frozen): "
                                 "detach the inputs of a stride-%d %s conv" % (stride, "SAME" if same else "VALID"))
            ctx.unsupported = True
            return
        ctx.unsupported = False
        ctx.relu = relu
        ctx.save_for_backward(x, w, output)

    def _conv_backward(ctx, gy):
        if ctx.unsupported:
            raise SqdetError("sqdet::conv2d: only stride-1 SAME convs have a backward kernel (the reference's conv1 is frozen)")
        x, w, y = ctx.saved_tensors
        dx, dw, db = torch.ops.sqdet.conv2d_bwd(x, w, y, gy, ctx.relu, ctx.needs_input_grad[0])
        return (dx if ctx.needs_input_grad[0] else None), dw.to(w.dtype), db, None, None, None

    _conv.register_autograd(_conv_backward, setup_context=_conv_setup)

    @custom_op("sqdet::maxpool", mutates_args=())
    def _pool(x: torch.Tensor, size: int, stride: int, same: bool) -> torch.Tensor:
        return ops.maxpool_nhwc(x, size, stride, _pad(same))

    @_pool.register_fake
    def _(x, size, stride, same):
        p = _pad(same)
        return x.new_empty((x.shape[0], ops._out_size(x.shape[1], size, stride, p), ops._out_size(x.shape[2], size, stride, p), x.shape[3]))

    @custom_op("sqdet::maxpool_bwd", mutates_args=())
    def _pool_bwd(x: torch.Tensor, gy: torch.Tensor, size: int, stride: int, same: bool) -> torch.Tensor:
        return ops.maxpool_bwd(x, gy.contiguous(), size, stride, _pad(same))

    @_pool_bwd.register_fake
    def _(x, gy, size, stride, same):
        return torch.empty_like(x)

    def _pool_setup(ctx, inputs, output):
        x, ctx.size, ctx.stride, ctx.same = inputs
        ctx.save_for_backward(x)

    def _pool_backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return torch.ops.sqdet.maxpool_bwd(x, gy, ctx.size, ctx.stride, ctx.same), None, None, None

    _pool.register_autograd(_pool_backward, setup_context=_pool_setup)

    @custom_op("sqdet::fire", mutates_args=())
    def _fire(x: torch.Tensor, ws: torch.Tensor, bs: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w3: torch.Tensor,
              b3: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """-> (concat(expand1x1, expand3x3), the squeeze tensor).  Three launches, the expand convs writing the two halves of
        the concat tensor (the fused fire kernels keep the squeeze tensor in LDS; the backward needs it in memory)."""
        pk = lambda w: ops.pack_conv_weights(w, x.dtype)
        sq = ops.conv2d_nhwc(x, pk(ws), bs, 1, "SAME", True)
        ne1 = int(w1.shape[3])
        y = torch.empty((x.shape[0], x.shape[1], x.shape[2], ne1 + int(w3.shape[3])), dtype=x.dtype, device=x.device)
        ops.conv2d_nhwc(sq, pk(w1), b1, 1, "SAME", True, out=y, out_coffset=0)
        ops.conv2d_nhwc(sq, pk(w3), b3, 1, "SAME", True, out=y, out_coffset=ne1)
        return y, sq

    @_fire.register_fake
    def _(x, ws, bs, w1, b1, w3, b3):
        return (x.new_empty((x.shape[0], x.shape[1], x.shape[2], w1.shape[3] + w3.shape[3])),
                x.new_empty((x.shape[0], x.shape[1], x.shape[2], ws.shape[3])))

    @custom_op("sqdet::fire_bwd", mutates_args=())
    def _fire_bwd(x: torch.Tensor, sq: torch.Tensor, y: torch.Tensor, gy: torch.Tensor, ws: torch.Tensor, w1: torch.Tensor,
                  w3: torch.Tensor, need_dx: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
                                                           torch.Tensor, torch.Tensor]:
        """-> (dx, dws, dbs, dw1, db1, dw3, db3): the backward of SqueezeDetTrainer's fire record (train.py), kernel for kernel"""
        ns, ne1, ne3, cin = int(ws.shape[3]), int(w1.shape[3]), int(w3.shape[3]), int(x.shape[3])
        g = gy.contiguous().clone()
        ops.relu_bwd(y, g)                                                # both expand convs end in ReLU
        dw1, db1 = ops.conv2d_bwd_filter(sq, g, 1, ns, ne1, dy_coffset=0)
        dw3, db3 = ops.conv2d_bwd_filter(sq, g, 3, ns, ne3, dy_coffset=ne1)
        ds = ops.conv2d_bwd_data(g, ops.PackedConvBwd(w1, x.dtype), dy_coffset=0)
        ops.conv2d_bwd_data(g, ops.PackedConvBwd(w3, x.dtype), dx=ds, dy_coffset=ne1, accumulate=True, relu_of=sq)   # + the squeeze's ReLU
        dws, dbs = ops.conv2d_bwd_filter(x, ds, 1, cin, ns)
        dx = ops.conv2d_bwd_data(ds, ops.PackedConvBwd(ws, x.dtype)) if need_dx else torch.zeros_like(x)
        return dx, dws, dbs, dw1, db1, dw3, db3

    @_fire_bwd.register_fake
    def _(x, sq, y, gy, ws, w1, w3, need_dx):
        f = lambda w: torch.empty_like(w, dtype=torch.float32)
        v = lambda w: w.new_empty((w.shape[3],), dtype=torch.float32)
        return torch.empty_like(x), f(ws), v(ws), f(w1), v(w1), f(w3), v(w3)

    def _fire_setup(ctx, inputs, output):
        x, ws, bs, w1, b1, w3, b3 = inputs
        y, sq = output
        # the squeeze tensor is an auxiliary output (what the backward kernels need in memory): a graph that consumes it must not
        # silently lose that gradient -- it is marked non-differentiable
        ctx.mark_non_differentiable(sq)
        ctx.save_for_backward(x, sq, y, ws, w1, w3)

    def _fire_backward(ctx, gy, gsq):
        # (the squeeze tensor is an auxiliary output: no gradient flows into it from outside the module)
        x, sq, y, ws, w1, w3 = ctx.saved_tensors
        dx, dws, dbs, dw1, db1, dw3, db3 = torch.ops.sqdet.fire_bwd(x, sq, y, gy, ws, w1, w3, ctx.needs_input_grad[0])
        return (dx if ctx.needs_input_grad[0] else None), dws.to(ws.dtype), dbs, dw1.to(w1.dtype), db1, dw3.to(w3.dtype), db3

    _fire.register_autograd(_fire_backward, setup_context=_fire_setup)

    # ------------------------------------------------------------------ the detection half
    @custom_op("sqdet::interpret_output", mutates_args=())
    def _interp(preds: torch.Tensor, anchors: torch.Tensor, classes: int, apg: int, img_w: float, img_h: float,
                exp_thresh: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return ops.interpret_output(preds, anchors, classes, apg, img_w, img_h, exp_thresh)

    @_interp.register_fake
    def _(preds, anchors, classes, apg, img_w, img_h, exp_thresh):
        n, A = preds.shape[0], anchors.shape[0]
        return (preds.new_empty((n, A, 4), dtype=torch.float32), preds.new_empty((n, A), dtype=torch.float32),
                preds.new_empty((n, A), dtype=torch.int64))

    @custom_op("sqdet::filter_prediction", mutates_args=())
    def _filt(boxes: torch.Tensor, probs: torch.Tensor, cls: torch.Tensor, classes: int, top_n: int, nms_thresh: float,
              prob_thresh: float, max_out: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        return ops.filter_prediction(boxes, probs, cls, classes, top_n, nms_thresh, prob_thresh, max_out)

    @_filt.register_fake
    def _(boxes, probs, cls, classes, top_n, nms_thresh, prob_thresh, max_out):
        n = probs.shape[0]
        return (probs.new_empty((n, max_out, 4)), probs.new_empty((n, max_out)),
                probs.new_empty((n, max_out), dtype=torch.int32), probs.new_empty((n, max_out), dtype=torch.int32),
                probs.new_empty((n,), dtype=torch.int32))

    @custom_op("sqdet::detect_filter", mutates_args=())
    def _detf(preds: torch.Tensor, anchors: torch.Tensor, classes: int, apg: int, img_w: float, img_h: float, exp_thresh: float,
              top_n: int, nms_thresh: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        return ops.detect_filter(preds, anchors, classes, apg, img_w, img_h, exp_thresh, top_n, nms_thresh)

    @_detf.register_fake
    def _(preds, anchors, classes, apg, img_w, img_h, exp_thresh, top_n, nms_thresh):
        n = preds.shape[0]
        f = lambda shape, dt: preds.new_empty(shape, dtype=dt)
        return (f((n, top_n, 4), torch.float32), f((n, top_n), torch.float32), f((n, top_n), torch.int32), f((n, top_n), torch.int32),
                f((n,), torch.int32))

    @custom_op("sqdet::net_forward", mutates_args=())
    def _net(image_input: torch.Tensor, plan_id: int) -> torch.Tensor:
        plan = _PLANS.get(int(plan_id))
        if plan is None:
            raise SqdetError("sqdet::net_forward: unknown plan id %d (torch_ops.register_plan)" % plan_id)
        return plan.forward(image_input)

    @_net.register_fake
    def _(image_input, plan_id):
        plan = _PLANS[int(plan_id)]
        return image_input.new_empty((image_input.shape[0], plan.gh, plan.gw, plan.out_ch))


try:
    _register()
except Exception as e:  # pragma: no cover -- a torch without torch.library.custom_op: the ctypes surface (ops.py) is unaffected
    import warnings
    warnings.warn("squeezedet_amd: torch.ops.sqdet.* not registered (%r)" % (e,))
