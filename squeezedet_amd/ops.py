"""Tensor-level entry points of the HIP hot path (torch is plumbing only: device memory and
streams).  Every function launches a kernel of libsqdet_hip.so through the C ABI
(include/sqdet.h) on the current torch stream; none has a CPU implementation.

Also registered as PyTorch custom ops under ``torch.ops.sqdet.*`` (see the bottom of the
file) so graph-level callers can use them.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, dtype_code, lib, pad_code, stream_ptr


def _dev(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.SqdetError("%s must be a CUDA(HIP) tensor -- there is no CPU path" % name)
    if dtype is not None and t.dtype != dtype:
        raise _lib.SqdetError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.SqdetError("%s must be contiguous" % name)
    return C.c_void_p(t.data_ptr())


def _out_size(n, k, s, padding):
    return -(-n // s) if padding.upper() == "SAME" else (n - k) // s + 1


# ---------------------------------------------------------------- conv
class PackedConv:
    """A conv kernel re-laid-out in MFMA fragment order (sqdet_conv_pack_weights)."""

    def __init__(self, w_hwio, dtype):
        w = w_hwio.detach().to(torch.float32).contiguous()
        self.k, k2, self.cin, self.cout = [int(v) for v in w.shape]
        assert self.k == k2, "square kernels only"
        self.dtype = dtype
        code = dtype_code(dtype)
        nbytes = lib().sqdet_conv_packed_bytes(self.k, self.cin, self.cout, code)
        self.data = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        check(lib().sqdet_conv_pack_weights(_dev(w, "w_hwio"), _dev(self.data, "packed"), self.k, self.cin, self.cout,
                                            code, stream_ptr()), "sqdet_conv_pack_weights")


def pack_conv_weights(w_hwio, dtype):
    return PackedConv(w_hwio, dtype)


def fold_batchnorm(w_hwio, conv_bias, gamma, beta, mean, var, eps):
    """Frozen batch norm of _conv_bn_layer (nn_skeleton.py:374-468) folded into its conv
    (sqdet_fold_batchnorm): returns (w_folded [k,k,cin,cout] f32, b_folded [cout] f32)."""
    w = w_hwio.detach().to(torch.float32).contiguous()
    k, _, cin, cout = [int(v) for v in w.shape]
    vecs =[v.detach().to(torch.float32).contiguous() for v in (gamma, beta, mean, var)]
    cb = conv_bias.detach().to(torch.float32).contiguous() if conv_bias is not None else None
    wf = torch.empty_like(w)
    bf = torch.empty(cout, dtype=torch.float32, device=w.device)
    check(lib().sqdet_fold_batchnorm(_dev(w, "w_hwio"), _dev(cb, "conv_bias") if cb is not None else None,
                                     _dev(vecs[0], "gamma"), _dev(vecs[1], "beta"), _dev(vecs[2], "mean"),
                                     _dev(vecs[3], "var"), float(eps), _dev(wf, "w_folded"), _dev(bf, "b_folded"),
                                     k, cin, cout, stream_ptr()), "sqdet_fold_batchnorm")
    return wf, bf


def conv2d_nhwc(x, packed, bias, stride=1, padding="SAME", relu=True, out=None, out_coffset=0, accumulate=False, residual=None):
    """relu?(conv2d(x, W) + b) with TF SAME/VALID semantics (nn_skeleton.py:471-563).
    x: [N,H,W,Cin] f16/f32 NHWC; packed: PackedConv; bias: f32 [Cout].  ``out`` (optional)
    is a [N,Ho,Wo,Ctot] tensor whose channels [out_coffset, out_coffset+Cout) are written
    (fire-module concat without a concat pass).  accumulate=True: out = relu?(conv + b + out), the
    residual add of a ResNet bottleneck (resnet50_convDet.py:55) done in the conv epilogue;
    residual=<tensor shaped like out>: out = relu?(conv + b + residual), the shortcut left untouched."""
    if residual is not None:
        n, h, w, cin = [int(v) for v in x.shape]
        ho, wo = _out_size(h, packed.k, stride, padding), _out_size(w, packed.k, stride, padding)
        if cin != packed.cin or x.dtype != packed.dtype:
            raise _lib.SqdetError("conv2d_nhwc: input does not match the packed kernel")
        if out is None:
            out = torch.empty((n, ho, wo, packed.cout), dtype=x.dtype, device=x.device)
            out_coffset = 0
        if tuple(out.shape[:3]) != (n, ho, wo) or out.dtype != x.dtype or tuple(residual.shape) != tuple(out.shape) or residual.dtype != out.dtype:
            raise _lib.SqdetError("conv2d_nhwc: `residual` must have the shape and dtype of `out`")
        check(lib().sqdet_conv2d_res_nhwc_fwd(_dev(x, "x"), _dev(packed.data, "packed"), _dev(bias, "bias", torch.float32), _dev(residual, "residual"),
                                              _dev(out, "out"), n, h, w, cin, packed.cout, packed.k, int(stride), pad_code(padding), int(bool(relu)),
                                              dtype_code(x.dtype), int(out.shape[3]), int(out_coffset), stream_ptr()), "sqdet_conv2d_res_nhwc_fwd")
        return out
    n, h, w, cin = [int(v) for v in x.shape]
    if cin != packed.cin or x.dtype != packed.dtype:
        raise _lib.SqdetError("conv2d_nhwc: input [%d ch, %s] does not match packed kernel [%d ch, %s]"
                              % (cin, x.dtype, packed.cin, packed.dtype))
    ho, wo = _out_size(h, packed.k, stride, padding), _out_size(w, packed.k, stride, padding)
    if out is None:
        out = torch.empty((n, ho, wo, packed.cout), dtype=x.dtype, device=x.device)
        out_coffset = 0
        if accumulate:
            raise _lib.SqdetError("conv2d_nhwc: accumulate=True needs `out` (the shortcut branch)")
    elif tuple(out.shape[:3]) != (n, ho, wo) or out.dtype != x.dtype:
        raise _lib.SqdetError("conv2d_nhwc: bad `out` shape/dtype")
    fn = lib().sqdet_conv2d_add_nhwc_fwd if accumulate else lib().sqdet_conv2d_nhwc_fwd
    check(fn(_dev(x, "x"), _dev(packed.data, "packed"), _dev(bias, "bias", torch.float32),
             _dev(out, "out"), n, h, w, cin, packed.cout, packed.k, int(stride),
             pad_code(padding), int(bool(relu)), dtype_code(x.dtype), int(out.shape[3]),
             int(out_coffset), stream_ptr()), "sqdet_conv2d_add_nhwc_fwd" if accumulate else "sqdet_conv2d_nhwc_fwd")
    return out


def maxpool_nhwc(x, size, stride, padding="SAME"):
    """tf.nn.max_pool semantics (nn_skeleton.py:565-586)."""
    n, h, w, c = [int(v) for v in x.shape]
    ho, wo = _out_size(h, size, stride, padding), _out_size(w, size, stride, padding)
    y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    check(lib().sqdet_maxpool_nhwc_fwd(_dev(x, "x"), _dev(y, "y"), n, h, w, c, int(size), int(stride),
                                       pad_code(padding), dtype_code(x.dtype), stream_ptr()), "sqdet_maxpool_nhwc_fwd")
    return y


def maxpool_nhwc_idx(x, size, stride, padding="SAME"):
    """maxpool_nhwc that also returns the uint8 window index of every output element (sqdet_maxpool_nhwc_fwd_idx): what
    maxpool_bwd_idx needs instead of x."""
    n, h, w, c = [int(v) for v in x.shape]
    ho, wo = _out_size(h, size, stride, padding), _out_size(w, size, stride, padding)
    y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    check(lib().sqdet_maxpool_nhwc_fwd_idx(_dev(x, "x"), _dev(y, "y"), _dev(idx, "idx"), n, h, w, c, int(size), int(stride),
                                           pad_code(padding), dtype_code(x.dtype), stream_ptr()), "sqdet_maxpool_nhwc_fwd_idx")
    return y, idx


def stem_supported(cout, k):
    """Shapes the fused conv1 + pool1 launch covers (sqdet_stem_conv_pool_fwd): SqueezeDet, SqueezeDet+ and ResNet50 stems."""
    return (k == 3 and cout == 64) or (k == 7 and cout in (64, 96))


def stem_conv_pool(x, packed, bias, conv_padding="SAME", pool_padding="SAME"):
    """conv1 + pool1 fused: max_pool3x3/s2(relu(conv(x, stride 2) + b)) (nets/squeezeDet.py:40-44)."""
    n, h, w, cin = [int(v) for v in x.shape]
    if cin != 3 or packed.cin != 3 or x.dtype != packed.dtype:
        raise _lib.SqdetError("stem_conv_pool: needs a 3-channel input matching the packed kernel")
    hc, wc = _out_size(h, packed.k, 2, conv_padding), _out_size(w, packed.k, 2, conv_padding)
    hp, wp = _out_size(hc, 3, 2, pool_padding), _out_size(wc, 3, 2, pool_padding)
    y = torch.empty((n, hp, wp, packed.cout), dtype=x.dtype, device=x.device)
    check(lib().sqdet_stem_conv_pool_fwd(_dev(x, "x"), _dev(packed.data, "packed"), _dev(bias, "bias", torch.float32),
                                         _dev(y, "y"), n, h, w, packed.cout, packed.k, pad_code(conv_padding),
                                         pad_code(pool_padding), dtype_code(x.dtype), stream_ptr()),
          "sqdet_stem_conv_pool_fwd")
    return y


def stem_conv_pool_squeeze(x, packed, bias, p_next_s, b_next_s, conv_padding="SAME", pool_padding="SAME"):
    """conv1 + pool1 + the next module's squeeze1x1 in one launch (sqdet_stem_conv_pool_squeeze_fwd): only the squeeze tensor
    [N, Hp, Wp, next_s] is written."""
    n, h, w, cin = [int(v) for v in x.shape]
    if cin != 3 or packed.cin != 3 or x.dtype != packed.dtype:
        raise _lib.SqdetError("stem_conv_pool_squeeze: needs a 3-channel input matching the packed kernel")
    hc, wc = _out_size(h, packed.k, 2, conv_padding), _out_size(w, packed.k, 2, conv_padding)
    hp, wp = _out_size(hc, 3, 2, pool_padding), _out_size(wc, 3, 2, pool_padding)
    so = torch.empty((n, hp, wp, p_next_s.cout), dtype=x.dtype, device=x.device)
    check(lib().sqdet_stem_conv_pool_squeeze_fwd(_dev(x, "x"), _dev(packed.data, "packed"), _dev(bias, "bias", torch.float32),
                                                 _dev(p_next_s.data, "w_next_s"), _dev(b_next_s, "b_next_s", torch.float32), _dev(so, "sq_out"),
                                                 n, h, w, packed.cout, packed.k, pad_code(conv_padding), pad_code(pool_padding),
                                                 p_next_s.cout, dtype_code(x.dtype), stream_ptr()), "sqdet_stem_conv_pool_squeeze_fwd")
    return so


def fire(x, p_s, b_s, p_e1, b_e1, p_e3, b_e3, keep_squeeze=False):
    """SqueezeDet._fire_layer (nets/squeezeDet.py:81-106).  keep_squeeze: returns (y, squeeze tensor) -- sqdet_fire_fwd_keep,
    the fused launch also writes the squeeze tensor its backward needs."""
    n, h, w, cin = [int(v) for v in x.shape]
    sq = torch.empty((n, h, w, p_s.cout), dtype=x.dtype, device=x.device)
    y = torch.empty((n, h, w, p_e1.cout + p_e3.cout), dtype=x.dtype, device=x.device)
    fn = lib().sqdet_fire_fwd_keep if keep_squeeze else lib().sqdet_fire_fwd
    check(fn(_dev(x, "x"), _dev(p_s.data, "w_s"), _dev(b_s, "b_s", torch.float32),
                               _dev(p_e1.data, "w_e1"), _dev(b_e1, "b_e1", torch.float32),
                               _dev(p_e3.data, "w_e3"), _dev(b_e3, "b_e3", torch.float32),
                               _dev(sq, "sq"), _dev(y, "y"), n, h, w, cin, p_s.cout, p_e1.cout, p_e3.cout,
                               dtype_code(x.dtype), stream_ptr()), "sqdet_fire_fwd")
    return (y, sq) if keep_squeeze else y


def fire_maxpool(x, p_s, b_s, p_e1, b_e1, p_e3, b_e3):
    """_fire_layer followed by max_pool 3x3/s2 SAME (nets/squeezeDet.py:49-57) -> the pooled tensor, in one launch
    where the streaming kernel covers the shape (sqdet_fire_maxpool_fwd)."""
    n, h, w, cin = [int(v) for v in x.shape]
    ctot = p_e1.cout + p_e3.cout
    sq = torch.empty((n, h, w, p_s.cout), dtype=x.dtype, device=x.device)
    full = torch.empty((n, h, w, ctot), dtype=x.dtype, device=x.device)
    y = torch.empty((n, -(-h // 2), -(-w // 2), ctot), dtype=x.dtype, device=x.device)
    check(lib().sqdet_fire_maxpool_fwd(_dev(x, "x"), _dev(p_s.data, "w_s"), _dev(b_s, "b_s", torch.float32),
                                       _dev(p_e1.data, "w_e1"), _dev(b_e1, "b_e1", torch.float32),
                                       _dev(p_e3.data, "w_e3"), _dev(b_e3, "b_e3", torch.float32),
                                       _dev(sq, "sq"), _dev(full, "fire_scratch"), _dev(y, "y"), n, h, w, cin, p_s.cout,
                                       p_e1.cout, p_e3.cout, dtype_code(x.dtype), stream_ptr()), "sqdet_fire_maxpool_fwd")
    return y


def fire_expand(sq_in, p_e1, b_e1, p_e3, b_e3, pool=False):
    """Expand half of a fire module from its squeeze tensor (+ the 3x3/s2 SAME max-pool behind it): sqdet_fire_expand_fwd."""
    n, h, w, s = [int(v) for v in sq_in.shape]
    ctot = p_e1.cout + p_e3.cout
    shape = (n, -(-h // 2), -(-w // 2), ctot) if pool else (n, h, w, ctot)
    y = torch.empty(shape, dtype=sq_in.dtype, device=sq_in.device)
    check(lib().sqdet_fire_expand_fwd(_dev(sq_in, "sq_in"), _dev(p_e1.data, "w_e1"), _dev(b_e1, "b_e1", torch.float32),
                                      _dev(p_e3.data, "w_e3"), _dev(b_e3, "b_e3", torch.float32), _dev(y, "y"), n, h, w, s,
                                      p_e1.cout, p_e3.cout, int(bool(pool)), dtype_code(sq_in.dtype), stream_ptr()), "sqdet_fire_expand_fwd")
    return y


def fire_squeeze_next(x, p_s, b_s, p_e1, b_e1, p_e3, b_e3, p_next_s, b_next_s):
    """A whole fire module from x, emitting the NEXT module's squeeze tensor instead of its concat tensor
    (sqdet_fire_squeeze_next_fwd)."""
    n, h, w, cin = [int(v) for v in x.shape]
    so = torch.empty((n, h, w, p_next_s.cout), dtype=x.dtype, device=x.device)
    check(lib().sqdet_fire_squeeze_next_fwd(_dev(x, "x"), _dev(p_s.data, "w_s"), _dev(b_s, "b_s", torch.float32),
                                            _dev(p_e1.data, "w_e1"), _dev(b_e1, "b_e1", torch.float32),
                                            _dev(p_e3.data, "w_e3"), _dev(b_e3, "b_e3", torch.float32),
                                            _dev(p_next_s.data, "w_next_s"), _dev(b_next_s, "b_next_s", torch.float32), _dev(so, "sq_out"),
                                            n, h, w, cin, p_s.cout, p_e1.cout, p_e3.cout, p_next_s.cout, dtype_code(x.dtype), stream_ptr()),
          "sqdet_fire_squeeze_next_fwd")
    return so


def fire_expand_squeeze_next(sq_in, p_e1, b_e1, p_e3, b_e3, p_next_s, b_next_s, pool=False):
    """Expand half of a module from its squeeze tensor (+ pool) emitting the NEXT module's squeeze tensor
    (sqdet_fire_expand_squeeze_next_fwd)."""
    n, h, w, s = [int(v) for v in sq_in.shape]
    shape = (n, -(-h // 2), -(-w // 2), p_next_s.cout) if pool else (n, h, w, p_next_s.cout)
    so = torch.empty(shape, dtype=sq_in.dtype, device=sq_in.device)
    check(lib().sqdet_fire_expand_squeeze_next_fwd(_dev(sq_in, "sq_in"), _dev(p_e1.data, "w_e1"), _dev(b_e1, "b_e1", torch.float32),
                                                   _dev(p_e3.data, "w_e3"), _dev(b_e3, "b_e3", torch.float32),
                                                   _dev(p_next_s.data, "w_next_s"), _dev(b_next_s, "b_next_s", torch.float32),
                                                   _dev(so, "sq_out"), n, h, w, s, p_e1.cout, p_e3.cout, p_next_s.cout, int(bool(pool)),
                                                   dtype_code(sq_in.dtype), stream_ptr()), "sqdet_fire_expand_squeeze_next_fwd")
    return so


class FireChainStream:
    """The packed weight stream of sqdet_fire_chain_fwd: expand1x1 + expand3x3 kernels of one fire module and,
    optionally, the squeeze1x1 kernel of the next one (float32 HWIO in, float16 stream out)."""

    def __init__(self, w_e1_hwio, w_e3_hwio, w_next_s_hwio=None, dtype=torch.float16):
        w1 = w_e1_hwio.detach().to(torch.float32).contiguous()
        w3 = w_e3_hwio.detach().to(torch.float32).contiguous()
        ws = w_next_s_hwio.detach().to(torch.float32).contiguous() if w_next_s_hwio is not None else None
        self.s, self.e1, self.e3 = int(w1.shape[2]), int(w1.shape[3]), int(w3.shape[3])
        self.s2 = int(ws.shape[3]) if ws is not None else 0
        if tuple(w1.shape[:2]) != (1, 1) or tuple(w3.shape[:3]) != (3, 3, self.s) or \
                (ws is not None and tuple(ws.shape[:3]) != (1, 1, self.e1 + self.e3)):
            raise _lib.SqdetError("FireChainStream: kernel shapes do not form expand1x1 / expand3x3 / next squeeze1x1")
        self.dtype = dtype
        code = dtype_code(dtype)
        nbytes = lib().sqdet_fire_chain_stream_bytes(self.s, self.e1, self.e3, self.s2, code)
        if nbytes == 0:
            raise _lib.SqdetError("fire chain kernel does not cover s=%d e1=%d e3=%d next_s=%d %s"
                                  % (self.s, self.e1, self.e3, self.s2, dtype))
        self.data = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
        check(lib().sqdet_fire_chain_pack(_dev(w1, "w_e1"), _dev(w3, "w_e3"), _dev(ws, "w_next_s") if ws is not None else None,
                                          _dev(self.data, "stream"), self.s, self.e1, self.e3, self.s2, code, stream_ptr()),
              "sqdet_fire_chain_pack")


def fire_chain_supported(s, e1, e3, next_s, dtype):
    return lib().sqdet_fire_chain_stream_bytes(int(s), int(e1), int(e3), int(next_s), dtype_code(dtype)) > 0


def fire_chain(sq_in, chain, b_e1, b_e3, b_next_s=None, want_y=False):
    """Expand half of a fire module + the next module's squeeze in one launch (sqdet_fire_chain_fwd).
    sq_in [N,H,W,S] float16 = the module's squeeze tensor.  Returns (y or None, sq_out or None)."""
    n, h, w, s = [int(v) for v in sq_in.shape]
    if s != chain.s or sq_in.dtype != chain.dtype:
        raise _lib.SqdetError("fire_chain: input does not match the packed stream")
    y = torch.empty((n, h, w, chain.e1 + chain.e3), dtype=sq_in.dtype, device=sq_in.device) if (want_y or chain.s2 == 0) else None
    so = torch.empty((n, h, w, chain.s2), dtype=sq_in.dtype, device=sq_in.device) if chain.s2 > 0 else None
    check(lib().sqdet_fire_chain_fwd(_dev(sq_in, "sq_in"), _dev(chain.data, "stream"), _dev(b_e1, "b_e1", torch.float32),
                                     _dev(b_e3, "b_e3", torch.float32),
                                     _dev(b_next_s, "b_next_s", torch.float32) if chain.s2 > 0 else None,
                                     _dev(y, "y") if y is not None else None, _dev(so, "sq_out") if so is not None else None,
                                     n, h, w, chain.s, chain.e1, chain.e3, chain.s2, dtype_code(sq_in.dtype), stream_ptr()),
          "sqdet_fire_chain_fwd")
    return y, so


# ---------------------------------------------------------------- post-processing
def interpret_output(preds, anchors_f32, classes, anchors_per_grid, img_w, img_h, exp_thresh=1.0,
                     with_class_probs=False, out=None):
    """_add_interpretation_graph (nn_skeleton.py:142-283).  preds [N,gh,gw,K*(C+5)] f16/f32;
    anchors_f32 [A,4] float32 device tensor.  Returns det_boxes [N,A,4] f32, det_probs [N,A] f32,
    det_class [N,A] int64 (+ pred_class_probs [N,A,C], pred_conf [N,A] when asked).  out: optional preallocated
    (det_boxes, det_probs, det_class) to write into (serving loops: no allocation per step)."""
    n, gh, gw, ch = [int(v) for v in preds.shape]
    if ch != anchors_per_grid * (classes + 5):
        raise _lib.SqdetError("interpret_output: %d channels != %d*(%d+5)" % (ch, anchors_per_grid, classes))
    A = gh * gw * anchors_per_grid
    if tuple(anchors_f32.shape) != (A, 4):
        raise _lib.SqdetError("interpret_output: anchors must be [%d,4]" % A)
    dev = preds.device
    if out is not None:
        boxes, probs, cls = out
        if tuple(boxes.shape) != (n, A, 4) or tuple(probs.shape) != (n, A) or tuple(cls.shape) != (n, A) or cls.dtype != torch.int64:
            raise _lib.SqdetError("interpret_output: out tensors must be [N,A,4] f32, [N,A] f32, [N,A] int64")
    else:
        boxes = torch.empty((n, A, 4), dtype=torch.float32, device=dev)
        probs = torch.empty((n, A), dtype=torch.float32, device=dev)
        cls = torch.empty((n, A), dtype=torch.int64, device=dev)
    pcp = torch.empty((n, A, classes), dtype=torch.float32, device=dev) if with_class_probs else None
    pconf = torch.empty((n, A), dtype=torch.float32, device=dev) if with_class_probs else None
    check(lib().sqdet_interpret_output(_dev(preds, "preds"), _dev(anchors_f32, "anchors", torch.float32),
                                       _dev(boxes, "boxes"), _dev(probs, "probs"), _dev(cls, "cls"),
                                       _dev(pcp, "pcp") if pcp is not None else None,
                                       _dev(pconf, "pconf") if pconf is not None else None,
                                       n, gh, gw, int(anchors_per_grid), int(classes), float(img_w), float(img_h),
                                       float(exp_thresh), dtype_code(preds.dtype), stream_ptr()),
          "sqdet_interpret_output")
    if with_class_probs:
        return boxes, probs, cls, pcp, pconf
    return boxes, probs, cls


def filter_prediction(boxes, probs, cls, classes, top_n, nms_thresh, prob_thresh, max_out=None, out=None):
    """Batched ModelSkeleton.filter_prediction (nn_skeleton.py:696-734).  boxes [N,A,4] f32,
    probs [N,A] f32, cls [N,A] int64 (device).  Returns device tensors
    (out_boxes [N,M,4], out_probs [N,M], out_cls [N,M] i32, out_index [N,M] i32, out_count [N] i32)
    with rows [0,count) valid, ordered by class then descending prob.  out: optional preallocated tuple of those five."""
    n, A = int(probs.shape[0]), int(probs.shape[1])
    use_topn = 0 < top_n < A
    if max_out is None:
        max_out = top_n if use_topn else min(A, 1024)
    dev = probs.device
    if out is not None:
        ob, op, oc, oi, cnt = out
        if tuple(ob.shape) != (n, max_out, 4) or tuple(op.shape) != (n, max_out) or tuple(cnt.shape) != (n,):
            raise _lib.SqdetError("filter_prediction: out tensors do not match [N,%d,...]" % max_out)
    else:
        ob = torch.empty((n, max_out, 4), dtype=torch.float32, device=dev)
        op = torch.empty((n, max_out), dtype=torch.float32, device=dev)
        oc = torch.empty((n, max_out), dtype=torch.int32, device=dev)
        oi = torch.empty((n, max_out), dtype=torch.int32, device=dev)
        cnt = torch.empty((n,), dtype=torch.int32, device=dev)
    check(lib().sqdet_filter_prediction(_dev(boxes, "boxes", torch.float32), _dev(probs, "probs", torch.float32),
                                        _dev(cls, "cls", torch.int64), _dev(ob, "ob"), _dev(op, "op"), _dev(oc, "oc"),
                                        _dev(oi, "oi"), _dev(cnt, "cnt"), n, A, int(classes), int(top_n), int(max_out),
                                        float(nms_thresh), float(prob_thresh), stream_ptr()), "sqdet_filter_prediction")
    return ob, op, oc, oi, cnt


def detect_filter_supported(num_anchors, top_n):
    return 0 < top_n <= 64 < num_anchors <= 20480


def detect_filter(preds, anchors_f32, classes, anchors_per_grid, img_w, img_h, exp_thresh, top_n, nms_thresh, scratch=None, out=None,
                  scores_ready=False, max_workgroups=0):
    """interpret_output + filter_prediction (top-N branch) in one call (sqdet_detect_filter): preds [N,gh,gw,K*(C+5)] ->
    the five filter_prediction outputs; det_boxes / det_class are never materialised.
    scores_ready: `scratch` already holds det_probs (written by the ConvDet epilogue: convdet / NetPlan.forward(scores=)) --
    sqdet_detect_filter_scored, the filter launch only."""
    n, gh, gw, ch = [int(v) for v in preds.shape]
    A = gh * gw * anchors_per_grid
    dev = preds.device
    scratch = torch.empty((n, A), dtype=torch.float32, device=dev) if scratch is None else scratch
    M = int(top_n)
    if out is not None:
        ob, op, oc, oi, cnt = out
    else:
        ob = torch.empty((n, M, 4), dtype=torch.float32, device=dev)
        op = torch.empty((n, M), dtype=torch.float32, device=dev)
        oc = torch.empty((n, M), dtype=torch.int32, device=dev)
        oi = torch.empty((n, M), dtype=torch.int32, device=dev)
        cnt = torch.empty((n,), dtype=torch.int32, device=dev)
    if scores_ready and (scratch is None or tuple(scratch.shape) != (n, A)):
        raise _lib.SqdetError("detect_filter: scores_ready needs the [N, A] float32 score tensor as `scratch`")
    args = (_dev(preds, "preds"), _dev(anchors_f32, "anchors", torch.float32), _dev(scratch, "scratch", torch.float32),
            _dev(ob, "ob"), _dev(op, "op"), _dev(oc, "oc"), _dev(oi, "oi"), _dev(cnt, "cnt"), n, gh, gw,
            int(anchors_per_grid), int(classes), float(img_w), float(img_h), float(exp_thresh), M, int(ob.shape[1]),
            float(nms_thresh), dtype_code(preds.dtype))
    if scores_ready:     # max_workgroups > 0: the images are walked by at most that many workgroups
        check(lib().sqdet_detect_filter_scored(*args, int(max_workgroups), stream_ptr()), "sqdet_detect_filter_scored")
    else:
        check(lib().sqdet_detect_filter(*args, stream_ptr()), "sqdet_detect_filter")
    return ob, op, oc, oi, cnt


def convdet_scores_supported(cin, anchors_per_grid, classes, dtype):
    return bool(lib().sqdet_convdet_scores_supported(int(cin), int(anchors_per_grid), int(classes), dtype_code(dtype)))


def convdet(x, packed, bias, anchors_per_grid, classes, preds=None, scores=None):
    """The ConvDet head + interpret_output's det_probs in one launch (sqdet_convdet_fwd): x [N,H,W,Cin] float16 ->
    (preds [N,H,W,K*(C+5)], scores [N, H*W*K] float32)."""
    n, h, w, cin = [int(v) for v in x.shape]
    cout = int(anchors_per_grid) * (int(classes) + 5)
    if packed.k != 3 or packed.cin != cin or packed.cout != cout or packed.dtype != x.dtype:
        raise _lib.SqdetError("convdet: packed kernel does not match x / the head's %d channels" % cout)
    preds = torch.empty((n, h, w, cout), dtype=x.dtype, device=x.device) if preds is None else preds
    scores = torch.empty((n, h * w * int(anchors_per_grid)), dtype=torch.float32, device=x.device) if scores is None else scores
    check(lib().sqdet_convdet_fwd(_dev(x, "x"), _dev(packed.data, "packed"), _dev(bias, "bias", torch.float32), _dev(preds, "preds"),
                                  _dev(scores, "scores", torch.float32), n, h, w, cin, int(anchors_per_grid), int(classes),
                                  dtype_code(x.dtype), stream_ptr()), "sqdet_convdet_fwd")
    return preds, scores


# ---------------------------------------------------------------- training (float32)
class PackedConvBwd:
    """rot180(W)^T in fragment order: the kernel of the backward-data conv (sqdet_conv_pack_weights_bwd_data)."""

    def __init__(self, w_hwio, dtype=torch.float32):
        w = w_hwio.detach().to(torch.float32).contiguous()
        self.k, _, self.cin, self.cout = [int(v) for v in w.shape]
        self.dtype = dtype
        code = dtype_code(dtype)
        nbytes = lib().sqdet_conv_packed_bytes(self.k, self.cout, self.cin, code)
        self.data = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        check(lib().sqdet_conv_pack_weights_bwd_data(_dev(w, "w_hwio"), _dev(self.data, "packed"), self.k, self.cin,
                                                     self.cout, code, stream_ptr()), "sqdet_conv_pack_weights_bwd_data")


class PackPlan:
    """Every trainable conv kernel of a trainer packed in ONE launch (sqdet_conv_pack_many): the forward fragment order
    (PackedConv) and, for the convs whose input needs a gradient, the backward-data conv's order (PackedConvBwd), into
    persistent buffers.  weights: {name: float32 HWIO tensor (a VIEW whose storage does not move: the trainer's flat
    parameter buffer)}; run() re-packs all of them from their current values.
    bn: {name: (gamma, beta, mean, var, conv_bias or None)} for the _conv_bn_layer convs among them -- their batch norm is
    folded on the way (sqdet_conv_pack_many_prepare_bn: bitwise fold_batchnorm + the packer) and self.bias[name] receives
    the folded bias."""

    def __init__(self, weights, dtype, bwd_names=(), bn=None, eps=0.0):
        self.dtype = dtype
        code = dtype_code(dtype)
        bn = bn or {}
        self.fwd, self.bwd, self.bias = {}, {}, {}
        items = []
        for name, w in weights.items():
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise _lib.SqdetError("PackPlan: %s must be a contiguous float32 HWIO tensor" % name)
            k, k2, cin, cout = [int(v) for v in w.shape]
            fold = bn.get(name)
            if fold is not None:
                for t in fold[:4]:
                    if t.dtype != torch.float32 or not t.is_contiguous() or int(t.numel()) != cout:
                        raise _lib.SqdetError("PackPlan: %s: batch-norm vectors must be contiguous float32 [cout]" % name)
                self.bias[name] = torch.empty(cout, dtype=torch.float32, device=w.device)
            pc = PackedConv.__new__(PackedConv)
            pc.k, pc.cin, pc.cout, pc.dtype = k, cin, cout, dtype
            pc.data = torch.empty(int(lib().sqdet_conv_packed_bytes(k, cin, cout, code)), dtype=torch.uint8, device=w.device)
            self.fwd[name] = pc
            items.append((w, pc.data, k, cin, cout, 0, fold, self.bias.get(name)))
            if name in bwd_names:
                pb = PackedConvBwd.__new__(PackedConvBwd)
                pb.k, pb.cin, pb.cout, pb.dtype = k, cin, cout, dtype
                pb.data = torch.empty(int(lib().sqdet_conv_packed_bytes(k, cout, cin, code)), dtype=torch.uint8, device=w.device)
                self.bwd[name] = pb
                items.append((w, pb.data, k, cin, cout, 1, fold, None))
        n = len(items)
        self.n = n
        self._keep = items
        dev = items[0][0].device
        arr = lambda vals, ct: (ct * n)(*vals)
        ptr = lambda t: t.data_ptr() if t is not None else None
        wp = arr([it[0].data_ptr() for it in items], C.c_void_p)
        op = arr([it[1].data_ptr() for it in items], C.c_void_p)
        ks, cis, cos, bw = [arr([it[j] for it in items], C.c_int) for j in (2, 3, 4, 5)]
        bnp = [arr([ptr(it[6][j]) if it[6] is not None else None for it in items], C.c_void_p) for j in range(5)]   # gamma beta mean var conv_bias
        bfp = arr([ptr(it[7]) for it in items], C.c_void_p)
        nbytes = int(lib().sqdet_conv_pack_many_table_bytes(n))
        host = (C.c_ubyte * nbytes)()
        blocks = C.c_int()
        check(lib().sqdet_conv_pack_many_prepare_bn(wp, op, ks, cis, cos, bw, bnp[0], bnp[1], bnp[2], bnp[3], bnp[4], bfp, float(eps), n,
                                                    code, host, C.byref(blocks)), "sqdet_conv_pack_many_prepare_bn")
        self.blocks = int(blocks.value)
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)

    def run(self):
        check(lib().sqdet_conv_pack_many(_dev(self.table, "table"), self.n, self.blocks, dtype_code(self.dtype), stream_ptr()),
              "sqdet_conv_pack_many")


class FoldBwdPlan:
    """sqdet_fold_batchnorm_bwd for MANY _conv_bn_layer convs in two launches.  items: [(w, dw_folded, db_folded, conv_bias or
    None, gamma, mean, var, dw, dgamma, dbeta)] -- persistent float32 tensors; run() after the folded gradients are in place."""

    def __init__(self, items, eps):
        self.n, self.eps, self._keep = len(items), float(eps), items
        dev = items[0][0].device
        dims = [[int(v) for v in it[0].shape] for it in items]
        self.ws = [torch.empty(int(lib().sqdet_fold_batchnorm_bwd_workspace_bytes(d[0], d[2], d[3])) // 4 + 16, dtype=torch.float32, device=dev)
                   for d in dims]
        arr = lambda vals, ct: (ct * self.n)(*vals)
        ptr = lambda t: t.data_ptr() if t is not None else None
        cols = [arr([ptr(it[j]) for it in items], C.c_void_p) for j in range(10)]
        wsp = arr([t.data_ptr() for t in self.ws], C.c_void_p)
        ks, cis, cos = [arr([d[j] for d in dims], C.c_int) for j in (0, 2, 3)]
        host = (C.c_ubyte * int(lib().sqdet_fold_batchnorm_bwd_many_table_bytes(self.n)))()
        b1, b2 = C.c_int(), C.c_int()
        check(lib().sqdet_fold_batchnorm_bwd_many_prepare(*cols, wsp, ks, cis, cos, self.n, host, C.byref(b1), C.byref(b2)),
              "sqdet_fold_batchnorm_bwd_many_prepare")
        self.blocks, self.finish_blocks = int(b1.value), int(b2.value)
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)

    def run(self):
        check(lib().sqdet_fold_batchnorm_bwd_many(_dev(self.table, "table"), self.n, self.blocks, self.finish_blocks, self.eps, stream_ptr()),
              "sqdet_fold_batchnorm_bwd_many")


def conv2d_bwd_data(dy, packed_bwd, dx=None, dy_coffset=0, accumulate=False, relu_of=None):
    """dx = conv(dy[..., dy_coffset:dy_coffset+cout], rot180(W)^T) (stride-1 SAME convs).  dy [N,H,W,Ctot].
    relu_of [N,H,W,cin]: the ReLU output dx is the gradient of -- dx (after the accumulation) is zeroed where it is <= 0
    (sqdet_conv2d_nhwc_bwd_data_relu: the ReLU backward of the layer below without a pass of its own)."""
    n, h, w, ctot = [int(v) for v in dy.shape]
    if dx is None:
        dx = torch.empty((n, h, w, packed_bwd.cin), dtype=dy.dtype, device=dy.device)
        accumulate = False
    if relu_of is not None:
        if tuple(relu_of.shape) != tuple(dx.shape) or relu_of.dtype != dx.dtype or not relu_of.is_contiguous():
            raise _lib.SqdetError("conv2d_bwd_data: relu_of must be a contiguous tensor shaped like dx")
        check(lib().sqdet_conv2d_nhwc_bwd_data_relu(_dev(dy, "dy"), _dev(packed_bwd.data, "packed"), _dev(dx, "dx"), _dev(relu_of, "relu_of"),
                                                    n, h, w, packed_bwd.cin, packed_bwd.cout, packed_bwd.k, dtype_code(dy.dtype), ctot,
                                                    int(dy_coffset), int(bool(accumulate)), stream_ptr()), "sqdet_conv2d_nhwc_bwd_data_relu")
        return dx
    check(lib().sqdet_conv2d_nhwc_bwd_data(_dev(dy, "dy"), _dev(packed_bwd.data, "packed"), _dev(dx, "dx"), n, h, w,
                                           packed_bwd.cin, packed_bwd.cout, packed_bwd.k, dtype_code(dy.dtype), ctot,
                                           int(dy_coffset), int(bool(accumulate)), stream_ptr()), "sqdet_conv2d_nhwc_bwd_data")
    return dx


def conv2d_bwd_filter(x, dy, k, cin, cout, w_for_decay=None, weight_decay=0.0, x_coffset=0, dy_coffset=0, want_bias=True,
                      dw=None, db=None, grad_scale=1.0):
    """dW [k,k,cin,cout] float32 HWIO = grad_scale * sum x*dy (+ weight_decay*W), dbias [cout] float32.
    x [N,H,W,Cx], dy [N,H,W,Cy] float32 or float16 (same dtype); the conv's input / output are the channel slices
    [x_coffset,+cin) / [dy_coffset,+cout)."""
    n, h, w, cx = [int(v) for v in x.shape]
    cy = int(dy.shape[3])
    dev = x.device
    if x.dtype != dy.dtype:
        raise _lib.SqdetError("conv2d_bwd_filter: x and dy must share a dtype")
    if dw is None:
        dw = torch.empty((k, k, cin, cout), dtype=torch.float32, device=dev)
    if db is None and want_bias:
        db = torch.empty((cout,), dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib().sqdet_conv2d_bwd_filter_workspace_bytes(n, h, w, cin, cout, k)) // 4 + 64, dtype=torch.float32, device=dev)
    check(lib().sqdet_conv2d_nhwc_bwd_filter(_dev(x, "x"), _dev(dy, "dy"), _dev(dw, "dw", torch.float32),
                                             _dev(db, "db", torch.float32) if db is not None else None,
                                             _dev(w_for_decay, "w", torch.float32) if w_for_decay is not None else None,
                                             float(weight_decay), float(grad_scale), _dev(ws, "ws"), n, h, w, int(cin), int(cout),
                                             int(k), cx, int(x_coffset), cy, int(dy_coffset), dtype_code(x.dtype), stream_ptr()),
          "sqdet_conv2d_nhwc_bwd_filter")
    return dw, db


class WgradPlan:
    """The weight gradients of a whole backward pass with ONE slab reduction (sqdet_conv2d_nhwc_bwd_filter_partial per conv
    into a workspace of its own, sqdet_slab_reduce_many at the end) instead of a reduction launch behind every gradient
    kernel.  items: [(key, (n, h, w, cin, cout, k), dw, db or None, w_for_decay or None, weight_decay)] -- dw / db are
    persistent float32 tensors (a trainer's flat gradient views).  Per conv the results are conv2d_bwd_filter's, bitwise."""

    def __init__(self, items):
        self.n = len(items)
        self.shape, self.ws, self.has_bias, self._keep = {}, {}, {}, items
        dev = items[0][2].device
        for key, shp, dw, db, wd, decay in items:
            n, h, w, cin, cout, k = [int(v) for v in shp]
            if tuple(dw.shape) != (k, k, cin, cout) or dw.dtype != torch.float32 or not dw.is_contiguous():
                raise _lib.SqdetError("WgradPlan: %s: dw must be a contiguous float32 [k,k,cin,cout] tensor" % (key,))
            self.shape[key] = (n, h, w, cin, cout, k)
            self.has_bias[key] = db is not None
            self.ws[key] = torch.empty(int(lib().sqdet_conv2d_bwd_filter_workspace_bytes(n, h, w, cin, cout, k)) // 4 + 64,
                                       dtype=torch.float32, device=dev)
        arr = lambda vals, ct: (ct * self.n)(*vals)
        ptr = lambda t: t.data_ptr() if t is not None else None
        wsp = arr([self.ws[it[0]].data_ptr() for it in items], C.c_void_p)
        dwp = arr([it[2].data_ptr() for it in items], C.c_void_p)
        dbp = arr([ptr(it[3]) for it in items], C.c_void_p)
        wdp = arr([ptr(it[4]) for it in items], C.c_void_p)
        dec = arr([float(it[5]) for it in items], C.c_float)
        dims = [arr([int(it[1][j]) for it in items], C.c_int) for j in range(6)]
        host = (C.c_ubyte * int(lib().sqdet_slab_reduce_many_table_bytes(self.n)))()
        blocks = C.c_int()
        check(lib().sqdet_slab_reduce_many_prepare(wsp, dwp, dbp, wdp, dec, *dims, self.n, host, C.byref(blocks)), "sqdet_slab_reduce_many_prepare")
        self.blocks = int(blocks.value)
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)

    def partial(self, key, x, dy, x_coffset=0, dy_coffset=0):
        """The gradient kernel of conv `key` on (x, dy): its partial slabs only."""
        n, h, w, cin, cout, k = self.shape[key]
        if tuple(x.shape[:3]) != (n, h, w) or tuple(dy.shape[:3]) != (n, h, w) or x.dtype != dy.dtype:
            raise _lib.SqdetError("WgradPlan.partial: %s: tensors do not match the planned shape" % (key,))
        check(lib().sqdet_conv2d_nhwc_bwd_filter_partial(_dev(x, "x"), _dev(dy, "dy"), _dev(self.ws[key], "ws"), int(self.has_bias[key]),
                                                         n, h, w, cin, cout, k, int(x.shape[3]), int(x_coffset), int(dy.shape[3]),
                                                         int(dy_coffset), dtype_code(x.dtype), stream_ptr()), "sqdet_conv2d_nhwc_bwd_filter_partial")

    def reduce(self, grad_scale=1.0):
        """dw / db of every item from the slabs its partial() wrote."""
        check(lib().sqdet_slab_reduce_many(_dev(self.table, "table"), self.n, self.blocks, float(grad_scale), stream_ptr()), "sqdet_slab_reduce_many")


def fold_batchnorm_bwd(w_hwio, dw_folded, db_folded, conv_bias, gamma, mean, var, eps, dw=None, dgamma=None, dbeta=None):
    """Gradients of kernels / gamma / beta of a _conv_bn_layer from the gradients of its folded kernel / bias
    (sqdet_fold_batchnorm_bwd).  Returns (dw, dgamma, dbeta); dw may be dw_folded itself (in place)."""
    k, _, cin, cout = [int(v) for v in w_hwio.shape]
    dev = w_hwio.device
    dw = torch.empty_like(dw_folded) if dw is None else dw
    dgamma = torch.empty(cout, dtype=torch.float32, device=dev) if dgamma is None else dgamma
    dbeta = torch.empty(cout, dtype=torch.float32, device=dev) if dbeta is None else dbeta
    ws = torch.empty(int(lib().sqdet_fold_batchnorm_bwd_workspace_bytes(k, cin, cout)) // 4 + 16, dtype=torch.float32, device=dev)
    check(lib().sqdet_fold_batchnorm_bwd(_dev(w_hwio, "w", torch.float32), _dev(dw_folded, "dw_folded", torch.float32),
                                         _dev(db_folded, "db_folded", torch.float32),
                                         _dev(conv_bias, "conv_bias", torch.float32) if conv_bias is not None else None,
                                         _dev(gamma, "gamma", torch.float32), _dev(mean, "mean", torch.float32),
                                         _dev(var, "var", torch.float32), float(eps), _dev(dw, "dw", torch.float32),
                                         _dev(dgamma, "dgamma", torch.float32), _dev(dbeta, "dbeta", torch.float32),
                                         _dev(ws, "ws"), k, cin, cout, stream_ptr()), "sqdet_fold_batchnorm_bwd")
    return dw, dgamma, dbeta


def subsample_nhwc(x, stride):
    """x[:, ::stride, ::stride, :] as a dense tensor (sqdet_subsample_nhwc)."""
    n, h, w, c = [int(v) for v in x.shape]
    y = torch.empty((n, -(-h // stride), -(-w // stride), c), dtype=x.dtype, device=x.device)
    check(lib().sqdet_subsample_nhwc(_dev(x, "x"), _dev(y, "y"), n, h, w, c, int(stride), dtype_code(x.dtype), stream_ptr()),
          "sqdet_subsample_nhwc")
    return y


def relu_bwd(y, dy):
    """In place: dy *= (y > 0).  float32 or float16."""
    check(lib().sqdet_relu_bwd(_dev(y, "y", dy.dtype), _dev(dy, "dy"), y.numel(), dtype_code(dy.dtype), stream_ptr()), "sqdet_relu_bwd")
    return dy


def scale_mask(x, mask, scale, relu_of=None):
    """x * mask * scale (tf.nn.dropout and its backward); relu_of: also zeroed where that ReLU output is <= 0 (the ReLU
    backward of the layer x is the gradient of, sqdet_scale_mask_relu)."""
    y = torch.empty_like(x)
    if relu_of is not None:
        if tuple(relu_of.shape) != tuple(x.shape):
            raise _lib.SqdetError("scale_mask: relu_of must be shaped like x")
        check(lib().sqdet_scale_mask_relu(_dev(x, "x"), _dev(mask, "mask", x.dtype), _dev(relu_of, "relu_of", x.dtype), _dev(y, "y"),
                                          float(scale), x.numel(), dtype_code(x.dtype), stream_ptr()), "sqdet_scale_mask_relu")
        return y
    check(lib().sqdet_scale_mask(_dev(x, "x"), _dev(mask, "mask", x.dtype), _dev(y, "y"), float(scale), x.numel(),
                                 dtype_code(x.dtype), stream_ptr()), "sqdet_scale_mask")
    return y


def convert_scale(x, dtype, scale=1.0):
    """(dtype)(x * scale) on the device: float16 <-> float32 (sqdet_convert_scale)."""
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().sqdet_convert_scale(_dev(x, "x"), dtype_code(x.dtype), _dev(y, "y"), dtype_code(dtype), float(scale), x.numel(),
                                    stream_ptr()), "sqdet_convert_scale")
    return y


def maxpool_bwd(x, dy, size, stride, padding="SAME", relu=False):
    """tf.nn.max_pool's gradient; relu=True: x is a ReLU output and dx is also zeroed where x <= 0 (the ReLU backward of
    the layer below, sqdet_maxpool_nhwc_bwd_relu)."""
    n, h, w, c = [int(v) for v in x.shape]
    dx = torch.empty_like(x)
    fn = lib().sqdet_maxpool_nhwc_bwd_relu if relu else lib().sqdet_maxpool_nhwc_bwd
    check(fn(_dev(x, "x"), _dev(dy, "dy", x.dtype), _dev(dx, "dx"), n, h, w, c, int(size), int(stride),
             pad_code(padding), dtype_code(x.dtype), stream_ptr()), "sqdet_maxpool_nhwc_bwd")
    return dx


def maxpool_bwd_idx(idx, y, dy, in_hw, size, stride, padding="SAME", relu=False):
    """maxpool_bwd from the forward's window index (maxpool_nhwc_idx) and the pooled y; in_hw = (h, w) of the pool's input."""
    n, ho, wo, c = [int(v) for v in dy.shape]
    h, w = int(in_hw[0]), int(in_hw[1])
    dx = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
    check(lib().sqdet_maxpool_nhwc_bwd_idx(_dev(idx, "idx", torch.uint8), _dev(y, "y", dy.dtype), _dev(dy, "dy"), _dev(dx, "dx"),
                                           n, h, w, c, int(size), int(stride), pad_code(padding), dtype_code(dy.dtype),
                                           1 if relu else 0, stream_ptr()), "sqdet_maxpool_nhwc_bwd_idx")
    return dx


def sum_f32(x, out=None):
    """Deterministic sum of a float32 tensor into a device scalar (sqdet_sum_f32): num_objects = sum(input_mask)."""
    out = torch.empty(1, dtype=torch.float32, device=x.device) if out is None else out
    check(lib().sqdet_sum_f32(_dev(x, "x", torch.float32), int(x.numel()), _dev(out, "out", torch.float32), stream_ptr()), "sqdet_sum_f32")
    return out


def add_relu(a, b, out=None):
    """max(a + b, 0) (sqdet_add_relu)."""
    if a.shape != b.shape or a.dtype != b.dtype:
        raise _lib.SqdetError("add_relu: operands differ in shape / dtype")
    out = torch.empty_like(a) if out is None else out
    check(lib().sqdet_add_relu(_dev(a, "a"), _dev(b, "b"), _dev(out, "out"), int(a.numel()), dtype_code(a.dtype), stream_ptr()), "sqdet_add_relu")
    return out


def copy_channels(x, out, out_coffset):
    """out[..., out_coffset : out_coffset + C] = x (sqdet_copy_channels): one input of a channel concat."""
    c = int(x.shape[-1])
    check(lib().sqdet_copy_channels(_dev(x, "x"), _dev(out, "out"), int(x.numel() // c), c, int(out.shape[-1]), int(out_coffset),
                                    dtype_code(x.dtype), stream_ptr()), "sqdet_copy_channels")
    return out


def dropout_mask(shape, keep_prob, seed, dtype, device):
    """floor(keep_prob + U[0,1)) per element from a counter-based generator (sqdet_dropout_mask)."""
    m = torch.empty(shape, dtype=dtype, device=device)
    check(lib().sqdet_dropout_mask(_dev(m, "mask"), int(m.numel()), float(keep_prob), int(seed) & (2 ** 64 - 1), dtype_code(dtype),
                                   stream_ptr()), "sqdet_dropout_mask")
    return m


def copy_to_pinned_host(src, dst_pinned):
    """src (device, contiguous) -> dst_pinned (a pin_memory() host tensor of the same byte size) by a kernel launch on the
    current stream (sqdet_copy_to_mapped_host): never blocks the host thread."""
    if not dst_pinned.is_pinned() or dst_pinned.numel() * dst_pinned.element_size() != src.numel() * src.element_size():
        raise _lib.SqdetError("copy_to_pinned_host: dst must be pinned host memory of the same size")
    check(lib().sqdet_copy_to_mapped_host(_dev(src, "src"), C.c_void_p(dst_pinned.data_ptr()), int(src.numel() * src.element_size()),
                                          stream_ptr()), "sqdet_copy_to_mapped_host")
    return dst_pinned


def dropout_mask_into(mask, keep_prob, seed):
    """Refills an existing mask tensor (static buffer of a captured step)."""
    check(lib().sqdet_dropout_mask(_dev(mask, "mask"), int(mask.numel()), float(keep_prob), int(seed) & (2 ** 64 - 1),
                                   dtype_code(mask.dtype), stream_ptr()), "sqdet_dropout_mask")
    return mask


def loss_fwd_bwd_mixed(preds, anchors_f32, input_mask, box_delta_input, box_input, labels, mc, num_objects, loss_scale, global_batch=0):
    """loss_fwd_bwd on FLOAT16 preds (sqdet_loss_fwd_bwd_mixed): also returns the loss-scaled float16 gradient the float16
    backward starts from -- bitwise convert_scale(dpreds, float16, loss_scale) -- written in the same pass.
    Returns (g16, dpreds float32, ious, losses)."""
    n, gh, gw, ch = [int(v) for v in preds.shape]
    A = gh * gw * mc.ANCHOR_PER_GRID
    dev = preds.device
    dpreds = torch.empty(preds.shape, dtype=torch.float32, device=dev)
    g16 = torch.empty_like(preds)
    ious = torch.empty((n, A), dtype=torch.float32, device=dev)
    losses = torch.empty((3,), dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib().sqdet_loss_workspace_bytes()) // 4 + 16, dtype=torch.float32, device=dev)
    on_dev = isinstance(num_objects, torch.Tensor)
    check(lib().sqdet_loss_fwd_bwd_mixed(_dev(preds, "preds", torch.float16), _dev(anchors_f32, "anchors", torch.float32),
                                         _dev(input_mask, "mask", torch.float32), _dev(box_delta_input, "delta", torch.float32),
                                         _dev(box_input, "box", torch.float32), _dev(labels, "labels", torch.float32),
                                         _dev(dpreds, "dpreds"), _dev(g16, "g16"), float(loss_scale), _dev(ious, "ious"),
                                         _dev(losses, "losses"), _dev(ws, "ws"), n, gh, gw, int(mc.ANCHOR_PER_GRID), int(mc.CLASSES),
                                         float(mc.IMAGE_WIDTH), float(mc.IMAGE_HEIGHT), float(mc.EXP_THRESH), float(mc.EPSILON),
                                         float(mc.LOSS_COEF_CLASS), float(mc.LOSS_COEF_CONF_POS), float(mc.LOSS_COEF_CONF_NEG),
                                         float(mc.LOSS_COEF_BBOX), 1.0 if on_dev else float(num_objects),
                                         _dev(num_objects, "num_objects", torch.float32) if on_dev else None, int(global_batch),
                                         stream_ptr()), "sqdet_loss_fwd_bwd_mixed")
    return g16, dpreds, ious, losses


def loss_fwd_bwd(preds, anchors_f32, input_mask, box_delta_input, box_input, labels, mc, num_objects, global_batch=0):
    """ModelSkeleton._add_loss_graph forward + backward (nn_skeleton.py:285-327).  All tensors float32 on
    the device.  num_objects: a Python number, or a float32 DEVICE scalar (no host round trip: hipGraph-capturable).
    global_batch: divisor of the confidence term's mean over the batch when this call is one replica's share of a larger
    batch (0 = this call's batch).  Returns (dpreds, ious [B,A], losses [3] = class, conf, bbox)."""
    n, gh, gw, ch = [int(v) for v in preds.shape]
    A = gh * gw * mc.ANCHOR_PER_GRID
    dev = preds.device
    dpreds = torch.empty_like(preds)
    ious = torch.empty((n, A), dtype=torch.float32, device=dev)
    losses = torch.empty((3,), dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib().sqdet_loss_workspace_bytes()) // 4 + 16, dtype=torch.float32, device=dev)
    if isinstance(num_objects, torch.Tensor):
        check(lib().sqdet_loss_fwd_bwd_dev(_dev(preds, "preds", torch.float32), _dev(anchors_f32, "anchors", torch.float32),
                                           _dev(input_mask, "mask", torch.float32), _dev(box_delta_input, "delta", torch.float32),
                                           _dev(box_input, "box", torch.float32), _dev(labels, "labels", torch.float32),
                                           _dev(dpreds, "dpreds"), _dev(ious, "ious"), _dev(losses, "losses"), _dev(ws, "ws"),
                                           n, gh, gw, int(mc.ANCHOR_PER_GRID), int(mc.CLASSES), float(mc.IMAGE_WIDTH),
                                           float(mc.IMAGE_HEIGHT), float(mc.EXP_THRESH), float(mc.EPSILON), float(mc.LOSS_COEF_CLASS),
                                           float(mc.LOSS_COEF_CONF_POS), float(mc.LOSS_COEF_CONF_NEG), float(mc.LOSS_COEF_BBOX),
                                           _dev(num_objects, "num_objects", torch.float32), int(global_batch), stream_ptr()), "sqdet_loss_fwd_bwd_dev")
        return dpreds, ious, losses
    check(lib().sqdet_loss_fwd_bwd(_dev(preds, "preds", torch.float32), _dev(anchors_f32, "anchors", torch.float32),
                                   _dev(input_mask, "mask", torch.float32), _dev(box_delta_input, "delta", torch.float32),
                                   _dev(box_input, "box", torch.float32), _dev(labels, "labels", torch.float32),
                                   _dev(dpreds, "dpreds"), _dev(ious, "ious"), _dev(losses, "losses"), _dev(ws, "ws"),
                                   n, gh, gw, int(mc.ANCHOR_PER_GRID), int(mc.CLASSES), float(mc.IMAGE_WIDTH),
                                   float(mc.IMAGE_HEIGHT), float(mc.EXP_THRESH), float(mc.EPSILON), float(mc.LOSS_COEF_CLASS),
                                   float(mc.LOSS_COEF_CONF_POS), float(mc.LOSS_COEF_CONF_NEG), float(mc.LOSS_COEF_BBOX),
                                   float(num_objects), int(global_batch), stream_ptr()), "sqdet_loss_fwd_bwd")
    return dpreds, ious, losses


class MomentumOptimizer:
    """sqdet_optimizer_*: Momentum + per-variable clip_by_norm (+ weight decay) over flat buffers."""

    def __init__(self, offsets, counts, decays, device):
        nv = len(offsets)
        self._h = C.c_void_p()
        off = (C.c_long * nv)(*[int(v) for v in offsets])
        cnt = (C.c_long * nv)(*[int(v) for v in counts])
        dec = (C.c_float * nv)(*[float(v) for v in decays])
        check(lib().sqdet_optimizer_create(C.byref(self._h), off, cnt, dec, nv), "sqdet_optimizer_create")
        self.ws = torch.empty(int(lib().sqdet_optimizer_workspace_bytes(self._h)) + 256, dtype=torch.uint8, device=device)

    def step(self, params, grads, accum, lr, momentum, max_grad_norm, grad_scale=1.0, found_inf=None):
        """found_inf: optional int32 device tensor [1]; set to 1 (and the step skipped) when a gradient norm is inf / NaN."""
        check(lib().sqdet_optimizer_step(self._h, _dev(params, "params", torch.float32), _dev(grads, "grads", torch.float32),
                                         _dev(accum, "accum", torch.float32), _dev(self.ws, "ws"), float(lr), float(momentum),
                                         float(max_grad_norm), float(grad_scale),
                                         _dev(found_inf, "found_inf", torch.int32) if found_inf is not None else None,
                                         stream_ptr()), "sqdet_optimizer_step")

    def __del__(self):
        try:
            if self._h:
                lib().sqdet_optimizer_destroy(self._h)
                self._h = None
        except Exception:
            pass


def build_labels(anchor_box, gt_boxes, gt_classes, gt_counts, classes, device=None):
    """Anchor assignment + dense label tensors on the GPU (imdb.py:195-239, train.py:163-224).
    anchor_box: mc.ANCHOR_BOX [A,4] float64; gt_boxes [B,M,4] float64 (cx,cy,w,h in network-input pixels, rows
    beyond gt_counts[b] ignored); gt_classes [B,M]; gt_counts [B].  Returns (input_mask [B,A], box_delta_input
    [B,A,4], box_input [B,A,4], labels [B,A,C]) float32 and anchor_index [B,M] int32, all on the device."""
    dev = torch.device(device) if device is not None else (gt_boxes.device if isinstance(gt_boxes, torch.Tensor) else torch.device("cuda", torch.cuda.current_device()))
    to = lambda v, dt: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).to(dev, dt).contiguous()
    anc, gt = to(anchor_box, torch.float64), to(gt_boxes, torch.float64)
    cls, cnt = to(gt_classes, torch.int32), to(gt_counts, torch.int32)
    B, M = int(gt.shape[0]), int(gt.shape[1])
    A = int(anc.shape[0])
    mask = torch.empty((B, A), dtype=torch.float32, device=dev)
    delta = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    box = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    lab = torch.empty((B, A, int(classes)), dtype=torch.float32, device=dev)
    aidx = torch.empty((B, M), dtype=torch.int32, device=dev)
    check(lib().sqdet_build_labels(_dev(anc, "anchors"), _dev(gt, "gt_boxes"), _dev(cls, "gt_classes"), _dev(cnt, "gt_counts"),
                                   _dev(mask, "mask"), _dev(delta, "delta"), _dev(box, "box"), _dev(lab, "labels"),
                                   _dev(aidx, "aidx"), B, A, M, int(classes), stream_ptr()), "sqdet_build_labels")
    return mask, delta, box, lab, aidx


def preprocess_bgr(images_u8, dst_h, dst_w, bgr_means, dtype=torch.float32):
    """uint8 BGR [N,H,W,3] (device) -> resized (cv2 INTER_LINEAR), mean-subtracted NHWC network input
    (demo.py:186-190) in `dtype`."""
    n, h, w, c = [int(v) for v in images_u8.shape]
    if c != 3 or images_u8.dtype != torch.uint8:
        raise _lib.SqdetError("preprocess_bgr: expected uint8 [N,H,W,3]")
    out = torch.empty((n, int(dst_h), int(dst_w), 3), dtype=dtype, device=images_u8.device)
    m = [float(v) for v in np.asarray(bgr_means).reshape(-1)[:3]]
    check(lib().sqdet_preprocess_bgr(_dev(images_u8, "images"), _dev(out, "out"), n, h, w, int(dst_h), int(dst_w), m[0], m[1], m[2],
                                     dtype_code(dtype), stream_ptr()), "sqdet_preprocess_bgr")
    return out


def box_calibration(device, mfma_iters=8192, copy_mib=1024, reps=3):
    """Two fixed microkernels timed with events on the current stream (sqdet_calib_mfma / sqdet_calib_copy): what THIS box
    sustains on a bare MFMA loop (TFLOP/s, float16 16x16x32) and on a plain device copy far larger than the Infinity Cache
    (GB/s, read + write).  bench.py prints both beside its result so that a slow box is identifiable from the JSON line alone.
    Best of `reps`."""
    scratch = torch.empty(512 * 256, dtype=torch.float32, device=device)
    n = copy_mib << 20
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    src.fill_(1)
    flops = C.c_double(0.0)
    best_m, best_c = None, None
    for r in range(reps + 1):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        check(lib().sqdet_calib_mfma(_dev(scratch, "scratch"), scratch.numel(), int(mfma_iters), C.byref(flops), stream_ptr()), "sqdet_calib_mfma")
        e1.record()
        check(lib().sqdet_calib_copy(_dev(src, "src"), _dev(dst, "dst"), n, stream_ptr()), "sqdet_calib_copy")
        e2.record()
        torch.cuda.synchronize()
        if r == 0:
            continue                                   # warm-up
        tm, tc = e0.elapsed_time(e1), e1.elapsed_time(e2)
        best_m = tm if best_m is None else min(best_m, tm)
        best_c = tc if best_c is None else min(best_c, tc)
    res = {"box_mfma_tflops": round(flops.value / (best_m * 1e-3) / 1e12, 1), "box_mfma_ms": round(best_m, 4),
           "box_copy_gbs": round(2.0 * n / (best_c * 1e-3) / 1e9, 1), "box_copy_ms": round(best_c, 4)}
    # round 5: what IS the box's MFMA ceiling -- 32x32x16 at four waves per SIMD (the guide's 2495 TF/s form) on non-trivial and
    # on all-zero operands, with the effective shader clock and the matrix pipe's busy fraction from in-kernel counters
    try:
        v = calib_mfma_variant(device, 1, 4, False, reps=reps)
        z = calib_mfma_variant(device, 1, 4, True, reps=reps)
        res.update(box_mfma_tflops_32x32=v["tflops"], effective_clock_mhz=v["effective_clock_mhz"], mfma_busy_frac_32x32=v["mfma_busy_frac"],
                   box_mfma_tflops_32x32_zero_operands=z["tflops"], effective_clock_mhz_zero_operands=z["effective_clock_mhz"])
    except _lib.SqdetError as e:      # (never voids the two figures above)
        res["box_mfma_32x32_error"] = str(e)[:120]
    return res


# pipe cycles one MFMA occupies its SIMD's matrix unit for (MI355X_MICROARCH.md, per-instruction cycle constants: 32x32x16 f16 issues
# back to back at 32 cycles per SIMD = SQ_VALU_MFMA_BUSY_CYCLES per instruction; 16x16x32 at half that)
MFMA_PIPE_CYCLES = {0: 16.0, 1: 32.0}


def calib_mfma_variant(device, shape, waves_per_simd, zero_operands, iters=None, reps=3):
    """sqdet_calib_mfma2 timed with events: {tflops, ms, effective_clock_mhz, cycles_per_mfma_per_simd, mfma_busy_frac, workgroups}.
    shape 0 = 16x16x32, 1 = 32x32x16 (float16); `waves_per_simd` co-resident workgroups per CU; zero_operands: all-zero A / B.
    effective_clock_mhz = 100 MHz x (s_memtime cycles / s_memrealtime ticks), median over the workgroups; cycles_per_mfma_per_simd =
    a wave's cycles / its MFMAs / waves_per_simd (the SIMD's issue interval); mfma_busy_frac = MFMA_PIPE_CYCLES / that."""
    import numpy as np
    nacc = 8 if shape == 0 else 4
    if iters is None:
        iters = (8192 * 2 // waves_per_simd) if shape == 0 else (8192 // waves_per_simd)       # ~2 ms at the guide's rate
    # (the kernel's grid is cu_count() x waves_per_simd workgroups: sized from THIS device's compute units, not an MI355X's 256)
    cap = max(256, int(torch.cuda.get_device_properties(device).multi_processor_count)) * 8 * 8
    scratch = torch.empty(cap * 256 // 8, dtype=torch.float32, device=device)
    ticks = torch.zeros(cap * 2 // 8 * 8, dtype=torch.int64, device=device)
    flops, wgs = C.c_double(0.0), C.c_int(0)
    best = None
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().sqdet_calib_mfma2(_dev(scratch, "scratch"), scratch.numel(), _dev(ticks, "ticks"), ticks.numel(), int(iters), int(shape),
                                      int(waves_per_simd), int(bool(zero_operands)), C.byref(flops), C.byref(wgs), stream_ptr()), "sqdet_calib_mfma2")
        e1.record()
        torch.cuda.synchronize()
        if r == 0:
            continue
        ms = e0.elapsed_time(e1)
        if best is None or ms < best[0]:
            t = ticks[:2 * wgs.value].cpu().numpy().reshape(-1, 2).astype(np.float64)
            best = (ms, float(np.median(t[:, 0])), float(np.median(t[:, 1])))
    ms, cyc, wall = best
    clock_mhz = 100.0 * cyc / max(wall, 1.0)
    # the SIMD's issue interval: kernel duration x effective clock / MFMAs per SIMD (workgroups are not necessarily spread evenly over
    # the CUs, so a single wave's cycle count is not the kernel's; the launch-level figure is what rocprofv3's
    # SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE gives too: profiles/r05_box_calibration.txt)
    simds = (wgs.value // waves_per_simd) * 4
    per_simd = (ms * 1e-3 * clock_mhz * 1e6) / (wgs.value * 4.0 * iters * nacc / simds)
    return {"shape": "16x16x32" if shape == 0 else "32x32x16", "waves_per_simd": waves_per_simd, "zero_operands": bool(zero_operands),
            "tflops": round(flops.value / (ms * 1e-3) / 1e12, 1), "ms": round(ms, 4), "workgroups": wgs.value,
            "effective_clock_mhz": round(clock_mhz, 1), "cycles_per_mfma_per_simd": round(per_simd, 2),
            "mfma_busy_frac": round(MFMA_PIPE_CYCLES[shape] / per_simd, 4)}


def set_option(name, value):
    """Process-wide tuning knob (sqdet_set_option), e.g. set_option("conv_algo", 1) = generic kernels only."""
    check(lib().sqdet_set_option(name.encode(), int(value)), "sqdet_set_option")


def probe_mfma_layout():
    """[2 shapes][64 lanes][4 regs][row, col] observed accumulator layout."""
    import numpy as np
    buf = (C.c_int32 * 1024)()
    check(lib().sqdet_probe_mfma_layout(buf, 1024), "sqdet_probe_mfma_layout")
    return np.frombuffer(buf, dtype=np.int32).reshape(2, 64, 4, 2).copy()


# ---------------------------------------------------------------- compiled network plan
class NetPlan:
    """sqdet_net_*: the whole forward graph (nets/squeezeDet.py:30-79 /
    nets/squeezeDetPlus.py:30-79) as one native plan; device memory is torch-allocated."""

    ARCH = {"squeezeDet": _lib.ARCH_SQUEEZEDET, "squeezeDet+": _lib.ARCH_SQUEEZEDET_PLUS,
            "resnet50": _lib.ARCH_RESNET50}

    def set_bn_epsilon(self, eps):
        check(lib().sqdet_net_set_bn_epsilon(self._h, float(eps)), "sqdet_net_set_bn_epsilon")

    def __init__(self, arch, dtype, batch, img_h, img_w, classes, anchors_per_grid, device):
        self.arch, self.dtype, self.batch, self.img_h, self.img_w = arch, dtype, batch, img_h, img_w
        self.classes, self.anchors_per_grid = int(classes), int(anchors_per_grid)
        self.device = torch.device(device)
        self._h = C.c_void_p()
        check(lib().sqdet_net_create(C.byref(self._h), self.ARCH[arch], dtype_code(dtype), batch, img_h, img_w,
                                     classes, anchors_per_grid), "sqdet_net_create")
        gh, gw, ch = C.c_int(), C.c_int(), C.c_int()
        check(lib().sqdet_net_output_dims(self._h, C.byref(gh), C.byref(gw), C.byref(ch)))
        self.gh, self.gw, self.out_ch = gh.value, gw.value, ch.value
        self.param_mem = torch.zeros(max(int(lib().sqdet_net_param_bytes(self._h)), 256), dtype=torch.uint8, device=self.device)
        self.workspace = torch.empty(max(int(lib().sqdet_net_workspace_bytes(self._h)), 256), dtype=torch.uint8, device=self.device)
        check(lib().sqdet_net_bind(self._h, _dev(self.param_mem, "param_mem"), _dev(self.workspace, "workspace")),
              "sqdet_net_bind")

    def __del__(self):
        try:
            if self._h:
                lib().sqdet_net_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def param_specs(self):
        out = []
        name = C.create_string_buffer(128)
        shape = (C.c_int * 4)()
        nd = C.c_int()
        for i in range(lib().sqdet_net_num_params(self._h)):
            check(lib().sqdet_net_param_info(self._h, i, name, 128, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def set_param(self, name, value):
        v = value.detach().to(device=self.device, dtype=torch.float32).contiguous()
        check(lib().sqdet_net_set_param(self._h, name.encode(), _dev(v, name), stream_ptr()), "sqdet_net_set_param(%s)" % name)
        torch.cuda.current_stream().synchronize()  # `v` may be a temporary

    def overlap_layer(self):
        """Index of the launch beside which side work is cheapest (first fire_chain launch), or -1."""
        return int(lib().sqdet_net_overlap_layer(self._h))

    def set_signal(self, layer_index, event):
        """Every following forward records `event` (a torch.cuda.Event that has been recorded once, so that its handle
        exists; None: no signal) right before the launch of layer `layer_index` (sqdet_net_set_signal)."""
        h = None if event is None else C.c_void_p(event.cuda_event)
        check(lib().sqdet_net_set_signal(self._h, int(layer_index), h), "sqdet_net_set_signal")

    def rider_capacity(self):
        """Images the plan's fire_chain launches can carry as riders (sqdet_net_rider_capacity); 0 = none."""
        return int(lib().sqdet_net_rider_capacity(self._h))

    def set_post_job(self, preds, scores, anchors_f32, out, classes, anchors_per_grid, img_w, img_h, exp_thresh, top_n, nms_thresh):
        """The decode + filter of a PREVIOUS batch (its preds / scores; out = the five output tensors of filter_prediction,
        device or pinned host) rides in the next forward's fire_chain launches (sqdet_net_set_post_job).  One-shot."""
        n, gh, gw, ch = [int(v) for v in preds.shape]
        ob, op, oc, oi, cnt = out
        ptr = lambda t: C.c_void_p(t.data_ptr())          # (pinned host tensors are device-accessible at the same address)
        for t in out:
            if not (t.is_cuda or t.is_pinned()):
                raise _lib.SqdetError("set_post_job: outputs must be device or pinned host tensors")
        check(lib().sqdet_net_set_post_job(self._h, _dev(preds, "preds"), _dev(scores, "scores", torch.float32),
                                           _dev(anchors_f32, "anchors", torch.float32), ptr(ob), ptr(op), ptr(oc), ptr(oi), ptr(cnt),
                                           n, gh, gw, int(anchors_per_grid), int(classes), float(img_w), float(img_h), float(exp_thresh),
                                           int(top_n), int(ob.shape[1]), float(nms_thresh), dtype_code(preds.dtype)), "sqdet_net_set_post_job")

    def scores_supported(self):
        """The plan's ConvDet launch can also write interpret_output's det_probs (float16 SqueezeDet-style head)."""
        return bool(lib().sqdet_net_scores_supported(self._h))

    def forward(self, image_input, preds=None, scores=None):
        """scores: float32 [batch, A] -- when given, the ConvDet launch's epilogue also writes det_probs there
        (sqdet_net_set_scores); follow with detect_filter(..., scratch=scores, scores_ready=True)."""
        exp = (self.batch, self.img_h, self.img_w, 3)
        if tuple(image_input.shape) != exp or image_input.dtype != self.dtype:
            raise _lib.SqdetError("forward: image_input must be %s %s, got %s %s"
                                  % (exp, self.dtype, tuple(image_input.shape), image_input.dtype))
        if preds is None:
            preds = torch.empty((self.batch, self.gh, self.gw, self.out_ch), dtype=self.dtype, device=self.device)
        if scores is not None:
            if scores.numel() * self.out_ch != preds.numel() * self.anchors_per_grid or scores.dtype != torch.float32:
                raise _lib.SqdetError("forward: scores must be float32 [batch, gh*gw*anchors_per_grid]")
            check(lib().sqdet_net_set_scores(self._h, _dev(scores, "scores", torch.float32)), "sqdet_net_set_scores")
        try:
            check(lib().sqdet_net_forward(self._h, _dev(image_input, "image_input"), _dev(preds, "preds"), stream_ptr()),
                  "sqdet_net_forward")
        finally:
            if scores is not None:
                lib().sqdet_net_set_scores(self._h, None)
        return preds

    def layer_table(self):
        out = []
        name = C.create_string_buffer(128)
        fl, by = C.c_double(), C.c_double()
        for i in range(lib().sqdet_net_num_layers(self._h)):
            check(lib().sqdet_net_layer_info(self._h, i, name, 128, C.byref(fl), C.byref(by)))
            out.append((name.value.decode(), fl.value, by.value))
        return out

    def set_probe(self, layer_index, max_records):
        check(lib().sqdet_net_set_probe(self._h, int(layer_index), int(max_records)), "sqdet_net_set_probe")

    def read_probe(self, capacity):
        ms = (C.c_float * capacity)()
        cnt = C.c_int()
        check(lib().sqdet_net_read_probe(self._h, ms, capacity, C.byref(cnt)), "sqdet_net_read_probe")
        return [float(ms[i]) for i in range(cnt.value)]

    def forward_timed(self, image_input, preds=None):
        """Per-launch milliseconds measured with HIP events on the launch stream."""
        if preds is None:
            preds = torch.empty((self.batch, self.gh, self.gw, self.out_ch), dtype=self.dtype, device=self.device)
        nl = lib().sqdet_net_num_layers(self._h)
        ms = (C.c_float * nl)()
        check(lib().sqdet_net_forward_timed(self._h, _dev(image_input, "image_input"), _dev(preds, "preds"), ms,
                                            stream_ptr()), "sqdet_net_forward_timed")
        return preds, [float(v) for v in ms]


# ---------------------------------------------------------------- torch.ops.sqdet.*
from . import torch_ops as _torch_ops  # noqa: E402,F401  (registers the custom ops; see that module)
