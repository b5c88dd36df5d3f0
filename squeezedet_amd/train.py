"""Training step of SqueezeDet on MI355X: forward (training mode) -> loss -> backward -> gradient
all-reduce -> clipped Momentum update, every arithmetic step a HIP kernel of libsqdet_hip.so.
Mirrors the reference's train graph (src/nn_skeleton.py:285-361: _add_loss_graph,
_add_train_graph) and the training loop body of src/train.py:302-304
(`sess.run([train_op, loss, conf_loss, bbox_loss, class_loss])`).

Data parallelism: one process per GPU, replicas hold full weights, ONE flat float32
gradient bucket all-reduced (SUM) per step over RCCL (torch.distributed backend "nccl"), divided by
the world size inside the optimizer kernel, then per-variable clip_by_norm and Momentum -- applied
identically (and deterministically) on every rank, so replicas stay bit-identical (the flat parameter and momentum
buffers are broadcast from rank 0 when a trainer is built).  Loss normalisation: "replica-mean" by default -- each
replica normalises by its own num_objects (the reference at its own batch size), gradients are averaged --, or, with
`global_num_objects=True`, the exact global-batch form (SURVEY.md 8e option b): num_objects is SUM-all-reduced over the
replicas before the loss (one extra scalar collective) and the gradient bucket is summed, not averaged, so N replicas of
batch B compute what the reference computes at batch N*B (nn_skeleton.py:180,297,307-308,321).

num_objects lives on the device (sqdet_sum_f32 of the mask) and the overflow flag is read lazily, so the forward +
loss + backward of a step has no host round trip and can be replayed as one hipGraph (GraphedStep below).
"""
import collections
import os

import numpy as np
import torch

from . import ops
from ._lib import SqdetError, SqdetUnsupported


def allreduce_gradients(flat_grads, world, group=None):
    """The ONE collective of a training step: SUM all-reduce of the flat float32 gradient bucket over
    RCCL (xGMI).  Returns the factor (1/world) the optimizer kernel multiplies the summed gradients
    by before clipping -- i.e. replicas apply the MEAN gradient, identically on every rank."""
    if world > 1:
        torch.distributed.all_reduce(flat_grads, op=torch.distributed.ReduceOp.SUM, group=group)
    return 1.0 / world


def reduce_num_objects(num_objects, world, group=None):
    """global_num_objects mode: the scalar num_objects = sum(input_mask) (nn_skeleton.py:180) SUM-all-reduced in place over
    the replicas -- the second (one-element) collective of a step in that mode."""
    if world > 1:
        torch.distributed.all_reduce(num_objects, op=torch.distributed.ReduceOp.SUM, group=group)
    return num_objects


def step_normalisation(global_num_objects, local_batch, world):
    """(global_batch argument of the loss kernel, factor applied to the SUM-all-reduced gradient bucket).
    replica-mean (default): every replica is the reference at its own batch -- loss divisors local (0 = this call's
    batch), gradients averaged (1/world).  global: the replicas' losses are shares of ONE graph of batch world*B -- the
    confidence term's batch divisor is world*B (the class / bbox terms only see the all-reduced num_objects), and the
    shares' gradients are SUMMED (factor 1)."""
    if global_num_objects:
        return int(local_batch) * int(world), 1.0
    return 0, 1.0 / world


class _TrainerBase:
    """Flat float32 parameter / gradient / momentum buffers over the trainable variables of a model built with
    mc.IS_TRAINING = True, the gradient all-reduce and the optimizer step.

    model.dtype float32 is the reference's training dtype.  model.dtype float16 is mixed-precision training
    (BASELINE.json configs[4]): activations and activation gradients are float16, the master weights, weight / bias
    gradients, momentum and the loss stay float32; d(loss)/d(preds) is multiplied by `loss_scale` before it is cast to
    float16 and the backward-filter kernels divide it out again, so the flat gradients are true-scale.  A step whose
    gradients overflowed (inf / NaN norm in any variable, on any rank -- the SUM all-reduce spreads it) is skipped by
    the optimizer kernel and halves the scale; `growth_interval` clean steps double it."""

    def __init__(self, model, process_group=None, loss_scale=1024.0, growth_interval=200, lazy_overflow_check=None,
                 global_num_objects=False, seed=0, overlap_wgrad=True):
        if model.dtype not in (torch.float32, torch.float16):
            raise SqdetError("training runs in float32 (the reference's training dtype) or float16 (mixed precision)")
        self.adt = model.dtype                       # activation dtype
        self.half = model.dtype == torch.float16
        self.loss_scale = float(loss_scale) if self.half else 1.0
        self.growth_interval, self._clean_steps, self.skipped_steps = int(growth_interval), 0, 0
        # lazy_overflow_check: read a step's overflow flag at the END OF THE NEXT step (asynchronous copy to pinned
        # memory + event) instead of synchronising with the device every step.  The skipped update itself happens on the
        # device either way; only the loss-scale / counter bookkeeping lags one step.  flush() settles the last one.
        # Default: lazy in float32 (the flag can only raise the divergence error there -- no reason to pay a device
        # synchronisation per step for it; global_step advances at once, so the staircase decay fires on the reference's
        # step), eager in float16 (the next step's loss scale depends on it).
        self.lazy_overflow_check = (not self.half) if lazy_overflow_check is None else bool(lazy_overflow_check)
        self._pending_flag = None
        if not model.has_device:
            raise SqdetError("squeezedet_amd needs a HIP device: there is no CPU path")
        self.model, self.mc, self.dev = model, model.mc, model.device
        # process_group=None: the default (WORLD) group when torch.distributed is initialised, else single-process
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.global_num_objects = bool(global_num_objects)
        # overlap_wgrad: a layer's backward-filter launches (weight gradient + its slab reduction, BN-fold backward) run on a
        # side stream beside the backward-data chain -- the two only share inputs.  On the late 24 x 78 maps a launch cannot
        # fill the chip (15-37 k pixels), so the pair really runs concurrently; inside a captured step the fork / join
        # become graph edges.  Same kernels, same order per buffer: bitwise the sequential results.
        self.overlap_wgrad = bool(overlap_wgrad) and os.environ.get("SQDET_WGRAD_OVERLAP", "1") != "0"
        self._wg_stream, self._wg_keep = None, []
        # (one WgradPlan per input shape, each pinning its conv's split-K slab workspaces: bounded like GraphedStep's cache,
        # least recently used first out)
        self._wplans, self.plan_wgrads = {}, os.environ.get("SQDET_WGRAD_PLAN", "1") != "0"
        # dropout masks are independent across the global batch (one tf.nn.dropout over all samples in the reference,
        # nets/squeezeDet.py:74): every replica draws from its own counter stream -- only parameters and momentum must
        # match across ranks, masks must not
        self.rank = torch.distributed.get_rank(process_group) if self.world > 1 else 0
        self.seed, self._mask_calls = int(seed) * self.world + self.rank, 0
        self.global_step = 0
        # trainable variables (conv1 is frozen: nets/squeezeDet.py:40-42) packed into flat buffers
        self.names = [n for n in model.params if model.trainable[n]]
        offs, cnts, decs, o = [], [], [], 0
        for n in self.names:
            c = model.params[n].numel()
            offs.append(o)
            cnts.append(c)
            decs.append(self.mc.WEIGHT_DECAY if n.endswith("/kernels") else 0.0)   # nn_skeleton.py:66-69
            o += (c + 63) // 64 * 64
        self.total = o
        self.flat_params = torch.zeros(o, dtype=torch.float32, device=self.dev)
        self.flat_grads = torch.zeros(o, dtype=torch.float32, device=self.dev)
        self.flat_accum = torch.zeros(o, dtype=torch.float32, device=self.dev)
        self.view, self.gview = {}, {}
        for n, off, c in zip(self.names, offs, cnts):
            shp = tuple(model.params[n].shape)
            self.view[n] = self.flat_params[off:off + c].view(shp)
            self.view[n].copy_(model.params[n])
            model.params[n] = self.view[n]           # the model now reads the flat buffer
            self.gview[n] = self.flat_grads[off:off + c].view(shp)
        model._packed.clear()
        model._plan_stale = True
        self.opt = ops.MomentumOptimizer(offs, cnts, decs, self.dev)
        self.found_inf = torch.zeros(1, dtype=torch.int32, device=self.dev)
        if self.world > 1:      # replicas start from rank 0's variables and momentum: they stay bit-identical from here on
            torch.distributed.broadcast(self.flat_params, src=torch.distributed.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
            torch.distributed.broadcast(self.flat_accum, src=torch.distributed.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
            # ... and from rank 0's FROZEN variables (conv1 of SqueezeDet, conv1 .. res3d and the batch-norm statistics of ResNet50):
            # they are not in the flat buffers, and a replica that loaded other values would compute other gradients for good
            src0 = torch.distributed.get_global_rank(self.pg, 0) if self.pg is not None else 0
            for n in model.params:
                if not model.trainable[n]:
                    t = model.params[n].contiguous()
                    torch.distributed.broadcast(t, src=src0, group=self.pg)
                    model.params[n].copy_(t)
            model._packed.clear()
            model._plan_stale = True

    MAX_WPLANS = 4

    def _wplan_get(self, key):
        plan = self._wplans.get(key)
        if plan is not None:
            self._wplans[key] = self._wplans.pop(key)            # most recently used last
        return plan

    def _wplan_put(self, key, plan):
        self._wplans[key] = plan
        while len(self._wplans) > self.MAX_WPLANS:
            del self._wplans[next(iter(self._wplans))]

    def _wgrad(self, fn, *reads):
        """Runs fn() -- weight-gradient launches writing into the flat gradient bucket -- behind everything issued so far,
        on the side stream when overlap_wgrad; `reads`: tensors it reads that the caller is about to drop (kept alive
        until _join_wgrad)."""
        if not self.overlap_wgrad:
            fn()
            return
        if self._wg_stream is None:
            self._wg_stream = torch.cuda.Stream(device=self.dev)
        cur = torch.cuda.current_stream()
        self._wg_stream.wait_stream(cur)
        with torch.cuda.stream(self._wg_stream):
            fn()
        self._wg_keep.extend(reads)

    def fork_labels(self, build):
        """Runs build() -- the label construction of a step: a handful of small latency-bound launches nothing in the forward
        depends on -- on the side stream, beside the forward; forward_backward joins it ahead of the loss.  Everything build()
        returns must be produced there (pass num_objects in: _labels would sum the mask on the main stream)."""
        if not self.overlap_wgrad:
            return build()
        if self._wg_stream is None:
            self._wg_stream = torch.cuda.Stream(device=self.dev)
        self._wg_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._wg_stream):
            out = build()
        self._labels_forked = True
        return out

    def _join_labels(self):
        if getattr(self, "_labels_forked", False):
            torch.cuda.current_stream().wait_stream(self._wg_stream)
            self._labels_forked = False

    def _before_inplace(self, t):
        """The caller is about to write `t` in place on its stream: if a pending side-stream weight gradient reads it, wait for
        the side stream first (ResNet's residual gradients are shared by the branch and the shortcut and accumulated into)."""
        if self._wg_stream is not None and any(t is r for r in self._wg_keep):
            torch.cuda.current_stream().wait_stream(self._wg_stream)
            self._wg_keep = []

    def _join_wgrad(self):
        if self._wg_stream is not None and self.overlap_wgrad:
            torch.cuda.current_stream().wait_stream(self._wg_stream)
        self._wg_keep = []

    def learning_rate(self):
        mc = self.mc
        return mc.LEARNING_RATE * mc.LR_DECAY_FACTOR ** (self.global_step // mc.DECAY_STEPS)   # staircase decay

    def _labels(self, B, input_mask, box_delta_input, box_input, labels, num_objects=None, num_objects_is_global=False):
        t = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).to(self.dev, torch.float32).contiguous()
        mask = t(input_mask).reshape(B, -1)
        if num_objects is None:       # sum(input_mask) on the device (nn_skeleton.py:180): no host round trip
            num_objects = ops.sum_f32(mask)
        elif not isinstance(num_objects, torch.Tensor):
            num_objects = torch.full((1,), float(num_objects), dtype=torch.float32, device=self.dev)
        if self.global_num_objects and self.world > 1 and not num_objects_is_global:      # exact global-batch normalisation: one scalar all-reduce
            if torch.cuda.is_current_stream_capturing():
                raise SqdetError("global_num_objects at world > 1: the num_objects all-reduce must run eagerly -- pass "
                                 "num_objects (already all-reduced) into the captured step (GraphedStep does)")
            reduce_num_objects(num_objects, self.world, self.pg)
        return t, mask, t(box_delta_input), t(box_input), t(labels), num_objects

    def _mask_tensor(self, dropout_mask, t):
        """A caller-supplied or generated keep mask as a device tensor of the activation dtype."""
        if isinstance(dropout_mask, torch.Tensor) and dropout_mask.is_cuda and dropout_mask.dtype == self.adt:
            return dropout_mask.contiguous()
        dm = t(dropout_mask)                           # float32 on the device
        return dm if self.adt == torch.float32 else ops.convert_scale(dm, self.adt)

    def _dropout_mask(self, shape, keep):
        """tf.nn.dropout's keep mask floor(keep_prob + U) from the counter-based HIP generator; a new seed per call."""
        self._mask_calls += 1
        return ops.dropout_mask(tuple(shape), keep, (self.seed << 32) + self._mask_calls, self.adt, self.dev)

    def _loss(self, preds, mask, delta, box, lab, num_objects):
        """Loss forward + backward in float32; returns (gradient w.r.t. preds in the activation dtype -- times
        loss_scale in float16 mode --, float32 dpreds, ious, losses)."""
        # global_num_objects: this replica's loss is its share of ONE graph of batch world*B -- the confidence term's
        # reduce_mean over the batch (nn_skeleton.py:304-312) divides by the global batch too; the bucket is then SUMMED
        gb, _ = step_normalisation(self.global_num_objects, int(preds.shape[0]), self.world)
        if self.half:     # float16 preds in, float32 dpreds + its loss-scaled float16 copy out: one launch, no conversions around it
            return ops.loss_fwd_bwd_mixed(preds, self.model.anchors_f32(), mask, delta, box, lab, self.mc, num_objects, self.loss_scale, global_batch=gb)
        dpreds, ious, losses = ops.loss_fwd_bwd(preds, self.model.anchors_f32(), mask, delta, box, lab, self.mc, num_objects, global_batch=gb)
        return dpreds, dpreds, ious, losses

    def _finish_step(self, apply_update):
        """Gradient all-reduce (the one collective) + clipped Momentum update on the flat buffers."""
        allreduce_gradients(self.flat_grads, self.world, self.pg)
        # (global mode: every replica's loss is its share of the global-batch graph: the bucket is a sum, factor 1)
        _, grad_scale = step_normalisation(self.global_num_objects, 0, self.world)
        if apply_update:
            if self.lazy_overflow_check:
                # settle the PREVIOUS step's flag (its event completed long ago) BEFORE this step's update is enqueued: a
                # float32 overflow raises here with global_step / skipped_steps describing exactly the updates applied
                self.flush()
            # (the kernel skips the whole update when any gradient norm is inf / NaN, in either precision: found_inf tells)
            self.opt.step(self.flat_params, self.flat_grads, self.flat_accum, self.learning_rate(), self.mc.MOMENTUM,
                          self.mc.MAX_GRAD_NORM, grad_scale, found_inf=self.found_inf)
            # packed / BN-folded kernels of the layers whose variables just changed are stale; the frozen layers'
            # (conv1 of SqueezeDet, conv1 .. res3d of ResNet50 -- re-folded and re-packed every step before) are not
            tr = self.model.trainable
            for k in [k for k in self.model._packed if tr.get(k + "/kernels", True)]:
                del self.model._packed[k]
            self.model._plan_stale = True
            if not self.lazy_overflow_check:
                self._account(bool(int(self.found_inf.item())))
            else:
                if getattr(self, "_flag_host", None) is None:
                    self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                    self._flag_event = torch.cuda.Event()
                self._flag_host.copy_(self.found_inf, non_blocking=True)
                self._flag_event.record(torch.cuda.current_stream())
                self._pending_flag = True
                if not self.half:
                    self.global_step += 1                     # float32: counted now, the late flag can only raise

    def flush(self):
        """lazy_overflow_check: settle the bookkeeping of the last step (loss scale, skipped / global step counters).  MANDATORY
        after the last step of a run in lazy mode (the float32 default): a divergence in the final step is reported here and
        nowhere else -- the reference asserts on the loss every step (train.py:302-310); bench.py and the tests call it."""
        if self._pending_flag:
            self._flag_event.synchronize()
            self._pending_flag = None
            overflowed = bool(int(self._flag_host[0]))
            if self.half:
                self._account(overflowed)
            elif overflowed:
                self.global_step -= 1                         # the kernel skipped that update
                self._account(True)                           # raises FloatingPointError

    def _account(self, overflowed):
        if overflowed and not self.half:
            # float32 has no loss scale to lower: a non-finite gradient means the run has diverged.  The reference asserts
            # on the loss (train.py:313); here the kernel has left weights and momentum untouched and the caller is told.
            self.skipped_steps += 1
            raise FloatingPointError("non-finite gradient norm in float32 training (step %d): the update was skipped" % self.global_step)
        if not self.half:
            self.global_step += 1
            return
        if overflowed:
            # the kernel left weights and momentum untouched; retry the next batch at half the scale
            self.loss_scale = max(self.loss_scale / 2.0, 2.0 ** -14)
            self._clean_steps = 0
            self.skipped_steps += 1
            return
        self._clean_steps += 1
        if self._clean_steps >= self.growth_interval:
            self.loss_scale, self._clean_steps = min(self.loss_scale * 2.0, 65536.0), 0
        self.global_step += 1

    def weight_decay_loss(self):
        """sum wd * l2_loss(kernel) over trainable kernels (the 'losses' collection of nn_skeleton.py:66-69)."""
        tot = 0.0
        for n in self.names:
            if n.endswith("/kernels"):
                tot += float((self.view[n].double() ** 2).sum().item()) * self.mc.WEIGHT_DECAY / 2
        return tot


class SqueezeDetTrainer(_TrainerBase):
    """model: a squeezedet_amd.nets.SqueezeDet built with mc.IS_TRAINING = True and dtype float32."""

    def __init__(self, model, process_group=None, **kw):
        _TrainerBase.__init__(self, model, process_group, **kw)
        self.layers = self._layer_list()
        # every trainable kernel is re-packed (forward order; backward-data order for all but the lowest trainable conv) by
        # ONE launch at the top of a step instead of 62 (ops.PackPlan); the frozen conv1 is packed once
        kernels = collections.OrderedDict((n[:-len("/kernels")], self.view[n]) for n in self.names if n.endswith("/kernels"))
        self.packplan = ops.PackPlan(kernels, self.adt, bwd_names=set(kernels))
        self._frozen_packed = {}

    # ---- the forward graph as a list (nets/squeezeDet.py:30-79) ----
    def _layer_list(self):
        m = self.model
        seq, node = [], m.preds
        chain = []
        # walk back from preds through the graph
        def walk(n):
            if n.op == "placeholder":
                return
            walk(n.inputs[0] if n.op != "concat" else n.inputs[0].inputs[0].inputs[0])
            chain.append(n)
        walk(node)
        for n in chain:
            if n.op == "concat":
                e1, e3 = n.inputs
                seq.append(("fire", n.name.split("/")[0], e1.inputs[0], e1, e3))
            elif n.op == "conv":
                seq.append(("conv", n.name, n))
            elif n.op == "pool":
                seq.append(("pool", n.name, n))
        return seq

    def _pack(self, name):
        if name in self.packplan.fwd:
            return self.packplan.fwd[name]
        w = self.model.params[name + "/kernels"]     # frozen layers (conv1): re-packed only when someone wrote the variable (load_params)
        hit = self._frozen_packed.get(name)
        if hit is None or hit[1] != w._version:
            hit = self._frozen_packed[name] = (ops.pack_conv_weights(w, self.adt), w._version)
        return hit[0]

    def step(self, images, input_mask, box_delta_input, box_input, labels, dropout_mask=None, apply_update=True,
             keep_activations=False, num_objects=None):
        """One training step.  images [B,H,W,3]; input_mask [B,A] or [B,A,1]; box_delta_input / box_input
        [B,A,4]; labels [B,A,C] (the reference's placeholders, nn_skeleton.py:81-97).  Returns a dict
        with loss, class_loss, conf_loss, bbox_loss (device scalars).  keep_activations: also return every stored
        forward activation as out["activations"] = {name: tensor} (names as oracle/train_oracle.py forward_train's
        `override`), for parity tests of the backward pass."""
        with torch.cuda.device(self.dev):
            out = self.forward_backward(images, input_mask, box_delta_input, box_input, labels, dropout_mask, keep_activations, num_objects)
            self._finish_step(apply_update)
        return out

    def forward_backward(self, images, input_mask, box_delta_input, box_input, labels, dropout_mask=None,
                         keep_activations=False, num_objects=None, num_objects_is_global=False):
        """Forward + loss + backward into the flat gradient bucket: kernel launches and stream-ordered allocations only,
        no host round trip (hipGraph-capturable).  _finish_step (all-reduce + update) completes the step."""
        m, mc, P = self.model, self.mc, self.model.params
        acts = {}
        self.packplan.run()          # the variables may have changed since the last step (optimizer, load_params): one launch
        x = m._to_input(images)
        B = int(x.shape[0])
        t, mask, delta, box, lab, num_objects = self._labels(B, input_mask, box_delta_input, box_input, labels, num_objects, num_objects_is_global)
        keep = m.keep_prob
        # ---------------- forward, keeping what the backward needs ----------------
        saved = []
        cur = x
        drop_in = None
        skip_pool = None
        for li, item in enumerate(self.layers):
            if item[0] == "pool" and item[2] is skip_pool:
                continue
            if item[0] == "conv":
                node = item[2]
                nxt = self.layers[li + 1] if li + 1 < len(self.layers) else None
                if (li == 0 and not keep_activations and not m.trainable[node.name + "/kernels"] and nxt is not None
                        and nxt[0] == "pool" and nxt[2].attrs["size"] == 3 and nxt[2].attrs["stride"] == 2
                        and node.attrs["stride"] == 2 and node.attrs["relu"]
                        and ops.stem_supported(int(P[node.name + "/kernels"].shape[3]), int(P[node.name + "/kernels"].shape[0]))):
                    # frozen conv1 + pool1 (nets/squeezeDet.py:40-44): nothing below pool1's output is needed by the
                    # backward, so the fused stem launch serves the training forward too
                    try:
                        y = ops.stem_conv_pool(cur, self._pack(node.name), P[node.name + "/biases"], node.attrs["padding"],
                                               nxt[2].attrs["padding"])
                    except SqdetUnsupported:
                        y = None        # e.g. set_option("conv_algo", 1): the separate conv + pool kernels below
                    if y is not None:
                        saved.append(("conv", node, cur, None))
                        saved.append(("pool", nxt[2], None, y, None))
                        acts[nxt[2].name] = y
                        cur = y
                        skip_pool = nxt[2]
                        continue
                if node.name == "conv12":
                    drop_in = cur
                    if dropout_mask is None:
                        dropout_mask = self._dropout_mask(cur.shape, keep)                           # tf.nn.dropout's mask
                    dm = self._mask_tensor(dropout_mask, t)
                    cur = ops.scale_mask(cur, dm, 1.0 / keep) if keep != 1.0 else cur
                    acts["drop"] = cur
                y = ops.conv2d_nhwc(cur, self._pack(node.name), P[node.name + "/biases"], node.attrs["stride"],
                                    node.attrs["padding"], node.attrs["relu"])
                saved.append(("conv", node, cur, y))
                acts[node.name] = y
                cur = y
            elif item[0] == "pool":
                node = item[2]
                if node.attrs["size"] == 3 and node.attrs["stride"] == 2:
                    # the window index rides along: the backward reads it (+ y, dy) instead of searching x again
                    y, widx = ops.maxpool_nhwc_idx(cur, 3, 2, node.attrs["padding"])
                else:
                    y, widx = ops.maxpool_nhwc(cur, node.attrs["size"], node.attrs["stride"], node.attrs["padding"]), None
                saved.append(("pool", node, cur, y, widx))
                acts[node.name] = y
                cur = y
            else:
                _, fname, sq, e1, e3 = item
                if self.half:
                    # the module in ONE launch (sqdet_fire_fwd_keep: the fused kernel's squeeze epilogue also writes the
                    # squeeze tensor the backward reads): mixed-precision step 3.40 -> 3.25 ms
                    y, s = ops.fire(cur, self._pack(sq.name), P[sq.name + "/biases"], self._pack(e1.name), P[e1.name + "/biases"],
                                    self._pack(e3.name), P[e3.name + "/biases"], keep_squeeze=True)
                else:
                    # float32: the fused tile kernels spill at this width -- squeeze + the two expand convs are faster (9.92 vs 10.16 ms)
                    s = ops.conv2d_nhwc(cur, self._pack(sq.name), P[sq.name + "/biases"], 1, "SAME", True)
                    y = torch.empty((B, int(s.shape[1]), int(s.shape[2]), e1.shape[3] + e3.shape[3]), dtype=self.adt, device=self.dev)
                    ops.conv2d_nhwc(s, self._pack(e1.name), P[e1.name + "/biases"], 1, "SAME", True, out=y, out_coffset=0)
                    ops.conv2d_nhwc(s, self._pack(e3.name), P[e3.name + "/biases"], 1, "SAME", True, out=y, out_coffset=e1.shape[3])
                saved.append(("fire", (sq, e1, e3), cur, s, y))
                acts[fname + "/squeeze1x1"], acts[fname] = s, y
                cur = y
        preds = cur
        # ---------------- loss ----------------
        self._join_labels()
        g, dpreds, ious, losses = self._loss(preds, mask, delta, box, lab, num_objects)
        # ---------------- backward ----------------
        self.flat_grads.zero_()
        gs = 1.0 / self.loss_scale      # g: gradient w.r.t. the current layer's OUTPUT (pre-activation mask applied below)
        bwd = lambda name: self.packplan.bwd[name] if name in self.packplan.bwd else ops.PackedConvBwd(P[name + "/kernels"], self.adt)
        # weight gradients: every conv's gradient kernel writes its partial slabs into a workspace of its own and ONE launch
        # at the end sums them all (ops.WgradPlan); the first step of an input shape runs the per-conv two-launch form and
        # records what the plan needs
        wkey = tuple(int(v) for v in x.shape)
        wplan = self._wplan_get(wkey)
        witems = []

        def wg(name, xt, gt, k, cin, cout, dy_coffset=0):
            if wplan is not None:
                wplan.partial(name, xt, gt, dy_coffset=dy_coffset)
                return
            witems.append((name, (int(xt.shape[0]), int(xt.shape[1]), int(xt.shape[2]), cin, cout, k),
                           self.gview[name + "/kernels"], self.gview[name + "/biases"], None, 0.0))
            ops.conv2d_bwd_filter(xt, gt, k, cin, cout, dy_coffset=dy_coffset, dw=self.gview[name + "/kernels"],
                                  db=self.gview[name + "/biases"], grad_scale=gs)

        def has_trainable(rec):
            if rec[0] == "conv":
                return m.trainable[rec[1].name + "/kernels"]
            return rec[0] == "fire"
        first_tr = min(i for i, r in enumerate(saved) if has_trainable(r))
        # ReLU backward: every gradient w.r.t. a ReLU output is masked in the epilogue of the kernel that PRODUCES it (the
        # backward-data conv or the max-pool backward above it) -- `masked` says g already carries the mask of the layer
        # whose output it is the gradient of; the loss gradient of conv12 (no ReLU) and any other case fall back to relu_bwd
        masked = False

        def relu_src(rj):
            """The ReLU output the record below ri produces (= the tensor dx is the gradient of), or None."""
            r = saved[rj]
            if r[0] == "fire":
                return r[4]
            if r[0] == "conv" and r[1].attrs["relu"] and r[3] is not None:
                return r[3]
            return None
        for ri in range(len(saved) - 1, first_tr - 1, -1):
            rec = saved[ri]
            need_dx = ri > first_tr     # nothing trainable (and no image gradient) below the first trainable layer
            below = relu_src(ri - 1) if ri > 0 else None          # x of this layer, when it is a ReLU output
            if rec[0] == "conv":
                _, node, xin, y = rec
                name = node.name
                if node.attrs["relu"] and not masked:
                    ops.relu_bwd(y, g)
                k = node.attrs["size"]
                cin, cout = int(xin.shape[3]), int(y.shape[3])
                self._wgrad(lambda xin=xin, g=g, k=k, cin=cin, cout=cout, name=name: wg(name, xin, g, k, cin, cout), g)
                masked = False
                if need_dx:
                    if name == "conv12" and keep != 1.0:
                        g = ops.conv2d_bwd_data(g, bwd(name))
                        g = ops.scale_mask(g, dm, 1.0 / keep, relu_of=below)      # dropout backward (+ the ReLU backward below it)
                        masked = below is not None
                    else:
                        g = ops.conv2d_bwd_data(g, bwd(name), relu_of=below)
                        masked = below is not None
            elif rec[0] == "pool":
                _, node, xin, y, widx = rec
                if widx is not None:
                    g = ops.maxpool_bwd_idx(widx, y, g, xin.shape[1:3], 3, 2, node.attrs["padding"], relu=below is not None)
                else:
                    g = ops.maxpool_bwd(xin, g, node.attrs["size"], node.attrs["stride"], node.attrs["padding"], relu=below is not None)
                masked = below is not None
            else:
                _, (sq, e1, e3), xin, s, y = rec
                ne1, ne3, ns = e1.shape[3], e3.shape[3], sq.shape[3]
                if not masked:
                    ops.relu_bwd(y, g)      # both expand convs end in ReLU
                def expand_wgrads(s=s, g=g, ns=ns, ne1=ne1, ne3=ne3, e1=e1, e3=e3):
                    wg(e1.name, s, g, 1, ns, ne1, dy_coffset=0)
                    wg(e3.name, s, g, 3, ns, ne3, dy_coffset=ne1)
                self._wgrad(expand_wgrads, g)
                ds = ops.conv2d_bwd_data(g, bwd(e1.name), dy_coffset=0)
                ops.conv2d_bwd_data(g, bwd(e3.name), dx=ds, dy_coffset=ne1, accumulate=True, relu_of=s)   # + the squeeze's ReLU backward
                self._wgrad(lambda xin=xin, ds=ds, ns=ns, sq=sq: wg(sq.name, xin, ds, 1, int(xin.shape[3]), ns), ds)
                masked = False
                if need_dx:
                    g = ops.conv2d_bwd_data(ds, bwd(sq.name), relu_of=below)
                    masked = below is not None
        self._join_wgrad()
        if wplan is not None:
            wplan.reduce(gs)
        elif self.plan_wgrads:
            self._wplan_put(wkey, ops.WgradPlan(witems))
        out = collections.OrderedDict(class_loss=losses[0], conf_loss=losses[1], bbox_loss=losses[2], ious=ious, preds=preds,
                                      dpreds=dpreds, num_objects=num_objects)
        if keep_activations:
            out["activations"] = acts
        return out


class ResNet50ConvDetTrainer(_TrainerBase):
    """Training step of ResNet50ConvDet (nets/resnet50_convDet.py:20-169 + nn_skeleton.py:285-361): conv1..res3d
    are frozen (:41-92), so the forward below res4a runs on the inference kernels and the backward stops at
    res4a's input.  The trainable _conv_bn_layer convs run with the batch norm FOLDED (frozen statistics make
    it affine): forward = sqdet_fold_batchnorm + conv, backward = the conv backward kernels on the folded
    kernel + sqdet_fold_batchnorm_bwd for d(kernels), d(gamma), d(beta).  float32, like the reference."""

    def __init__(self, model, process_group=None, **kw):
        _TrainerBase.__init__(self, model, process_group, **kw)
        m = model
        order, seen = [], set()

        def topo(n):
            if n in seen:
                return
            seen.add(n)
            for i in n.inputs:
                topo(i)
            order.append(n)
        topo(m.preds)
        has_tr = lambda n: n.op in ("conv", "conv_bn") and m.trainable[n.name + "/kernels"]
        first = next(n for n in order if has_tr(n))
        self.boundary = first.inputs[0]                 # last frozen activation (res3d)
        below, stack = set(), [self.boundary]
        while stack:
            n = stack.pop()
            if n not in below:
                below.add(n)
                stack.extend(n.inputs)
        self.region = [n for n in order if n not in below]   # trainable suffix, topological order
        for n in self.region:
            if n.op in ("conv", "conv_bn") and not has_tr(n):
                raise SqdetError("frozen conv %s above the first trainable one is not supported" % n.name)
        # gradient contributions each node of the region receives in the backward (one per reader)
        self._grad_fanin = collections.Counter(i for n in self.region for i in n.inputs if i is not self.boundary)
        self.fuse_relu_bwd = os.environ.get("SQDET_FUSE_RELU_BWD", "1") != "0"
        self._plans_for = None
        self._build_plans()

    def _bn_pointers(self):
        P = self.model.params
        return tuple(P[n.name + s].data_ptr() for n in self.region if n.op == "conv_bn" for s in ("/mean", "/var"))

    def _build_plans(self):
        """One launch per step for everything that is per-variable bookkeeping (was ~6 launches of 5-10 us for each of the 19
        trainable convs): PackPlan folds the batch norms and packs every kernel in both orders; FoldBwdPlan turns the
        folded gradients (staged in self.dwf / self.dbf by the weight-gradient kernels) into d(kernels), d(gamma), d(beta)."""
        P, eps = self.model.params, self.mc.BATCH_NORM_EPSILON
        convs = [n for n in self.region if n.op in ("conv", "conv_bn")]
        kernels = collections.OrderedDict((n.name, self.view[n.name + "/kernels"]) for n in convs)
        bn = {n.name: (P[n.name + "/gamma"], P[n.name + "/beta"], P[n.name + "/mean"], P[n.name + "/var"], None)
              for n in convs if n.op == "conv_bn"}
        bwd_names = set(n.name for n in convs if n.attrs["stride"] == 1 and n.inputs[0] is not self.boundary)
        self.packplan = ops.PackPlan(kernels, self.adt, bwd_names=bwd_names, bn=bn, eps=eps)
        self.dwf = {n.name: torch.empty_like(self.view[n.name + "/kernels"]) for n in convs if n.op == "conv_bn"}
        self.dbf = {n.name: torch.empty(int(self.view[n.name + "/kernels"].shape[3]), dtype=torch.float32, device=self.dev) for n in convs if n.op == "conv_bn"}
        self.foldplan = ops.FoldBwdPlan([(self.view[n.name + "/kernels"], self.dwf[n.name], self.dbf[n.name], None, P[n.name + "/gamma"],
                                          P[n.name + "/mean"], P[n.name + "/var"], self.gview[n.name + "/kernels"],
                                          self.gview[n.name + "/gamma"], self.gview[n.name + "/beta"]) for n in convs if n.op == "conv_bn"], eps)
        self._plans_for = self._bn_pointers()

    def step(self, images, input_mask, box_delta_input, box_input, labels, dropout_mask=None, apply_update=True,
             keep_activations=False, num_objects=None):
        with torch.cuda.device(self.dev):
            out = self.forward_backward(images, input_mask, box_delta_input, box_input, labels, dropout_mask, keep_activations, num_objects)
            self._finish_step(apply_update)
        return out

    def forward_backward(self, images, input_mask, box_delta_input, box_input, labels, dropout_mask=None,
                         keep_activations=False, num_objects=None, num_objects_is_global=False):
        m, mc, P = self.model, self.mc, self.model.params
        eps = mc.BATCH_NORM_EPSILON
        if self._plans_for != self._bn_pointers():     # load_params replaced the (non-trainable) moving statistics
            self._build_plans()
        self.packplan.run()     # every trainable kernel, batch norm folded, in both fragment orders + the folded biases: one launch
        pk, pbias = self.packplan.fwd, self.packplan.bias
        (xb,) = m.run([self.boundary], {m.image_input: images}, use_plan=False)
        B = int(xb.shape[0])
        t, mask, delta, box, lab, num_objects = self._labels(B, input_mask, box_delta_input, box_input, labels, num_objects, num_objects_is_global)
        # ---------------- forward over the trainable region ----------------
        val, aux = {self.boundary: xb}, {}
        for n in self.region:
            if n.op == "conv_bn":
                x = val[n.inputs[0]]
                fused_add = len(n.readers) == 1 and n.readers[0].op == "add_relu" and n.readers[0].inputs[1] is n and not n.attrs["relu"]
                if fused_add:
                    continue        # evaluated by its add_relu reader (residual epilogue)
                val[n] = ops.conv2d_nhwc(x, pk[n.name], pbias[n.name], n.attrs["stride"], n.attrs["padding"], n.attrs["relu"])
            elif n.op == "add_relu":
                sc, br = n.inputs
                if br in val:
                    val[n] = ops.add_relu(val[sc], val[br])
                else:
                    # (the shortcut is branch2a's input: its value is needed by the backward -- the conv reads it as a residual tensor
                    #  of its own instead of adding into a copy)
                    val[n] = ops.conv2d_nhwc(val[br.inputs[0]], pk[br.name], pbias[br.name], br.attrs["stride"],
                                             br.attrs["padding"], True, residual=val[sc])
            elif n.op == "dropout":
                x = val[n.inputs[0]]
                keep = n.attrs["keep_prob"]
                if dropout_mask is None:
                    dropout_mask = self._dropout_mask(x.shape, keep)
                aux[n] = self._mask_tensor(dropout_mask, t)
                val[n] = ops.scale_mask(x, aux[n], 1.0 / keep)
            elif n.op == "conv":
                x = val[n.inputs[0]]
                val[n] = ops.conv2d_nhwc(x, pk[n.name], P[n.name + "/biases"], n.attrs["stride"], n.attrs["padding"], n.attrs["relu"])
            else:
                raise SqdetError("ResNet50ConvDetTrainer: unsupported op %s in the trainable region" % n.op)
        preds = val[m.preds]
        self._join_labels()
        g0, dpreds, ious, losses = self._loss(preds, mask, delta, box, lab, num_objects)
        # ---------------- backward ----------------
        self.flat_grads.zero_()
        g = {m.preds: g0}
        gs = 1.0 / self.loss_scale
        # weight gradients: partial slabs per conv, ONE reduction (ops.WgradPlan; the folded gradients of the conv_bn convs
        # land in self.dwf / self.dbf), then ONE fold backward; the first step of an input shape runs the per-conv
        # two-launch gradient and records what the plan needs
        wkey = tuple(int(v) for v in xb.shape)
        wplan = self._wplan_get(wkey)
        witems = []

        def wg(n, xt, gt, k, cin, cout):
            dw, db = (self.dwf[n.name], self.dbf[n.name]) if n.op == "conv_bn" else (self.gview[n.name + "/kernels"], self.gview[n.name + "/biases"])
            if wplan is not None:
                wplan.partial(n.name, xt, gt)
                return
            witems.append((n.name, (int(xt.shape[0]), int(xt.shape[1]), int(xt.shape[2]), cin, cout, k), dw, db, None, 0.0))
            ops.conv2d_bwd_filter(xt, gt, k, cin, cout, dw=dw, db=db, grad_scale=gs)

        # ReLU backward without a pass of its own: the launch that delivers the LAST contribution to a ReLU output's gradient
        # also zeroes it where that output is <= 0 (conv2d_bwd_data / scale_mask relu_of) -- the bits of a relu_bwd launch
        # behind it.  left[node] = contributions still to come; a node whose last contribution is a plain alias (add_relu
        # handing its gradient to both summands) keeps its relu_bwd launch.
        left = dict(self._grad_fanin)
        relu_done = set()
        has_relu = lambda node: node.op == "add_relu" or (node.op in ("conv", "conv_bn") and node.attrs["relu"])

        def last_relu(node):
            """Counts one contribution to g[node]; the ReLU output to mask with if it is the last one."""
            left[node] -= 1
            if self.fuse_relu_bwd and left[node] == 0 and has_relu(node):
                relu_done.add(node)
                return val[node]
            return None

        def give(node, dy, packed_bwd):
            """d(input) of a stride-1 conv into g[node] (accumulating when the node already has a gradient)."""
            if node is self.boundary:
                return
            r = last_relu(node)
            if node in g:
                self._before_inplace(g[node])
                ops.conv2d_bwd_data(dy, packed_bwd, dx=g[node], accumulate=True, relu_of=r)
            else:
                g[node] = ops.conv2d_bwd_data(dy, packed_bwd, relu_of=r)

        for n in reversed(self.region):
            gy = g.pop(n)
            if n.op == "conv":
                x = val[n.inputs[0]]
                if n.attrs["relu"] and n not in relu_done:
                    self._before_inplace(gy)
                    ops.relu_bwd(val[n], gy)
                k, cin, cout = n.attrs["size"], int(x.shape[3]), int(n.shape[3])
                self._wgrad(lambda x=x, gy=gy, k=k, cin=cin, cout=cout, n=n: wg(n, x, gy, k, cin, cout), gy, x)
                give(n.inputs[0], gy, self.packplan.bwd.get(n.name))
            elif n.op == "dropout":
                g[n.inputs[0]] = ops.scale_mask(gy, aux[n], 1.0 / n.attrs["keep_prob"], relu_of=last_relu(n.inputs[0]))
            elif n.op == "add_relu":
                if n not in relu_done:
                    self._before_inplace(gy)
                    ops.relu_bwd(val[n], gy)
                sc, br = n.inputs
                g[br] = gy                              # both summands receive the same gradient;
                left[br] -= 1
                if sc is not self.boundary:             # the branch's last d(input) is accumulated into it
                    g[sc] = gy
                    left[sc] -= 1
            elif n.op == "conv_bn":
                x = val[n.inputs[0]]
                if n.attrs["relu"] and n not in relu_done:
                    self._before_inplace(gy)
                    ops.relu_bwd(val[n], gy)
                k, stride = n.attrs["size"], n.attrs["stride"]
                cin, cout = int(x.shape[3]), int(n.shape[3])
                if stride != 1:
                    if k != 1 or n.inputs[0] is not self.boundary:
                        raise SqdetError("ResNet50ConvDetTrainer: strided conv %s needs an input gradient" % n.name)
                    x = ops.subsample_nhwc(x, stride)
                self._wgrad(lambda x=x, gy=gy, k=k, cin=cin, cout=cout, n=n: wg(n, x, gy, k, cin, cout), gy, x)
                if stride == 1:
                    give(n.inputs[0], gy, self.packplan.bwd.get(n.name))
        self._join_wgrad()
        if wplan is not None:
            wplan.reduce(gs)
        elif self.plan_wgrads:
            self._wplan_put(wkey, ops.WgradPlan(witems))
        self.foldplan.run()      # d(kernels), d(gamma), d(beta) of every conv_bn conv from its folded gradients: two launches
        out = collections.OrderedDict(class_loss=losses[0], conf_loss=losses[1], bbox_loss=losses[2], ious=ious, preds=preds,
                                      dpreds=dpreds, num_objects=num_objects)
        if keep_activations:     # names as oracle/resnet_oracle.py forward_train's `override`
            out["activations"] = {n.name: v for n, v in val.items()}
        return out


class GraphedStep:
    """A training step replayed as ONE hipGraph (the reference issues one sess.run per step, train.py:302-304; issuing the
    ~300 launches of a step from Python took longer than the GPU needs to run them).  The graph holds the GPU label build
    (sqdet_build_labels), forward, loss and backward into the flat gradient bucket -- static shapes, no host round trip;
    the inputs are copied into static buffers, the dropout mask is drawn by one launch ahead of the replay, and the
    gradient all-reduce + optimizer step follow it eagerly (so the collective is an ordinary RCCL call).  Anything that
    enters the captured launches BY VALUE (the loss scale) triggers a re-capture when it changes; at most MAX_GRAPHS
    captured graphs are kept (each pins its own static buffers and activation pool), least recently used first out."""

    MAX_GRAPHS = 4

    def __init__(self, trainer, anchors_f64, classes):
        self.tr, self.anchors, self.classes = trainer, anchors_f64, int(classes)
        self.graph, self.static, self.out, self.key = None, None, None, None
        self.cache = {}      # (loss scale, input shape) -> (graph, static buffers, outputs): a scale seen before is not re-captured

    def _capture(self, x, gt, gcls, gcnt):
        tr = self.tr
        st = dict(x=x.clone(), gt=gt.clone(), gcls=gcls.clone(), gcnt=gcnt.clone())
        keep = tr.model.keep_prob
        # the dropout sits in front of the last conv: its input has the last conv's Cin channels on the output grid
        last = tr.model.preds
        din = last.inputs[0]
        dshape = (int(x.shape[0]),) + tuple(int(v) for v in din.get_shape()[1:])
        st["mask"] = torch.ones(dshape, dtype=tr.adt, device=tr.dev)
        # global_num_objects at world > 1: the scalar all-reduce is an ordinary eager RCCL call AHEAD of the replay (step()
        # below fills st["nobj"]); the captured launches read the already-global count from that static buffer
        self.eager_nobj = tr.global_num_objects and tr.world > 1
        if self.eager_nobj:
            st["nobj"] = torch.ones(1, dtype=torch.float32, device=tr.dev)
        def build():       # label build + num_objects = sum(input_mask): on the trainer's side stream, beside the forward
            lab = ops.build_labels(self.anchors, st["gt"], st["gcls"], st["gcnt"], self.classes)[:4]
            return lab, (st["nobj"] if self.eager_nobj else ops.sum_f32(lab[0].reshape(int(x.shape[0]), -1)))

        def run():
            lab, nobj = tr.fork_labels(build)
            return tr.forward_backward(st["x"], *lab, dropout_mask=st["mask"] if keep != 1.0 else None,
                                       num_objects=nobj, num_objects_is_global=self.eager_nobj)
        side = torch.cuda.Stream(device=tr.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()                                     # warm-up on the side stream (allocator pools, lazy one-time set-up)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                out = run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph, self.static, self.out = g, st, out
        self.key = (tr.loss_scale, tuple(x.shape))
        self.cache[self.key] = (g, st, out)
        while len(self.cache) > self.MAX_GRAPHS:          # dict order = insertion / last-use order (step() re-inserts on a hit)
            del self.cache[next(iter(self.cache))]

    def step(self, x, gt, gcls, gcnt, apply_update=True):
        tr = self.tr
        with torch.cuda.device(tr.dev):
            key = (tr.loss_scale, tuple(x.shape))
            if self.key != key:
                if key in self.cache:
                    self.graph, self.static, self.out = self.cache[key]
                    self.cache[key] = self.cache.pop(key)          # most recently used last
                    self.key = key
                    self.eager_nobj = "nobj" in self.static
                else:
                    self._capture(x, gt, gcls, gcnt)
            st = self.static
            st["x"].copy_(x); st["gt"].copy_(gt); st["gcls"].copy_(gcls); st["gcnt"].copy_(gcnt)
            if tr.model.keep_prob != 1.0:
                tr._mask_calls += 1
                ops.dropout_mask_into(st["mask"], tr.model.keep_prob, (tr.seed << 32) + tr._mask_calls)
            if getattr(self, "eager_nobj", False):
                mask = ops.build_labels(self.anchors, st["gt"], st["gcls"], st["gcnt"], self.classes)[0]
                ops.sum_f32(mask, out=st["nobj"])
                reduce_num_objects(st["nobj"], tr.world, tr.pg)
            self.graph.replay()
            tr._finish_step(apply_update)
        return self.out
