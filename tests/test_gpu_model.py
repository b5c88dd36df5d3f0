"""GPU parity tests, model level: the native plan (sqdet_net_forward) and the builder graph
(ModelSkeleton._conv_layer/_pooling_layer/_fire_layer) against the CPU oracle; the demo.py
call shape; stage-wise bit-exact picks; size-independent properties at BASELINE.json's full
config (batch 32, 375x1242, fp16)."""
import numpy as np
import pytest
import torch

from oracle import sqdet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(arch="squeezeDet", dtype=torch.float32, batch=1, size=None, seed=0):
    import squeezedet_amd as S
    from squeezedet_amd import nets
    if arch == "squeezeDet":
        mc = S.kitti_squeezeDet_config() if size is None else S.kitti_squeezeDet_config_for_input(*size)
        cls = nets.SqueezeDet
    else:
        mc = S.kitti_squeezeDetPlus_config()
        cls = nets.SqueezeDetPlus
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = batch
    m = cls(mc, gpu_id="0", dtype=dtype)
    storage = "fp16" if dtype == torch.float16 else "fp32"
    params = O.init_params(arch, seed=seed, storage=storage)
    m.load_params(params)
    return m, mc, params, storage


OBSERVED = {}   # what -> largest observed max-error / tensor-scale (printed by test_zz_report_observed_errors)


def _check_layers(got, ref, dtype, what):
    got = got.float().cpu().numpy()
    ref = ref.numpy() if isinstance(ref, torch.Tensor) else ref
    assert got.shape == ref.shape, what
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    key = ("fp32 " if dtype == torch.float32 else "fp16 ") + what.split("[")[0]
    OBSERVED[key] = max(OBSERVED.get(key, 0.0), float(err / max(scale, 1e-30)))
    if dtype == torch.float32:
        assert err <= 1e-3 * scale + 1e-5, "%s: max err %g vs scale %g" % (what, err, scale)  # north_star 1e-3 rel
    else:
        # fp16 storage: both sides round every activation to fp16; rounding decisions can flip by one fp16 ulp and
        # propagate.  Observed (test_zz_report_observed_errors, round 2): 5e-4 of the tensor scale after conv1, growing
        # to 2.2e-3 at preds (conv12) -> 5e-3 of the scale allowed (2x the largest observed)
        assert err <= 5e-3 * scale + 1e-3, "%s: max err %g vs scale %g" % (what, err, scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("size", [(384, 1248), (375, 1242)], ids=["384x1248", "375x1242"])
def test_squeezedet_layer_by_layer_vs_oracle(dtype, size):
    """Builder-graph path: every intermediate tensor against the oracle's, so a wrong layer is
    named.  Covers both the reference input (1248x384) and BASELINE.json's 1242x375 with its
    irregular SAME paddings (conv1 (1,1,0,1), pool1 (0,1,1,1), pool5 (1,1,0,1))."""
    m, mc, params, storage = _model("squeezeDet", dtype, 1, size)
    x = O.synthetic_images(1, size[0], size[1], seed=1, storage=storage)
    col = {}
    O.forward("squeezeDet", params, x, storage, collect=col)
    # walk the graph: every conv / pool / concat node by name
    nodes = {}
    stack = [m.preds]
    while stack:
        n = stack.pop()
        if n.name and n.name not in nodes:
            nodes[n.name] = n
        stack.extend(n.inputs)
    names = [k for k in col if k in nodes or (k + "/concat") in nodes]
    fetch = [nodes.get(k, nodes.get(k + "/concat")) for k in names]
    outs = m.run(fetch, {m.image_input: x}, use_plan=False)
    torch.cuda.synchronize()
    for k, o in zip(names, outs):
        _check_layers(o, col[k], dtype, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_native_plan_equals_builder_graph_and_oracle(dtype):
    m, mc, params, storage = _model("squeezeDet", dtype, 2, (375, 1242))
    x = O.synthetic_images(2, 375, 1242, seed=2, storage=storage)
    p_plan = m.run([m.preds], {m.image_input: x}, use_plan=True)[0]
    p_graph = m.run([m.preds], {m.image_input: x}, use_plan=False)[0]
    torch.cuda.synchronize()
    if dtype == torch.float32:
        assert torch.equal(p_plan, p_graph), "native plan and op-by-op graph must run the same kernels"
    else:
        # fp16: the plan's persistent stem (stem3 / stem4.hip, with fire2's squeeze folded in) sums the 27 im2col products in another
        # order inside the MFMA than the strip stem / conv1 -> pool1: 1-ulp flips in ~6e-5 of pool1's elements, which reach preds as
        # the same noise any fp16 rounding difference does (observed 2e-3 of the largest value; the oracle check below allows 5e-3).
        # (Since round 6 the op-by-op graph runs conv1 + pool1 as the fused stem launch too -- whichever kernel "stem_algo" selects.)
        d = (p_plan.float() - p_graph.float()).abs().max().item()
        scale = p_graph.float().abs().max().item()
        assert d <= 4e-3 * scale + 1e-4, "plan vs graph: %g of %g" % (d, scale)
        from squeezedet_amd import ops
        ops.set_option("stem_algo", 2)       # decided at plan creation: a second model whose plan keeps to the strip stem; the graph's
        try:                                 # stem launch follows the knob at run time -- with the strip stem both paths are bitwise equal
            m2 = _model("squeezeDet", dtype, 2, (375, 1242))[0]
            p_strip = m2.run([m2.preds], {m2.image_input: x}, use_plan=True)[0]
            p_graph2 = m2.run([m2.preds], {m2.image_input: x}, use_plan=False)[0]
            torch.cuda.synchronize()
        finally:
            ops.set_option("stem_algo", 0)
        assert torch.equal(p_strip, p_graph2), "native plan (strip stem) and op-by-op graph must run the same kernels"
    ref = O.forward("squeezeDet", params, x, storage)
    _check_layers(p_plan, ref, dtype, "preds")


def test_native_plan_scores_from_the_convdet_epilogue():
    """NetPlan.forward(scores=): the plan's ConvDet launch also writes interpret_output's det_probs (sqdet_net_set_scores);
    preds are bitwise those of the plain forward, scores bitwise sqdet_interpret_output's on those preds; a float32 plan
    reports no support and refuses; unbinding restores the plain launch."""
    from squeezedet_amd import _lib, ops
    m, mc, params, storage = _model("squeezeDet", torch.float16, 3, (375, 1242))
    x = O.synthetic_images(3, 375, 1242, seed=12, storage=storage).to(DEV, torch.float16)
    plan = m._native_plan(3)
    assert plan.scores_supported()
    p0 = plan.forward(x).clone()
    scores = torch.full((3, mc.ANCHORS), -1.0, dtype=torch.float32, device=DEV)
    p1 = plan.forward(x, scores=scores)
    probs = ops.interpret_output(p1, m.anchors_f32(), mc.CLASSES, mc.ANCHOR_PER_GRID, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH)[1]
    torch.cuda.synchronize()
    assert torch.equal(p0, p1) and torch.equal(scores, probs) and float(scores.min()) >= 0.0
    scores.fill_(-1.0)
    p2 = plan.forward(x)                     # unbound again: the buffer is not touched
    torch.cuda.synchronize()
    assert torch.equal(p2, p0) and float(scores.max()) == -1.0
    m32 = _model("squeezeDet", torch.float32, 1, (128, 256))[0]
    plan32 = m32._native_plan(1)
    assert not plan32.scores_supported()
    with pytest.raises(_lib.SqdetUnsupported):
        plan32.forward(torch.zeros((1, 128, 256, 3), device=DEV), scores=torch.zeros((1, plan32.gh * plan32.gw * 9), device=DEV))


@pytest.mark.parametrize("shape", [(3, 200, 600), (1, 97, 333), (5, 130, 236), (2, 64, 1000)], ids=lambda v: "x".join(map(str, v)))
def test_native_plan_other_sizes_and_batches_fp16_vs_oracle(shape):
    """The chained fp16 plan (stem + fire2's squeeze, squeeze-tensor launches through the pools, chain launches with two
    images per workgroup, ConvDet) away from the two benchmark sizes: odd batches, ragged tiles everywhere, an odd width
    (strip stem, no stem-squeeze launch), maps smaller than one tile row -- against the oracle in float16-storage mode."""
    n, h, w = shape
    m, mc, params, storage = _model("squeezeDet", torch.float16, n, (h, w))
    x = O.synthetic_images(n, h, w, seed=7, storage=storage)
    got = m.run([m.preds], {m.image_input: x}, use_plan=True)[0]
    torch.cuda.synchronize()
    ref = O.forward("squeezeDet", params, x, storage)
    _check_layers(got, ref, torch.float16, "preds %dx%dx%d" % shape)
    g2 = m.run([m.preds], {m.image_input: x}, use_plan=False)[0]       # the op-by-op graph agrees to fp16 noise as well
    torch.cuda.synchronize()
    d = (got.float() - g2.float()).abs().max().item()
    assert d <= 4e-3 * g2.float().abs().max().item() + 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_squeezedet_plus_vs_oracle(dtype):
    m, mc, params, storage = _model("squeezeDet+", dtype, 1)
    x = O.synthetic_images(1, 375, 1242, seed=3, storage=storage)
    got = m.run([m.preds], {m.image_input: x})[0]
    torch.cuda.synchronize()
    assert tuple(got.shape) == (1, 22, 76, 72)
    ref = O.forward("squeezeDet+", params, x, storage)
    _check_layers(got, ref, dtype, "squeezeDet+ preds")


def test_demo_call_shape_and_stagewise_bit_exact_picks():
    """demo.py:193-205 shaped caller: sess.run([det_boxes, det_probs, det_class], feed_dict) ->
    model.filter_prediction(...).  Stage-wise parity: the SAME float32
    det_boxes/probs/class fed to the oracle's filter_prediction give identical picks."""
    from squeezedet_amd.nn_skeleton import Session
    m, mc, params, storage = _model("squeezeDet", torch.float32, 1)
    x = O.synthetic_images(1, 384, 1248, seed=4)
    with Session() as sess:
        det_boxes, det_probs, det_class = sess.run([m.det_boxes, m.det_probs, m.det_class],
                                                   feed_dict={m.image_input: [x[0].numpy()]})
    assert det_boxes.shape == (1, 16848, 4) and det_boxes.dtype == np.float32
    assert det_probs.shape == (1, 16848) and det_class.dtype == np.int64
    final_boxes, final_probs, final_class = m.filter_prediction(det_boxes[0], det_probs[0], det_class[0])
    assert isinstance(final_boxes, list) and len(final_boxes) == len(final_probs) == len(final_class) <= 64
    fb, fp, fc = O.filter_prediction(O.kitti_squeezeDet_config(), det_boxes[0], det_probs[0], det_class[0])
    assert final_class == fc
    np.testing.assert_array_equal(np.array(final_probs, np.float32), np.array(fp, np.float32))
    np.testing.assert_array_equal(np.array(final_boxes, np.float32).reshape(-1, 4), np.array(fb, np.float32).reshape(-1, 4))
    keep_idx = [i for i in range(len(final_probs)) if final_probs[i] > mc.PLOT_PROB_THRESH]  # demo.py:201-205
    assert all(0 <= final_class[i] < 3 for i in keep_idx)
    # end-to-end vs the oracle run from the image: decoded outputs within float tolerance
    _, out, _ = O.detect("squeezeDet", O.kitti_squeezeDet_config(), params, x)
    np.testing.assert_allclose(det_probs, out["det_probs"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(det_boxes, out["det_boxes"], rtol=2e-3, atol=5e-2)


def test_full_config_properties_batch32_fp16():
    """BASELINE.json configs[1]: SqueezeDet fp16, batch 32, synthetic 1242x375.  The oracle
    needs seconds per image, so here: (i) images 0 and 17 against the oracle, (ii) batch
    independence -- every image's result equals the same image run in a batch of 1 (bitwise),
    (iii) the post-processing invariants every image must satisfy."""
    m, mc, params, storage = _model("squeezeDet", torch.float16, 32, (375, 1242))
    x = O.synthetic_images(32, 375, 1242, seed=5, storage="fp16")
    xd = x.to(DEV, torch.float16)
    preds = m.run([m.preds], {m.image_input: xd})[0]
    boxes, probs, cls = m.detect(xd)
    ob, op, oc, oi, cnt = m.filter_prediction_batch(boxes, probs, cls)
    torch.cuda.synchronize()
    for i in (0, 17):
        ref = O.forward("squeezeDet", params, x[i:i + 1], "fp16")
        _check_layers(preds[i:i + 1], ref, torch.float16, "preds[%d]" % i)
    m1, _, _, _ = _model("squeezeDet", torch.float16, 1, (375, 1242))
    for i in (3, 31):
        p1 = m1.run([m1.preds], {m1.image_input: xd[i:i + 1].contiguous()})[0]
        assert torch.equal(p1[0], preds[i]), "image %d differs between batch 32 and batch 1" % i
    n = cnt.cpu().numpy()
    assert (n >= 1).all() and (n <= 64).all()
    oc_, op_, oi_ = oc.cpu().numpy(), op.cpu().numpy(), oi.cpu().numpy()
    pr = probs.cpu().numpy()
    for i in range(32):
        k = n[i]
        c = oc_[i, :k]
        assert (np.diff(c) >= 0).all()                                   # ordered by class
        for cc in range(3):
            pp = op_[i, :k][c == cc]
            assert (np.diff(pp) <= 0).all()                              # then descending prob
        assert len(set(oi_[i, :k].tolist())) == k                        # distinct anchors
        thr = np.sort(pr[i])[-64]
        assert (pr[i][oi_[i, :k]] >= thr).all()                          # all from the top-64
        np.testing.assert_array_equal(pr[i][oi_[i, :k]], op_[i, :k])
    # idempotence: filtering the already-filtered detections changes nothing
    fb, fp, fc, fi = O.filter_prediction(O.squeezeDet_config_for_input(375, 1242), boxes[7].cpu().numpy(),
                                         probs[7].cpu().numpy(), cls[7].cpu().numpy(), return_index=True)
    np.testing.assert_array_equal(oi_[7, :n[7]], np.array(fi, np.int32))


def test_pipelined_step_equals_sequential():
    """detect_filter_pipelined (decode + NMS on a side stream behind an event, as bench.py runs the step) returns
    exactly what detect -> filter_prediction_batch returns, step after step with different inputs in flight."""
    m, mc, params, storage = _model("squeezeDet", torch.float16, 4, (375, 1242))
    xs = [O.synthetic_images(4, 375, 1242, seed=s, storage=storage).to(DEV, torch.float16) for s in (3, 4, 5)]
    seq = []
    for x in xs:
        b, p, c = m.detect(x)
        seq.append([t.clone() for t in m.filter_prediction_batch(b, p, c)])
    torch.cuda.synchronize()
    outs = []
    for x in xs + xs:                                           # six steps in flight, no sync in between
        out = m.detect_filter_pipelined(x)                      # (the slot's tensors: valid until the second-next call)
        with torch.cuda.stream(m.post_stream):
            outs.append([t.clone() for t in out])               # snapshot, stream-ordered behind the step's NMS
    torch.cuda.synchronize()
    seq = seq + seq
    for got, want in zip(outs, seq):
        n = want[4].cpu().numpy()
        assert np.array_equal(got[4].cpu().numpy(), n)
        for i in range(4):
            for t in range(4):
                assert torch.equal(got[t][i, :n[i]], want[t][i, :n[i]])


@pytest.mark.parametrize("lanes", [3, 2, 1])
@pytest.mark.parametrize("mode", ["ride", "signal"])
@pytest.mark.parametrize("batch", [32, 5, 1])
def test_deferred_pipelined_step_equals_sequential(mode, batch, lanes, monkeypatch):
    """defer=True (bench.py's step): the decode + filter of call k is carried out by call k+1 of its lane -- "ride": as rider
    workgroups of that forward's fire_chain launches (sqdet_net_set_post_job: same stream, rows written straight to pinned host
    memory; 512-thread form of the filter body); "signal": on the side stream behind that forward's mid-point event, the
    images walked by 16 workgroups -- and flush_pipeline() carries out the last ones.  `lanes` (the explicit argument; 2 is the
    model default, bench.py's sqdet_sample_b1 config runs 3 at batch 1, 1 is the single-stream form): consecutive calls alternate
    between `lanes` serving lanes (plans, HIP streams, nothing ordering them), so a call's rows are carried out by the call
    `lanes` later -- read there, and for the last `lanes` calls after flush_pipeline() + a synchronisation of the CALLER's stream
    only, they must equal detect -> filter_prediction_batch exactly."""
    monkeypatch.setenv("SQDET_POST_DEFER", mode)
    monkeypatch.delenv("SQDET_SERVE_LANES", raising=False)
    m, mc, params, storage = _model("squeezeDet", torch.float16, batch, (375, 1242))
    xs = [O.synthetic_images(batch, 375, 1242, seed=s, storage=storage).to(DEV, torch.float16) for s in (3, 4, 5)]
    seq = []
    for x in xs:
        b, p, c = m.detect(x)
        seq.append([t.clone() for t in m.filter_prediction_batch(b, p, c)])
    torch.cuda.synchronize()
    plan = m._native_plan(batch)
    assert plan.overlap_layer() >= 0 and plan.scores_supported() and plan.rider_capacity() >= batch
    outs, hist = [], []
    for x in xs + xs + xs[:1]:
        hist.append(m.detect_filter_pipelined(x, to_host=True, defer=True, lanes=lanes))
        if len(hist) > lanes:                                   # the rows of the call `lanes` back: enqueued by THIS call
            torch.cuda.synchronize()
            outs.append([t.clone() for t in hist[-1 - lanes]])
    assert (m._lanes is not None and len(m._lanes) == lanes) if lanes > 1 else True
    if lanes > 1:
        chk = m._lane_check                                     # warm_up_lanes ran at the first call: every lane k >= 1 was paired with lane 0
        assert chk is not None and len(chk["pairs"]) == lanes - 1 and chk["forwards_per_sample"] >= 16
    m.flush_pipeline()
    torch.cuda.current_stream().synchronize()                   # the caller's stream alone (flush_pipeline made it wait for every lane)
    for out in hist[-lanes:]:
        outs.append([t.clone() for t in out])
    seq = seq + seq + seq[:1]
    assert len(outs) == 7
    for got, want in zip(outs, seq):
        n = want[4].cpu().numpy()
        assert np.array_equal(got[4].numpy(), n)
        for i in range(batch):
            for t in range(4):
                assert torch.equal(got[t][i, :n[i]], want[t][i, :n[i]].cpu())


@pytest.mark.parametrize("arch,dtype,batch", [("squeezeDet", torch.float32, 3), ("squeezeDet+", torch.float16, 8)], ids=["sqdet-fp32", "plus-fp16-b8"])
def test_deferred_step_on_plans_without_riders_is_complete_after_flush(arch, dtype, batch):
    """Plans that cannot carry riders (float32 SqueezeDet; SqueezeDet+, which has no fire_chain launches -- BASELINE configs[3],
    bench.py's sqdetplus_infer: batch 8, two lanes) run decode + filter + the row copy on each lane's own side stream.
    flush_pipeline() must cover those too: after it, a synchronisation of the caller's stream ALONE makes every row readable, and
    every step's rows equal detect -> filter_prediction_batch bitwise."""
    size = (375, 1242) if arch == "squeezeDet+" else (128, 256)
    m, mc, params, storage = _model(arch, dtype, batch, None if arch == "squeezeDet+" else size)
    xs = [O.synthetic_images(batch, size[0], size[1], seed=s, storage=storage).to(DEV, dtype) for s in (21, 22, 23)]
    seq = []
    for x in xs:
        b, p, c = m.detect(x)
        seq.append([t.clone() for t in m.filter_prediction_batch(b, p, c)])
    torch.cuda.synchronize()
    assert m._native_plan(batch).rider_capacity() < batch
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                               # a caller on its own stream
        hist = [m.detect_filter_pipelined(x, to_host=True, defer=True) for x in xs]        # (default lane count: 2)
        assert m._lanes is not None and len(m._lanes) == 2 and all(ln["post_stream"] is not None for ln in m._lanes)
        m.flush_pipeline()
        side.synchronize()                                      # NOT a device-wide synchronize
        outs = [[t.clone() for t in out] for out in hist[-2:]]  # (slot reuse: the rows of the last `lanes` x 2 calls are live)
    for got, want in zip(outs, seq[-2:]):
        n = want[4].cpu().numpy()
        assert np.array_equal(got[4].numpy(), n) and (n >= 1).all()
        for i in range(batch):
            for t in range(4):
                assert torch.equal(got[t][i, :n[i]], want[t][i, :n[i]].cpu())


def test_squeezedet_plus_serving_step_picks_identical_planted_objects(capsys):
    """BASELINE configs[3] as bench.py runs it (sqdetplus_infer: SqueezeDet+ float16, batch 8, detect_filter_pipelined(defer=True) on two
    lanes) pinned image -> picks on the 22x76 / 15048-anchor grid against the float16-storage oracle, on planted objects
    (squeezedet_amd/synthetic.py, SqueezeDet+ geometry: 7x7 objects through the 7x7/s2 VALID stem and the VALID pools): all 8
    images decidable, anchor indices in output order / classes identical, boxes to 2e-6."""
    from tests import decision_margins as DM
    rows, summary = DM.run(None, "fp16", nimg=8, seed=60, planted=True, batch=8, pipelined=True, arch="squeezeDet+", lanes=2)
    with capsys.disabled():
        print("\n[planted objects, SqueezeDet+] " + DM.format_report(rows, summary))
    for r in rows:
        assert r["decidable"], "image %d: a decision margin is within the measured noise: %r" % (r["image"], r)
        assert r["same_picks"] and r["same_boxes"], "image %d: the picks differ: %r" % (r["image"], r)
        assert r["n_strong"] >= 8 and r["m_iou"] >= 0.03, r
    assert summary["all_same"] == 8 and summary["decidable"] == 8, summary


def test_pipelined_default_path_preds_vs_oracle_random_weights():
    """The serving step's OWN forward -- the default kernels of the float16 plan (DMA stem, fire_dma, fire_chain, ConvDet's SCORE
    form) as detect_filter_pipelined(defer=True) launches them on its lanes, batch 32, random weights (every channel matters, unlike
    the planted head) -- against the float16-storage oracle at preds level: the slot's preds tensor of images 0 and 17 within the
    float16 tolerance, and bitwise equal to the plain plan forward."""
    m, mc, params, storage = _model("squeezeDet", torch.float16, 32, (375, 1242))
    x = O.synthetic_images(32, 375, 1242, seed=5, storage="fp16")
    xd = x.to(DEV, torch.float16)
    for _ in range(2):                                           # one call per lane
        m.detect_filter_pipelined(xd, to_host=True, defer=True)
    m.flush_pipeline()
    torch.cuda.synchronize()
    plain = m.run([m.preds], {m.image_input: xd})[0]
    torch.cuda.synchronize()
    for lane in m._lanes:
        used = [s for s in lane["pipe"]["slots"] if s.get("ride")]
        assert len(used) == 1
        preds = used[0]["preds"]
        assert torch.equal(preds, plain)
        for i in (0, 17):
            ref = O.forward("squeezeDet", params, x[i:i + 1], "fp16")
            _check_layers(preds[i:i + 1], ref, torch.float16, "preds[%d] (pipelined, lane %d)" % (i, lane["which"]))


def test_post_job_riders_equal_the_filter_launch():
    """sqdet_net_set_post_job: the previous batch's decode + filter as riders of the next forward's fire_chain launches gives
    exactly sqdet_detect_filter's outputs (device outputs here), is consumed by ONE forward, and a plan without fire_chain
    launches (float32) refuses."""
    from squeezedet_amd import _lib, ops
    m, mc, params, storage = _model("squeezeDet", torch.float16, 32, (375, 1242))
    plan = m._native_plan(32)
    assert plan.rider_capacity() == 96                                   # 6 fire_chain launches x 16 idle CUs
    x = O.synthetic_images(32, 375, 1242, seed=8, storage=storage).to(DEV, torch.float16)
    scores = torch.empty((32, mc.ANCHORS), dtype=torch.float32, device=DEV)
    preds = plan.forward(x, scores=scores).clone()
    want = ops.detect_filter(preds, m.anchors_f32(), mc.CLASSES, mc.ANCHOR_PER_GRID, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH,
                             mc.TOP_N_DETECTION, mc.NMS_THRESH, scratch=scores, scores_ready=True)
    out = [torch.full_like(t, -7) for t in want]
    plan.set_post_job(preds, scores, m.anchors_f32(), out, mc.CLASSES, mc.ANCHOR_PER_GRID, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT,
                      mc.EXP_THRESH, mc.TOP_N_DETECTION, mc.NMS_THRESH)
    p2 = plan.forward(x)                                                  # carries the riders
    torch.cuda.synchronize()
    assert torch.equal(p2, preds)
    for g_, w_ in zip(out, want):
        assert torch.equal(g_, w_)
    for t in out:
        t.fill_(-7)
    plan.forward(x)                                                       # one-shot: no riders this time
    torch.cuda.synchronize()
    assert all(int((t == -7).all()) for t in out)
    m32 = _model("squeezeDet", torch.float32, 2, (128, 256))[0]
    assert m32._native_plan(2).rider_capacity() == 0
    with pytest.raises(_lib.SqdetUnsupported):
        m32._native_plan(2).set_post_job(preds[:2], scores[:2], m.anchors_f32(), [t[:2] for t in out], mc.CLASSES, mc.ANCHOR_PER_GRID,
                                         mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH, mc.TOP_N_DETECTION, mc.NMS_THRESH)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("size", [(384, 1248), (375, 1242)], ids=["384x1248", "375x1242"])
def test_end_to_end_decision_margins(size, dtype, capsys):
    """image -> picks on the device against image -> picks in the oracle for 16 seeded images.  Where
    every decision of filter_prediction (top-64 boundary, rank inside the class, class arg-max, IoU vs NMS_THRESH) has
    a margin above twice the measured float noise, the picks -- anchor indices, in output order -- must be IDENTICAL;
    the margins are printed (pytest -s, or tools/decision_margins.py)."""
    from tests import decision_margins as DM
    rows, summary = DM.run(size, dtype, nimg=16, seed=40)
    with capsys.disabled():
        print("\n" + DM.format_report(rows, summary))
    for r in rows:
        if r["decidable"]:
            assert r["same_picks"], "image %d: margins above the noise but the picks differ: %r" % (r["image"], r)
    # The comparison must not be vacuous in float32: about half of the images are decidable (observed 9 of 16 at noise
    # 1.8e-6, and all 16 pick identically).  In float16 the score noise (~1e-3: one-ulp activation flips propagate) is
    # ABOVE the gap between the 64th and 65th of 16848 random-weight scores (~1e-4), so no image is decidable end to
    # end -- stated, not hidden: the float16 path is pinned stage-wise instead (its own decoded arrays through the
    # oracle's filter_prediction give identical picks: test_full_config_properties_batch32_fp16, smoke()).
    if dtype == "fp32":
        assert summary["decidable"] >= 6 and summary["all_same"] >= summary["decidable"], summary
        assert summary["max_n_p"] < 2e-5, summary
    else:
        # (random-weight float16: only the noise bound -- the pick identity of the float16 path is asserted on the
        # planted-object head below, where the decisions have margins)
        assert summary["max_n_p"] < 1e-2, summary


@pytest.mark.parametrize("size", [(375, 1242), (384, 1248)], ids=["375x1242", "384x1248"])
def test_fp16_headline_path_picks_identical_planted_objects(size, capsys):
    """The benchmarked path pinned end to end (north_star: bit-exact anchor indices / NMS picks): float16, batch 32,
    through detect_filter_pipelined (bench.py's step: forward + decode + top-N + NMS + rows to pinned host memory), against
    the float16-storage oracle, image -> picks, on planted objects (squeezedet_amd/synthetic.py: 12 objects per image in the
    IMAGE, carried by three exact detector channels through every launch of the forward, a head that saturates on them -- 36
    anchors per image score 0.58-0.91, all others tie exactly at their shape's background level; object layouts keep every
    same-class IoU >= 0.04 away from NMS_THRESH).  EVERY one of the 16 compared images must be decidable (all margins above
    twice the measured noise) and give identical anchor indices in output order, identical classes, and boxes equal to 2e-6
    relative (device expf against glibc's)."""
    from tests import decision_margins as DM
    rows, summary = DM.run(size, "fp16", nimg=16, seed=40, planted=True, batch=32, pipelined=True)
    with capsys.disabled():
        print("\n[planted objects] " + DM.format_report(rows, summary))
    for r in rows:
        assert r["decidable"], "image %d: a decision margin is within the measured noise: %r" % (r["image"], r)
        assert r["same_picks"] and r["same_boxes"], "image %d: the picks differ: %r" % (r["image"], r)
        assert r["n_strong"] >= 8, r                       # the comparison is not vacuous: the planted objects were found
        assert r["m_iou"] >= 0.03, r                       # (layout margin 0.04 minus the +1 of bbox_transform_inv's width)
    assert summary["all_same"] == 16 and summary["decidable"] == 16, summary


def test_zz_report_observed_errors(capsys):
    """Prints the largest observed error / tensor scale of every _check_layers comparison of this module (runs last)."""
    with capsys.disabled():
        print("\nobserved max-error / tensor-scale vs the oracle:")
        for k in sorted(OBSERVED):
            print("  %-40s %.3e" % (k, OBSERVED[k]))


def test_sample_png_config0_picks_vs_oracle():
    """BASELINE configs[0]: the reference's data/sample.png (tests/golden/sample.png) through demo.py:186-199's path -- BGR
    uint8 -> sqdet_preprocess_bgr (resize to 1248x384, mean subtraction) -> float32 SqueezeDet at batch 1 -> picks -- against
    the oracle on the same file (its own preprocessing restatement, forward, interpret_output, filter_prediction).  The
    prepared input agrees to float32 rounding, preds to 1e-3 (north_star), and the picks are identical when every decision has
    a margin above the measured noise (else their overlap is reported and must still be high)."""
    import os
    from PIL import Image
    from oracle import preproc_oracle as PO
    from squeezedet_amd import ops
    from tests import decision_margins as DM
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample.png")
    bgr = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    assert bgr.shape == (375, 1242, 3)
    m, mc, params, storage = _model("squeezeDet", torch.float32, 1, None)
    omc = O.kitti_squeezeDet_config()
    x_ref = PO.preprocess_bgr(bgr, omc.IMAGE_HEIGHT, omc.IMAGE_WIDTH, omc.BGR_MEANS)[None]
    x_dev = ops.preprocess_bgr(torch.from_numpy(bgr).to(DEV)[None], mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, mc.BGR_MEANS, torch.float32)
    np.testing.assert_allclose(x_dev.cpu().numpy(), x_ref, rtol=0, atol=2e-4)
    outs = m.run([m.preds, m.det_boxes, m.det_probs, m.det_class], {m.image_input: x_dev})
    ob, op, oc, oi, cnt = m.filter_prediction_batch(outs[1], outs[2], outs[3])
    torch.cuda.synchronize()
    preds_ref, ref, dets = O.detect("squeezeDet", omc, params, torch.from_numpy(x_ref))
    err = np.abs(outs[0].cpu().numpy() - preds_ref).max() / np.abs(preds_ref).max()
    assert err < 1e-3, err
    r = {k: ref[k][0] for k in ("det_boxes", "det_probs", "det_class", "pred_class_probs", "pred_conf")}
    g = dict(det_boxes=outs[1][0].cpu().numpy(), det_probs=outs[2][0].cpu().numpy(), det_class=outs[3][0].cpu().numpy())
    row = DM.image_margins(omc, r, g)
    got, want = oi[0, :int(cnt[0])].cpu().tolist(), list(dets[0][3])
    jac = len(set(got) & set(want)) / float(max(len(set(got) | set(want)), 1))
    if row["decidable"]:
        assert got == want, (row, got, want)
    assert jac > 0.9, (row, jac)
