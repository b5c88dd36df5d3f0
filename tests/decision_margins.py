"""End-to-end decision margins: image -> picks on the GPU against image -> picks in the CPU oracle.

The forward's float arithmetic differs between the two in the last bits (MFMA accumulation order, device expf), so
end-to-end equality of the DISCRETE outputs -- which 64 anchors enter NMS, which class each has, which boxes NMS keeps,
in which order -- can only be demanded where every decision the reference's filter_prediction takes
(nn_skeleton.py:696-734, util.nms :56-76) has a margin above that float noise.  Per image this module measures
  margins  m_sel  = p[63] - p[64]            gap between the last selected and the first rejected score
           m_ord  = min adjacent gap among the 65 best scores (rank inside a class decides who suppresses whom)
           m_cls  = min over the 64 selected anchors of (best - second best) class probability * confidence
           m_iou  = min over same-class pairs of selected anchors of |IoU - NMS_THRESH|
  noise    n_p    = max |prob_gpu - prob_oracle| over all anchors
           n_iou  = max |IoU_gpu - IoU_oracle| over the same-class selected pairs
and an image is DECIDABLE when every margin exceeds twice the matching noise.  A pair of anchors whose scores are
BITWISE equal on both sides (the exact background ties of the planted head, squeezedet_amd/synthetic.py) is decided by
the tie rule (higher anchor index first) on both sides alike and does not count as a zero margin.  Used by
tests/test_gpu_model.py::test_end_to_end_decision_margins and tools/decision_margins.py (test infrastructure)."""
import numpy as np

from oracle import sqdet_oracle as O


def image_margins(mc, ref, got):
    """ref / got: dicts with det_boxes [A,4], det_probs [A], det_class [A], pred_class_probs [A,C], pred_conf [A] of ONE
    image from the oracle / the device."""
    n = mc.TOP_N_DETECTION
    p = ref["det_probs"].astype(np.float64)
    order = O.rank_order(ref["det_probs"])
    top = p[order[:n + 1]]
    sel = order[:n]
    gaps = top[:-1] - top[1:]
    gp = got["det_probs"][order[:n + 1]]
    exact_tie = (gaps == 0) & (gp[:-1] == gp[1:])          # same bits in the oracle AND on the device: the index decides
    gaps = np.where(exact_tie, np.inf, gaps)
    m_sel = float(gaps[n - 1])
    m_ord = float(np.min(gaps))
    pc = ref["pred_class_probs"][sel].astype(np.float64) * ref["pred_conf"][sel].astype(np.float64)[:, None]
    pcs = np.sort(pc, axis=1)
    m_cls = float(np.min(pcs[:, -1] - pcs[:, -2]))
    n_p = float(np.max(np.abs(got["det_probs"].astype(np.float64) - p)))
    m_iou, n_iou = np.inf, 0.0
    cls = ref["det_class"][sel]
    for c in range(mc.CLASSES):
        idx = sel[cls == c]
        for a in range(len(idx)):
            if a + 1 >= len(idx):
                break
            ir = O.batch_iou(ref["det_boxes"][idx[a + 1:]], ref["det_boxes"][idx[a]]).astype(np.float64)
            ig = O.batch_iou(got["det_boxes"][idx[a + 1:]], got["det_boxes"][idx[a]]).astype(np.float64)
            m_iou = min(m_iou, float(np.min(np.abs(ir - mc.NMS_THRESH))))
            n_iou = max(n_iou, float(np.max(np.abs(ig - ir))))
    decidable = m_sel > 2 * n_p and m_ord > 2 * n_p and m_cls > 2 * n_p and m_iou > 2 * n_iou
    return dict(m_sel=m_sel, m_ord=m_ord, m_cls=m_cls, m_iou=float(m_iou), n_p=n_p, n_iou=n_iou, decidable=bool(decidable))


def run(size, dtype_name, nimg=16, seed=40, device="cuda:0", planted=False, batch=None, pipelined=False, arch="squeezeDet", lanes=None):
    """Runs seeded images through the device path and the oracle; returns (rows, summary) for the first nimg.
    planted: planted-object images + detector channels + head (squeezedet_amd/synthetic.py) on both sides.  batch: device batch
    (>= nimg; default nimg).  pipelined: the device picks come from detect_filter_pipelined (bench.py's step) instead of
    detect -> filter_prediction_batch; lanes: that call is then a DEFERRED one on `lanes` serving lanes, completed by
    flush_pipeline().  arch: "squeezeDet" (size = (H, W)) or "squeezeDet+" (its one size, 375x1242: size is ignored)."""
    import torch

    import squeezedet_amd as S
    from squeezedet_amd import nets
    tdt = torch.float16 if dtype_name == "fp16" else torch.float32
    if arch == "squeezeDet+":
        mc, omc, cls = S.kitti_squeezeDetPlus_config(), O.kitti_squeezeDetPlus_config(), nets.SqueezeDetPlus
        size = (omc.IMAGE_HEIGHT, omc.IMAGE_WIDTH)
    else:
        mc, omc, cls = S.kitti_squeezeDet_config_for_input(*size), O.squeezeDet_config_for_input(*size), nets.SqueezeDet
    mc.LOAD_PRETRAINED_MODEL = False
    batch = batch or nimg
    mc.BATCH_SIZE = batch
    m = cls(mc, gpu_id="0", dtype=tdt)
    params = O.init_params(arch, seed=seed, storage=dtype_name)
    if planted:
        # planted objects (squeezedet_amd/synthetic.py: a data generator, not arithmetic under test): objects in the image, three
        # exact detector channels through every launch, a saturating head -- every decision has a margin by construction
        from squeezedet_amd import synthetic as SY
        params = SY.planted_params(params, omc.ANCHOR_PER_GRID, omc.CLASSES, arch=arch)
        x, _ = SY.planted_images(omc, batch, seed=seed + 1, arch=arch)
    else:
        x = O.synthetic_images(batch, size[0], size[1], seed=seed + 1, storage=dtype_name)
    m.load_params(params)
    xd = x.to(device, tdt)
    outs = m.run([m.det_boxes, m.det_probs, m.det_class, m.pred_class_probs, m.pred_conf], {m.image_input: xd})
    if pipelined and lanes:
        ob, op, oc, oi, cnt = m.detect_filter_pipelined(xd, to_host=True, defer=True, lanes=lanes)   # pinned host rows ...
        m.flush_pipeline()                                                     # ... carried out by the flush, complete after the sync below
    elif pipelined:
        ob, op, oc, oi, cnt = m.detect_filter_pipelined(xd, to_host=True)      # pinned host rows, complete after the sync below
    else:
        ob, op, oc, oi, cnt = m.filter_prediction_batch(outs[0], outs[1], outs[2])
    torch.cuda.synchronize()
    g = [o.cpu().numpy()[:nimg] for o in outs]
    ob, op, oc = ob.cpu().numpy(), op.cpu().numpy(), oc.cpu().numpy()
    oi, cnt = oi.cpu().numpy(), cnt.cpu().numpy()
    _, ref, dets = O.detect(arch, omc, params, x[:nimg], storage=dtype_name)
    rows = []
    for i in range(nimg):
        r = {k: ref[k][i] for k in ("det_boxes", "det_probs", "det_class", "pred_class_probs", "pred_conf")}
        gi = dict(det_boxes=g[0][i], det_probs=g[1][i], det_class=g[2][i])
        row = image_margins(omc, r, gi)
        picks_gpu = oi[i, :cnt[i]].tolist()
        picks_ref = list(dets[i][3])
        inter = len(set(picks_gpu) & set(picks_ref))
        same_cls = oc[i, :cnt[i]].tolist() == [int(c) for c in dets[i][2]]
        same_box = len(picks_gpu) == len(picks_ref) and bool(np.allclose(ob[i, :cnt[i]], np.asarray(dets[i][0], np.float32).reshape(-1, 4), rtol=2e-6, atol=0))   # decoded floats: expf ulp
        strong = [k for k, pr in enumerate(dets[i][1]) if pr > omc.PLOT_PROB_THRESH]      # what demo.py:201-205 draws
        row.update(image=i, same_picks=picks_gpu == picks_ref and same_cls, same_boxes=same_box, n_picks=len(picks_ref), n_strong=len(strong),
                   jaccard=inter / float(max(len(set(picks_gpu) | set(picks_ref)), 1)),
                   same_class=bool(np.array_equal(g[2][i], r["det_class"])))
        rows.append(row)
    dec = [r for r in rows if r["decidable"]]
    summary = dict(size="%dx%d" % size, dtype=dtype_name, images=nimg, decidable=len(dec),
                   decidable_same=sum(r["same_picks"] for r in dec), all_same=sum(r["same_picks"] for r in rows),
                   max_n_p=max(r["n_p"] for r in rows), max_n_iou=max(r["n_iou"] for r in rows),
                   mean_jaccard=float(np.mean([r["jaccard"] for r in rows])),
                   min_m_sel=min(r["m_sel"] for r in rows), min_m_iou=min(r["m_iou"] for r in rows))
    return rows, summary


def format_report(rows, summary):
    lines = ["%s %s: %d images, %d decidable (all margins > 2x noise), picks identical on %d decidable / %d of all; "
             "noise: prob %.3g, IoU %.3g; mean Jaccard overlap of the pick sets %.3f"
             % (summary["size"], summary["dtype"], summary["images"], summary["decidable"], summary["decidable_same"],
                summary["all_same"], summary["max_n_p"], summary["max_n_iou"], summary["mean_jaccard"]),
             "  img  picks same  decidable   m_sel      m_ord      m_cls      m_iou      n_p        n_iou      jaccard"]
    for r in rows:
        lines.append("  %3d  %5d %-5s %-9s %.3e  %.3e  %.3e  %.3e  %.3e  %.3e  %.3f" % (
            r["image"], r["n_picks"], r["same_picks"], r["decidable"], r["m_sel"], r["m_ord"], r["m_cls"], r["m_iou"], r["n_p"], r["n_iou"],
            r["jaccard"]))
    return "\n".join(lines)
