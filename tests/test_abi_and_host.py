"""CPU-side checks: the C-ABI library loads and exports every symbol include/sqdet.h declares
(no compute calls without a GPU), argument validation returns error codes instead of
crashing, and the host-side mirror (config, graph builders, counters) matches the reference's
numbers."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import squeezedet_amd as S
from squeezedet_amd import _lib, nets
from squeezedet_amd import build as sqbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sqbuild.build(verbose=False)
    return _lib.lib()


def header_functions():
    src = open(os.path.join(ROOT, "include", "sqdet.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sqdet_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol(lib):
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libsqdet_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "ctypes binding missing for %s" % n
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.sqdet_version()


def test_argument_validation_returns_codes(lib):
    assert lib.sqdet_conv2d_nhwc_fwd(None, None, None, None, 1, 8, 8, 8, 8, 3, 1, 0, 1, 1, 8, 0, None) == -1
    assert b"null" in lib.sqdet_last_error()
    assert lib.sqdet_maxpool_nhwc_fwd(None, None, 1, 8, 8, 8, 3, 2, 0, 1, None) == -1
    h = C.c_void_p()
    assert lib.sqdet_net_create(C.byref(h), 7, 1, 1, 384, 1248, 3, 9) == -1      # bad arch
    assert lib.sqdet_net_create(C.byref(h), 0, 5, 1, 384, 1248, 3, 9) == -1      # bad dtype
    assert lib.sqdet_conv_packed_bytes(0, 3, 3, 1) == 0


def test_many_table_builders_are_host_side(lib):
    """The `*_prepare` halves of the one-launch-for-many entry points (weight-gradient slab reduction, packing with the batch norm
    folded, fold backward) are host functions: table sizes, block ranges and argument validation without a device."""
    n = 3
    arr = lambda vals, ct: (ct * n)(*vals)
    fake = lambda k: arr([0x10000 * (k + i + 1) for i in range(n)], C.c_void_p)      # (never dereferenced on the host)
    nul = arr([None] * n, C.c_void_p)
    dims = [arr(v, C.c_int) for v in ([20, 20, 8], [24, 47, 24], [78, 156, 78], [768, 32, 1024], [72, 128, 256], [3, 3, 1])]
    # --- sqdet_slab_reduce_many_prepare
    tb = lib.sqdet_slab_reduce_many_table_bytes(n)
    assert tb > 0 and tb % n == 0 and lib.sqdet_slab_reduce_many_table_bytes(0) == 0
    host, blocks = (C.c_ubyte * tb)(), C.c_int()
    dec = arr([0.0, 1e-4, 0.0], C.c_float)
    assert lib.sqdet_slab_reduce_many_prepare(fake(0), fake(10), fake(20), nul, dec, *dims, n, host, C.byref(blocks)) == 0
    want = 0
    for i in range(n):
        cnt = dims[5][i] ** 2 * dims[3][i] * dims[4][i] + dims[4][i]
        want += min((cnt + 63) // 64, 4096)
        ws = lib.sqdet_conv2d_bwd_filter_workspace_bytes(*[d[i] for d in dims])
        assert ws > 0 and ws % (4 * cnt) == 0                     # ksplit whole slabs of k*k*cin*cout + cout floats
    assert blocks.value == want
    bad = arr([3, 5, 1], C.c_int)                                  # k = 5 is not a trainable conv of these nets
    assert lib.sqdet_slab_reduce_many_prepare(fake(0), fake(10), fake(20), nul, dec, *dims[:5], bad, n, host, C.byref(blocks)) == -1
    # --- sqdet_conv_pack_many_prepare_bn: plain items and folded items in one table
    tb = lib.sqdet_conv_pack_many_table_bytes(n)
    host, blocks = (C.c_ubyte * tb)(), C.c_int()
    ks, cis, cos, bw = dims[5], dims[3], dims[4], arr([0, 1, 0], C.c_int)
    gam = arr([0x500000, None, 0x510000], C.c_void_p)
    assert lib.sqdet_conv_pack_many_prepare_bn(fake(0), fake(10), ks, cis, cos, bw, gam, fake(30), fake(40), fake(50), nul, fake(60),
                                               1e-5, n, _lib.F16, host, C.byref(blocks)) == 0
    assert blocks.value > 0
    b2 = C.c_int()
    assert lib.sqdet_conv_pack_many_prepare(fake(0), fake(10), ks, cis, cos, bw, n, _lib.F16, host, C.byref(b2)) == 0
    assert b2.value == blocks.value                                 # the fold changes what is written, not the grid
    # --- sqdet_fold_batchnorm_bwd_many_prepare
    tb = lib.sqdet_fold_batchnorm_bwd_many_table_bytes(n)
    host, b1, bf = (C.c_ubyte * tb)(), C.c_int(), C.c_int()
    assert lib.sqdet_fold_batchnorm_bwd_many_prepare(fake(0), fake(10), fake(20), nul, fake(30), fake(40), fake(50), fake(60), fake(70),
                                                     fake(80), fake(90), ks, cis, cos, n, host, C.byref(b1), C.byref(bf)) == 0
    rows = [ks[i] ** 2 * cis[i] for i in range(n)]
    assert b1.value == sum(((cos[i] + 63) // 64) * ((rows[i] + 31) // 32) for i in range(n))
    assert bf.value == sum((cos[i] + 255) // 256 for i in range(n))
    for i in range(n):
        assert lib.sqdet_fold_batchnorm_bwd_workspace_bytes(ks[i], cis[i], cos[i]) == ((rows[i] + 31) // 32) * cos[i] * 4


def test_net_plan_tables_without_gpu(lib):
    """The plan is host-side: parameter table, workspace sizes and the layer table can be
    inspected without a device."""
    h = C.c_void_p()
    # default plan: conv1+pool1 fused; one launch per fire module on the large few-channel maps (fire2-5, the persistent
    # streaming kernel); the late 24x78 modules as a CHAIN: fire6's squeeze, then expand_i + squeeze_{i+1} per launch
    def layer_names():
        nm = C.create_string_buffer(128)
        out = []
        for i in range(lib.sqdet_net_num_layers(h)):
            assert lib.sqdet_net_layer_info(h, i, nm, 128, None, None) == 0
            out.append(nm.value.decode())
        return out
    assert lib.sqdet_net_create(C.byref(h), _lib.ARCH_SQUEEZEDET, _lib.F16, 32, 375, 1242, 3, 9) == 0
    names_default = layer_names()
    # conv1 to fire11 is ONE run: between its launches only squeeze tensors travel (through the pools too) -- the stem launch
    # ends with fire2's squeeze1x1
    assert names_default == ["conv1+pool1+fire2/squeeze1x1", "fire2/expand+fire3/squeeze1x1", "fire3/expand+pool3+fire4/squeeze1x1", "fire4/expand+fire5/squeeze1x1",
                             "fire5/expand+pool5+fire6/squeeze1x1", "fire6/expand+fire7/squeeze1x1", "fire7/expand+fire8/squeeze1x1",
                             "fire8/expand+fire9/squeeze1x1", "fire9/expand+fire10/squeeze1x1", "fire10/expand+fire11/squeeze1x1",
                             "fire11/expand", "conv12"]
    tot_f = C.c_double()
    fsum = 0.0
    for i in range(lib.sqdet_net_num_layers(h)):
        assert lib.sqdet_net_layer_info(h, i, None, 0, C.byref(tot_f), None) == 0
        fsum += tot_f.value
    assert abs(fsum / 32 / 1e9 - 10.492) < 0.01     # the chained plan does the same arithmetic: GFLOP / image @375x1242
    lib.sqdet_net_destroy(h)
    # "fire_fuse" = 9: the stem keeps to conv1 + pool1 and fire2 runs whole from pool1's tensor (the round-2 'b' plan)
    assert lib.sqdet_set_option(b"fire_fuse", 9) == 0
    assert lib.sqdet_net_create(C.byref(h), _lib.ARCH_SQUEEZEDET, _lib.F16, 32, 375, 1242, 3, 9) == 0
    assert lib.sqdet_set_option(b"fire_fuse", 0) == 0
    assert layer_names()[:3] == ["conv1+pool1", "fire2+fire3/squeeze1x1", "fire3/expand+pool3+fire4/squeeze1x1"]
    lib.sqdet_net_destroy(h)
    # "fire_fuse" = 8: no streaming expand + next-squeeze launches -- the pooled modules end their runs (round-2 midpoint plan)
    assert lib.sqdet_set_option(b"fire_fuse", 8) == 0
    assert lib.sqdet_net_create(C.byref(h), _lib.ARCH_SQUEEZEDET, _lib.F16, 32, 375, 1242, 3, 9) == 0
    assert lib.sqdet_set_option(b"fire_fuse", 0) == 0
    assert layer_names()[:6] == ["conv1+pool1", "fire2+fire3/squeeze1x1", "fire3/expand+pool3", "fire4+fire5/squeeze1x1", "fire5/expand+pool5",
                                 "fire6/squeeze1x1"]
    lib.sqdet_net_destroy(h)
    # without the chains ("fire_fuse" = 5): one launch per fire module everywhere (the tile kernel on the late maps);
    # float32 plans never chain (the chain kernel is float16)
    assert lib.sqdet_set_option(b"fire_fuse", 5) == 0
    assert lib.sqdet_net_create(C.byref(h), _lib.ARCH_SQUEEZEDET, _lib.F16, 32, 375, 1242, 3, 9) == 0
    assert lib.sqdet_set_option(b"fire_fuse", 0) == 0
    names_nochain = layer_names()
    assert "fire6" in names_nochain and "fire11" in names_nochain and len(names_nochain) == 34 - 2 * 10 - 2
    assert "fire2" in names_nochain and "fire3+pool3" in names_nochain and "fire5+pool5" in names_nochain and "pool3" not in names_nochain
    lib.sqdet_net_destroy(h)
    assert lib.sqdet_net_create(C.byref(h), _lib.ARCH_SQUEEZEDET, _lib.F32, 2, 384, 1248, 3, 9) == 0
    assert not any("+fire" in n or n.endswith("/expand") for n in layer_names())
    lib.sqdet_net_destroy(h)
    # the rest of this test inspects the per-conv plan (fire fusion off)
    assert lib.sqdet_set_option(b"fire_fuse", 2) == 0
    assert lib.sqdet_net_create(C.byref(h), _lib.ARCH_SQUEEZEDET, _lib.F16, 32, 375, 1242, 3, 9) == 0
    assert lib.sqdet_set_option(b"fire_fuse", 0) == 0
    gh, gw, ch = C.c_int(), C.c_int(), C.c_int()
    assert lib.sqdet_net_output_dims(h, C.byref(gh), C.byref(gw), C.byref(ch)) == 0
    assert (gh.value, gw.value, ch.value) == (24, 78, 72)
    name = C.create_string_buffer(128)
    shape = (C.c_int * 4)()
    nd = C.c_int()
    names, nparam = [], 0
    for i in range(lib.sqdet_net_num_params(h)):
        assert lib.sqdet_net_param_info(h, i, name, 128, shape, C.byref(nd)) == 0
        names.append(name.value.decode())
        nparam += int(np.prod([shape[j] for j in range(nd.value)]))
    assert nparam == 2082120                       # SqueezeDet parameters
    assert names[0] == "conv1/kernels" and names[-1] == "conv12/biases" and "fire7/expand3x3/kernels" in names
    fl, by = C.c_double(), C.c_double()
    tot_f = tot_b = 0.0
    nl = lib.sqdet_net_num_layers(h)
    assert nl == 34                                # 32 convs + 3 pools, conv1+pool1 fused into one launch
    for i in range(nl):
        assert lib.sqdet_net_layer_info(h, i, name, 128, C.byref(fl), C.byref(by)) == 0
        tot_f += fl.value
        tot_b += by.value
    assert abs(tot_f / 32 / 1e9 - 10.492) < 0.01   # GFLOP / image @375x1242
    # MB / image fp16, every tensor touched once: 133.2 MB at batch 1; at batch 32 the
    # 4.16 MB of weights are read once per launch, not once per image
    fused_saving = 2 * 188 * 621 * 64 * 2 / 1e6    # conv1's activations are neither written nor re-read
    assert abs(tot_b / 32 / 1e6 - (133.244 - 4.164 * 31 / 32 - fused_saving)) < 0.05
    assert lib.sqdet_net_workspace_bytes(h) > 32 * 188 * 621 * 64 * 2
    x = C.c_void_p(256)
    assert lib.sqdet_net_forward(h, x, x, None) == -4    # SQDET_ESTATE: not bound
    lib.sqdet_net_destroy(h)
    assert lib.sqdet_net_create(C.byref(h), _lib.ARCH_SQUEEZEDET_PLUS, _lib.F16, 1, 375, 1242, 3, 9) == 0
    assert lib.sqdet_net_output_dims(h, C.byref(gh), C.byref(gw), C.byref(ch)) == 0
    assert (gh.value, gw.value) == (22, 76)
    lib.sqdet_net_destroy(h)


def test_packed_bytes_geometry(lib):
    # [groups][steps][NT][64 lanes][16 B]
    assert lib.sqdet_conv_packed_bytes(1, 64, 16, _lib.F16) == 1 * 2 * 1 * 1024       # K=64 -> 2 chunk-steps, NT=1
    assert lib.sqdet_conv_packed_bytes(3, 96, 384, _lib.F16) == 4 * 27 * 6 * 1024     # 9 taps x 3 chunks, NT=6 x 4 groups
    assert lib.sqdet_conv_packed_bytes(3, 3, 64, _lib.F16) == 1 * 1 * 4 * 1024        # gather: K'=27 -> 1 step
    assert lib.sqdet_conv_packed_bytes(3, 768, 72, _lib.F32) == 1 * (9 * 48) * 5 * 1024
    assert lib.sqdet_conv_packed_bytes(7, 3, 96, _lib.F32) == 1 * 10 * 6 * 1024       # K'=147 -> 10 steps of 16


def test_config_matches_reference_fields(golden_dir):
    import hashlib
    g = np.load(os.path.join(golden_dir, "anchors.npz"))
    for key, fn in (("squeezeDet", S.kitti_squeezeDet_config), ("squeezeDetPlus", S.kitti_squeezeDetPlus_config),
                    ("res50", S.kitti_res50_config)):
        mc = fn()
        ab = mc.ANCHOR_BOX
        assert ab.dtype == np.float64 and mc.ANCHORS == len(ab) and mc.ANCHOR_PER_GRID == 9
        assert hashlib.sha256(np.ascontiguousarray(ab).tobytes()).hexdigest() == str(g["anchors_%s_sha256" % key])
    mc = S.kitti_squeezeDet_config()
    assert (mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.TOP_N_DETECTION, mc.NMS_THRESH, mc.PROB_THRESH, mc.PLOT_PROB_THRESH,
            mc.EXP_THRESH, mc.CLASSES) == (1248, 384, 64, 0.4, 0.005, 0.4, 1.0, 3)
    assert mc.CLASS_NAMES == ("car", "pedestrian", "cyclist")
    mc2 = S.kitti_squeezeDet_config_for_input(375, 1242)
    assert mc2.ANCHORS == 16848 and mc2.ANCHOR_BOX[0][0] == 1242.0 / 79 and mc2.ANCHOR_BOX[0][1] == 375.0 / 25


def test_graph_builders_and_reference_counters():
    """The builder graph reproduces the reference's analytical counters
    (nn_skeleton.py:549-561): params (1+k^2 C)F, flops (1+2Ck^2)FHW (+2FHW relu)."""
    mc = S.kitti_squeezeDet_config()
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = 1
    m = nets.SqueezeDet(mc, gpu_id="0")
    assert m.preds.get_shape() == (1, 24, 78, 72)
    assert m.det_boxes.get_shape() == (1, 16848, 4) and m.det_class.get_shape() == (1, 16848)
    assert sum(c for _, c in m.model_size_counter) == 2082120
    assert abs(sum(c for _, c in m.flop_counter) / 1e9 - 10.649) < 1e-3     # BASELINE.md, reference formula
    assert len(m.model_params) == 64 and list(m.params)[:2] == ["conv1/kernels", "conv1/biases"]
    assert tuple(m.params["fire2/expand3x3/kernels"].shape) == (3, 3, 16, 64)
    assert m.trainable["conv1/kernels"] is False and m.trainable["conv12/kernels"] is True   # conv1 frozen
    k = m.params["fire2/squeeze1x1/kernels"]
    assert float(k.abs().max()) <= 0.02 + 1e-6 and 0.005 < float(k.std()) < 0.012               # trunc normal, sigma 0.01
    mcp = S.kitti_squeezeDetPlus_config()
    mcp.LOAD_PRETRAINED_MODEL = False
    mp = nets.SqueezeDetPlus(mcp)
    assert sum(c for _, c in mp.model_size_counter) == 7021640
    assert abs(sum(c for _, c in mp.flop_counter) / 1e9 - 77.246) < 1e-3
    if not torch.cuda.is_available():
        with pytest.raises(_lib.SqdetError):   # no CPU execution path
            m.run([m.preds], {m.image_input: np.zeros((1, 384, 1248, 3), np.float32)})
        with pytest.raises(_lib.SqdetError):
            m.filter_prediction(np.zeros((10, 4), np.float32), np.zeros(10, np.float32), np.zeros(10, np.int64))


def test_loss_scale_bookkeeping_state_machine():
    """Mixed-precision trainers: the host-side bookkeeping of the dynamic loss scale (the skipped update itself happens
    inside the optimizer kernel) -- halve on overflow down to 2^-14, double after `growth_interval` clean steps up to
    65536, global_step counts applied steps only."""
    from squeezedet_amd.train import _TrainerBase
    tr = _TrainerBase.__new__(_TrainerBase)          # no device needed for the state machine
    tr.loss_scale, tr.growth_interval, tr._clean_steps, tr.skipped_steps, tr.global_step = 1024.0, 3, 0, 0, 0
    tr.half = True
    tr._account(True)
    assert (tr.loss_scale, tr.skipped_steps, tr.global_step, tr._clean_steps) == (512.0, 1, 0, 0)
    for _ in range(2):
        tr._account(False)
    assert (tr.loss_scale, tr.global_step, tr._clean_steps) == (512.0, 2, 2)
    tr._account(False)                               # third clean step: the scale grows, the counter restarts
    assert (tr.loss_scale, tr.global_step, tr._clean_steps) == (1024.0, 3, 0)
    tr._account(True)
    assert tr._clean_steps == 0 and tr.loss_scale == 512.0 and tr.skipped_steps == 2
    tr.loss_scale = 2.0 ** -14
    tr._account(True)
    assert tr.loss_scale == 2.0 ** -14               # floor
    tr.loss_scale, tr.growth_interval = 65536.0, 1
    tr._account(False)
    assert tr.loss_scale == 65536.0                  # ceiling
    tr._pending_flag = None
    tr.flush()                                       # nothing pending: a no-op
    assert tr.global_step == 4
    # float32 training has no scale to lower: clean steps just count, a non-finite gradient norm is an error (the
    # reference asserts on a NaN loss, train.py:313) -- the kernel has skipped the update, the caller is told
    tr.half, tr.loss_scale = False, 1.0
    for _ in range(5):
        tr._account(False)
    assert tr.global_step == 9 and tr.loss_scale == 1.0
    with pytest.raises(FloatingPointError):
        tr._account(True)
    assert tr.skipped_steps == 4 and tr.global_step == 9


def test_bench_launch_roofline_picks_the_bound_by_intensity():
    """bench.py's roofline entry of one launch: MFMA-bound above the ridge (2.5 PF/s / 8 TB/s = 312 flop/B) in float16,
    else HBM-bound; achieved = algorithmic work / measured duration."""
    import bench
    r = bench.launch_roofline(59.6e9, 102e6, 0.0632, "fp16")          # ConvDet: 584 flop/B
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["achieved"] - 59.6e9 / 0.0632e-3 / 1e12) < 1e-2
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    r = bench.launch_roofline(14.8e9, 119e6, 0.0735, "fp16")          # the stem launch: 124 flop/B
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["achieved"] - 119e6 / 0.0735e-3 / 1e9) < 1e-1
    assert bench.launch_roofline(59.6e9, 102e6, 0.0632, "fp32")["bound"] == "hbm"   # the dense peak quoted is float16's


def test_one_hip_runtime_whatever_the_import_order():
    """libsqdet_hip.so loaded BEFORE torch must not pull the system libamdhip64 in beside torch's bundled one (two runtimes:
    the second sees no device -- `python __graft_entry__.py smoke` after build() failed that way): _lib.lib() imports
    torch first.  Checked in a fresh interpreter, library first."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from squeezedet_amd import _lib\n_lib.lib()\nimport torch\n"
            "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == "1", r.stdout


def test_bench_rocprof_launch_ms_reads_only_a_fingerprinted_profile(tmp_path, monkeypatch):
    """bench.rocprof_launch_ms: a kernel's average duration is quoted from profiles/<tag>_kernel_stats(_1lane).txt only when that
    summary carries the SAME build fingerprint as the PMC profile next to it; the layer -> kernel map is the PMC profile's."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    kname = "void sqdet::convdet_dma_kernel<true>(sqdet::TileArgs, int, int)"
    (prof / "rX_hbm_traffic_pmc.json").write_text(json.dumps({
        "build_fingerprint": "abc", "config": "sqdet_infer", "by_layer": {"conv12": 1},
        "kernels": [{"layer": "conv12", "kernel": kname}, {"layer": "fire7", "kernel": "k2"}, {"layer": "fire8", "kernel": "k2"}]}))
    row = "%-90s %7d %12.1f %10.2f %7.2f\n"
    (prof / "rX_kernel_stats.txt").write_text("# x\n# build_fingerprint: abc\n" + row % ("kernel", 0, 0, 0, 0) + row % (kname[:90], 100, 8800.0, 88.0, 10.0)
                                              + row % ("k2", 200, 5000.0, 25.0, 5.0))
    (prof / "rX_kernel_stats_1lane.txt").write_text("# x\n# build_fingerprint: OTHER\n" + row % (kname[:90], 100, 6000.0, 60.0, 10.0))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.rocprof_launch_ms("rX_hbm_traffic_pmc.json", "conv12")
    assert got == {"ms": 0.088, "profile": "rX_kernel_stats.txt", "kernel_shared_by_launches": 1, "shared_layers": ["conv12"]}
    assert bench.rocprof_launch_ms("rX_hbm_traffic_pmc.json", "fire7")["kernel_shared_by_launches"] == 2
    assert bench.rocprof_launch_ms("rX_hbm_traffic_pmc.json", "conv12", "_kernel_stats_1lane.txt") is None     # another build's trace
    assert bench.rocprof_launch_ms("rX_hbm_traffic_pmc.json", "nope") is None
    assert bench.rocprof_launch_ms("missing.json", "conv12") is None


def test_serving_lane_state_is_swapped_and_restored():
    """ModelSkeleton._lane_state (the two-batches-in-flight serving loop): the pipeline state of a lane is installed for the duration
    of a call and written back, the single-lane state is restored afterwards -- also when the call raises."""
    from squeezedet_amd.nn_skeleton import ModelSkeleton

    class M(ModelSkeleton):
        def __init__(self):          # (no device, no graph: only the attributes _lane_state touches)
            self._pipe, self.post_stream, self._post_event = "P0", "S0", "E0"
    m = M()
    lane = dict(which=1, pipe=None, post_stream=None, post_event=None)
    with m._lane_state(lane):
        assert (m._pipe, m.post_stream, m._post_event, m._lane_plan) == (None, None, None, 1)
        m._pipe, m.post_stream, m._post_event = "P1", "S1", "E1"           # what the first call of a lane creates
    assert (lane["pipe"], lane["post_stream"], lane["post_event"]) == ("P1", "S1", "E1")
    assert (m._pipe, m.post_stream, m._post_event, m._lane_plan) == ("P0", "S0", "E0", 0)
    with pytest.raises(RuntimeError):
        with m._lane_state(lane):
            m._pipe = "P2"
            raise RuntimeError("boom")
    assert lane["pipe"] == "P2" and m._pipe == "P0" and m._lane_plan == 0


def test_tools_and_scripts_parse():
    """Every tools/*.py compiles and every tools/*.sh passes `bash -n` (they run on the GPU box only; a syntax error there costs a
    gpurun call)."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pys = sorted(glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "profiles", "*.py")) +
                 [os.path.join(root, n) for n in ("bench.py", "demo.py", "__graft_entry__.py")])
    assert len(pys) > 10
    for p in pys:
        compile(open(p).read(), p, "exec")
    for p in sorted(glob.glob(os.path.join(root, "tools", "*.sh"))):
        r = subprocess.run(["bash", "-n", p], capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stderr)
