"""GPU parity tests, op level: every HIP kernel called through the C ABI (squeezedet_amd.ops
-> ctypes -> libsqdet_hip.so) against the CPU oracle on the same seeded inputs.

Tolerances (stated per test):
  * fp32 convs: 1e-3 relative (BASELINE.json north_star); observed error is ~1e-6 (exact-f32 MFMA).
  * fp16 convs: compared with the oracle in fp16-storage mode (same fp16 operands, fp32
    accumulate, one fp16 rounding of the result) -> 2 fp16 ulps (2^-9 relative) + 1e-3 abs.
  * max-pool, top-N/NMS picks, indices, classes: bit-exact.
  * interpret_output floats: 2e-6 relative (device expf vs NumPy exp differ in the last ulp).
"""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import sqdet_oracle as O
from tests.golden import cases

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from squeezedet_amd import ops
    return ops


def _rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_mfma_fragment_layout_probe():
    """The conv kernels assume: operand lane l = (row/col l&15, k-group l>>4); accumulator
    register r of lane l = D[4*(l>>4)+r][l&15] (cdna_hip_programming.md section 3)."""
    lay = _ops().probe_mfma_layout()
    for shape in range(2):
        for l in range(64):
            for r in range(4):
                row, col = lay[shape, l, r]
                assert (row, col) == (4 * (l >> 4) + r, l & 15), "shape %d lane %d reg %d -> (%d,%d)" % (shape, l, r, row, col)


# (name, N, H, W, Cin, Cout, k, stride, padding, relu)
CONV_CASES = [
    ("conv1_375x1242_like", 2, 37, 53, 3, 64, 3, 2, "SAME", True),      # pads (1,1)/(0,1): odd/odd sizes
    ("conv1_384x1248_like", 1, 36, 48, 3, 64, 3, 2, "SAME", True),      # pads (0,1)
    ("conv1_plus_7x7_valid", 1, 41, 57, 3, 96, 7, 2, "VALID", True),
    ("fire2_squeeze", 2, 23, 31, 64, 16, 1, 1, "SAME", True),
    ("fire3_squeeze", 1, 23, 31, 128, 16, 1, 1, "SAME", True),
    ("fire2_expand1x1", 2, 23, 31, 16, 64, 1, 1, "SAME", True),
    ("fire2_expand3x3", 2, 23, 31, 16, 64, 3, 1, "SAME", True),
    ("fire4_expand3x3", 1, 12, 39, 32, 128, 3, 1, "SAME", True),
    ("fire6_squeeze", 1, 12, 20, 256, 48, 1, 1, "SAME", True),
    ("fire6_expand1x1", 1, 12, 20, 48, 192, 1, 1, "SAME", True),
    ("fire6_expand3x3", 1, 12, 20, 48, 192, 3, 1, "SAME", True),
    ("fire8_expand3x3", 1, 9, 14, 64, 256, 3, 1, "SAME", True),
    ("fire11_squeeze", 1, 9, 14, 768, 96, 1, 1, "SAME", True),
    ("fire10_expand3x3", 1, 9, 14, 96, 384, 3, 1, "SAME", True),
    ("conv12_convdet", 2, 7, 13, 768, 72, 3, 1, "SAME", False),
    ("plus_fire6_squeeze", 1, 9, 11, 256, 288, 1, 1, "SAME", True),
    ("stride2_3x3_valid", 1, 15, 17, 16, 32, 3, 2, "VALID", True),
    ("stride2_3x3_same_even", 1, 16, 18, 8, 20, 3, 2, "SAME", False),
    ("one_pixel", 1, 1, 1, 16, 16, 3, 1, "SAME", True),
    # multi-tile shapes for the LDS-tiled 3x3 kernels (8x16 tiles, ragged right/bottom edges)
    ("e3_multitile_32_128", 2, 19, 37, 32, 128, 3, 1, "SAME", True),
    ("e3_multitile_48_192", 1, 24, 78, 48, 192, 3, 1, "SAME", True),
    ("e3_multitile_16_64", 1, 47, 63, 16, 64, 3, 1, "SAME", True),
    ("e3_multitile_64_256", 1, 24, 40, 64, 256, 3, 1, "SAME", True),
    ("e3_multitile_96_384", 1, 17, 30, 96, 384, 3, 1, "SAME", True),
    ("convdet_full_24x78", 1, 24, 78, 768, 72, 3, 1, "SAME", False),
    ("plus_convdet_22x76", 1, 22, 76, 512, 72, 3, 1, "SAME", False),
    # odd tile counts per group (8-byte-aligned rows only) and shallow-K 72-cout 3x3
    ("c1_k64_n48", 1, 12, 20, 64, 48, 1, 1, "SAME", True),
    ("c1_k32_n80", 2, 9, 21, 32, 80, 1, 1, "SAME", False),
    ("c1_k128_n32", 1, 33, 45, 128, 32, 1, 1, "SAME", True),
    ("e3_cin32_cout72", 1, 11, 19, 32, 72, 3, 1, "SAME", True),
    ("e3_cin16_cout32", 1, 10, 33, 16, 32, 3, 1, "SAME", True),
    ("e3_cin64_cout48", 1, 9, 17, 64, 48, 3, 1, "SAME", True),
    # deep-K 1x1 as a workgroup GEMM tile (gemm1x1.hip): 3 / 4 / 5 cout tiles per wave, ragged last pixel tile, K staged in
    # several 4-chunk stages, more slices than one workgroup's four waves (SqueezeDet+ squeeze / expand, ResNet50 1x1)
    ("g1_k512_n384", 1, 22, 76, 512, 384, 1, 1, "SAME", True),
    ("g1_k384_n256", 2, 9, 23, 384, 256, 1, 1, "SAME", True),
    ("g1_k256_n72", 1, 13, 29, 256, 72, 1, 1, "SAME", False),
    ("g1_k2048_n512", 1, 7, 9, 2048, 512, 1, 1, "SAME", True),
    ("g1_k1024_n2048", 1, 5, 11, 1024, 2048, 1, 1, "SAME", False),
    ("g1_k264_n40", 3, 11, 7, 264, 40, 1, 1, "SAME", True),
    ("g1_stride2_k256_n512", 2, 23, 31, 256, 512, 1, 2, "SAME", False),
    ("g1_stride2_k256_n128", 1, 94, 61, 256, 128, 1, 2, "SAME", True),
    # 3x3 with K walked in 4-chunk LDS stages (SqueezeDet+ fire6-11: 9 / 12 chunks in fp16; ResNet50 res4 / res5)
    ("e3_k288_n192", 1, 17, 30, 288, 192, 3, 1, "SAME", True),
    ("e3_k384_n256", 2, 9, 19, 384, 256, 3, 1, "SAME", True),
    ("e3_k512_n512", 1, 8, 16, 512, 512, 3, 1, "SAME", True),
    # deep-K 1x1 on large maps: the streaming kernel with the cout group's weights resident in LDS (conv1x1k.hip: >= 8192 pixels,
    # 5..48 K chunks) -- ragged last pixel block, K not a multiple of the 8-chunk load group, channel padding in the last chunk,
    # ragged couts, 1..6 cout tiles per wave, several cout groups
    ("k1_k256_n32", 1, 94, 311, 256, 32, 1, 1, "SAME", True),
    ("k1_k768_n96", 1, 47, 175, 768, 96, 1, 1, "SAME", True),
    ("k1_k264_n40", 2, 70, 67, 264, 40, 1, 1, "SAME", True),
    ("k1_k160_n72", 1, 100, 90, 160, 72, 1, 1, "SAME", False),
    ("k1_k1024_n256", 1, 64, 130, 1024, 256, 1, 1, "SAME", True),
    ("k1_k384_n16", 1, 91, 93, 384, 16, 1, 1, "SAME", True),
]


@pytest.fixture(params=["auto", "generic"])
def conv_algo(request):
    """auto = specialised kernels where eligible (LDS-tiled 3x3, split-K ConvDet, ...);
    generic = the implicit-GEMM conv_direct / conv_gather kernels only.  Both must match the oracle."""
    ops = _ops()
    ops.set_option("conv_algo", 1 if request.param == "generic" else 0)
    yield request.param
    ops.set_option("conv_algo", 0)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_parity(case, dtype, conv_algo):
    ops = _ops()
    name, N, H, W, Cin, Cout, k, s, pad, relu = case
    rs = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
    x = rs.randn(N, H, W, Cin).astype(np.float32)
    w = (rs.randn(k, k, Cin, Cout) * (2.0 / (k * k * Cin)) ** 0.5).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, Cout).astype(np.float32)
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    xt, wt, bt = torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b)
    if dtype == "fp16":
        xt, wt = xt.half().float(), wt.half().float()
    ref = O.conv_layer(xt, wt, bt, s, pad, relu, storage=dtype).numpy()
    packed = ops.pack_conv_weights(wt.to(DEV), tdt)
    y = ops.conv2d_nhwc(xt.to(DEV, tdt).contiguous(), packed, bt.to(DEV), s, pad, relu)
    torch.cuda.synchronize()
    got = y.float().cpu().numpy()
    assert got.shape == ref.shape
    if dtype == "fp32":
        # north_star tolerance: 1e-3 relative; the exact-f32 MFMA path lands around 1e-6
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-4)
        assert _rel_err(got, ref) < 1e-4
    else:
        np.testing.assert_allclose(got, ref, rtol=2 ** -9, atol=1e-3)


@pytest.mark.parametrize("batch", [20, 52])
def test_convdet_persistent_workgroups_vs_single_tile_launch(batch):
    """The fp16 ConvDet kernel is persistent above 256 tiles (csrc/convdet.hip): a workgroup walks 2-4 tiles and prefetches
    the next tile's first stage across the reduction.  Size-independent property: with the SAME image in every batch slot,
    every slot's output must be bitwise the batch-1 result (15 tiles, one per workgroup, checked against the oracle by
    test_conv2d_parity[convdet_full_24x78]); a second run with distinct images must match the generic kernels."""
    ops = _ops()
    rs = np.random.RandomState(77)
    w = torch.from_numpy((rs.randn(3, 3, 768, 72) * (2.0 / (9 * 768)) ** 0.5).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.uniform(-0.5, 0.5, 72).astype(np.float32)).to(DEV)
    packed = ops.pack_conv_weights(w, torch.float16)
    img = torch.from_numpy(rs.randn(1, 24, 78, 768).astype(np.float32)).to(DEV, torch.float16)
    y1 = ops.conv2d_nhwc(img, packed, b, 1, "SAME", False)
    yb = ops.conv2d_nhwc(img.expand(batch, -1, -1, -1).contiguous(), packed, b, 1, "SAME", False)
    torch.cuda.synchronize()
    for n in range(batch):
        assert torch.equal(yb[n], y1[0]), "slot %d differs from the batch-1 result" % n
    x = torch.from_numpy(rs.randn(batch, 24, 78, 768).astype(np.float32)).to(DEV, torch.float16)
    y = ops.conv2d_nhwc(x, packed, b, 1, "SAME", False)
    ops.set_option("conv_algo", 1)
    try:
        yg = ops.conv2d_nhwc(x, packed, b, 1, "SAME", False)
    finally:
        ops.set_option("conv_algo", 0)
    torch.cuda.synchronize()
    # both accumulate in float32 (different K order) and round once to float16
    np.testing.assert_allclose(y.float().cpu().numpy(), yg.float().cpu().numpy(), rtol=2 ** -9, atol=2e-3)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_conv2d_concat_offset_and_fire(dtype):
    """expand1x1 / expand3x3 write the two halves of one concat tensor (nets/squeezeDet.py:106);
    sqdet_fire_fwd == the three convs."""
    ops = _ops()
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    p = {k: v for k, v in O.init_params("squeezeDet", seed=3, storage=dtype).items() if k.startswith("fire4/")}
    rs = np.random.RandomState(5)
    x = torch.from_numpy(np.maximum(rs.randn(2, 13, 21, 128), 0).astype(np.float32))
    if dtype == "fp16":
        x = x.half().float()
    ref = O.fire_layer(p, "fire4", x, storage=dtype).numpy()
    pk = {n: ops.pack_conv_weights(p["fire4/%s/kernels" % n].to(DEV), tdt) for n in ("squeeze1x1", "expand1x1", "expand3x3")}
    bs = {n: p["fire4/%s/biases" % n].to(DEV) for n in pk}
    y = ops.fire(x.to(DEV, tdt).contiguous(), pk["squeeze1x1"], bs["squeeze1x1"], pk["expand1x1"], bs["expand1x1"],
                 pk["expand3x3"], bs["expand3x3"])
    torch.cuda.synchronize()
    got = y.float().cpu().numpy()
    assert got.shape == ref.shape == (2, 13, 21, 256)
    tol = dict(rtol=1e-3, atol=1e-4) if dtype == "fp32" else dict(rtol=2 ** -8, atol=2e-3)
    np.testing.assert_allclose(got, ref, **tol)


def _same_as_unfused(got, want, s, dtype, what):
    """Fused launches are compared BITWISE with the generic kernel path, except the float16 squeeze-depth-16 modules
    (fire2 / fire3): their streaming kernel sums two taps of the expand3x3 inside one MFMA (fire2.hip, PAIR), which
    equals the three-conv path up to float32 summation order -- at most one float16 ulp after rounding, and rarely."""
    if dtype == "fp16" and s == 16:
        a, b = got.float().cpu().numpy(), want.float().cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=2 ** -9, atol=2 ** -12, err_msg=what)
        assert (a != b).mean() < 0.02, what
    else:
        assert torch.equal(got, want), what


FIRE_CASES = [("fire2", 64, 16, 64, 19, 37), ("fire3", 128, 16, 64, 9, 33), ("fire4", 128, 32, 128, 17, 20),
              ("fire5", 256, 32, 128, 8, 16), ("fire6", 256, 48, 192, 24, 78), ("fire7", 384, 48, 192, 11, 19),
              ("fire8", 384, 64, 256, 10, 17), ("fire9", 512, 64, 256, 9, 31), ("fire10", 512, 96, 384, 24, 78),
              ("fire11", 768, 96, 384, 13, 21)]


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("case", FIRE_CASES, ids=[c[0] for c in FIRE_CASES])
def test_fused_fire_module_parity(case, dtype):
    """sqdet_fire_fwd as ONE launch (squeeze halo in LDS -> expand1x1 || expand3x3 -> concat,
    nets/squeezeDet.py:81-106) against the oracle, and BITWISE against the three separate convs."""
    ops = _ops()
    name, cin, s, e, H, W = case
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    rs = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
    p = {}
    for sub, (k, ci, co) in (("squeeze1x1", (1, cin, s)), ("expand1x1", (1, s, e)), ("expand3x3", (3, s, e))):
        w = torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32))
        p["%s/%s/kernels" % (name, sub)] = w.half().float() if dtype == "fp16" else w
        p["%s/%s/biases" % (name, sub)] = torch.from_numpy(rs.uniform(-0.3, 0.3, co).astype(np.float32))
    x = torch.from_numpy(np.maximum(rs.randn(2, H, W, cin), 0).astype(np.float32))
    if dtype == "fp16":
        x = x.half().float()
    ref = O.fire_layer(p, name, x, storage=dtype).numpy()
    pk = {n: ops.pack_conv_weights(p["%s/%s/kernels" % (name, n)].to(DEV), tdt) for n in ("squeeze1x1", "expand1x1", "expand3x3")}
    bs = {n: p["%s/%s/biases" % (name, n)].to(DEV) for n in pk}
    xd = x.to(DEV, tdt).contiguous()
    args = (xd, pk["squeeze1x1"], bs["squeeze1x1"], pk["expand1x1"], bs["expand1x1"], pk["expand3x3"], bs["expand3x3"])
    ops.set_option("fire_fuse", 1)
    y_fused = ops.fire(*args)
    ops.set_option("fire_fuse", 2)
    y_sep = ops.fire(*args)
    ops.set_option("fire_fuse", 0)
    torch.cuda.synchronize()
    _same_as_unfused(y_fused, y_sep, s, dtype, "fused fire differs from squeeze -> expand1x1 / expand3x3")
    tol = dict(rtol=1e-3, atol=1e-4) if dtype == "fp32" else dict(rtol=2 ** -8, atol=2e-3)
    np.testing.assert_allclose(y_fused.float().cpu().numpy(), ref, **tol)


FIRE_POOL_CASES = [("fire3", 128, 16, 64, 94, 311, 2), ("fire3-odd", 128, 16, 64, 19, 37, 2), ("fire3-even", 128, 16, 64, 20, 28, 1),
                   ("fire5", 256, 32, 128, 47, 156, 2), ("fire5-small", 256, 32, 128, 9, 15, 3), ("fire2-shape", 64, 16, 64, 33, 30, 1),
                   ("fire5-many-tiles", 256, 32, 128, 47, 156, 10), ("fire3-many-tiles", 128, 16, 64, 94, 311, 5),
                   ("fallback", 256, 48, 192, 13, 21, 1)]


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("case", FIRE_POOL_CASES, ids=[c[0] for c in FIRE_POOL_CASES])
def test_fire_plus_maxpool_one_launch(case, dtype):
    """sqdet_fire_maxpool_fwd (fire3+pool3 / fire5+pool5, nets/squeezeDet.py:49-57: the 3x3/s2 SAME pool taken in
    registers, only the pooled tensor written) BITWISE against fire -> max-pool, over odd / even sizes (SAME pads
    (0,1) and (1,1)), image edges inside tiles, and a shape the streaming kernel does not cover (fallback)."""
    ops = _ops()
    name, cin, s, e, H, W, N = case
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    rs = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32))
    ws, w1, w3 = mk(1, cin, s), mk(1, s, e), mk(3, s, e)
    bs, b1, b3 = [torch.from_numpy(rs.uniform(-0.3, 0.3, c).astype(np.float32)).to(DEV) for c in (s, e, e)]
    x = torch.from_numpy((rs.randn(N, H, W, cin) - 0.3).astype(np.float32)).to(DEV, tdt).contiguous()   # many negative pre-activations
    ps, p1, p3 = [ops.pack_conv_weights(w.to(DEV), tdt) for w in (ws, w1, w3)]
    args = (x, ps, bs, p1, b1, p3, b3)
    y = ops.fire_maxpool(*args)
    ops.set_option("fire_fuse", 2)
    want = ops.maxpool_nhwc(ops.fire(*args), 3, 2, "SAME")
    ops.set_option("fire_fuse", 0)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (N, -(-H // 2), -(-W // 2), 2 * e)
    _same_as_unfused(y, want, s, dtype, "fire+pool in one launch differs from fire -> pool")


POOL_CASES = [(2, 37, 53, 64, 3, 2, "SAME"), (1, 188, 621, 8, 3, 2, "SAME"), (1, 47, 156, 16, 3, 2, "SAME"),
              (1, 94, 311, 8, 3, 2, "SAME"), (1, 41, 57, 96, 3, 2, "VALID"), (1, 12, 14, 8, 2, 2, "SAME"), (1, 3, 3, 8, 3, 2, "SAME")]


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("case", POOL_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_maxpool_parity_bit_exact(case, dtype):
    ops = _ops()
    N, H, W, C, k, s, pad = case
    rs = np.random.RandomState(H * 1000 + W)
    x = torch.from_numpy((rs.randn(N, H, W, C) - 1.0).astype(np.float32))  # mostly negative: zero padding would win
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    xq = x.to(tdt)
    ref = O.pooling_layer(xq.float(), k, s, pad).numpy()
    y = ops.maxpool_nhwc(xq.to(DEV).contiguous(), k, s, pad)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y.float().cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("cfg", ["squeezeDet", "squeezeDet+"])
def test_interpret_output_parity(cfg, dtype):
    ops = _ops()
    mc = O.kitti_squeezeDet_config() if cfg == "squeezeDet" else O.kitti_squeezeDetPlus_config()
    gh, gw = (24, 78) if cfg == "squeezeDet" else (22, 76)
    rs = np.random.RandomState(11)
    preds = (rs.randn(3, gh, gw, 72) * 1.7).astype(np.float32)
    preds[0, :, :, 36:] *= 2.0  # push some dw/dh beyond EXP_THRESH so both safe_exp branches run
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    pt = torch.from_numpy(preds).to(tdt)
    ref = O.interpret_output(pt.float().numpy(), mc)
    anchors = torch.from_numpy(mc.ANCHOR_BOX.astype(np.float32)).to(DEV)
    boxes, probs, cls, pcp, pconf = ops.interpret_output(pt.to(DEV).contiguous(), anchors, mc.CLASSES, mc.ANCHOR_PER_GRID,
                                                         mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH, with_class_probs=True)
    torch.cuda.synchronize()
    assert boxes.dtype == torch.float32 and probs.dtype == torch.float32 and cls.dtype == torch.int64
    # 2e-6 relative: device expf vs NumPy exp may differ in the last ulp
    np.testing.assert_allclose(pcp.cpu().numpy(), ref["pred_class_probs"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(pconf.cpu().numpy(), ref["pred_conf"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(probs.cpu().numpy(), ref["det_probs"], rtol=3e-6, atol=1e-9)
    # box corners are differences of ~1e3-magnitude terms through exp(): a last-ulp exp difference is ~1e-4 absolute
    np.testing.assert_allclose(boxes.cpu().numpy(), ref["det_boxes"], rtol=2e-6, atol=1e-3)
    # classes: exact wherever the top-2 class scores are separated by more than the exp ulp noise
    pr = ref["pred_class_probs"] * ref["pred_conf"][..., None]
    srt = np.sort(pr, axis=2)
    clear = (srt[..., -1] - srt[..., -2]) > 1e-5 * srt[..., -1]
    got_cls = cls.cpu().numpy()
    assert clear.mean() > 0.99
    np.testing.assert_array_equal(got_cls[clear], ref["det_class"][clear])
    # boxes that do not touch an exp() are bit-exact: cx/cy only depend on mul/add when unclipped
    assert (boxes.cpu().numpy()[..., 2:] >= 1.0).all()


def _gpu_filter(boxes, probs, cls, mc, max_out=None):
    ops = _ops()
    b = torch.from_numpy(boxes).to(DEV).reshape(1, -1, 4).contiguous()
    p = torch.from_numpy(probs).to(DEV).reshape(1, -1).contiguous()
    c = torch.from_numpy(cls).to(DEV).reshape(1, -1).contiguous()
    ob, op, oc, oi, cnt = ops.filter_prediction(b, p, c, mc.CLASSES, mc.TOP_N_DETECTION, mc.NMS_THRESH, mc.PROB_THRESH, max_out)
    torch.cuda.synchronize()
    n = int(cnt[0])
    return ob[0, :n].cpu().numpy(), op[0, :n].cpu().numpy(), oc[0, :n].cpu().numpy(), oi[0, :n].cpu().numpy(), n


@pytest.mark.parametrize("name", cases.FILTER_CASES)
def test_filter_prediction_vs_reference_golden(golden_dir, name):
    """Bit-exact against the outputs of the reference's own filter_prediction."""
    g = np.load(os.path.join(golden_dir, "filter_prediction.npz"))
    boxes, probs, cls, overrides = cases.make_filter_case(name)
    mc = O.kitti_squeezeDet_config()
    for k, v in overrides.items():
        mc[k] = v
    ob, op, oc, oi, n = _gpu_filter(boxes, probs, cls, mc)
    assert n == len(g[name + "_probs"])
    np.testing.assert_array_equal(ob, g[name + "_boxes"])
    np.testing.assert_array_equal(op, g[name + "_probs"])
    np.testing.assert_array_equal(oc.astype(np.int64), g[name + "_cls"])
    # anchor indices: bit-exact against the oracle (the reference does not return them)
    _, _, _, idx = O.filter_prediction(mc, boxes, probs, cls, return_index=True)
    np.testing.assert_array_equal(oi, np.array(idx, np.int32))
    np.testing.assert_array_equal(boxes[oi], ob)


def test_filter_prediction_batched_random_vs_oracle():
    ops = _ops()
    mc = O.kitti_squeezeDet_config()
    B, A = 5, 16848
    rs = np.random.RandomState(123)
    centers = rs.uniform([100, 60, 40, 30], [1100, 320, 250, 150], size=(B, 8, 4))
    which = rs.randint(0, 8, (B, A))
    boxes = np.take_along_axis(centers, which[..., None].repeat(4, 2), 1) + rs.normal(0, 1, (B, A, 4)) * [14, 9, 12, 9]
    boxes[..., 2:] = np.maximum(boxes[..., 2:], 1.0)
    boxes = boxes.astype(np.float32)
    probs = (rs.uniform(0, 1, (B, A)) ** 6).astype(np.float32)
    cls = rs.randint(0, 3, (B, A)).astype(np.int64)
    ob, op, oc, oi, cnt = ops.filter_prediction(torch.from_numpy(boxes).to(DEV), torch.from_numpy(probs).to(DEV),
                                                torch.from_numpy(cls).to(DEV), 3, 64, 0.4, 0.005)
    torch.cuda.synchronize()
    for i in range(B):
        fb, fp, fc, fi = O.filter_prediction(mc, boxes[i], probs[i], cls[i], return_index=True)
        n = int(cnt[i])
        assert n == len(fp)
        np.testing.assert_array_equal(oi[i, :n].cpu().numpy(), np.array(fi, np.int32))
        np.testing.assert_array_equal(op[i, :n].cpu().numpy(), np.array(fp, np.float32))
        np.testing.assert_array_equal(ob[i, :n].cpu().numpy(), np.array(fb, np.float32).reshape(-1, 4))
        np.testing.assert_array_equal(oc[i, :n].cpu().numpy(), np.array(fc, np.int32))
        assert (oc[i, n:].cpu().numpy() == -1).all()


def test_filter_prediction_tie_rule_and_small_inputs():
    """Ties: descending prob, then HIGHER anchor index first -- the rule the
    oracle's stable_desc_order states.  Also A < TOP_N (threshold branch) and A == 1."""
    mc = O.kitti_squeezeDet_config()
    rs = np.random.RandomState(9)
    A = 500
    boxes = np.stack([rs.uniform(0, 1247, A), rs.uniform(0, 383, A), rs.uniform(1, 40, A), rs.uniform(1, 40, A)], 1).astype(np.float32)
    probs = np.round(rs.uniform(0, 1, A), 2).astype(np.float32)  # heavy ties
    cls = rs.randint(0, 3, A).astype(np.int64)
    ob, op, oc, oi, n = _gpu_filter(boxes, probs, cls, mc)
    fb, fp, fc, fi = O.filter_prediction(mc, boxes, probs, cls, return_index=True)
    np.testing.assert_array_equal(oi, np.array(fi, np.int32))
    np.testing.assert_array_equal(op, np.array(fp, np.float32))
    for A in (1, 7, 64):  # A <= TOP_N -> nn_skeleton.py:716-720 threshold branch
        ob, op, oc, oi, n = _gpu_filter(boxes[:A], probs[:A], cls[:A], mc, max_out=64)
        fb, fp, fc, fi = O.filter_prediction(mc, boxes[:A], probs[:A], cls[:A], return_index=True)
        np.testing.assert_array_equal(oi, np.array(fi, np.int32))


@pytest.mark.parametrize("kind", ["constant", "two_levels", "few_distinct"])
def test_filter_prediction_massive_ties_full_size(kind):
    """Score maps with thousands of equal values (a saturated or constant head): the fast top-N
    kernel's candidate list overflows and it must fall back to the exact radix select; the result
    still follows the repo's total order (descending prob, ties -> higher anchor index first)."""
    mc = O.kitti_squeezeDet_config()
    rs = np.random.RandomState(31)
    A = 16848
    boxes = np.stack([rs.uniform(0, 1247, A), rs.uniform(0, 383, A), rs.uniform(1, 60, A), rs.uniform(1, 60, A)], 1).astype(np.float32)
    if kind == "constant":
        probs = np.full(A, 0.5, np.float32)
    elif kind == "two_levels":
        probs = np.where(rs.uniform(0, 1, A) < 0.002, 0.9, 0.25).astype(np.float32)   # ~34 high, the rest tied
    else:
        probs = rs.choice(np.array([0.1, 0.2, 0.7], np.float32), A)
    cls = rs.randint(0, 3, A).astype(np.int64)
    ob, op, oc, oi, n = _gpu_filter(boxes, probs, cls, mc)
    fb, fp, fc, fi = O.filter_prediction(mc, boxes, probs, cls, return_index=True)
    np.testing.assert_array_equal(oi, np.array(fi, np.int32))
    np.testing.assert_array_equal(op, np.array(fp, np.float32))
    np.testing.assert_array_equal(oc, np.array(fc, np.int32))


def test_filter_prediction_threshold_overflow_reports_count():
    mc = O.kitti_squeezeDet_config()
    mc.TOP_N_DETECTION = 0
    mc.PROB_THRESH = 0.5
    boxes, probs, cls, _ = cases.make_filter_case("uniform0")
    ops = _ops()
    out = ops.filter_prediction(torch.from_numpy(boxes).to(DEV)[None], torch.from_numpy(probs).to(DEV)[None],
                                torch.from_numpy(cls).to(DEV)[None], 3, 0, 0.4, 0.5, max_out=128)
    torch.cuda.synchronize()
    assert int(out[4][0]) == -int((probs > 0.5).sum())


def test_util_nms_known_answers(golden_dir):
    from squeezedet_amd import util
    g = np.load(os.path.join(golden_dir, "util_kat.npz"))
    assert util.nms(g["iou_boxes"], g["nms_probs"], 0.4) == [True, False, True, False]
    assert util.nms(g["chain_boxes"], g["chain_probs"], 0.4) == [True, False, False]  # non-greedy
    rs = np.random.RandomState(4)
    n = 300
    bx = np.stack([rs.uniform(0, 400, n), rs.uniform(0, 300, n), rs.uniform(1, 200, n), rs.uniform(1, 150, n)], 1).astype(np.float32)
    pr = rs.uniform(0, 1, n).astype(np.float32)
    assert util.nms(bx, pr, 0.4) == O.nms(bx, pr, 0.4)


def test_cpu_tensors_are_rejected():
    from squeezedet_amd._lib import SqdetError
    ops = _ops()
    with pytest.raises(SqdetError):
        ops.maxpool_nhwc(torch.zeros(1, 8, 8, 8), 3, 2, "SAME")
    with pytest.raises(SqdetError):
        ops.maxpool_nhwc(torch.zeros(1, 8, 8, 6, device=DEV), 3, 2, "SAME")  # channels not a multiple of 4


STEM_CASES = [(2, 37, 53, 3, 64, "SAME", "SAME"), (1, 375, 1242, 3, 64, "SAME", "SAME"), (1, 384, 1248, 3, 64, "SAME", "SAME"),
              (1, 64, 80, 3, 64, "SAME", "SAME"), (1, 75, 131, 7, 96, "VALID", "VALID"), (1, 375, 1242, 7, 96, "VALID", "VALID"),
              # the persistent fp16 stem (even W >= 236): several images, ragged last tiles, VALID conv padding
              (3, 45, 250, 3, 64, "SAME", "SAME"), (2, 100, 236, 3, 64, "VALID", "SAME"), (5, 19, 480, 3, 64, "SAME", "VALID"),
              # the float16 7x7 stems on even widths (stem5.hip: one K chunk per kernel row): ResNet50's SAME conv + VALID pool at full
              # size and on ragged maps (last strip / last row segment partly outside), SqueezeDet+'s with several images
              (1, 375, 1242, 7, 64, "SAME", "VALID"), (2, 61, 134, 7, 64, "SAME", "VALID"), (3, 45, 250, 7, 96, "VALID", "VALID"),
              (2, 23, 30, 7, 64, "SAME", "VALID")]


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("case", STEM_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_fused_stem_conv_pool_parity(case, dtype):
    """conv1+pool1 in one launch (nets/squeezeDet.py:40-44, nets/squeezeDetPlus.py:40-44) against the
    oracle's conv_layer -> pooling_layer, and BITWISE against the unfused HIP conv -> pool."""
    ops = _ops()
    N, H, W, k, cout, cpad, ppad = case
    rs = np.random.RandomState(H * 7 + W)
    x = torch.from_numpy(rs.uniform(-2, 2, (N, H, W, 3)).astype(np.float32))
    w = torch.from_numpy((rs.randn(k, k, 3, cout) * (2.0 / (k * k * 3)) ** 0.5).astype(np.float32))
    b = torch.from_numpy(rs.uniform(-0.5, 0.5, cout).astype(np.float32))
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    if dtype == "fp16":
        x, w = x.half().float(), w.half().float()
    ref = O.pooling_layer(O.conv_layer(x, w, b, 2, cpad, True, storage=dtype), 3, 2, ppad).numpy()
    packed = ops.pack_conv_weights(w.to(DEV), tdt)
    xd, bd = x.to(DEV, tdt).contiguous(), b.to(DEV)
    y = ops.stem_conv_pool(xd, packed, bd, cpad, ppad)
    y2 = ops.maxpool_nhwc(ops.conv2d_nhwc(xd, packed, bd, 2, cpad, True), 3, 2, ppad)
    torch.cuda.synchronize()
    assert tuple(y.shape) == ref.shape
    if dtype == "fp16" and k == 7 and W % 2 == 0 and ppad == "VALID":
        # stem5.hip walks K as 7 kernel rows of 8 pixels x 4 channels (zero-weight padding) -- another float32 summation order than the
        # gather kernels' 147 products in chunks of 32: float16-identical but for rare 1-ulp flips (and absolute float32 noise next to 0)
        ulps = (y.view(torch.int16).int() - y2.view(torch.int16).int()).abs()
        gap = (y.float() - y2.float()).abs()
        assert bool(((ulps <= 1) | (gap <= 5e-5)).all()) and float((ulps != 0).float().mean()) < 2e-3, \
            "7x7 stem vs conv -> pool: %d ulps, %g, %g flipped" % (int(ulps.max()), float(gap.max()), float((ulps != 0).float().mean()))
        ops.set_option("stem_algo", 2)
        y3 = ops.stem_conv_pool(xd, packed, bd, cpad, ppad)
        ops.set_option("stem_algo", 0)
        assert torch.equal(y3, y2), "strip stem differs from conv -> pool"
    elif dtype == "fp16" and k == 3 and W % 2 == 0 and W * 6 >= 1408:
        # the persistent fp16 stem (stem3.hip) walks the 27 im2col products in a different K order inside the MFMA:
        # float32 summation order -> identical after fp16 rounding except for a 1-ulp flip in < 1e-3 of the elements
        # (observed 6e-5); every other path is bitwise
        ulps = (y.view(torch.int16).int() - y2.view(torch.int16).int()).abs()
        gap = (y.float() - y2.float()).abs()
        # (next to zero -- a cancelling sum behind the ReLU -- the ulp is tiny and the float32 noise is not: absolute bound)
        assert bool(((ulps <= 1) | (gap <= 2e-5)).all()) and float((ulps != 0).float().mean()) < 1e-3, \
            "persistent stem vs conv -> pool: %d ulps, %g" % (int(ulps.max()), float(gap.max()))
        ops.set_option("stem_algo", 2)
        y3 = ops.stem_conv_pool(xd, packed, bd, cpad, ppad)
        ops.set_option("stem_algo", 0)
        assert torch.equal(y3, y2), "strip stem differs from conv -> pool"
    else:
        assert torch.equal(y, y2), "fused stem differs from conv -> pool"
    tol = dict(rtol=1e-3, atol=1e-4) if dtype == "fp32" else dict(rtol=2 ** -9, atol=1e-3)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref, **tol)


# (name, S, E, next_S, H, W, N): SqueezeDet's fire6..fire11 shapes on full and ragged maps; odd batch (the second image
# of the last workgroup's pair does not exist); next_S = 0: expand only (fire11: the concat tensor is the output)
CHAIN_CASES = [("fire6-7", 48, 192, 48, 24, 78, 2), ("fire7-8", 48, 192, 64, 11, 19, 3), ("fire8-9", 64, 256, 64, 24, 78, 1),
               ("fire9-10", 64, 256, 96, 9, 31, 2), ("fire10-11", 96, 384, 96, 24, 78, 3), ("fire11", 96, 384, 0, 24, 78, 2),
               ("fire11-ragged", 96, 384, 0, 13, 21, 5), ("fire7-only", 48, 192, 0, 8, 16, 1), ("one-pixel", 64, 256, 64, 1, 1, 1),
               # the large early maps (one-chunk squeezes): fire2 -> 3, fire4 -> 5, fire5 -> (pool) fire6's width, expand only
               ("fire2-3", 16, 64, 16, 47, 83, 2), ("fire4-5", 32, 128, 32, 23, 40, 3), ("fire5-w48", 32, 128, 48, 9, 17, 1),
               ("fire3-only", 16, 64, 0, 19, 33, 1), ("fire2-3-many-quads", 16, 64, 16, 94, 311, 3), ("fire4-5-odd", 32, 128, 32, 47, 156, 5)]


@pytest.mark.parametrize("want_y", [False, True])
@pytest.mark.parametrize("case", CHAIN_CASES, ids=[c[0] for c in CHAIN_CASES])
def test_fire_chain_parity(case, want_y):
    """sqdet_fire_chain_fwd (expand1x1 || expand3x3 of one fire module + the squeeze1x1 of the next in one launch,
    nets/squeezeDet.py:58-69,81-106; float16) BITWISE against the separate convs through memory, and against the
    oracle in float16-storage mode (2 float16 ulps + 2e-3 abs: same tolerance as the fused fire module)."""
    ops = _ops()
    name, s, e, s2, H, W, N = case
    tdt = torch.float16
    rs = np.random.RandomState(zlib.crc32(("chain" + name).encode()) % (2 ** 31))
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).half().float()
    w1, w3 = mk(1, s, e), mk(3, s, e)
    b1 = torch.from_numpy(rs.uniform(-0.3, 0.3, e).astype(np.float32))
    b3 = torch.from_numpy(rs.uniform(-0.3, 0.3, e).astype(np.float32))
    ws = mk(1, 2 * e, s2) if s2 else None
    bs = torch.from_numpy(rs.uniform(-0.3, 0.3, s2).astype(np.float32)) if s2 else None
    sq = torch.from_numpy(np.maximum(rs.randn(N, H, W, s), 0).astype(np.float32)).half()
    chain = ops.FireChainStream(w1.to(DEV), w3.to(DEV), ws.to(DEV) if s2 else None, tdt)
    sqd = sq.to(DEV).contiguous()
    y, so = ops.fire_chain(sqd, chain, b1.to(DEV), b3.to(DEV), bs.to(DEV) if s2 else None, want_y=want_y)
    if s <= 32 and s2 and not want_y:
        # one-chunk squeezes on large maps run the PERSISTENT, weights-resident kernel: force it here at test size
        # ("dbg" 31) -- it must give the ring kernel's result bit for bit (both are checked against the separate convs)
        ops.set_option("dbg", 31)
        _, so_p = ops.fire_chain(sqd, chain, b1.to(DEV), b3.to(DEV), bs.to(DEV), want_y=False)
        ops.set_option("dbg", 0)
        torch.cuda.synchronize()
        assert torch.equal(so_p, so), "persistent chain kernel differs from the ring kernel"
    # the separate launches
    p1, p3 = ops.pack_conv_weights(w1.to(DEV), tdt), ops.pack_conv_weights(w3.to(DEV), tdt)
    y_sep = torch.empty((N, H, W, 2 * e), dtype=tdt, device=DEV)
    ops.conv2d_nhwc(sqd, p1, b1.to(DEV), 1, "SAME", True, out=y_sep, out_coffset=0)
    ops.conv2d_nhwc(sqd, p3, b3.to(DEV), 1, "SAME", True, out=y_sep, out_coffset=e)
    torch.cuda.synchronize()
    if want_y or not s2:
        assert y is not None and torch.equal(y, y_sep), "chain concat tensor differs from expand1x1 / expand3x3"
    else:
        assert y is None
    # oracle (float16 storage)
    e1o = O.conv_layer(sq.float(), w1, b1, 1, "SAME", True, storage="fp16")
    e3o = O.conv_layer(sq.float(), w3, b3, 1, "SAME", True, storage="fp16")
    cat = torch.cat([e1o, e3o], dim=3)
    tol = dict(rtol=2 ** -8, atol=2e-3)
    np.testing.assert_allclose(y_sep.float().cpu().numpy(), cat.numpy(), **tol)
    if s2:
        so_sep = ops.conv2d_nhwc(y_sep, ops.pack_conv_weights(ws.to(DEV), tdt), bs.to(DEV), 1, "SAME", True)
        torch.cuda.synchronize()
        assert torch.equal(so, so_sep), "chain squeeze tensor differs from the separate squeeze conv"
        # the oracle squeeze on the DEVICE's concat tensor (one conv of error, like every other conv test)
        s_ref = O.conv_layer(y_sep.float().cpu(), ws, bs, 1, "SAME", True, storage="fp16")
        np.testing.assert_allclose(so.float().cpu().numpy(), s_ref.numpy(), **tol)
    else:
        assert so is None


@pytest.mark.parametrize("case", [("fire3+pool3", 128, 16, 64, 94, 311, 2), ("fire5+pool5", 256, 32, 128, 47, 156, 2)], ids=["fire3+pool3", "fire5+pool5"])
def test_fire_pool_full_size_fp16_vs_oracle(case):
    """fire3+pool3 / fire5+pool5 (nets/squeezeDet.py:49-57) as ONE launch at their real SqueezeDet shapes in float16,
    DIRECTLY against the oracle (fire_layer -> pooling_layer in float16-storage mode) -- not only against the repo's own
    unfused path: 2 float16 ulps + 2e-3 abs, the fused-fire tolerance (the S = 16 module pairs two taps per MFMA)."""
    ops = _ops()
    name, cin, s, e, H, W, N = case
    tdt = torch.float16
    rs = np.random.RandomState(zlib.crc32(("full" + name).encode()) % (2 ** 31))
    fn = name.split("+")[0]
    p = {}
    for sub, (k, ci, co) in (("squeeze1x1", (1, cin, s)), ("expand1x1", (1, s, e)), ("expand3x3", (3, s, e))):
        w = torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32))
        p["%s/%s/kernels" % (fn, sub)] = w.half().float()
        p["%s/%s/biases" % (fn, sub)] = torch.from_numpy(rs.uniform(-0.3, 0.3, co).astype(np.float32))
    x = torch.from_numpy(np.maximum(rs.randn(N, H, W, cin), 0).astype(np.float32)).half().float()
    ref = O.pooling_layer(O.fire_layer(p, fn, x, storage="fp16"), 3, 2, "SAME").numpy()
    pk = {n: ops.pack_conv_weights(p["%s/%s/kernels" % (fn, n)].to(DEV), tdt) for n in ("squeeze1x1", "expand1x1", "expand3x3")}
    bs = {n: p["%s/%s/biases" % (fn, n)].to(DEV) for n in pk}
    y = ops.fire_maxpool(x.to(DEV, tdt).contiguous(), pk["squeeze1x1"], bs["squeeze1x1"], pk["expand1x1"], bs["expand1x1"],
                         pk["expand3x3"], bs["expand3x3"])
    torch.cuda.synchronize()
    got = y.float().cpu().numpy()
    assert got.shape == ref.shape == (N, -(-H // 2), -(-W // 2), 2 * e)
    np.testing.assert_allclose(got, ref, rtol=2 ** -8, atol=2e-3)
    frac = float((got != ref).mean())
    assert frac < 0.05, "more than 5 %% of the pooled float16 values differ from the oracle's (%g)" % frac


EXPAND_CASES = [("fire3", 128, 16, 64, 94, 311, 2, True), ("fire3-odd", 128, 16, 64, 19, 37, 3, True), ("fire5", 256, 32, 128, 47, 156, 2, True),
                ("fire5-small", 256, 32, 128, 9, 15, 1, True), ("fire2-nopool", 64, 16, 64, 33, 30, 2, False), ("fire4-nopool", 128, 32, 128, 17, 20, 1, False),
                ("fallback-nopool", 256, 48, 192, 13, 21, 1, False)]


@pytest.mark.parametrize("case", EXPAND_CASES, ids=[c[0] for c in EXPAND_CASES])
def test_fire_expand_from_squeeze_tensor(case):
    """sqdet_fire_expand_fwd (the expand half of a fire module -- and the max-pool behind it -- from the module's squeeze
    tensor, as the chained plan runs fire3+pool3 / fire5+pool5) BITWISE against the whole-module launches
    sqdet_fire_fwd / sqdet_fire_maxpool_fwd on the same weights, whose squeeze conv produces that tensor."""
    ops = _ops()
    name, cin, s, e, H, W, N, pool = case
    tdt = torch.float16
    rs = np.random.RandomState(zlib.crc32(("expand" + name).encode()) % (2 ** 31))
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).half().float()
    ws, w1, w3 = mk(1, cin, s), mk(1, s, e), mk(3, s, e)
    bs, b1, b3 = [torch.from_numpy(rs.uniform(-0.3, 0.3, c).astype(np.float32)).to(DEV) for c in (s, e, e)]
    ps, p1, p3 = [ops.pack_conv_weights(w_.to(DEV), tdt) for w_ in (ws, w1, w3)]
    x = torch.from_numpy(np.maximum(rs.randn(N, H, W, cin), 0).astype(np.float32)).to(DEV, tdt).contiguous()
    sq = ops.conv2d_nhwc(x, ps, bs, 1, "SAME", True)
    got = ops.fire_expand(sq, p1, b1, p3, b3, pool=pool)
    want = ops.fire_maxpool(x, ps, bs, p1, b1, p3, b3) if pool else ops.fire(x, ps, bs, p1, b1, p3, b3)
    torch.cuda.synchronize()
    assert got.shape == want.shape and torch.equal(got, want)


# (name, S, E, H, W, N, dbg): SqueezeDet+'s deep squeezes -- fire4 / fire5 (S = 192: 6 K chunks, two workgroups per CU) and fire9-11
# (S = 384: 12 chunks resident, one workgroup per CU: taken where the launch has no more workgroups than CUs; "dbg" 94 forces it)
PAIR_CASES = [("plus-fire4", 192, 128, 92, 61, 2, 0), ("plus-fire5-ragged", 192, 128, 13, 21, 3, 0), ("plus-fire9", 384, 256, 22, 76, 2, 0),
              ("plus-fire9-ragged", 384, 256, 9, 31, 3, 94), ("one-pixel", 192, 128, 1, 1, 1, 0)]


@pytest.mark.parametrize("case", PAIR_CASES, ids=[c[0] for c in PAIR_CASES])
def test_fire_expand_pair_tile_parity(case):
    """sqdet_fire_expand_fwd on squeeze depths the streaming kernels do not cover (nets/squeezeDetPlus.py:46-73,81-106): expand1x1 and
    expand3x3 from ONE staged squeeze tile in one launch of the tile kernel's PAIR form -- BITWISE the two separate convs (and "dbg" 93,
    which switches the form off), and against the oracle in float16-storage mode."""
    ops = _ops()
    name, s, e, H, W, N, dbg = case
    tdt = torch.float16
    rs = np.random.RandomState(zlib.crc32(("pair" + name).encode()) % (2 ** 31))
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).half().float()
    w1, w3 = mk(1, s, e), mk(3, s, e)
    b1 = torch.from_numpy(rs.uniform(-0.3, 0.3, e).astype(np.float32))
    b3 = torch.from_numpy(rs.uniform(-0.3, 0.3, e).astype(np.float32))
    sq = torch.from_numpy(np.maximum(rs.randn(N, H, W, s), 0).astype(np.float32)).half()
    sqd = sq.to(DEV).contiguous()
    p1, p3 = ops.pack_conv_weights(w1.to(DEV), tdt), ops.pack_conv_weights(w3.to(DEV), tdt)
    ops.set_option("dbg", dbg)
    try:
        assert ops.lib().sqdet_fire_expand_pair_supported(N, H, W, s, e, e, 1) == 1
        got = ops.fire_expand(sqd, p1, b1.to(DEV), p3, b3.to(DEV), pool=False)
        ops.set_option("dbg", 93)
        off = ops.fire_expand(sqd, p1, b1.to(DEV), p3, b3.to(DEV), pool=False)
    finally:
        ops.set_option("dbg", 0)
    y_sep = torch.empty((N, H, W, 2 * e), dtype=tdt, device=DEV)
    ops.conv2d_nhwc(sqd, p1, b1.to(DEV), 1, "SAME", True, out=y_sep, out_coffset=0)
    ops.conv2d_nhwc(sqd, p3, b3.to(DEV), 1, "SAME", True, out=y_sep, out_coffset=e)
    torch.cuda.synchronize()
    assert torch.equal(off, y_sep), "two-launch fallback differs from the separate convs"
    assert torch.equal(got[..., e:], y_sep[..., e:]), "PAIR form: expand3x3 half differs from the separate conv"
    assert torch.equal(got[..., :e], y_sep[..., :e]), "PAIR form: expand1x1 half differs from the separate conv"
    e1o = O.conv_layer(sq.float(), w1, b1, 1, "SAME", True, storage="fp16")
    e3o = O.conv_layer(sq.float(), w3, b3, 1, "SAME", True, storage="fp16")
    np.testing.assert_allclose(got.float().cpu().numpy(), torch.cat([e1o, e3o], dim=3).numpy(), rtol=2 ** -8, atol=2e-3)


SQNEXT_CASES = [("fire2-3", 64, 16, 64, 16, 94, 311, 2), ("fire2-3-ragged", 64, 16, 64, 16, 19, 37, 3), ("fire4-5", 128, 32, 128, 32, 47, 156, 2),
                ("fire4-5-small", 128, 32, 128, 32, 9, 15, 5)]


@pytest.mark.parametrize("case", SQNEXT_CASES, ids=[c[0] for c in SQNEXT_CASES])
def test_fire_squeeze_next_one_launch(case):
    """sqdet_fire_squeeze_next_fwd (a whole fire module from x whose concat tensor is replaced by the NEXT module's squeeze
    tensor: fire2 -> fire3's squeeze, fire4 -> fire5's, nets/squeezeDet.py:46-53) BITWISE against sqdet_fire_fwd followed
    by the squeeze conv, and against the oracle in float16-storage mode."""
    ops = _ops()
    name, cin, s, e, s2, H, W, N = case
    tdt = torch.float16
    assert ops.lib().sqdet_fire_squeeze_next_supported(cin, s, e, e, s2, 1) == 1
    rs = np.random.RandomState(zlib.crc32(("sqnext" + name).encode()) % (2 ** 31))
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).half().float()
    ws, w1, w3, wn = mk(1, cin, s), mk(1, s, e), mk(3, s, e), mk(1, 2 * e, s2)
    bs, b1, b3, bn = [torch.from_numpy(rs.uniform(-0.3, 0.3, c).astype(np.float32)) for c in (s, e, e, s2)]
    ps, p1, p3, pn = [ops.pack_conv_weights(w_.to(DEV), tdt) for w_ in (ws, w1, w3, wn)]
    x = torch.from_numpy(np.maximum(rs.randn(N, H, W, cin), 0).astype(np.float32)).half()
    xd = x.to(DEV).contiguous()
    got = ops.fire_squeeze_next(xd, ps, bs.to(DEV), p1, b1.to(DEV), p3, b3.to(DEV), pn, bn.to(DEV))
    y = ops.fire(xd, ps, bs.to(DEV), p1, b1.to(DEV), p3, b3.to(DEV))
    want = ops.conv2d_nhwc(y, pn, bn.to(DEV), 1, "SAME", True)
    torch.cuda.synchronize()
    if s == 16:   # (the float16 S = 16 module pairs two taps per MFMA in BOTH launches: same kernel arithmetic)
        pass
    assert got.shape == want.shape and torch.equal(got, want), "squeeze-out form differs from fire -> squeeze conv"
    ref = O.conv_layer(y.float().cpu(), wn, bn, 1, "SAME", True, storage="fp16")
    np.testing.assert_allclose(got.float().cpu().numpy(), ref.numpy(), rtol=2 ** -8, atol=2e-3)


EXPSQ_CASES = [("fire2-3", 16, 64, 16, 94, 311, 2, False), ("fire2-3-ragged", 16, 64, 16, 21, 45, 3, False), ("fire3p-4", 16, 64, 32, 94, 311, 2, True), ("fire3p-4-odd", 16, 64, 32, 19, 37, 3, True), ("fire4-5", 32, 128, 32, 47, 156, 2, False),
               ("fire4-5-small", 32, 128, 32, 9, 15, 5, False), ("fire5p-6", 32, 128, 48, 47, 156, 2, True), ("fire5p-6-even", 32, 128, 48, 20, 28, 1, True)]


@pytest.mark.parametrize("case", EXPSQ_CASES, ids=[c[0] for c in EXPSQ_CASES])
def test_fire_expand_squeeze_next(case):
    """sqdet_fire_expand_squeeze_next_fwd (expand half of a module from its squeeze tensor, its pool, and the NEXT module's
    squeeze in one streaming launch: fire3+pool3 -> fire4's squeeze, fire4 -> fire5's, fire5+pool5 -> fire6's) BITWISE
    against sqdet_fire_expand_fwd followed by the squeeze conv."""
    ops = _ops()
    name, s, e, s2, H, W, N, pool = case
    tdt = torch.float16
    assert ops.lib().sqdet_fire_expand_squeeze_next_supported(s, e, e, s2, int(pool), 1) == 1
    rs = np.random.RandomState(zlib.crc32(("expsq" + name).encode()) % (2 ** 31))
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).half().float()
    w1, w3, wn = mk(1, s, e), mk(3, s, e), mk(1, 2 * e, s2)
    b1, b3, bn = [torch.from_numpy(rs.uniform(-0.3, 0.3, c).astype(np.float32)).to(DEV) for c in (e, e, s2)]
    p1, p3, pn = [ops.pack_conv_weights(w_.to(DEV), tdt) for w_ in (w1, w3, wn)]
    sq = torch.from_numpy(np.maximum(rs.randn(N, H, W, s), 0).astype(np.float32)).to(DEV, tdt).contiguous()
    got = ops.fire_expand_squeeze_next(sq, p1, b1, p3, b3, pn, bn, pool=pool)
    y = ops.fire_expand(sq, p1, b1, p3, b3, pool=pool)
    want = ops.conv2d_nhwc(y, pn, bn, 1, "SAME", True)
    torch.cuda.synchronize()
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("shape", [(2, 375, 1242), (3, 45, 250), (1, 100, 236), (5, 19, 480)], ids=lambda v: "x".join(map(str, v)))
def test_stem_conv_pool_squeeze(shape):
    """sqdet_stem_conv_pool_squeeze_fwd (conv1 + pool1 + fire2/squeeze1x1 in the persistent stem launch: pool1's tensor is
    never written) BITWISE against the persistent stem followed by the squeeze conv, and against the oracle."""
    ops = _ops()
    N, H, W = shape
    rs = np.random.RandomState(H * 3 + W)
    x = torch.from_numpy(rs.uniform(-2, 2, (N, H, W, 3)).astype(np.float32)).half()
    w = torch.from_numpy((rs.randn(3, 3, 3, 64) * (2.0 / 27) ** 0.5).astype(np.float32)).half().float()
    b = torch.from_numpy(rs.uniform(-0.5, 0.5, 64).astype(np.float32))
    ws = torch.from_numpy((rs.randn(1, 1, 64, 16) * (2.0 / 64) ** 0.5).astype(np.float32)).half().float()
    bs = torch.from_numpy(rs.uniform(-0.3, 0.3, 16).astype(np.float32))
    assert ops.lib().sqdet_stem_conv_pool_squeeze_supported(H, W, 64, 3, 0, 0, 16, 1, N) == 1
    p, ps = ops.pack_conv_weights(w.to(DEV), torch.float16), ops.pack_conv_weights(ws.to(DEV), torch.float16)
    xd = x.to(DEV).contiguous()
    got = ops.stem_conv_pool_squeeze(xd, p, b.to(DEV), ps, bs.to(DEV))
    pool1 = ops.stem_conv_pool(xd, p, b.to(DEV))
    want = ops.conv2d_nhwc(pool1, ps, bs.to(DEV), 1, "SAME", True)
    torch.cuda.synchronize()
    assert got.shape == want.shape and torch.equal(got, want)
    ref = O.conv_layer(O.pooling_layer(O.conv_layer(x.float(), w, b, 2, "SAME", True, storage="fp16"), 3, 2, "SAME"), ws, bs, 1, "SAME", True, storage="fp16")
    np.testing.assert_allclose(got.float().cpu().numpy(), ref.numpy(), rtol=2 ** -8, atol=2e-3)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("ties", [False, True], ids=["spread", "mass-ties"])
def test_detect_filter_one_launch(dtype, ties):
    """sqdet_detect_filter (interpret_output + the top-N branch of filter_prediction in one launch: scores on the fly, boxes
    and classes decoded for the selected anchors only) gives EXACTLY what sqdet_interpret_output -> sqdet_filter_prediction
    give -- also when thousands of anchors tie at the selection boundary (constant score map: the radix-select fallback
    reads the scores back from the scratch)."""
    ops = _ops()
    mc = O.kitti_squeezeDet_config()
    n, gh, gw, K, C = 3, 24, 78, mc.ANCHOR_PER_GRID, mc.CLASSES
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    rs = np.random.RandomState(31 + ties)
    preds = (rs.randn(n, gh, gw, K * (C + 5)) * 2.0).astype(np.float32)
    if ties:
        preds[1, :, :, :K * (C + 1)] = 0.25            # image 1: every anchor has the same score
    pd = torch.from_numpy(preds).to(DEV, tdt).contiguous()
    anchors = torch.from_numpy(np.asarray(mc.ANCHOR_BOX).astype(np.float32)).to(DEV)
    boxes, probs, cls = ops.interpret_output(pd, anchors, C, K, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH)[:3]
    want = ops.filter_prediction(boxes, probs, cls, C, mc.TOP_N_DETECTION, mc.NMS_THRESH, mc.PROB_THRESH)
    got = ops.detect_filter(pd, anchors, C, K, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH, mc.TOP_N_DETECTION, mc.NMS_THRESH)
    torch.cuda.synchronize()
    for g_, w_ in zip(got, want):
        assert torch.equal(g_, w_)
    assert int(got[4].min()) >= 1


@pytest.mark.parametrize("shape", [(3, 24, 78), (2, 22, 76), (1, 5, 19), (32, 22, 76)], ids=lambda s: "%dx%dx%d" % s)
def test_convdet_score_epilogue_bitwise(shape):
    """sqdet_convdet_fwd (the ConvDet launch with the score half of interpret_output in its epilogue): preds are BITWISE those
    of sqdet_conv2d_nhwc_fwd, scores are BITWISE the det_probs sqdet_interpret_output computes from those preds (one shared
    float expression, postproc.h), and sqdet_detect_filter_scored on them gives exactly sqdet_detect_filter's outputs --
    full 22x76 / 24x78 maps, a ragged 5x19 map (edge tiles only), and the benchmark's batch 32."""
    ops = _ops()
    mc = O.kitti_squeezeDet_config()
    n, gh, gw = shape
    K, C = mc.ANCHOR_PER_GRID, mc.CLASSES
    rs = np.random.RandomState(101 + n)
    x = torch.from_numpy(np.maximum(rs.randn(n, gh, gw, 768), 0).astype(np.float32)).to(DEV, torch.float16)
    w = torch.from_numpy((rs.randn(3, 3, 768, 72) * 0.03).astype(np.float32)).to(DEV)
    b = torch.from_numpy((rs.randn(72) * 0.5).astype(np.float32)).to(DEV)
    packed = ops.pack_conv_weights(w, torch.float16)
    assert ops.convdet_scores_supported(768, K, C, torch.float16)
    assert not ops.convdet_scores_supported(768, K, C, torch.float32) and not ops.convdet_scores_supported(768, K, 4, torch.float16)
    want_preds = ops.conv2d_nhwc(x, packed, b, 1, "SAME", False)
    preds, scores = ops.convdet(x, packed, b, K, C)
    torch.cuda.synchronize()
    assert torch.equal(preds, want_preds)
    # anchors of a gh x gw grid are only needed for boxes: any [A,4] array serves the score / pick comparison
    A = gh * gw * K
    anchors = torch.from_numpy(np.abs(rs.randn(A, 4)).astype(np.float32) * 50 + 10).to(DEV)
    probs = ops.interpret_output(preds, anchors, C, K, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH)[1]
    assert scores.shape == probs.shape and torch.equal(scores, probs)
    assert float(scores.max()) > 0.3 and float(scores.min()) >= 0.0
    want = ops.detect_filter(preds, anchors, C, K, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH, mc.TOP_N_DETECTION, mc.NMS_THRESH)
    got = ops.detect_filter(preds, anchors, C, K, mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT, mc.EXP_THRESH, mc.TOP_N_DETECTION, mc.NMS_THRESH,
                            scratch=scores, scores_ready=True)
    torch.cuda.synchronize()
    for g_, w_ in zip(got, want):
        assert torch.equal(g_, w_)


def test_convdet_score_epilogue_rejects_other_heads():
    ops = _ops()
    from squeezedet_amd import _lib
    x = torch.zeros((1, 8, 16, 768), dtype=torch.float32, device=DEV)
    w = torch.zeros((3, 3, 768, 72), dtype=torch.float32, device=DEV)
    packed = ops.pack_conv_weights(w, torch.float32)
    with pytest.raises(_lib.SqdetUnsupported):
        ops.convdet(x, packed, torch.zeros(72, device=DEV), 9, 3)


PHASE_STEM_CASES = [(2, 375, 1242, "SAME", "SAME"), (1, 384, 1248, "SAME", "SAME"), (3, 97, 600, "SAME", "SAME"), (1, 64, 1000, "SAME", "VALID"),
                    (2, 31, 524, "VALID", "SAME"), (1, 200, 2050, "SAME", "SAME"), (5, 19, 786, "VALID", "VALID")]


@pytest.mark.parametrize("sq", [False, True], ids=["plain", "squeeze"])
@pytest.mark.parametrize("case", PHASE_STEM_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_phase_stem_equals_persistent_stem(case, sq):
    """stem4.hip (the default on images >= 523 wide: lane = pooled column, three column phases, lane-local pooling)
    BITWISE against stem3.hip's persistent strip-lane kernel (stem_algo = 3), plain and squeeze forms: full KITTI sizes,
    several images, ragged right / bottom tiles, VALID paddings, an image wider than 32 tiles."""
    ops = _ops()
    N, H, W, cpad, ppad = case
    rs = np.random.RandomState(H + 3 * W)
    x = torch.from_numpy((rs.randint(0, 256, (N, H, W, 3)) - 110.0).astype(np.float32)).to(DEV, torch.float16)
    w = torch.from_numpy((rs.randn(3, 3, 3, 64) * (2.0 / 27) ** 0.5 / 64).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.uniform(-0.5, 0.5, 64).astype(np.float32)).to(DEV)
    ws = torch.from_numpy((rs.randn(1, 1, 64, 16) * 0.2).astype(np.float32)).to(DEV)
    bs = torch.from_numpy(rs.uniform(-0.1, 0.1, 16).astype(np.float32)).to(DEV)
    pk, pks = ops.pack_conv_weights(w, torch.float16), ops.pack_conv_weights(ws, torch.float16)
    run = lambda: ops.stem_conv_pool_squeeze(x, pk, b, pks, bs, cpad, ppad) if sq else ops.stem_conv_pool(x, pk, b, cpad, ppad)
    got = run()
    ops.set_option("stem_algo", 3)
    try:
        want = run()
    finally:
        ops.set_option("stem_algo", 0)
    torch.cuda.synchronize()
    assert got.shape == want.shape and float(want.float().abs().max()) > 0.1
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("case", [("fire2", 64, 16, 64, 94, 311, 2), ("fire4", 128, 32, 128, 47, 156, 3), ("fire6", 256, 48, 192, 24, 78, 2),
                                  ("fire10", 512, 96, 384, 22, 76, 1), ("ragged", 128, 32, 128, 13, 21, 2)], ids=lambda c: c[0])
def test_fire_keep_squeeze(case, dtype):
    """sqdet_fire_fwd_keep: the fused fire launch that ALSO writes its squeeze tensor (the training forward) -- y is bitwise
    sqdet_fire_fwd's, the squeeze tensor bitwise the stand-alone squeeze conv's, streaming and tile kernels, both dtypes,
    full and ragged maps."""
    ops = _ops()
    name, cin, s, e, H, W, N = case
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    rs = np.random.RandomState(H + W + cin)
    x = torch.from_numpy(np.maximum(rs.randn(N, H, W, cin), 0).astype(np.float32)).to(DEV, tdt).contiguous()
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).to(DEV)
    ws, w1, w3 = mk(1, cin, s), mk(1, s, e), mk(3, s, e)
    bs, b1, b3 = [torch.from_numpy(rs.uniform(-0.2, 0.2, c).astype(np.float32)).to(DEV) for c in (s, e, e)]
    ps, p1, p3 = [ops.pack_conv_weights(w_, tdt) for w_ in (ws, w1, w3)]
    y, sq = ops.fire(x, ps, bs, p1, b1, p3, b3, keep_squeeze=True)
    y0 = ops.fire(x, ps, bs, p1, b1, p3, b3)
    sq0 = ops.conv2d_nhwc(x, ps, bs, 1, "SAME", True)
    torch.cuda.synchronize()
    assert torch.equal(y, y0) and torch.equal(sq, sq0)
