"""CPU checks of the ResNet50+ConvDet restatement (oracle/resnet_oracle.py) and of the host graph /
native plan tables for it (no GPU compute): SURVEY.md section 8 row a6, BASELINE.json config 5."""
import ctypes as C

import numpy as np
import torch

from oracle import resnet_oracle as R
from oracle import sqdet_oracle as O


def test_param_table_matches_reference_counts():
    """9 191 624 conv params by the formula of nn_skeleton.py:549-551 ((1+k*k*cin)*cout
    per conv), 13 bottlenecks (res2a-c, res3a-d, res4a-f) = 1 + 3 + 13*3 + 1 = 44 convs."""
    shapes = R.param_shapes()
    kernels = [s for n, s in shapes.items() if n.endswith("/kernels")]
    assert len(kernels) == 1 + 3 + 13 * 3 + 1
    assert sum((1 + k * k2 * ci) * co for k, k2, ci, co in kernels) == 9191624
    # variable creation order of one _conv_bn_layer (nn_skeleton.py:427-439)
    names = list(shapes)
    assert names[:6] == ["conv1/kernels", "conv1/biases", "conv1/gamma", "conv1/beta", "conv1/mean", "conv1/var"]
    assert "conv2_x/res2a/res2a_branch2/res2a_branch2a/kernels" in shapes
    assert "conv4_x/res4f/res4f_branch2/res4f_branch2c/var" in shapes
    assert shapes["conv5/kernels"] == (3, 3, 1024, 72)


def test_folded_equals_unfolded_and_float64():
    """The BN fold is the same function as conv -> bias -> batch_normalization (float32 op for op),
    and both agree with a float64 evaluation of the graph."""
    params = R.init_params(seed=3)
    x = O.synthetic_images(1, 96, 160, seed=5)
    col_a, col_b = {}, {}
    a = R.forward(params, x, "fp32", collect=col_a)
    b = R.forward(params, x, "fp32", collect=col_b, folded=True)
    ref = R.forward_float64(params, x)
    assert a.shape == (1, 6, 10, 72)
    scale = float(ref.abs().max())
    assert scale > 0.5          # activations neither collapse nor blow up through 13 residual blocks
    assert float((a.double() - ref).abs().max()) <= 1e-4 * scale
    assert float((b.double() - ref).abs().max()) <= 1e-4 * scale
    for name in col_a:
        s = float(col_a[name].abs().max())
        assert 1e-2 < s < 1e3, (name, s)
        assert float((col_a[name] - col_b[name]).abs().max()) <= 1e-4 * s, name


def test_fp16_storage_model_is_close_to_fp32():
    params = R.init_params(seed=3)
    x = O.synthetic_images(1, 96, 160, seed=5)
    a = R.forward(params, x, "fp32")
    h = R.forward(params, O._round_storage(x, "fp16"), "fp16")
    assert float((a - h).abs().max()) <= 3e-2 * float(a.abs().max())


def test_grid_of_the_reference_input():
    import squeezedet_amd as S
    mc = S.kitti_res50_config_for_input(375, 1242)
    ref = S.kitti_res50_config()
    assert mc.ANCHORS == ref.ANCHORS == 24 * 78 * 9
    assert np.array_equal(mc.ANCHOR_BOX, ref.ANCHOR_BOX)
    x = torch.zeros(1, 375, 1242, 3)
    t = O.conv_layer(x, torch.zeros(7, 7, 3, 4), torch.zeros(4), 2, "SAME", True)
    assert tuple(t.shape[1:3]) == (188, 621)         # conv1 7x7/s2 SAME pad (3,3,2,3)
    assert O.same_pads(375, 7, 2) == (3, 3) and O.same_pads(1242, 7, 2) == (2, 3)
    t = O.pooling_layer(t, 3, 2, "VALID")
    assert tuple(t.shape[1:3]) == (93, 310)


def test_host_graph_and_native_plan_tables():
    """The python builder graph (nets.ResNet50ConvDet) and the native plan (SQDET_ARCH_RESNET50) declare
    the oracle's variables, in the reference's creation order; FLOPs = 61.13 G/img."""
    import squeezedet_amd as S
    from squeezedet_amd import _lib, nets
    mc = S.kitti_res50_config()
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = 1
    m = nets.ResNet50ConvDet(mc, gpu_id="0", dtype=torch.float16)
    shapes = R.param_shapes()
    assert list(m.params) == list(shapes)
    assert all(tuple(m.params[n].shape) == shapes[n] for n in shapes)
    assert m.preds.get_shape() == (1, 24, 78, 72)
    assert sum(c for _, c in m.model_size_counter) == 9191624
    # conv1..res3d frozen, res4* + conv5 trained (resnet50_convDet.py:41-118); mean/var never trained
    assert not m.trainable["conv3_x/res3d/res3d_branch2/res3d_branch2c/kernels"]
    assert m.trainable["conv4_x/res4a/res4a_branch1/gamma"] and not m.trainable["conv4_x/res4a/res4a_branch1/mean"]
    assert m.trainable["conv5/kernels"]

    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.sqdet_net_create(C.byref(h), _lib.ARCH_RESNET50, _lib.F16, 8, 375, 1242, 3, 9) == 0
    name, shape, nd = C.create_string_buffer(128), (C.c_int * 4)(), C.c_int()
    plan = []
    for i in range(lib.sqdet_net_num_params(h)):
        assert lib.sqdet_net_param_info(h, i, name, 128, shape, C.byref(nd)) == 0
        plan.append((name.value.decode(), tuple(shape[j] for j in range(nd.value))))
    assert plan == [(n, tuple(s)) for n, s in shapes.items()]
    gh, gw, ch = C.c_int(), C.c_int(), C.c_int()
    assert lib.sqdet_net_output_dims(h, C.byref(gh), C.byref(gw), C.byref(ch)) == 0
    assert (gh.value, gw.value, ch.value) == (24, 78, 72)
    fl, by = C.c_double(), C.c_double()
    tot, nl = 0.0, lib.sqdet_net_num_layers(h)
    assert nl == 1 + 3 + 13 * 3 + 1          # conv1+pool1 (one fused-stem launch), 3 projection shortcuts, 39 branch convs, conv5
    for i in range(nl):
        assert lib.sqdet_net_layer_info(h, i, name, 128, C.byref(fl), C.byref(by)) == 0
        tot += fl.value
    assert abs(tot / 8 - 61.13e9) < 0.05e9
    assert lib.sqdet_net_set_bn_epsilon(h, 1e-5) == 0
    assert lib.sqdet_net_forward(h, None, None, None) == -1          # null pointers are refused, not run
    lib.sqdet_net_destroy(h)
