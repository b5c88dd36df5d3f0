"""Host logic of squeezedet_amd.weights: variable names / layouts of the reference's parameter
files (nn_skeleton.py:397-412, 493-502, 531-536).  No GPU needed: models are built, not run."""
import numpy as np
import torch


def _model(cls_name, cfg_name):
    import squeezedet_amd as S
    from squeezedet_amd import nets
    mc = getattr(S, cfg_name)()
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = 1
    return getattr(nets, cls_name)(mc, gpu_id="0", dtype=torch.float16)


def test_npz_round_trip_and_tf_names(tmp_path):
    from squeezedet_amd import synthetic, weights
    m = _model("SqueezeDet", "kitti_squeezeDet_config")
    vals = synthetic.synthetic_params(m, seed=3)
    path = str(tmp_path / "w.npz")
    weights.save_params(path, vals)
    back = weights.load_params(path)
    assert list(back) == list(m.params)
    assert all(np.array_equal(back[k], vals[k].numpy()) for k in vals)
    tf_named = {k + ":0": v for k, v in vals.items()}            # what {v.name: sess.run(v)} yields
    conv = weights.from_reference_names(tf_named)
    assert set(conv) == set(m.params) and np.array_equal(conv["fire4/expand3x3/kernels"], vals["fire4/expand3x3/kernels"].numpy())


def test_caffe_pickle_conversion_squeezenet_and_resnet():
    from squeezedet_amd import weights
    rs = np.random.RandomState(0)
    m = _model("SqueezeDet", "kitti_squeezeDet_config")
    # SqueezeNet v1.1 pickle: {layer: [W (OIHW), b]}; conv1 matches, a wrong-shaped fire2 blob is skipped (:497-502)
    cw = {"conv1": [rs.randn(64, 3, 3, 3).astype(np.float32), rs.randn(64).astype(np.float32)],
          "fire2/squeeze1x1": [rs.randn(16, 64, 1, 1).astype(np.float32), rs.randn(16).astype(np.float32)],
          "fire2/expand1x1": [rs.randn(99, 16, 1, 1).astype(np.float32), rs.randn(99).astype(np.float32)]}
    out = weights.from_caffe_weights(cw, m)
    assert set(out) == {"conv1/kernels", "conv1/biases", "fire2/squeeze1x1/kernels", "fire2/squeeze1x1/biases"}
    assert np.array_equal(out["conv1/kernels"], np.transpose(cw["conv1"][0], [2, 3, 1, 0]))
    r = _model("ResNet50ConvDet", "kitti_res50_config")
    cw = {"res4b_branch2a": [rs.randn(256, 1024, 1, 1).astype(np.float32)],
          "bn4b_branch2a": [rs.randn(256).astype(np.float32), rs.rand(256).astype(np.float32)],
          "scale4b_branch2a": [rs.rand(256).astype(np.float32), rs.randn(256).astype(np.float32)],
          "conv1": [rs.randn(64, 3, 7, 7).astype(np.float32), rs.randn(64).astype(np.float32)],
          "bn_conv1": [rs.randn(64).astype(np.float32), rs.rand(64).astype(np.float32)],
          "scale_conv1": [rs.rand(64).astype(np.float32), rs.randn(64).astype(np.float32)]}
    out = weights.from_caffe_weights(cw, r)
    p = "conv4_x/res4b/res4b_branch2/res4b_branch2a/"
    assert set(out) == {p + s for s in ("kernels", "mean", "var", "gamma", "beta")} | {"conv1/" + s for s in ("kernels", "biases", "mean", "var", "gamma", "beta")}
    assert np.array_equal(out[p + "var"], cw["bn4b_branch2a"][1]) and np.array_equal(out[p + "gamma"], cw["scale4b_branch2a"][0])
    assert np.array_equal(out["conv1/beta"], cw["scale_conv1"][1])
    assert out[p + "kernels"].shape == (1, 1, 1024, 256)
