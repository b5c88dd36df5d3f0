"""Weight import / export with the reference's variable names and layouts.

The reference keeps parameters in three forms:
  * TF variables named '<scope>/<layer>/kernels' [k,k,Cin,Cout] (HWIO), '<layer>/biases' [Cout] and, for
    _conv_bn_layer, 'gamma' / 'beta' / 'mean' / 'var' (nn_skeleton.py:427-439, 531-536) -- what
    `tf.train.Saver(model.model_params)` writes (demo.py:181-184, train.py:128-131);
  * the ImageNet-pretrained backbones as a joblib pickle {caffe layer name: [W (OIHW), b]} and, for ResNet,
    {bn name: [mean, var]}, {scale name: [gamma, beta]} (nn_skeleton.py:397-412, 493-502);
Here a parameter set is a flat {variable name: float32 array}; files are .npz (NumPy, no pickle needed).  A TF
checkpoint can be converted by anyone who has TensorFlow with
    {v.name: sess.run(v) for v in model.model_params}  ->  from_reference_names()  ->  save_params().
"""
import numpy as np
import torch


def _np(v):
    return v.detach().float().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float32)


def save_params(path, params):
    """params: a model (its .params) or a {name: array/tensor} dict -> .npz with the reference's variable names."""
    d = params.params if hasattr(params, "params") else params
    np.savez(path, **{k: _np(v) for k, v in d.items()})


def load_params(path):
    """.npz written by save_params -> {name: float32 array} for ModelSkeleton.load_params."""
    with np.load(path, allow_pickle=False) as z:
        return {k: np.asarray(z[k], dtype=np.float32) for k in z.files}


def from_reference_names(values):
    """{TF variable name: array} -> our names: drops the ':0' output suffix TF appends to variable names."""
    return {(k[:-2] if k.endswith(":0") else k): _np(v) for k, v in values.items()}


def from_caffe_weights(caffemodel_weight, model):
    """The pretrained-backbone pickle (joblib.load(mc.PRETRAINED_MODEL_PATH)) -> {variable name: array} for every
    variable of `model` it covers: conv kernels OIHW -> HWIO (nn_skeleton.py:496), biases, and for
    _conv_bn_layer the bn / scale blobs (nn_skeleton.py:403-412).  Layers missing from the pickle or with a
    different shape are skipped, as the reference does (:497-502)."""
    cw, out = caffemodel_weight, {}
    for name, p in model.params.items():
        layer, leaf = name.rsplit("/", 1)
        # caffe names: the layer name itself ('fire2/squeeze1x1'), without the tf.variable_scope prefixes of ResNet
        base = layer if layer in cw else layer.rsplit("/", 1)[-1]
        if leaf == "kernels" and base in cw:
            k = np.transpose(np.asarray(cw[base][0]), [2, 3, 1, 0])
            if k.shape == tuple(p.shape):
                out[name] = k.astype(np.float32)
        elif leaf == "biases" and base in cw and len(cw[base]) > 1:
            b = np.asarray(cw[base][1]).reshape(-1)
            if b.shape == tuple(p.shape):
                out[name] = b.astype(np.float32)
        elif leaf in ("mean", "var", "gamma", "beta") and (base.startswith("res") or base == "conv1"):
            # resXY_branchZ -> bnXY_branchZ / scaleXY_branchZ ; conv1 -> bn_conv1 / scale_conv1 (resnet50_convDet.py:41-44,150-168)
            suffix = base[3:] if base.startswith("res") else "_" + base
            blob = cw.get(("bn" if leaf in ("mean", "var") else "scale") + suffix)
            if blob is not None:
                v = np.asarray(blob[{"mean": 0, "var": 1, "gamma": 0, "beta": 1}[leaf]]).reshape(-1)
                if v.shape == tuple(p.shape):
                    out[name] = v.astype(np.float32)
    return out
