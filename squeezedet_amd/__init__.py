"""squeezedet_amd: the SqueezeDet detection hot path (fire-module / ResNet50 backbone, ConvDet,
interpret_output, filter_prediction, and the training step) as hand-written HIP kernels for MI355X
(gfx950) behind the reference's Python surface.  See DESIGN.md / INTEGRATION.md."""
from .config import (base_model_config, kitti_res50_config, kitti_res50_config_for_input,  # noqa: F401
                     kitti_squeezeDet_config, kitti_squeezeDet_config_for_input, kitti_squeezeDetPlus_config,
                     kitti_vgg16_config)


def __getattr__(name):  # lazy: importing the package must not need torch / a GPU
    if name in ("SqueezeDet", "SqueezeDetPlus", "ResNet50ConvDet"):
        from . import nets
        return getattr(nets, name)
    if name in ("ModelSkeleton", "Session"):
        from . import nn_skeleton
        return getattr(nn_skeleton, name)
    raise AttributeError(name)
