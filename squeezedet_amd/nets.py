"""SqueezeDet / SqueezeDet+ model classes with the reference's constructor and graph
(reference src/nets/squeezeDet.py:19-106, src/nets/squeezeDetPlus.py:19-106).  The graph is
declared with the same builder calls as the reference; `preds` is executed by the native plan
(sqdet_net_forward) and, equivalently, op by op through the builders' own HIP kernels."""
import os

import torch

from .nn_skeleton import ModelSkeleton


class _FireNet(ModelSkeleton):
    def __init__(self, mc, gpu_id=0, dtype=torch.float32, seed=0):
        ModelSkeleton.__init__(self, mc, gpu_id, dtype, seed)
        self._add_forward_graph()
        self._add_interpretation_graph()
        # _add_loss_graph / _add_train_graph / _add_viz_graph (nets/squeezeDet.py:26-28) belong to
        # the training path, which is outside the inference hot path built here.

    def _load_pretrained(self):
        mc = self.mc
        if mc.LOAD_PRETRAINED_MODEL:
            assert os.path.exists(mc.PRETRAINED_MODEL_PATH), \
                "Cannot find pretrained model at the given path:  {}".format(mc.PRETRAINED_MODEL_PATH)
            import joblib
            self.caffemodel_weight = joblib.load(mc.PRETRAINED_MODEL_PATH)

    def _fire_layer(self, layer_name, inputs, s1x1, e1x1, e3x3, stddev=0.01, freeze=False):
        """Fire layer constructor (nets/squeezeDet.py:81-106)."""
        sq1x1 = self._conv_layer(layer_name + "/squeeze1x1", inputs, filters=s1x1, size=1, stride=1,
                                 padding="SAME", stddev=stddev, freeze=freeze)
        ex1x1 = self._conv_layer(layer_name + "/expand1x1", sq1x1, filters=e1x1, size=1, stride=1,
                                 padding="SAME", stddev=stddev, freeze=freeze)
        ex3x3 = self._conv_layer(layer_name + "/expand3x3", sq1x1, filters=e3x3, size=3, stride=1,
                                 padding="SAME", stddev=stddev, freeze=freeze)
        return self._concat([ex1x1, ex3x3], 3, name=layer_name + "/concat")


class SqueezeDet(_FireNet):
    NATIVE_ARCH = "squeezeDet"

    def _add_forward_graph(self):
        """NN architecture (nets/squeezeDet.py:30-79)."""
        mc = self.mc
        self._load_pretrained()
        conv1 = self._conv_layer("conv1", self.image_input, filters=64, size=3, stride=2, padding="SAME", freeze=True)
        pool1 = self._pooling_layer("pool1", conv1, size=3, stride=2, padding="SAME")
        fire2 = self._fire_layer("fire2", pool1, s1x1=16, e1x1=64, e3x3=64, freeze=False)
        fire3 = self._fire_layer("fire3", fire2, s1x1=16, e1x1=64, e3x3=64, freeze=False)
        pool3 = self._pooling_layer("pool3", fire3, size=3, stride=2, padding="SAME")
        fire4 = self._fire_layer("fire4", pool3, s1x1=32, e1x1=128, e3x3=128, freeze=False)
        fire5 = self._fire_layer("fire5", fire4, s1x1=32, e1x1=128, e3x3=128, freeze=False)
        pool5 = self._pooling_layer("pool5", fire5, size=3, stride=2, padding="SAME")
        fire6 = self._fire_layer("fire6", pool5, s1x1=48, e1x1=192, e3x3=192, freeze=False)
        fire7 = self._fire_layer("fire7", fire6, s1x1=48, e1x1=192, e3x3=192, freeze=False)
        fire8 = self._fire_layer("fire8", fire7, s1x1=64, e1x1=256, e3x3=256, freeze=False)
        fire9 = self._fire_layer("fire9", fire8, s1x1=64, e1x1=256, e3x3=256, freeze=False)
        fire10 = self._fire_layer("fire10", fire9, s1x1=96, e1x1=384, e3x3=384, freeze=False)
        fire11 = self._fire_layer("fire11", fire10, s1x1=96, e1x1=384, e3x3=384, freeze=False)
        dropout11 = self._dropout(fire11, self.keep_prob, name="drop11")
        num_output = mc.ANCHOR_PER_GRID * (mc.CLASSES + 1 + 4)
        self.preds = self._conv_layer("conv12", dropout11, filters=num_output, size=3, stride=1, padding="SAME",
                                      xavier=False, relu=False, stddev=0.0001)


class ResNet50ConvDet(_FireNet):
    """ResNet50 conv1..res4f + ConvDet (nets/resnet50_convDet.py:20-169)."""
    NATIVE_ARCH = "resnet50"

    def _add_forward_graph(self):
        """NN architecture (nets/resnet50_convDet.py:31-132)."""
        mc = self.mc
        self._load_pretrained()
        conv1 = self._conv_bn_layer(self.image_input, "conv1", "bn_conv1", "scale_conv1", filters=64, size=7, stride=2,
                                    freeze=True, conv_with_bias=True)
        pool1 = self._pooling_layer("pool1", conv1, size=3, stride=2, padding="VALID")
        # (stage scope, block names, branch2a/b filters, output filters, frozen) -- :47-118
        stages = [("conv2_x", ["2a", "2b", "2c"], 64, 256, True),
                  ("conv3_x", ["3a", "3b", "3c", "3d"], 128, 512, True),
                  ("conv4_x", ["4a", "4b", "4c", "4d", "4e", "4f"], 256, 1024, False)]
        x = pool1
        for scope, blocks, in_f, out_f, freeze in stages:
            with self.variable_scope(scope):
                for i, n in enumerate(blocks):
                    with self.variable_scope("res" + n):
                        first = i == 0
                        down = first and scope != "conv2_x"
                        shortcut = x
                        if first:   # projection shortcut: 1x1, stride on the shortcut too, no relu (:51-53,71-73,97-99)
                            shortcut = self._conv_bn_layer(x, "res%s_branch1" % n, "bn%s_branch1" % n, "scale%s_branch1" % n,
                                                           filters=out_f, size=1, stride=2 if down else 1, freeze=freeze,
                                                           relu=False)
                        branch2 = self._res_branch(x, layer_name=n, in_filters=in_f, out_filters=out_f, down_sample=down,
                                                   freeze=freeze)
                        x = self._add_relu(shortcut, branch2, name="res" + n)
        dropout4 = self._dropout(x, self.keep_prob, name="drop4")
        num_output = mc.ANCHOR_PER_GRID * (mc.CLASSES + 1 + 4)
        self.preds = self._conv_layer("conv5", dropout4, filters=num_output, size=3, stride=1, padding="SAME",
                                      xavier=False, relu=False, stddev=0.0001)

    def _res_branch(self, inputs, layer_name, in_filters, out_filters, down_sample=False, freeze=False):
        """Residual branch constructor (nets/resnet50_convDet.py:134-169): 1x1 (stride 2 when
        down-sampling) -> 3x3 -> 1x1 without relu, each conv + frozen BN."""
        with self.variable_scope("res" + layer_name + "_branch2"):
            stride = 2 if down_sample else 1
            n = layer_name
            out = self._conv_bn_layer(inputs, "res%s_branch2a" % n, "bn%s_branch2a" % n, "scale%s_branch2a" % n,
                                      filters=in_filters, size=1, stride=stride, freeze=freeze)
            out = self._conv_bn_layer(out, "res%s_branch2b" % n, "bn%s_branch2b" % n, "scale%s_branch2b" % n,
                                      filters=in_filters, size=3, stride=1, freeze=freeze)
            out = self._conv_bn_layer(out, "res%s_branch2c" % n, "bn%s_branch2c" % n, "scale%s_branch2c" % n,
                                      filters=out_filters, size=1, stride=1, freeze=freeze, relu=False)
            return out


class SqueezeDetPlus(_FireNet):
    NATIVE_ARCH = "squeezeDet+"

    def _add_forward_graph(self):
        """NN architecture (nets/squeezeDetPlus.py:30-79)."""
        mc = self.mc
        self._load_pretrained()
        conv1 = self._conv_layer("conv1", self.image_input, filters=96, size=7, stride=2, padding="VALID", freeze=True)
        pool1 = self._pooling_layer("pool1", conv1, size=3, stride=2, padding="VALID")
        fire2 = self._fire_layer("fire2", pool1, s1x1=96, e1x1=64, e3x3=64, freeze=False)
        fire3 = self._fire_layer("fire3", fire2, s1x1=96, e1x1=64, e3x3=64, freeze=False)
        fire4 = self._fire_layer("fire4", fire3, s1x1=192, e1x1=128, e3x3=128, freeze=False)
        pool4 = self._pooling_layer("pool4", fire4, size=3, stride=2, padding="VALID")
        fire5 = self._fire_layer("fire5", pool4, s1x1=192, e1x1=128, e3x3=128, freeze=False)
        fire6 = self._fire_layer("fire6", fire5, s1x1=288, e1x1=192, e3x3=192, freeze=False)
        fire7 = self._fire_layer("fire7", fire6, s1x1=288, e1x1=192, e3x3=192, freeze=False)
        fire8 = self._fire_layer("fire8", fire7, s1x1=384, e1x1=256, e3x3=256, freeze=False)
        pool8 = self._pooling_layer("pool8", fire8, size=3, stride=2, padding="VALID")
        fire9 = self._fire_layer("fire9", pool8, s1x1=384, e1x1=256, e3x3=256, freeze=False)
        fire10 = self._fire_layer("fire10", fire9, s1x1=384, e1x1=256, e3x3=256, freeze=False)
        fire11 = self._fire_layer("fire11", fire10, s1x1=384, e1x1=256, e3x3=256, freeze=False)
        dropout11 = self._dropout(fire11, self.keep_prob, name="drop11")
        num_output = mc.ANCHOR_PER_GRID * (mc.CLASSES + 1 + 4)
        self.preds = self._conv_layer("conv12", dropout11, filters=num_output, size=3, stride=1, padding="SAME",
                                      xavier=False, relu=False, stddev=0.0001)
