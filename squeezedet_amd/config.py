"""`mc` model configurations with the reference's field names (reference
src/config/config.py:10-142, src/config/kitti_*_config.py).  `mc.ANCHOR_BOX` is built
with the closed form  ANCHOR_BOX[(h*W+w)*B+k] = [(w+1)*IMG_W/(W+1), (h+1)*IMG_H/(H+1), aw_k, ah_k]
(float64), which is bit-identical to the reference's reshape/transpose construction
(kitti_squeezeDet_config.py:45-79; checked in tests against vectors produced by the reference).
"""
import numpy as np


class EasyDict(dict):
    """Attribute-access dict (what easydict.EasyDict gives the reference)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


edict = EasyDict


def base_model_config(dataset="PASCAL_VOC"):
    """config/config.py:10-142."""
    assert dataset.upper() == "PASCAL_VOC" or dataset.upper() == "KITTI", \
        "Currently only support PASCAL_VOC or KITTI dataset"
    cfg = edict()
    cfg.DATASET = dataset.upper()
    if cfg.DATASET == "PASCAL_VOC":
        cfg.CLASS_NAMES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
                           "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa",
                           "train", "tvmonitor")
    elif cfg.DATASET == "KITTI":
        cfg.CLASS_NAMES = ("car", "pedestrian", "cyclist")
    cfg.CLASSES = len(cfg.CLASS_NAMES)
    cfg.GRID_POOL_WIDTH = 7
    cfg.GRID_POOL_HEIGHT = 7
    cfg.LEAKY_COEF = 0.1
    cfg.KEEP_PROB = 0.5
    cfg.IMAGE_WIDTH = 224
    cfg.IMAGE_HEIGHT = 224
    cfg.ANCHOR_BOX = []
    cfg.ANCHORS = len(cfg.ANCHOR_BOX)
    cfg.ANCHOR_PER_GRID = -1
    cfg.BATCH_SIZE = 20
    cfg.PROB_THRESH = 0.005
    cfg.PLOT_PROB_THRESH = 0.5
    cfg.NMS_THRESH = 0.2
    cfg.BGR_MEANS = np.array([[[103.939, 116.779, 123.68]]])
    cfg.LOSS_COEF_CONF = 1.0
    cfg.LOSS_COEF_CLASS = 1.0
    cfg.LOSS_COEF_BBOX = 10.0
    cfg.DECAY_STEPS = 10000
    cfg.LR_DECAY_FACTOR = 0.1
    cfg.LEARNING_RATE = 0.005
    cfg.MOMENTUM = 0.9
    cfg.WEIGHT_DECAY = 0.0005
    cfg.LOAD_PRETRAINED_MODEL = True
    cfg.PRETRAINED_MODEL_PATH = ""
    cfg.DEBUG_MODE = False
    cfg.EPSILON = 1e-16
    cfg.EXP_THRESH = 1.0
    cfg.MAX_GRAD_NORM = 10.0
    cfg.DATA_AUGMENTATION = False
    cfg.DRIFT_X = 0
    cfg.DRIFT_Y = 0
    cfg.EXCLUDE_HARD_EXAMPLES = True
    cfg.BATCH_NORM_EPSILON = 1e-5
    cfg.NUM_THREAD = 4
    cfg.QUEUE_CAPACITY = 100
    cfg.IS_TRAINING = False
    return cfg


SQUEEZEDET_ANCHOR_SHAPES = np.array([[36., 37.], [366., 174.], [115., 59.], [162., 87.], [38., 90.],
                                     [258., 173.], [224., 108.], [78., 170.], [72., 43.]])
RES50_ANCHOR_SHAPES = np.array([[94., 49.], [225., 161.], [170., 91.], [390., 181.], [41., 32.],
                                [128., 64.], [298., 164.], [232., 99.], [65., 42.]])


def set_anchors(mc, H=24, W=78, anchor_shapes=SQUEEZEDET_ANCHOR_SHAPES):
    """kitti_squeezeDet_config.py:45-79 in closed form; returns float64 [H*W*B, 4]."""
    B = len(anchor_shapes)
    cx = np.arange(1, W + 1) * float(mc.IMAGE_WIDTH) / (W + 1)
    cy = np.arange(1, H + 1) * float(mc.IMAGE_HEIGHT) / (H + 1)
    out = np.empty((H, W, B, 4), np.float64)
    out[..., 0] = cx[None, :, None]
    out[..., 1] = cy[:, None, None]
    out[..., 2:] = np.asarray(anchor_shapes, np.float64)[None, None, :, :]
    return out.reshape(-1, 4)


def _kitti_common(mc):
    mc.BATCH_SIZE = 20
    mc.WEIGHT_DECAY = 0.0001
    mc.LEARNING_RATE = 0.01
    mc.DECAY_STEPS = 10000
    mc.MAX_GRAD_NORM = 1.0
    mc.MOMENTUM = 0.9
    mc.LR_DECAY_FACTOR = 0.5
    mc.LOSS_COEF_BBOX = 5.0
    mc.LOSS_COEF_CONF_POS = 75.0
    mc.LOSS_COEF_CONF_NEG = 100.0
    mc.LOSS_COEF_CLASS = 1.0
    mc.PLOT_PROB_THRESH = 0.4
    mc.NMS_THRESH = 0.4
    mc.PROB_THRESH = 0.005
    mc.TOP_N_DETECTION = 64
    mc.DATA_AUGMENTATION = True
    mc.DRIFT_X = 150
    mc.DRIFT_Y = 100
    mc.EXCLUDE_HARD_EXAMPLES = False
    return mc


def _finish(mc, H, W, shapes):
    mc.ANCHOR_BOX = set_anchors(mc, H, W, shapes)
    mc.ANCHORS = len(mc.ANCHOR_BOX)
    mc.ANCHOR_PER_GRID = 9
    return mc


def kitti_squeezeDet_config():
    """config/kitti_squeezeDet_config.py:9-43 (network input 1248x384)."""
    mc = base_model_config("KITTI")
    mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT = 1248, 384
    return _finish(_kitti_common(mc), 24, 78, SQUEEZEDET_ANCHOR_SHAPES)


def kitti_squeezeDetPlus_config():
    """config/kitti_squeezeDetPlus_config.py:9-43 (1242x375, 22x76 grid)."""
    mc = base_model_config("KITTI")
    mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT = 1242, 375
    return _finish(_kitti_common(mc), 22, 76, SQUEEZEDET_ANCHOR_SHAPES)


def kitti_res50_config():
    """config/kitti_res50_config.py:9-43."""
    mc = base_model_config("KITTI")
    mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT = 1242, 375
    return _finish(_kitti_common(mc), 24, 78, RES50_ANCHOR_SHAPES)


def kitti_vgg16_config():
    """config/kitti_vgg16_config.py:9-43."""
    mc = base_model_config("KITTI")
    mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT = 1242, 375
    _kitti_common(mc)
    mc.BATCH_SIZE = 5
    return _finish(mc, 24, 78, SQUEEZEDET_ANCHOR_SHAPES)


def kitti_squeezeDet_config_for_input(image_height, image_width):
    """SqueezeDet on another input size (BASELINE.json quotes the metric on 1242x375): the
    grid is what conv1/s2 + three SAME 3x3/s2 pools give, anchors follow the same formula."""
    mc = base_model_config("KITTI")
    mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT = int(image_width), int(image_height)
    gh, gw = int(image_height), int(image_width)
    for _ in range(4):
        gh, gw = -(-gh // 2), -(-gw // 2)
    return _finish(_kitti_common(mc), gh, gw, SQUEEZEDET_ANCHOR_SHAPES)


def kitti_res50_config_for_input(image_height, image_width):
    """ResNet50+ConvDet on another input size: the grid is what conv1 7x7/s2 SAME, pool1 3x3/s2 VALID
    and the stride-2 res3a / res4a blocks give (nets/resnet50_convDet.py:41-99); 375x1242 -> 24x78."""
    mc = base_model_config("KITTI")
    mc.IMAGE_WIDTH, mc.IMAGE_HEIGHT = int(image_width), int(image_height)
    g = []
    for n in (int(image_height), int(image_width)):
        n = -(-n // 2)            # conv1
        n = (n - 3) // 2 + 1      # pool1 VALID
        n = -(-n // 2)            # res3a
        n = -(-n // 2)            # res4a
        g.append(n)
    return _finish(_kitti_common(mc), g[0], g[1], RES50_ANCHOR_SHAPES)
