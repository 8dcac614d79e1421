"""Static description of the siamese-FCOS hot path: parameter names/shapes and the config values it consumes.

The names are the reference's own state_dict keys so a reference `.pth` loads unchanged
(reference: modeling/backbone/resnet.py:80-145,228-293,318-337; fpn.py:28-41,82-94; rpn/fcos/fcos.py:12-81;
config of record configs/fcos/2019_10_25_vanilla_siamse_backbone.yaml + config/defaults.py).
tests/test_spec.py checks this list against tests/golden/state_dict_keys.json, which was dumped from the real
reference model.
"""
from collections import OrderedDict

# ---- config of record (only the values the hot path reads) ----
STEM_OUT = 64
RES2_OUT = 256
STAGE_BLOCKS = (3, 4, 6, 3)           # resnet.py:65-68 (R-50-FPN-RETINANET)
FPN_OUT = 256                         # BACKBONE_OUT_CHANNELS
FPN_STRIDES = (8, 16, 32, 64, 128)    # defaults.py FCOS.FPN_STRIDES
POOLER_SCALES = (0.125, 0.0625, 0.03125, 0.015625, 0.0078125)
POOLER_SAMPLING_RATIO = 2
NUM_CONVS = 4
GN_GROUPS = 32
GN_EPS = 1e-5
PRIOR_PROB = 0.01
PRE_NMS_TOP_N_TEST = 6000
PRE_NMS_TOP_N_TRAIN = 12000
POST_NMS_TOP_N_TEST = 2000
POST_NMS_TOP_N_TRAIN = 4000
NMS_THRESH = 0.8
FREEZE_CONV_BODY_AT = 2               # stem + layer1 frozen (resnet.py:127-136)
LOSS_ALPHA = 0.25
LOSS_GAMMA = 2.0
POS_RADIUS = 1.5
SIZE_DIVISIBILITY = 32
INF = 100000000
# second stage: yaml ROI_BOX_HEAD + defaults.py:196-229,511
BOX_POOL = 7                          # ROI_BOX_HEAD.POOLER_RESOLUTION
BOX_MLP_DIM = 1024                    # ROI_BOX_HEAD.MLP_HEAD_DIM
BOX_NUM_CLASSES = 2                   # ROI_BOX_HEAD.NUM_CLASSES (background, the query's class)
BOX_REG_WEIGHTS = (10.0, 10.0, 5.0, 5.0)   # ROI_HEADS.BBOX_REG_WEIGHTS
BOX_SCORE_THRESH = 0.0                # ROI_HEADS.SCORE_THRESH
BOX_NMS_THRESH = 0.5                  # ROI_HEADS.NMS
BOX_DETECTIONS_PER_IMG = 2000         # ROI_HEADS.DETECTIONS_PER_IMG
BOX_LEAKY_SLOPE = 0.2                 # box_head.py:46,49,64
# second stage, training (defaults.py:190-203, box_head.py:193-194)
BOX_FG_IOU_THRESH = 0.5               # ROI_HEADS.FG_IOU_THRESHOLD == BG_IOU_THRESHOLD
BOX_BATCH_PER_IMAGE = 128             # ROI_HEADS.BATCH_SIZE_PER_IMAGE
BOX_POSITIVE_FRACTION = 0.25          # ROI_HEADS.POSITIVE_FRACTION
BOX_LOSS_WEIGHTS = (5.0, 2.5)         # loss_classifier *= 5; loss_box_reg *= 2.5
LEVEL_MAP_SCALE = 224                 # poolers.py:16 LevelMapper canonical_scale / canonical_level / eps
LEVEL_MAP_LEVEL = 4
LEVEL_MAP_EPS = 1e-6


def _bn(prefix, n, out):
    for k in ("weight", "bias", "running_mean", "running_var"):
        out[prefix + "." + k] = (n,)


def resnet_body_shapes(prefix):
    """ResNet-50 body keys (resnet.py:80-125, Bottleneck :228-293, BaseStem :318-330)."""
    out = OrderedDict()
    out[prefix + "stem.conv1.weight"] = (STEM_OUT, 3, 7, 7)
    _bn(prefix + "stem.bn1", STEM_OUT, out)
    cin = STEM_OUT
    for si, nblocks in enumerate(STAGE_BLOCKS):
        mid = 64 * (2 ** si)
        cout = RES2_OUT * (2 ** si)
        for b in range(nblocks):
            p = "%slayer%d.%d." % (prefix, si + 1, b)
            if b == 0:  # in_channels != out_channels -> downsample branch (resnet.py:243-251)
                out[p + "downsample.0.weight"] = (cout, cin, 1, 1)
                _bn(p + "downsample.1", cout, out)
            out[p + "conv1.weight"] = (mid, cin, 1, 1)
            _bn(p + "bn1", mid, out)
            out[p + "conv2.weight"] = (mid, mid, 3, 3)
            _bn(p + "bn2", mid, out)
            out[p + "conv3.weight"] = (cout, mid, 1, 1)
            _bn(p + "bn3", cout, out)
            cin = cout
    return out


def fpn_shapes(prefix):
    """FPN keys; C2 lateral is skipped (fpn.py:33, backbone.py:59)."""
    out = OrderedDict()
    for idx, cin in ((2, 512), (3, 1024), (4, 2048)):
        out["%sfpn_inner%d.weight" % (prefix, idx)] = (FPN_OUT, cin, 1, 1)
        out["%sfpn_inner%d.bias" % (prefix, idx)] = (FPN_OUT,)
        out["%sfpn_layer%d.weight" % (prefix, idx)] = (FPN_OUT, FPN_OUT, 3, 3)
        out["%sfpn_layer%d.bias" % (prefix, idx)] = (FPN_OUT,)
    for name in ("p6", "p7"):
        out["%stop_blocks.%s.weight" % (prefix, name)] = (FPN_OUT, FPN_OUT, 3, 3)
        out["%stop_blocks.%s.bias" % (prefix, name)] = (FPN_OUT,)
    return out


def backbone_shapes(prefix):
    out = resnet_body_shapes(prefix + "body.")
    out.update(fpn_shapes(prefix + "fpn."))
    return out


def fcos_head_shapes(prefix="rpn.head."):
    """FCOSHead keys (fcos.py:27-81): Sequential indices 0,3,6,9 = conv; 1,4,7,10 = GroupNorm."""
    out = OrderedDict()
    for tower in ("cls_tower", "bbox_tower"):
        for i in range(NUM_CONVS):
            out["%s%s.%d.weight" % (prefix, tower, 3 * i)] = (FPN_OUT, FPN_OUT, 3, 3)
            out["%s%s.%d.bias" % (prefix, tower, 3 * i)] = (FPN_OUT,)
            out["%s%s.%d.weight" % (prefix, tower, 3 * i + 1)] = (FPN_OUT,)
            out["%s%s.%d.bias" % (prefix, tower, 3 * i + 1)] = (FPN_OUT,)
    for name, c in (("cls_logits", 1), ("bbox_pred", 4), ("centerness", 1)):
        out["%s%s.weight" % (prefix, name)] = (c, FPN_OUT, 3, 3)
        out["%s%s.bias" % (prefix, name)] = (c,)
    for i in range(5):
        out["%sscales.%d.scale" % (prefix, i)] = (1,)
    return out


def box_head_shapes(prefix="roi_heads.box."):
    """Second-stage few-shot ROI box head keys (modeling/roi_heads/box_head/box_head.py:40-78: compress_dim_conv =
    Sequential(conv1x1, GN, LeakyReLU, conv1x1, GN, LeakyReLU) -> indices 0,1,3,4; feature_aggreg = Sequential(conv3x3,
    GN, LeakyReLU); fc6/fc7 make_fc; roi_box_predictors.py:37-99 FPNPredictor with 2 classes and 2x4 box deltas)."""
    out = OrderedDict()
    c2 = 2 * FPN_OUT
    out[prefix + "compress_dim_conv.0.weight"] = (c2, c2, 1, 1)
    out[prefix + "compress_dim_conv.0.bias"] = (c2,)
    out[prefix + "compress_dim_conv.1.weight"] = (c2,)
    out[prefix + "compress_dim_conv.1.bias"] = (c2,)
    out[prefix + "compress_dim_conv.3.weight"] = (FPN_OUT, c2, 1, 1)
    out[prefix + "compress_dim_conv.3.bias"] = (FPN_OUT,)
    out[prefix + "compress_dim_conv.4.weight"] = (FPN_OUT,)
    out[prefix + "compress_dim_conv.4.bias"] = (FPN_OUT,)
    out[prefix + "feature_aggreg.0.weight"] = (FPN_OUT // 2, FPN_OUT, 3, 3)
    out[prefix + "feature_aggreg.0.bias"] = (FPN_OUT // 2,)
    out[prefix + "feature_aggreg.1.weight"] = (FPN_OUT // 2,)
    out[prefix + "feature_aggreg.1.bias"] = (FPN_OUT // 2,)
    out[prefix + "fc6.weight"] = (BOX_MLP_DIM, (FPN_OUT // 2) * BOX_POOL ** 2)
    out[prefix + "fc6.bias"] = (BOX_MLP_DIM,)
    out[prefix + "fc7.weight"] = (BOX_MLP_DIM, BOX_MLP_DIM)
    out[prefix + "fc7.bias"] = (BOX_MLP_DIM,)
    out[prefix + "predictor.cls_score.weight"] = (BOX_NUM_CLASSES, BOX_MLP_DIM)
    out[prefix + "predictor.cls_score.bias"] = (BOX_NUM_CLASSES,)
    out[prefix + "predictor.bbox_pred.weight"] = (BOX_NUM_CLASSES * 4, BOX_MLP_DIM)
    out[prefix + "predictor.bbox_pred.bias"] = (BOX_NUM_CLASSES * 4,)
    return out


def hot_path_shapes():
    """All state_dict entries of the hot path: target backbone, query backbone (separate weights,
    generalized_rcnn.py:69-71), FCOS head."""
    out = backbone_shapes("backbone.")
    out.update(backbone_shapes("supp_backbone."))
    out.update(fcos_head_shapes())
    return out


def full_model_shapes():
    """Hot path + second-stage box head = every state_dict entry of the reference model under the config of record."""
    out = hot_path_shapes()
    out.update(box_head_shapes())
    return out


def is_frozen(key):
    """Parameters with requires_grad=False: FrozenBN buffers, stem and layer1 (resnet.py:127-136)."""
    if ".bn" in key or "downsample.1." in key:
        return True
    return ".body.stem." in key or ".body.layer1." in key


def level_sizes(h, w):
    """Spatial sizes of P3..P7 for an (h, w) input that is a multiple of 32: stride-2 convs with pad 1 / k 3
    (P6, P7) give ceil(n/2) (fpn.py:95-99), the body gives exact halvings."""
    assert h % 32 == 0 and w % 32 == 0
    p5 = (h // 32, w // 32)
    p6 = ((p5[0] + 1) // 2, (p5[1] + 1) // 2)
    p7 = ((p6[0] + 1) // 2, (p6[1] + 1) // 2)
    return [(h // 8, w // 8), (h // 16, w // 16), p5, p6, p7]
