"""Static enumeration of the conv launches of one hot-path forward (same order as model.run_backbone / run_head),
with their implicit-GEMM shapes and algorithmic FLOPs (2*M*N*K with the REAL channel counts)."""
from . import spec
from .ops import conv_out


def _backbone(prefix, n, h, w):
    out = []
    ho, wo = conv_out(h, 7, 2, 3), conv_out(w, 7, 2, 3)
    out.append((prefix + "stem", n * ho * wo, 64, 147))
    h, w = conv_out(ho, 3, 2, 1), conv_out(wo, 3, 2, 1)
    cin = 64
    sizes = []
    for si, nblocks in enumerate(spec.STAGE_BLOCKS):
        mid, cout = 64 * 2 ** si, 256 * 2 ** si
        for b in range(nblocks):
            s = 2 if (b == 0 and si > 0) else 1
            h2, w2 = conv_out(h, 1, s, 0), conv_out(w, 1, s, 0)
            p = "%slayer%d.%d." % (prefix, si + 1, b)
            if b == 0:
                out.append((p + "downsample", n * h2 * w2, cout, cin))
            out.append((p + "conv1", n * h2 * w2, mid, cin))
            out.append((p + "conv2", n * h2 * w2, mid, mid * 9))
            out.append((p + "conv3", n * h2 * w2, cout, mid))
            h, w, cin = h2, w2, cout
        sizes.append((h, w))
    (h3, w3), (h4, w4), (h5, w5) = sizes[1], sizes[2], sizes[3]
    out.append((prefix + "fpn_inner4", n * h5 * w5, 256, 2048))
    out.append((prefix + "fpn_layer4", n * h5 * w5, 256, 2304))
    out.append((prefix + "fpn_inner3", n * h4 * w4, 256, 1024))
    out.append((prefix + "fpn_layer3", n * h4 * w4, 256, 2304))
    out.append((prefix + "fpn_inner2", n * h3 * w3, 256, 512))
    out.append((prefix + "fpn_layer2", n * h3 * w3, 256, 2304))
    h6, w6 = conv_out(h5, 3, 2, 1), conv_out(w5, 3, 2, 1)
    h7, w7 = conv_out(h6, 3, 2, 1), conv_out(w6, 3, 2, 1)
    out.append((prefix + "p6", n * h6 * w6, 256, 2304))
    out.append((prefix + "p7", n * h7 * w7, 256, 2304))
    return out, [(h3, w3), (h4, w4), (h5, w5), (h6, w6), (h7, w7)]


def conv_launches(batch, h, w, n_query, qh, qw):
    """[(name, M, N, K)] in launch order for one forward."""
    tb, levels = _backbone("backbone.", batch, h, w)
    qb, _ = _backbone("supp_backbone.", n_query, qh, qw)
    out = tb + qb
    # FCOS head: the levels share the weights, so each tower layer / prediction conv is ONE grouped launch (M summed)
    m = sum(batch * lh * lw for lh, lw in levels)
    for tower, pred, pn in (("cls_tower", "cls_logits+centerness", 2), ("bbox_tower", "bbox_pred", 4)):
        for i in range(spec.NUM_CONVS):
            out.append(("head.P3-P7.%s.%d" % (tower, i), m, 256, 2304))
        out.append(("head.P3-P7.%s" % pred, m, pn, 2304))
    return out


def conv_flops(batch, h, w, n_query, qh, qw):
    return sum(2.0 * m * n * k for _, m, n, k in conv_launches(batch, h, w, n_query, qh, qw))
