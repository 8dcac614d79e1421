"""Developer script (GPU box): per-stage error of the HIP engine vs the oracle on a named golden case."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_utils as gu  # noqa: E402
from oneshotdet_amd import model, ops, spec, synth  # noqa: E402
from oracle import hotpath_ref as orc  # noqa: E402


def nchw(t):
    return ops.nhwc_to_nchw_f32(t).cpu()


def err(a, b):
    a, b = a.float(), b.float()
    return "max|d|=%.3e  rel=%.3e  (absmax ref %.3e)" % ((a - b).abs().max().item(),
                                                         (a - b).abs().max().item() / max(b.abs().max().item(), 1e-9),
                                                         b.abs().max().item())


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "small"
    dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    sd = orc.to_torch_state_dict(np_sd)
    with torch.no_grad():
        o = orc.hot_path_forward(torch.from_numpy(img), torch.from_numpy(q), sd, shots=S)
        body = orc.resnet_body(torch.from_numpy(img), sd, "backbone.body.")
    eng = model.HotPathEngine(np_sd, dtype=dtype)
    images, queries = torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda()
    feats, bodyf = model.run_backbone(eng.backbone, images, dtype, return_body=True)
    for i in range(4):
        print("body C%d" % (i + 2), err(nchw(bodyf[i]), body[i]))
    out = eng.detect(images, queries, cuda_nms=False)
    torch.cuda.synchronize()
    for l in range(5):
        print("P%d" % (l + 3), err(nchw(out["features"][l]), o["features"][l]))
    for l in range(5):
        print("Q%d" % (l + 3), err(nchw(out["query_features"][l]), o["query_features"][l]))
    for l in range(5):
        print("pooled%d" % l, err(out["pooled"][l].cpu(), o["pooled"][l].reshape(B, -1)))
        print("combined%d" % l, err(nchw(out["combined"][l]), o["combined"][l]))
    for l in range(5):
        cc, rg = out["head"][l]
        cc, rg = nchw(cc), nchw(rg)
        print("logits%d" % l, err(cc[:, 0:1], o["logits"][l]), "| ctr", err(cc[:, 1:2], o["centerness"][l]),
              "| reg", err(rg, o["bbox_reg"][l]))
    props = orc.fcos_postprocess(o["logits"], o["bbox_reg"], o["centerness"], [(H, W)] * B)
    ob, os_, oc = [t.cpu() for t in out["proposals"]]
    for i in range(B):
        k = int(oc[i])
        frac = gu.match_boxes(props[i][0].numpy(), props[i][1].numpy(), ob[i, :k].numpy(), os_[i, :k].numpy())
        print("image %d: oracle %d proposals, hip %d, overlap %.4f" % (i, len(props[i][0]), k, frac))
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        eng.detect(images, queries)
        torch.cuda.synchronize()
        print("detect wall %.2f ms" % ((time.time() - t0) * 1e3))


if __name__ == "__main__":
    main()
