"""Developer script (GPU box): training step gradients of the HIP engine vs the oracle's autograd on the small case."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_utils as gu  # noqa: E402
from oneshotdet_amd import spec, synth, train  # noqa: E402
from oracle import hotpath_ref as orc  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "small"
    dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    sd = {k: v.clone().requires_grad_(not spec.is_frozen(k)) for k, v in orc.to_torch_state_dict(np_sd).items()}
    o = orc.hot_path_forward(torch.from_numpy(img), torch.from_numpy(q), sd, shots=S)
    c, r, t, info = orc.fcos_loss(o["logits"], o["bbox_reg"], o["centerness"], gts, focal="cuda")
    (c + r + t).backward()
    print("oracle losses", c.item(), r.item(), t.item(), "num_pos", info["num_pos"])
    eng = train.TrainEngine(np_sd, dtype=dtype)
    G = max(len(g) for g in gts)
    gtb = torch.zeros(B, G, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
    losses = eng.forward_backward(torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda(), gtb.cuda(), cnt.cuda())
    torch.cuda.synchronize()
    print("hip losses", losses.cpu().tolist())
    grads = eng.named_grads()
    worst = []
    for k, g in grads.items():
        ref = sd[k].grad
        if ref is None:
            print("no ref grad for", k)
            continue
        gg = g.float().cpu()
        err = (gg - ref).abs().max().item()
        scale = ref.abs().max().item()
        worst.append((err / max(scale, 1e-12), k, err, scale))
    worst.sort(reverse=True)
    for w in worst[:25]:
        print("rel %.3e  %-55s abs %.3e  refmax %.3e" % w)
    print("...", len(worst), "tensors; median rel", sorted(x[0] for x in worst)[len(worst) // 2])


if __name__ == "__main__":
    main()
