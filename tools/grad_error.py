import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import golden_utils as gu
from oneshotdet_amd import spec, synth, train
for name in ("small", "config1"):
    f = gu.load("train_%s.npz" % name)
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
    eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.float32)
    G = max(len(g) for g in gts)
    gtb = torch.zeros(B, G, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
    losses = eng.forward_backward(torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda(), gtb.cuda(), cnt.cuda()).cpu().numpy()
    print(name, "loss rel err", np.abs(losses[:3] - f["losses_cuda_formula"]) / np.abs(f["losses_cuda_formula"]))
    grads = eng.named_grads()
    worst = 0
    for key in f.files:
        if key.startswith("fullgrad_oracle.") and key.endswith(".samples"):
            k = key[len("fullgrad_oracle."):-len(".samples")]
            g = grads[k].float().cpu().numpy().reshape(-1)
            idx = gu.sample_indices(g.size, "grad." + k)[:256]
            scale = float(f["fullgrad_oracle.%s.absmax" % k])
            e = np.abs(g[idx] - f[key]).max() / scale
            worst = max(worst, e)
            print("   %-45s %.2e" % (k, e))
    print(name, "worst", worst)
