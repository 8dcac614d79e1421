"""GPU box: distribution of the gradient errors against the fixtures (tests/golden/train_*.npz) — what the two-tier fp32
bound and the bf16 cosine bound of tests/test_gpu_train.py are set from.
    python tools/grad_stats.py [case ...]
fp32: per tensor the 4 largest |error| / absmax over the 256 sampled elements and how many exceed 1e-4;
bf16: per tensor relative L2 error and cosine."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import golden_utils as gu
from oneshotdet_amd import spec, synth, train

cases = sys.argv[1:] or ["small", "nonsquare", "shots5", "tall", "config1", "config1x2"]
for name in cases:
    if not os.path.exists(os.path.join(gu.GOLDEN_DIR, "train_%s.npz" % name)):
        continue
    f = gu.load("train_%s.npz" % name)
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
    G = max(len(g) for g in gts)
    gtb = torch.zeros(B, G, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
    for dt in (torch.float32, torch.bfloat16):
        eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=dt)
        losses = eng.forward_backward(torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda(), gtb.cuda(), cnt.cuda()).cpu().numpy()
        print("== %s %s: loss rel err %s" % (name, dt, np.abs(losses[:3] - f["losses_cuda_formula"]) / np.abs(f["losses_cuda_formula"])))
        grads = eng.named_grads()
        for key in f.files:
            if key.startswith("fullgrad_oracle.") and key.endswith(".samples"):
                k = key[len("fullgrad_oracle."):-len(".samples")]
                g = grads[k].float().cpu().numpy().reshape(-1)
                idx = gu.sample_indices(g.size, "grad." + k)[:256]
                scale = float(f["fullgrad_oracle.%s.absmax" % k])
                ref = f[key]
                e = np.sort(np.abs(g[idx] - ref) / scale)[::-1]
                cos = float(np.dot(g[idx], ref) / max(np.linalg.norm(g[idx]) * np.linalg.norm(ref), 1e-30))
                l2 = float(np.linalg.norm(g[idx] - ref) / max(np.linalg.norm(ref), 1e-30))
                print("   %-45s top4 %s  n>1e-4: %3d  n>1e-3: %3d  L2 %.2e  cos %.6f  absmax %.2e" % (
                    k, " ".join("%.1e" % v for v in e[:4]), int((e > 1e-4).sum()), int((e > 1e-3).sum()), l2, cos, scale))
        del eng
        torch.cuda.empty_cache()
