"""GPU box: tune the bs=8 bf16 training step and print the weight-gradient algorithm the tuner chose per launch shape
(variant, split target code) — what the timed steps of bench.py run with.   python tools/show_wgrad_algos.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import ops, spec, synth, train

B = 8
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
gtb = np.zeros((B, max(len(g) for g in gts), 4), dtype=np.float32)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = g
with ops.tuning():
    eng.train_step(images, queries, torch.from_numpy(gtb).cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda())
torch.cuda.synchronize()
for k, a in ops.WGRAD_ALGO_CACHE.items():
    kind = k[0] if isinstance(k[0], str) else "single"
    npix = 0
    try:
        shapes = k[2] if kind in ("mixed",) else None
        if shapes:
            npix = sum(int(np.prod(s[1][:3])) for s in shapes)
    except Exception:
        pass
    print("%-8s variant %2d  target code %d   segments/pixels %s   key %s" % (kind, (a - 1) & 15, (a - 1) >> 4, npix or "", str(k)[:110]))
