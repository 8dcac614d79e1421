"""GPU box: tune the default training workload (bs 8, bf16, 800 x 1024 + 127 x 127) as bench.py does, count how often each tuned shape is
launched in one step, and print for the shapes that weigh most what the isolated timing saw: the choice and the runners-up.
python tools/tune_report.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import ops, spec, synth, train, tuner

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, H, W = 8, 800, 1024
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
images = torch.from_numpy(synth.make_images("bench.target", B, H, W, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
gts = synth.make_gt_boxes(B, H, W, seed=1000, max_boxes=6)
gtb = np.zeros((B, 6, 4), np.float32)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = g
bt = (images, queries, torch.from_numpy(gtb).cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda())
with ops.tuning():
    eng.forward_backward(*bt)
torch.cuda.synchronize()
caches = {"ALGO_CACHE": tuner.ALGO_CACHE, "WGRAD_ALGO_CACHE": tuner.WGRAD_ALGO_CACHE, "SPLIT_CACHE": tuner.SPLIT_CACHE}
for c in caches.values():
    c.hits = {}
    c.census = True
eng.train_step(*bt)
torch.cuda.synchronize()
for c in caches.values():
    c.census = False
out = []
for name, cache in caches.items():
    for key, hits in cache.hits.items():
        log = tuner.TUNE_LOG[name].get(key)
        if not log or key not in cache:
            continue
        cur = cache[key]
        t_cur = min([t for t, a in log if a == cur] or [log[0][0]])
        out.append((hits * t_cur, hits, name, key, cur, t_cur, log))
out.sort(key=lambda r: -r[0])


def nm(name, a):
    if name != "ALGO_CACHE":
        return str(a)
    a0 = a - 1
    return {50: "pred", 56: "deep64x32x8", 57: "deep64x64x5", 58: "deep64x64x8", 59: "deep32x64x8"}.get(a0, "%s.v%d.t%d" % ("dma" if a0 < 32 else "reg", (a0 >> 3) & 3, a0 & 7))


print("ms/step  launches  cache  key  ->  choice us | runners-up us")
for tot, hits, name, key, cur, t_cur, log in out[:rows]:
    alts = "  ".join("%s %.1f" % (nm(name, a), t * 1e3) for t, a in log[:6] if a != cur)
    print("%6.3f  %3d  %s %s\n        -> %s %.1f | %s" % (tot, hits, name.replace("_CACHE", ""), key, nm(name, cur), t_cur * 1e3, alts))
