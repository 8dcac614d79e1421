import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oneshotdet_amd import ops, spec, synth, train
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
B = 8
images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
gtb = np.zeros((B, 6, 4), np.float32)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = g
gt_boxes = torch.from_numpy(gtb).cuda(); gt_count = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
with ops.tuning():
    eng.forward_backward(images, queries, gt_boxes, gt_count)
for _ in range(3):
    eng.train_step(images, queries, gt_boxes, gt_count)
eng.join(); torch.cuda.synchronize()
pg = eng.last_pred_grads
h = "rpn.head."
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for k, name in enumerate(("cls_ctr", "bbox_pred")):
    pc = eng.convs[h + name]
    dys = [pg[l][k] for l in range(5)]
    for tag, xs in (("real dy", dys), ("random dy", [torch.randn_like(d.float()).to(d.dtype) for d in dys]), ("zeros", [torch.zeros_like(d) for d in dys])):
        t = timeit(lambda: eng._dgrad_levels(pc, xs))
        fin = [float(torch.isfinite(x.float()).float().mean()) for x in xs]
        absmax = [float(x.float().abs().max()) for x in xs]
        tiny = [float(((x.float().abs() > 0) & (x.float().abs() < 1.2e-38)).float().mean()) for x in xs]
        print("%s dgrad, %s: %.1f us; finite %s absmax %s denormal share %s" % (name, tag, t, ["%.3f" % f for f in fin], ["%.2g" % a for a in absmax], ["%.4f" % a for a in tiny]))
print({k: v for k, v in ops.ALGO_CACHE.items() if k[0] == "grouped" and k[4] == 64})
print({k: v for k, v in ops.SPLIT_CACHE.items() if 64 in k[2][0]})
