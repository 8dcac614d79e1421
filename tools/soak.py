"""GPU box: soak test of the bs=8 bf16 training step: N steps on fixed data, the loss must stay finite and fall; every 50
steps the gradient buffer of a forward_backward is compared with a second engine's (same weights) as a race screen."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import ops, spec, synth, train

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = 8
sd = synth.make_state_dict(spec.hot_path_shapes())
eng = train.TrainEngine(sd, dtype=torch.bfloat16, lr=0.0005)
eng.defer_join = not os.environ.get("OSD_NO_DEFER_JOIN")        # bench.py's mode: the step's tail is joined by the next step
ref = train.TrainEngine(sd, dtype=torch.bfloat16, lr=0.0005)
images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
gtb = np.zeros((B, max(len(g) for g in gts), 4), dtype=np.float32)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = g
gt_boxes = torch.from_numpy(gtb).cuda()
gt_count = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
with ops.tuning():
    eng.forward_backward(images, queries, gt_boxes, gt_count)
torch.cuda.synchronize()
first = None
t0 = time.time()
for it in range(steps):
    losses = eng.train_step(images, queries, gt_boxes, gt_count)
    if it % 50 == 0 or it == steps - 1:
        l = losses.cpu().numpy()
        assert np.isfinite(l).all(), (it, l)
        first = l[:3].sum() if first is None else first
        eng.join()                                                # the update / repack of the last buckets may still be in flight
        ref.flat_w.copy_(eng.flat_w)
        ref.repack()
        ga = eng.forward_backward(images, queries, gt_boxes, gt_count)
        gx = eng.flat_g.clone()
        gb = ref.forward_backward(images, queries, gt_boxes, gt_count)
        torch.cuda.synchronize()
        rel = float((gx - ref.flat_g).norm() / ref.flat_g.norm())
        print("step %4d  loss %.4f (cls %.4f reg %.4f ctr %.4f)  two-engine gradient rel. diff %.2e  %.1f s" % (
            it, l[:3].sum(), l[0], l[1], l[2], rel, time.time() - t0), flush=True)
        assert rel < 5e-2, rel
assert l[:3].sum() < first, (first, l[:3].sum())
print("soak ok: %d steps, loss %.4f -> %.4f" % (steps, first, l[:3].sum()))
