"""GPU box: host time to ENQUEUE one training step (bs 8, bf16) versus the step's wall time: after a device sync the host enqueues
one step and the clock stops before any wait; then the same with the sync at the end.  python tools/host_time.py"""
import os, sys, time
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
gt_boxes = torch.from_numpy(gtb).cuda()
gt_count = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
with ops.tuning():
    eng.train_step(images, queries, gt_boxes, gt_count)
for _ in range(5):
    eng.train_step(images, queries, gt_boxes, gt_count)
torch.cuda.synchronize()
enq, wall = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_step(images, queries, gt_boxes, gt_count)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
print("one step from an idle device: host enqueue %.2f ms (min %.2f), wall %.2f ms" % (np.median(enq), min(enq), np.median(wall)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.train_step(images, queries, gt_boxes, gt_count)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("20 steps back to back: host returns after %.2f ms/step, device done after %.2f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
if os.environ.get("OSD_CPROFILE"):      # where the host time goes: the top functions by own time over 9 steps
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(9):
        eng.train_step(images, queries, gt_boxes, gt_count)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(int(os.environ.get("OSD_CPROFILE")) if os.environ["OSD_CPROFILE"].isdigit() and int(os.environ["OSD_CPROFILE"]) > 1 else 40)
