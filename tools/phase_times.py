"""GPU box: the main chain's phases alone on an idle GPU (events around each phase, 10 repetitions, tuned kernels) — target backbone
forward, cls + bbox tower forward, loss, tower backward, FPN + backbone backward with the weight gradients inline — against
the multi-stream step.  What does each phase cost when nothing runs beside it?   python tools/phase_times.py"""
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
gtb, gtc = torch.from_numpy(gtb).cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
with ops.tuning():
    eng.train_step(images, queries, gtb, gtc)
for _ in range(3):
    eng.train_step(images, queries, gtb, gtc)
eng.join()
torch.cuda.synchronize()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


x = ops.pack_stem_input(images, eng.dtype) if hasattr(ops, "pack_stem_input") else None
t_bb, (feats_ctx) = timed(lambda: eng._backbones_forward(eng.BBS[:1], (images,)))
(feats,), (tctx,) = feats_ctx
t_q, qout = timed(lambda: eng._backbones_forward(eng.BBS[1:], (queries,)))
print("target backbone + FPN forward alone: %.3f ms; query backbone alone: %.3f ms" % (t_bb, t_q))
# one whole step on ONE stream (no side streams): the sum of all kernels' isolated durations + launch gaps
eng1 = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16, wgrad_side_stream=False)
with ops.tuning():
    eng1.train_step(images, queries, gtb, gtc)
t_one, _ = timed(lambda: eng1.train_step(images, queries, gtb, gtc), reps=10)
t_multi, _ = timed(lambda: eng.train_step(images, queries, gtb, gtc), reps=20)
print("whole step: %.3f ms on one stream, %.3f ms on the engine's streams" % (t_one, t_multi))
