"""GPU box: does the training-proposal path stay on its fast (head-of-the-order) phase while the weights train?  Runs N
train steps of the bench workload and times model.run_proposals on the step's own head outputs after every step (HIP
events, idle stream), next to the number of proposals kept per image."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from oneshotdet_amd import model, ops, spec, synth, train

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
B = 8
images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
gtb = np.zeros((B, 6, 4), np.float32)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = g
gt_boxes = torch.from_numpy(gtb).cuda()
gt_count = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
with ops.tuning():
    eng.forward_backward(images, queries, gt_boxes, gt_count)
torch.cuda.synchronize()
eng.defer_join = True
for step in range(n_steps):
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3):
        losses = eng.train_step(images, queries, gt_boxes, gt_count)
    eng.join()
    t1.record()
    torch.cuda.synchronize()
    step_ms = t0.elapsed_time(t1) / 3
    ho = eng.last_head_out
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    pb, ps, pc = model.run_proposals(ho, 800, 1024, spec.PRE_NMS_TOP_N_TRAIN, spec.POST_NMS_TOP_N_TRAIN, spec.NMS_THRESH)
    b.record()
    torch.cuda.synchronize()
    # depth of the greedy scan: position (in score order) of the last kept box, from the two-call pipeline
    n = ho[0][0].shape[0]
    sizes = [(c.shape[1], c.shape[2]) for c, _ in ho]
    total = sum(h * w for h, w in sizes)
    scores = torch.empty((n, total), device="cuda", dtype=torch.float32)
    boxes = torch.empty((n, total, 4), device="cuda", dtype=torch.float32)
    off, levels = 0, []
    for (cc, rg), stride in zip(ho, spec.FPN_STRIDES):
        ops.fcos_score_decode(cc, rg, scores, boxes, stride, off, 800, 1024, None)
        levels.append((off, cc.shape[1] * cc.shape[2]))
        off += cc.shape[1] * cc.shape[2]
    mc = sum(min(c, spec.PRE_NMS_TOP_N_TRAIN) for _, c in levels)
    bs, ss, idx, cnt = ops.rank_sort_gather(scores, boxes, mc, levels, spec.PRE_NMS_TOP_N_TRAIN)
    ob, os_, op, oc = ops.nms_sorted(bs, ss, cnt, spec.NMS_THRESH, spec.POST_NMS_TOP_N_TRAIN, cuda_semantics=True)
    depth = [int(op[i, :int(oc[i])].max()) + 1 for i in range(n)]
    hint = eng._prop_depth.hint if eng._prop_depth is not None else -1
    print("hint %5d  step %2d  %.2f ms/step  losses %s  proposals %.0f us  depth of the 4000th survivor %s" % (
        hint, 3 * step + 2, step_ms, " ".join("%.3f" % v for v in losses[:3].tolist()), a.elapsed_time(b) * 1e3, depth), flush=True)
