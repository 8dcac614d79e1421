"""GPU box: the training proposal pipeline (score / decode, head sort, NMS; model.run_proposals) on saved head outputs — a fresh model's
and one trained for N steps on one batch (tools/sustained.py with OSD_SAVE_HEAD) — timed with HIP events, with and without the depth
feedback warmed up.  Under `rocprofv3 --kernel-trace --stats` the same script gives the per-kernel split.
python tools/proposals_probe.py <head.pt> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import model, spec

head = [(a.cuda().bfloat16(), b.cuda().bfloat16()) for a, b in torch.load(sys.argv[1])]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
depth = model.ProposalDepth()
for phase in ("cold hint", "warm hint"):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        pb, ps, pc = model.run_proposals(head, 800, 1024, spec.PRE_NMS_TOP_N_TRAIN, spec.POST_NMS_TOP_N_TRAIN, spec.NMS_THRESH, depth=depth)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("%s: %.2f ms per call (min %.2f), depth hint now %d, proposals kept per image %s" % (phase, sorted(ts)[len(ts) // 2], min(ts), depth.hint, pc.tolist()))
s = torch.cat([torch.sigmoid(a[..., 0].float()).reshape(a.shape[0], -1) * torch.sigmoid(a[..., 1].float()).reshape(a.shape[0], -1) for a, _ in head], 1)
print("score quantiles per image (0.5 / 0.9 / 0.99 / max):", [[round(float(v), 4) for v in torch.quantile(r, torch.tensor([0.5, 0.9, 0.99, 1.0], device=r.device))] for r in s[:2]])
# the two-call form (rank every candidate, then NMS on the sorted list): where the time of a full-depth call goes
from oneshotdet_amd import ops
n = head[0][0].shape[0]
sizes = [(c.shape[1], c.shape[2]) for c, _ in head]
total = sum(h * w for h, w in sizes)
scores = torch.empty((n, total), device="cuda", dtype=torch.float32)
boxes = torch.empty((n, total, 4), device="cuda", dtype=torch.float32)
off, offs = 0, []
for (cls_ctr, reg), stride in zip(head, spec.FPN_STRIDES):
    ops.fcos_score_decode(cls_ctr, reg, scores, boxes, stride, off, 800, 1024, None)
    offs.append(off)
    off += cls_ctr.shape[1] * cls_ctr.shape[2]
levels = [(lo, h * w) for (h, w), lo in zip(sizes, offs)]
max_count = sum(min(c, spec.PRE_NMS_TOP_N_TRAIN) for _, c in levels)
for rep in range(3):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    bs, ss, idx, cnt = ops.rank_sort_gather(scores, boxes, max_count, levels, spec.PRE_NMS_TOP_N_TRAIN)
    e[1].record()
    ob, os_, op, oc = ops.nms_sorted(bs, ss, cnt, spec.NMS_THRESH, spec.POST_NMS_TOP_N_TRAIN, cuda_semantics=True)
    e[2].record()
    torch.cuda.synchronize()
    print("two-call form: rank + gather of all %d candidates %.2f ms, NMS (mask + scan) %.2f ms, kept %s" % (total, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), oc.tolist()[:3]))
