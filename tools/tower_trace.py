"""GPU box: replay the cls tower's backward of the training engine step by step (fp32, full-size config1 features) against
a float64 torch reference built from the same master weights; prints the relative max error of every intermediate."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
import torch.nn.functional as F
import golden_utils as gu
from oneshotdet_amd import ops, spec, synth, train

name, tower = (sys.argv[1] if len(sys.argv) > 1 else "config1"), (sys.argv[2] if len(sys.argv) > 2 else "cls_tower")
B, H, W, S, qh, qw = gu.CASES[name]
img, q = gu.case_inputs(name)
gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.float32, wgrad_side_stream=False)
G = max(len(g) for g in gts)
gtb = torch.zeros(B, G, 4)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = torch.from_numpy(g)
cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
_hf = eng.head_forward
eng.head_forward = lambda feats: (lambda r: (setattr(eng, "_dbg_ctx", r[1]), r)[1])(_hf(feats))
eng.forward_backward(torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda(), gtb.cuda(), cnt.cuda())
torch.cuda.synchronize()
dq, dP, d_comb, pred_grads = eng._keep[-1]
layers, t_last = eng._dbg_ctx[tower]
gi = 0 if tower == "cls_tower" else 1
h = "rpn.head."
nl = 5


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def nchw(t):
    return t.double().cpu().permute(0, 3, 1, 2).contiguous()


# ---- float64 reference of the tower on the engine's own inputs
feats = [nchw(layers[0][0][l]).requires_grad_(True) for l in range(nl)]
ws = []
for i in range(4):
    c = eng.convs["%s%s.%d" % (h, tower, 3 * i)]
    ws.append((c.w.double().cpu().permute(0, 3, 1, 2).contiguous(), c.b.double().cpu(),
               eng.extra["%s%s.%d.weight" % (h, tower, 3 * i + 1)][0].double().cpu(),
               eng.extra["%s%s.%d.bias" % (h, tower, 3 * i + 1)][0].double().cpu()))
pc = eng.convs[h + ("cls_ctr" if tower == "cls_tower" else "bbox_pred")]
wp, bp = pc.w.double().cpu().permute(0, 3, 1, 2).contiguous(), pc.b.double().cpu()
us, ts = [], []
for l in range(nl):
    t = feats[l]
    ul, tl = [], []
    for i in range(4):
        u = F.conv2d(t, ws[i][0], ws[i][1], padding=1)
        u.retain_grad()
        t = F.relu(F.group_norm(u, 32, ws[i][2], ws[i][3], 1e-5))
        t.retain_grad()
        ul.append(u); tl.append(t)
    out = F.conv2d(t, wp, bp, padding=1)
    dpred = nchw(pred_grads[l][gi])[:, :out.shape[1]]
    if tower == "bbox_tower":
        raise SystemExit("bbox tower: the engine's d pred is taken before the exp(scale * x) epilogue; use cls_tower")
    out.backward(dpred)
    us.append(ul); ts.append(tl)
print("forward: saved conv outputs vs fp64:", ["%.1e" % max(rel(nchw(layers[i][1][l]), us[l][i].detach()) for l in range(nl)) for i in range(4)])

# ---- engine replay
dpred = [pred_grads[l][gi] for l in range(nl)]
d_t = eng._dgrad_levels(pc, dpred)
print("d t3 (dgrad of the prediction conv):", ["%.1e" % rel(nchw(d_t[l]), ts[l][3].grad) for l in range(nl)])
for i in range(3, -1, -1):
    (gw, ggw), (gbeta, ggb) = eng.gn("%s%s.%d" % (h, tower, 3 * i + 1))
    c = eng.convs["%s%s.%d" % (h, tower, 3 * i)]
    t_in, u, ab = layers[i]
    du = ops.groupnorm_relu_bwd_levels(u, d_t, ab, gw, gbeta, torch.zeros_like(ggw), torch.zeros_like(ggb), spec.GN_GROUPS)
    print("layer %d  du (GroupNorm+ReLU backward):" % i, ["%.1e" % rel(nchw(du[l]), us[l][i].grad) for l in range(nl)])
    d_t = eng._dgrad_levels(c, du)
    ref = [ts[l][i - 1].grad if i > 0 else feats[l].grad for l in range(nl)]
    print("layer %d  d input (dgrad):            " % i, ["%.1e" % rel(nchw(d_t[l]), ref[l]) for l in range(nl)])

# ---- where is layer 3's GroupNorm backward off at P4?
i, l = 3, 1
(gw, ggw), (gbeta, ggb) = eng.gn("%s%s.%d" % (h, tower, 3 * i + 1))
t_in, u, ab = layers[i]
d_t3 = eng._dgrad_levels(pc, dpred)
du = ops.groupnorm_relu_bwd_levels(u, d_t3, ab, gw, gbeta, torch.zeros_like(ggw), torch.zeros_like(ggb), spec.GN_GROUPS)
a, b = nchw(du[l]), us[l][i].grad
err = (a - b).abs()
print("P4 layer 3: max |du ref| %.3e, max err %.3e" % (float(b.abs().max()), float(err.max())))
flat = err.flatten().topk(5)
for v, idx in zip(flat.values, flat.indices):
    n_, c_, y_, x_ = np.unravel_index(int(idx), err.shape)
    print("  err %.3e at (n %d, c %d [group %d], y %d, x %d): engine %.4e ref %.4e   d_t %.4e  u %.4e" % (
        float(v), n_, c_, c_ // 8, y_, x_, float(a[n_, c_, y_, x_]), float(b[n_, c_, y_, x_]),
        float(nchw(d_t3[l])[n_, c_, y_, x_]), float(nchw(u[l])[n_, c_, y_, x_])))
# error by group
eg = err.view(1, 32, 8, -1).amax(dim=(2, 3)).flatten()
print("  max err per group:", ["%.1e" % float(x) for x in eg])
ul = nchw(u[l]).view(1, 32, -1)
print("  u: per-group mean", ["%.2f" % float(x) for x in ul.mean(-1).flatten()[:8]], "std", ["%.3f" % float(x) for x in ul.std(-1).flatten()[:8]])
