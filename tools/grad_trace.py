"""GPU box: where does the engine's backward first deviate from the oracle's autograd (fp32, full-size config1)?
Compares d loss / d (logits, centerness, bbox_reg), d / d correlated features and d / d target FPN features per level."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import golden_utils as gu
from oneshotdet_amd import spec, synth, train
from oracle import hotpath_ref as orc

name = sys.argv[1] if len(sys.argv) > 1 else "config1"
B, H, W, S, qh, qw = gu.CASES[name]
img, q = gu.case_inputs(name)
gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
np_sd = synth.make_state_dict(spec.hot_path_shapes())
sd = {k: v.clone().requires_grad_(not spec.is_frozen(k)) for k, v in orc.to_torch_state_dict(np_sd).items()}
images, queries = torch.from_numpy(img), torch.from_numpy(q)
f = orc.backbone(images, sd, "backbone.")
qf = orc.backbone(queries, sd, "supp_backbone.")
pl = orc.query_pool(qf, [(qh, qw)] * (B * S), B)
comb = orc.correlate(f, pl)
for t in list(f) + list(comb):
    t.retain_grad()
lg, br, ct = orc.fcos_head(comb, sd)
for t in list(lg) + list(br) + list(ct):
    t.retain_grad()
c, r, t_, info = orc.fcos_loss(lg, br, ct, gts, focal="cuda")
(c + r + t_).backward()

eng = train.TrainEngine(np_sd, dtype=torch.float32, wgrad_side_stream=not (len(sys.argv) > 2 and sys.argv[2] == "single"))
G = max(len(g) for g in gts)
gtb = torch.zeros(B, G, 4)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = torch.from_numpy(g)
cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
_hf = eng.head_forward


def _stash(feats):
    out, ctxs = _hf(feats)
    eng._dbg_ctx = ctxs
    return out, ctxs


eng.head_forward = _stash
eng.forward_backward(images.cuda(), queries.cuda(), gtb.cuda(), cnt.cuda())
torch.cuda.synchronize()
dq, dP, d_comb, pred_grads = eng._keep[-1]


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


for l in range(5):
    dcc, drg = pred_grads[l]
    dcc, drg = dcc.float().cpu(), drg.float().cpu()
    print("P%d  d logits %.2e  d ctr %.2e  d reg %.2e   d comb %.2e   d feat %.2e" % (
        l + 3, rel(dcc[..., 0], lg[l].grad[:, 0]), rel(dcc[..., 1], ct[l].grad[:, 0]),
        rel(drg[..., :4].permute(0, 3, 1, 2), br[l].grad), rel(d_comb[l].float().cpu().permute(0, 3, 1, 2), comb[l].grad),
        rel(dP[l].float().cpu().permute(0, 3, 1, 2), comb[l].grad * pl[l].detach())), flush=True)
grads = eng.named_grads()
for k in ("rpn.head.cls_logits.weight", "rpn.head.centerness.weight", "rpn.head.bbox_pred.weight", "rpn.head.cls_tower.9.weight",
          "rpn.head.cls_tower.10.weight", "rpn.head.cls_tower.6.weight", "rpn.head.cls_tower.3.weight", "rpn.head.cls_tower.0.weight",
          "rpn.head.bbox_tower.0.weight", "rpn.head.bbox_tower.9.weight"):
    print("%-34s %.2e" % (k, rel(grads[k].float().cpu(), sd[k].grad)))

# ---- ReLU-mask flips: GroupNorm outputs within rounding of zero can land on different sides of the ReLU in the engine
# (z = fma(u, a, b) with a = gamma * rstd) and in the oracle ((u - mean) * rstd * gamma + beta)
if len(sys.argv) > 3 or True:
    ctxs = eng._dbg_ctx
    for tower in ("cls_tower", "bbox_tower"):
        layers, _ = ctxs[tower]
        for i, (t_in, u, ab) in enumerate(layers):
            gw = eng.extra["rpn.head.%s.%d.weight" % (tower, 3 * i + 1)][0].double().cpu()
            gb = eng.extra["rpn.head.%s.%d.bias" % (tower, 3 * i + 1)][0].double().cpu()
            flips, near = 0, 0
            for l in range(5):
                ul = u[l].double().cpu()                       # [N, H, W, C]
                n_, h_, w_, c_ = ul.shape
                g = ul.view(n_, h_ * w_, 32, c_ // 32)
                mean = g.mean(dim=(1, 3), keepdim=True)
                var = g.var(dim=(1, 3), unbiased=False, keepdim=True)
                z64 = ((g - mean) / torch.sqrt(var + 1e-5)).view(n_, h_, w_, c_) * gw + gb
                a, b = ab[l, 0].float().cpu(), ab[l, 1].float().cpu()      # [N, C]
                z32 = torch.addcmul(b.view(n_, 1, 1, c_), u[l].float().cpu(), a.view(n_, 1, 1, c_))
                flips += int(((z32 > 0) != (z64 > 0)).sum())
                near += int((z64.abs() < 1e-6).sum())
            print("%s layer %d: mask flips vs fp64 %d, |z| < 1e-6: %d" % (tower, i, flips, near))
