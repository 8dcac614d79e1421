"""GPU box: time every weight-gradient variant x split target on the shapes that dominate the training step (bf16, bs=8).
    python tools/wgrad_variants.py [tower|backbone|all]
Prints, per shape, the best time of every variant (and its split-target code) in us and TFLOP/s."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oneshotdet_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if os.environ.get("WG_ORDERED"):          # ordered mode: partial tiles stored + fixed-order reduction pass
    ops.wgrad_set_workspace(nbytes=2 << 30)
    print("ordered mode")
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)


def rnd(*shape):
    return (torch.randn(*shape, device="cuda", generator=g) * 0.5).to(dt)


def time_it(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def sweep(name, launch, flops, cout, cin, xr=None):
    best = {}
    cands = ops.wgrad_algo_candidates(ops.OSD_BF16, cout, cin)
    if xr is not None:
        cands = cands + ops.wgrad_xr_candidates(ops.OSD_BF16, cout, cin, 3, 3, 1, 1, xr)
    only = [int(v) for v in os.environ.get("WG_VARIANTS", "").split(",") if v]      # WG_VARIANTS=5,13: these variants only
    for algo in cands:
        v, t = (algo - 1) & 15, ((algo - 1) >> 4) & 7
        if algo > 128:
            v += 100
        if only and v not in only:
            continue
        try:
            us = time_it(lambda: launch(algo))
        except Exception as e:
            print("   variant %d target %d failed: %s" % (v, t, e))
            continue
        if os.environ.get("WG_ALL"):
            print("   v%d t%d %7.1f us" % (v, t, us))
        if v not in best or us < best[v][0]:
            best[v] = (us, t)
    print("%-44s %s" % (name, "  ".join("v%d: %6.1f us (t%d) %4.0f TF" % (v, us, t, flops / us / 1e6) for v, (us, t) in sorted(best.items()))))


levels = [(8, 100, 128), (8, 50, 64), (8, 25, 32), (8, 13, 16), (8, 7, 8)]
if which in ("tower", "all"):
    dws = [torch.zeros(256, 3, 3, 256, device="cuda") for _ in range(4)]
    dbs = [torch.zeros(256, device="cuda") if os.environ.get("WG_BIAS") else None for _ in range(4)]      # WG_BIAS=1: with the fused bias gradient
    items = []
    for i in range(4):
        for (n, h, w) in levels:
            items.append((rnd(n, h, w, 256), rnd(n, h, w, 256), dws[i], None, dbs[i]))
    m = sum(n * h * w for n, h, w in levels) * 4
    sweep("tower: 4 convs x 5 levels 3x3 256->256", lambda a: ops.conv2d_wgrad_multi(items, 3, 3, 1, 1, 256, algo=a),
          2.0 * m * 256 * 2304, 256, 256, xr=[w for _, _, w in levels])
if which in ("backbone", "all"):
    for (name, n, h, w, cin, cout, k, stride, pad) in (
            ("layer3 conv2 3x3 256->256 M=25600", 8, 50, 64, 256, 256, 3, 1, 1),
            ("layer3 conv3 1x1 256->1024 M=25600", 8, 50, 64, 256, 1024, 1, 1, 0),
            ("layer3 conv1 1x1 1024->256 M=25600", 8, 50, 64, 1024, 256, 1, 1, 0),
            ("layer4 conv2 3x3 512->512 M=6400", 8, 25, 32, 512, 512, 3, 1, 1),
            ("layer4 conv3 1x1 512->2048 M=6400", 8, 25, 32, 512, 2048, 1, 1, 0),
            ("layer2 conv2 3x3 128->128 M=102400", 8, 100, 128, 128, 128, 3, 1, 1),
            ("layer2 conv3 1x1 128->512 M=102400", 8, 100, 128, 128, 512, 1, 1, 0),
            ("fpn P3 3x3 256->256 M=102400", 8, 100, 128, 256, 256, 3, 1, 1)):
        x = rnd(n, h, w, cin)
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        dy = rnd(n, ho, wo, cout)
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        sweep(name, lambda a: ops.conv2d_wgrad(x, dy, dw, k, k, stride, pad, cout, algo=a), 2.0 * n * ho * wo * cout * cin * k * k,
              cout, cin, xr=[w] if (k, stride, pad) == (3, 1, 1) else None)


def stage_items(cin0, width, cout, blocks, h, w, n=8):
    """The weight gradients of one ResNet stage as the engine queues them (train.py:_flush_wgrads): block 0 = conv1 1x1 s2
    (cin0 -> width), conv2 3x3, conv3 1x1 (width -> cout), downsample 1x1 s2 (cin0 -> cout); later blocks conv1 (cout -> width)."""
    items, fl = [], 0.0
    hi, wi = 2 * h, 2 * w
    def add(x, dy, k, stride, pad, co):
        nonlocal fl
        ci = x.shape[-1]
        items.append((x, dy, torch.zeros(co, k, k, ci, device="cuda"), None, None, k, k, stride, pad, co))
        fl += 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * co * ci * k * k
    for b in range(blocks):
        if b == 0:
            add(rnd(n, hi, wi, cin0), rnd(n, h, w, width), 1, 2, 0, width)
            add(rnd(n, hi, wi, cin0), rnd(n, h, w, cout), 1, 2, 0, cout)
        else:
            add(rnd(n, h, w, cout), rnd(n, h, w, width), 1, 1, 0, width)
        add(rnd(n, h, w, width), rnd(n, h, w, width), 3, 1, 1, width)
        add(rnd(n, h, w, width), rnd(n, h, w, cout), 1, 1, 0, cout)
    return items, fl


if which in ("stages", "all"):
    for name, args in (("layer2 (13 convs, 100x128)", (256, 128, 512, 4, 100, 128)), ("layer3 (19 convs, 50x64)", (512, 256, 1024, 6, 50, 64)),
                       ("layer4 (10 convs, 25x32)", (1024, 512, 2048, 3, 25, 32))):
        items, fl = stage_items(*args)
        sweep(name, lambda a: ops.conv2d_wgrad_mixed(items, algo=a), fl, max(i[9] for i in items), max(i[0].shape[-1] for i in items))
