"""GPU box: the input pipeline of SURVEY.md 8f #4 (`osd_image_transform` through `oneshotdet_amd.transforms`) measured
against its HBM roofline, with the reference's own CPU chain (Pillow resize -> float -> BGR255 - mean -> padded batch) timed
beside it on the box's host cores.
    python tools/transforms_bench.py [out.md]
Workload: the config of record's sizes (yaml :39-47): 8 COCO-sized (480 x 640) uint8 targets -> 800 x 1066, padded batch
[8, 800, 1088] as the stem conv's NHWC4 bf16 input; 8 support crops 180 x 140 -> short side 200 (max 400).
Algorithmic bytes per image = source bytes read once + destination bytes written once (the horizontal pass's uint8
intermediate is the kernel chain's own traffic and is NOT counted)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from oneshotdet_amd import transforms as T

out_path = sys.argv[1] if len(sys.argv) > 1 else ""
rng = np.random.RandomState(0)
lines = []


def say(s):
    print(s)
    lines.append(s)


def gpu_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3      # us


def cpu_chain(srcs, min_size, max_size, div):
    """The reference's chain with its own third-party pieces: PIL resize (what torchvision 0.2.1 F.resize calls), to_tensor,
    [2, 1, 0] * 255 - mean, zero-padded batch (image_list.py:52-70)."""
    from PIL import Image
    rs = T.Resize(min_size, max_size)
    outs = []
    for a in srcs:
        im = Image.fromarray(a)
        oh, ow = rs.get_size(im.size)
        im = im.resize((ow, oh), Image.BILINEAR)
        t = torch.from_numpy(np.asarray(im)).permute(2, 0, 1).float().div(255)
        t = t[[2, 1, 0]] * 255
        t = t - torch.tensor(T.PIXEL_MEAN).view(3, 1, 1)
        outs.append(t)
    mh = max(t.shape[1] for t in outs)
    mw = max(t.shape[2] for t in outs)
    mh, mw = -(-mh // div) * div, -(-mw // div) * div
    batch = torch.zeros(len(outs), 3, mh, mw)
    for i, t in enumerate(outs):
        batch[i, :, :t.shape[1], :t.shape[2]] = t
    return batch


say("# Input pipeline (`osd_image_transform`: PIL-exact bilinear resize + flip + BGR255 - mean + pad, one chain per image)")
say("")
say("| workload | images | us / batch | images/s | algorithmic MB / batch | GB/s | fraction of 8 TB/s | CPU chain (Pillow + torch, host) ms / batch | GPU / CPU |")
say("|---|---|---|---|---|---|---|---|---|")
for name, n, (h, w), (mn, mx) in (("targets 480x640 -> 800x1066, NHWC4 bf16 batch", 8, (480, 640), (800, 1200)),
                                   ("targets 1080x1920 -> 675x1200, NHWC4 bf16 batch", 8, (1080, 1920), (800, 1200)),
                                   ("supports 180x140 -> 257x200, NHWC4 bf16 batch", 8, (180, 140), (200, 400))):
    srcs = [rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n)]
    dev = [T.DeviceImage(a) for a in srcs]
    rs = T.Resize(mn, mx)
    ims = [rs(d)[0] for d in dev]

    def run():
        return T.collate(ims, 32, stem_dtype=torch.bfloat16)
    packed = run()
    us = gpu_time(run)
    dst_bytes = packed.tensor.numel() * 2
    mb = (n * h * w * 3 + dst_bytes) / 1e6
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        cpu_chain(srcs, mn, mx, 32)
    cpu_ms = (time.perf_counter() - t0) / reps * 1e3
    say("| %s | %d | %.1f | %.0f | %.1f | %.0f | %.3f | %.1f | %.0fx |" % (name, n, us, n / us * 1e6, mb, mb / us * 1e3,
                                                                  mb / us / 8.0, cpu_ms, cpu_ms * 1e3 / us))
if out_path:
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
