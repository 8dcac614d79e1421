"""GPU box: is BASELINE.json configs[4]'s per-GPU workload (bs 4, 5 shots, 640 / 800 / 1024 short edge cycling) HOST-bound?
For every geometry by itself and for the cycling mix: host time to enqueue a step (clock stops before any wait) against the time the
device needs (sync at the end), back to back; then a cProfile of the host side of the cycling mix (top functions by own time).
python tools/config5_host_probe.py [--profile]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import ops, spec, synth, train

B, S = 4, 5
shapes = ((640, 832), (800, 1024), (1024, 1312))
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
batches = []
for gi, (H, W) in enumerate(shapes):
    images = torch.from_numpy(synth.make_images("bench.target", B, H, W, seed=1000 + 97 * gi)).cuda()
    queries = torch.from_numpy(synth.make_images("bench.query", B * S, 127, 127, seed=1000 + 97 * gi)).cuda()
    gts = synth.make_gt_boxes(B, H, W, seed=1000 + 97 * gi, max_boxes=6)
    gtb = np.zeros((B, 6, 4), np.float32)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = g
    batches.append((images, queries, torch.from_numpy(gtb).cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()))
for bt in batches:
    with ops.tuning():
        eng.forward_backward(*bt)
    torch.cuda.synchronize()
eng.defer_join = True


def run(seq, n):
    for i in range(6):
        eng.train_step(*batches[seq[i % len(seq)]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        eng.train_step(*batches[seq[i % len(seq)]])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


def idle_enqueue(gi, n=8):
    enq, wall = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train_step(*batches[gi])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
    return float(np.median(enq)), float(np.median(wall))


print("cpus visible: %d; loadavg %s" % (len(os.sched_getaffinity(0)), open("/proc/loadavg").read().strip()))
for rep in range(2):
    for gi, (H, W) in enumerate(shapes):
        e, w_ = idle_enqueue(gi)
        h, d = run([gi], 30)
        print("rep %d  %4dx%-4d alone: from idle host enqueue %.2f ms / wall %.2f ms; back to back host returns after %.2f ms/step, device done after %.2f ms/step -> %s"
              % (rep, H, W, e, w_, h, d, "HOST-bound" if h > 0.93 * d else "device-bound"))
    h, d = run([0, 1, 2], 30)
    print("rep %d  cycling mix: host %.2f ms/step, device done after %.2f ms/step (%.1f images/s) -> %s"
          % (rep, h, d, B / d * 1e3, "HOST-bound" if h > 0.93 * d else "device-bound"))
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for i in range(9):
        eng.train_step(*batches[i % 3])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
