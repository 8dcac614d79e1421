"""Un-profiled per-stream timeline of ONE training step (GPU box): every C-ABI launch is bracketed by HIP events on the
stream it goes to (after that stream's waits have been enqueued), the host runs ahead as in production (the timed step
is preceded by 3 untimed ones and parked behind a spin kernel), and the events are read afterwards.
Prints: per-stream busy time, union busy / idle time of the step, the largest idle windows and what ran around them.
usage: python tools/step_timeline.py [--steps-ahead 1]"""
import argparse
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2, help="instrumented consecutive steps")
    ap.add_argument("--dump", action="store_true", help="list every launch of the main stream (start, duration) and what ran beside it")
    args = ap.parse_args()
    from oneshotdet_amd import _lib, ops, spec, synth, train
    eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
    B = args.batch
    images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
    queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
    gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
    G = max(len(g) for g in gts)
    gtb = torch.zeros(B, G, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
    gtb = gtb.cuda()
    with ops.tuning():
        eng.train_step(images, queries, gtb, cnt)
    torch.cuda.synchronize()
    for _ in range(3):
        eng.train_step(images, queries, gtb, cnt)
    torch.cuda.synchronize()

    rec = []
    real_call = _lib.call

    def timed_call(name, *a):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        real_call(name, *a)
        e1.record(st)
        rec.append((name, st.cuda_stream, e0, e1))
    _lib.call = timed_call
    ops._lib.call = timed_call
    base = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(40e6))
    base.record()
    marks = []
    for _ in range(args.steps):
        eng.train_step(images, queries, gtb, cnt)
        marks.append(len(rec))
    torch.cuda.synchronize()
    _lib.call = real_call
    ops._lib.call = real_call
    ev = [(n, s, base.elapsed_time(a) * 1e3, base.elapsed_time(b) * 1e3) for n, s, a, b in rec]
    streams = {s: i for i, s in enumerate(sorted({e[1] for e in ev}))}
    lo = marks[-2] if len(marks) > 1 else 0
    step = ev[lo:marks[-1]]
    # the step's span on the GPU: from the first start to the last end of the LAST instrumented step
    t0, t1 = min(e[2] for e in step), max(e[3] for e in step)
    print("launches in the step: %d, span %.2f ms (includes overlap with the neighbouring steps' tails)" % (len(step), (t1 - t0) / 1e3))
    per = collections.defaultdict(float)
    for n, s, a, b in step:
        per[streams[s]] += b - a
    print("per-stream busy ms:", {k: round(v / 1e3, 2) for k, v in sorted(per.items())})
    iv = sorted((a, b) for _, _, a, b in ev if b > t0 and a < t1)
    busy, (ca, cb), gaps = 0.0, iv[0], []
    for a, b in iv[1:]:
        if a > cb:
            busy += cb - ca
            gaps.append((cb, a))
            ca, cb = a, b
        else:
            cb = max(cb, b)
    busy += cb - ca
    print("union busy %.2f ms, idle %.2f ms in %d gaps (> 20 us: %d, sum %.2f ms)"
          % (busy / 1e3, sum(b - a for a, b in gaps) / 1e3, len(gaps), sum(1 for a, b in gaps if b - a > 20),
             sum(b - a for a, b in gaps if b - a > 20) / 1e3))
    for a, b in sorted(gaps, key=lambda g: g[0] - g[1])[:12]:
        before = [e for e in ev if abs(e[3] - a) < 0.5]
        after = [e for e in ev if abs(e[2] - b) < 0.5]
        print("  idle %7.1f us at +%.2f ms: after %s, before %s" % (b - a, (a - t0) / 1e3,
              [(e[0], streams[e[1]]) for e in before][:2], [(e[0], streams[e[1]]) for e in after][:2]))
    # main-chain stream = the one with the largest busy time: its own gaps (waiting on other streams or on the host)
    main = max(per, key=per.get)
    ms = sorted((a, b, n) for n, s, a, b in step if streams[s] == main)
    g2 = [(ms[i + 1][0] - ms[i][1], ms[i][2], ms[i + 1][2], ms[i][1]) for i in range(len(ms) - 1)]
    print("main stream %d: %d launches, busy %.2f ms, own gaps %.2f ms" % (main, len(ms), per[main] / 1e3, sum(g[0] for g in g2) / 1e3))
    for g, a, b, at in sorted(g2, reverse=True)[:15]:
        print("  gap %7.1f us at +%.2f ms between %s and %s" % (g, (at - t0) / 1e3, a, b))
    if args.dump:
        for a, b, n in ms:
            beside = collections.Counter()
            for n2, s2, a2, b2 in ev:
                if streams[s2] != main and b2 > a and a2 < b:
                    beside["%s@%d" % (n2.replace("osd_", ""), streams[s2])] += min(b, b2) - max(a, a2)
            print("   +%8.1f us  %7.1f us  %-38s | %s" % (a - t0, b - a, n, ", ".join("%s %.0f" % kv for kv in beside.most_common(4))))
    by = collections.defaultdict(lambda: [0, 0.0])
    for a, b, n in ms:
        by[n][0] += 1
        by[n][1] += b - a
    print("  main stream by entry point (launches, ms):",
          sorted(((n, c, round(t / 1e3, 2)) for n, (c, t) in by.items()), key=lambda x: -x[2]))
    for st in sorted(per):
        if st == main:
            continue
        by = collections.defaultdict(lambda: [0, 0.0])
        for n, s_, a, b in step:
            if streams[s_] == st:
                by[n][0] += 1
                by[n][1] += b - a
        print("  stream %d:" % st, sorted(((n, c, round(t / 1e3, 2)) for n, (c, t) in by.items()), key=lambda x: -x[2])[:6])
    hist = collections.Counter(int(min(g[0], 99) // 5) * 5 for g in g2)
    print("  gap histogram (us bucket: count):", sorted(hist.items()))


if __name__ == "__main__":
    main()
