"""GPU box: the PCIe-inclusive training step (the reference's loop moves every batch host -> device, engine/trainer.py:75-78).
Per step: 8 uint8 RGB targets (600 x 768, what a COCO image is before Resize) and 8 uint8 queries (127 x 127) leave PINNED host
memory by asynchronous copies on a copy stream, the fused transform chain (osd_image_transform_batch: PIL-exact resize to
800 x 1024, BGR255 - mean, pad, written straight into the stem conv's NHWC4 bf16 input) runs behind them on that stream, and the
training step waits for that batch's event; batch t + 1 is produced while step t runs (two slots).  Compared, in the same
process, with the resident-input step bench.py times.   python tools/pcie_step.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops, spec, synth, train  # noqa: E402
from oneshotdet_amd import transforms as T  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    B = 8
    eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
    rng = np.random.RandomState(0)
    host_t = [[torch.from_numpy(rng.randint(0, 256, (600, 768, 3), dtype=np.uint8)).pin_memory() for _ in range(B)] for _ in range(2)]
    host_q = [[torch.from_numpy(rng.randint(0, 256, (127, 127, 3), dtype=np.uint8)).pin_memory() for _ in range(B)] for _ in range(2)]
    rs_t, rs_q = T.Resize(800, 1200), T.Resize(127, 127)
    gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
    gtb = np.zeros((B, 6, 4), np.float32)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = g
    gt_boxes = torch.from_numpy(gtb).cuda()
    gt_count = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
    copy = torch.cuda.Stream()
    slots = [None, None]
    events = [torch.cuda.Event(), torch.cuda.Event()]
    done = [torch.cuda.Event(), torch.cuda.Event()]          # the step that read a slot has been enqueued and finished reading

    def produce(k):
        with torch.cuda.stream(copy):
            copy.wait_event(done[k])
            imgs = [rs_t(T.DeviceImage(h.to("cuda", non_blocking=True)), None)[0] for h in host_t[k]]
            qs = [rs_q(T.DeviceImage(h.to("cuda", non_blocking=True)), None)[0] for h in host_q[k]]
            slots[k] = (T.collate(imgs, spec.SIZE_DIVISIBILITY, stem_dtype=torch.bfloat16),
                        T.collate(qs, 0, stem_dtype=torch.bfloat16))
            events[k].record(copy)
    for k in (0, 1):
        done[k].record()
    produce(0)
    torch.cuda.synchronize()
    images, queries = slots[0]
    assert tuple(images.shape) == (B, 3, 800, 1024) and tuple(queries.shape)[-2:] == (127, 127), (images.shape, queries.shape)
    with ops.tuning():
        eng.forward_backward(images, queries, gt_boxes, gt_count)
    torch.cuda.synchronize()
    eng.defer_join = True

    def run(n, from_host):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            k = i & 1
            if from_host:
                produce(k ^ 1)                       # next batch travels while this step runs
                torch.cuda.current_stream().wait_event(events[k])
            im, qu = slots[k] if from_host else slots[0]
            eng.train_step(im, qu, gt_boxes, gt_count)
            if from_host:
                done[k].record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    produce(1)
    run(8, False)
    res = {}
    for rep in range(3):
        for mode in (False, True):
            res.setdefault(mode, []).append(run(steps, mode))
    h2d = sum(h.numel() for h in host_t[0]) + sum(h.numel() for h in host_q[0])
    r, f = sorted(res[False])[1], sorted(res[True])[1]
    print("resident inputs      : %.3f ms / step (%.1f images/s)   [median of 3 x %d steps]" % (r, B / r * 1e3, steps))
    print("from pinned host uint8: %.3f ms / step (%.1f images/s)   %.1f MB H2D + resize/normalise/pad per step, overlapped on a copy stream"
          % (f, B / f * 1e3, h2d / 1e6))
    print("PCIe-inclusive cost: %+.2f %%" % (100.0 * (f - r) / r))


if __name__ == "__main__":
    main()
