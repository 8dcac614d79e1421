"""GPU box: tower conv over the 5 FPN levels — one grouped launch vs big levels / small levels as two launches
(256x256 tiles on 256 CUs: 534 tiles = 3 rounds with the third almost empty; 500 tiles = 2 rounds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops

sizes = [(8, 100, 128), (8, 50, 64), (8, 25, 32), (8, 13, 16), (8, 7, 8)]
xs = [torch.randn(n, h, w, 256, device="cuda").bfloat16() for n, h, w in sizes]
wt = torch.randn(256, 256, 3, 3, device="cuda") / 48
pc = ops.pack_conv(wt, bias=torch.zeros(256, device="cuda"), dtype=torch.bfloat16)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2e8))
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


with ops.tuning():
    ops.conv2d_grouped(xs, pc, pad=1)
    for cut in (1, 2, 3):
        ops.conv2d_grouped(xs[:cut], pc, pad=1)
        ops.conv2d_grouped(xs[cut:], pc, pad=1)
print("one grouped launch (5 levels): %.1f us" % timeit(lambda: ops.conv2d_grouped(xs, pc, pad=1)))
for cut in (1, 2, 3):
    t = timeit(lambda: (ops.conv2d_grouped(xs[:cut], pc, pad=1), ops.conv2d_grouped(xs[cut:], pc, pad=1)))
    ta, tb = timeit(lambda: ops.conv2d_grouped(xs[:cut], pc, pad=1)), timeit(lambda: ops.conv2d_grouped(xs[cut:], pc, pad=1))
    print("levels [:%d] + [%d:] as two launches: %.1f us  (%.1f + %.1f)" % (cut, cut, t, ta, tb))
