import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oneshotdet_amd import ops
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).to(dt)
levels = [(8, 100, 128), (8, 50, 64), (8, 25, 32), (8, 13, 16), (8, 7, 8)]
dws = [torch.zeros(256, 3, 3, 256, device="cuda") for _ in range(4)]
items = [(rnd(n, h, w, 256), rnd(n, h, w, 256), dws[i], None, None) for i in range(4) for (n, h, w) in levels]
def t(algo):
    f = lambda: ops.conv2d_wgrad_multi(items, 3, 3, 1, 1, 256, algo=algo)
    f(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(5): f()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / 5 * 1e3
out = []
for name, algo in (("v11 t7", 1 + 11 + 16 * 7), ("xr64 t4", 129 + 1 + 16 * 4), ("xr64 t0", 129 + 1), ("xr64 t7", 129 + 1 + 16 * 7), ("xr32x6 t4", 129 + 16 * 4), ("xr32x8 t4", 129 + 2 + 16 * 4)):
    out.append("%s %.0f" % (name, t(algo)))
print(os.environ.get("OSD_LIB_PATH", "default").split("hip")[-1], " | ".join(out))
