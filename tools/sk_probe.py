"""GPU box: the tower weight-gradient launch (4 convs x 5 levels, bs 8) with ONE algo (env SK_ALGO), 6 launches — for rocprofv3 passes."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from oneshotdet_amd import ops
algo = int(os.environ["SK_ALGO"])
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
levels = [(8, 100, 128), (8, 50, 64), (8, 25, 32), (8, 13, 16), (8, 7, 8)]
dws = [torch.zeros(256, 3, 3, 256, device="cuda") for _ in range(4)]
items = [(rnd(n, h, w, 256), rnd(n, h, w, 256), dws[i], None, None) for i in range(4) for (n, h, w) in levels]
for _ in range(6):
    ops.conv2d_wgrad_multi(items, 3, 3, 1, 1, 256, algo=algo)
torch.cuda.synchronize()
