"""Worker for tests/test_dist_cpu.py::test_gradient_average_two_ranks (launched by torch.distributed.run, gloo)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd.dist_utils import average_flat_  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = torch.Generator().manual_seed(123)
base = torch.randn(100003, generator=g)           # odd length: ragged last bucket
flat = base * (rank + 1)
average_flat_(flat, None, 4)
expect = base * (sum(range(1, world + 1)) / world)
ok = torch.allclose(flat, expect, rtol=1e-6, atol=1e-6)
empty = average_flat_(torch.zeros(0), None, 4)
print("RANK %d OK=%s EMPTY=%d" % (rank, ok, empty.numel()), flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
