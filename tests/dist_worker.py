"""Worker for tests/test_dist_cpu.py::test_gradient_average_two_ranks (launched by torch.distributed.run, gloo)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import spec  # noqa: E402
from oneshotdet_amd.dist_utils import GradExchange, average_flat_, bucket_ranges  # noqa: E402

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

# the overlapped exchange of TrainEngine: buckets of the real parameter plan announced in backward order (head first,
# then per backbone layer4+fpn, layer3, layer2), one bucket deliberately never announced (finish() must pick it up), and a
# second step to check that finish() re-arms the buckets
plan = []
for name, shape in spec.hot_path_shapes().items():
    if spec.is_frozen(name) or "running_" in name or ".bn" in name or "downsample.1" in name:
        continue
    plan.append((name, (min(int(shape[0]), 8),) + tuple(min(int(d), 3) for d in shape[1:])))   # shrunk: structure only
order = [n for n, _ in plan]
plan.sort(key=lambda kv: (0 if kv[0].startswith("backbone.") else 1 if kv[0].startswith("supp_backbone.") else 2,
                          0 if ".body." in kv[0] else 1, order.index(kv[0])))
import math
total = (sum(int(math.prod(s)) for _, s in plan) + 63) // 64 * 64
ranges = bucket_ranges(plan, total)
cover = sorted((lo, hi) for _, lo, hi in ranges)
ok2 = cover[0][0] == 0 and cover[-1][1] == total and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
names = [n for n, _, _ in ranges]
ok2 = ok2 and names == ["backbone.layer2", "backbone.layer3", "backbone.layer4+fpn", "supp_backbone.layer2",
                        "supp_backbone.layer3", "supp_backbone.layer4+fpn", "head"]
base2 = torch.randn(total, generator=g)
for step in range(2):
    flat2 = base2 * (rank + 1 + step)
    ex = GradExchange(flat2, ranges) if step == 0 else ex
    if step == 1:
        ex.flat = flat2
    for n in ("head", "backbone.layer4+fpn", "supp_backbone.layer4+fpn", "backbone.layer3", "supp_backbone.layer3",
              "backbone.layer2"):
        ex.ready(n)
        ex.ready(n)                       # announcing twice must not reduce twice
    ex.finish()                           # supp_backbone.layer2 was never announced
    expect2 = base2 * (sum(r + 1 + step for r in range(world)) / world)
    ok2 = ok2 and torch.allclose(flat2, expect2, rtol=1e-6, atol=1e-6)
# the same with the second stage's parameters (TrainEngine(second_stage=True)): one more bucket, `box_head`, announced
# right after the first-stage head; every rank ends with the average
plan3 = []
for name, shape in spec.full_model_shapes().items():
    if spec.is_frozen(name) or "running_" in name or ".bn" in name or "downsample.1" in name:
        continue
    plan3.append((name, (min(int(shape[0]), 8),) + tuple(min(int(d), 3) for d in shape[1:])))
order3 = [n for n, _ in plan3]
plan3.sort(key=lambda kv: (0 if kv[0].startswith("backbone.") else 1 if kv[0].startswith("supp_backbone.") else
                           2 if kv[0].startswith("rpn.") else 3, 0 if ".body." in kv[0] else 1, order3.index(kv[0])))
total3 = (sum(int(math.prod(s)) for _, s in plan3) + 63) // 64 * 64
ranges3 = bucket_ranges(plan3, total3)
names3 = [n for n, _, _ in ranges3]
ok2 = ok2 and names3[-2:] == ["head", "box_head"] and ranges3[-1][2] == total3 and any(n.startswith("roi_heads.") for n, _ in plan3)
flat3 = torch.randn(total3, generator=g) * 0 + (rank + 1.0)
ex3 = GradExchange(flat3, ranges3)
for n in ("head", "box_head", "backbone.layer4+fpn", "supp_backbone.layer4+fpn"):
    ex3.ready(n)
ex3.finish()
ok2 = ok2 and torch.allclose(flat3, torch.full_like(flat3, sum(r + 1.0 for r in range(world)) / world))
print("RANK %d EXCHANGE=%s" % (rank, ok2), flush=True)

# kernel choices: every rank tunes by its own timing, so before the timed region rank 0's tuner caches replace everyone's
# (dist_utils.broadcast_tuner_choices); here the ranks start with DIFFERENT choices for the same shapes and extra private keys
import types
from oneshotdet_amd import dist_utils
fake = types.SimpleNamespace(ALGO_CACHE={(1, 8, 100, 128, 256, 256, 3, 3, 1, 1, 0, 0, 0, False): 15 if rank == 0 else 13,
                                         ("grouped", 1, ((8, 100, 128), (8, 50, 64)), 256): 15 + rank},
                             SPLIT_CACHE={("split", 1, ((8, 100, 128, 256),), 256, 3, 1, 1, 0, 0, False): rank},
                             WGRAD_ALGO_CACHE={("w", 1, 256, 256, 3): 4 + 16 * rank})
if rank == 1:
    fake.ALGO_CACHE[("only rank 1 met this shape",)] = 7
ok3 = not dist_utils.tuner_choices_agree(fake)
dist_utils.broadcast_tuner_choices(fake, src=0)
ok3 = ok3 and dist_utils.tuner_choices_agree(fake)
ok3 = ok3 and fake.ALGO_CACHE[(1, 8, 100, 128, 256, 256, 3, 3, 1, 1, 0, 0, 0, False)] == 15 and len(fake.ALGO_CACHE) == 2
ok3 = ok3 and list(fake.SPLIT_CACHE.values()) == [0] and list(fake.WGRAD_ALGO_CACHE.values()) == [4]
import tempfile
path = os.path.join(tempfile.gettempdir(), "osd_tuner_%d_%d.json" % (os.getpid(), rank))
dist_utils.save_tuner_choices(fake, path)
fake2 = types.SimpleNamespace(ALGO_CACHE={}, SPLIT_CACHE={}, WGRAD_ALGO_CACHE={})
ok3 = ok3 and dist_utils.load_tuner_choices(fake2, path) is True
ok3 = ok3 and fake2.ALGO_CACHE == fake.ALGO_CACHE and fake2.WGRAD_ALGO_CACHE == fake.WGRAD_ALGO_CACHE
# a file of another library generation (algorithm ids were reused for other kernels: ADVICE r5) is ignored, not replayed
import json, warnings
with open(path) as f:
    stale = json.load(f)
stale["abi"] = 3
with open(path, "w") as f:
    json.dump(stale, f)
fake3 = types.SimpleNamespace(ALGO_CACHE={}, SPLIT_CACHE={}, WGRAD_ALGO_CACHE={})
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ok3 = ok3 and dist_utils.load_tuner_choices(fake3, path) is False and not fake3.ALGO_CACHE and not fake3.WGRAD_ALGO_CACHE
os.remove(path)
print("RANK %d TUNER=%s" % (rank, ok3), flush=True)
dist.destroy_process_group()
sys.exit(0 if (ok and ok2 and ok3) else 1)
