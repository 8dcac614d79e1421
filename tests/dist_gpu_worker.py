"""Worker for tests/test_gpu_train.py::test_two_ranks_on_one_gpu_average_gradients (torch.distributed.run, 2 ranks, both
on cuda:0, gloo backend with device tensors) and tests/test_gpu_multi.py (OSD_DIST_BACKEND=nccl: one rank per GPU over RCCL, on
boxes with >= 2 GPUs): the REAL training engine with the overlapped gradient exchange against the average of the two ranks'
gradients computed without any exchange."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_utils as gu  # noqa: E402
from oneshotdet_amd import spec, synth, train  # noqa: E402

BACKEND = os.environ.get("OSD_DIST_BACKEND", "gloo")
if BACKEND == "nccl":      # RCCL: one GPU per rank, the engine's streams first (stream-to-queue assignment), then the group
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
else:
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
rank, world = dist.get_rank(), dist.get_world_size()
name = "small"
B, H, W, S, qh, qw = gu.CASES[name]


def inputs(r):
    img, q = synth.make_images("t%d" % r, B, H, W, seed=10 + r), synth.make_images("q%d" % r, B * S, qh, qw, seed=20 + r)
    gts = synth.make_gt_boxes(B, H, W, seed=30 + r, max_boxes=3)
    G = max(len(g) for g in gts)
    gtb = torch.zeros(B, G, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
    return torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda(), gtb.cuda(), cnt.cuda()


sd = synth.make_state_dict(spec.hot_path_shapes())
eng = train.TrainEngine(sd, dtype=torch.float32, lr=0.01)
assert eng.exchange.active and eng.exchange.world == world
assert eng.exchange.avg == (BACKEND == "nccl")      # RCCL averages natively; gloo: sum + scale
# reference: both ranks' gradients computed locally with the exchange switched off
eng._overlap = False
eng.exchange.active = False
ref = torch.zeros_like(eng.flat_g)
for r in range(world):
    eng.forward_backward(*inputs(r))
    ref += eng.flat_g
ref /= world
# the overlapped exchange on this rank's own batch
eng._overlap = True
eng.exchange.active = True
w0 = eng.flat_w.clone()
eng.forward_backward(*inputs(rank))
ok_pending = eng.exchange.pending == set()
eng.reduce_gradients()
torch.cuda.synchronize()
err = float((eng.flat_g - ref).norm() / ref.norm())
# and a whole train_step (exchange + SGD + repack behind backward) must apply the SAME update on both ranks
eng.flat_w.copy_(w0)
eng.repack()
eng.train_step(*inputs(rank))
torch.cuda.synchronize()
mine = (eng.flat_w - w0) if BACKEND == "nccl" else (eng.flat_w - w0).cpu()
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
same = float((both[0] - both[1]).abs().max())
moved = float(mine.abs().max())
ok = ok_pending and err < 1e-4 and same <= 1e-7 * max(moved, 1e-30) + 1e-9 and moved > 0
print("RANK %d BACKEND=%s WORLD=%d" % (rank, dist.get_backend(), dist.get_world_size()), flush=True)
print("RANK %d GPU_EXCHANGE=%s pending_empty=%s rel_err=%.2e update_max_diff=%.2e update_max=%.2e" % (
    rank, ok, ok_pending, err, same, moved), flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
