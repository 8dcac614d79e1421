"""GPU box: the dominant launch (FCOS tower 3x3 256 -> 256 over P3 + P4 at bs 8, conv_sp 256 x 256 tile) under the library named by
OSD_LIB_PATH (tagged diagnostic builds): best / median of ROUNDS brackets of 20 back-to-back launches on post-ReLU operands.
python tools/sp_lib_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops

g = torch.Generator(device="cuda").manual_seed(0)
xs = [torch.relu(torch.randn(8, h, w, 256, device="cuda", generator=g)).bfloat16() for h, w in ((100, 128), (50, 64))]
pc = ops.pack_conv(torch.randn(256, 256, 3, 3, device="cuda", generator=g) / 48, bias=torch.zeros(256, device="cuda"), dtype=torch.bfloat16)
algo = int(os.environ.get("SP_ALGO", str(ops.ALGO_SP)))
for _ in range(3):
    ops.conv2d_grouped(xs, pc, pad=1, algo=algo, _whole=True)
torch.cuda.synchronize()
ts = []
for r in range(int(os.environ.get("ROUNDS", "6"))):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.conv2d_grouped(xs, pc, pad=1, algo=algo, _whole=True)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 20 * 1e3)
ts.sort()
print("%s: best %.1f us, median %.1f us (%.0f TFLOP/s)" % (os.path.basename(os.environ.get("OSD_LIB_PATH", "default")), ts[0], ts[len(ts) // 2], 151.0 / ts[len(ts) // 2] * 1e3))
