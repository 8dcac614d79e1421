"""GPU box, diagnostic build (OSD_LIB_PATH=oneshotdet_amd/lib/liboneshotdet_hip_stamps.so): where the cycles of the
ping-pong conv kernel go.  Prints, for wave 0 (group 0) and wave 4 (group 1) of workgroup 0, the cycles per K tile spent
in: loads + DMA issue, waiting at the first barrier, issuing the 16 MFMAs, waiting at the second barrier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from oneshotdet_amd import ops, _lib

lib = _lib.load()
fn = lib.osd_debug_p8_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
for (n, h, w, cin, cout, k, pad) in [(8, 100, 128, 256, 256, 3, 1), (8, 50, 64, 1024, 256, 1, 0)]:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    wt = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    pc = ops.pack_conv(wt, bias=torch.randn(cout, device="cuda"), dtype=torch.bfloat16)
    for _ in range(5):
        y = ops.conv2d(x, pc, pad=pad, algo=6)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    assert fn(out) == 0
    kt = cin * k * k // 64
    for g in range(2):
        v = [out[g * 4 + i] / kt for i in range(4)]
        print("cin%d k%d group %d: per K tile (4 phases): loads+issue %.0f  barrier-1 wait %.0f  mfma issue %.0f  barrier-2 wait %.0f  total %.0f cycles (100 MHz ticks? see note)"
              % (cin, k, g, v[0], v[1], v[2], v[3], sum(v)))
