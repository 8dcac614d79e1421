"""GPU box: is the big 3x3 conv bound by the operand feed (L2 / Infinity Cache -> LDS) or by the MFMA/LDS pipeline?
Runs the same launch twice: on the real NHWC input, and with the input's row / image strides set to 0 so that every
pixel row aliases row 0 (the pixel operand then lives in L2: same instruction stream, same MFMA work, no feed misses)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from oneshotdet_amd import ops, _lib

dt = torch.bfloat16
for (n, h, w, cin, cout, k, pad) in [(8, 100, 128, 256, 256, 3, 1), (8, 50, 64, 256, 256, 3, 1), (8, 50, 64, 256, 1024, 1, 0),
                                     (8, 100, 128, 128, 512, 1, 0)]:
    x = torch.randn(n, h, w, cin, device="cuda").to(dt)
    wt = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=dt)
    y = torch.empty(n, h, w, cout, device="cuda", dtype=dt)
    fl = 2.0 * n * h * w * cout * cin * k * k
    for algo in (1 + 8 + 4, 1 + 16 + 4, 1 + 8 + 0, 1 + 0):
        out = []
        for alias in (False, True):
            d = ops.ConvDesc()
            d.dtype = ops._dt(x)
            d.n, d.h, d.w, d.cin, d.r, d.s = n, h, w, cin, k, k
            d.in_stride_n, d.in_stride_h, d.in_stride_w = (0, 0, cin) if alias else (h * w * cin, w * cin, cin)
            d.stride_h = d.stride_w = 1
            d.pad_h = d.pad_w = pad
            d.ho, d.wo, d.cout, d.w_rows, d.out_stride = h, w, cout, pc.w_rows, cout
            d.algo = algo
            args = (ops._ptr(x), ops._ptr(pc.w), ops._ptr(pc.bias), None, None, None, None, ops._ptr(y), ops._stream())
            try:
                _lib.call("osd_conv2d_fwd", C.byref(d), *args)
            except _lib.OsdError:
                continue
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                _lib.call("osd_conv2d_fwd", C.byref(d), *args)
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) / 10
            out.append("%s %.1f us %.0f TF" % ("aliased" if alias else "real", t * 1e3, fl / t / 1e9))
        print("M=%d N=%d K=%d algo %d: %s" % (n * h * w, cout, cin * k * k, algo, "   ".join(out)), flush=True)
