"""GPU box: the prediction convs' weight gradient over the five FPN levels at bs = 8 (cout 2 and 4): read-once kernel, or
(OSD_NO_PRED_WGRAD=1) the 128-channel MFMA tile.  python tools/pred_wgrad_bench.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oneshotdet_amd import ops
levels = [(8, 100, 128), (8, 50, 64), (8, 25, 32), (8, 13, 16), (8, 7, 8)]
g = torch.Generator(device="cuda").manual_seed(0)
for cout in (2, 4):
    pairs = [((torch.randn(n, h, w, 256, device="cuda", generator=g)).bfloat16(), (torch.randn(n, h, w, 64, device="cuda", generator=g) * 0.1).bfloat16())
             for n, h, w in levels]
    dw, db = torch.zeros(cout, 3, 3, 256, device="cuda"), torch.zeros(cout, device="cuda")
    fn = lambda: ops.conv2d_wgrad_grouped(pairs, dw, 3, 3, 1, 1, cout, db=db)      # noqa: E731
    fn(); torch.cuda.synchronize()
    if not os.environ.get("OSD_NO_PRED_WGRAD"):            # the two launches separately (HIP events around direct C-ABI calls)
        import ctypes as C
        from oneshotdet_amd import _lib
        k = len(pairs)
        d = ops._conv_desc(pairs[0][0].shape, ops.OSD_BF16, cout, 3, 3, 1, 1, 64)
        xs = (C.c_void_p * k)(*[x.data_ptr() for x, _ in pairs]); dys = (C.c_void_p * k)(*[y.data_ptr() for _, y in pairs])
        ns = (C.c_int32 * k)(*[x.shape[0] for x, _ in pairs]); hs = (C.c_int32 * k)(*[x.shape[1] for x, _ in pairs]); ws = (C.c_int32 * k)(*[x.shape[2] for x, _ in pairs])
        need = int(_lib.load().osd_conv2d_wgrad_pred_workspace_bytes(k, ns, hs, ws, 256))
        wsp = torch.empty((need // 4 + 1,), device="cuda")
        call = lambda: _lib.call("osd_conv2d_wgrad_pred", C.byref(d), k, xs, dys, ns, hs, ws, ops._ptr(dw), ops._ptr(db), ops._ptr(wsp), ops._stream())   # noqa: E731
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record(); torch.cuda.synchronize()
        print("   C-ABI call with a preallocated workspace: %.1f us" % (e0.elapsed_time(e1) * 100))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record(); torch.cuda.synchronize()
    mb = sum(x.numel() * 2 for x, _ in pairs) / 1e6
    us = a.elapsed_time(b) * 100
    print("cout %d: %.1f us per launch (x operand %.0f MB -> %.2f TB/s)  [%s]" % (cout, us, mb, mb / us,
          "MFMA tile" if os.environ.get("OSD_NO_PRED_WGRAD") else "read-once kernel"))
