"""GPU box: what does the vendor GEMM (hipBLASLt through torch.matmul) reach on the GEMM shapes of the big convs?
A yardstick for the implicit-GEMM kernels, not a code path of the product."""
import torch


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(40e6))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


import sys
SHAPES = (("tower P3+P4 3x3", 128000, 256, 2304), ("tower P3 3x3", 102400, 256, 2304),
          ("same K, square-ish", 8192, 8192, 2304), ("large square", 8192, 8192, 8192),
          ("layer3 3x3", 25600, 256, 2304), ("fc6", 16000, 1024, 6272))
if len(sys.argv) > 1 and sys.argv[1] == "backbone":      # (M, N, K) of the backbone convs at bs=8, 800x1024
    SHAPES = (("layer2 conv3 1x1", 102400, 512, 128), ("layer2 conv1 1x1", 102400, 128, 512), ("layer2 3x3", 102400, 128, 1152),
              ("layer3 conv3 1x1", 25600, 1024, 256), ("layer3 conv1 1x1", 25600, 256, 1024), ("layer3 3x3", 25600, 256, 2304),
              ("layer4 3x3", 6400, 512, 4608), ("layer4 conv3 1x1", 6400, 2048, 512), ("layer4 conv1 1x1", 6400, 512, 2048),
              ("layer1 conv3 1x1", 409600, 256, 64), ("layer1 3x3", 409600, 64, 576), ("layer1 conv1 1x1", 409600, 64, 256),
              ("fpn inner2", 102400, 256, 512), ("layer2 ds", 102400, 512, 256))
for name, m, n, k in SHAPES:
    a = torch.relu(torch.randn(m, k, device="cuda")).bfloat16()
    b = torch.randn(n, k, device="cuda").bfloat16() / k ** 0.5
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    t = bench(lambda: torch.matmul(a, b.t(), out=out))
    print("%-22s M=%d N=%d K=%d: %.1f us, %.0f TFLOP/s" % (name, m, n, k, t, 2.0 * m * n * k / t / 1e6), flush=True)
