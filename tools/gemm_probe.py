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


for name, m, n, k in (("tower P3+P4 3x3", 128000, 256, 2304), ("tower P3 3x3", 102400, 256, 2304),
                      ("same K, square-ish", 8192, 8192, 2304), ("large square", 8192, 8192, 8192),
                      ("layer3 3x3", 25600, 256, 2304), ("fc6", 16000, 1024, 6272)):
    a = torch.relu(torch.randn(m, k, device="cuda")).bfloat16()
    b = torch.randn(n, k, device="cuda").bfloat16() / k ** 0.5
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    t = bench(lambda: torch.matmul(a, b.t(), out=out))
    print("%-22s M=%d N=%d K=%d: %.1f us, %.0f TFLOP/s" % (name, m, n, k, t, 2.0 * m * n * k / t / 1e6), flush=True)
