"""CPU: could Winograd F(2x2, 3x3) replace the direct bf16 MFMA conv of the FCOS towers (fcos.py:26-37,
3x3 256 -> 256 on post-GroupNorm-ReLU activations) UNDER THE PER-LAUNCH PARITY BAR of tests/test_gpu_launch_replay.py?

The bar ("The measured dtype"): every bf16 output element within ONE bf16 ulp of the fp32 restatement rounded once
(+ 1e-5 x absmax), at most 2 % of the elements on the neighbouring value.  The direct kernel meets it because bf16 x bf16 products
are exact in fp32 and only the fp32 summation order differs.  Winograd must ROUND its transformed operands to the MFMA's input type:
V = B^T d B (sums of four activations) and U = G g G^T (sums of up to nine weights / 4) are not bf16 values.  This script emulates
the arithmetic exactly (transforms in fp32, operands rounded to bf16 — or fp16 as the best case an MFMA offers at the same rate —,
products and sums in fp32, output transform in fp32, one final rounding) and measures it against that bar.

python tools/probe/winograd_numerics.py      (CPU only, ~20 s)"""
import torch

torch.manual_seed(0)
C, K, H, W = 256, 256, 32, 32
x = torch.relu(torch.randn(1, C, H, W) * 1.0 + 0.2).bfloat16().float()          # post-ReLU activations
w = (torch.randn(K, C, 3, 3) / (C * 9) ** 0.5).bfloat16().float()
ref64 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
ref = ref64.float().bfloat16().float()                                            # the restatement rounded once

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd(x, w, op_dtype):
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    # 4x4 input tiles with stride 2: [1, C, th, tw, 4, 4]
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum("ij,nchwjk,lk->nchwil", Bt, t, Bt)                            # fp32 transform
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)
    V = V.to(op_dtype).float()                                                     # MFMA operands
    U = U.to(op_dtype).float()
    M = torch.einsum("kcij,nchwij->nkhwij", U, V)                                  # fp32 products and sums over c
    Y = torch.einsum("ij,nkhwjl,ml->nkhwim", At, M, At)                            # fp32 output transform: [n, k, th, tw, 2, 2]
    n, k, th, tw = Y.shape[:4]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, th * 2, tw * 2)


def report(name, y):
    y = y.bfloat16().float()
    ulp = torch.maximum(ref.abs(), y.abs()) * 2.0 ** -7 + 1e-5 * float(ref.abs().max())
    d = (y - ref).abs()
    off = d > 0
    beyond = d > ulp
    rel = float((y.double() - ref64).norm() / ref64.norm())
    print("%-34s differ from the once-rounded restatement: %5.1f %% of elements (bar: 2 %%), beyond one ulp: %5.2f %% (bar: 0), "
          "relative L2 to the fp64 result after the final rounding %.1e" % (name, 100 * float(off.float().mean()), 100 * float(beyond.float().mean()), rel))


direct = torch.nn.functional.conv2d(x, w, padding=1)                               # fp32 accumulation, another summation order
report("direct conv, fp32 accumulate", direct)
report("Winograd F(2x2,3x3), bf16 operands", winograd(x, w, torch.bfloat16))
report("Winograd F(2x2,3x3), fp16 operands", winograd(x, w, torch.float16))
report("Winograd F(2x2,3x3), fp32 operands", winograd(x, w, torch.float32))
# the data gradient sees dY instead of post-ReLU activations: zero-mean values with a wide dynamic range
g = (torch.randn(1, C, H, W) * torch.exp(torch.randn(1, C, 1, 1) * 2.0) * 1e-3).bfloat16().float()
ref64 = torch.nn.functional.conv2d(g.double(), w.double(), padding=1)
ref = ref64.float().bfloat16().float()
print("data gradient (zero-mean dY, per-channel scales over e^+-2):")
report("direct conv, fp32 accumulate", torch.nn.functional.conv2d(g, w, padding=1))
report("Winograd F(2x2,3x3), bf16 operands", winograd(g, w, torch.bfloat16))
report("Winograd F(2x2,3x3), fp16 operands", winograd(g, w, torch.float16))
