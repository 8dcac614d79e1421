import torch, time
x = torch.zeros(1 << 20, device="cuda")
def chain(n):
    for _ in range(n):
        x.add_(1.0)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda._sleep(int(30e6))
    e[0].record()
    for _ in range(reps): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps * 1e3
N = 200
print("eager: %.2f us per dependent 4 MB add kernel" % (t(lambda: chain(N)) / N))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain(3)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    chain(N)
print("graph: %.2f us per kernel" % (t(lambda: g.replay()) / N))
y = torch.zeros(256, device="cuda")
def chain2(n):
    for _ in range(n):
        y.add_(1.0)
print("eager tiny: %.2f us" % (t(lambda: chain2(N)) / N))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    chain2(N)
print("graph tiny: %.2f us" % (t(lambda: g2.replay()) / N))
