"""[D2D copy, fill kernel, graph launch] on one stream, no host syncs: does the graph see both writes?"""
import torch
dev = torch.device("cuda:0")
N = 1 << 22
srcs = [torch.full((N,), float(i), device=dev) for i in range(8)]
a = torch.zeros(N, device=dev)
h = torch.zeros(4, device=dev, dtype=torch.float64)
acc = torch.zeros(3, device=dev, dtype=torch.float64)
big = torch.zeros(1 << 24, device=dev)

def body():
    big.mul_(1.0001)                                  # some work first, like a forward pass
    d = (a[0].double() - h[0]).abs() + (a[-1].double() - h[0]).abs()
    acc[0] += d
    acc[1] += 1

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
acc.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for mode in ("copy+fill", "fill only", "copy only"):
    acc.zero_(); torch.cuda.synchronize()
    for i in range(3000):
        k = i % 8
        if mode != "fill only":
            a.copy_(srcs[k])
        else:
            a.fill_(float(k))
        if mode != "copy only":
            h[0:1].fill_(float(k))
        else:
            h[0:1].copy_(srcs[k][0:1].double())
        g.replay()
    torch.cuda.synchronize()
    print(mode, acc.tolist())
