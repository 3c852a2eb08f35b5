"""per-node cost of a replayed HIP graph of dependent tiny kernels (what every sub-10-us launch of the step costs at least)"""
import time, torch
dev = torch.device("cuda:0")
t = torch.zeros(64, device=dev)
big = torch.zeros(1 << 22, device=dev)
for N, what in ((1000, "tiny"), (1000, "16MB")):
    g = torch.cuda.CUDAGraph()
    buf = t if what == "tiny" else big
    buf.add_(1); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(N):
            buf.add_(1)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 / N * 1e6
    print(f"{what}: {dt:.2f} us per node in a {N}-node graph")
    # eager launches for comparison
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        buf.add_(1)
    torch.cuda.synchronize()
    print(f"{what}: {(time.perf_counter() - t0) / N * 1e6:.2f} us per eager launch")
