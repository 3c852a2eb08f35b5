"""Is a kernel launched on a stream ordered before a HIP graph launched right after it on the same stream?"""
import torch
dev = torch.device("cuda:0")
for N in (1, 1 << 10, 1 << 26):
    for dt in (torch.float32, torch.float64):
        a = torch.zeros(N, device=dev, dtype=dt)
        out = torch.zeros(N, device=dev, dtype=dt)
        h = torch.zeros(4, device=dev, dtype=torch.float64)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            out.copy_(a * 2 + h[0].to(dt))
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out.copy_(a * 2 + h[0].to(dt))
        bad = 0
        for i in range(200):
            a.fill_(float(i))
            h[0:1].fill_(float(i) * 0.5)
            g.replay()
            if i % 7 == 0:
                v = float(out[0]); w = float(out[-1])
                if v != 2.5 * i or w != 2.5 * i:
                    bad += 1
        torch.cuda.synchronize()
        print("N", N, dt, "bad", bad, float(out[0]), 2.5 * 199)
