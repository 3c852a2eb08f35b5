"""is the forward tap fold worth building?  70 -> 70 15x15 (DRCNN:L, batch 64) as 70 -> 64 plus the 6 remaining couts as
12 rows of a 17x15 stride-(2,1) convolution, both through the existing kernels; correctness of the strided piece vs torch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from multipitch_architectures_amd import ops
dev = torch.device("cuda:0")

def timed(fn, n=3):
    best = 1e9
    for _ in range(n):
        keys = []
        ops.set_kernel_probe(lambda k, kind: keys.append(kind) or True)
        fn()
        ms = ops.probe_results_ms(); ops.set_kernel_probe(None)
        best = min(best, sum(ms))
    return best

# correctness on a small case (odd and even heights)
for H in (9, 12, 75):
    x = torch.randn(2, 5, H, 20); w = torch.randn(6, 5, 15, 15) * 0.05; b = torch.randn(6)
    ref = F.conv2d(x, w, b, padding=7)
    wf = torch.zeros(12, 5, 17, 15)
    for s in (0, 1):
        wf[s * 6:(s + 1) * 6, :, 1 + s:16 + s] = w
    yf = ops.conv2d(x.to(dev), wf.to(dev), b.repeat(2).to(dev), (2, 1), (8, 7), ops.ACT_NONE, 0.0).cpu()
    y = torch.empty_like(ref)
    y[:, :, 0::2] = yf[:, :6, :(H + 1) // 2]
    y[:, :, 1::2] = yf[:, 6:, :H // 2]
    print("H", H, "fold err", float((y - ref).abs().max()), "yf", tuple(yf.shape))

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 70, 75, 216, device=dev)
for name, (co, kh, sh, ph) in {"70->70 15x15": (70, 15, 1, 7), "70->64 15x15": (64, 15, 1, 7), "70->12 17x15 s2": (12, 17, 2, 8),
                               "70->16 17x15 s2": (16, 17, 2, 8), "70->12 16x15 s2": (12, 16, 2, 7), "70->6 15x15": (6, 15, 1, 7)}.items():
    w = torch.randn(co, 70, kh, 15, device=dev) * 0.02
    b = torch.zeros(co, device=dev)
    try:
        t = timed(lambda: ops.conv2d(x, w, b, (sh, 1), (ph, 7), ops.ACT_LRELU, 0.3))
        print(f"{name:18s} {t:7.3f} ms", flush=True)
    except RuntimeError as e:
        print(name, "unsupported", str(e)[:80])
