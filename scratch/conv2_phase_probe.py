"""would the head conv2 (128->Cout 3x3 stride (1,3) @75x216) run faster as a stride-1 (3,1) convolution over the
phase-major layout (384 channels @75x72)?  times fwd / dgrad / wgrad of both forms with the existing kernels"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
def run(tag, xs, ws, stride, pad):
    x = torch.randn(*xs, device=dev).requires_grad_(True)
    w = (torch.randn(*ws, device=dev) * 0.02).requires_grad_(True)
    b = torch.zeros(ws[0], device=dev, requires_grad=True)
    keys = []
    ops.set_kernel_probe(lambda k, kind: (keys.append(kind) or True))
    for _ in range(3):
        y = ops.conv2d(x, w, b, stride, pad)
        y.backward(torch.ones_like(y))
    ms = ops.probe_results_ms()
    ops.set_kernel_probe(None)
    fl = 2.0 * y.numel() * ws[1] * ws[2] * ws[3]
    best = {}
    for k, t in zip(keys, ms):
        best[k] = min(best.get(k, 1e9), t)
    print(tag, " ".join(f"{k} {t:.3f} ms {fl / t / 1e9:.1f} TF/s" for k, t in best.items()), flush=True)
for Cout in (80, 200):
    run(f"strided  128->{Cout}", (B, 128, 75, 216), (Cout, 128, 3, 3), (1, 3), (1, 0))
    run(f"phase    384->{Cout}", (B, 384, 75, 72), (Cout, 384, 3, 1), (1, 1), (1, 0))
