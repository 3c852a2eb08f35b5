"""time the MLP's two short-K products through ops.linear: python scratch/panel_time.py [rows]
(forward 128 -> 8192 with ReLU: B k-contiguous; input gradient of 8192 -> 128: B n-contiguous)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import ops
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 13312
dev = torch.device("cuda:0")
x = torch.randn(rows, 128, device=dev); w = torch.randn(8192, 128, device=dev) * 0.05; b = torch.randn(8192, device=dev)
gy = torch.randn(rows, 128, device=dev); w2 = torch.randn(128, 8192, device=dev) * 0.05
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fl = 2.0 * rows * 8192 * 128
y = torch.empty(rows, 8192, device=dev); dx = torch.empty(rows, 8192, device=dev)
P = ops._p
a = t(lambda: ops._gemm(P(x), 128, 1, P(w), 1, 128, P(b), P(y), 8192, rows, 8192, 128, 0, 1))
c = t(lambda: ops._gemm(P(gy), 128, 1, P(w2), 8192, 1, None, P(dx), 8192, rows, 8192, 128))
print(f"{os.environ.get('MPA_GEMM_PANEL_WGS', '-'):>6s} {'old' if os.environ.get('MPA_GEMM_NO_PANEL') else 'panel'}  fwd {a:.3f} ms {fl/a/1e9:6.1f} TF   dgrad {c:.3f} ms {fl/c/1e9:6.1f} TF", flush=True)
