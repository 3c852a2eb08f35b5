"""per-pass timing of single 15x15 / 9x9 layers at a given batch (set MPA_FWD_FORCE="NB,PB" outside to pin the forward /
backward-data wave tile: the planner reads it once per process)
usage: python scratch/fwd_force.py B"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
LAYERS = [(16, 75, 216, 128, 15), (32, 75, 216, 16, 15), (16, 75, 216, 16, 15), (6, 75, 216, 16, 15), (32, 37, 108, 32, 15),
          (16, 37, 108, 32, 15), (64, 37, 108, 32, 9), (32, 37, 108, 16, 9), (64, 18, 54, 64, 9), (32, 18, 54, 64, 9)]
if len(sys.argv) > 2:
    LAYERS = [LAYERS[int(i)] for i in sys.argv[2].split(",")]
out = []
for Cin, H, W, Cout, k in LAYERS:
    x = torch.randn(B, Cin, H, W, device=dev, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.02).requires_grad_(True)
    b = torch.zeros(Cout, device=dev, requires_grad=True)
    gy = torch.randn(B, Cout, H, W, device=dev)
    best = {}
    for it in range(4):
        keys = []
        ops.set_kernel_probe(lambda key, kind: keys.append(kind) or True)
        try:
            y = ops.conv2d(x, w, b, (1, 1), (k // 2, k // 2), ops.ACT_NONE, 0.0)
            y.backward(gy)
        except RuntimeError as e:
            print("unsupported", Cin, Cout, k, str(e)[:60]); break
        ms = ops.probe_results_ms()
        ops.set_kernel_probe(None)
        x.grad = None; w.grad = None; b.grad = None
        if it:
            for kind, t in zip(keys, ms):
                best[kind] = min(best.get(kind, 1e9), t)
    fl = 2.0 * B * H * W * Cin * Cout * k * k
    out.append(f"{Cin:3d}->{Cout:3d} k{k:2d} {H}x{W}: " + "  ".join(f"{kind} {t:6.3f} ms {fl / t / 1e9:6.1f}" for kind, t in best.items()))
print(os.environ.get("MPA_FWD_FORCE", "planner"), "B", B)
print("\n".join(out))
