"""time the bf16x3 15x15 launches against the exact-fp32 ones (same process, interleaved)"""
import sys, torch
sys.path.insert(0, ".")
from multipitch_architectures_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
LAYERS = [(16, 128, 75, 216), (32, 16, 75, 216), (16, 16, 75, 216), (6, 16, 75, 216), (16, 32, 37, 108), (32, 32, 37, 108)]
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for cin, cout, H, W in LAYERS:
    x = torch.randn(B, cin, H, W, device=dev); w = torch.randn(cout, cin, 15, 15, device=dev) * 0.02
    dy = torch.randn(B, cout, H, W, device=dev)
    flop = 2.0 * B * cout * cin * 225 * H * W
    res = {}
    for prec in ("f32", "bf16x3"):
        ops.set_conv_precision(prec)
        xr = x.clone().requires_grad_(True)
        res[prec, "fwd"] = t(lambda: ops.conv2d(x, w, None, (1, 1), (7, 7)))
        y = ops.conv2d(xr, w, None, (1, 1), (7, 7))
        res[prec, "dgrad"] = t(lambda: torch.autograd.grad(y, xr, dy, retain_graph=True))
        wr = w.clone().requires_grad_(True)
        y2 = ops.conv2d(x, wr, None, (1, 1), (7, 7))
        res[prec, "wgrad"] = t(lambda: torch.autograd.grad(y2, wr, dy, retain_graph=True))
    if True:
        ops.set_conv_precision("bf16x3")
        res["split", "x"] = t(lambda: ops.split_bf16(x)); res["split", "dy"] = t(lambda: ops.split_bf16(dy))
    print(f"{cin:3d}->{cout:3d} {H}x{W} B={B}: fwd f32 {res['f32','fwd']:7.3f} ms ({flop/res['f32','fwd']/1e9:6.1f} TF)  bf16x3 {res['bf16x3','fwd']:7.3f} ms "
          f"({flop/res['bf16x3','fwd']/1e9:6.1f} TF incl. split {res['split','x']:.3f}) | dgrad f32 {res['f32','dgrad']:7.3f}  bf16x3 {res['bf16x3','dgrad']:7.3f} ms "
          f"({flop/res['bf16x3','dgrad']/1e9:6.1f} TF incl. split {res['split','dy']:.3f}) | wgrad f32 {res['f32','wgrad']:7.3f}  bf16x3 {res['bf16x3','wgrad']:7.3f} ms "
          f"({flop/res['bf16x3','wgrad']/1e9:6.1f} TF incl. split)", flush=True)
