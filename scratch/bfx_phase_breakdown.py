"""where a conv_bfx_kernel launch spends its time: MPA_BFX_DEBUG variants (results wrong, timing only), 16->128 and 128->16"""
import os, subprocess, sys
code = r'''
import sys, torch
sys.path.insert(0, ".")
from multipitch_architectures_amd import ops
ops.set_conv_precision("bf16x3")
dev = torch.device("cuda:0")
B = 64
for cin, cout in ((16, 128), (128, 16), (16, 16)):
    x = torch.randn(B, cin, 75, 216, device=dev); w = torch.randn(cout, cin, 15, 15, device=dev) * 0.02
    xs = ops.split_bf16(x)
    f = lambda: ops.conv2d(x, w, None, (1, 1), (7, 7))
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(5): ops.split_bf16(x)
    s1.record()
    a.record()
    for _ in range(5): f()
    b.record(); torch.cuda.synchronize()
    print(f"{cin}->{cout}: conv+split {a.elapsed_time(b)/5:.3f} ms, split {s0.elapsed_time(s1)/5:.3f} ms")
'''
for dbg in ("0", "1", "2", "3"):
    env = dict(os.environ, MPA_BFX_DEBUG=dbg)
    print("MPA_BFX_DEBUG =", dbg, flush=True)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:], flush=True)
