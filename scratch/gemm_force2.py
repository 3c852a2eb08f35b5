"""the MLP products of SAUnet:L under every tile variant / K split of gemm_kernel (MPA_GEMM_FORCE="variant,splits")"""
import subprocess, sys, os
shapes = [("fc1 fwd", 13312, 8192, 128, 128, 1, 1, 128), ("fc1 dgrad", 13312, 128, 8192, 8192, 1, 128, 1), ("fc1 wgrad", 8192, 128, 13312, 1, 8192, 128, 1),
          ("fc2 fwd", 13312, 128, 8192, 8192, 1, 1, 8192), ("fc2 dgrad", 13312, 8192, 128, 128, 1, 8192, 1), ("fc2 wgrad", 128, 8192, 13312, 1, 128, 8192, 1)]
code = r'''
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load(); P = lambda t: ctypes.c_void_p(t.data_ptr()); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, lam, lak, lbk, lbn = map(int, sys.argv[1:8])
A = torch.randn(M * K, device="cuda"); B = torch.randn(K * N, device="cuda"); C = torch.empty(M * N, device="cuda")
f = lambda: lib.mpa_gemm(P(A), lam, lak, P(B), lbk, lbn, None, P(C), N, M, N, K, 0, 0, st)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{ms:7.3f} ms {2.0*M*N*K/ms/1e9:6.1f} TF")
'''
open("/tmp/gf2.py", "w").write(code)
for name, M, N, K, lam, lak, lbk, lbn in shapes:
    res = []
    for force in ["auto"] + [f"{v},{s}" for v in range(4) for s in (1, 2, 4, 8, 16)]:
        if force != "auto" and int(force.split(",")[1]) > 1 and K < 1024: continue
        env = dict(os.environ)
        if force != "auto": env["MPA_GEMM_FORCE"] = force
        r = subprocess.run([sys.executable, "/tmp/gf2.py", *map(str, (M, N, K, lam, lak, lbk, lbn))], env=env, capture_output=True, text=True)
        out = r.stdout.strip().splitlines()
        res.append((force, out[-1] if out else "ERR " + r.stderr[-100:]))
    print(name, M, N, K, " | ".join(f"{f}: {o}" for f, o in res), flush=True)
