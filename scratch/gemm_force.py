import sys, os, ctypes, subprocess
shapes = [(18432,150,15000,1,1),(150,15000,18432,0,0),(18432,15000,150,1,0),(9216,100,11250,1,1)]
code = '''
import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M,N,K,ak,bk = map(int, sys.argv[1:6])
A = torch.randn(M, K, device="cuda") if ak else torch.randn(K, M, device="cuda")
Bm = torch.randn(N, K, device="cuda") if bk else torch.randn(K, N, device="cuda")
C = torch.empty(M, N, device="cuda")
la = (K, 1) if ak else (1, M); lb = (1, K) if bk else (N, 1)
f = lambda: lib.mpa_gemm(P(A), la[0], la[1], P(Bm), lb[0], lb[1], None, P(C), N, M, N, K, 0, 0, st)
f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"{os.environ.get('MPA_GEMM_FORCE','auto'):>6s} M={M} N={N} K={K}: {ms:7.3f} ms {2.0*M*N*K/ms/1e9:6.1f} TF/s")
'''
open("/tmp/gf.py","w").write(code)
for sh in shapes:
    for force in ["auto","0,1","0,2","0,4","1,2","1,4","2,2","2,4","4,1","4,2","4,4","5,1","5,2","5,4"]:
        env = dict(os.environ)
        if force != "auto": env["MPA_GEMM_FORCE"] = force
        r = subprocess.run([sys.executable, "/tmp/gf.py", *map(str, sh)], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:], flush=True)
