import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(M, N, K, ak=True, bk=True):
    A = torch.randn(M, K, device="cuda") if ak else torch.randn(K, M, device="cuda")
    Bm = torch.randn(N, K, device="cuda") if bk else torch.randn(K, N, device="cuda")
    C = torch.empty(M, N, device="cuda"); b = torch.randn(N, device="cuda")
    la = (K, 1) if ak else (1, M); lb = (1, K) if bk else (N, 1)
    f = lambda: lib.mpa_gemm(P(A), la[0], la[1], P(Bm), lb[0], lb[1], P(b), P(C), N, M, N, K, 0, 0, st)
    assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    Ad = (A if ak else A.T)[:96].double(); Bd = (Bm.T if bk else Bm).double()
    ref = Ad @ Bd + b.double()
    err = (C[:96].double() - ref).abs().max().item() / ref.abs().max().item()
    Ad2 = (A if ak else A.T)[-96:].double(); ref2 = Ad2 @ Bd + b.double()
    err2 = (C[-96:].double() - ref2).abs().max().item() / ref2.abs().max().item()
    print(f"M={M} N={N} K={K} A{'k' if ak else 'm'} B{'k' if bk else 'n'}: {ms:7.3f} ms {2.0*M*N*K/ms/1e9:6.1f} TF/s err {err:.1e} {err2:.1e}", flush=True)
for args in [(18432,150,15000,True,True),(18432,15000,150,True,False),(150,15000,18432,False,False),
             (9216,100,11250,True,True),(9216,11250,100,True,False),(100,11250,9216,False,False),
             (18432,50,6000,True,True),(18432,6000,50,True,False),(50,6000,18432,False,False),
             (13312,8192,128,True,True),(13312,128,8192,True,False),(4096,4096,4096,True,True),(1000,150,333,True,True),(150,160,77,False,False)]:
    run(*args)
