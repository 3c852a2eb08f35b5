import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(13312, 8192, 128), (13312, 8192, 512), (13312, 8192, 2048), (13312, 128, 8192), (8192, 128, 13312), (4096, 4096, 4096), (13312, 128, 128), (1664, 8192, 128)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(M, N, device="cuda"); b = torch.randn(N, device="cuda")
    f = lambda: lib.mpa_gemm(P(x), K, 1, P(w), 1, K, P(b), P(y), N, M, N, K, 0, 0, st)
    assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ref = x[:64].double() @ w.double().T + b.double()
    err = (y[:64].double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"M={M} N={N} K={K}: {ms:7.3f} ms {2.0*M*N*K/ms/1e9:6.1f} TF/s  C write {M*N*4/ms/1e6:6.0f} GB/s  err {err:.1e}")
