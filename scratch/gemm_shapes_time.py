"""time mpa_gemm (and the bf16x3 variant) on the products SAUnet:L / BLUnet:XXL issue at their BASELINE batch"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def cases():
    rows = 256 * 52
    for N, K, tag in [(8192, 128, "mlp fc1"), (128, 8192, "mlp fc2"), (384, 128, "in_proj"), (128, 128, "out_proj"),
                      (4 * 832, 1664, "blstm ih"), (1664, 1664, "blstm lin")]:
        r = rows if "blstm" not in tag else 256 * 75
        yield (tag + " fwd", r, N, K, K, 1, 1, K)
        yield (tag + " bwd-data", r, K, N, N, 1, K, 1)
        yield (tag + " bwd-w", N, K, r, 1, N, K, 1)
tot = [0.0, 0.0]
for tag, M, N, K, lda_m, lda_k, ldb_k, ldb_n in cases():
    A = torch.randn(M * K, device=dev); B = torch.randn(K * N, device=dev); C = torch.zeros(M * N, device=dev)
    out = []
    for fn in (lib.mpa_gemm, lib.mpa_gemm_bf16x3):
        if fn is lib.mpa_gemm_bf16x3 and not lib.mpa_gemm_bf16x3_supported(p(A), lda_m, lda_k, p(B), ldb_k, ldb_n, M, N, K):
            out.append(float("nan")); continue
        for _ in range(3): fn(p(A), lda_m, lda_k, p(B), ldb_k, ldb_n, None, p(C), N, M, N, K, 0, 0, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn(p(A), lda_m, lda_k, p(B), ldb_k, ldb_n, None, p(C), N, M, N, K, 0, 0, s)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    print(f"{tag:20s} M={M:6d} N={N:5d} K={K:6d}  f32 {out[0]:7.3f} ms {fl/out[0]/1e9:6.1f} TF   bf16x3 {out[1]:7.3f} ms {fl/out[1]/1e9:6.1f} TF", flush=True)
