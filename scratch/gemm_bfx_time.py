"""time mpa_gemm vs mpa_gemm_bf16x3 on the models' large products"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
# (tag, M, N, K, lda_m, lda_k, ldb_k, ldb_n)
def cases():
    for rows, N, K, tag in [(256*174, 64, 128, "conv3-like"), (256*75, 512, 512, "lstm ih B256"), (256*75, 2048, 512, "lstm ih 4H"),
                            (256*174, 8192//8, 8192//8, "sq 1024"), (4096, 4096, 4096, "sq 4096")]:
        yield (tag + " fwd", rows, N, K, K, 1, 1, K)
        yield (tag + " bwd-data", rows, K, N, N, 1, K, 1)
        yield (tag + " bwd-w", N, K, rows, 1, N, K, 1)
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
    print(f"{tag:24s} M={M:6d} N={N:5d} K={K:6d}  f32 {out[0]:7.3f} ms {fl/out[0]/1e9:6.1f} TF   bf16x3 {out[1]:7.3f} ms {fl/out[1]/1e9:6.1f} TF", flush=True)
