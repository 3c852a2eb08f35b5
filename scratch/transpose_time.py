"""the (B, R, C) -> (B, C, R) copies of the step at batch B (argv): shapes recorded from one SAUnet:L step"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for R, C in [(6000, 72), (72, 6000), (128, 52), (52, 128), (3750, 72), (72, 3750)]:
    x = torch.randn(B, R, C, device=dev); y = torch.empty(B, C, R, device=dev)
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lib.mpa_transpose_add(P(x), None, P(y), B, R, C, 0, st); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    assert torch.equal(y, x.transpose(1, 2).contiguous())
    print(f"B={B} ({R},{C}): {best*1e3:7.1f} us  {2*x.numel()*4/best/1e9:5.2f} TB/s")
