"""does a second pass over a buffer run faster when the buffer fits the 256 MB Infinity Cache?  in-tree kernels only:
mpa_colsum over a (rows, 1024) fp32 view, repeated back to back, for sizes from 32 MB to 2 GB"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
out = torch.empty(1024, device="cuda")
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    rows = mb * 256
    x = torch.randn(rows, 1024, device="cuda")
    for _ in range(3): lib.mpa_colsum(P(x), P(out), rows, 1024, 0, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): lib.mpa_colsum(P(x), P(out), rows, 1024, 0, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{mb:5d} MB  {ms:7.3f} ms  {mb / 1024 / (ms / 1e3) / 1e3:6.2f} TB/s", flush=True)
    del x
