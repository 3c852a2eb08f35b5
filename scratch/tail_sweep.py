import ctypes, os, sys, torch
sys.path.insert(0, '.')
from multipitch_architectures_amd import _lib
from multipitch_architectures_amd._lib import ConvDesc
lib = _lib.load()
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
Cin, H, W, Cout = 128, 75, 216, 16
for B_ in (93, 104, 105, 116, 128, 139, 140, 151, 256):
    d = ConvDesc(B_, Cin, H, W, Cout, 15, 15, 1, 1, 7, 7)
    x = torch.randn(B_, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 15, 15, device=dev); y = torch.empty(B_, Cout, H, W, device=dev)
    n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), 0); wp = torch.empty(n, device=dev)
    lib.mpa_conv2d_pack(ctypes.byref(d), 0, P(w), P(wp), None)
    ts = []
    for it in range(3):
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        rc = lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp), None, P(y), 0, ctypes.c_float(0.0), None)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    blocks = B_ * 22
    print('B %3d blocks %5d rounds(768) %.2f  %.2f ms  %.4f ms/img' % (B_, blocks, blocks / 768, min(ts), min(ts) / B_))
