import ctypes, os, sys, torch
sys.path.insert(0, '.')
from multipitch_architectures_amd import _lib
from multipitch_architectures_amd._lib import ConvDesc
lib = _lib.load()
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (B_, Cin, H, W, Cout) in [(256, 16, 75, 216, 128), (256, 32, 75, 216, 16)]:
    d = ConvDesc(B_, Cin, H, W, Cout, 15, 15, 1, 1, 7, 7)
    x = torch.randn(B_, Cin, H, W, device=dev); dy = torch.randn(B_, Cout, H, W, device=dev)
    dw = torch.empty(Cout, Cin, 15, 15, device=dev); db = torch.empty(Cout, device=dev)
    n = lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d)); ws = torch.empty(n // 4, device=dev)
    for dbg in ('0', '1', '2'):
        os.environ['MPA_DEBUG_WG15'] = dbg
        ts = []
        for it in range(3):
            a, b = torch.cuda.Event(True), torch.cuda.Event(True)
            a.record()
            rc = lib.mpa_conv2d_bwd_weight(ctypes.byref(d), P(x), P(dy), P(dw), P(db), P(ws), ctypes.c_int64(n), None)
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print((Cin, Cout), 'dbg', dbg, 'rc', rc, ['%.2f' % t for t in ts])
