import ctypes, os, sys, torch
sys.path.insert(0, '.')
from multipitch_architectures_amd import _lib
from multipitch_architectures_amd._lib import ConvDesc
lib = _lib.load()
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
buf = ctypes.create_string_buffer(512)
for (B_, Cin, H, W, Cout) in [(256, 16, 75, 216, 128), (256, 128, 75, 216, 16), (256, 32, 75, 216, 16)]:
    d = ConvDesc(B_, Cin, H, W, Cout, 15, 15, 1, 1, 7, 7)
    lib.mpa_conv2d_describe_plan(ctypes.byref(d), 0, buf, 512); print(buf.value.decode())
    x = torch.randn(B_, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 15, 15, device=dev); y = torch.empty(B_, Cout, H, W, device=dev)
    n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), 0); wp = torch.empty(n, device=dev)
    lib.mpa_conv2d_pack(ctypes.byref(d), 0, P(w), P(wp), None)
    for dbg in ('0', '4', '5'):
        os.environ['MPA_DEBUG_FWD'] = dbg
        ts = []
        for it in range(3):
            a, b = torch.cuda.Event(True), torch.cuda.Event(True)
            a.record()
            rc = lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp), None, P(y), 0, ctypes.c_float(0.0), None)
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print('  ', (Cin, Cout), 'dbg', dbg, 'rc', rc, ['%.2f' % t for t in ts])
