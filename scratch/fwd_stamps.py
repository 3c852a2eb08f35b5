import ctypes, os, subprocess, sys, torch
sys.path.insert(0, '.')
from multipitch_architectures_amd import build as B
from multipitch_architectures_amd._lib import ConvDesc
lib_path = '/tmp/libmpa_stamps.so'
subprocess.check_call(['hipcc', *B.FLAGS, '-DMPA_STAMPS', '-shared', '-o', lib_path, os.path.join(B.CSRC, 'conv.hip'), os.path.join(B.CSRC, 'pointwise.hip')])
lib = ctypes.CDLL(lib_path)
lib.mpa_conv2d_packed_floats.restype = ctypes.c_int64
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (B_, Cin, H, W, Cout) in [(256, 16, 75, 216, 128), (256, 128, 75, 216, 16)]:
    d = ConvDesc(B_, Cin, H, W, Cout, 15, 15, 1, 1, 7, 7)
    x = torch.randn(B_, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 15, 15, device=dev); y = torch.empty(B_, Cout, H, W, device=dev)
    n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), 0); wp = torch.empty(n, device=dev)
    lib.mpa_conv2d_pack(ctypes.byref(d), 0, P(w), P(wp), None)
    st = (ctypes.c_ulonglong * 8)()
    for dbg in ('0', '4'):
        os.environ['MPA_DEBUG_FWD'] = dbg
        for it in range(2):
            lib.mpa_debug_read_stamps(st, 1)
            a, b = torch.cuda.Event(True), torch.cuda.Event(True)
            a.record()
            rc = lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp), None, P(y), 0, ctypes.c_float(0.0), None)
            b.record(); torch.cuda.synchronize()
            lib.mpa_debug_read_stamps(st, 0)
        nb = max(st[3], 1)
        print((Cin, Cout), 'dbg', dbg, '%.2f ms' % a.elapsed_time(b), 'blocks', st[3], 'cycles/block: prologue %.0f  main %.0f  epilogue %.0f' % (st[0] / nb, st[1] / nb, st[2] / nb))
