"""diagnostic: build a stamped copy of the library and time the phases of conv_wgrad15 on one layer"""
import ctypes, os, subprocess, sys, torch
sys.path.insert(0, '.')
from multipitch_architectures_amd import build as B
from multipitch_architectures_amd._lib import ConvDesc
lib_path = '/tmp/libmpa_stamps.so'
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
subprocess.check_call(['hipcc', *B.FLAGS, '-DMPA_STAMPS', '-shared', '-o', lib_path, *srcs])
lib = ctypes.CDLL(lib_path)
lib.mpa_conv2d_bwd_weight_workspace.restype = ctypes.c_int64
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (B_, Cin, H, W, Cout) in [(256, 16, 75, 216, 128), (256, 32, 75, 216, 16)]:
    d = ConvDesc(B_, Cin, H, W, Cout, 15, 15, 1, 1, 7, 7)
    x = torch.randn(B_, Cin, H, W, device=dev); dy = torch.randn(B_, Cout, H, W, device=dev)
    dw = torch.empty(Cout, Cin, 15, 15, device=dev); db = torch.empty(Cout, device=dev)
    n = lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d)); ws = torch.empty(n // 4, device=dev)
    st = (ctypes.c_ulonglong * 8)()
    for it in range(2):
        lib.mpa_debug_read_stamps(st, 1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        rc = lib.mpa_conv2d_bwd_weight(ctypes.byref(d), P(x), P(dy), P(dw), P(db), P(ws), ctypes.c_int64(n), None)
        b.record(); torch.cuda.synchronize()
        lib.mpa_debug_read_stamps(st, 0)
    tiles = st[6]
    names = ['top barrier', 'glds issue', 'vmcnt wait', 'post barrier', 'bias sums', 'mfma loop']
    tot = sum(st[i] for i in range(6))
    print((Cin, Cout), 'rc', rc, '%.2f ms' % a.elapsed_time(b), 'tiles', tiles, 'cycles/tile (100MHz ticks?)', tot / max(tiles, 1))
    for i, nm in enumerate(names):
        print('   %-13s %8.0f per tile  %5.1f%%' % (nm, st[i] / max(tiles, 1), 100.0 * st[i] / max(tot, 1)))
