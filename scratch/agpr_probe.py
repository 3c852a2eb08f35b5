"""build a copy of the library WITHOUT -amdgpu-mfma-vgpr-form and time the 16->128 forward in dbg modes"""
import ctypes, os, subprocess, sys, torch
sys.path.insert(0, '.')
from multipitch_architectures_amd import build as B
from multipitch_architectures_amd._lib import ConvDesc
flags = [f for f in B.FLAGS if f not in ('-mllvm', '-amdgpu-mfma-vgpr-form')]
lib_path = '/tmp/libmpa_agpr.so'
subprocess.check_call(['hipcc', *flags, '-shared', '-o', lib_path, os.path.join(B.CSRC, 'conv.hip'), os.path.join(B.CSRC, 'pointwise.hip')])
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
for name, path in (('agpr', lib_path), ('vgpr', B.LIB)):
    lib = ctypes.CDLL(path)
    lib.mpa_conv2d_packed_floats.restype = ctypes.c_int64
    for (B_, Cin, H, W, Cout) in [(256, 16, 75, 216, 128), (256, 32, 37, 108, 32)]:
        d = ConvDesc(B_, Cin, H, W, Cout, 15, 15, 1, 1, 7, 7)
        x = torch.randn(B_, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 15, 15, device=dev); y = torch.empty(B_, Cout, H, W, device=dev)
        n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), 0); wp = torch.empty(n, device=dev)
        lib.mpa_conv2d_pack(ctypes.byref(d), 0, P(w), P(wp), None)
        for dbg in ('0', '4'):
            os.environ['MPA_DEBUG_FWD'] = dbg
            ts = []
            for it in range(3):
                a, b = torch.cuda.Event(True), torch.cuda.Event(True)
                a.record()
                rc = lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp), None, P(y), 0, ctypes.c_float(0.0), None)
                b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            print(name, (Cin, Cout, H), 'dbg', dbg, 'rc', rc, ['%.2f' % t for t in ts])
