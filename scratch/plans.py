"""print the planner's choice for a convolution problem (host-only: works without a GPU)
usage: python scratch/plans.py B Cin H W Cout kh kw sh sw ph pw"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multipitch_architectures_amd import _lib as L
lib = L.load()
def show(*a):
    d = L.ConvDesc(*a)
    for mode, nm in ((0, "fwd "), (1, "dgrd"), (2, "wgrd")):
        buf = ctypes.create_string_buffer(512)
        rc = lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512)
        print(nm, a, buf.value.decode() if rc == 0 else rc)
if len(sys.argv) > 2:
    show(*map(int, sys.argv[1:12]))
else:
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for g in [(16,75,216,128,15,15,1,1,7,7),(32,75,216,16,15,15,1,1,7,7),(6,75,216,16,15,15,1,1,7,7),(16,75,216,16,15,15,1,1,7,7),
              (32,37,108,32,15,15,1,1,7,7),(16,37,108,32,15,15,1,1,7,7),(64,37,108,32,9,9,1,1,4,4),(32,37,108,16,9,9,1,1,4,4),
              (128,75,216,80,3,3,1,3,1,0)]:
        show(B, *g)
