"""describe the plans of one conv layer: python scratch/plan_one.py B Cin H W Cout kh kw [sh sw ph pw]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multipitch_architectures_amd import _lib as L
a = [int(v) for v in sys.argv[1:]]
a += [1, 1, 0, 0][len(a) - 7:]
lib = L.load()
d = L.ConvDesc(*a)
buf = ctypes.create_string_buffer(1024)
for mode, name in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
    lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 1024)
    print(name, buf.value.decode())
