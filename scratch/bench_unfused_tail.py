"""bench.py with the CNN families' prefilter tail as three separate kernels (pool, dropout, add), for A/B runs"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multipitch_architectures_amd import ops
from multipitch_architectures_amd.nn_models import layers


def forward(self, x, residual=None):
    mods = list(self)
    h = mods[0](x, mods[1].act, mods[1].slope)
    for m in mods[2:]:
        h = m(h)
    return h if residual is None else ops.add(h, residual)


layers.ConvActPoolDrop.forward = forward
import bench
bench.main()
