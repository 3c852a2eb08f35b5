"""bench.py with the head stage's tail (13-row pool, dropout) as separate kernels, for A/B runs"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multipitch_architectures_amd import ops
ops.POOLROWS_KH = (3,)
import bench
bench.main()
