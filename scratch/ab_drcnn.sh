#!/bin/bash
cd /root/repo
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step')"; }
A="--config DRCNN:L --global-batch 64 --steps 8 --warmup 3 --no-cpu-baseline --no-extras"
python3 bench.py $A 2>/dev/null | line "fold+fused"
MPA_WG15_NOFOLD=1 python3 bench.py $A 2>/dev/null | line "nofold+fused"
python3 scratch/bench_unfused_tail.py $A 2>/dev/null | line "fold+unfused"
MPA_WG15_NOFOLD=1 python3 scratch/bench_unfused_tail.py $A 2>/dev/null | line "nofold+unfused"
python3 bench.py $A 2>/dev/null | line "fold+fused(again)"
