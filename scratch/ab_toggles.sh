#!/bin/bash
cd /root/repo
run() { env $2 python3 bench.py --global-batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'], 3), 'ms')"; }
for rep in 1 2; do
run all-new X=1
run attn-valu MPA_ATTN_VALU=1
run no-fanout MPA_NO_FANOUT=1
run no-poolskip MPA_NO_POOLSKIP=1
run none-of-them "MPA_ATTN_VALU=1 MPA_NO_FANOUT=1 MPA_NO_POOLSKIP=1"
(cd scratch/_r03 && python3 bench.py --global-batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r03-tree', round(d['ms_per_step'], 3), 'ms')")
done
