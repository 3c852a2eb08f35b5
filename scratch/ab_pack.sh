#!/bin/bash
cd /root/repo
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), 'ms/step')"; }
for spec in "SAUnet:L 32" "SAUnet:L 64" "SAUnet:L 256" "DRCNN:L 64"; do
  set -- $spec
  for g in "" "--no-graph"; do
    A="--config $1 --global-batch $2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras $g"
    python3 bench.py $A 2>/dev/null | line "table   $1 b$2 $g"
    MPA_PACK_TABLES=0 python3 bench.py $A 2>/dev/null | line "lazy    $1 b$2 $g"
    python3 bench.py $A 2>/dev/null | line "table   $1 b$2 $g"
  done
done
