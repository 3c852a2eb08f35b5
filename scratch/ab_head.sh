#!/bin/bash
cd /root/repo
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), 'ms/step', round(d['step_mfma_frac']*100,1), '%')"; }
for spec in "SAUnet:L 256" "Unet:L 128" "BLUnet:XXL 256" "SAUnet:L 32"; do
  set -- $spec
  A="--config $1 --global-batch $2 --steps 8 --warmup 3 --no-cpu-baseline --no-extras"
  python3 bench.py $A 2>/dev/null | line "fused   $1 b$2"
  python3 scratch/bench_head_unfused.py $A 2>/dev/null | line "unfused $1 b$2"
  python3 bench.py $A 2>/dev/null | line "fused   $1 b$2"
done
