#!/bin/bash
# A/B of whole-step time with and without one environment switch, inside one gpurun call: ab_env.sh VAR=VALUE "<config> <batch>" ...
cd /root/repo
sw=$1; shift
specs=("$@")
run() { env $3 python3 bench.py --config $1 --global-batch $2 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$4', '$1', 'b$2', round(d['ms_per_step'], 2), 'ms', round(d['step_mfma_frac'] * 100, 1), '%')"; }
for rep in 1 2; do
  for spec in "${specs[@]}"; do run $spec X_UNUSED=1 default; done
  for spec in "${specs[@]}"; do run $spec $sw "$sw"; done
done
