#!/bin/bash
# A/B of whole-step time, current library against multipitch_architectures_amd/csrc/libbase.keep, inside one gpurun
# call (boxes differ by a few per cent).  usage: ab_bench.sh "<config> <batch>" ...
cd /root/repo
specs=("$@")
run() { python3 bench.py --config $1 --global-batch $2 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', '$1', 'b$2', round(d['ms_per_step'], 2), 'ms', round(d['step_mfma_frac'] * 100, 1), '%')"; }
for spec in "${specs[@]}"; do run $spec new; done
cp multipitch_architectures_amd/csrc/libmpa_hip.so /tmp/libnew.so
cp multipitch_architectures_amd/csrc/libbase.keep multipitch_architectures_amd/csrc/libmpa_hip.so
for spec in "${specs[@]}"; do run $spec base; done
cp /tmp/libnew.so multipitch_architectures_amd/csrc/libmpa_hip.so
for spec in "${specs[@]}"; do run $spec new; done
