#!/bin/bash
# A/B of whole-step time, current library against multipitch_architectures_amd/csrc/libbase.keep, inside one gpurun
# call (boxes differ by a few per cent).  usage: ab_bench.sh "<config> <batch>" ...
cd /root/repo
run() { python3 bench.py --config $1 --global-batch $2 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', '$1', 'b$2', d['ms_per_step'], 'ms', d.get('roofline',{}).get('step_frac', ''))"; }
for spec in "$@"; do set -- $spec; run $1 $2 new; done
cp multipitch_architectures_amd/csrc/libmpa_hip.so /tmp/libnew.so
cp multipitch_architectures_amd/csrc/libbase.keep multipitch_architectures_amd/csrc/libmpa_hip.so
for spec in "$@"; do set -- $spec; MPA_BASELINE_LIB=1 run $1 $2 base; done
cp /tmp/libnew.so multipitch_architectures_amd/csrc/libmpa_hip.so
