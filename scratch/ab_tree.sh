#!/bin/bash
# A/B of whole-step time: this tree against the round-3 tree exported to scratch/_r03 (its own Python + its own library), inside
# one gpurun call.  usage: ab_tree.sh "<config> <batch>" ...
cd /root/repo
run() { (cd $1 && python3 bench.py --config $2 --global-batch $3 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$4', '$2', 'b$3', round(d['ms_per_step'], 2), 'ms', round(d.get('step_mfma_frac', 0) * 100, 1), '%')"); }
for spec in "$@"; do run . $spec new; done
for spec in "$@"; do run scratch/_r03 $spec r03; done
for spec in "$@"; do run . $spec new; done
for spec in "$@"; do run scratch/_r03 $spec r03; done
