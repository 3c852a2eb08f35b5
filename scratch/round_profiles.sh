#!/bin/bash
# everything profiles/r02_* is made of, at the current state of the tree (run on the GPU box through gpurun):
#   bench lines (unprofiled) + rocprofv3 kernel stats of the same commands for every BASELINE.json configuration and
#   the local-batch-32 share of the 8-GPU run, then the three PMC passes on the headline configuration
cd /root/repo; export TMPDIR=/tmp
for spec in "SAUnet:L 256" "SAUnet:L 32" "DRCNN:L 64" "Unet:L 128" "BLUnet:XXL 256" "PUnet:XL 128"; do
  set -- $spec; cfg=$1; b=$2; tag=$(echo ${cfg}_b$b | tr ':' '_')
  python3 bench.py --config $cfg --global-batch $b --steps 10 --warmup 3 > gpurun_out/r02_bench_$tag.json 2> gpurun_out/r02_bench_$tag.err
  echo "$tag bench rc=$?"
  bash tools_profile.sh r02_$tag --config $cfg --global-batch $b --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "$tag prof rc=$?"
done
python3 bench.py --config SAUnet:L --global-batch 32 --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-extras > gpurun_out/r02_bench_SAUnet_L_b32_nograph.json 2>/dev/null
bash scratch/pmc_passes.sh r02 --steps 2 --warmup 2
