#!/bin/bash
# round-3 extras: the T = 174 shape in both precisions, kernel stats of the remaining BASELINE configurations, and the
# three PMC passes of the exact-fp32 step at the final commit
cd /root/repo; export TMPDIR=/tmp
python3 bench.py --frames 174 --global-batch 64 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03_bench_SAUnet_L_T174_b64.json 2>/dev/null
python3 bench.py --frames 174 --global-batch 64 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --conv-precision bf16x3 > gpurun_out/r03_bench_SAUnet_L_T174_b64_bf16x3.json 2>/dev/null
for spec in "Unet_L_b128 --config Unet:L --global-batch 128" "PUnet_XL_b128 --config PUnet:XL --global-batch 128"; do
  set -- $spec; tag=$1; shift
  bash tools_profile.sh r03_$tag "$@" --steps 5 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "$tag prof rc=$?"
done
bash scratch/pmc_passes.sh r03f --steps 2 --warmup 2
