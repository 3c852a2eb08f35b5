#!/bin/bash
# everything profiles/r03_* is made of at the final state of round 3 (run on the GPU box through gpurun):
#   1. unprofiled bench.py lines of every BASELINE.json configuration and of the local batches of the 2/4/8-GPU shares, with
#      the exact-fp32 convolutions and with the opt-in bf16x3 ones (scratch/round3_benches.sh)
#   2. rocprofv3 kernel stats of the headline configuration in both precisions and of the local-batch-32 share
#   3. the default `python bench.py` line (all probes, cpu_baseline) with its wall time
cd /root/repo; export TMPDIR=/tmp
bash scratch/round3_benches.sh
for spec in "SAUnet_L_b256 --global-batch 256" "SAUnet_L_b32 --global-batch 32" "SAUnet_L_b256_bf16x3 --global-batch 256 --conv-precision bf16x3" "DRCNN_L_b64 --config DRCNN:L --global-batch 64" "BLUnet_XXL_b256 --config BLUnet:XXL --global-batch 256"; do
  set -- $spec; tag=$1; shift
  bash tools_profile.sh r03_$tag "$@" --steps 5 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "$tag prof rc=$?"
done
t0=$(date +%s)
python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
echo "default bench rc=$? wall=$(( $(date +%s) - t0 )) s"
