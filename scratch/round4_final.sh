#!/bin/bash
# everything profiles/r04_* is made of at the final state of round 4 (run on the GPU box through gpurun):
#   1. unprofiled bench.py lines (scratch/round4_benches.sh)
#   2. rocprofv3 kernel stats of the headline configuration, of the local-batch-32 share, of the data-parallel rehearsal and of
#      the other BASELINE configurations
#   3. the default `python bench.py` line (all probes, t174 and bf16x3 sub-objects, cpu_baseline) with its wall time
cd /root/repo; export TMPDIR=/tmp
bash scratch/round4_benches.sh
for spec in "SAUnet_L_b256 --global-batch 256" "SAUnet_L_b32 --global-batch 32" "SAUnet_L_b32_dp_rehearsal --global-batch 32 --dp-rehearsal" "DRCNN_L_b64 --config DRCNN:L --global-batch 64" "Unet_L_b128 --config Unet:L --global-batch 128" "BLUnet_XXL_b256 --config BLUnet:XXL --global-batch 256" "PUnet_XL_b256 --config PUnet:XL --global-batch 256"; do
  set -- $spec; tag=$1; shift
  bash tools_profile.sh r04_$tag "$@" --steps 5 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "$tag prof rc=$?"
done
t0=$(date +%s)
python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
echo "default bench rc=$? wall=$(( $(date +%s) - t0 )) s"
