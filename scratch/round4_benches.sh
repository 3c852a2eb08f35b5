#!/bin/bash
# profiles/r04_bench_*: every BASELINE.json configuration (and the local batches of the 2/4/8-GPU shares, PUnet:XL at the 256
# of SURVEY 8(d), SAUnet:L at T = 174) with the exact-fp32 convolutions, and the headline shapes with the opt-in bf16x3 ones;
# unprofiled bench.py lines, one gpurun call (boxes differ by a few per cent)
cd /root/repo
one() {   # tag, extra args...
  tag=$1; shift
  python3 bench.py "$@" --steps 8 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r04_bench_$tag.json 2> gpurun_out/r04_bench_$tag.err
  python3 -c "import json; d=json.load(open('gpurun_out/r04_bench_$tag.json')); print('$tag', round(d['ms_per_step'],2), 'ms', round(d['patches_per_s'],1), 'patches/s', round(d.get('step_mfma_frac',0)*100,1), '% of 157.3', 'roofline kernel', round(d['roofline']['launch_ms'],3), 'ms', round(d['roofline']['frac'],3), d.get('dp_segments'))"
}
for spec in "SAUnet:L 256" "SAUnet:L 128" "SAUnet:L 64" "SAUnet:L 32" "DRCNN:L 64" "Unet:L 128" "BLUnet:XXL 256" "PUnet:XL 128" "PUnet:XL 256"; do
  set -- $spec; cfg=$1; b=$2; tag=$(echo ${cfg}_b$b | tr ':' '_')
  one $tag --config $cfg --global-batch $b
done
one SAUnet_L_T174_b64 --frames 174 --global-batch 64
one SAUnet_L_b32_dp_rehearsal --global-batch 32 --dp-rehearsal
one SAUnet_L_b32_nograph --global-batch 32 --no-graph
for spec in "SAUnet:L 256" "SAUnet:L 32" "DRCNN:L 64" "BLUnet:XXL 256"; do
  set -- $spec; cfg=$1; b=$2; tag=$(echo ${cfg}_b$b | tr ':' '_')
  one ${tag}_bf16x3 --config $cfg --global-batch $b --conv-precision bf16x3
done
