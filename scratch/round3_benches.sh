#!/bin/bash
# profiles/r03_bench_*: every BASELINE.json configuration (and the local batches of the 2/4/8-GPU shares) with the exact-fp32
# convolutions and with the opt-in bf16x3 ones, unprofiled bench.py lines, one gpurun call (boxes differ by a few per cent)
cd /root/repo
for spec in "SAUnet:L 256" "SAUnet:L 128" "SAUnet:L 64" "SAUnet:L 32" "DRCNN:L 64" "Unet:L 128" "BLUnet:XXL 256" "PUnet:XL 128"; do
  set -- $spec; cfg=$1; b=$2; tag=$(echo ${cfg}_b$b | tr ':' '_')
  for prec in f32 bf16x3; do
    sfx=""; [ $prec = bf16x3 ] && sfx="_bf16x3"
    python3 bench.py --config $cfg --global-batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-extras --conv-precision $prec > gpurun_out/r03_bench_$tag$sfx.json 2> gpurun_out/r03_bench_$tag$sfx.err
    python3 -c "import json; d=json.load(open('gpurun_out/r03_bench_$tag$sfx.json')); print('$tag', '$prec', round(d['ms_per_step'],2), 'ms', round(d['patches_per_s'],1), 'patches/s', round(d.get('step_mfma_frac',0)*100,1), '% of 157.3', 'roofline kernel', round(d['roofline']['launch_ms'],3), 'ms', round(d['roofline']['frac'],3))"
  done
done
