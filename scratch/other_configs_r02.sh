#!/bin/bash
# bench.py for the other BASELINE.json configurations (unprofiled JSON line + rocprofv3 kernel stats of the same command)
cd /root/repo; export TMPDIR=/tmp
for spec in "DRCNN:L 64" "Unet:L 128" "BLUnet:XXL 256" "PUnet:XL 128" "SAUnet:L 256" "SAUnet:L 32"; do
  set -- $spec; cfg=$1; b=$2; tag=$(echo ${cfg}_b$b | tr ':' '_')
  python3 bench.py --config $cfg --global-batch $b --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02_cfg_$tag.json 2> gpurun_out/r02_cfg_$tag.err
  echo "$tag rc=$?"
  bash tools_profile.sh r02_$tag --config $cfg --global-batch $b --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "$tag prof rc=$?"
done
