#!/bin/bash
# Two data-parallel ranks over gloo sharing the one GPU of the box, each under its own rocprofv3 (kernel + memory-copy trace):
# where do the bucket all-reduces (device->host copy, host reduction, host->device copy with gloo) sit relative to the graph
# segments of backward?  Output: gpurun_out/prof_dpov/rank*/...
cd /root/repo; export TMPDIR=/tmp
out=gpurun_out/prof_dpov; rm -rf $out; mkdir -p $out
port=$((20000 + RANDOM % 20000))
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port MPA_DIST_BACKEND=gloo \
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/rank$r -o dp -- \
    python3 bench.py --gpus 2 --global-batch 64 --steps 3 --warmup 3 --no-cpu-baseline --no-extras > $out/rank$r.log 2>&1 &
  pids[$r]=$!
done
wait ${pids[0]}; rc0=$?; wait ${pids[1]}; rc1=$?
echo rc $rc0 $rc1
tail -1 $out/rank0.log | cut -c1-400
python3 scratch/dp_overlap_report.py $out/rank0 | tee $out/report.txt
