#!/bin/bash
# kernel traces of the local-batch-32 step of this tree and of the round-3 tree (scratch/_r03) in one call: where are the gaps?
cd /root/repo; export TMPDIR=/tmp
for t in new r03; do
  dir=.; [ $t = r03 ] && dir=scratch/_r03
  out=/root/repo/gpurun_out/gap_$t; rm -rf $out; mkdir -p $out
  (cd $dir && rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py --global-batch 32 --steps 5 --warmup 3 --no-cpu-baseline --no-extras > $out/bench.log 2>&1)
  echo "== $t"; python3 scratch/gap_report.py $(find $out -name "*kernel_trace.csv" | head -1)
done
