#!/bin/bash
# usage: gap_one.sh <tag> <bench args...>: kernel trace of a bench run + gap report
cd /root/repo; export TMPDIR=/tmp
tag=$1; shift
out=/root/repo/gpurun_out/gap_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/bench.log 2>&1
python3 scratch/gap_report.py $(find $out -name "*kernel_trace.csv" | head -1)
