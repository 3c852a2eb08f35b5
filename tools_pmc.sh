#!/bin/bash
# usage: tools_pmc.sh <tag> "<counters>" <bench args...>  -- rocprofv3 PMC pass (kernel-trace only, as the pool requires)
set -e
tag=$1; shift
ctrs=$1; shift
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o $tag -- python3 bench.py "$@" > $out/bench.log 2>&1
ls $out
