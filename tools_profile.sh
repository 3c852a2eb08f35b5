#!/bin/bash
# usage: tools_profile.sh <tag> <bench args...>   -- rocprofv3 kernel trace + stats of one bench.py run
set -e
tag=$1; shift
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py "$@" > $out/bench.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
tail -2 $out/bench.log
