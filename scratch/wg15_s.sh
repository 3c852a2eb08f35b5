#!/bin/bash
# 15x15 backward-weight with the slice count forced (MPA_WG15_S), layers given as indices of scratch/fwd_force.py
cd /root/repo
for S in $3; do
  echo "== S=$S"; MPA_WG15_S=$S python3 scratch/fwd_force.py $1 $2 2>/dev/null | sed 's/fwd.*wgrad/wgrad/' | grep -v planner
done
