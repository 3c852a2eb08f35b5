#!/bin/bash
# backward-data at a given batch with the channel-split count forced (MPA_FWD_KS_FORCE): does splitting the input
# channels over more workgroups beat the planner's choice when the tile count per CU is a bad fraction?
cd /root/repo
for ks in 0 2 4 8; do
  echo "== KS_FORCE=$ks"
  MPA_FWD_KS_FORCE=$ks python3 scratch/fwd_force.py $1 $2 2>/dev/null | sed 's/fwd *[0-9.]* ms *[0-9.]* *//; s/wgrad.*//'
done
