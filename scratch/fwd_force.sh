#!/bin/bash
cd /root/repo
for f in "" "1,12" "1,6" "2,12" "2,6" "1,8" "2,8" "1,4" "2,4" "4,6" "4,4"; do
  if [ -z "$f" ]; then python3 scratch/fwd_force.py $1 $2 2>/dev/null; else MPA_FWD_FORCE=$f python3 scratch/fwd_force.py $1 $2 2>/dev/null; fi
done
