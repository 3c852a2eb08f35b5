#!/bin/bash
# head conv2 backward-weight: the planner's choice against forced variants (MPA_WG_GA, MPA_WG_VARIANT, MPA_WG_TXN)
cd /root/repo
run() { echo "== $*"; env "$@" python3 scratch/wgrad_layers.py 2>/dev/null | grep "s3:" ; }
run X=1
run MPA_WG_GA=force
run MPA_WG_GA=0
for v in 0 1 2 3 4; do run MPA_WG_VARIANT=$v; run MPA_WG_VARIANT=$v MPA_WG_GA=force; done
for t in 2 3 4 8; do run MPA_WG_TXN=$t; done
