#!/bin/bash
# rocprofv3 kernel trace of a data-parallel rank's step (world of one over RCCL): which kernels run on which queue, and where
# the collectives' kernels (if a one-rank RCCL launches any) sit relative to the graph segments
cd /root/repo; export TMPDIR=/tmp
out=gpurun_out/prof_dp; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o dp -- python3 bench.py --global-batch 32 --steps 4 --warmup 3 --no-cpu-baseline --no-extras --dp-rehearsal > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-300
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "kernel records; columns:", list(rows[0].keys()))
q = collections.Counter((r.get("Queue_Id"), r.get("Stream_Id")) for r in rows)
print("queues/streams:", q.most_common(8))
names = collections.Counter(r["Kernel_Name"][:60] for r in rows if "ccl" in r["Kernel_Name"].lower() or "Reduce" in r["Kernel_Name"] or "rccl" in r["Kernel_Name"].lower())
print("collective kernels:", names)
PY
