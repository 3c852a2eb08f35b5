"""per-kernel MFMA-pipe / wave-state summary of one rocprofv3 --pmc pass: python scratch/pmc_quick.py gpurun_out/<dir>"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d = per[int(r["Dispatch_Id"])]
    d["kernel"], d["grid"] = r["Kernel_Name"], int(r["Grid_Size"])
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(lambda: [0, collections.Counter()])
for d in per.values():
    k = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", d["kernel"].replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
    a = agg[(k, d["grid"])]; a[0] += 1
    for n, v in d.items():
        if n not in ("kernel", "grid"): a[1][n] += v
for (k, g), (n, c) in sorted(agg.items(), key=lambda kv: -kv[1][1]["GRBM_GUI_ACTIVE"]):
    cyc = c["GRBM_GUI_ACTIVE"] / n / 8.0; wc = c["SQ_WAVE_CYCLES"] or 1.0
    print(f"{k:60s} grid {g:8d} x{n:3d} cyc {cyc:9.0f} mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/n/(cyc*1024):.3f} "
          f"wait_any {c['SQ_WAIT_ANY']/wc:.2f} wait_inst {c['SQ_WAIT_INST_ANY']/wc:.2f} active {c['SQ_ACTIVE_INST_ANY']/wc:.2f} "
          f"lds_conf {c['SQ_LDS_BANK_CONFLICT']/(c['SQ_LDS_IDX_ACTIVE'] or 1):.3f} waves/simd {c['SQ_WAVE_CYCLES']/n/(c['SQ_BUSY_CYCLES']/n or 1):.2f}")
