"""Summarise the three rocprofv3 --pmc passes of scratch/pmc_passes.sh into profiles/:
   <out>_pmc_sq_summary.csv   per kernel: dispatches, mean counters, MFMA-pipe busy fraction, wave-state shares
   <out>_pmc_hbm_summary.csv  per kernel: mean FETCH_SIZE / WRITE_SIZE (KB) per dispatch, corrected traffic
   <out>_pmc_traffic.json     the roofline kernel of bench.py (largest-grid dispatch of the 16->128 forward)
usage: python scratch/pmc_summary.py <tag> <out prefix, e.g. profiles/r02> [kernel name prefix of the roofline kernel]
(run where gpurun_out/pmc_<tag>_* exist).  The traffic file is stamped with the commit of the tree the passes ran on
(MPA_COMMIT in the environment, else `git rev-parse HEAD`)."""
import collections, csv, glob, json, os, re, subprocess, sys
tag, out = sys.argv[1], sys.argv[2]
roof_prefix = sys.argv[3] if len(sys.argv) > 3 else "conv_fwd_kernel<2, 12, 15"
try:
    commit = os.environ.get("MPA_COMMIT") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                                            cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:       # noqa: BLE001
    commit = None
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")

def load(part):
    f = glob.glob(os.path.join(root, f"pmc_{tag}_{part}", "*counter_collection.csv"))[0]
    per = collections.defaultdict(dict)        # dispatch id -> {kernel, grid, counters}
    for r in csv.DictReader(open(f)):
        d = per[int(r["Dispatch_Id"])]
        d["kernel"], d["grid"] = r["Kernel_Name"], int(r["Grid_Size"])
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return per

def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", k)[:90]

sq = load("sq")
names = ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
agg = collections.defaultdict(lambda: [0, collections.Counter()])
for d in sq.values():
    a = agg[(short(d["kernel"]), d["grid"])]
    a[0] += 1
    for n in names:
        a[1][n] += d.get(n, 0.0)
rows = []
for (k, grid), (n, c) in agg.items():
    m = {x: c[x] / n for x in names}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0                              # summed over the 8 XCDs by rocprofv3
    busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0) if cyc else 0.0     # 256 CUs x 4 SIMDs
    wc = m["SQ_WAVE_CYCLES"] or 1.0
    rows.append((m["GRBM_GUI_ACTIVE"] * n, k, grid, n, m, busy, m["SQ_WAIT_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc,
                 m["SQ_ACTIVE_INST_ANY"] / wc, m["SQ_LDS_BANK_CONFLICT"] / (m["SQ_LDS_IDX_ACTIVE"] or 1.0)))
rows.sort(reverse=True)
with open(out + "_pmc_sq_summary.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "dispatches"] + [f"mean_{n}" for n in names] +
               ["mfma_busy_frac = MFMA_BUSY/(GUI_ACTIVE/8*1024 SIMDs)", "wait_any_share", "wait_inst_share", "active_inst_share",
                "lds_bank_conflict_per_idx_active"])
    for _, k, grid, n, m, busy, wa, wi, ai, bc in rows:
        w.writerow([k, grid, n] + [f"{m[x]:.6g}" for x in names] + [f"{busy:.4f}", f"{wa:.3f}", f"{wi:.3f}", f"{ai:.3f}", f"{bc:.4f}"])
hb = {}
for part, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    a = collections.defaultdict(lambda: [0, 0.0])
    for d in load(part).values():
        e = a[(short(d["kernel"]), d["grid"])]
        e[0] += 1
        e[1] += d.get(ctr, 0.0)
    hb[ctr] = a
keys = sorted(set(hb["FETCH_SIZE"]) | set(hb["WRITE_SIZE"]),
              key=lambda k: -(hb["FETCH_SIZE"].get(k, [1, 0])[1] + hb["WRITE_SIZE"].get(k, [1, 0])[1]))
with open(out + "_pmc_hbm_summary.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "dispatches", "mean_FETCH_SIZE_KB", "mean_WRITE_SIZE_KB",
                "traffic_bytes_per_dispatch = 2*FETCH_KB*1024 + WRITE_KB*1024 (gfx950: FETCH_SIZE counts half of 16-B/lane reads)"])
    for k in keys:
        fn, fs = hb["FETCH_SIZE"].get(k, [0, 0.0]); wn, ws = hb["WRITE_SIZE"].get(k, [0, 0.0])
        fm, wm = (fs / fn if fn else 0.0), (ws / wn if wn else 0.0)
        w.writerow([k[0], k[1], max(fn, wn), f"{fm:.1f}", f"{wm:.1f}", f"{2 * fm * 1024 + wm * 1024:.0f}"])
# the roofline kernel: 16->128 15x15 forward = the conv_fwd_kernel<2, 12, 15...> dispatches with the largest grid
cand = [k for k in keys if k[0].startswith(roof_prefix)]
if cand:
    k = max(cand, key=lambda k: k[1])
    fn, fs = hb["FETCH_SIZE"][k]; wn, ws = hb["WRITE_SIZE"][k]
    fm, wm = fs / fn, ws / wn
    sqk = [r for r in rows if r[1] == k[0] and r[2] == k[1]]
    json.dump({"kernel": k[0], "grid_size": k[1], "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm,
               "fetch_correction": "x2 (gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streaming reads; "
                                   "MI355X_MICROARCH.md 'HBM')",
               "traffic_bytes": 2 * fm * 1024 + wm * 1024,
               "mfma_busy": sqk[0][5] if sqk else None,
               "lds_bank_conflict_per_idx_active": sqk[0][9] if sqk else None,
               "commit": commit,
               "source": f"gpurun_out/pmc_{tag}_fetch, pmc_{tag}_write, pmc_{tag}_sq (separate --pmc passes with --kernel-trace "
                         f"only; mean over {fn} dispatches)"}, open(out + "_pmc_traffic.json", "w"), indent=1)
print(open(out + "_pmc_sq_summary.csv").read()[:3000])
