"""From one rank's rocprofv3 kernel + memory-copy traces: per timed step, the device<->host copies of the gradient buckets
(gloo's all-reduce of a HIP tensor = copy out, host reduction, copy back) against the kernels that ran meanwhile."""
import csv, glob, os, sys
d = sys.argv[1]
kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
mf = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)[0]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kf))]
M = list(csv.DictReader(open(mf)))
print("memory-copy columns:", list(M[0].keys()))
K.sort()
# steps: adamw_kernel marks the end of each step's update graph
ends = [e for s, e, n in K if "adamw_kernel" in n]
# (this rocprofv3 build reports no size: bucket copies are told from the small ones by their duration)
copies = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "").replace("MEMORY_COPY_", ""),
           int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in M]
copies.sort()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
for si in range(max(0, len(ends) - 3), len(ends)):
    t1 = ends[si]; t0 = ends[si - 1] if si > 0 else t1 - 100_000_000
    ks = [(s, e, n) for s, e, n in K if t0 < s <= t1]
    cs = [c for c in copies if t0 < c[0] <= t1 and c[3] >= 30_000]
    if not ks: continue
    base = ks[0][0]
    print(f"\nstep ending at kernel #{si}: {len(ks)} kernels over {(ks[-1][1] - base) / 1e6:.2f} ms, {len(cs)} copies of >= 30 us (gradient buckets)")
    for s, e, direction, size in cs:
        over = [(ks_, ke, n) for ks_, ke, n in ks if ks_ < e and ke > s]
        busy = sum(min(e, ke) - max(s, ks_) for ks_, ke, n in over)
        names = sorted({short(n) for _, _, n in over})
        print(f"  copy {direction:>16s} {size / 1e3:7.1f} us  t = {(s - base) / 1e6:7.3f} .. {(e - base) / 1e6:7.3f} ms;"
              f" kernels running meanwhile: {len(over)} ({100.0 * busy / max(e - s, 1):.0f} % of the copy) {names[:4]}")
