"""launches of the 15x15 templates grouped by grid size from a rocprofv3 kernel trace (the per-name averages of
*_kernel_stats.csv mix the layers an instantiation serves): python scratch/roofline_launches.py <kernel_trace.csv> [skip]
skip = launches of each group to leave out from the front (the initialising / capturing steps)."""
import collections, csv, re, sys
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", k)
    if not re.match(r"conv_(fwd_kernel<\d+, \d+, 15|wgrad15g_kernel)", k): continue
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    rows[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
print(f"{'kernel':46s} {'grid threads':>12s} {'launches':>8s} {'avg ms':>8s} {'min ms':>8s} {'max ms':>8s}")
for (k, g), v in sorted(rows.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    v = v[skip:] if len(v) > skip else v
    print(f"{k:46s} {g:12d} {len(v):8d} {sum(v)/len(v):8.3f} {min(v):8.3f} {max(v):8.3f}")
