"""per-step time of the top kernels of a rocprofv3 kernel_stats.csv: python scratch/stats_top.py file.csv steps [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total per step {tot / steps / 1e6:.2f} ms over {len(rows)} kernels")
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms {int(r['Calls']) / steps:6.1f}  {name[:100]}")
