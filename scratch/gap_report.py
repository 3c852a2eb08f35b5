"""inter-kernel gaps of the last three steps of a rocprofv3 kernel trace of bench.py (steps end with u64_add_kernel)"""
import csv, re, sys, collections
R = list(csv.DictReader(open(sys.argv[1])))
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '').split('(')[0][:46]) for r in R)
ends = [i for i, r in enumerate(rows) if r[2].startswith('u64_add_kernel')]
for si in range(len(ends) - 3, len(ends)):
    seg = rows[ends[si - 1] + 1: ends[si] + 1]
    span = (seg[-1][1] - seg[0][0]) / 1e6; busy = sum(e - s for s, e, _ in seg) / 1e6
    gaps = [(seg[j + 1][0] - seg[j][1], seg[j][2], seg[j + 1][2]) for j in range(len(seg) - 1)]
    print(f"step {si}: {len(seg)} kernels, span {span:.3f} ms, busy {busy:.3f} ms, gaps {sum(max(g[0], 0) for g in gaps) / 1e6:.3f} ms")
    for g in sorted(gaps, reverse=True)[:6]:
        print(f"    gap {g[0] / 1e3:7.1f} us  {g[1]} -> {g[2]}")
