"""profiles/r03_kernel_resources.txt: VGPR / AGPR / SGPR / occupancy / spills of every kernel of the library, from
hipcc -Rpass-analysis=kernel-resource-usage with the build flags of multipitch_architectures_amd/build.py.  A diff of the table
after an edit shows register / occupancy changes of kernels the edit did not touch (the stream-K episode of round 2)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multipitch_architectures_amd.build import CSRC, FLAGS, SOURCES
rows = []
for src in SOURCES:
    r = subprocess.run(["hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/dev/null"],
                       capture_output=True, text=True)
    cur = None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1), "src": src}
            rows.append(cur)
            continue
        for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("spill", r"VGPRs Spill: (\d+)")):
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
seen = set()
rows = [r for r in rows if not (r["name"] in seen or seen.add(r["name"]))]
dem = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
out = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r03_kernel_resources.txt")
with open(out, "w") as f:
    f.write("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, flags: " + " ".join(FLAGS) + "\n")
    f.write("# regenerate: python scratch/kernel_resources.py ; a diff after an edit shows kernels whose registers moved\n")
    f.write(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'occ':>4} {'spill':>5} {'scratch':>7}  {'source':16s} kernel\n")
    for r, d in sorted(zip(rows, dem), key=lambda t: (t[0]["src"], t[1])):
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"\(.*$", "", d)
        f.write(f"{r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('occ', 0):4d} {r.get('spill', 0):5d} "
                f"{r.get('scratch', 0):7d}  {r['src']:16s} {d}\n")
print(out, len(rows))
