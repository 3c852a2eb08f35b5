"""scan ISA listings for loads that are waited for right where they are issued: a vector-memory load followed, within a few
instructions and before any MFMA, by `s_waitcnt vmcnt(0)` inside a loop -- a prefetch that is not one (typically a select on
the loaded value, or a load first used in the block it is issued in).  usage: python scratch/load_wait_scan.py file.s ..."""
import re, subprocess, sys, collections
for path in sys.argv[1:]:
    fn = None; hits = collections.Counter(); loads = collections.Counter(); dist = None; inloop = False
    for ln in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", ln)
        if m: fn = m.group(1); dist = None; inloop = False; continue
        t = ln.strip()
        if "in Loop:" in ln or "Loop Header" in ln: inloop = True
        if re.match(r"(global|buffer|flat)_load", t) and " lds" not in t:
            if inloop: loads[fn] += 1; dist = 0
        elif dist is not None and t and not t.startswith((";", ".")):
            if t.startswith("v_mfma"): dist = None
            elif re.match(r"s_waitcnt.*vmcnt\(0\)", t):
                hits[fn] += 1; dist = None
            else:
                dist += 1
                if dist > 12: dist = None
        if t.startswith("s_endpgm"): inloop = False
    for f, n in hits.most_common():
        name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"\(.*$", "", name)
        print(f"{n:4d} / {loads[f]:4d} loads in loops  {name[:110]}")
