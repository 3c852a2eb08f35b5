"""scan ISA listings for serialised stores: a vector-memory store, then `s_waitcnt vmcnt(0)` with no load issued in between,
then another store -- the wave waits for the first store to be acknowledged (one in-order counter for loads and stores on
gfx9), typically because a load issued before the first store is first used inside a divergent block.
usage: python scratch/store_wait_scan.py file.s ..."""
import re, subprocess, sys, collections
for path in sys.argv[1:]:
    fn = None; state = 0; hits = collections.Counter(); stores = collections.Counter()
    for ln in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", ln)
        if m: fn = m.group(1); state = 0; continue
        t = ln.strip()
        if re.match(r"(global|buffer|flat|scratch)_(store|atomic)", t):
            stores[fn] += 1
            if state == 2: hits[fn] += 1
            state = 1
        elif re.match(r"(global|buffer|flat|scratch)_load", t): state = 0
        elif state == 1 and re.match(r"s_waitcnt.*vmcnt\(0\)", t): state = 2
        elif t.startswith("s_endpgm"): state = 0
    for f, n in hits.most_common():
        name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"\(.*$", "", name)
        print(f"{n:4d} / {stores[f]:4d} stores  {name[:110]}")
