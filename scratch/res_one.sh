#!/bin/bash
# usage: scratch/res_one.sh <file.hip> <kernel-name-regex> [extra flags]   -- VGPR / AGPR / occupancy / spills of the matching kernels
f=$1; pat=$2; shift 2
cd /root/repo/multipitch_architectures_amd/csrc
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form "$@" \
  -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | python3 -c "
import re,sys,subprocess
cur=None;rows=[]
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur={'n':m.group(1)}; rows.append(cur); continue
    for k,p in (('v',r' VGPRs: (\d+)'),('a',r'AGPRs: (\d+)'),('o',r'Occupancy \[waves/SIMD\]: (\d+)'),('s',r'VGPRs Spill: (\d+)'),('sg',r'TotalSGPRs: (\d+)')):
        m=re.search(p,ln)
        if m and cur is not None: cur[k]=int(m.group(1))
seen=set()
for r in rows:
    if r['n'] in seen: continue
    seen.add(r['n'])
    d=subprocess.run(['c++filt',r['n']],capture_output=True,text=True).stdout.strip()
    d=re.sub(r'\(anonymous namespace\)::','',d); d=re.sub(r'\(.*$','',d)
    if re.search(r'''$pat''',d): print(f\"{r.get('v',0):4d} {r.get('a',0):4d} sg{r.get('sg',0):4d} occ{r.get('o',0)} spill{r.get('s',0)}  {d}\")
"
