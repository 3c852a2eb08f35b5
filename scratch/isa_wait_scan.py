"""per kernel of a HIP source: number of global / buffer loads against number of `s_waitcnt vmcnt` -- a kernel whose waits
are about as many as its loads makes one trip to memory after the other (typical cause: a guarded load inside an unrolled
loop, which hipcc compiles to a branch and a wait per element)
usage: python scratch/isa_wait_scan.py pointwise norm pool_up attn ..."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in sys.argv[1:] or ["pointwise", "norm", "pool_up", "attn"]:
    asm = f"/tmp/{f}.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-ffp-contract=off", "-mllvm",
                    "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-o", asm,
                    os.path.join(root, "multipitch_architectures_amd", "csrc", f + ".hip")], check=True, capture_output=True)
    txt = open(asm).read()
    for m in re.finditer(r'^(_Z\w+):\s*;[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        loads = len(re.findall(r'global_load|buffer_load', body)); waits = len(re.findall(r's_waitcnt vmcnt', body))
        if loads >= 6 and waits >= 0.6 * loads:
            print(f"{f:9s} loads {loads:4d} waits {waits:4d}  {re.sub(r'_ZN12_GLOBAL__N_1[0-9]+', '', name)[:70]}")
