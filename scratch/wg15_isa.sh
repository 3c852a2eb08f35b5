#!/bin/bash
# quick ISA check of the wgrad15 kernels: registers, spills, instruction mix
python /root/repo/scratch/slice_wg15.py && cd /tmp/st && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I/root/repo/include -I/root/repo/multipitch_architectures_amd/csrc -S --cuda-device-only -o wg15g.s wg15g.hip 2>&1 | grep -i "error" -A5 | head -20
grep "^\s*\.\(vgpr_count\|vgpr_spill_count\):" wg15g.s | tr '\n' ' '; echo
for k in 1 2; do awk "/^_ZN12_GLOBAL__N_120conv_wgrad15g_kernelILi${k}ELb1ELb1EEEvNS_10Wg15ParamsE:/,/s_endpgm/" wg15g.s > g$k.s; echo "NBC=$k lines $(wc -l < g$k.s) mfma $(grep -c v_mfma g$k.s) ds_read $(grep -c ds_read g$k.s) vmov $(grep -c 'v_mov_b32' g$k.s) scratch $(grep -c scratch_ g$k.s)"; done
