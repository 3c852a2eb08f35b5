#!/bin/bash
# A/B of the current library against multipitch_architectures_amd/csrc/libbase.keep inside one gpurun call
cd /root/repo
for cfg in "256 SAUnet:L" "128 Unet:L" "256 BLUnet:XXL" "32 SAUnet:L"; do
  set -- $cfg
  python scratch/gemm_table.py $1 $2 2>&1 | grep -v "Warn\|amdgpu" | head -1 | sed "s/^/new $2 b$1: /"
done
cp multipitch_architectures_amd/csrc/libmpa_hip.so /tmp/libnew.so
cp multipitch_architectures_amd/csrc/libbase.keep multipitch_architectures_amd/csrc/libmpa_hip.so
for cfg in "256 SAUnet:L" "128 Unet:L" "256 BLUnet:XXL" "32 SAUnet:L"; do
  set -- $cfg
  python scratch/gemm_table.py $1 $2 2>&1 | grep -v "Warn\|amdgpu" | head -1 | sed "s/^/base $2 b$1: /"
done
cp /tmp/libnew.so multipitch_architectures_amd/csrc/libmpa_hip.so
