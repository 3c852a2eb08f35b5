#!/bin/bash
cd /root/repo
echo new; python3 scratch/transpose_time.py 256 2>/dev/null
cp multipitch_architectures_amd/csrc/libmpa_hip.so /tmp/libnew.so
cp multipitch_architectures_amd/csrc/libbase.keep multipitch_architectures_amd/csrc/libmpa_hip.so
echo base; python3 scratch/transpose_time.py 256 2>/dev/null
cp /tmp/libnew.so multipitch_architectures_amd/csrc/libmpa_hip.so
