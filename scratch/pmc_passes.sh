#!/bin/bash
# three rocprofv3 counter passes over the same bench command (kernel-by-kernel launches, so every dispatch is attributed):
#   sq    : MFMA-pipe / wave-state / LDS counters   fetch : FETCH_SIZE   write : WRITE_SIZE
# usage: pmc_passes.sh <tag> <bench args...>; outputs gpurun_out/pmc_<tag>_{sq,fetch,write}/
cd /root/repo; export TMPDIR=/tmp
tag=$1; shift
bash tools_pmc.sh ${tag}_sq "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@" --no-graph --no-extras --no-cpu-baseline > /dev/null 2>&1; echo sq rc=$?
bash tools_pmc.sh ${tag}_fetch "FETCH_SIZE" "$@" --no-graph --no-extras --no-cpu-baseline > /dev/null 2>&1; echo fetch rc=$?
bash tools_pmc.sh ${tag}_write "WRITE_SIZE" "$@" --no-graph --no-extras --no-cpu-baseline > /dev/null 2>&1; echo write rc=$?
ls gpurun_out/pmc_${tag}_sq | head
