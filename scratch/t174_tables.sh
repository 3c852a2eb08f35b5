#!/bin/bash
cd /root/repo
python scratch/layer_table.py 64 SAUnet:L 174 > gpurun_out/lt_t174_b64.txt 2>&1
head -30 gpurun_out/lt_t174_b64.txt
