echo even; timeout -k 10 100 python scratch/wg15_time.py 2>&1 | grep -v amdgpu.ids
echo no-unroll; MPA_WG15_NOUNROLL=1 timeout -k 10 100 python scratch/wg15_time.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3
