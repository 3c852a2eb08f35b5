echo "== global dY"; timeout -k 10 200 python scratch/wgrad_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
echo "== LDS dY"; MPA_WG_GA=0 timeout -k 10 200 python scratch/wgrad_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3
