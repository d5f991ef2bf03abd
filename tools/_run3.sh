mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4c/pytest.txt
timeout 900 python tools/ab_ggs.py > gpurun_out/r4c/ab.txt 2>&1
tail -5 gpurun_out/r4c/pytest.txt; grep -v Warning gpurun_out/r4c/ab.txt | grep -v "return nn\|amdgpu.ids"
