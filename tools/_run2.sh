mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r4b/pytest.txt
timeout 120 python tools/ggs_prof.py 1 0 0 0,3,7 > gpurun_out/r4b/prof_b1.txt 2>&1
timeout 200 python tools/ggs_prof.py 64 1 0 0,2,7 > gpurun_out/r4b/prof_b64.txt 2>&1
timeout 200 python tools/ggs_prof.py 64 1 4 0,2,7 > gpurun_out/r4b/prof_b64_w8.txt 2>&1
tail -15 gpurun_out/r4b/pytest.txt; grep -h "launch\|wave" gpurun_out/r4b/prof_b1.txt gpurun_out/r4b/prof_b64.txt gpurun_out/r4b/prof_b64_w8.txt
