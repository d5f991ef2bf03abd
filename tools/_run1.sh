mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r4a/pytest_r4.txt
timeout 120 python tools/ggs_prof.py 1 0 0 0,3,7 > gpurun_out/r4a/prof_b1.txt 2>&1
timeout 120 python tools/ggs_prof.py 8 0 0 0,7 > gpurun_out/r4a/prof_b8.txt 2>&1
timeout 200 python tools/ggs_prof.py 64 1 0 0,2,7 > gpurun_out/r4a/prof_b64.txt 2>&1
cat gpurun_out/r4a/pytest_r4.txt | tail -40; cat gpurun_out/r4a/prof_b1.txt gpurun_out/r4a/prof_b8.txt gpurun_out/r4a/prof_b64.txt
