mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_gpu_parity_r3.py -m gpu -x -q -k "lane" 2>&1 | tail -15 > gpurun_out/r4i/pytest_lane.txt
tail -12 gpurun_out/r4i/pytest_lane.txt
export PD_AB_SHAPES="256,1,0;256,1,8;64,1,0;64,1,8;256,1,8;256,1,0"
timeout 900 python tools/ab_ggs.py posediffusion_amd/lib/libpd_engine.so 2>&1 | grep "B=\|round\|Error\|error" | tee gpurun_out/r4i/ab_lane.txt
