mkdir -p gpurun_out/r4e
timeout 1200 python tools/ab_ggs.py gpurun_ab/libpd_wavevgpr.so posediffusion_amd/lib/libpd_engine.so > gpurun_out/r4e/ab_wave.txt 2>&1
grep -v "Warning\|return nn\|amdgpu.ids" gpurun_out/r4e/ab_wave.txt
