mkdir -p gpurun_out/r4g
export PD_AB_SHAPES="256,1,0;256,1,0;64,1,0"
timeout 1500 python tools/ab_ggs.py posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_glds_nt.so gpurun_ab/libpd_glds_sc1.so gpurun_ab/libpd_glds_sc0sc1.so > gpurun_out/r4g/ab_policy.txt 2>&1
grep -v "Warning\|return nn\|amdgpu.ids" gpurun_out/r4g/ab_policy.txt
