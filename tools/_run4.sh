mkdir -p gpurun_out/r4d
export PD_AB_ABLATION=1
L=posediffusion_amd/lib/libpd_engine.so
timeout 1200 python tools/ab_ggs.py $L gpurun_ab/libpd_abl1.so gpurun_ab/libpd_abl2.so gpurun_ab/libpd_abl4.so gpurun_ab/libpd_abl8.so gpurun_ab/libpd_abl16.so gpurun_ab/libpd_abl32.so gpurun_ab/libpd_abl63.so > gpurun_out/r4d/ablation.txt 2>&1
grep -v "Warning\|return nn\|amdgpu.ids" gpurun_out/r4d/ablation.txt
