mkdir -p gpurun_out/r4j
export PD_AB_SHAPES="256,1,8;64,1,8;256,1,8"
timeout 1500 python tools/ab_ggs.py gpurun_ab/libpd_lane_12_8.so gpurun_ab/libpd_lane_10_8.so gpurun_ab/libpd_lane_8_8.so gpurun_ab/libpd_lane_4_8.so 2>&1 | grep "B=\|round\|Error\|error" | tee gpurun_out/r4j/ab_lane_variants.txt
