#!/bin/bash
# A variant of libpd_engine.so for same-box A / B runs (tools/ab_ggs.py, PD_ENGINE_LIB): tools/build_variant.sh <name> "<-D flags>"
# -> gpurun_ab/libpd_<name>.so (git-ignored; travels to the GPU box with gpurun).  Objects go to a build directory of their own.
set -e
cd "$(dirname "$0")/../posediffusion_amd/csrc"
name=$1; flags=$2; B=build_$name; mkdir -p $B ../../gpurun_ab
C="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags"
for f in pd_engine pd_denoiser pd_ggs_ingest pd_metrics pd_vit; do hipcc $C -ffp-contract=fast -c $f.hip -o $B/$f.o & done
hipcc $C -ffp-contract=on -fno-slp-vectorize -c pd_ggs.hip -o $B/pd_ggs.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o ../../gpurun_ab/libpd_$name.so
rm -rf $B; ls -la ../../gpurun_ab/libpd_$name.so
