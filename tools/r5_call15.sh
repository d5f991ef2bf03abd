cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/demo
timeout 500 python -m pytest tests/test_demo_dropin.py -m gpu -q 2>&1 | tail -3 > gpurun_out/demo/pytest_demo.txt; cat gpurun_out/demo/pytest_demo.txt
python tools/make_synthetic_ckpt.py /tmp/synth.pth --cfg _ref_stage/cfgs/default.yaml > /dev/null 2>&1
(cd _ref_stage/pose_diffusion && PYTHONPATH=$R timeout 600 python -m posediffusion_amd.run_reference demo.py image_folder=samples/apple ckpt=/tmp/synth.pth GGS.enable=False 2>&1 | grep -v Warning | tail -8) > gpurun_out/demo/demo_ggs_off.log; tail -4 gpurun_out/demo/demo_ggs_off.log
