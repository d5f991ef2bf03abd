cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r2b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_r2.py tests/test_demo_dropin.py -m gpu -q 2>&1 | tail -25 > $O/pytest_r2.txt; tail -12 $O/pytest_r2.txt
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity_r2.py --deselect tests/test_demo_dropin.py 2>&1 | tail -15 > $O/pytest_all.txt; tail -5 $O/pytest_all.txt
timeout 300 python tools/ggs_prof_k1.py 2>&1 | grep -v Warn | tee $O/ggs_prof_k1.txt
