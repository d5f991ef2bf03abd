# ablations of the persistent small-batch denoiser kernel (csrc/pd_den_small.inc): PD_SMALL_DBG bits -- 1 no LayerNorm statistics, 2 no weight
# stream, 4 no bias / residual loads, 8 relaxed barrier arrival, 16 no activation loads, 32 no attention; 63 all (barriers + MFMA + stores only)
for d in 0 1 2 4 8 16 32 63; do echo "PD_SMALL_DBG=$d"; PD_SMALL_DBG=$d python tools/den_small.py 1 2 2>&1 | grep -E "^B=1, one|layers 1..7"; done
