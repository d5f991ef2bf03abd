#!/bin/bash
# HBM-side traffic per launch: two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots") over tools/pmc_target.py (the bench default engine batch: 256 sequences, one GGS workgroup each), then tools/pmc_summary.py.
# usage (on the GPU box): tools/collect_pmc.sh [sequences = 256]   -> gpurun_out/pmc_summary.json
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
CMD="python tools/pmc_target.py ${1:-256} 1"   # one batch of the default size, one GGS workgroup per sequence, no graphs (rocprofv3 attributes counters to the dispatches it sees); bench.py itself crashed inside rocprofv3 --pmc at this batch size
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- $CMD > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE "$CMD" > gpurun_out/pmc_summary.json
head -c 600 gpurun_out/pmc_summary.json
f=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); head -3 "$f"
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
