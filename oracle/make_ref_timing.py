"""Timing fixture of the REFERENCE FILES EXECUTED IN PLACE next to the oracle port, on one host (test infrastructure, like everything
under oracle/):  python -m oracle.make_ref_timing  ->  tests/golden/ref_timing.json

bench.py's `cpu_baseline` runs the reference's own files (`kind: "reference"`) only where /root/reference exists; the GPU box never has
it and reports the oracle port (`kind: "port"`).  This script measures BOTH on the same host with the same sample and thread count
(bench.cpu_baseline itself, the port leg with the reference hidden), so that a "port" figure from another box can be read against the
reference: the two run the same torch-CPU operator sequence and differ by the ratio recorded here.
Follows models/denoiser.py:53-98 (one forward at B = 1, N = 20) and util/geometry_guided_sampling.py:67-126 (GGS_optimize iterations
at M = 57 000 for the four stage types)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(budget_s=24.0):
    import torch
    import bench
    from oracle import ref_stubs as RS
    if not RS.available():
        raise SystemExit("the reference tree is not mounted (PD_REFERENCE_ROOT or /root/reference)")
    out = {"host_cores": os.cpu_count(), "torch": torch.__version__, "budget_s_per_leg": budget_s, "legs": {}}
    real_available = RS.available
    for kind in ("reference", "port"):
        RS.available = real_available if kind == "reference" else (lambda: False)
        r = bench.cpu_baseline(budget_s)
        assert r["kind"] == kind, r
        ms = {k: float(v) for k, v in re.findall(r"(all|fl|r|t) x\d+: ([0-9.]+) ms/it", r["sample"])}
        den = float(re.search(r"\(B=1, N=20\): ([0-9.]+) ms/step", r["sample"]).group(1))
        out["legs"][kind] = {"threads": r["cores"], "denoiser_ms_per_step": den, "ggs_ms_per_iteration": ms, "sequences_per_s": r["value"],
                             "sample": r["sample"]}
    RS.available = real_available
    a, b = out["legs"]["reference"], out["legs"]["port"]
    out["port_over_reference"] = {"denoiser": b["denoiser_ms_per_step"] / a["denoiser_ms_per_step"],
                                  "ggs_all": b["ggs_ms_per_iteration"]["all"] / a["ggs_ms_per_iteration"]["all"],
                                  "sequences_per_s": b["sequences_per_s"] / a["sequences_per_s"]}
    path = os.path.join(ROOT, "tests", "golden", "ref_timing.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["port_over_reference"]), "->", path)


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 24.0)
