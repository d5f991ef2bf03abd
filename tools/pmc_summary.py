"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel traffic per dispatch.
usage: python tools/pmc_summary.py <dir FETCH_SIZE pass> <dir WRITE_SIZE pass> "<profiled command>"  > summary.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): rocprofv3 reports the two derived
counters in KB per dispatch; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read stream,
so it is doubled in the *_corrected figures; the counters sit on the fabric side of the L2, i.e. Infinity-Cache hits are
included: this is L2-miss traffic, an upper bound on HBM bytes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(d, counter):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return per


def main():
    fdir, wdir, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, write = read(fdir, "FETCH_SIZE"), read(wdir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("pd_") and "pd_" not in name:
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        kernels[name] = {
            "FETCH_SIZE": {"dispatches": len(f), "avg_KB": fa, "min_KB": min(f) if f else 0.0, "max_KB": max(f) if f else 0.0},
            "WRITE_SIZE": {"dispatches": len(w), "avg_KB": wa, "min_KB": min(w) if w else 0.0, "max_KB": max(w) if w else 0.0},
            "traffic_bytes_per_dispatch_raw": (fa + wa) * 1024.0,
            "traffic_bytes_per_dispatch_corrected": (2.0 * fa + wa) * 1024.0,
        }

    def corrected(prefix):
        return sum(v["traffic_bytes_per_dispatch_corrected"] for k, v in kernels.items() if prefix in k)

    # one denoiser step, launches per kind (kinds that did not run in the profiled process contribute nothing; a kind's
    # figure is the average over its dispatches, so kinds that serve layers of different sizes are weighted by their count).
    # < 1024 token rows: 34 tile-GEMM launches of 7 kinds + 8 attention + 1 tail
    per_step = {"pd_gemm_kernel<704": 1, "pd_gemm_kernel<512, 1, 0": 8, "pd_gemm_kernel<512, 0, 2": 8,
                "pd_gemm_kernel<512, 1, 1": 8, "pd_gemm_kernel<1024, 0, 2": 8, "pd_gemm_kernel<512, 0, 0": 1,
                "pd_attn_kernel": 8, "pd_tail_kernel": 1,
                # >= 1024 token rows, common: the input-row kernel of _first, _first + _last.0 on the exact LDS-DMA GEMM
                "pd_embed_rows_kernel": 1, "pd_gemm_dma_kernel<0, false": 2,
                # default mode there (fp16-plane encoder GEMMs): LayerNorm rows x 16, QKV x 8, out-projection + FF2 x 16, FF1 x 8, attention x 8
                "pd_ln_rows_kernel<512, 2>": 16, "pd_gemm_strip_kernel<0, 2, true": 8, "pd_gemm_strip_kernel<2, 2, true": 16,
                "pd_gemm_strip_kernel<4, 2, true": 8, "pd_attn_seq_kernel<2>": 8, "pd_attn_mma_kernel<2>": 8,
                # round 5 / 6 (MISSING from this table until the end of round 6 -- the step figure of rounds 5 and 6 up to then left out the fused attention kernel
                # and the 96-row strip GEMMs, i.e. most of the step): in_proj + attention in one kernel x 8 (then no <0, 2, true> / pd_attn_* launches), out-projection +
                # FF2 on 96-row tiles x 16 where the launch takes them (then no <2, 2, true> launches), _first's step piece on the LDS-DMA kernel (EPI 4)
                "pd_qkv_attn_kernel": 8, "pd_gemm_strip_kernel<2, 3, true": 16, "pd_gemm_strip_kernel<4, 3, true": 8, "pd_gemm_dma_kernel<4, false": 1,
                # exact mode (PD_OPT_DENOISER_SPLIT = 0): statistics x 16, QKV / FF1 with LayerNorm at the fragment reads x 8 each, out-projection + FF2 x 16
                "pd_ln_stats_kernel": 16, "pd_gemm_dma_kernel<0, true": 8, "pd_gemm_dma_kernel<1, true": 8, "pd_gemm_dma_kernel<2, false": 16,
                "pd_attn_seq_kernel<0>": 8}
    den = 0.0
    names = [k.replace("void ", "") for k in kernels]
    for a, b in (("pd_gemm_strip_kernel<2, 2, true", "pd_gemm_strip_kernel<2, 3, true"), ("pd_gemm_strip_kernel<4, 2, true", "pd_gemm_strip_kernel<4, 3, true"),
                 ("pd_gemm_strip_kernel<0, 2, true", "pd_qkv_attn_kernel")):
        if any(n.startswith(a) for n in names) and any(n.startswith(b) for n in names):
            sys.exit(f"pmc_summary: both {a} and {b} ran in the profiled process -- alternatives for the same launches of a step; profile one batch size")
    for k, v in kernels.items():
        for pre, n in per_step.items():
            if k.replace("void ", "").startswith(pre):
                den += n * v["traffic_bytes_per_dispatch_corrected"]
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "posediffusion_amd", "lib", "libpd_engine.so")
    with open(lib, "rb") as fh:
        lib_hash = hashlib.sha256(fh.read()).hexdigest()
    out = {
        "libpd_engine_sha256": lib_hash,   # bench.py quotes these figures only for the binary they were measured on
        "batch_sequences": int(cmd.split()[2]) if len(cmd.split()) > 2 and cmd.split()[2].isdigit() else None,   # ... and for this engine batch
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `{cmd}` "
                  "(one engine batch of the bench default size, N=20, GGS on, one GGS workgroup per sequence), 1x MI355X; tools/collect_pmc.sh + tools/pmc_summary.py",
        "units": "KB per dispatch as reported by rocprofv3 (FETCH_SIZE / WRITE_SIZE); bytes = KB * 1024",
        "gfx950_correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) "
                             "coalesced read stream on gfx950 -> doubled in *_corrected; Infinity-Cache hits are counted "
                             "(fabric-side counter), so this is L2-miss traffic, an upper bound on HBM bytes",
        "kernels": kernels,
        "ggs_launch": {"traffic_bytes_corrected": corrected("pd_ggs_lane_kernel") or corrected("pd_ggs_kernel"),
                       "kernel": "pd_ggs_lane_kernel" if corrected("pd_ggs_lane_kernel") else "pd_ggs_kernel"},
        "denoiser_step": {"traffic_bytes_corrected": den},
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
