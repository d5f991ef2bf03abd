"""Register / scratch budget of the GGS kernel variants, read from hipcc's own resource remarks (cross-compiled for gfx950, no GPU
needed).  The design rests on these numbers: the 8-wave variants must hold two waves per SIMD without touching scratch, the
12-wave variants must fit three waves per SIMD (168 VGPRs) with the match pass free of spills -- what little is spilled are launch
constants of the serial phases."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "posediffusion_amd", "csrc", "pd_ggs.hip")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_ggs_kernel_variants_keep_their_register_budget(tmp_path):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage",
                          "-c", SRC, "-o", str(tmp_path / "pd_ggs.o")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    ggs = {k: v for k, v in kernels.items() if "pd_ggs_kernel" in k}
    assert len(ggs) == 8, sorted(kernels)                       # <0,true,8> <0,false,8> <3|5|6,false,8> <3|5|6,false,12>
    for name, r in ggs.items():
        twelve = "ELi12EE" in name
        if twelve:
            assert r["VGPRs"] <= 168 and r["Occupancy"] == 3, (name, r)
            assert r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (name, r)       # (round 4: built with -fno-slp-vectorize, nothing spills)
        else:
            assert r["VGPRs"] <= 256 and r["Occupancy"] >= 2, (name, r)
            assert r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (name, r)
    # the lane-per-item kernel: 8 waves of up to 256 registers (two per SIMD), fourteen steps of every lane item resident in
    # them for the whole launch and NOTHING spilled (round 4: its pass is bound by instruction issue; 16 resident steps spill 26 registers and
    # cost 4 % of a launch, profiles/round4_lane_ring.txt)
    lane = {k: v for k, v in kernels.items() if "pd_ggs_lane_kernel" in k}
    assert len(lane) == 1, sorted(kernels)
    for name, r in lane.items():
        assert r["VGPRs"] <= 256 and r["Occupancy"] == 2 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (name, r)
    two_hop = [v for k, v in kernels.items() if "pd_ggs2_kernel" in k]
    assert two_hop and two_hop[0]["VGPRs Spill"] == 0 and two_hop[0]["ScratchSize"] == 0


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_denoiser_kernels_keep_their_register_budget(tmp_path):
    """The large-batch denoiser's GEMM kernels: the strip kernel of the fp16-plane mode lives on occupancy (>= 4 waves per
    SIMD at the 64 x 128 tile), the LDS-DMA exact kernel on >= 6; nothing touches scratch."""
    src = os.path.join(ROOT, "posediffusion_amd", "csrc", "pd_denoiser.hip")
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Rpass-analysis=kernel-resource-usage",
                          "-c", src, "-o", str(tmp_path / "pd_denoiser.o")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split("[")[0].strip()] = int(m.group(2))
    strip = {k: v for k, v in kernels.items() if "pd_gemm_strip_kernel" in k}
    # EPI 0 / 2 / 4 at 64-row tiles (4 waves per SIMD: several workgroups share a CU there) + (round 6) EPI 2 / 4 at 96-row tiles (one workgroup per
    # CU by construction: 48 accumulator registers more, 3 waves per SIMD suffice); fp16 planes; nothing spilled
    assert len(strip) == 5, sorted(kernels)
    # (round 6) the EPI 2 variants also hold the residual tile, requested a K chunk before the epilogue needs it: 4 RT more 16-byte registers, one wave
    # per SIMD less -- their launches are one workgroup per CU at the bench's shapes (96-row tiles: 216 workgroups; 64-row tiles at 2 060 rows: 132)
    for name, r in strip.items():
        rt3, epi2 = "ELi3ELb1E" in name, "kernelILi2E" in name
        assert r["Occupancy"] >= (3 if rt3 else 4) - (1 if epi2 else 0) and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (name, r)
    dma = {k: v for k, v in kernels.items() if "pd_gemm_dma_kernel" in k}
    assert len(dma) == 5, sorted(kernels)                       # EPI 0, 0 + LN, 1 + LN, 2, and (round 5) 4: _first's step piece + the hoisted z piece
    for name, r in dma.items():
        assert r["Occupancy"] >= (5 if "ILi4E" in name else 6) and r["ScratchSize"] == 0, (name, r)    # (EPI 4 carries one more pointer: 5 waves, K = 192 only)
    # round 5: in_proj + attention in one workgroup of 12 waves (three per SIMD: <= 168 registers), Q / K / V in LDS only -- nothing spilled
    fused = {k: v for k, v in kernels.items() if "pd_qkv_attn_kernel" in k}
    assert len(fused) == 1, sorted(kernels)                     # (the BARE > 0 variants exist in -DPD_DEV_KNOBS builds only)
    for name, r in fused.items():
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 168 and r["Occupancy"] >= 3 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (name, r)
