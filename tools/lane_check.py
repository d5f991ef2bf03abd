"""Development check of the lane-per-item GGS kernel (pd_ggs_lane_kernel) on the GPU box: against the wave-per-item kernels on small /
ragged / full-size sequences (the comparison with the CPU restatement lives in tests/test_gpu_parity_r3.py), host-built vs device-built tables, its
launch time and phase clocks at the bench shape.
    python tools/lane_check.py [--time-only] [--seqs 256]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posediffusion_amd import _lib, synth  # noqa: E402
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg  # noqa: E402
from posediffusion_amd.host import denoiser_state  # noqa: E402

DEV = torch.device("cuda:0")
LANE, NOLANE = _lib.PD_GGS_CFG_LANE_ITEMS, _lib.PD_GGS_CFG_NO_LANE_ITEMS


def rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def ragged_matches(enc, h, w, seed, lo=3, hi=400):
    """Matches with a different count per pair (some pairs missing altogether), pair-grouped like hloc's output."""
    rng = np.random.default_rng(seed)
    md = synth.make_matches(enc, h, w, per_pair=hi, seed=seed)
    key = md["i12"][:, 0] * len(enc) + md["i12"][:, 1]
    keep = np.zeros(len(key), dtype=bool)
    for k in np.unique(key):
        idx = np.nonzero(key == k)[0]
        n = int(rng.integers(lo, hi + 1)) if rng.random() > 0.1 else 0
        keep[idx[:n]] = True
    return {"kp1": md["kp1"][keep], "kp2": md["kp2"][keep], "i12": md["i12"][keep], "img_shape": md["img_shape"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--seqs", type=int, default=256)
    ap.add_argument("--scenes", type=int, default=0, help="distinct scenes at the bench shape (0 = all distinct)")
    ap.add_argument("--per-pair", type=int, default=300, help="matches per frame pair at the bench shape")
    args = ap.parse_args()
    torch.cuda.set_device(DEV)
    diff = synth.make_diffuser(seed=0).to(DEV)
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    ok = True
    if not args.time_only:
        eng = PoseEngine(denoiser_state(diff.model), tables, device=DEV, max_B=8, max_N=32)
        cases = [("N20 x 300 (bench sequence)", 20, 224, lambda e, s: synth.make_matches(e, 224, 224, per_pair=300, seed=s)),
                 ("N6 x 40", 6, 224, lambda e, s: synth.make_matches(e, 224, 224, per_pair=40, seed=s)),
                 ("N12 ragged 3..400", 12, 224, lambda e, s: ragged_matches(e, 224, 224, s)),
                 ("N20 x 7 (odd, tiny)", 20, 224, lambda e, s: synth.make_matches(e, 224, 224, per_pair=7, seed=s)),
                 ("N27 x 64 (351 pairs)", 27, 336, lambda e, s: synth.make_matches(e, 336, 336, per_pair=64, seed=s))]
        for name, N, hw, gen in cases:
            B = 3
            encs = [synth.make_cameras(N, seed=500 + b) for b in range(B)]
            mds = [gen(encs[b], 900 + b) for b in range(B)]
            for b in range(B):
                eng.set_matches(b, mds[b]["kp1"], mds[b]["kp2"], mds[b]["i12"], mds[b]["img_shape"])
            x0 = torch.cat([synth.perturb_pose(encs[b], seed=70 + b) for b in range(B)]).to(DEV)
            res = {}
            for tag, flags in (("wave", NOLANE), ("lane", LANE)):
                loss, grad = eng.ggs_loss_grad(x0, cfg=make_ggs_cfg(reserved=flags))
                eng.check_async()
                o5, st5, _ = eng.ggs_optimize(x0, cfg=make_ggs_cfg(iter_num=5, reserved=flags))
                eng.check_async()
                g, stg = eng.ggs_guide(x0, 3, make_ggs_cfg(synth.GGS_CFG, iter_num=20, reserved=flags))
                eng.check_async()
                res[tag] = (loss.cpu(), grad.cpu(), o5.cpu(), st5.cpu(), g.cpu(), stg.cpu())
            lw, ll = res["wave"], res["lane"]
            nv_equal = torch.equal(lw[0][:, 1], ll[0][:, 1])
            e_loss, e_grad, e_o5, e_g = rel(ll[0][:, 0], lw[0][:, 0]), rel(ll[1], lw[1]), rel(ll[2], lw[2]), rel(ll[4], lw[4])
            e_print = rel(ll[0][:, 2], lw[0][:, 2])
            good = nv_equal and e_loss < 2e-6 and e_grad < 2e-5 and e_o5 < 1e-4 and e_print < 2e-6 and \
                torch.equal(lw[3][:, 1], ll[3][:, 1]) and torch.equal(lw[5][:, :, 1], ll[5][:, :, 1])
            ok &= bool(good)
            print(f"{name:28s} n_valid equal {nv_equal} ({ll[0][:, 1].tolist()}); lane vs wave: loss {e_loss:.1e} print {e_print:.1e} grad {e_grad:.1e} "
                  f"5 it {e_o5:.1e} guide(5x20 it) {e_g:.1e} -> {'OK' if good else 'FAIL'}", flush=True)
        # device-built tables == host-built tables (lane kernel on both)
        N, B = 20, 4
        encs = [synth.make_cameras(N, seed=600 + b) for b in range(B)]
        mds = [ragged_matches(encs[b], 224, 224, 950 + b, lo=100, hi=300) for b in range(B)]
        x0 = torch.cat([synth.perturb_pose(encs[b], seed=80 + b) for b in range(B)]).to(DEV)
        cfg = make_ggs_cfg(synth.GGS_CFG, iter_num=10, reserved=LANE)
        for b in range(B):
            eng.set_matches(b, mds[b]["kp1"], mds[b]["kp2"], mds[b]["i12"], mds[b]["img_shape"])
        g_host, _ = eng.ggs_guide(x0, 3, cfg)
        eng.check_async()
        off = np.cumsum([0] + [len(m["kp1"]) for m in mds])
        kp1 = torch.from_numpy(np.concatenate([m["kp1"] for m in mds])).to(DEV)
        kp2 = torch.from_numpy(np.concatenate([m["kp2"] for m in mds])).to(DEV)
        i12 = torch.from_numpy(np.concatenate([m["i12"] for m in mds])).to(DEV)
        eng.set_matches_async(0, kp1, kp2, i12, off, mds[0]["img_shape"], max_pairs=190, max_matches_per_pair=300)
        g_dev, _ = eng.ggs_guide(x0, 3, cfg)
        eng.check_async()
        same = torch.equal(g_host, g_dev)
        ok &= same
        print(f"device-built lane tables vs host-built: bitwise equal {same} (rel {rel(g_dev, g_host):.1e})", flush=True)
        eng.close()

    # ---- the bench shape: one launch = 700 iterations x `seqs` sequences
    EB = args.seqs
    eng = PoseEngine(denoiser_state(diff.model), tables, device=DEV, max_B=EB, max_N=20)
    n_scenes = args.scenes if args.scenes > 0 else EB
    t0 = time.time()
    scenes = []
    for s in range(n_scenes):
        enc = synth.make_cameras(20, seed=2000 + s)
        scenes.append((enc, synth.make_matches(enc, 224, 224, per_pair=args.per_pair, seed=2000 + s)))
    for b in range(EB):
        md = scenes[b % n_scenes][1]
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    print(f"{EB} sequences ({n_scenes} scenes) uploaded in {time.time() - t0:.1f} s", flush=True)
    x0 = torch.cat([synth.perturb_pose(scenes[b % n_scenes][0], seed=7 + b) for b in range(EB)]).to(DEV)
    outs = {}
    for tag, flags in (("lane-per-item", LANE), ("wave-per-item 12 waves", NOLANE), ("lane-per-item", LANE)):
        cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=flags)
        out, stats = eng.ggs_guide(x0, 3, cfg)
        eng.check_async()
        iters = stats[:, :, 1].sum(dim=1)
        outs[tag] = out.cpu()
        ms = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.ggs_guide(x0, 3, cfg)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        flop = EB * 190.0 * args.per_pair * 700 * 100.0
        print(f"{tag:26s} {EB} sequences (iterations run: {iters.min().item():.0f}..{iters.max().item():.0f}): {np.mean(ms):.2f} ms per launch "
              f"({min(ms):.2f} .. {max(ms):.2f}) = {np.mean(ms) / 700 * 1e3:.2f} us per iteration, "
              f"{flop / np.mean(ms) / 1e9:.1f} TFLOP/s = {flop / np.mean(ms) / 1e9 / 157.3:.3f} of the fp32 ALU roof", flush=True)
    for w in (0, 2, 4):                               # phase clocks of waves on SIMD 0 (two waves), SIMD 2 (one wave) and the last wave
        eng.ggs_prof(1 + w)
        cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=LANE)
        eng.ggs_guide(x0, 3, cfg)
        torch.cuda.synchronize()
        pr = eng.ggs_prof(0)
        print(f"lane kernel phase clocks, wave {w}: pass {pr['P2']:.0f}  barrier wait {pr['xchg']:.0f}  P3a+P3b {pr['P3']:.0f}  P4 {pr['P4']:.0f} ticks per iteration "
              f"({pr['iters']} iterations)", flush=True)
    print(f"700-iteration guide, lane vs wave kernels at {EB} sequences: rel {rel(outs['lane-per-item'], outs['wave-per-item 12 waves']):.2e}")
    print("LANE CHECK", "OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
