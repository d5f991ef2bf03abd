"""Timings of the small stateless kernels of SURVEY 8f rows N3 (evaluation metrics) and N4 (image preprocessing), with the
CPU restatement (oracle, torch / numpy) timed beside them.  usage: python tests/perf/aux_bench.py  -> one JSON line"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pd_oracle as O                          # CPU baseline only
from posediffusion_amd import _lib, synth

DEV = torch.device("cuda:0")


def gpu_time(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def cpu_time(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6     # us


def random_rotations(n, seed):
    q = torch.randn(n, 4, generator=torch.Generator().manual_seed(seed))
    return O.quaternion_to_matrix(q / q.norm(dim=-1, keepdim=True))


def main():
    lib = _lib.load()
    stream = torch.cuda.current_stream(DEV).cuda_stream
    out = {}
    # ---- N3: pair errors + summary, alignment
    for B, N in ((8, 20), (1, 200)):
        Rp, Rg = random_rotations(B * N, 1), random_rotations(B * N, 2)
        Tp, Tg = torch.randn(B * N, 3, generator=torch.Generator().manual_seed(3)), torch.randn(B * N, 3, generator=torch.Generator().manual_seed(4))
        d = [t.to(DEV).contiguous() for t in (Rp, Tp, Rg, Tg)]
        total = B * N * (N - 1) // 2
        r, t, summ = torch.empty(total, device=DEV), torch.empty(total, device=DEV), torch.empty(7, device=DEV)

        def gpu():
            _lib.check(lib.pd_metrics_rel_pose_errors(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), B, N,
                                                      r.data_ptr(), t.data_ptr(), stream), "pairs")
            _lib.check(lib.pd_metrics_summary(r.data_ptr(), t.data_ptr(), total, 30, summ.data_ptr(), stream), "summary")

        def cpu():
            rr, tt = O.camera_to_rel_deg(Rp, Tp, Rg, Tg, B)
            O.calculate_auc_np(rr.numpy(), tt.numpy(), 30)

        out[f"metrics_B{B}_N{N}"] = {"pairs": total, "gpu_us": round(gpu_time(gpu), 2), "cpu_us": round(cpu_time(cpu), 1),
                                     "cpu_threads": torch.get_num_threads()}
    N = 20
    Rs, Rt = random_rotations(N, 5), random_rotations(N, 6)
    Ts, Tt = torch.randn(N, 3), torch.randn(N, 3)
    d = [t.to(DEV).contiguous() for t in (Rs, Ts, Rt, Tt)]
    Ro, To = torch.empty(N, 3, 3, device=DEV), torch.empty(N, 3, device=DEV)
    out["align_N20"] = {
        "gpu_us": round(gpu_time(lambda: _lib.check(lib.pd_align_cameras(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), N, 1,
                                                                        1e-9, Ro.data_ptr(), To.data_ptr(), None, stream), "align")), 2),
        "cpu_us": round(cpu_time(lambda: O.corresponding_cameras_alignment(Rs, Ts, Rt, Tt, True)), 1)}
    # ---- N4: one 1080 x 1920 uint8 frame -> 3 x 224 x 224 float (centre crop + bilinear), frame resident in HBM
    H, W, S = 1080, 1920, 224
    frame = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    fd, od = frame.to(DEV), torch.empty(3, S, S, device=DEV)
    us = gpu_time(lambda: _lib.check(lib.pd_preprocess_image(fd.data_ptr(), H, W, S, od.data_ptr(), stream), "prep"), 200)

    def cpu_prep():      # util/load_img_folder.py:58-73 + :35-40 on one decoded frame
        x = torch.from_numpy(frame.numpy().astype(np.float32) / 255.0).permute(2, 0, 1)
        m = min(H, W)
        t0, l0 = (H - m) // 2, (W - m) // 2
        F.interpolate(x[None, :, t0:t0 + m, l0:l0 + m], size=(S, S), mode="bilinear", align_corners=False)

    touched = 3 * S * S * 4 + 4 * 3 * S * S            # output floats + <= 4 source bytes per output value
    out["preprocess_1080p_to_224"] = {"gpu_us": round(us, 2), "cpu_us": round(cpu_time(cpu_prep), 1),
                                      "algorithmic_bytes": touched, "gpu_GBps": round(touched / us / 1e3, 2),
                                      "note": "launch-latency bound: 1.2 MB touched per frame"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
