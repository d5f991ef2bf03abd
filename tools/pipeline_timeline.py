"""Timeline of the two-stage sampling pipeline: when each batch's unguided / guided half ran.
usage: python tools/pipeline_timeline.py depth slots wgs priority [n_submissions]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state
from posediffusion_amd.pipeline import SamplingPipeline

depth, slots, wgs, prio = (int(a) for a in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 12
nu = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda", 0)
B, N = 8, 20
diff = synth.make_diffuser(seed=0).to(dev)
tables = {k: v for k, v in diff.named_buffers(recurse=False)}
engines = [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N) for _ in range(depth)]
inputs = [bench.make_batch_inputs(engines[j], diff, B, dev, seed0=j * B) for j in range(depth)]
cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=wgs)
pipe = SamplingPipeline(engines, slots, dev, trace=False, unguided_streams=nu)
for j in range(depth):
    pipe.submit(inputs[j][0], inputs[j][1], bench.COND_START, cfg)
pipe.synchronize()
pipe = SamplingPipeline(engines, slots, dev, trace=True, unguided_streams=nu)
for i in range(n):
    j = pipe.next_context()
    pipe.submit(inputs[j][0], inputs[j][1], bench.COND_START, cfg)
tl = pipe.timeline()
print(f"depth={depth} slots={slots} wgs={wgs} prio={prio} unguided_streams={nu}")
for i, (a, b, c, d) in enumerate(tl):
    print(f"  sub {i:2d} ctx {i % depth} slot {i % slots}: U {a:7.1f} -> {b:7.1f} ({b - a:5.1f})   G {c:7.1f} -> {d:7.1f} ({d - c:5.1f})")
print(f"  total {tl[-1][3]:.1f} ms for {n} batches -> {n * B / tl[-1][3] * 1e3:.1f} seq/s")
