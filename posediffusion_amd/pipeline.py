"""Software pipeline over independent batches of sequences on ONE GPU.

The reference samples one sequence at a time (test.py:153-216, demo.py:108).  Its loop has two
very different halves (models/gaussian_diffuser.py:284-300 with the ``t < cond_start_step``
branch at :270):

* the unguided steps t = T-1 .. cond_start_step: 43 small launches per step that need few CUs;
* the guided steps: per step one persistent ``pd_ggs_kernel`` launch that holds
  ``wgs_per_seq`` co-resident workgroups (= CUs) per sequence for milliseconds.

Sequences never interact, so several batches can be in flight, each on its own engine
context (buffers + hipGraphs).  A process gets 4 hardware queues and streams that share one
serialise, so at most 4 streams are used and the pipeline measures which torch streams really
overlap (`pick_concurrent_streams`).  Two ways to use them:

* ``unguided_streams = 0`` (default of bench.py): every stream runs whole passes (unguided half,
  then guided half) of every 4th batch.  `wgs_per_seq(B)` sizes a guided kernel to a quarter of
  the CUs (8 workgroups per sequence for batches of 8, one for batches of 64), so even four guided
  halves at once are co-resident (4 x 64 = 256 CUs) and the denoiser launches of the other batches
  fill whatever is free.  Fewer workgroups per sequence cost less CU time per sequence (the serial
  part of an iteration is replicated on every workgroup): 272 sequences/s with 4 x 8 in flight,
  about 500 with 4 x 64.
* ``unguided_streams = u > 0``: a two-stage pipeline, u streams run unguided halves back to back
  and ``ggs_slots`` streams run guided halves (submission i uses slot i % ggs_slots);
  ``ggs_slots * B * wgs_per_seq`` is sized to leave a quarter of the CUs to the unguided streams.
  The hand-over between the halves is ``hipStreamWaitEvent`` (include/pd_engine.h PD_PHASE_*).

No host thread and no host synchronisation in either mode.  Raising the stream priority of the
guided halves, more than 4 hardware queues, or touching the default stream while a guided half is
queued all cost throughput (profiles/round1_e_pipeline_notes.md).
"""
from __future__ import annotations

import warnings
from typing import List, Optional

import torch

from .engine import PoseEngine, make_ggs_cfg

# two-stage mode: fraction of the CUs the co-resident guided kernels may hold; the rest serve the unguided streams
GGS_CU_FRACTION = 0.75


def _overlaps(a: torch.cuda.Stream, b: torch.cuda.Stream, spin_cycles: int, scratch: torch.Tensor) -> bool:
    """True if work on `b` can overtake a busy `a`, i.e. the two streams do not share a hardware queue."""
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin_cycles)
        ea.record(a)
    with torch.cuda.stream(b):
        scratch.add_(1.0)
        eb.record(b)
    eb.synchronize()
    overtook = not ea.query()
    ea.synchronize()
    return overtook


def pick_concurrent_streams(device: torch.device, n: int, priority: int = 0, candidates: int = 16) -> List[torch.cuda.Stream]:
    """Up to `n` torch streams that pairwise run concurrently.

    HIP multiplexes streams onto a few hardware queues (4 per process by default) and two streams on
    one queue serialise: a guided half (tens of ms of persistent kernels) then blocks whatever shares
    its queue.  Which stream lands on which queue is not specified, so it is measured: a spinning
    kernel on one stream, a trivial one on the other, and the question whether the second finished
    first (both directions)."""
    pool = [torch.cuda.Stream(device=device, priority=priority) for _ in range(candidates)]
    if n <= 1:
        return pool[:1]
    scratch = torch.zeros(64, device=device)
    torch.cuda.synchronize(device)
    # calibrate the spin to ~2 ms
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(pool[0]):
        torch.cuda._sleep(1000)
        e0.record(pool[0])
        torch.cuda._sleep(1_000_000)
        e1.record(pool[0])
    e1.synchronize()
    ms = max(e0.elapsed_time(e1), 1e-3)
    spin = max(int(2.0 / ms * 1_000_000), 10_000)
    chosen = [pool[0]]
    for cand in pool[1:]:
        if len(chosen) == n:
            break
        if any(cand.cuda_stream == c.cuda_stream for c in chosen):
            continue
        if all(_overlaps(c, cand, spin, scratch) and _overlaps(cand, c, spin, scratch) for c in chosen):
            chosen.append(cand)
    torch.cuda.synchronize(device)
    return chosen


class PendingSample:
    """Result of `SamplingPipeline.submit`: tensors are valid once `done` has completed."""

    def __init__(self, pose, process, stats, done: torch.cuda.Event, context: int, stream: torch.cuda.Stream):
        self.pose, self.process, self.stats, self.done, self.context, self.stream = pose, process, stats, done, context, stream

    def wait(self):
        self.done.synchronize()
        return self.pose, self.process, self.stats


class SamplingPipeline:
    def __init__(self, engines: List[PoseEngine], ggs_slots: int, device: torch.device, trace: bool = False,
                 unguided_streams: int = 1):
        if not engines:
            raise ValueError("SamplingPipeline needs at least one engine context")
        if ggs_slots < 1 or ggs_slots > len(engines):
            raise ValueError(f"ggs_slots must be in [1, {len(engines)}] (got {ggs_slots})")
        self.engines = engines
        self.device = device
        self.ggs_slots = ggs_slots
        if len(engines) == 1:
            self.u_streams = [torch.cuda.Stream(device=device)]
            self.g_streams = [self.u_streams[0]]
        else:
            nu = max(0, unguided_streams)
            found = pick_concurrent_streams(device, nu + ggs_slots)
            if len(found) < nu + ggs_slots:
                # e.g. under a profiler that serialises dispatches: same results, less overlap
                warnings.warn(f"SamplingPipeline: only {len(found)} of the {nu + ggs_slots} requested HIP streams run "
                              "concurrently; batches will share streams", RuntimeWarning)
                found = [found[i % len(found)] for i in range(nu + ggs_slots)]
            # guided slots first (they must never share a queue); what is left serves the unguided halves.
            # unguided_streams = 0: every slot stream runs whole passes (its unguided half, then its guided half)
            self.g_streams = found[:ggs_slots]
            self.u_streams = found[ggs_slots:] if nu > 0 else []
        self.whole_pass_streams = not self.u_streams
        if self.whole_pass_streams:
            self.u_streams = list(self.g_streams)
        self.u_stream = self.u_streams[0]
        self._ctx_free: List[Optional[torch.cuda.Event]] = [None] * len(engines)
        self._submitted = 0
        # trace=True: per submission (u_begin, u_end, g_begin, g_end) timing events, see `timeline()`
        self._trace = [] if trace else None

    @property
    def contexts(self) -> int:
        return len(self.engines)

    @property
    def streams(self):
        seen, out = set(), []
        for st in self.u_streams + self.g_streams:
            if st.cuda_stream not in seen:
                seen.add(st.cuda_stream)
                out.append(st)
        return out

    def wgs_per_seq(self, B: int) -> int:
        """Workgroups per sequence so that `ggs_slots` guided kernels of B sequences are co-resident.
        A lone context (nothing to overlap with) returns 0 = the engine's own choice."""
        if self.contexts == 1:
            return 0
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        if not self.whole_pass_streams:
            cus = int(cus * GGS_CU_FRACTION)
        return max(1, cus // (B * self.ggs_slots))

    def make_cfg(self, ggs_cfg: dict, B: int, reserved: int = 0):
        """The GGS configuration of a pass of B sequences: workgroups per sequence from the pipeline shape; where that is ONE the
        lane-per-item kernel is asked for (PD_GGS_CFG_LANE_ITEMS: 20 - 30 % faster than the wave-per-item kernel there; the engine
        falls back to the latter where the lane tables do not exist or do not fit -- more than 24 frames or 512 frame pairs)."""
        from . import _lib
        wgs = self.wgs_per_seq(B)
        if wgs == 1 and not (reserved & _lib.PD_GGS_CFG_NO_LANE_ITEMS):
            reserved |= _lib.PD_GGS_CFG_LANE_ITEMS
        return make_ggs_cfg(ggs_cfg, wgs_per_seq=wgs, reserved=reserved)

    def next_context(self) -> int:
        """The context (engine index) the next `submit` will use; upload that batch's matches there."""
        return self._submitted % self.contexts

    def next_stream(self) -> torch.cuda.Stream:
        """The stream the next `submit` starts on: work enqueued there first (e.g. the image feature extractor that
        produces ``z``) is ordered before the sampler without an event."""
        i = self._submitted
        return self.g_streams[i % self.ggs_slots] if self.whole_pass_streams else self.u_streams[i % len(self.u_streams)]

    def submit(self, z: torch.Tensor, noise: torch.Tensor, cond_start_step: int = 0, ggs_cfg=None,
               use_graph: bool = True, want_process: bool = False,
               inputs_ready: Optional[torch.cuda.Event] = None) -> PendingSample:
        """Enqueue GaussianDiffusion.sample for one batch; returns immediately.

        ``z`` / ``noise`` must be complete when the unguided stream reaches them: pass the event that
        follows their producer as ``inputs_ready`` (or synchronise before submitting).  The pipeline
        deliberately does not touch the caller's current stream: a marker on the default stream can
        sit behind a guided half when the two share a hardware queue, which stalls the whole pipe."""
        i = self._submitted
        j = i % self.contexts
        eng = self.engines[j]
        guided = ggs_cfg is not None and cond_start_step > 0
        us = self.u_streams[i % len(self.u_streams)]
        gs = self.g_streams[i % self.ggs_slots] if guided else us
        if self.whole_pass_streams:
            us = gs
        if inputs_ready is not None:
            us.wait_event(inputs_ready)
        if self._ctx_free[j] is not None:
            us.wait_event(self._ctx_free[j])                       # the context's previous batch has left the engine
        tr = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self._trace is not None else None
        with torch.cuda.stream(us):
            if tr:
                tr[0].record(us)
            if not guided:
                out = eng.sample(z, noise, cond_start_step, ggs_cfg, use_graph=use_graph, want_process=want_process)
            else:
                out = eng.sample(z, noise, cond_start_step, ggs_cfg, use_graph=use_graph, want_process=want_process, phase=1)
        if tr:
            tr[1].record(us)
        if gs is not us:
            ev = torch.cuda.Event()
            ev.record(us)
            gs.wait_event(ev)
        done = torch.cuda.Event()
        with torch.cuda.stream(gs):
            if tr:
                tr[2].record(gs)
            if guided:
                out = eng.sample(z, noise, cond_start_step, ggs_cfg, use_graph=use_graph, want_process=want_process,
                                 phase=2, out=out)
            if tr:
                tr[3].record(gs)
                self._trace.append(tr)
            done.record(gs)
        self._ctx_free[j] = done
        self._submitted += 1
        return PendingSample(out[0], out[1], out[2], done, j, gs)

    def timeline(self):
        """[(u_begin, u_end, g_begin, g_end)] in ms relative to the first submission (trace=True; synchronises)."""
        self.synchronize()
        if not self._trace:
            return []
        t0 = self._trace[0][0]
        return [tuple(t0.elapsed_time(e) for e in tr) for tr in self._trace]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def check_async(self):
        for e in self.engines:
            e.check_async()
