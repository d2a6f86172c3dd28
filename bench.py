#!/usr/bin/env python
"""Headline benchmark: BASELINE.json config 2 -- 256 synthetic f32 stereo sources per GPU,
44.1 -> 48 kHz linear resample + low_pass(200) + ordered Mixer sum, 1 Mi-frame blocks.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the fused hot path (rh_rlm_run) over one batch of S x N input frames
already resident in HBM, plus -- for N > 1 ranks -- the RCCL all-reduce of the mixed block
(sources are sharded S per rank: weak scaling).  Rank 0 prints ONE JSON line.

  value      = interleaved INPUT samples consumed per second over all ranks, in Msamples/s
               (S*N*C / t).
  roofline   = algorithmic bytes of one launch (4*S*N*C read + 4*M*C written) / the kernel's mean
               duration, measured with HIP events on the launch stream inside the timed region,
               against the 8 TB/s HBM3E peak.
  cpu_baseline = the restated rodio CPU iterator path (oracle/, "port": the reference is Rust
               and cannot be compiled here) on a bounded sample of the same workload, 1 thread
               (rodio's mixer runs on the single cpal callback thread).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sources", type=int, default=256, help="sources per GPU")
    ap.add_argument("--frames", type=int, default=1 << 20, help="input frames per source")
    ap.add_argument("--span", type=int, default=0, help="current_span_len of the sources (0 = None)")
    ap.add_argument("--freq", type=int, default=200)
    ap.add_argument("--frames-per-lane", type=int, default=0)
    ap.add_argument("--ring-stages", type=int, default=0)
    ap.add_argument("--no-balance", type=int, default=0)
    ap.add_argument("--force-general", type=int, default=0)
    ap.add_argument("--no-autotune", action="store_true", help="keep the cost model's launch geometry")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=1 << 20)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import rodio_amd as rh
    from rodio_amd import _lib
    from rodio_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # RH_BENCH_ONE_DEVICE=1: development aid for a 1-GPU box -- every rank time-shares cuda:0 and the
    # collective goes through gloo (RCCL refuses two ranks on one device).  Not a measurement mode.
    one_dev = os.environ.get("RH_BENCH_ONE_DEVICE") == "1"
    dev = 0 if one_dev else local_rank
    torch.cuda.set_device(dev)
    rh.init(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    S, N, Cn = args.sources, args.frames, 2
    # synthetic 44.1 kHz stereo sources, U(-1,1) / (total sources) so |mix| <= 1
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    data = (torch.rand((S, N * Cn), generator=g, device="cuda", dtype=torch.float32) * 2 - 1) * (1.0 / (S * world))
    pipe = rh.ResampleLowpassMix(44100, 48000, Cn, args.span or None, "low_pass", args.freq, 0.5, max_sources=S,
                                 max_in_frames=N, frames_per_lane=args.frames_per_lane, ring_stages=args.ring_stages, no_balance=args.no_balance, force_general=args.force_general)
    pipe.set_sources([data[s] for s in range(S)])
    M = pipe.out_frames
    tuned = None
    if not (args.no_autotune or args.frames_per_lane or args.ring_stages):
        tuned = pipe.autotune()  # untimed set-up, like a BLAS find step: picks the launch geometry on this GPU
        pipe.late_carries()  # reset the diagnostics counter
    outs = [torch.empty(M * Cn, device="cuda", dtype=torch.float32) for _ in range(2)]
    works = [None, None]
    lib = _lib.lib
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def new_event():
        e = C.c_void_p()
        _lib.check(lib.rh_event_create(C.byref(e)), "rh_event_create")
        return e

    def step(k, ev=None):
        buf = outs[k & 1]
        if works[k & 1] is not None:  # the all-reduce that last used this buffer (stream-level wait)
            works[k & 1].wait()
            works[k & 1] = None
        if ev is not None:
            lib.rh_event_record(ev[0], stream)
        pipe.run(buf)
        if ev is not None:
            lib.rh_event_record(ev[1], stream)
        if world > 1:  # the mixer sum across the source shards: one RCCL all-reduce over xGMI,
            works[k & 1] = D.all_reduce_mix(buf, async_op=True)  # rodio_amd/distributed.py; overlaps step k+1

    def drain():
        for i in (0, 1):
            if works[i] is not None:
                works[i].wait()
                works[i] = None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    pipe.check_status()
    events = [(new_event(), new_event()) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, events[k])
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    pipe.check_status()
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = []
    for a, b in events:
        ms = C.c_float(0)
        _lib.check(lib.rh_event_elapsed_ms(a, b, C.byref(ms)), "rh_event_elapsed_ms")
        kernel_ms.append(ms.value)
    kernel_avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)

    if rank == 0:
        in_samples = S * N * Cn
        alg_bytes = 4 * in_samples + 4 * M * Cn
        achieved = alg_bytes / (kernel_avg_ms * 1e-3) / 1e9 if kernel_avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC-derived bytes per launch, see profiles/README.md
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                g0 = pipe.geometry()
                same_geo = tj.get("geometry", {}).get("frames_per_lane") == g0["frames_per_lane"] and not g0["general_kernel"]  # the ring depth does not change what is fetched
                if tj.get("sources") == S and tj.get("frames") == N and tj.get("span", 0) == args.span and same_geo:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        geo = pipe.geometry()
        ph = pipe.phase_cycles()
        if ph is not None:
            geo["phase_cycles"] = [round(x) for x in ph]
        lc = pipe.late_carries()
        # fast kernel: polls of the end-of-kernel look-back that found a predecessor not finished yet;
        # general kernel: (source, tile) carries that were not published in time
        geo["late_carries_per_launch"] = (lc & 0xffffffff) / max(args.steps + args.warmup, 1)
        if tuned:
            geo["autotuned"] = True
        if lc >> 32:
            geo["empty_polls_per_launch"] = (lc >> 32) / max(args.steps + args.warmup, 1)
        res = {
            "metric": "Msamples/s through resample+low_pass+mix pipeline",
            "value": in_samples * world * args.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{S} f32 stereo sources/GPU x {N} frames, 44.1->48 kHz linear resample + low_pass({args.freq}) + ordered Mixer sum"
                            + (f", span_len={args.span}" if args.span else ", span_len=None")
                            + (f"; {world} ranks, sources sharded {S}/rank, RCCL all-reduce of the mixed block" if world > 1 else ""),
                "sources_per_gpu": S, "in_frames": N, "out_frames": M, "channels": Cn,
                "kernel": "k_rlm_wave" if pipe.geometry()["general_kernel"] else "k_rlm_fast", "geometry": geo,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kernel_avg_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(data, S, min(args.cpu_sample_frames, N), args.span, args.freq)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(data, S, frames, span, freq):
    """The oracle's iterator-pull pipeline (virtual next() per adapter like rodio's Box<dyn Source>),
    one thread, on the first `frames` frames of every source of the benchmark batch."""
    from oracle import rodio_oracle as O

    x = data[:, : frames * 2].cpu().numpy().reshape(S, frames, 2)
    t0 = time.perf_counter()
    n_out = O.pipeline_resample_lowpass_mix(x, 44100, 48000, span if span else O.SPAN_NONE, freq, 0.5, want_output=False)
    dt = time.perf_counter() - t0
    assert n_out > 0
    return {
        "value": S * frames * 2 / dt / 1e6,
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{S} sources x {frames} frames (first {frames / (1 << 20):.3g} of each 1 Mi-frame source), {dt:.1f} s; "
                  f"restated rodio CPU iterator path (not rustc-compiled); host has {os.cpu_count()} logical cores",
    }


if __name__ == "__main__":
    main()
