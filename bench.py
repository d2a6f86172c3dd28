#!/usr/bin/env python
"""Benchmarks of the hot path.  The driver line (no flags) is BASELINE.json config 2:

    python bench.py --gpus N --steps K --warmup W          # 256 sources/GPU x 1 Mi stereo frames, 44.1 -> 48 kHz
                                                           # linear resample + low_pass(200) + ordered Mixer sum

A "step" is one pass of the fused hot path (rh_rlm_run) over one batch of S x N input frames already resident in
HBM, plus -- for N > 1 ranks -- the RCCL all-reduce of the mixed block (sources are sharded S per rank: weak scaling;
SURVEY.md 8(e)).  Rank 0 prints ONE JSON line.

  value        = interleaved INPUT samples consumed per second over all ranks, in Msamples/s (S*N*C / t).
  roofline     = algorithmic bytes of one launch (4*S*N*C read + 4*M*C written) / the kernel's mean duration, measured with
                 HIP events on the launch stream inside the timed region, against the 8 TB/s HBM3E peak.  `traffic` = HBM bytes
                 per launch from the TCC counters, collected live by two `rocprofv3 --pmc` passes of this script (FETCH_SIZE,
                 WRITE_SIZE; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md "HBM") unless RH_BENCH_NO_PMC=1.
  cpu_baseline = the restated rodio CPU iterator path (oracle/, "port": the reference is Rust and cannot be compiled here), one
                 thread, on the WHOLE workload (12-13 s); `all_cores` = the same with the sources sharded over the host's cores
                 (an upper bound: rodio's mixer is single-threaded by construction).
  parity       = every output frame of the timed launch against that oracle run.
  roofline.per_source = the same batch in the same run with every source on its own through the converter and the filter
                 (rh_rlm_set_mix_first(0): the path sources with filters or lengths of their own take): kernel time, fraction,
                 traffic and parity of the GENERAL kernel beside the headline's specialisation.

N > 1: `python bench.py --gpus N` starts its own N ranks (torch.distributed.run on 127.0.0.1) when no launcher has; under the
driver's `python -m torch.distributed.run ... bench.py --gpus N` it is a rank.  The line then also carries `multi_gpu` (ranks
seen, the collective alone through torch.distributed AND through the C ABI's rh_allreduce_sum_f32, overlap, the reduced block
against the ranks' partials), `parity` of the ALL-REDUCED block against the oracle over all S*N sources (every rank runs the
oracle over its shard on its share of the cores), `cpu_baseline` and rank 0's `roofline` with live traffic.  The kernel's tiles go
by ticket there (rh_rlm_set_exclusive(0)): RCCL's kernels share the CUs.

The other configurations (single GPU, one JSON line each; evidence for the numbers in DESIGN.md / README.md):

    python bench.py --config 2span      # config 2 with current_span_len = 32768 (uniform.rs:56-67 restarts the converter)
    python bench.py --config 2mono      # config 2 on MONO sources (the fused kernels' channel count is a template parameter)
    python bench.py --config 3          # reverb(65 536 samples) -> Spatial on 64 sources (rh_reverb_spatial)
    python bench.py --config 5          # i16 -> f32 and 6 -> 2 channels on the music.wav excerpt, tiled
    python bench.py --config ragged     # config 2 with source lengths uniform in [N/2, N]
    python bench.py --config limit      # limiter, 64 streams x 1 Mi stereo frames (and --sources 2048 --frames 32768)
    python bench.py --config agc        # automatic gain control, same shapes
    python bench.py --config biquad     # stand-alone low_pass: mode 1 (time-parallel) and mode 0 (reference order)
    python bench.py --config stream     # the headline workload STREAMED in blocks of --block input frames (rh_rlm_stream_block_v on resident rows)
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def make_sources(S, N, first, total, ch=2):
    """[S, N*ch] f32 on the host: source `first + s` of a `total`-source job, U(-1,1) / total from
    numpy.random.default_rng(1234 + global index) -- BASELINE.md section 3, cfg 2 and cfg 4."""
    import numpy as np

    x = np.empty((S, N * ch), dtype=np.float32)
    for s in range(S):
        x[s] = (np.random.default_rng(1234 + first + s).uniform(-1.0, 1.0, ch * N) * (1.0 / total)).astype(np.float32)
    return x


def events(lib, _lib, n):
    out = []
    for _ in range(n):
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(lib.rh_event_create(C.byref(a)), "rh_event_create")
        _lib.check(lib.rh_event_create(C.byref(b)), "rh_event_create")
        out.append((a, b))
    return out


def elapsed(lib, _lib, evs):
    ms = []
    for a, b in evs:
        v = C.c_float(0)
        _lib.check(lib.rh_event_elapsed_ms(a, b, C.byref(v)), "rh_event_elapsed_ms")
        ms.append(v.value)
    return ms


def pmc_traffic(argv, kernel_like, per_call=False):
    """HBM bytes per launch of the kernels matching `kernel_like`: two rocprofv3 --pmc passes of this script (TCC has 4 slots:
    FETCH_SIZE takes 3, WRITE_SIZE 2), each a short run.  per_call: the operation is several kernels -- the sum over every
    matching dispatch of the child run, divided by the calls the child says it made.  Returns (bytes, detail) or (None, reason)."""
    exe = shutil.which("rocprofv3")
    if not exe or os.environ.get("RH_BENCH_NO_PMC") == "1":
        return None, "rocprofv3 not available" if not exe else "disabled (RH_BENCH_NO_PMC=1)"
    vals = {}
    env = dict(os.environ, RH_BENCH_CHILD="1", TMPDIR="/tmp")
    for k in list(env):  # an N > 1 run profiles rank 0's kernel in a single-process child: nothing of the launcher's environment goes along
        if k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT",
                 "RH_BENCH_ONE_DEVICE", "RH_BENCH_SPAWNED") or k.startswith("TORCHELASTIC_") or k.startswith("NCCL_ASYNC"):
            env.pop(k)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rh_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__)] + argv + ["--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
            for attempt in (0, 1):  # (a profiler pass that dies on its way up is tried once more: seen once in a long session, never twice)
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
                if r.returncode == 0 and dbs:
                    break
                shutil.rmtree(d, ignore_errors=True)
                os.makedirs(d, exist_ok=True)
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): " + r.stderr.decode(errors="replace")[-300:]
            con = sqlite3.connect(dbs[0])
            # the timed launches are the last ones of the run: the geometry autotune's dispatches of other template
            # instances are left out by taking the kernel name of the last dispatch
            likes = kernel_like if isinstance(kernel_like, (list, tuple)) else [kernel_like]
            rows = con.execute("select kernel_name, value from counters_collection where counter_name=? and (" + " or ".join("kernel_name like ?" for _ in likes) + ") order by dispatch_id",
                               (counter, *likes)).fetchall()
            if not rows:
                return None, f"no {kernel_like} dispatch in the {counter} pass"
            if per_call:
                calls = None
                for line in r.stdout.decode(errors="replace").splitlines():
                    if line.startswith("{") and '"child"' in line:
                        calls = json.loads(line).get("calls")
                if not calls:
                    return None, "the profiling child did not report its calls"
                vals[counter] = sum(v for _, v in rows) / calls
            else:
                last = rows[-1][0]
                sel = [v for k, v in rows if k == last][-4:]
                vals[counter] = sum(sel) / len(sel)
        except Exception as e:  # noqa: BLE001 -- the benchmark line must still come out
            return None, f"{counter}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    b = vals["FETCH_SIZE"] * 1024.0 * 2.0 + vals["WRITE_SIZE"] * 1024.0
    return b, {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "fetch_correction": 2.0,
               "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of this script, mean over the last dispatches"}


def cpu_baseline(x, S, N, span, freq, want_all_cores=True, ch=2):
    """The oracle's iterator-pull pipeline (a virtual next() per adapter like rodio's Box<dyn Source>), one thread, the whole
    workload; its output is kept for the parity check.  Then the same with the sources sharded over the cores (first 1/4 of
    every source: a bounded sample)."""
    import numpy as np

    from oracle import rodio_oracle as O

    data = x.reshape(S, N, ch)
    t0 = time.perf_counter()
    ref = O.pipeline_resample_lowpass_mix(data, 44100, 48000, span if span else O.SPAN_NONE, freq, 0.5, want_output=True)
    dt = time.perf_counter() - t0
    res = {
        "value": S * N * ch / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": f"the whole workload: {S} sources x {N} frames, {dt:.1f} s; restated rodio CPU iterator path (not rustc-compiled); host has {os.cpu_count()} logical cores",
    }
    if want_all_cores:
        from concurrent.futures import ThreadPoolExecutor

        cores = min(os.cpu_count() or 1, S)
        n4 = max(N // 4, 1)
        shards = [np.ascontiguousarray(data[i::cores, :n4]) for i in range(cores)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:  # ctypes drops the GIL
            list(ex.map(lambda sh: O.pipeline_resample_lowpass_mix(sh, 44100, 48000, span if span else O.SPAN_NONE, freq, 0.5, want_output=False), shards))
        dm = time.perf_counter() - t0
        res["all_cores"] = {"value": S * n4 * ch / dm / 1e6, "unit": "Msamples/s", "cores": cores,
                            "sample": f"sources sharded over {cores} threads, first {n4} frames of each, {dm:.2f} s; an upper bound (rodio's mixer is single-threaded, stream.rs:538-545)"}
    return res, ref


def f64_truth(host, S, N, Cn, span, freq):
    """The exact response of the chain in f64: the converter and the filter are linear, so the f64 sum of the sources goes through one
    f64 lerp (sample_rate.rs:131-201 per span of min(span, 32768) samples, each span's last frame verbatim) and one f64 recurrence with
    the coefficients rodio computes in f32 (blt.rs:502-544).  What the GPU and the oracle (rodio's own f32 recurrence) are each held against."""
    import numpy as np
    from scipy.signal import lfilter

    from oracle import rodio_oracle as O

    mix = np.zeros((N, Cn), dtype=np.float64)
    data = host.reshape(S, N, Cn)
    for s in range(S):
        mix += data[s]
    F, T = 147, 160  # 44.1 -> 48 kHz reduced (sample_rate.rs:74)
    span_f = N if not span else min(span, 32768) // Cn
    parts = []
    for f0 in range(0, N, span_f):
        xs = mix[f0:f0 + span_f]
        n = len(xs)
        M = (((n - 1) * T) + F - 1) // F + 1 if n > 0 else 0
        m = np.arange(M, dtype=np.int64)
        i = m * F // T
        num = (m * F - i * T).astype(np.float64)
        i1 = np.minimum(i + 1, n - 1)
        y = xs[i] + (xs[i1] - xs[i]) * (num / T)[:, None]
        y[i >= n - 1] = xs[n - 1]
        parts.append(y)
    y = np.concatenate(parts)
    co = O.blt_coeffs("low_pass", freq, 0.5, 48000).astype(np.float64)
    z = lfilter(co[:3], np.concatenate([[1.0], co[3:]]), y, axis=0)
    return z.reshape(-1)


def self_spawn(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, one process per GPU, under
    torch.distributed.run on 127.0.0.1 (the form the driver uses for N > 1, spelled out by us when it does not).  The ranks'
    stdout passes through: rank 0 prints the one JSON line."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RH_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on these hosts
    sys.exit(subprocess.run(cmd, env=env).returncode)


def shard_oracle(host, S, N, Cn, span, freq, threads):
    """The oracle's mix of `host`'s S sources, the sources dealt over `threads` host threads: every thread runs the restated
    rodio pipeline (ordered f32 sum inside) over its sources, the threads' mixes are added in f64.  (ctypes drops the GIL.)"""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    from oracle import rodio_oracle as O

    data = host.reshape(S, N, Cn)
    threads = max(1, min(threads, S))
    shards = [np.ascontiguousarray(data[i::threads]) for i in range(threads)]
    with ThreadPoolExecutor(threads) as ex:
        outs = list(ex.map(lambda sh: O.pipeline_resample_lowpass_mix(sh, 44100, 48000, span if span else O.SPAN_NONE, freq, 0.5, want_output=True), shards))
    ref = np.zeros(outs[0].shape, dtype=np.float64)
    for o in outs:
        ref += o
    return ref


def headline(args, argv):
    import numpy as np
    import torch
    import torch.distributed as dist

    import rodio_amd as rh
    from rodio_amd import _lib
    from rodio_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py --gpus {args.gpus} inside a launcher of {world} rank(s): the two must agree")
    # RH_BENCH_ONE_DEVICE=1: development aid for a 1-GPU box -- every rank time-shares cuda:0 and the
    # collective goes through gloo (RCCL refuses two ranks on one device).  Not a measurement mode.
    one_dev = os.environ.get("RH_BENCH_ONE_DEVICE") == "1"
    if world > 1 and not one_dev and torch.cuda.device_count() < world:
        sys.exit(f"bench.py --gpus {world}: {torch.cuda.device_count()} GPU(s) visible (RH_BENCH_ONE_DEVICE=1 time-shares one device over gloo: a development aid, not a measurement)")
    dev = 0 if one_dev else local_rank
    torch.cuda.set_device(dev)
    rh.init(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    cdev = "cpu" if (world > 1 and one_dev) else "cuda"  # where small tensors of host-side collectives live

    def allreduce_scalar(v, op):
        if world == 1:
            return v
        t = torch.tensor([v], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=op)
        return float(t.item())

    # ---- which collective the timed loop runs (mixer.rs:185-198 completed across the shards) --------------------------------------
    #   native  rh_allreduce_sum_f32 of the C ABI (rh_comm.hip: RCCL dlopen'ed, no PyTorch in the data path) -- what a Rust host calls.
    #           The default whenever RCCL takes the ranks (N > 1 on N devices); `--collective native` at --gpus 1 runs the same code with
    #           a communicator of one rank (the path differs from N > 1 by `nranks` only).
    #   native-reduce  the same through rh_reduce_sum_f32 to rank 0 (north_star's "reduce": only the sink-owning rank needs the mix)
    #   torch   torch.distributed's all_reduce (the cross-check; the only choice when the ranks time-share one device over gloo)
    # torch.distributed stays the control plane either way: rendezvous, the 128-byte id, barriers.
    want = args.collective
    comm = None
    comm_err = None
    if want.startswith("native") or (want == "auto" and world > 1 and not one_dev):
        # Every rank takes the same path: rank 0's failure to make the id travels WITH the broadcast (an exception in front of it would leave the
        # others waiting there), and the ranks agree on the outcome before the first launch -- RCCL not loadable / refusing the ranks on any
        # of them sends all of them to the cross-check path, and the line says so.
        uid = None
        if rank == 0:
            try:
                uid = D.NativeComm.unique_id()
            except Exception as e:  # noqa: BLE001
                comm_err = str(e)[:300]
        box = [(uid, comm_err)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        uid, comm_err = box[0]
        if uid is not None:
            try:
                comm = D.NativeComm(rank, world, uid)
                # ... and one small all-reduce through it before anything is timed: a communicator that initialises but cannot reduce (a
                # transport RCCL picks and the box refuses) shows here, where the ranks can still agree to take the other path
                probe = torch.full((1024,), float(rank + 1), device="cuda", dtype=torch.float32)
                comm.all_reduce(probe)
                torch.cuda.synchronize()
                if abs(float(probe[0].item()) - world * (world + 1) / 2.0) > 1e-3:
                    raise RuntimeError(f"rh_allreduce_sum_f32 probe: {float(probe[0].item())} on rank {rank} of {world}")
            except Exception as e:  # noqa: BLE001
                comm_err = str(e)[:300]
                comm = None
        if world > 1:
            agreed = torch.tensor([1 if comm is not None else 0], device=cdev, dtype=torch.int32)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 0 and comm is not None:
                comm = None
                comm_err = comm_err or "another rank could not join the communicator"
        if comm is None and want.startswith("native") and world == 1:
            sys.exit(f"--collective {want}: {comm_err}")
    reduce_only = comm is not None and want == "native-reduce"
    cstream = torch.cuda.Stream() if comm is not None else None  # the collective's stream: the all-reduce of block k runs beside the kernel of block k+1
    done = [None, None]  # native: the event behind the collective that last used buffer i

    S, N, Cn = args.sources, args.frames, (1 if args.config == "2mono" else 2)
    span = 32768 if args.config == "2span" else args.span
    ragged = args.config == "ragged"
    child = os.environ.get("RH_BENCH_CHILD") == "1"  # a rocprofv3 --pmc pass of ourselves: same launches, no extras
    # sources [rank*S, (rank+1)*S) of a (S*world)-source job: seeds 1234 + global index, scale 1/(S*world) (cfg 2 / cfg 4)
    host = make_sources(S, N, rank * S, S * world, Cn)
    data = torch.from_numpy(host).cuda()
    lens = [N] * S
    if ragged:
        lens = [int(v) for v in np.random.default_rng(1).integers(N // 2, N + 1, S)]
    pipe = rh.ResampleLowpassMix(44100, 48000, Cn, span or None, "low_pass", args.freq, 0.5, max_sources=S,
                                 max_in_frames=N, frames_per_lane=args.frames_per_lane, ring_stages=args.ring_stages, no_balance=args.no_balance, force_general=args.force_general)
    # N > 1: the all-reduce of block k runs on RCCL's stream beside the kernel of block k+1 -- the kernel does not have the CUs to
    # itself, so its tiles go by ticket (rodio_hip.h: rh_rlm_set_exclusive)
    pipe.set_exclusive(world == 1 and not args.shared_device and comm is None)
    if args.per_source:
        pipe.set_mix_first(False)
    pipe.set_sources([data[s, : Cn * lens[s]] for s in range(S)])
    M = pipe.out_frames
    tuned = None
    if not (args.no_autotune or args.frames_per_lane or args.ring_stages):
        tuned = pipe.autotune()  # untimed set-up, like a BLAS find step: picks the launch geometry on this GPU
        pipe.late_carries()  # reset the diagnostics counter
    outs = [torch.empty(M * Cn, device="cuda", dtype=torch.float32) for _ in range(2)]
    works = [None, None]
    lib = _lib.lib
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def step(k, ev=None, reduce=True):
        buf = outs[k & 1]
        if works[k & 1] is not None:  # the all-reduce that last used this buffer (stream-level wait)
            works[k & 1].wait()
            works[k & 1] = None
        if done[k & 1] is not None:
            torch.cuda.current_stream().wait_event(done[k & 1])
            done[k & 1] = None
        if ev is not None:
            lib.rh_event_record(ev[0], stream)
        pipe.run(buf)
        if ev is not None:
            lib.rh_event_record(ev[1], stream)
        if not reduce:
            return
        if comm is not None:  # the C ABI's collective on its own stream, ordered behind the kernel by an event
            e = torch.cuda.Event()
            e.record()
            cstream.wait_event(e)
            if reduce_only:
                comm.reduce(buf, 0, stream=cstream)
            else:
                comm.all_reduce(buf, stream=cstream)
            d_ = torch.cuda.Event()
            d_.record(cstream)
            done[k & 1] = d_
        elif world > 1:  # the mixer sum across the source shards: one RCCL all-reduce over xGMI,
            works[k & 1] = D.all_reduce_mix(buf, async_op=True)  # rodio_amd/distributed.py; overlaps step k+1

    def drain():
        for i in (0, 1):
            if works[i] is not None:
                works[i].wait()
                works[i] = None
            if done[i] is not None:
                torch.cuda.current_stream().wait_event(done[i])
                done[i] = None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    pipe.check_status()
    evs = events(lib, _lib, args.steps)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, evs[k])
    drain()
    fence()
    dt = time.perf_counter() - t0
    pipe.check_status()  # a tile hand-off that timed out fails the run here (and would have poisoned the block)
    if world > 1:
        dt = allreduce_scalar(dt, dist.ReduceOp.MAX)
    kernel_ms = elapsed(lib, _lib, evs)
    kernel_avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
    ms_per_step = dt / args.steps * 1e3
    mixed_last = outs[(args.steps - 1) & 1]  # the last timed step's block: the FULL mix on every rank once its all-reduce has run

    if child:
        if rank == 0:
            emit({"child": True, "kernel_ms": kernel_avg_ms, "calls": args.warmup + args.steps})
        return

    # ---- untimed diagnostics of the N > 1 path: the collective alone, the reduced block against the ranks' partials, the ranks seen --
    multi = None
    local = None
    if world > 1:
        kmax = allreduce_scalar(kernel_avg_ms, dist.ReduceOp.MAX)
        kmin = allreduce_scalar(kernel_avg_ms, dist.ReduceOp.MIN)
        seen = int(round(allreduce_scalar(1.0, dist.ReduceOp.SUM)))
        local = torch.empty_like(outs[0])
        pipe.run(local)  # this rank's partial mix
        torch.cuda.synchronize()
        red = local.clone()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        e0.record()
        for _ in range(reps):
            D.all_reduce_mix(red, async_op=False)
        e1.record()
        torch.cuda.synchronize()
        allreduce_ms = e0.elapsed_time(e1) / reps
        red = local.clone()
        D.all_reduce_mix(red, async_op=False)
        k = min(local.numel(), 1 << 16)  # a slice of every rank's partial, summed in rank order on rank 0
        parts = torch.zeros((world, k), device=local.device, dtype=torch.float32)  # (all-reduce of one-hot rows: gloo has no GPU all_gather)
        parts[rank] = local[:k]
        dist.all_reduce(parts, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        acc = torch.zeros(k, device=local.device, dtype=torch.float32)
        for r_ in range(world):
            acc += parts[r_]
        exposed = max(ms_per_step - kmax, 0.0)
        multi = {"n_ranks_seen": seen, "backend": dist.get_backend(),
                 "collective": {"timed": ("rh_reduce_sum_f32 to rank 0" if reduce_only else "rh_allreduce_sum_f32") + " (C ABI, rh_comm.hip over RCCL) on its own stream, block k beside the kernel of block k+1" if comm is not None
                                else "torch.distributed all_reduce (async_op), block k beside the kernel of block k+1", "requested": want, **({"native_unavailable": comm_err} if comm_err else {})},
                 "kernel_ms_max_over_ranks": kmax, "kernel_ms_min_over_ranks": kmin,
                 "allreduce_ms": allreduce_ms, "allreduce_bytes": int(local.numel() * 4),
                 "exposed_ms_per_step": exposed, "overlap_frac": max(0.0, min(1.0, 1.0 - exposed / allreduce_ms)) if allreduce_ms > 0 else None,
                 "reduce_check": {"samples": k, "max_abs_err_vs_rank_ordered_sum_of_partials": float((red[:k] - acc).abs().max()), "peak": float(acc.abs().max())}}
        # the C ABI's own collective (rh_comm_* over a dlopen'ed RCCL, what a Rust host would call): the same reduction, timed the same way
        if one_dev:
            multi["native_comm"] = {"skipped": "RCCL refuses two ranks on one device (RH_BENCH_ONE_DEVICE=1)"}
        else:
            try:
                if comm is None:
                    box = [D.NativeComm.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    comm = D.NativeComm(rank, world, box[0])
                nat = local.clone()
                comm.all_reduce(nat)
                torch.cuda.synchronize()
                diff = float((nat - red).abs().max())
                dist.barrier()
                e0.record()
                for _ in range(reps):
                    comm.all_reduce(nat)
                e1.record()
                torch.cuda.synchronize()
                multi["native_comm"] = {"allreduce_ms": e0.elapsed_time(e1) / reps, "max_abs_diff_vs_torch_distributed": diff, "entry": "rh_allreduce_sum_f32"}
            except Exception as e:  # noqa: BLE001 -- the line must still come out; the failure is in it
                multi["native_comm"] = {"error": str(e)[:300]}

    # ---- the per-source path in the same run: the same batch with every source on its own through the converter and the filter
    # (rh_rlm_set_mix_first(0): what sources with filters / lengths of their own take) -- the general number beside the headline's
    per_source = None
    if not ragged and not args.per_source and not args.no_per_source and pipe.geometry().get("mix_first"):
        pipe.set_mix_first(False)
        pipe.set_sources([data[s, : Cn * lens[s]] for s in range(S)])
        if not (args.no_autotune or args.frames_per_lane or args.ring_stages):
            pipe.autotune()
        ps_out = torch.empty(M * Cn, device="cuda", dtype=torch.float32)
        for _ in range(args.warmup):
            pipe.run(ps_out)
        pevs = events(lib, _lib, args.steps)
        torch.cuda.synchronize()
        for k in range(args.steps):
            lib.rh_event_record(pevs[k][0], stream)
            pipe.run(ps_out)
            lib.rh_event_record(pevs[k][1], stream)
        torch.cuda.synchronize()
        pipe.check_status()
        pms = elapsed(lib, _lib, pevs)
        pgeo = pipe.geometry()
        per_source = {"kernel_ms": sum(pms) / len(pms), "out": ps_out, "geometry": {k_: pgeo[k_] for k_ in ("frames_per_lane", "ring_stages", "n_tiles", "general_kernel", "mix_first")}}
        pipe.set_mix_first(True)
        pipe.set_sources([data[s, : Cn * lens[s]] for s in range(S)])

    # ---- a filter PER SOURCE in the same run: `mixer.add(a.low_pass(200)); mixer.add(b.high_pass(1000)); ...` (source/mod.rs:686-721) -- the same 256
    # sources dealt over four filters, 64 each (rh_rlm_set_filters: one mix-first launch per class, the classes' mixes added in order of first
    # appearance); parity against the oracle's Mixer over the per-source chains, the classes' oracle mixes computed on four host threads
    per_class = None
    if not ragged and not args.per_source and not args.no_per_class and world == 1 and Cn == 2 and not span and S % 4 == 0 and not child:
        classes = [("low_pass", args.freq), ("low_pass", 1000), ("high_pass", 1000), ("low_pass", 4000)]
        pc = rh.ResampleLowpassMix(44100, 48000, Cn, None, "low_pass", args.freq, 0.5, max_sources=S, max_in_frames=N)
        pc.set_filters([classes[s_ % 4] for s_ in range(S)])
        pc.set_sources([data[s_, : Cn * lens[s_]] for s_ in range(S)])
        pc_out = torch.empty(M * Cn, device="cuda", dtype=torch.float32)
        for _ in range(args.warmup):
            pc.run(pc_out)
        cevs = events(lib, _lib, 1)
        torch.cuda.synchronize()
        lib.rh_event_record(cevs[0][0], stream)
        for k in range(args.steps):
            pc.run(pc_out)
        lib.rh_event_record(cevs[0][1], stream)
        torch.cuda.synchronize()
        pc.check_status()
        cms = elapsed(lib, _lib, cevs)[0] / args.steps
        alg_c = 4 * sum(lens) * Cn + 4 * M * Cn
        per_class = {"classes": [f"{k}({f})" for k, f in classes], "sources_per_class": S // 4, "call_ms": cms, "achieved": alg_c / (cms * 1e-3) / 1e9, "frac": alg_c / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "how": "HIP events around the timed region / steps: one call = ONE launch that walks the classes and adds their mixes (k_rlm_chunk_classes; RH_CLASSES_ONE_BY_ONE=1: a launch per class + rh_mix_sum)"}
        if not args.no_cpu_baseline:
            from concurrent.futures import ThreadPoolExecutor

            from oracle import rodio_oracle as O

            def class_mix(ci):
                m_ = O.Mixer(Cn, 48000)
                for s_ in range(ci, S, 4):
                    u_ = O.UniformSourceIterator(O.TestSource(host[s_], Cn, 44100), Cn, 48000)
                    m_.add(u_.low_pass(classes[ci][1]) if classes[ci][0] == "low_pass" else u_.high_pass(classes[ci][1]))
                return m_.collect()

            with ThreadPoolExecutor(4) as ex:
                mixes = list(ex.map(class_mix, range(4)))
            refc = mixes[0].copy()
            for mx_ in mixes[1:]:
                refc = (refc + mx_).astype(np.float32)
            gotc = pc_out.cpu().numpy()
            if gotc.shape == refc.shape:
                ec = float(np.abs(gotc.astype(np.float64) - refc.astype(np.float64)).max())
                per_class["parity"] = {"max_abs_err": ec, "peak": float(np.abs(refc).max()), "tolerance": 1e-5, "ok": bool(ec <= 1e-5), "frames_compared": int(len(refc) // Cn),
                                       "vs": "the oracle's Mixer over the 256 per-source chains (every source its own filter), the classes' mixes added in the same order"}
            else:
                per_class["parity"] = {"ok": False, "error": f"length {gotc.shape} vs oracle {refc.shape}"}
        pc.close()

    # ---- parity and the CPU baseline ---------------------------------------------------------------------------------------
    in_samples = sum(lens) * Cn
    cores = os.cpu_count() or 1
    base = parity = None
    ref = None
    if not args.no_cpu_baseline and ragged and world == 1:  # sources of different lengths: the oracle's Mixer over per-source chains, one thread, the whole workload
        from oracle import rodio_oracle as O

        t0 = time.perf_counter()
        mx = O.Mixer(Cn, 48000)
        for s_ in range(S):
            mx.add(O.UniformSourceIterator(O.TestSource(host[s_, : Cn * lens[s_]], Cn, 44100), Cn, 48000).low_pass(args.freq))
        ref = mx.collect()
        dtc = time.perf_counter() - t0
        base = {"value": in_samples / dtc / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": f"the whole workload ({S} sources, {in_samples} samples), {dtc:.1f} s; restated rodio CPU iterator path (not rustc-compiled); host has {cores} logical cores"}
        got = mixed_last.cpu().numpy()
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64)) if got.shape == ref.shape else None
        parity = ({"frames_compared": int(len(ref) // Cn), "of": int(M), "max_abs_err": float(d.max()), "peak": float(np.abs(ref).max()), "tolerance": 1e-5, "ok": bool(d.max() <= 1e-5),
                   "vs": "oracle Mixer over the per-source chains, the timed launch's whole block"} if d is not None else {"ok": False, "error": f"length {got.shape} vs oracle {ref.shape}"})
    elif not args.no_cpu_baseline and not ragged and world == 1:
        base, ref = cpu_baseline(host, S, N, span, args.freq, ch=Cn)
        got = mixed_last.cpu().numpy()  # the last timed launch's block
        if got.shape == ref.shape:
            d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
            peak = float(np.abs(ref).max())
            parity = {"frames_compared": int(len(ref) // Cn), "of": int(M), "max_abs_err": float(d.max()), "peak": peak, "rel_to_peak": float(d.max()) / peak if peak else None,
                      "tolerance": 1e-5, "ok": bool(d.max() <= 1e-5), "vs": "oracle (restated rodio CPU path), the timed launch's whole block"}
            if not args.no_unscaled:
                # the inputs are scaled by 1/S so that |mix| <= 1, which makes a bare 1e-5 abs check S times weaker.  So: the SAME workload at
                # amplitude 1 (every source U(-1,1): the inputs multiplied by S = 2^8 in place, exact), one more launch, every frame against the oracle
                # (whose unscaled output is S x its scaled one bit for bit: every operation of the chain is linear and S is a power of two; checked below
                # on a prefix), and both against the exact response of the chain in f64.
                t64 = f64_truth(host, S, N, Cn, span, args.freq)
                sc = float(S * world)
                data.mul_(sc)
                pipe.set_sources([data[s, : Cn * lens[s]] for s in range(S)])
                out_u = torch.empty(M * Cn, device="cuda", dtype=torch.float32)
                pipe.run(out_u)
                torch.cuda.synchronize()
                pipe.check_status()
                got_u = out_u.cpu().numpy().astype(np.float64)
                data.mul_(1.0 / sc)
                pipe.set_sources([data[s, : Cn * lens[s]] for s in range(S)])
                ref_u = ref.astype(np.float64) * sc
                npre = min(N, 1 << 14)
                from oracle import rodio_oracle as O_
                pre = O_.pipeline_resample_lowpass_mix(np.ascontiguousarray(host.reshape(S, N, Cn)[:, :npre]) * np.float32(sc), 44100, 48000, span if span else O_.SPAN_NONE, args.freq, 0.5)
                pre_s = O_.pipeline_resample_lowpass_mix(np.ascontiguousarray(host.reshape(S, N, Cn)[:, :npre]), 44100, 48000, span if span else O_.SPAN_NONE, args.freq, 0.5)
                if got_u.shape == ref_u.shape == t64.shape:
                    du = np.abs(got_u - ref_u)
                    pk = float(np.abs(ref_u).max())
                    parity["vs_f64"] = {"gpu": float(np.abs(got.astype(np.float64) - t64).max()), "oracle": float(np.abs(ref.astype(np.float64) - t64).max()),
                                        "what": "max abs distance from the exact (f64) response of the same chain, all frames: the GPU's time-parallel filter and rodio's own f32 recurrence side by side"}
                    parity["unscaled"] = {"amplitude": 1.0, "frames_compared": int(len(ref) // Cn), "max_abs_err": float(du.max()), "peak": pk, "rel_to_peak": float(du.max()) / pk if pk else None,
                                          "vs_f64": {"gpu": float(np.abs(got_u - t64 * sc).max()), "oracle": float(np.abs(ref_u - t64 * sc).max())},
                                          "oracle_scaling_exact_on_prefix": bool(np.array_equal(pre_s.astype(np.float64) * sc, pre.astype(np.float64))),
                                          "note": f"the same {S} sources at full scale (U(-1,1) each, inputs x {int(sc)} in place), one launch of the same plan; the 1e-5 abs tolerance of north_star is stated for |mix| <= 1 "
                                                  "(the scaled run above); at full scale the mix peaks near 3.5 and the distance to rodio's f32 recurrence scales with it -- both sit at the same distance from the exact response"}
                else:
                    parity["unscaled"] = {"ok": False, "error": f"lengths {got_u.shape} / {ref_u.shape} / {t64.shape}"}
        else:
            parity = {"ok": False, "error": f"length {got.shape} vs oracle {ref.shape}"}
    elif not args.no_cpu_baseline and not ragged:
        # N > 1.  (1) rank 0 times the one-thread oracle on a bounded sample of ITS shard while the other ranks wait (no contention
        # for the cores); (2) every rank runs the oracle over its whole shard on its share of the cores; (3) its own partial mix is
        # compared with that, and the ALL-REDUCED block of the last timed step with the f64 sum of all ranks' oracle mixes.
        if rank == 0:
            from oracle import rodio_oracle as O

            nb = min(S, max(1, args.baseline_sources))
            t0 = time.perf_counter()
            O.pipeline_resample_lowpass_mix(host.reshape(S, N, Cn)[:nb], 44100, 48000, span if span else O.SPAN_NONE, args.freq, 0.5, want_output=False)
            dtc = time.perf_counter() - t0
            base = {"value": nb * N * Cn / dtc / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                    "sample": f"sources 0..{nb - 1} of rank 0's shard ({nb} x {N} frames), one thread, {dtc:.1f} s; restated rodio CPU iterator path (not rustc-compiled); host has {cores} logical cores"}
        dist.barrier()
        threads = max(1, cores // world)
        t0 = time.perf_counter()
        ref_shard = shard_oracle(host, S, N, Cn, span, args.freq, threads)
        dto = time.perf_counter() - t0
        mine = local.cpu().numpy().astype(np.float64)
        err_mine = float(np.abs(mine - ref_shard).max()) if mine.shape == ref_shard.shape else float("inf")
        err_parts = allreduce_scalar(err_mine, dist.ReduceOp.MAX)
        dto_max = allreduce_scalar(dto, dist.ReduceOp.MAX)
        tot = torch.from_numpy(ref_shard.copy()).to(cdev)  # (a copy: on the CPU path the all-reduce below would otherwise sum INTO ref_shard)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        ref = tot.cpu().numpy()
        if rank == 0:
            got = mixed_last.cpu().numpy().astype(np.float64)
            ok_shape = got.shape == ref.shape
            err = float(np.abs(got - ref).max()) if ok_shape else float("inf")
            parity = {"frames_compared": int(len(ref) // Cn) if ok_shape else 0, "of": int(M), "max_abs_err": err, "peak": float(np.abs(ref).max()), "tolerance": 1e-5,
                      "per_rank_partial_max_abs_err": err_parts, "ok": bool(err <= 1e-5 and err_parts <= 1e-5),
                      "vs": f"the all-reduced block of the last timed step against the oracle over all {S * world} sources (every rank its shard on {threads} threads, {dto_max:.1f} s; "
                            f"thread and rank mixes added in f64), and every rank's partial mix against its shard's oracle mix"}
            base["all_cores"] = {"value": S * world * N * Cn / dto_max / 1e6, "unit": "Msamples/s", "cores": threads * world,
                                 "sample": f"the whole job: {S * world} sources sharded over {world} ranks x {threads} threads, {dto_max:.2f} s; an upper bound (rodio's mixer is single-threaded, stream.rs:538-545)"}
    if per_source is not None:
        alg_b = 4 * in_samples + 4 * M * Cn
        ps = {"kernel": "k_rlm_fast (every source through the converter and the filter on its own, merged filter state)", "kernel_ms": per_source["kernel_ms"],
              "achieved": alg_b / (per_source["kernel_ms"] * 1e-3) / 1e9, "frac": alg_b / (per_source["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "geometry": per_source["geometry"]}
        cmp_ref = ref if world == 1 else None
        if world > 1 and not args.no_cpu_baseline:
            cmp_ref = ref_shard
        if cmp_ref is not None:
            gotp = per_source["out"].cpu().numpy().astype(np.float64)
            if gotp.shape == cmp_ref.shape:
                e_ = float(np.abs(gotp - cmp_ref.astype(np.float64)).max())
                ps["parity"] = {"max_abs_err": e_, "tolerance": 1e-5, "ok": bool(e_ <= 1e-5), "frames_compared": int(len(cmp_ref) // Cn),
                                "vs": "the same oracle run" if world == 1 else "this rank's shard of the oracle run"}
            else:
                ps["parity"] = {"ok": False, "error": f"length {gotp.shape} vs oracle {cmp_ref.shape}"}
        per_source = ps

    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # ---- rank 0 alone from here: the counter passes (single-process children on this rank's GPU) and the line ----------------
    alg_bytes = 4 * in_samples + 4 * M * Cn
    achieved = alg_bytes / (kernel_avg_ms * 1e-3) / 1e9 if kernel_avg_ms > 0 else 0.0
    geo = pipe.geometry()
    # the profiling passes repeat this run's launches: same workload, the geometry the autotune kept, no autotune of their own
    child_argv = [a for a in argv]
    child_argv[child_argv.index("--frames-per-lane") + 1] = str(geo["frames_per_lane"])
    child_argv[child_argv.index("--ring-stages") + 1] = str(geo["ring_stages"])
    if world > 1 or args.shared_device or want.startswith("native"):
        child_argv.append("--shared-device")  # tiles by ticket, as in this run
    # (a ragged batch and a mix-first batch are two kernels per launch: the sum over both, per call)
    two = bool(geo["ragged_pair"]) or geo.get("mix_first") == 1
    traffic, traffic_how = pmc_traffic(child_argv, ["%k_rlm%", "%k_mix_%"], per_call=two)
    if per_source is not None:
        pa = [a for a in argv]
        pa[pa.index("--frames-per-lane") + 1] = str(per_source["geometry"]["frames_per_lane"])
        pa[pa.index("--ring-stages") + 1] = str(per_source["geometry"]["ring_stages"])
        per_source["traffic"], per_source["traffic_detail"] = pmc_traffic(pa + ["--per-source"], ["%k_rlm%"], per_call=False)
    ph = pipe.phase_cycles()
    if ph is not None:
        geo["phase_cycles"] = [round(x) for x in ph]
    lc = pipe.late_carries()
    geo["late_carries_per_launch"] = (lc & 0xffffffff) / max(args.steps + args.warmup, 1)
    if tuned:
        geo["autotuned"] = True
    geo["tiles_by"] = "workgroup index (exclusive device)" if (world == 1 and not args.shared_device and want == "auto") else "ticket (the device is shared with the collective's kernels)"
    # ---- roofline.side: BASELINE's other single-GPU configurations in the driver's line: config 3 (reverb + spatial),
    # config 5 (i16 -> f32 + 6 -> 2 channels on music.wav) and the headline workload STREAMED in blocks (what GpuMixer runs), each with its
    # time per call, fraction of the 8 TB/s roofline and parity; bounded (a few seconds each; python bench.py --config {3,5,stream} are the
    # full lines with cpu_baseline and counter traffic).
    side_legs = None
    if world == 1 and args.config == "2" and not child and not args.per_source and not args.no_side and not ragged:
        import copy

        side_legs = {}
        t_side = time.perf_counter()
        try:
            # the stream: the very rows of the headline launch, resident, in blocks of B input frames through rh_rlm_stream_block_v; parity against the
            # oracle's ONE-pass block of the headline's parity leg (`ref`), or -- without it -- against the one-shot launch's block
            for B in (65536, 16384):
                sp = rh.ResampleLowpassMix(44100, 48000, Cn, None, "low_pass", args.freq, 0.5, max_sources=S, max_in_frames=B + 4096)
                sp.set_exclusive(True)
                so = torch.empty(M * Cn + 4096, device="cuda", dtype=torch.float32)
                nb = (N + B - 1) // B
                basep = [data[s_].data_ptr() for s_ in range(S)]
                cache = {}
                got_m = [0]

                def one_stream():
                    sp.stream_begin(keep_history=True)
                    g0 = m_ = 0
                    for k_ in range(nb):
                        hi = min(N, (k_ + 1) * B)
                        if (k_, g0) not in cache:
                            cache[(k_, g0)] = ((C.c_void_p * S)(*[b_ + g0 * 4 * Cn for b_ in basep]), (C.c_uint64 * S)(*([hi - g0] * S)), (C.c_uint8 * S)(*([1 if hi >= N else 0] * S)))
                        ptrs, avail, ended = cache[(k_, g0)]
                        o_, c_ = C.c_uint64(0), C.c_uint64(0)
                        _lib.check(lib.rh_rlm_stream_block_v(sp._h, ptrs, avail, ended, S, C.c_void_p(so.data_ptr() + m_ * 4 * Cn), M + 512 - m_, C.byref(o_), C.byref(c_), stream), "rh_rlm_stream_block_v")
                        m_ += o_.value
                        g0 += c_.value
                    got_m[0] = m_

                one_stream()
                one_stream()
                torch.cuda.synchronize()
                evs_ = events(lib, _lib, 1)
                reps = 5
                lib.rh_event_record(evs_[0][0], stream)
                for _ in range(reps):
                    one_stream()
                lib.rh_event_record(evs_[0][1], stream)
                torch.cuda.synchronize()
                sms = elapsed(lib, _lib, evs_)[0] / reps
                sp.check_status()
                one_l = C.c_uint32(0)
                _lib.check(lib.rh_rlm_stream_one_launch_blocks(sp._h, C.byref(one_l)), "rh_rlm_stream_one_launch_blocks")
                gs = so[: got_m[0] * Cn].cpu().numpy()
                against = ref if ref is not None else mixed_last.cpu().numpy()
                ok_len = gs.shape == against.shape
                es = float(np.abs(gs.astype(np.float64) - against.astype(np.float64)).max()) if ok_len else float("inf")
                side_legs["stream" if B == 65536 else f"stream_{B}"] = {
                    "ms": sms, "frac": alg_bytes / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, "block_frames": B, "blocks": nb, "blocks_in_one_launch": one_l.value, "max_abs_err": es, "parity_ok": bool(es <= 1e-5),
                    "vs": "the oracle's one-pass block (the headline's parity reference)" if ref is not None else "the one-shot launch's block",
                    "what": f"the headline's {S} resident sources through rh_rlm_stream_block_v in blocks of {B} input frames; one call = one whole stream = the headline's bytes"}
                sp.close()
                del so
            try:  # a block of a 5.1 mixer in one launch (rh_wide_mix_block) beside the chain of stand-alone launches it replaces
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                import bench_wide

                wm = bench_wide.measure(16, 16384, 6, 44100, 48000, 30)
                side_legs["wide"] = {"ms": wm["new"], "ms_chain_of_launches": wm["old"], "frac": wm["algorithmic_bytes"] / (wm["new"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "bit_identical_to_chain": wm["bit_identical_to_chain"], "parity_ok": bool(wm["oracle_ok"]) and wm["bit_identical_to_chain"],
                                     "vs": "the oracle's mixer on the block's first 2048 frames, bit for bit; the whole block against the chain of stand-alone launches",
                                     "what": "one block of mixer::mixer(6, 48 kHz): 16 resident 5.1 sources at 44.1 kHz, amplify -> convert -> ordered sum, 16384 frames (a short launch-bound kernel: see ms)"}
            except Exception as e:  # noqa: BLE001
                side_legs["wide"] = {"error": str(e)[:200]}
            try:  # the one-row launches a chain falls back to (SURVEY 8(a) row by row), at a quarter of tools/bench_rows.py's size; the resampler against the oracle
                import bench_rows
                import numpy as np

                names = ["resample_linear ch=2 44100->48000", "resample_linear ch=1 44100->48000", "resample_linear ch=6 44100->48000", "mix_sum S=32", "amplify", "echo_mix D=65536",
                         "channel_volume 2->6", "i16_to_f32", "linear_gain_ramp", "delay D=65536"]
                rr = bench_rows.measure(128, 5, names)
                xs = (np.random.default_rng(77).uniform(-1, 1, 2 * 300_001)).astype(np.float32)
                import rodio_amd as rh_
                from oracle import rodio_oracle as O_

                got_ = rh_.SampleRateConverter(rh_.TestSource(xs, 2, 44100), 44100, 48000, 2).collect()
                ref_ = O_.SampleRateConverter(O_.TestSource(xs, 2, 44100), 44100, 48000, 2).collect()
                side_legs["rows"] = {"frac": {r["row"]: r["frac"] for r in rr}, "ms": {r["row"]: r["ms"] for r in rr},
                                     "parity_ok": bool(got_.shape == ref_.shape and np.array_equal(got_.view(np.uint32), ref_.view(np.uint32))),
                                     "vs": "the oracle's SampleRateConverter on 300 001 stereo frames, bit for bit (the other rows: pytest -m gpu)",
                                     "what": "stand-alone C-ABI entries on 128 MiB of f32 input, HIP events, fraction of 8 TB/s on algorithmic bytes (tools/bench_rows.py; 512 MiB: profiles/r06_rows.txt)"}
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                side_legs["rows"] = {"error": str(e)[:200]}
            for cfg_ in ("3", "5"):
                a2 = copy.copy(args)
                a2.config, a2.steps, a2.warmup, a2.no_cpu_baseline = cfg_, 10, 2, False
                side_legs[cfg_] = side(a2, [], lite=True)
                torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001  (a side leg never takes the headline line down)
            side_legs["error"] = str(e)[:300]
        side_legs["seconds"] = time.perf_counter() - t_side
    kern = "k_rlm_fast+k_rlm_resid" if geo["ragged_pair"] else ("k_rlm_wave" if geo["general_kernel"] else "k_rlm_chunk" if geo.get("mix_first") == 2 else "k_mix_ring+k_rlm_fast" if geo.get("mix_first") else "k_rlm_fast")
    res = {
        "metric": "Msamples/s through resample+low_pass+mix pipeline",
        "value": in_samples * world * args.steps / dt / 1e6,
        "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{S} f32 {'mono' if Cn == 1 else 'stereo'} sources/GPU x {N} frames" + (" (lengths uniform in [N/2, N])" if ragged else "")
                        + f", 44.1->48 kHz linear resample + low_pass({args.freq}) + ordered Mixer sum"
                        + (f", span_len={span}" if span else ", span_len=None")
                        + f"; numpy default_rng(1234+s) U(-1,1)/{S * world}"
                        + (f"; {world} ranks, sources sharded {S}/rank, RCCL all-reduce of the mixed block" if world > 1 else ""),
            "sources_per_gpu": S, "in_frames": N, "out_frames": M, "channels": Cn, "kernel": kern, "geometry": geo,
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_detail": traffic_how, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kernel_avg_ms,
                     "of": "rank 0's kernel (every rank runs the same kernel on its own 256-source shard)" if world > 1 else "the one kernel of a step"},
    }
    if per_source is not None:
        res["roofline"]["per_source"] = per_source
    if per_class is not None:
        res["roofline"]["per_class"] = per_class
    if side_legs:
        res["roofline"]["side"] = side_legs
    if world == 1 and want.startswith("native"):
        res["config"]["collective"] = ("rh_reduce_sum_f32" if reduce_only else "rh_allreduce_sum_f32") + " (C ABI, rh_comm.hip over RCCL) with a communicator of ONE rank in the timed loop, on its own stream: the N > 1 code path, nranks = 1"
    if multi:
        one = args.one_gpu_value
        if one is None and args.one_gpu_line:  # a JSON line (or a file holding one) of the same bench at --gpus 1
            txt = open(args.one_gpu_line).read() if os.path.exists(args.one_gpu_line) else args.one_gpu_line
            one = float(json.loads(txt.strip().splitlines()[-1])["value"])
        if one:
            multi["efficiency_vs_1gpu"] = res["value"] / (world * one)  # weak scaling: N ranks do N times the work of one
            multi["one_gpu_value"] = one
        res["multi_gpu"] = multi
    if base is not None:
        res["cpu_baseline"] = base
    if parity is not None:
        res["parity"] = parity
    emit(res)


def _timed_oracle(fn, units, what, cores=1):
    t0 = time.perf_counter()
    out = fn()
    dt = time.perf_counter() - t0
    return out, {"value": units / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
                 "sample": f"{what}, {dt:.2f} s; restated rodio CPU iterator path (not rustc-compiled); host has {os.cpu_count()} logical cores"}


def _parity(got, ref, tol, what):
    import numpy as np

    if got.shape != ref.shape:
        return {"ok": False, "error": f"length {got.shape} vs oracle {ref.shape}", "vs": what}
    if tol == 0:
        return {"samples_compared": int(ref.size), "bit_exact": bool(np.array_equal(got, ref)), "tolerance": 0, "ok": bool(np.array_equal(got, ref)), "vs": what}
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    return {"samples_compared": int(ref.size), "max_abs_err": float(d.max()) if d.size else 0.0, "peak": float(np.abs(ref).max()) if ref.size else 0.0, "tolerance": tol,
            "ok": bool(d.size == 0 or d.max() <= tol), "vs": what}


def side(args, argv, lite=False):
    """Single-GPU configurations other than the headline: one JSON line with the same fields -- `parity` against the oracle,
    `cpu_baseline` (the oracle timed on a bounded sample of the same workload) and live `roofline.traffic` included."""
    import numpy as np
    import torch

    import rodio_amd as rh
    from rodio_amd import _lib

    torch.cuda.set_device(0)
    rh.init(0)
    lib = _lib.lib
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cfg = args.config
    child = os.environ.get("RH_BENCH_CHILD") == "1"
    kernels = []  # (name, fn, algorithmic bytes per launch, units per launch)
    checks = None  # () -> (parity dict, cpu_baseline dict), run once after the timing
    like = "%"
    per_call = False

    if cfg == "3":
        from oracle import rodio_oracle as O

        S, n, ns = 64, 2 << 20, 682_666_667
        host = np.stack([(np.random.default_rng(5678 + s).uniform(-1, 1, n) * 0.25).astype(np.float32) for s in range(S)])
        x = torch.from_numpy(host).cuda()
        em = [[0.5 + 0.01 * s, 0, 1] for s in range(S)]
        d = rh.delay_samples(ns, 48000, 2)
        out = torch.empty((S, n + d), device="cuda")
        gd = rh.spatial_gains_batch(em, [-1, 0, 0], [1, 0, 0])
        alg = 4 * S * n + 4 * S * (n + d)
        kernels.append(("reverb_spatial", lambda: rh.reverb_spatial_batch(x, 48000, ns, 0.3, None, None, None, out=out, gains_dev=gd), alg, S * n))
        workload = f"reverb({ns} ns = {d} samples, 0.3) -> Spatial on {S} sources x {n} interleaved stereo samples @ 48 kHz, default_rng(5678+s) U(-1,1)*0.25 (BASELINE config 3), one fused launch"
        metric = "Msamples/s through reverb+spatial"
        like = "%reverb_spatial%"

        def checks():
            from concurrent.futures import ThreadPoolExecutor

            def row(s_):
                return O.Spatial(O.TestSource(host[s_], 2, 48000).reverb(ns, 0.3), em[s_], [-1, 0, 0], [1, 0, 0]).collect()

            t0 = time.perf_counter()
            first = [row(s_) for s_ in range(8)]  # the baseline: 8 sources, one thread
            dt = time.perf_counter() - t0
            base = {"value": 8 * n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                    "sample": f"sources 0..7 of the workload (8 x {n} samples through reverb -> Spatial), one thread, {dt:.2f} s; restated rodio CPU iterator path (not rustc-compiled); host has {os.cpu_count()} logical cores"}
            with ThreadPoolExecutor(min(os.cpu_count() or 1, S)) as ex:  # every row of the timed launch (ctypes drops the GIL)
                ref = np.stack(first + list(ex.map(row, range(8, S))))
            return _parity(out.cpu().numpy(), ref, 1e-5, f"oracle Spatial(reverb(x)) chains, all {S} rows of the timed launch"), base
    elif cfg == "5":
        from oracle import rodio_oracle as O

        wav = open(os.path.join(ROOT, "tests", "golden", "music.wav"), "rb").read()  # the reference's assets/music.wav (tests/golden/make_golden.py)
        info = rh.wav_probe(wav)
        raw = np.frombuffer(wav, dtype=np.uint8, count=info["data_bytes"], offset=info["data_offset"])
        pcm = raw.view("<i2")
        assert info["samples"] == 894654 and info["bits_per_sample"] == 16 and not info["is_float"], info
        tile = 1024
        d_raw = torch.from_numpy(np.tile(raw, tile)).cuda()  # the data chunk, tiled x1024 so that a launch is not launch-bound
        n = info["samples"] * tile
        f32 = torch.empty(n + 8, device="cuda", dtype=torch.float32)
        frames6 = n // 6
        out2 = torch.empty(frames6 * 2, device="cuda", dtype=torch.float32)
        m = C.c_uint64(0)
        kernels.append(("wav_decode_i16_to_f32", lambda: _lib.check(lib.rh_wav_decode(C.c_void_p(f32.data_ptr()), C.c_void_p(d_raw.data_ptr()), n, 2, 16, 0, C.byref(m), stream), "rh_wav_decode"), 6 * n, n))
        kernels.append(("channels_6_to_2", lambda: _lib.check(lib.rh_channels_convert(C.c_void_p(out2.data_ptr()), C.c_void_p(f32.data_ptr()), frames6, 6, 2, stream), "rh_channels_convert"), 32 * frames6, frames6 * 6))
        # ... and both in ONE launch (round 6: rh_wav_decode_channels): the decoded block in between -- 4 B a sample written and read again -- never exists
        out2f = torch.empty(frames6 * 2 + 4, device="cuda", dtype=torch.float32)
        mf = C.c_uint64(0)
        kernels.append(("wav_decode_and_channels_6_to_2_one_launch", lambda: _lib.check(lib.rh_wav_decode_channels(C.c_void_p(out2f.data_ptr()), C.c_void_p(d_raw.data_ptr()), frames6 * 6, 6, 16, 0, 2, C.byref(mf), stream),
                                                                                     "rh_wav_decode_channels"), 12 * frames6 + 8 * frames6, frames6 * 6))
        workload = (f"assets/music.wav data chunk ({info['samples']} PCM16 samples, RIFF probed by rh_wav_probe_host) tiled x{tile} = {n} samples: i16 -> f32 on the device "
                    f"(rh_wav_decode), then the f32 stream re-framed as {frames6} frames x 6 ch -> ChannelCountConverter(6 -> 2) (BASELINE config 5)")
        metric = "Msamples/s through i16->f32 DataConverter"
        like = "%k_int_to_f32%"

        def checks():
            ns_ = info["samples"]
            reps = 400  # a measurable sample: the chunk converts in about a millisecond
            ref1, base = _timed_oracle(lambda: [O.convert("i16_to_f32", pcm) for _ in range(reps)][-1], ns_ * reps, f"the whole data chunk ({ns_} samples) i16 -> f32, {reps} passes, one thread")
            ref2 = O.ChannelCountConverter(O.TestSource(ref1[: ns_ // 6 * 6], 6, 44100), 6, 2).collect()
            got1 = f32[:ns_].cpu().numpy()
            got2 = out2[: ns_ // 6 * 2].cpu().numpy()  # the first tile = the untiled chunk: 149 109 frames -> 298 218 samples
            p1 = _parity(got1, ref1, 0, "oracle i16 -> f32 (dasp_sample 0.11.0 restated), the whole data chunk")
            p2 = _parity(got2, ref2, 0, "oracle ChannelCountConverter(6 -> 2), 149 109 frames")
            whole = bool(torch.equal(f32[: ns_ * tile].view(tile, ns_), f32[:ns_].expand(tile, ns_)))  # every tile decoded alike
            fused_same = bool(mf.value == frames6 * 2 and torch.equal(out2f[: frames6 * 2].view(torch.int32), out2.view(torch.int32)))  # the one-launch form: the same bits, all of them
            return {"ok": p1["ok"] and p2["ok"] and whole and fused_same, "i16_to_f32": p1, "channels_6_to_2": p2, "all_tiles_equal": whole, "one_launch_bit_identical": fused_same}, base
    elif cfg in ("limit", "agc", "biquad"):
        from oracle import rodio_oracle as O

        S = args.sources if args.sources != 256 else 64
        n = args.frames
        host = np.stack([(np.random.default_rng(4321 + s).uniform(-1, 1, 2 * n) * 0.9).astype(np.float32) for s in range(S)])
        x = torch.from_numpy(host).cuda()
        out = torch.empty_like(x)
        alg = 8 * S * 2 * n
        if cfg == "limit":
            kernels.append(("limit", lambda: rh.limit_batch(x, 2, 48000, out=out), alg, S * 2 * n))
            like, chain, tol = "%k_limit_scan%", (lambda src: src.limit()), 1e-5
        elif cfg == "agc":
            kernels.append(("agc", lambda: rh.agc_batch(x, 48000, out=out), alg, S * 2 * n))
            like, chain, tol, per_call = "%k_agc%", (lambda src: src.automatic_gain_control()), 1e-5, True
        else:
            co = rh.biquad_coeffs("low_pass", 200, 0.5, 48000)
            outs_b = {}
            out0 = torch.empty_like(x)  # (outputs allocated once: a 512 MiB torch.empty_like per step costs the allocator more than the kernel takes)
            kernels.append(("biquad_time_parallel", lambda: outs_b.__setitem__("o", rh.biquad_batch(x, co, mode=1, out=out)), alg, S * 2 * n))
            kernels.append(("biquad_reference_order", lambda: rh.biquad_batch(x, co, mode=0, out=out0), alg, S * 2 * n))
            like, chain, tol = "%k_biquad_scan%", (lambda src: src.low_pass(200)), 1e-5
        workload = f"{cfg}: {S} stereo streams x {n} frames @ 48 kHz, default settings, default_rng(4321+s) U(-1,1)*0.9, 4 B in + 4 B out per sample"
        metric = f"Msamples/s through {cfg}"

        def checks():
            # streams of the timed launch through the oracle's chain, one thread, until about 5 s of CPU work are spent (at least 3)
            rows_, refs, t0 = [], [], time.perf_counter()
            for r in range(S):
                rows_.append(r)
                refs.append(chain(O.TestSource(host[r], 2, 48000)).collect())
                if len(rows_) >= 3 and time.perf_counter() - t0 > 5.0:
                    break
            dt = time.perf_counter() - t0
            base = {"value": len(rows_) * 2 * n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                    "sample": f"streams 0..{rows_[-1]} of the workload ({len(rows_)} x {2 * n} samples), one thread, {dt:.2f} s; restated rodio CPU iterator path (not rustc-compiled); host has {os.cpu_count()} logical cores"}
            got = (outs_b["o"] if cfg == "biquad" else out)[rows_].cpu().numpy()
            return _parity(got, np.stack(refs), tol, f"oracle chains of streams 0..{rows_[-1]} of the timed launch"), base
    elif cfg == "stream":
        # BLOCK STREAMING: what a drop-in runs.  The cpal callback pulls blocks (src/stream.rs:538-545), GpuMixer answers with rh_rlm_stream_block_v per
        # block.  Here the device side of that path alone: the headline's 256 sources RESIDENT in HBM, streamed in blocks of --block input frames
        # through the C ABI -- every block passes `row + consumed so far` (no staging copy: *consumed_frames is a whole number of 16-byte vectors),
        # the sources run together, so the stream carries their summed state (rh_rlm_stream_keep_history: mix first).  One "step" =
        # one whole stream = the headline's bytes; parity = the concatenated blocks against the one-pass oracle.
        from oracle import rodio_oracle as O

        S, N, B = args.sources, args.frames, args.block
        host = make_sources(S, N, 0, S, 2)
        data = torch.from_numpy(host).cuda()
        # (frames per lane: the library's choice for a handle of this block size, as GpuMixer leaves it -- short runs: a block's one mixed row is a
        # latency chain of a few dozen tiles, profiles/r05_stream_frames_per_lane.txt; a block emits whole 16-byte vectors whatever the run length)
        pipe = rh.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", args.freq, 0.5, max_sources=S, max_in_frames=B + 4096, frames_per_lane=args.frames_per_lane or 0)
        pipe.set_exclusive(True)
        # --overlap: rh_rlm_stream_overlap (the rows are resident and complete before the first call: the promise it asks for) -- a block is launched
        # on the same stream WITHOUT a barrier behind the block in front (hipExtAnyOrderLaunch) and waits for the filter state inside the kernel.
        # Bit-identical, +1-3 % (profiles/r06_bench_stream_*_overlap.json); opt-in because of the promise: the default is one block after the other.
        _lib.check(lib.rh_rlm_stream_overlap(pipe._h, 1 if args.overlap else 0), "rh_rlm_stream_overlap")
        mo = C.c_uint64(0)
        _lib.check(lib.rh_resample_out_frames(N, 44100, 48000, 2, 0, C.byref(mo)), "rh_resample_out_frames")
        M = mo.value
        out = torch.empty(M * 2 + 4096, device="cuda", dtype=torch.float32)
        nblocks = (N + B - 1) // B
        base = [data[s_].data_ptr() for s_ in range(S)]
        plan_cache = {}  # block index -> the ctypes arrays of the call (the stream is deterministic: the same g0 every time)
        emitted = []

        # --short-source F: source 0 has only F x N frames -- the stream leaves the summed state when it ends (a state per source, recovered from
        # the block before) and returns to it once that source has given everything (rh_pipeline_stream.hip); RH_STREAM_NO_REJOIN=1: a state per
        # source from there to the end, as before round 5.  No parity in this mode (tests/test_gpu_mix_first.py has the oracle's comparison).
        N0 = max(4, int(N * args.short_source) // 4 * 4) if args.short_source else N

        def one_stream():
            pipe.stream_begin(keep_history=True)
            g0 = m = 0
            for k in range(nblocks):
                hi = min(N, (k + 1) * B)
                key = (k, g0)
                if key not in plan_cache:
                    av = [max(0, min(hi, N0 if s_ == 0 else N) - g0) for s_ in range(S)]
                    plan_cache[key] = ((C.c_void_p * S)(*[b_ + g0 * 8 for b_ in base]), (C.c_uint64 * S)(*av), (C.c_uint8 * S)(*[1 if hi >= (N0 if s_ == 0 else N) else 0 for s_ in range(S)]))
                ptrs, avail, ended = plan_cache[key]
                o, c = C.c_uint64(0), C.c_uint64(0)
                _lib.check(lib.rh_rlm_stream_block_v(pipe._h, ptrs, avail, ended, S, C.c_void_p(out.data_ptr() + m * 8), M + 512 - m, C.byref(o), C.byref(c), stream), "rh_rlm_stream_block_v")
                m += o.value
                g0 += c.value
            emitted.append(m)

        alg = 4 * S * N * 2 + 4 * M * 2
        kernels.append((f"stream_block_v x {nblocks}", one_stream, alg, S * N * 2))
        workload = (f"block streaming of the headline workload: {S} resident f32 stereo sources x {N} frames in {nblocks} blocks of {B} input frames, 44.1->48 kHz + low_pass({args.freq}) + ordered "
                    f"Mixer sum per block through rh_rlm_stream_block_v (sources that run together: the summed state, mix first); one step = one whole stream")
        metric = "Msamples/s through the block-streamed resample+low_pass+mix pipeline"
        like = ["%k_rlm%", "%k_mix_%"]
        per_call = True

        def checks():
            if args.short_source:
                return {"ok": None, "note": "--short-source: timing only", "stream_stats": dict(zip(("blocks_on_the_summed_state", "blocks_with_per_source_states", "recoveries"), pipe.stream_stats()))}, None
            base_, ref = cpu_baseline(host, S, N, 0, args.freq, want_all_cores=False)
            got = out[: emitted[-1] * 2].cpu().numpy()
            pr = _parity(got, ref, 1e-5, f"oracle (restated rodio CPU path) in ONE pass over the whole sources; the GPU output is the concatenation of {nblocks} streamed blocks")
            pr["stream_stats"] = dict(zip(("blocks_on_the_summed_state", "blocks_with_per_source_states", "recoveries"), pipe.stream_stats()))
            ol = C.c_uint32(0)
            _lib.check(lib.rh_rlm_stream_one_launch_blocks(pipe._h, C.byref(ol)), "rh_rlm_stream_one_launch_blocks")
            pr["stream_stats"]["blocks_in_one_launch"] = ol.value
            pr["stream_stats"]["blocks_side_by_side"] = bool(args.overlap) and not os.environ.get("RH_SBLK_NO_OVERLAP")
            return pr, base_
    else:
        sys.exit(f"unknown --config {cfg}")

    rows = []
    calls = 0
    for name, fn, alg, units in kernels:
        fn()
        calls += 1
        torch.cuda.synchronize()
        slow = name == "biquad_reference_order"
        steps = max(2, args.steps // 10) if slow else args.steps
        for _ in range(0 if slow else args.warmup):
            fn()
            calls += 1
        # One pair of HIP events around the timed region, divided by its launches: the launches run back to back as a caller's would,
        # gaps included.  (A pair per launch measures the same within the spread between boxes.  rocprofv3's trace of the scan kernels
        # reads 7 % lower -- 0.186 / 0.248 ms against 0.20 / 0.27 -- because the profiler serialises launches with idle gaps between
        # them: sustained back-to-back launches run at lower clocks.  The line reports the sustained number.)
        evs = events(lib, _lib, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.rh_event_record(evs[0][0], stream)
        for k in range(steps):
            fn()
        lib.rh_event_record(evs[0][1], stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        calls += steps
        kms = elapsed(lib, _lib, evs)[0] / steps
        rows.append({"kernel": name, "steps": steps, "ms_per_step": dt / steps * 1e3, "kernel_ms": kms, "kernel_ms_how": "HIP events around the timed region / steps", "algorithmic_bytes_per_launch": alg,
                     "achieved_GBps": alg / kms / 1e6, "frac": alg / kms / 1e6 / HBM_PEAK_GBS, "Msamples_per_s": units / (dt / steps) / 1e6})
        if child:
            break  # the counter passes look at the head kernel only
    if child:
        emit({"child": True, "calls": calls})
        return
    rh.async_status()
    head = rows[0]
    if lite:  # a leg of the default line (roofline.side): the timing and the parity, no counter passes, nothing printed
        out_ = {"ms": head["kernel_ms"], "frac": head["frac"], "kernels": [{"kernel": r_["kernel"], "ms": r_["kernel_ms"], "frac": r_["frac"]} for r_ in rows], "workload": workload}
        if checks is not None:
            pr_, _ = checks()
            out_["parity_ok"] = bool(pr_.get("ok"))
            if "max_abs_err" in pr_:
                out_["max_abs_err"] = pr_["max_abs_err"]
            if pr_.get("bit_exact") is not None or "i16_to_f32" in pr_:
                out_["bit_exact"] = bool(pr_.get("ok"))
        if cfg == "5" and len(rows) >= 3:  # the config is a CHAIN (decode, then 6 -> 2): its time as two launches and as the one launch of round 6
            out_["chain_ms_two_launches"] = rows[0]["kernel_ms"] + rows[1]["kernel_ms"]
            out_["chain_ms_one_launch"] = rows[2]["kernel_ms"]
        return out_
    traffic, traffic_how = pmc_traffic(argv, like, per_call)
    res = {"metric": metric, "value": head["Msamples_per_s"], "unit": "Msamples/s", "n_gpus": 1, "steps": head["steps"], "warmup": args.warmup,
           "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i16->f32" if cfg == "5" else "f32", "data": "synthetic" if cfg != "5" else "the reference's assets/music.wav, tiled",
           "config": {"workload": workload, "kernels": rows},
           "roofline": {"bound": "hbm", "achieved": head["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac"], "traffic": traffic, "traffic_detail": traffic_how,
                        "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"], "kernel_ms": head["kernel_ms"]}}
    if not args.no_cpu_baseline and checks is not None:
        res["parity"], res["cpu_baseline"] = checks()
    emit(res)


_REAL_STDOUT = None


def emit(obj):
    """The one JSON line, on the process's real stdout (see main())."""
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2", help="2 (default, the driver line), 2span, 2mono, 3, 5, ragged, limit, agc, biquad")
    ap.add_argument("--sources", type=int, default=256, help="sources per GPU (side configs: streams)")
    ap.add_argument("--frames", type=int, default=1 << 20, help="input frames per source")
    ap.add_argument("--span", type=int, default=0, help="current_span_len of the sources (0 = None)")
    ap.add_argument("--block", type=int, default=65536, help="--config stream: input frames per block")
    ap.add_argument("--overlap", action="store_true", help="--config stream: rh_rlm_stream_overlap (a block launched without a barrier behind the block in front; +1-3 %%)")
    ap.add_argument("--short-source", type=float, default=0.0, help="--config stream: source 0 ends after this fraction of the frames (timing only)")
    ap.add_argument("--freq", type=int, default=200)
    ap.add_argument("--frames-per-lane", type=int, default=0)
    ap.add_argument("--ring-stages", type=int, default=0)
    ap.add_argument("--no-balance", type=int, default=0)
    ap.add_argument("--force-general", type=int, default=0)
    ap.add_argument("--no-autotune", action="store_true", help="keep the cost model's launch geometry")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-source", action="store_true", help="the headline batch on the per-source path (rh_rlm_set_mix_first(0)); the default line carries it as roofline.per_source")
    ap.add_argument("--no-per-source", action="store_true", help="skip the per-source leg of the default line")
    ap.add_argument("--no-per-class", action="store_true", help="skip roofline.per_class (the same sources with a filter per source: four classes of 64)")
    ap.add_argument("--no-side", action="store_true", help="skip roofline.side (configs 3, 5 and the block-streamed headline workload in the default line)")
    ap.add_argument("--no-unscaled", action="store_true", help="skip parity.unscaled / parity.vs_f64 (the same workload at amplitude 1, and both sides against the f64 response)")
    ap.add_argument("--shared-device", action="store_true", help="tiles by ticket although one rank runs (rh_rlm_set_exclusive(0)): what the N > 1 ranks do")
    ap.add_argument("--collective", default="auto", choices=["auto", "native", "native-reduce", "torch"],
                    help="the collective of the timed loop: native = rh_allreduce_sum_f32 of the C ABI (default for N > 1 when RCCL takes the ranks), native-reduce = rh_reduce_sum_f32 to rank 0, torch = torch.distributed")
    ap.add_argument("--baseline-sources", type=int, default=64, help="N > 1: sources of rank 0's shard the one-thread CPU baseline times")
    ap.add_argument("--one-gpu-value", type=float, default=None, help="Msamples/s of this bench at --gpus 1: an N > 1 run then prints multi_gpu.efficiency_vs_1gpu")
    ap.add_argument("--one-gpu-line", default=None, help="... or the 1-GPU JSON line itself / a file that holds it")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_spawn(args)  # (does not return)
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner there from C (whatever NCCL_DEBUG says on this image), and other
    # libraries may: file descriptor 1 is pointed at stderr for the rest of the run and the line goes out through a duplicate of the real one.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    import torch

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.config in ("2", "2span", "2mono", "ragged"):
        # the arguments a profiling child must repeat to launch the same kernels
        argv = ["--config", args.config, "--sources", str(args.sources), "--frames", str(args.frames), "--span", str(args.span), "--freq", str(args.freq),
                "--frames-per-lane", str(args.frames_per_lane), "--ring-stages", str(args.ring_stages), "--no-balance", str(args.no_balance), "--force-general", str(args.force_general)]
        if args.no_autotune:
            argv.append("--no-autotune")
        if args.per_source:
            argv.append("--per-source")
        headline(args, argv)
    else:
        side(args, ["--config", args.config, "--sources", str(args.sources), "--frames", str(args.frames), "--block", str(args.block), "--short-source", str(args.short_source)])


if __name__ == "__main__":
    main()
