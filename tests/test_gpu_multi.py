"""The N > 1 path with the HIP kernel in it.  The box has one GPU, so:

  * the source shards of two "ranks" run on one device and their partial mixes meet in rh_mix_sum -- the arithmetic of
    the sharded job (shard ownership, the ordered sum of partials) with the real kernel, no collective;
  * `bench.py --gpus 2` runs for real under torch.distributed.run with both ranks time-sharing cuda:0 (RH_BENCH_ONE_DEVICE=1:
    gloo carries the collective because RCCL refuses two ranks on one device): sharding by rank, seeds 1234+s over all
    sources, the all-reduce, the reduced block checked against the ranks' partials, the JSON contract;
  * the native RCCL entry points with two ranks are exercised the first time two devices exist (skipped otherwise).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G(rh):
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    rh.init(0)
    return rh


def test_two_source_shards_on_one_device_sum_to_the_single_rank_mix(G, O):
    import ctypes as C

    import torch

    from rodio_amd import _lib
    from rodio_amd.distributed import shard_range

    S, n = 12, 60000
    xs = [(np.random.default_rng(1234 + s).uniform(-1, 1, 2 * n) / S).astype(np.float32) for s in range(S)]
    dev = [torch.from_numpy(x).cuda() for x in xs]

    def mix(tensors):
        p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=len(tensors), max_in_frames=n)
        p.set_sources(tensors)
        out = p.run().clone()
        p.check_status()
        p.close()
        return out

    whole = mix(dev)
    parts = []
    for r in range(2):
        lo, hi = shard_range(S, r, 2)
        parts.append(mix(dev[lo:hi]))
    assert parts[0].numel() == parts[1].numel() == whole.numel()
    summed = torch.empty_like(whole)
    ptrs = (C.c_void_p * 2)(parts[0].data_ptr(), parts[1].data_ptr())
    starts = (C.c_uint64 * 2)(0, 0)
    lens = (C.c_uint64 * 2)(parts[0].numel(), parts[1].numel())
    _lib.check(_lib.lib.rh_mix_sum(C.c_void_p(summed.data_ptr()), summed.numel(), ptrs, starts, lens, 2, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rh_mix_sum")
    torch.cuda.synchronize()
    # the rank sum re-associates the f32 mixer sum; inputs are scaled by 1/S
    assert float((summed - whole).abs().max()) <= 1e-6
    m = O.Mixer(2, 48000)
    for x in xs:
        m.add(O.UniformSourceIterator(O.TestSource(x, 2, 44100), 2, 48000).low_pass(200))
    assert float(np.max(np.abs(summed.cpu().numpy() - m.collect()))) <= 1e-5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_time_sharing_one_device(G):
    env = dict(os.environ, RH_BENCH_ONE_DEVICE="1", RH_BENCH_NO_PMC="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--sources", "8", "--frames", "65536", "--no-autotune"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "Msamples/s" and d["value"] > 0
    assert "1234+s" in d["config"]["workload"] and "/16" in d["config"]["workload"]  # 8 sources per rank, scaled by the job's 16
    mg = d["multi_gpu"]
    assert mg["allreduce_ms"] > 0 and mg["allreduce_bytes"] == d["config"]["out_frames"] * 2 * 4
    assert mg["reduce_check"]["max_abs_err_vs_rank_ordered_sum_of_partials"] <= 1e-6 * max(mg["reduce_check"]["peak"], 1e-3) + 1e-7
    assert 0.0 <= mg["overlap_frac"] <= 1.0


def test_bench_gpus_2_spawns_its_own_ranks_and_prints_a_complete_line(G):
    """`python bench.py --gpus 2` -- plain, no torch.distributed.run around it -- must start its ranks
    itself, and the N > 1 line must carry parity against the ORACLE (not only against the ranks' own partials), the CPU
    baseline, the roofline of the per-rank kernel with live traffic, and the ranks the communicator saw.  One GPU here, so
    the two ranks time-share it over gloo (RH_BENCH_ONE_DEVICE=1); the tiles of the kernel go by ticket as on the real node."""
    import shutil

    env = dict(os.environ, RH_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RH_BENCH_NO_PMC"):  # (a caller's "no counter passes" is not this test's)
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--sources", "16", "--frames", "262144", "--baseline-sources", "4"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["sources_per_gpu"] == 16
    mg = d["multi_gpu"]
    assert mg["n_ranks_seen"] == 2 and mg["backend"] == "gloo"
    assert "skipped" in mg["native_comm"]  # RCCL refuses two ranks on one device; with two devices the entry holds allreduce_ms
    assert "ticket" in d["config"]["geometry"]["tiles_by"]
    par = d["parity"]
    assert par["ok"] and par["max_abs_err"] <= 1e-5 and par["per_rank_partial_max_abs_err"] <= 1e-5 and par["frames_compared"] == d["config"]["out_frames"]
    assert "oracle over all 32 sources" in par["vs"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["all_cores"]["value"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and rf["kernel_ms"] > 0
    if shutil.which("rocprofv3"):
        assert rf["traffic"] and 0.9 * rf["algorithmic_bytes_per_launch"] < rf["traffic"] < 1.3 * rf["algorithmic_bytes_per_launch"], rf
    ps = rf["per_source"]
    assert ps["parity"]["ok"] and ps["kernel_ms"] > 0 and ps["geometry"]["mix_first"] == 0


def test_tiles_by_ticket_give_the_bits_of_tiles_by_workgroup_index(G):
    """rh_rlm_set_exclusive(0) (what the N > 1 ranks and GpuMixer use: other kernels share the CUs) only changes how a launch
    numbers its tiles: k_rlm_chunk and the one-stream launch of the two-kernel form give the same bits either way."""
    import torch

    S, n = 24, 600_000  # >= 2 x 256 chunks of 1024 frames: k_rlm_chunk applies
    xs = [torch.from_numpy((np.random.default_rng(77 + s).uniform(-1, 1, 2 * n) / S).astype(np.float32)).cuda() for s in range(S)]
    outs = {}
    for excl in (True, False):
        p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=n)
        p.set_exclusive(excl)
        p.set_sources(xs)
        assert p.geometry()["mix_first"] == 2
        for _ in range(3):
            o = p.run().clone()
        p.check_status()
        outs[excl] = o
        p.close()
    assert torch.equal(outs[True], outs[False])


def _comm_worker(rank, uid, q):
    import torch

    sys.path.insert(0, ROOT)
    import rodio_amd as rh
    from rodio_amd.distributed import NativeComm

    torch.cuda.set_device(rank)
    rh.init(rank)
    comm = NativeComm(rank, 2, uid)
    x = torch.full((1 << 20,), float(rank + 1), device="cuda")
    comm.all_reduce(x)
    torch.cuda.synchronize()
    q.put((rank, float(x[0]), float(x[-1])))
    comm.close()


def test_native_rccl_all_reduce_two_ranks():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: rh_allreduce_sum_f32 with nranks > 1 (one GPU here)")
    import torch.multiprocessing as mp

    from rodio_amd.distributed import NativeComm

    uid = NativeComm.unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, uid, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert got == [(0, 3.0, 3.0), (1, 3.0, 3.0)]


def test_native_comm_with_one_rank_is_the_identity(G):
    """The C ABI's collective (rh_comm_*: RCCL dlopen'ed, no PyTorch in the data path -- what a Rust host calls) with a communicator of ONE
    rank: the N > 1 code path of bench.py differs from it by `nranks` only.  All-reduce and reduce-to-root leave the block as it is; on a
    second stream, ordered by events, like the timed loop's."""
    import torch

    from rodio_amd.distributed import NativeComm

    comm = NativeComm(0, 1, NativeComm.unique_id())
    x = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, 1 << 20).astype(np.float32)).cuda()
    want = x.clone()
    comm.all_reduce(x)
    comm.reduce(x, 0)
    side = torch.cuda.Stream()
    e = torch.cuda.Event()
    e.record()
    side.wait_event(e)
    comm.all_reduce(x, stream=side)
    d = torch.cuda.Event()
    d.record(side)
    torch.cuda.current_stream().wait_event(d)
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    with pytest.raises(Exception):
        comm.reduce(x, 1)  # no such root
    comm.close()


def test_bench_native_collective_at_one_gpu():
    """`bench.py --collective native` at --gpus 1: rh_allreduce_sum_f32 in the timed loop (a communicator of one rank), the line names it."""
    import json
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--collective", "native", "--sources", "32", "--frames", "65536", "--steps", "3", "--warmup", "1", "--no-per-source",
                        "--no-unscaled", "--no-autotune"], capture_output=True, text=True, timeout=600, env=dict(os.environ, RH_BENCH_NO_PMC="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # ONE JSON line on stdout (RCCL's version banner is silenced)
    d = json.loads(lines[0])
    assert "rh_allreduce_sum_f32" in d["config"]["collective"] and d["parity"]["ok"]
