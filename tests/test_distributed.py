"""The N > 1 path on CPU: two gloo ranks shard the sources, mix their shard (with the oracle standing
in for the HIP kernel -- tests may use it as the checker's twin), all-reduce, and must agree with the
single-rank mix.  What is under test is rodio_amd/distributed.py: shard ownership, the agreed block
length for ragged shards, and the collective."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_in_order():
    from rodio_amd.distributed import shard, shard_range

    for n in (0, 1, 2, 7, 256, 257, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))  # contiguous, insertion order kept
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard(list("abcdefg"), 1, 3) == ["d", "e"]
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ns, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import rodio_oracle as O
    from rodio_amd import distributed as D

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = len(ns)
        xs = [(np.random.default_rng(50 + s).uniform(-1, 1, 2 * n) / total).astype(np.float32) for s, n in enumerate(ns)]
        lo, hi = D.shard_range(total, rank, world)
        m = O.Mixer(2, 48000)
        for x in xs[lo:hi]:
            m.add(O.UniformSourceIterator(O.TestSource(x, 2, 44100), 2, 48000).low_pass(200))
        part = m.collect()
        frames = D.max_out_frames(len(part) // 2)
        block = torch.zeros(frames * 2, dtype=torch.float32)
        block[: len(part)] = torch.from_numpy(part)
        w = D.all_reduce_mix(block, async_op=True)
        w.wait()
        q.put((rank, lo, hi, block.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ns", [[3000] * 6, [3000, 10, 2999, 0, 4000, 1, 2500]])
def test_two_rank_source_shards_sum_to_the_single_rank_mix(O, ns):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ns, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = len(ns)
    xs = [(np.random.default_rng(50 + s).uniform(-1, 1, 2 * n) / total).astype(np.float32) for s, n in enumerate(ns)]
    m = O.Mixer(2, 48000)
    for x in xs:
        m.add(O.UniformSourceIterator(O.TestSource(x, 2, 44100), 2, 48000).low_pass(200))
    ref = m.collect()
    got.sort(key=lambda t: t[0])
    assert (got[0][1], got[0][2], got[1][2]) == (0, (total + 1) // 2, total)
    for _, _, _, block in got:  # every rank holds the full mix
        assert block.shape == ref.shape
        # the rank sum re-associates the f32 mixer sum: inputs are scaled by 1/total
        assert float(np.max(np.abs(block - ref))) <= 1e-6
    assert np.array_equal(got[0][3], got[1][3])
