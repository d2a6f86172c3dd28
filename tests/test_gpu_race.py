"""Deterministic reproducer of the round-1 "wrong block / RH_ERR_TIMEOUT when the GPU is shared" failure.

Root cause: state the fused kernels read and write in HBM (ticket counter, aggregate table, stream filter states, the
echo history) was initialised with hipMemset, which is only ENQUEUED on the null stream.  The C++ shim (and any host
that follows rodio_hip.h) launches on hipStreamNonBlocking streams, which do not wait for the null stream: when the null
stream was slow -- another process holding the GPU -- the fill landed AFTER the first kernels had taken tickets,
published aggregates or stored states, and wiped them: tiles renumbered in mid-launch (frames never written: zeros),
carries that never arrive (timeout), a filter state reset between two blocks (a wrong block).

The test makes the null stream slow on purpose (a thread that keeps large fills queued on it), then creates handles and runs their first blocks
on a non-blocking stream (torch's pool streams are created with cudaStreamNonBlocking).  Against the round-1 library this
fails within a few trials (RODIO_HIP_LIB=variants/r1/librodio_hip.so); with rh::fill_now it cannot.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TRIALS = 12


@pytest.fixture(scope="module")
def G(rh):
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    rh.init(0)
    return rh


class _BusyNullStream:
    """A thread that keeps a few milliseconds of 1 GiB fills queued on the legacy default (null) stream -- its own current
    stream -- for as long as the block runs.  (A queue filled once would be drained by the first synchronous hipMemcpy of
    rh_rlm_create; the fills that matter come after it.)"""

    def __init__(self, big):
        import threading

        self.big, self.stop = big, False
        self.t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import torch

        evs = [torch.cuda.Event(), torch.cuda.Event()]
        k = 0
        while not self.stop:
            for _ in range(8):
                self.big.zero_()
            evs[k & 1].record()
            if k:
                evs[(k - 1) & 1].synchronize()  # at most two batches in flight
            k += 1

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.t.join()


def _stream_job(G, xd, ns, block):
    p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=len(xd), max_in_frames=block + 4096, frames_per_lane=4)
    p.stream_begin()
    outs, a, nmax = [], 0, max(ns)
    while a < nmax:
        b = a + block
        outs.append(p.stream_feed_v([x[2 * min(a, n): 2 * min(b, n)] for x, n in zip(xd, ns)], [n <= b for n in ns]))
        a = b
    return p, outs


def test_first_blocks_on_a_nonblocking_stream_behind_a_busy_null_stream(G, O):
    import torch

    ns = [30000, 21000, 30000, 12345, 26000]
    xs = [(np.random.default_rng(70 + i).uniform(-1, 1, 2 * n) * 0.15).astype(np.float32) for i, n in enumerate(ns)]
    m = O.Mixer(2, 48000)
    for x in xs:
        m.add(O.UniformSourceIterator(O.TestSource(x, 2, 44100), 2, 48000).low_pass(200))
    ref = m.collect()
    xd = [torch.from_numpy(x).cuda() for x in xs]
    big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = []
    with _BusyNullStream(big):
        for trial in range(TRIALS):
            with torch.cuda.stream(side):
                try:
                    p, outs = _stream_job(G, xd, ns, 8192)
                    side.synchronize()
                    p.check_status()
                    got = torch.cat(outs).cpu().numpy()
                    err = float(np.max(np.abs(got - ref))) if len(got) == len(ref) else float("inf")
                    if not err <= 1e-5:
                        bad.append((trial, "mismatch", err, int(np.count_nonzero(got == 0.0))))
                    p.close()
                except G.RhError as e:
                    bad.append((trial, str(e)))
    torch.cuda.synchronize()
    assert not bad, bad


def test_echo_history_fill_cannot_land_on_the_first_block(G, O):
    import torch

    n, rate = 60000, 48000
    x = (np.random.default_rng(5).uniform(-1, 1, n) * 0.5).astype(np.float32)
    ref = O.TestSource(x, 2, rate).reverb(100_000_000, 0.4).collect()
    xd = torch.from_numpy(x).cuda()
    big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with _BusyNullStream(big):
        for trial in range(TRIALS):
            with torch.cuda.stream(side):
                r = G.StreamingReverb(100_000_000, 0.4, rate, 2)
                outs = [r.feed(xd[a: a + 7000]) for a in range(0, n, 7000)]
                outs.append(r.flush())
                side.synchronize()
                got = torch.cat(outs).cpu().numpy()
                r.close()
            assert np.array_equal(got, ref), trial
    torch.cuda.synchronize()
